// Native composition of the primitives: one AR decode step (eager / hipGraph-captured) and the
// full-sequence transformer forward.  Host code only — every launch goes through the C-ABI
// primitives so that what the tests check piecewise is exactly what the composites run.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "vh_common.h"

static thread_local char g_err[512] = "";

void vh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int g_tuning[VH_TUNE_COUNT] = {0};
int vh_tuning(int knob) { return (knob >= 0 && knob < VH_TUNE_COUNT) ? g_tuning[knob] : 0; }
extern "C" int vh_set_tuning(int knob, int value) {
    VH_REQUIRE(knob >= 0 && knob < VH_TUNE_COUNT, VH_EINVAL, "vh_set_tuning: knob=%d", knob);
    g_tuning[knob] = value;
    return VH_OK;
}

extern "C" const char* vh_last_error(void) { return g_err; }
extern "C" int vh_version(void) { return VH_VERSION; }

#define HIP_TRY(call)                                                             \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        VH_REQUIRE(e_ == hipSuccess, VH_ELAUNCH, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

#define TRY(call)                \
    do {                         \
        int rc_ = (call);        \
        if (rc_ != VH_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// AR decoder
// ---------------------------------------------------------------------------------------------
struct vh_ar_decoder {
    vh_ar_decoder_desc d;
    std::vector<vh_layer> layers;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph_n = nullptr;        // VH_GRAPH_STEPS consecutive steps in one graph (fewer graph launches per generate)
    hipGraphExec_t exec_n = nullptr;
    int ldl = 0;
    // pipelined attention (d.qkv_ll): the attention launches live on their own stream, in their own graphs; the two
    // streams hand data to each other through (value, tag) pairs only — no event between them inside a run of steps
    hipStream_t side = nullptr;          // created on first use, at a priority OTHER than the caller's stream's
    int side_prio = 0;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipGraph_t side_graph = nullptr, side_graph_n = nullptr;
    hipGraphExec_t side_exec = nullptr, side_exec_n = nullptr;
};

enum { PART_ALL = 0, PART_MAIN = 1, PART_SIDE = 2 };   // which launches of a pipelined step decoder_enqueue issues

static int decoder_check(const vh_ar_decoder_desc* d) {
    VH_REQUIRE(d, VH_EINVAL, "vh_ar_decoder: null desc");
    VH_REQUIRE(d->B > 0 && d->B <= 64, VH_EUNSUPPORTED,
               "vh_ar_decoder: B=%d (decode rows per GPU must be 1..64)", d->B);
    VH_REQUIRE(d->n_layers > 0 && d->layers, VH_EINVAL, "vh_ar_decoder: no layers");
    VH_REQUIRE(d->d_model == d->n_heads * VH_HEAD_DIM, VH_EUNSUPPORTED,
               "vh_ar_decoder: d_model=%d != n_heads=%d x 64", d->d_model, d->n_heads);
    VH_REQUIRE(d->dff % 16 == 0 && d->V > 0 && d->S_max > 0 && d->n_split >= 1, VH_EINVAL,
               "vh_ar_decoder: bad dff/V/S_max/n_split");
    VH_REQUIRE(d->proj_w && d->audio_emb && d->audio_pe && d->x && d->q && d->attn && d->hidden &&
                   d->logits && d->cache_len && d->audio_pos && d->eos_count && d->codes,
               VH_EINVAL, "vh_ar_decoder: null buffer in desc");
    VH_REQUIRE(d->n_split == 1 || d->attn_partial, VH_EINVAL, "vh_ar_decoder: n_split>1 needs attn_partial");
    VH_REQUIRE((d->x64 == nullptr) == (d->xmid == nullptr), VH_EINVAL, "vh_ar_decoder: x64 and xmid go together");
    VH_REQUIRE(!d->chain_ws || (d->chain_sync && d->chain_ws_bytes >= vh_decode_chain_ws_bytes(d->B, d->d_model, d->dff)),
               VH_EINVAL, "vh_ar_decoder: chain_ws needs chain_sync and vh_decode_chain_ws_bytes() bytes");
    if (d->x64 || d->xs || d->chain_ws)
        for (int i = 0; i < d->n_layers; ++i)
            VH_REQUIRE(d->layers[i].wqkv_f && d->layers[i].w1_f, VH_EINVAL,
                       "vh_ar_decoder: the fp64 accumulator / two-slab forms need folded weights (layer %d)", i);
    VH_REQUIRE(!d->xs || d->x64 || (d->dff % 2048 == 0 && d->d_model <= 1024 && d->d_model % 128 == 0), VH_EUNSUPPORTED,
               "vh_ar_decoder: the two-slab form needs dim_feedforward %% 2048 == 0 and d_model <= 1024 (dff=%d d=%d)",
               d->dff, d->d_model);
    VH_REQUIRE(d->top_k == 1 || d->temperature > 0.f, VH_EINVAL,
               "vh_ar_decoder: sampling (top_k=%d) needs temperature > 0", d->top_k);
    if (d->qkv_ll) {
        VH_REQUIRE(d->pipe_err && d->attn_ll && d->n_layers >= 2 && d->n_layers <= 64 && d->d_model == 512 &&
                       d->S_max % 32 == 0, VH_EINVAL,
                   "vh_ar_decoder: qkv_ll needs attn_ll, pipe_err, 2 <= n_layers <= 64, d_model == 512 and S_max %% 32 == 0");
        for (int i = 0; i < d->n_layers; ++i)
            VH_REQUIRE(d->layers[i].wqkv_f, VH_EINVAL, "vh_ar_decoder: pipelined attention needs folded weights (layer %d)", i);
    }
    return VH_OK;
}

extern "C" void vh_ar_decoder_destroy(vh_ar_decoder* dec);

extern "C" vh_ar_decoder* vh_ar_decoder_create(const vh_ar_decoder_desc* desc) {
    if (decoder_check(desc) != VH_OK) return nullptr;
    auto* dec = new vh_ar_decoder();
    dec->d = *desc;
    dec->layers.assign(desc->layers, desc->layers + desc->n_layers);
    dec->d.layers = dec->layers.data();
    dec->ldl = (desc->V + 3) & ~3;
    if (dec->d.x64 || dec->d.xs || dec->d.chain_ws) dec->d.qkv_ll = nullptr;
    if (dec->d.qkv_ll) {
        if (hipEventCreateWithFlags(&dec->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&dec->ev_join, hipEventDisableTiming) != hipSuccess) {
            vh_set_error("vh_ar_decoder: could not create the attention stream");
            vh_ar_decoder_destroy(dec);
            return nullptr;
        }
    }
    return dec;
}

extern "C" void vh_ar_decoder_destroy(vh_ar_decoder* dec) {
    if (!dec) return;
    if (dec->exec) (void)hipGraphExecDestroy(dec->exec);
    if (dec->graph) (void)hipGraphDestroy(dec->graph);
    if (dec->exec_n) (void)hipGraphExecDestroy(dec->exec_n);
    if (dec->graph_n) (void)hipGraphDestroy(dec->graph_n);
    if (dec->side_exec) (void)hipGraphExecDestroy(dec->side_exec);
    if (dec->side_graph) (void)hipGraphDestroy(dec->side_graph);
    if (dec->side_exec_n) (void)hipGraphExecDestroy(dec->side_exec_n);
    if (dec->side_graph_n) (void)hipGraphDestroy(dec->side_graph_n);
    if (dec->ev_fork) (void)hipEventDestroy(dec->ev_fork);
    if (dec->ev_join) (void)hipEventDestroy(dec->ev_join);
    delete dec;
}

// One decode step for every row: x (B,d) holds the new token's embedding on entry and the NEXT
// token's embedding on exit.  `ev` (optional) brackets each decode-attention launch.
void vh_internal_attn_decode_events(hipEvent_t start, hipEvent_t stop);   // attention.hip

static int decoder_enqueue(vh_ar_decoder* dec, hipStream_t s, std::vector<hipEvent_t>* ev,
                           std::vector<hipEvent_t>* kev = nullptr, int part = PART_ALL) {
    const vh_ar_decoder_desc& d = dec->d;
    const int B = d.B, D = d.d_model;
    const bool pipe = d.qkv_ll && !ev;           // the attention profile times the stand-alone kernel
    // decode attention of one layer, optionally bracketed by events (vh_ar_decoder_profile_attn)
    auto run_attention = [&](const vh_layer& L) -> int {
        if (!ev)
            return vh_attn_decode(d.q, D, L.kcache, L.vcache, d.attn, D, d.cache_len, 1, B, d.n_heads, d.S_max, d.n_split,
                                  d.attn_partial, s);
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            vh_set_error("vh_ar_decoder: hipEventCreate failed");
            return VH_ELAUNCH;
        }
        hipEvent_t k0 = nullptr, k1 = nullptr;
        if (kev && hipEventCreate(&k0) == hipSuccess && hipEventCreate(&k1) == hipSuccess) {
            vh_internal_attn_decode_events(k0, k1);
            kev->push_back(k0);
            kev->push_back(k1);
        }
        (void)hipEventRecord(e0, s);
        const int arc = vh_attn_decode(d.q, D, L.kcache, L.vcache, d.attn, D, d.cache_len, 1, B, d.n_heads, d.S_max,
                                       d.n_split, d.attn_partial, s);
        (void)hipEventRecord(e1, s);
        vh_internal_attn_decode_events(nullptr, nullptr);
        ev->push_back(e0);
        ev->push_back(e1);
        return arc;
    };
    for (int i = 0; i < d.n_layers; ++i) {
        const vh_layer& L = dec->layers[i];
        // LN1 fused into the QKV GEMM; K/V rows appended at cache_len[b]  (modules.py:146-157,271)
        if (d.chain_ws) {
            // one QKV launch for layer 0, then per layer: attention + ONE persistent launch for the whole GEMM chain
            // (out-projection, LN2 + linear_1 + GELU, linear_2 + residual, then the next layer's LN1 + QKV or the head)
            if (i == 0)
                TRY(vh_linear_qkv_folded(d.x, 0, D, L.wqkv_f, L.qkv_c1, L.qkv_c2, d.q, D, L.kcache, L.vcache,
                                         d.cache_len, B, 1, D, d.n_heads, d.S_max, d.ln_eps, 0, s));
            TRY(run_attention(L));
            const bool last = i + 1 == d.n_layers;
            const vh_layer* nx = last ? nullptr : &dec->layers[i + 1];
            TRY(vh_decode_chain(d.attn, d.x, d.q, L.wo, L.bo, L.w1_f, L.w1_c1, L.w1_c2, L.w2, L.b2,
                                nx ? nx->wqkv_f : nullptr, nx ? nx->qkv_c1 : nullptr, nx ? nx->qkv_c2 : nullptr,
                                nx ? nx->kcache : nullptr, nx ? nx->vcache : nullptr, d.cache_len,
                                last ? d.proj_w : nullptr, d.logits, dec->ldl, d.V, B, D, d.dff, d.n_heads, d.S_max, i,
                                d.ln_eps, d.chain_ws, d.chain_ws_bytes, d.chain_sync, s));
            continue;
        }
        const bool x2 = d.xs && !d.x64;              // residual stream between layers in the two-slab form
        const int64_t ss = (int64_t)B * D;           // slab stride
        if (pipe) {
            // QKV publishes q and the newest K / V row as (value, tag) pairs; the attention launch — normally on the
            // decoder's second stream (PART_SIDE), where it started as soon as the previous layer's attention ended
            // and has requested its first keys since — takes them from there and publishes its output the same way
            // for the out-projection, which is launched right behind the QKV and waits for it with its weights loaded.
            if (part != PART_SIDE)
                TRY(vh_linear_qkv_folded_pipe(d.x, D, L.wqkv_f, L.qkv_c1, L.qkv_c2, L.kcache, L.vcache, d.cache_len, B, D,
                                              d.n_heads, d.S_max, d.ln_eps, d.qkv_ll, i, s));
            if (part != PART_MAIN)
                TRY(vh_attn_decode_pipe(d.qkv_ll, L.kcache, L.vcache, d.attn_ll, d.cache_len, B, d.n_heads, d.S_max, i,
                                        d.pipe_err, s));
            if (part == PART_SIDE) continue;
            TRY(vh_linear_ll_in(d.attn_ll, L.wo, L.bo, d.x, D, d.x, D, B, D, D, d.cache_len, i, d.pipe_err, s));
            TRY(vh_linear_folded(d.x, D, L.w1_f, L.w1_c1, L.w1_c2, nullptr, 0, d.hidden, d.dff, B, d.dff, D,
                                 VH_ACT_GELU_ERF, d.ln_eps, s));
            TRY(vh_linear_ws(d.hidden, d.dff, L.w2, L.b2, d.x, D, d.x, D, B, D, d.dff, VH_ACT_NONE, d.gemm_ws,
                             d.gemm_ws_bytes, s));
            continue;
        } else if (d.x64)
            TRY(vh_linear_qkv_folded(d.x64, 1, D, L.wqkv_f, L.qkv_c1, L.qkv_c2, d.q, D, L.kcache, L.vcache,
                                     d.cache_len, B, 1, D, d.n_heads, d.S_max, d.ln_eps, 0, s));
        else if (x2 && i > 0)                        // layer 0 reads the token embedding (one tensor)
            TRY(vh_linear_qkv_folded(d.xs, 2, D, L.wqkv_f, L.qkv_c1, L.qkv_c2, d.q, D, L.kcache, L.vcache,
                                     d.cache_len, B, 1, D, d.n_heads, d.S_max, d.ln_eps, ss, s));
        else if (L.wqkv_f)
            TRY(vh_linear_qkv_folded(d.x, 0, D, L.wqkv_f, L.qkv_c1, L.qkv_c2, d.q, D, L.kcache, L.vcache,
                                     d.cache_len, B, 1, D, d.n_heads, d.S_max, d.ln_eps, 0, s));
        else
            TRY(vh_linear_qkv(d.x, D, L.wqkv, d.q, D, L.kcache, L.vcache, d.cache_len, B, 1, D, d.n_heads,
                              d.S_max, L.ln1_g, L.ln1_b, nullptr, nullptr, d.ln_eps, s));
        TRY(run_attention(L));
        if (d.x64) {
            // accumulator form: out-proj consumes (reads + clears) the fp64 rows → xmid; LN2 + linear_1
            // on xmid; the K slices of linear_2 add (partials + bias + xmid) back onto the fp64 rows
            TRY(vh_linear_x64(d.attn, 0, D, L.wo, L.bo, d.x64, D, d.xmid, D, B, D, D, s));
            TRY(vh_linear_folded(d.xmid, D, L.w1_f, L.w1_c1, L.w1_c2, nullptr, 0, d.hidden, d.dff, B, d.dff, D,
                                 VH_ACT_GELU_ERF, d.ln_eps, s));
            TRY(vh_linear_acc64(d.hidden, d.dff, L.w2, L.b2, d.xmid, D, d.x64, D, B, D, d.dff, s));
            continue;
        }
        if (x2) {
            // out-proj + bias + residual (two slabs, or the embedding in layer 0) → x; LN2 + linear_1 + GELU;
            // linear_2 as two K slices → the two slabs (slice 0 carries bias + residual x)
            if (i > 0)
                TRY(vh_linear_x2(d.attn, 0, D, L.wo, L.bo, d.xs, 1, D, ss, d.x, D, B, D, D, s));
            else
                TRY(vh_linear(d.attn, D, L.wo, L.bo, d.x, D, d.x, D, B, D, D, VH_ACT_NONE, nullptr, nullptr,
                              nullptr, nullptr, 0.f, s));
            TRY(vh_linear_folded(d.x, D, L.w1_f, L.w1_c1, L.w1_c2, nullptr, 0, d.hidden, d.dff, B, d.dff, D,
                                 VH_ACT_GELU_ERF, d.ln_eps, s));
            TRY(vh_linear_to_x2(d.hidden, d.dff, L.w2, L.b2, d.x, D, d.xs, D, ss, B, D, d.dff, s));
            continue;
        }
        // out-proj + bias + residual (modules.py:171,277)
        TRY(vh_linear(d.attn, D, L.wo, L.bo, d.x, D, d.x, D, B, D, D, VH_ACT_NONE, nullptr, nullptr,
                      nullptr, nullptr, 0.f, s));
        // LN2 fused + linear_1 + exact GELU (modules.py:221,278)
        if (L.w1_f)
            TRY(vh_linear_folded(d.x, D, L.w1_f, L.w1_c1, L.w1_c2, nullptr, 0, d.hidden, d.dff, B, d.dff, D,
                                 VH_ACT_GELU_ERF, d.ln_eps, s));
        else
            TRY(vh_linear(d.x, D, L.w1, L.b1, nullptr, 0, d.hidden, d.dff, B, d.dff, D, VH_ACT_GELU_ERF,
                          L.ln2_g, L.ln2_b, nullptr, nullptr, d.ln_eps, s));
        // linear_2 + bias + residual
        TRY(vh_linear_ws(d.hidden, d.dff, L.w2, L.b2, d.x, D, d.x, D, B, D, d.dff, VH_ACT_NONE, d.gemm_ws,
                         d.gemm_ws_bytes, s));
    }
    if (part == PART_SIDE) return VH_OK;
    // head (no bias, no final norm: valle_ar.py:29,158) then greedy sampling + state update
    if (d.chain_ws) {
        // the last layer's chain launch produced the logits
    } else if (d.x64)
        TRY(vh_linear_x64(d.x64, 1, D, d.proj_w, nullptr, nullptr, 0, d.logits, dec->ldl, B, d.V, D, s));
    else if (d.xs)
        TRY(vh_linear_x2(d.xs, 1, D, d.proj_w, nullptr, nullptr, 0, 0, (int64_t)B * D, d.logits, dec->ldl, B, d.V, D, s));
    else
        TRY(vh_linear(d.x, D, d.proj_w, nullptr, nullptr, 0, d.logits, dec->ldl, B, d.V, D, VH_ACT_NONE,
                      nullptr, nullptr, nullptr, nullptr, 0.f, s));
    float* xf = (d.x64 && !d.chain_ws) ? nullptr : d.x;
    if (d.top_k == 1)
        TRY(vh_greedy_step(d.logits, dec->ldl, d.V, d.eos, d.codes, d.codes_stride, d.eos_count,
                           d.pos_base, d.audio_emb, d.audio_pe, d.audio_pos, d.cache_len, xf, d.chain_ws ? nullptr : d.x64, B, D, s));
    else
        TRY(vh_sample_step(d.logits, dec->ldl, d.V, d.eos, d.top_k, d.top_p, d.temperature, d.seed,
                           d.codes, d.codes_stride, d.eos_count, d.pos_base, d.sum_logprobs, d.audio_emb,
                           d.audio_pe, d.audio_pos, d.cache_len, xf, d.chain_ws ? nullptr : d.x64, B, D, s));
    return VH_OK;
}

static bool two_streams(const vh_ar_decoder* dec) { return dec->d.qkv_ll && vh_tuning(VH_TUNE_PIPE_MODE) != 1; }

// The attention stream joins / leaves the caller's stream around a run of steps (never inside one).
// The two streams wait for each other's DATA, so they must sit on different hardware queues: the runtime multiplexes
// streams of one priority onto a few queues (in order within a queue: a launch waiting for data from a launch behind
// it would never end), but keeps separate queues per priority level — so the attention stream gets a priority the
// caller's stream does not have.
static int pipe_side_stream(vh_ar_decoder* dec, hipStream_t s) {
    int prio = 0, least = 0, greatest = 0;
    HIP_TRY(hipStreamGetPriority(s, &prio));
    if (dec->side) {
        VH_REQUIRE(prio != dec->side_prio, VH_ESTATE,
                   "vh_ar_decoder: the caller's stream changed to priority %d, the attention stream's own", prio);
        return VH_OK;
    }
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    VH_REQUIRE(least != greatest, VH_EUNSUPPORTED, "vh_ar_decoder: pipelined attention needs stream priorities");
    // the LOWER priority where there is a choice ("least"): the attention workgroups that wait for a CU must never
    // hold up the dispatch of the QKV launch the resident ones are waiting for
    dec->side_prio = prio == least ? greatest : least;
    // ONE attention stream per (device, priority) for the life of the process, shared by every decoder: where the
    // runtime puts a stream's hardware queue is decided when the stream is first used and depends on how many queues
    // exist by then — created once, early, it keeps its place; and a generate no longer pays
    // for creating and destroying a stream.  Decoders use it one after the other (single caller thread).
    static std::mutex mu;
    static std::map<std::pair<int, int>, hipStream_t> shared;
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(mu);
    hipStream_t& slot = shared[{device, dec->side_prio}];
    if (!slot) HIP_TRY(hipStreamCreateWithPriority(&slot, hipStreamNonBlocking, dec->side_prio));
    dec->side = slot;
    return VH_OK;
}

static int pipe_fork(vh_ar_decoder* dec, hipStream_t s) {
    TRY(pipe_side_stream(dec, s));
    HIP_TRY(hipEventRecord(dec->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(dec->side, dec->ev_fork, 0));
    return VH_OK;
}
static int pipe_join(vh_ar_decoder* dec, hipStream_t s) {
    HIP_TRY(hipEventRecord(dec->ev_join, dec->side));
    HIP_TRY(hipStreamWaitEvent(s, dec->ev_join, 0));
    return VH_OK;
}

extern "C" int vh_ar_decoder_step(vh_ar_decoder* dec, void* stream) {
    VH_REQUIRE(dec, VH_EINVAL, "vh_ar_decoder_step: null decoder");
    hipStream_t s = (hipStream_t)stream;
    if (!two_streams(dec)) return decoder_enqueue(dec, s, nullptr);
    TRY(pipe_fork(dec, s));
    TRY(decoder_enqueue(dec, dec->side, nullptr, nullptr, PART_SIDE));
    TRY(decoder_enqueue(dec, s, nullptr, nullptr, PART_MAIN));
    return pipe_join(dec, s);
}

#define VH_GRAPH_STEPS 8

static int capture_steps(vh_ar_decoder* dec, hipStream_t s, int n_steps, hipGraph_t* graph, hipGraphExec_t* exec,
                         int part = PART_ALL) {
    if (*exec) { (void)hipGraphExecDestroy(*exec); *exec = nullptr; }
    if (*graph) { (void)hipGraphDestroy(*graph); *graph = nullptr; }
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    VH_REQUIRE(e == hipSuccess, VH_ELAUNCH, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    int rc = VH_OK;
    for (int i = 0; i < n_steps && rc == VH_OK; ++i) rc = decoder_enqueue(dec, s, nullptr, nullptr, part);
    e = hipStreamEndCapture(s, graph);
    if (rc != VH_OK) return rc;
    VH_REQUIRE(e == hipSuccess && *graph, VH_ELAUNCH, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0);
    VH_REQUIRE(e == hipSuccess, VH_ELAUNCH, "hipGraphInstantiate: %s", hipGetErrorString(e));
    return VH_OK;
}

extern "C" int vh_ar_decoder_capture(vh_ar_decoder* dec, void* stream) {
    VH_REQUIRE(dec, VH_EINVAL, "vh_ar_decoder_capture: null decoder");
    VH_REQUIRE(stream, VH_EINVAL, "vh_ar_decoder_capture: capture needs a non-null stream");
    hipStream_t s = (hipStream_t)stream;
    // the step advances device-side state only, so N captured steps are the same step N times: one graph of a
    // single step (remainders) and one of VH_GRAPH_STEPS steps (the bulk: an eighth of the graph launches)
    const bool two = two_streams(dec);
    const bool many = vh_tuning(VH_TUNE_GRAPH_STEPS) != 1;
    TRY(capture_steps(dec, s, 1, &dec->graph, &dec->exec, two ? PART_MAIN : PART_ALL));
    if (many) TRY(capture_steps(dec, s, VH_GRAPH_STEPS, &dec->graph_n, &dec->exec_n, two ? PART_MAIN : PART_ALL));
    if (!two) {
        if (dec->side_exec) { (void)hipGraphExecDestroy(dec->side_exec); dec->side_exec = nullptr; }
        if (dec->side_exec_n) { (void)hipGraphExecDestroy(dec->side_exec_n); dec->side_exec_n = nullptr; }
    }
    if (two) TRY(pipe_side_stream(dec, s));
    if (two) {                       // the attention launches of a step: a second graph, replayed on the second stream
        TRY(capture_steps(dec, dec->side, 1, &dec->side_graph, &dec->side_exec, PART_SIDE));
        if (many) TRY(capture_steps(dec, dec->side, VH_GRAPH_STEPS, &dec->side_graph_n, &dec->side_exec_n, PART_SIDE));
    }
    return VH_OK;
}

extern "C" int vh_ar_decoder_replay(vh_ar_decoder* dec, int n_steps, void* stream) {
    VH_REQUIRE(dec && dec->exec, VH_ESTATE, "vh_ar_decoder_replay: capture first");
    VH_REQUIRE(n_steps >= 0, VH_EINVAL, "vh_ar_decoder_replay: n_steps=%d", n_steps);
    hipStream_t s = (hipStream_t)stream;
    const bool two = dec->side_exec != nullptr;
    if (two && n_steps > 0) TRY(pipe_fork(dec, s));
    int i = 0;
    if (dec->exec_n)
        for (; i + VH_GRAPH_STEPS <= n_steps; i += VH_GRAPH_STEPS) {
            if (two) HIP_TRY(hipGraphLaunch(dec->side_exec_n, dec->side));
            HIP_TRY(hipGraphLaunch(dec->exec_n, s));
        }
    for (; i < n_steps; ++i) {
        if (two) HIP_TRY(hipGraphLaunch(dec->side_exec, dec->side));
        HIP_TRY(hipGraphLaunch(dec->exec, s));
    }
    if (two && n_steps > 0) TRY(pipe_join(dec, s));
    return VH_OK;
}

static float mean_event_pairs(std::vector<hipEvent_t>& ev) {
    double total = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) { total += ms; ++n; }
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear();
    return n ? (float)(total / n) : 0.f;
}

extern "C" int vh_ar_decoder_profile_attn(vh_ar_decoder* dec, int n_steps, void* stream,
                                          float* mean_ms, float* floor_ms, float* kernel_ms) {
    VH_REQUIRE(dec && mean_ms && n_steps > 0, VH_EINVAL, "vh_ar_decoder_profile_attn: bad args");
    hipStream_t s = (hipStream_t)stream;
    std::vector<hipEvent_t> ev, fl, kev;
    int rc = VH_OK;
    for (int i = 0; i < n_steps && rc == VH_OK; ++i) {
        rc = decoder_enqueue(dec, s, &ev, kernel_ms ? &kev : nullptr);
        // measurement floor: the same bracket with nothing inside, once per step
        hipEvent_t f0, f1;
        if (floor_ms && hipEventCreate(&f0) == hipSuccess && hipEventCreate(&f1) == hipSuccess) {
            (void)hipEventRecord(f0, s);
            (void)hipEventRecord(f1, s);
            fl.push_back(f0);
            fl.push_back(f1);
        }
    }
    (void)hipStreamSynchronize(s);
    if (floor_ms) *floor_ms = mean_event_pairs(fl);
    if (kernel_ms) *kernel_ms = mean_event_pairs(kev);
    *mean_ms = mean_event_pairs(ev);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// full-sequence forward (prefill / NAR stage / training forward)
// ---------------------------------------------------------------------------------------------
extern "C" int vh_transformer_forward(const vh_forward_desc* f, void* stream) {
    VH_REQUIRE(f && f->layers && f->x && f->xn && f->q && f->attn && f->hidden, VH_EINVAL,
               "vh_transformer_forward: null pointer in desc");
    VH_REQUIRE(f->B > 0 && f->T > 0 && f->n_layers > 0 && f->S_max >= f->T, VH_EINVAL,
               "vh_transformer_forward: bad dims B=%d T=%d L=%d S_max=%d", f->B, f->T, f->n_layers,
               f->S_max);
    VH_REQUIRE(f->d_model == f->n_heads * VH_HEAD_DIM, VH_EUNSUPPORTED,
               "vh_transformer_forward: d_model=%d != n_heads=%d x 64", f->d_model, f->n_heads);
    const int B = f->B, T = f->T, D = f->d_model, M = B * T;
    for (int i = 0; i < f->n_layers; ++i) {
        const vh_layer& L = f->layers[i];
        const float* ada = f->ada ? f->ada + (int64_t)i * 4 * D : nullptr;
        const float* src = (i == 0 && f->x_in) ? f->x_in : f->x;     // layer 0 may read a caller-owned input
        TRY(vh_layernorm(src, L.ln1_g, L.ln1_b, ada, ada ? ada + D : nullptr, f->xn, M, D, f->ln_eps,
                         stream));
        TRY(vh_linear_qkv(f->xn, D, L.wqkv, f->q, D, L.kcache, L.vcache, nullptr, B, T, D, f->n_heads,
                          f->S_max, nullptr, nullptr, nullptr, nullptr, 0.f, stream));
        TRY(vh_attn_rows(f->q, D, L.kcache, L.vcache, f->attn, D, B, f->n_heads, T, T, f->S_max,
                         f->mode, f->x_len, f->x_len_dev, f->kv_len, f->mask, f->pad, stream));
        TRY(vh_linear_ws(f->attn, D, L.wo, L.bo, src, D, f->x, D, M, D, D, VH_ACT_NONE, f->gemm_ws,
                         f->gemm_ws_bytes, stream));
        TRY(vh_layernorm(f->x, L.ln2_g, L.ln2_b, ada ? ada + 2 * D : nullptr, ada ? ada + 3 * D : nullptr,
                         f->xn, M, D, f->ln_eps, stream));
        TRY(vh_linear_ws(f->xn, D, L.w1, L.b1, nullptr, 0, f->hidden, f->dff, M, f->dff, D, VH_ACT_GELU_ERF,
                         f->gemm_ws, f->gemm_ws_bytes, stream));
        TRY(vh_linear_ws(f->hidden, f->dff, L.w2, L.b2, f->x, D, f->x, D, M, D, f->dff, VH_ACT_NONE, f->gemm_ws,
                         f->gemm_ws_bytes, stream));
    }
    return VH_OK;
}
