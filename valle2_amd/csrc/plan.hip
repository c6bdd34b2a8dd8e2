// Native composition of the primitives: one AR decode step (eager / hipGraph-captured) and the
// full-sequence transformer forward.  Host code only — every launch goes through the C-ABI
// primitives so that what the tests check piecewise is exactly what the composites run.
#include <stdarg.h>

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
extern "C" int vh_h16_format(void) { return VH_H16_IS_BF16; }     // 0 = IEEE fp16 (default), 1 = bf16 (-DVH_PERF_BF16)
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
};

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
    VH_REQUIRE(d->prefix_len >= 0, VH_EINVAL, "vh_ar_decoder: prefix_len=%d", d->prefix_len);
    if (d->ffn_ws) {
        VH_REQUIRE(d->ffn_ws_bytes >= vh_ffn_decode_ws_bytes(d->B, d->d_model, d->dff) && d->ffn_ws_bytes > 0, VH_EINVAL,
                   "vh_ar_decoder: ffn_ws needs vh_ffn_decode_ws_bytes() bytes");
        for (int i = 0; i < d->n_layers; ++i)
            VH_REQUIRE(d->layers[i].w1_f && d->layers[i].w1_c1 && d->layers[i].w1_c2, VH_EINVAL,
                       "vh_ar_decoder: ffn_ws needs the folded linear_1 weights (layer %d)", i);
    }
    if (d->kv_bf16) {
        VH_REQUIRE(d->n_split == 1, VH_EUNSUPPORTED, "vh_ar_decoder: the bf16 K/V cache has no key-split form (n_split=%d)",
                   d->n_split);
        for (int i = 0; i < d->n_layers; ++i)
            VH_REQUIRE(d->layers[i].wqkv_f, VH_EINVAL, "vh_ar_decoder: the bf16 K/V cache needs folded weights (layer %d)", i);
    }
    if (d->prefix_len > 0) {
        VH_REQUIRE(!d->kv_bf16, VH_EUNSUPPORTED, "vh_ar_decoder: the shared prompt has no bf16 form");
        VH_REQUIRE((d->prefix_len + 31) / 32 + d->n_split <= 256, VH_EUNSUPPORTED,
                   "vh_ar_decoder: a shared prompt of %d keys with %d suffix splits exceeds the 256 records vh_attn_decode_shared merges",
                   d->prefix_len, d->n_split);
        VH_REQUIRE(d->prefix_len <= d->prefix_S && d->attn_partial &&
                       d->attn_partial_bytes >= vh_attn_decode_shared_ws_bytes(d->B, d->n_heads, d->prefix_len, d->n_split),
                   VH_EINVAL, "vh_ar_decoder: shared prompt of %d / %d keys needs attn_partial of %zu bytes (got %zu)",
                   d->prefix_len, d->prefix_S, vh_attn_decode_shared_ws_bytes(d->B, d->n_heads, d->prefix_len, d->n_split),
                   d->attn_partial_bytes);
        for (int i = 0; i < d->n_layers; ++i)
            VH_REQUIRE(d->layers[i].kprefix && d->layers[i].vprefix, VH_EINVAL,
                       "vh_ar_decoder: shared prompt needs kprefix / vprefix in every layer (layer %d)", i);
    }
    VH_REQUIRE(d->top_k == 1 || d->temperature > 0.f, VH_EINVAL,
               "vh_ar_decoder: sampling (top_k=%d) needs temperature > 0", d->top_k);
    if (d->head_ws) {
        VH_REQUIRE(d->top_k == 1 && d->B <= 64 && d->d_model <= 1024, VH_EUNSUPPORTED,
                   "vh_ar_decoder: head_ws (head + greedy step in one launch) is for top_k == 1, B <= 64, d_model <= 1024");
        VH_REQUIRE(d->head_ws_bytes >= vh_head_greedy_ws_bytes(d->B, d->V) && vh_aligned16(d->head_ws), VH_EINVAL,
                   "vh_ar_decoder: head_ws needs vh_head_greedy_ws_bytes() = %zu bytes, 16-byte aligned (got %zu)",
                   vh_head_greedy_ws_bytes(d->B, d->V), d->head_ws_bytes);
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
    return dec;
}

extern "C" void vh_ar_decoder_destroy(vh_ar_decoder* dec) {
    if (!dec) return;
    if (dec->exec) (void)hipGraphExecDestroy(dec->exec);
    if (dec->graph) (void)hipGraphDestroy(dec->graph);
    if (dec->exec_n) (void)hipGraphExecDestroy(dec->exec_n);
    if (dec->graph_n) (void)hipGraphDestroy(dec->graph_n);
    delete dec;
}

// One decode step for every row: x (B,d) holds the new token's embedding on entry and the NEXT
// token's embedding on exit.  `ev` (optional) brackets each decode-attention launch.
void vh_internal_attn_decode_events(hipEvent_t start, hipEvent_t stop);   // attention.hip
int vh_internal_linear_w16(const float* A, int lda, const uint16_t* W16, const float* bias, const float* residual, int ldr,
                           float* out, int ldo, int M, int N, int K, void* stream);                                   // gemm.hip
int vh_internal_linear_qkv_folded_kv16_w16(const float* A, int lda, const uint16_t* Wf16, const float* c1, const float* c2,
                                           float* q_out, int ldq, uint16_t* kcache16, uint16_t* vcache16, const int32_t* cache_len,
                                           int B, int d_model, int n_heads, int S_max, float ln_eps, void* stream);   // gemm.hip
int vh_internal_ffn_decode_w16(const float* x, int ldx, const uint16_t* w1f16, const float* c1, const float* c2,
                               const uint16_t* w2_16, const float* b2, float* out, int ldo, int M, int d_model, int dff,
                               float ln_eps, void* workspace, size_t workspace_bytes, void* stream);                  // ffn.hip
int vh_internal_sample_step(const float* logits, int ldl, int V, int eos, int top_k, float top_p, float temperature, uint64_t seed,
                            const uint64_t* seed_dev, int64_t* codes, int64_t codes_stride, int32_t* eos_count,
                            const int32_t* pos_base, float* sum_logprobs, const float* audio_emb, const float* pe,
                            int32_t* audio_pos, int32_t* cache_len, float* x_next, int B, int d, void* stream);   // elementwise.hip

static int decoder_enqueue(vh_ar_decoder* dec, hipStream_t s, std::vector<hipEvent_t>* ev,
                           std::vector<hipEvent_t>* kev = nullptr) {
    const vh_ar_decoder_desc& d = dec->d;
    const int B = d.B, D = d.d_model;
    // FeedForward as one launch split over dim_feedforward + the slab reduce (vh_ffn_decode) when the caller gave
    // the workspace and the folded weights; else linear_1 and linear_2 (split-K + reduce) as separate launches
    // measured: +1.6 us per step at 12L/512d x 32 rows, but 1.2 us per LAYER slower at 24L/1024d x 8 rows (256 slices of
    // 16 columns, 8 MB of slabs: profiles/r3_ab_config5_ffn.log) — the default follows the measurements
    const int ffn_knob = vh_tuning(VH_TUNE_FFN_FUSED);
    const bool ffn_fused = d.ffn_ws && ffn_knob != 1 && (d.d_model <= 512 || ffn_knob == 2);
    // decode attention of one layer, optionally bracketed by events (vh_ar_decoder_profile_attn)
    auto attention = [&](const vh_layer& L) -> int {
        if (d.kv_bf16)
            return vh_attn_decode_kv16(d.q, D, (const uint16_t*)L.kcache, (const uint16_t*)L.vcache, d.attn, D, d.cache_len,
                                       1, B, d.n_heads, d.S_max, s);
        if (d.prefix_len > 0)        // the beams of one utterance: prompt K/V read once, each beam's own rows after it
            return vh_attn_decode_shared(d.q, D, L.kprefix, L.vprefix, d.prefix_len, d.prefix_S, L.kcache, L.vcache, d.attn, D,
                                         d.cache_len, 1, B, d.n_heads, d.S_max, d.n_split, d.attn_partial,
                                         d.attn_partial_bytes, s);
        return vh_attn_decode(d.q, D, L.kcache, L.vcache, d.attn, D, d.cache_len, 1, B, d.n_heads, d.S_max, d.n_split,
                              d.attn_partial, s);
    };
    auto run_attention = [&](const vh_layer& L) -> int {
        if (!ev) return attention(L);
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
        const int arc = attention(L);
        (void)hipEventRecord(e1, s);
        vh_internal_attn_decode_events(nullptr, nullptr);
        ev->push_back(e0);
        ev->push_back(e1);
        return arc;
    };
    for (int i = 0; i < d.n_layers; ++i) {
        const vh_layer& L = dec->layers[i];
        // LN1 fused into the QKV GEMM; K/V rows appended at cache_len[b]  (modules.py:146-157,271)
        // (perf mode with h16 copies of the step's four matrices: the same launches over half the weight bytes)
        const bool w16 = d.kv_bf16 && L.wqkv_f16 && L.wo16 && L.w1_f16 && L.w2_16 && D <= 1024;
        if (w16)
            TRY(vh_internal_linear_qkv_folded_kv16_w16(d.x, D, L.wqkv_f16, L.qkv_c1, L.qkv_c2, d.q, D, (uint16_t*)L.kcache,
                                                       (uint16_t*)L.vcache, d.cache_len, B, D, d.n_heads, d.S_max, d.ln_eps, s));
        else if (d.kv_bf16)
            TRY(vh_linear_qkv_folded_kv16(d.x, D, L.wqkv_f, L.qkv_c1, L.qkv_c2, d.q, D, (uint16_t*)L.kcache,
                                          (uint16_t*)L.vcache, d.cache_len, B, D, d.n_heads, d.S_max, d.ln_eps, s));
        else if (L.wqkv_f)
            TRY(vh_linear_qkv_folded(d.x, D, L.wqkv_f, L.qkv_c1, L.qkv_c2, d.q, D, L.kcache, L.vcache,
                                     d.cache_len, B, 1, D, d.n_heads, d.S_max, d.ln_eps, s));
        else
            TRY(vh_linear_qkv(d.x, D, L.wqkv, d.q, D, L.kcache, L.vcache, d.cache_len, B, 1, D, d.n_heads,
                              d.S_max, L.ln1_g, L.ln1_b, nullptr, nullptr, d.ln_eps, s));
        TRY(run_attention(L));
        // out-proj + bias + residual (modules.py:171,277)
        if (w16)
            TRY(vh_internal_linear_w16(d.attn, D, L.wo16, L.bo, d.x, D, d.x, D, B, D, D, s));
        else
            TRY(vh_linear(d.attn, D, L.wo, L.bo, d.x, D, d.x, D, B, D, D, VH_ACT_NONE, nullptr, nullptr,
                          nullptr, nullptr, 0.f, s));
        // LN2 + linear_1 + exact GELU + linear_2 + bias + residual (modules.py:215-221,278-279)
        if (w16 && d.ffn_ws) {             // (the fused FeedForward is the only 16-bit-weight form: taken at every width)
            TRY(vh_internal_ffn_decode_w16(d.x, D, L.w1_f16, L.w1_c1, L.w1_c2, L.w2_16, L.b2, d.x, D, B, D, d.dff, d.ln_eps, d.ffn_ws,
                                           d.ffn_ws_bytes, s));
            continue;
        }
        if (ffn_fused) {
            TRY(vh_ffn_decode(d.x, D, L.w1_f, L.w1_c1, L.w1_c2, L.w2, L.b2, d.x, D, B, D, d.dff, d.ln_eps, d.ffn_ws,
                              d.ffn_ws_bytes, s));
            continue;
        }
        if (L.w1_f)
            TRY(vh_linear_folded(d.x, D, L.w1_f, L.w1_c1, L.w1_c2, nullptr, 0, d.hidden, d.dff, B, d.dff, D,
                                 VH_ACT_GELU_ERF, d.ln_eps, s));
        else
            TRY(vh_linear(d.x, D, L.w1, L.b1, nullptr, 0, d.hidden, d.dff, B, d.dff, D, VH_ACT_GELU_ERF,
                          L.ln2_g, L.ln2_b, nullptr, nullptr, d.ln_eps, s));
        TRY(vh_linear_ws(d.hidden, d.dff, L.w2, L.b2, d.x, D, d.x, D, B, D, d.dff, VH_ACT_NONE, d.gemm_ws,
                         d.gemm_ws_bytes, s));
    }
    // head (no bias, no final norm: valle_ar.py:29,158) then sampling + state update
    if (d.top_k == 1 && d.head_ws) {      // opt-in: one launch for both (the descriptor check made sure the shape is served)
        TRY(vh_head_greedy(d.x, D, d.proj_w, d.logits, dec->ldl, d.V, d.eos, d.codes, d.codes_stride, d.eos_count, d.pos_base,
                           d.audio_emb, d.audio_pe, d.audio_pos, d.cache_len, d.x, B, D, d.head_ws, d.head_ws_bytes, s));
        return VH_OK;
    }
    if (d.kv_bf16 && d.proj_w16 && D <= 1024)
        TRY(vh_internal_linear_w16(d.x, D, d.proj_w16, nullptr, nullptr, 0, d.logits, dec->ldl, B, d.V, D, s));
    else
        TRY(vh_linear(d.x, D, d.proj_w, nullptr, nullptr, 0, d.logits, dec->ldl, B, d.V, D, VH_ACT_NONE,
                      nullptr, nullptr, nullptr, nullptr, 0.f, s));
    if (d.top_k == 1)
        TRY(vh_greedy_step(d.logits, dec->ldl, d.V, d.eos, d.codes, d.codes_stride, d.eos_count,
                           d.pos_base, d.audio_emb, d.audio_pe, d.audio_pos, d.cache_len, d.x, B, D, s));
    else
        TRY(vh_internal_sample_step(d.logits, dec->ldl, d.V, d.eos, d.top_k, d.top_p, d.temperature, d.seed, d.seed_dev,
                                    d.codes, d.codes_stride, d.eos_count, d.pos_base, d.sum_logprobs, d.audio_emb,
                                    d.audio_pe, d.audio_pos, d.cache_len, d.x, B, D, s));
    return VH_OK;
}

extern "C" int vh_ar_decoder_step(vh_ar_decoder* dec, void* stream) {
    VH_REQUIRE(dec, VH_EINVAL, "vh_ar_decoder_step: null decoder");
    return decoder_enqueue(dec, (hipStream_t)stream, nullptr);
}

#define VH_GRAPH_STEPS 8

static int capture_steps(vh_ar_decoder* dec, hipStream_t s, int n_steps, hipGraph_t* graph, hipGraphExec_t* exec) {
    if (*exec) { (void)hipGraphExecDestroy(*exec); *exec = nullptr; }
    if (*graph) { (void)hipGraphDestroy(*graph); *graph = nullptr; }
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    VH_REQUIRE(e == hipSuccess, VH_ELAUNCH, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    int rc = VH_OK;
    for (int i = 0; i < n_steps && rc == VH_OK; ++i) rc = decoder_enqueue(dec, s, nullptr);
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
    TRY(capture_steps(dec, s, 1, &dec->graph, &dec->exec));
    if (vh_tuning(VH_TUNE_GRAPH_STEPS) != 1) TRY(capture_steps(dec, s, VH_GRAPH_STEPS, &dec->graph_n, &dec->exec_n));
    return VH_OK;
}

extern "C" int vh_ar_decoder_replay(vh_ar_decoder* dec, int n_steps, void* stream) {
    VH_REQUIRE(dec && dec->exec, VH_ESTATE, "vh_ar_decoder_replay: capture first");
    VH_REQUIRE(n_steps >= 0, VH_EINVAL, "vh_ar_decoder_replay: n_steps=%d", n_steps);
    hipStream_t s = (hipStream_t)stream;
    int i = 0;
    if (dec->exec_n)
        for (; i + VH_GRAPH_STEPS <= n_steps; i += VH_GRAPH_STEPS) HIP_TRY(hipGraphLaunch(dec->exec_n, s));
    for (; i < n_steps; ++i) HIP_TRY(hipGraphLaunch(dec->exec, s));
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
