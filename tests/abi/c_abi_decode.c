/* The AR decoder driven through the C ABI from plain C (no Python, no torch in this process): what a non-Python host
 * of the reference's generate() (valle/models/valle_ar.py:92-180) would bind.
 *
 *   1. loads the flat model file written by tests/abi/export_tiny_model.py (2L/128d, BASELINE configs[0]:
 *      128 text + BOS + 255 codec tokens, 4 beams) — weights, ids, and the REAL reference's 64 greedy tokens + margins;
 *   2. prompt pass:  vh_embed_sum_pe (text, audio) -> vh_transformer_forward (prefix-LM mask, K/V into the cache)
 *                    -> vh_linear (head on the last row) -> vh_greedy_step;
 *   3. decode:       vh_ln_fold per layer, vh_ar_decoder_create -> 5 eager vh_ar_decoder_step -> vh_ar_decoder_capture
 *                    -> vh_ar_decoder_replay for half of the rest -> vh_ar_decoder_destroy; a second decoder with the
 *                    one-launch head (head_ws -> vh_head_greedy) continues on the same caller-owned state to the end;
 *   4. asserts the tokens of every beam row equal the reference's (up to the first step whose reference margin is
 *      below 1e-4, as the Python parity tests do);
 *   5. vh_attn_rows (prefix mask, ragged key lengths) and vh_attn_decode (no split, 3 key splits) against a
 *      double-precision C loop;
 *   6. descriptor / state errors: a bad vh_ar_decoder_desc is refused with a reason, replay before capture is VH_ESTATE.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/abi/c_abi_decode.c \
 *       -Lvalle2_amd/csrc -lvalle_hip -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_decode
 *   ./c_abi_decode MODEL.bin            exit code 0 = every check passed
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "valle_hip.h"

#define CHECK(cond, ...)                                         \
    do {                                                         \
        if (!(cond)) {                                           \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                        \
            fprintf(stderr, "\n");                               \
            exit(1);                                             \
        }                                                        \
    } while (0)
#define HIP(call) CHECK((call) == hipSuccess, "%s", #call)
#define VH(call)                                                          \
    do {                                                                  \
        int rc_ = (call);                                                 \
        CHECK(rc_ == VH_OK, "%s: rc=%d (%s)", #call, rc_, vh_last_error()); \
    } while (0)

static void* dev_alloc(size_t bytes, const void* src) {
    void* p = NULL;
    HIP(hipMalloc(&p, bytes ? bytes : 16));
    if (src) HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    else HIP(hipMemset(p, 0, bytes ? bytes : 16));
    return p;
}

static float* read_f32(FILE* f, size_t n) {
    float* p = malloc(n * sizeof(float));
    CHECK(p && fread(p, sizeof(float), n, f) == n, "model file too short (%zu floats)", n);
    return p;
}

static float* dev_f32(FILE* f, size_t n) {
    float* h = read_f32(f, n);
    float* d = dev_alloc(n * sizeof(float), h);
    free(h);
    return d;
}

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

static float frand(unsigned* s) {
    *s = *s * 1664525u + 1013904223u;
    return ((*s >> 8) & 0xFFFF) / 65536.0f - 0.5f;
}

/* ------------------------------------------------------------------------------------------------------------- */
static int decode_against_the_reference(const char* path) {
    FILE* f = fopen(path, "rb");
    CHECK(f, "cannot open %s", path);
    int32_t hdr[12];
    CHECK(fread(hdr, sizeof(int32_t), 12, f) == 12 && hdr[0] == 0x314D4856, "bad model file header");
    const int d = hdr[1], h = hdr[2], dff = hdr[3], L = hdr[4], Vt = hdr[5], Va = hdr[6], n_text = hdr[7],
              n_prompt = hdr[8], n_new = hdr[9], B = hdr[10], n_pe = hdr[11];
    const int V = Va + 1, eos = Va, ldl = (V + 3) & ~3;
    const int S0 = n_text + n_prompt, S_max = (S0 + n_new + 31) / 32 * 32;
    printf("c_abi_decode: %dL/%dd/h%d dff %d, %d text + %d codec positions, %d beams, %d new tokens (S_max %d)\n", L, d, h,
           dff, n_text, n_prompt, B, n_new, S_max);
    CHECK(d == h * VH_HEAD_DIM && n_new > 6, "unexpected model shape");

    float* tokens_emb = dev_f32(f, (size_t)Vt * d);
    float* audio_emb = dev_f32(f, (size_t)(Va + 2) * d);
    float* pe_text = dev_f32(f, (size_t)n_pe * d);
    float* pe_audio = dev_f32(f, (size_t)n_pe * d);
    float* proj = dev_f32(f, (size_t)V * d);
    vh_layer* layers = calloc(L, sizeof(vh_layer));
    const size_t cache_elems = (size_t)B * h * S_max * VH_HEAD_DIM;
    for (int i = 0; i < L; ++i) {
        vh_layer* l = &layers[i];
        l->ln1_g = dev_f32(f, d);  l->ln1_b = dev_f32(f, d);
        l->wqkv = dev_f32(f, (size_t)3 * d * d);
        l->wo = dev_f32(f, (size_t)d * d);  l->bo = dev_f32(f, d);
        l->ln2_g = dev_f32(f, d);  l->ln2_b = dev_f32(f, d);
        l->w1 = dev_f32(f, (size_t)dff * d);  l->b1 = dev_f32(f, dff);
        l->w2 = dev_f32(f, (size_t)d * dff);  l->b2 = dev_f32(f, d);
        l->kcache = dev_alloc(cache_elems * sizeof(float), NULL);
        l->vcache = dev_alloc(cache_elems * sizeof(float), NULL);
    }
    int64_t* h_text = malloc(sizeof(int64_t) * n_text);
    int64_t* h_codes = malloc(sizeof(int64_t) * n_prompt);
    int64_t* gold = malloc(sizeof(int64_t) * n_new);
    float* margin = malloc(sizeof(float) * n_new);
    CHECK(fread(h_text, 8, n_text, f) == (size_t)n_text && fread(h_codes, 8, n_prompt, f) == (size_t)n_prompt &&
              fread(gold, 8, n_new, f) == (size_t)n_new && fread(margin, 4, n_new, f) == (size_t)n_new,
          "model file too short (ids)");
    fclose(f);

    hipStream_t s, cap;
    HIP(hipStreamCreate(&s));
    HIP(hipStreamCreate(&cap));

    /* ---- the growing code sequence of the reference (valle_ar.py:115-117,171): (B, codes_stride) int64, EOS-filled */
    const int64_t codes_stride = n_prompt + n_new;
    int64_t* h_all = malloc(sizeof(int64_t) * B * codes_stride);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < codes_stride; ++t) h_all[b * codes_stride + t] = t < n_prompt ? h_codes[t] : eos;
    int64_t* codes = dev_alloc(sizeof(int64_t) * B * codes_stride, h_all);
    int64_t* text = dev_alloc(sizeof(int64_t) * n_text, h_text);

    /* ---- prompt pass (valle_ar.py:141-158 at kv_cache=None) ---- */
    const size_t rows = (size_t)B * S0;
    float* x = dev_alloc(rows * d * sizeof(float), NULL);
    float* xn = dev_alloc(rows * d * sizeof(float), NULL);
    float* q = dev_alloc(rows * d * sizeof(float), NULL);
    float* attn = dev_alloc(rows * d * sizeof(float), NULL);
    float* hidden = dev_alloc(rows * dff * sizeof(float), NULL);
    int32_t* err_flag = dev_alloc(sizeof(int32_t), NULL);
    const float* t_text[1] = {tokens_emb};
    const float* t_audio[1] = {audio_emb};
    const int32_t v_text[1] = {Vt}, v_audio[1] = {Va + 2};
    /* every beam row reads the same ids: batch stride 0 for the text, the row's own codes for the audio stream */
    VH(vh_embed_sum_pe(text, 0, 1, 0, t_text, v_text, 1, pe_text, 0, NULL, x, (int64_t)S0 * d, 0, B, n_text, d, err_flag,
                       NULL, NULL, NULL, s));
    VH(vh_embed_sum_pe(codes, codes_stride, 1, 0, t_audio, v_audio, 1, pe_audio, 0, NULL, x, (int64_t)S0 * d, n_text, B,
                       n_prompt, d, err_flag, NULL, NULL, NULL, s));
    size_t ws_bytes = max_sz(vh_linear_ws_bytes((int)rows, d, d),
                             max_sz(vh_linear_ws_bytes((int)rows, d, dff), vh_linear_ws_bytes((int)rows, dff, d)));
    void* gemm_ws = ws_bytes ? dev_alloc(ws_bytes, NULL) : NULL;
    vh_forward_desc fd;
    memset(&fd, 0, sizeof fd);
    fd.B = B; fd.T = S0; fd.d_model = d; fd.n_heads = h; fd.dff = dff; fd.n_layers = L; fd.S_max = S_max;
    fd.mode = VH_MASK_PREFIX; fd.x_len = n_text; fd.ln_eps = 1e-5f; fd.layers = layers;
    fd.x = x; fd.xn = xn; fd.q = q; fd.attn = attn; fd.hidden = hidden; fd.gemm_ws = gemm_ws; fd.gemm_ws_bytes = ws_bytes;
    VH(vh_transformer_forward(&fd, s));

    /* ---- head + greedy step on the last prompt row (the tail of step 0, valle_ar.py:158-171) ---- */
    float* xs = dev_alloc((size_t)B * d * sizeof(float), NULL);       /* decode residual stream (B, d) */
    HIP(hipMemcpy2DAsync(xs, (size_t)d * sizeof(float), x + (size_t)(S0 - 1) * d, (size_t)S0 * d * sizeof(float),
                         (size_t)d * sizeof(float), B, hipMemcpyDeviceToDevice, s));
    float* logits = dev_alloc((size_t)B * ldl * sizeof(float), NULL);
    int32_t* h_i32 = malloc(sizeof(int32_t) * B);
    for (int b = 0; b < B; ++b) h_i32[b] = S0 - 1;                    /* +1 by the sampling step */
    int32_t* cache_len = dev_alloc(sizeof(int32_t) * B, h_i32);
    for (int b = 0; b < B; ++b) h_i32[b] = n_prompt;
    int32_t* audio_pos = dev_alloc(sizeof(int32_t) * B, h_i32);
    int32_t* eos_count = dev_alloc(sizeof(int32_t) * (codes_stride + 1), NULL);
    VH(vh_linear(xs, d, proj, NULL, NULL, 0, logits, ldl, B, V, d, VH_ACT_NONE, NULL, NULL, NULL, NULL, 0.f, s));
    VH(vh_greedy_step(logits, ldl, V, eos, codes, codes_stride, eos_count, NULL, audio_emb, pe_audio, audio_pos, cache_len,
                      xs, B, d, s));

    /* ---- the decoder: folded LayerNorm weights, key splits to fill the chip, the FeedForward as one launch ---- */
    for (int i = 0; i < L; ++i) {
        vh_layer* l = &layers[i];
        float *wf = dev_alloc((size_t)3 * d * d * sizeof(float), NULL), *c1 = dev_alloc(3 * d * sizeof(float), NULL),
              *c2 = dev_alloc(3 * d * sizeof(float), NULL);
        VH(vh_ln_fold(l->wqkv, l->ln1_g, l->ln1_b, NULL, wf, c1, c2, 3 * d, d, s));
        l->wqkv_f = wf; l->qkv_c1 = c1; l->qkv_c2 = c2;
        wf = dev_alloc((size_t)dff * d * sizeof(float), NULL);
        c1 = dev_alloc(dff * sizeof(float), NULL);
        c2 = dev_alloc(dff * sizeof(float), NULL);
        VH(vh_ln_fold(l->w1, l->ln2_g, l->ln2_b, l->b1, wf, c1, c2, dff, d, s));
        l->w1_f = wf; l->w1_c1 = c1; l->w1_c2 = c2;
    }
    int n_split = 1;
    if (B * h < 256) { n_split = (256 + B * h - 1) / (B * h); if (n_split > 16) n_split = 16; }
    vh_ar_decoder_desc dd;
    memset(&dd, 0, sizeof dd);
    dd.B = B; dd.d_model = d; dd.n_heads = h; dd.dff = dff; dd.n_layers = L; dd.S_max = S_max; dd.V = V; dd.eos = eos;
    dd.n_split = n_split; dd.ln_eps = 1e-5f; dd.layers = layers; dd.proj_w = proj; dd.audio_emb = audio_emb;
    dd.audio_pe = pe_audio; dd.x = xs;
    dd.q = dev_alloc((size_t)B * d * sizeof(float), NULL);
    dd.attn = dev_alloc((size_t)B * d * sizeof(float), NULL);
    dd.hidden = dev_alloc((size_t)B * dff * sizeof(float), NULL);
    dd.logits = logits;
    dd.attn_partial = n_split > 1 ? dev_alloc(vh_attn_decode_ws_bytes(B, h, n_split), NULL) : NULL;
    dd.gemm_ws_bytes = vh_linear_ws_bytes(B, d, dff);
    dd.gemm_ws = dd.gemm_ws_bytes ? dev_alloc(dd.gemm_ws_bytes, NULL) : NULL;
    dd.cache_len = cache_len; dd.audio_pos = audio_pos; dd.eos_count = eos_count; dd.pos_base = NULL;
    dd.codes = codes; dd.codes_stride = codes_stride; dd.top_k = 1; dd.top_p = 1.f; dd.temperature = 1.f;
    dd.ffn_ws_bytes = vh_ffn_decode_ws_bytes(B, d, dff);
    dd.ffn_ws = dd.ffn_ws_bytes ? dev_alloc(dd.ffn_ws_bytes, NULL) : NULL;

    /* a bad descriptor is refused with a reason; replay before capture is a state error */
    vh_ar_decoder_desc bad = dd;
    bad.B = 65;
    CHECK(vh_ar_decoder_create(&bad) == NULL && strstr(vh_last_error(), "B=65"), "B = 65 must be refused: '%s'", vh_last_error());
    bad = dd;
    bad.codes = NULL;
    CHECK(vh_ar_decoder_create(&bad) == NULL && strstr(vh_last_error(), "null buffer"), "null codes: '%s'", vh_last_error());
    bad = dd;
    bad.n_split = 2; bad.attn_partial = NULL;
    CHECK(vh_ar_decoder_create(&bad) == NULL && strstr(vh_last_error(), "attn_partial"), "split without workspace: '%s'", vh_last_error());
    CHECK(vh_ar_decoder_create(NULL) == NULL, "null desc");

    vh_ar_decoder* dec = vh_ar_decoder_create(&dd);
    CHECK(dec, "vh_ar_decoder_create: %s", vh_last_error());
    CHECK(vh_ar_decoder_replay(dec, 1, s) == VH_ESTATE, "replay before capture must be VH_ESTATE");
    CHECK(vh_ar_decoder_capture(dec, NULL) == VH_EINVAL, "capture needs a non-null stream");
    const int eager = 5, first = (n_new - 1 - eager) / 2, rest = n_new - 1 - eager - first;
    for (int i = 0; i < eager; ++i) VH(vh_ar_decoder_step(dec, s));                 /* steps 1..5 eagerly */
    VH(vh_ar_decoder_capture(dec, cap));
    VH(vh_ar_decoder_replay(dec, first, s));                                        /* half of the rest as graph replays */
    HIP(hipStreamSynchronize(s));
    vh_ar_decoder_destroy(dec);
    vh_ar_decoder_destroy(NULL);
    /* a SECOND decoder continues on the same caller-owned state (codes, positions, caches, residual rows) — this one with
     * the head and the greedy step as one launch (vh_head_greedy, opt-in through head_ws) */
    dd.head_ws_bytes = vh_head_greedy_ws_bytes(B, V);
    CHECK(dd.head_ws_bytes == 256 + (size_t)B * ((V + 15) / 16) * 8, "vh_head_greedy_ws_bytes = %zu", dd.head_ws_bytes);
    dd.head_ws = dev_alloc(dd.head_ws_bytes, NULL);                                 /* zeroed */
    bad = dd;
    bad.head_ws_bytes = 64;
    CHECK(vh_ar_decoder_create(&bad) == NULL && strstr(vh_last_error(), "head_ws"), "small head workspace: '%s'", vh_last_error());
    dec = vh_ar_decoder_create(&dd);
    CHECK(dec, "vh_ar_decoder_create (one-launch head): %s", vh_last_error());
    VH(vh_ar_decoder_capture(dec, cap));
    VH(vh_ar_decoder_replay(dec, rest, s));
    HIP(hipStreamSynchronize(s));
    vh_ar_decoder_destroy(dec);

    HIP(hipMemcpy(h_all, codes, sizeof(int64_t) * B * codes_stride, hipMemcpyDeviceToHost));
    int32_t flag = 0;
    HIP(hipMemcpy(&flag, err_flag, sizeof flag, hipMemcpyDeviceToHost));
    CHECK(flag == 0, "device error flag %d (an id outside its table)", flag);
    HIP(hipMemcpy(h_i32, audio_pos, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) CHECK(h_i32[b] == n_prompt + n_new, "row %d stopped at position %d", b, h_i32[b]);
    int first_bad = -1;
    for (int t = 0; t < n_new && first_bad < 0; ++t)
        for (int b = 0; b < B; ++b)
            if (h_all[b * codes_stride + n_prompt + t] != gold[t]) { first_bad = t; break; }
    if (first_bad >= 0)
        CHECK(margin[first_bad] <= 1e-4f, "greedy tokens diverge from the reference at step %d (margin %.3e): %lld vs %lld",
              first_bad, margin[first_bad], (long long)h_all[n_prompt + first_bad], (long long)gold[first_bad]);
    printf("c_abi_decode: %d greedy tokens x %d beams equal the reference's (%s; n_split %d, %d eager + %d replayed steps)\n",
           first_bad < 0 ? n_new : first_bad, B, first_bad < 0 ? "all" : "up to a near-tie", n_split, eager, n_new - 1 - eager);
    printf("c_abi_decode: a second decoder (vh_head_greedy: head + greedy step in one launch) took over after step %d\n", eager + first);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------- */
static int attention_against_double_loops(void) {
    enum { B = 2, H = 2, T = 40, XL = 13, SMAX = 128, D = H * 64 };
    unsigned seed = 11;
    static float hq[B * T * D], hk[B * H * SMAX * 64], hv[B * H * SMAX * 64], got[B * T * D];
    for (int i = 0; i < B * T * D; ++i) hq[i] = 2.f * frand(&seed);
    for (int i = 0; i < B * H * SMAX * 64; ++i) { hk[i] = 2.f * frand(&seed); hv[i] = 2.f * frand(&seed); }
    const int32_t kvl[B] = {T, 31};
    float *q = dev_alloc(sizeof hq, hq), *k = dev_alloc(sizeof hk, hk), *v = dev_alloc(sizeof hv, hv);
    float* out = dev_alloc(sizeof got, NULL);
    int32_t* kv_len = dev_alloc(sizeof kvl, kvl);
    hipStream_t s;
    HIP(hipStreamCreate(&s));
    /* many-row attention under build_attn_mask(x_len, y_len) + key padding (valle/models/utils.py:17-43) */
    VH(vh_attn_rows(q, D, k, v, out, D, B, H, T, T, SMAX, VH_MASK_PREFIX, XL, NULL, kv_len, NULL, NULL, s));
    HIP(hipStreamSynchronize(s));
    HIP(hipMemcpy(got, out, sizeof got, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int b = 0; b < B; ++b)
        for (int hh = 0; hh < H; ++hh)
            for (int i = 0; i < T; ++i) {
                double sc[T], mx = -1e300, den = 0, o[64] = {0};
                for (int j = 0; j < T; ++j) {
                    const int vis = j < kvl[b] && (j < XL || (i >= XL && j <= i));
                    double a = 0;
                    for (int c = 0; c < 64; ++c) a += (double)hq[(b * T + i) * D + hh * 64 + c] * hk[((b * H + hh) * SMAX + j) * 64 + c];
                    sc[j] = vis ? a / 8.0 : -1e300;
                    if (sc[j] > mx) mx = sc[j];
                }
                for (int j = 0; j < T; ++j) {
                    const double p = sc[j] <= -1e299 ? 0 : exp(sc[j] - mx);
                    den += p;
                    for (int c = 0; c < 64; ++c) o[c] += p * hv[((b * H + hh) * SMAX + j) * 64 + c];
                }
                for (int c = 0; c < 64; ++c) worst = fmax(worst, fabs(o[c] / den - got[(b * T + i) * D + hh * 64 + c]));
            }
    printf("c_abi_decode: vh_attn_rows (prefix mask, ragged keys)      max |err| = %.2e\n", worst);
    CHECK(worst < 3e-5, "vh_attn_rows differs from the double-precision loop");

    /* one-query decode attention over the cache: no split, then 3 key splits + combine */
    const int32_t clen[B] = {99, 36};                 /* keys 0..cache_len[b] (len_bias = 1) */
    int32_t* cache_len = dev_alloc(sizeof clen, clen);
    void* partial = dev_alloc(vh_attn_decode_ws_bytes(B, H, 3), NULL);
    for (int n_split = 1; n_split <= 3; n_split += 2) {
        VH(vh_attn_decode(q, T * D, k, v, out, D, cache_len, 1, B, H, SMAX, n_split, n_split > 1 ? partial : NULL, s));
        HIP(hipStreamSynchronize(s));
        HIP(hipMemcpy(got, out, sizeof(float) * B * D, hipMemcpyDeviceToHost));
        worst = 0;
        for (int b = 0; b < B; ++b)
            for (int hh = 0; hh < H; ++hh) {
                double sc[SMAX], mx = -1e300, den = 0, o[64] = {0};
                const int n = clen[b] + 1;
                for (int j = 0; j < n; ++j) {
                    double a = 0;
                    for (int c = 0; c < 64; ++c) a += (double)hq[(b * T) * D + hh * 64 + c] * hk[((b * H + hh) * SMAX + j) * 64 + c];
                    sc[j] = a / 8.0;
                    if (sc[j] > mx) mx = sc[j];
                }
                for (int j = 0; j < n; ++j) {
                    const double p = exp(sc[j] - mx);
                    den += p;
                    for (int c = 0; c < 64; ++c) o[c] += p * hv[((b * H + hh) * SMAX + j) * 64 + c];
                }
                for (int c = 0; c < 64; ++c) worst = fmax(worst, fabs(o[c] / den - got[b * D + hh * 64 + c]));
            }
        printf("c_abi_decode: vh_attn_decode (n_split = %d)                   max |err| = %.2e\n", n_split, worst);
        CHECK(worst < 3e-5, "vh_attn_decode differs from the double-precision loop");
    }
    CHECK(vh_attn_decode(q, D, k, v, out, D, cache_len, 1, B, H, SMAX, 2, NULL, s) != VH_OK, "split without workspace must be refused");
    return 0;
}

int main(int argc, char** argv) {
    CHECK(argc > 1, "usage: c_abi_decode MODEL.bin");
    CHECK(vh_version() == VH_VERSION, "vh_version() = %d, header says %d", vh_version(), VH_VERSION);
    if (attention_against_double_loops()) return 1;
    if (decode_against_the_reference(argv[1])) return 1;
    printf("c_abi_decode: all checks passed\n");
    return 0;
}
