// Embedding-sum + position add, LayerNorm / adaptive LayerNorm, greedy step.
// All HBM-bound row kernels: one wave64 per row, float4 per lane, wave-shuffle reductions.
#include "vh_common.h"

struct TablePtrs {
    const float* t[VH_MAX_TABLES];
    int vocab[VH_MAX_TABLES];
};

// ---------------------------------------------------------------------------------------------
// K1/K2  out[b, out_t0+t, :] = sum_j tables[j][ids[b,t,j], :] + pe[pos0+t, :]
// grid: (ceil(T/4), B), block 256 = 4 waves, one wave per (b,t) row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_sum_pe_kernel(
    const int64_t* __restrict__ ids, int64_t ids_bs, int64_t ids_ts, int64_t ids_js, TablePtrs tabs,
    int n_tables, const float* __restrict__ pe, int pos0, const int32_t* __restrict__ lens,
    float* __restrict__ out, int64_t out_bs, int out_t0, int T, int d, int32_t* __restrict__ err_flag,
    const int32_t* __restrict__ row_pos0, const int32_t* __restrict__ row_t0, DropArgs drop) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (t >= T) return;
    if (lens && t >= lens[b]) return;
    if (row_pos0) pos0 = row_pos0[b];               // ragged batches: every row its own position / output offset
    if (row_t0) out_t0 = row_t0[b];
    const int64_t* idp = ids + b * ids_bs + t * ids_ts;
    // ids are range-checked here (the reference's nn.Embedding raises IndexError): an id outside its
    // table reads row 0 and raises the device flag the host polls at its next synchronisation
    int64_t row[VH_MAX_TABLES];
    bool bad = false;
    for (int j = 0; j < n_tables; ++j) {
        const int64_t id = idp[j * ids_js];
        const bool ok = id >= 0 && id < tabs.vocab[j];
        bad |= !ok;
        row[j] = ok ? id : 0;
    }
    if (bad && err_flag && lane == 0) atomicOr(err_flag, VH_DEVERR_EMBED_ID);
    float* orow = out + b * out_bs + (int64_t)(out_t0 + t) * d;
    const float* prow = pe ? pe + (int64_t)(pos0 + t) * d : nullptr;
    // dropout after the position add (modules.py:80): the field's row is this row's index in `out`
    const uint32_t frow = (uint32_t)((b * out_bs) / d + out_t0 + t);
    for (int c = lane * 4; c < d; c += 256) {
        f32x4 acc = prow ? ld4(prow + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 e = ld4(tabs.t[0] + row[0] * d + c);
        // sum the codebooks first, then add the position row: same order as the reference
        // (emb0 + emb1 + ... then + pe), so fp32 rounding matches op for op.
        for (int j = 1; j < n_tables; ++j) e += ld4(tabs.t[j] + row[j] * d + c);
        e += acc;
        if (drop.thresh) e = e * vh_dropmul4(drop, frow, (uint32_t)c >> 2);
        st4(orow + c, e);
    }
}

extern "C" int vh_embed_sum_pe(const int64_t* ids, int64_t ids_bstride, int64_t ids_tstride,
                               int64_t ids_jstride, const float* const* tables, const int32_t* vocab,
                               int n_tables, const float* pe, int pos0, const int32_t* lens, float* out,
                               int64_t out_bstride, int out_t0, int B, int T, int d, int32_t* err_flag,
                               const int32_t* row_pos0, const int32_t* row_t0, const vh_dropout_spec* drop,
                               void* stream) {
    VH_REQUIRE(ids && tables && vocab && out, VH_EINVAL, "vh_embed_sum_pe: null pointer");
    VH_REQUIRE(n_tables >= 1 && n_tables <= VH_MAX_TABLES, VH_EINVAL,
               "vh_embed_sum_pe: n_tables=%d not in 1..%d", n_tables, VH_MAX_TABLES);
    VH_REQUIRE(B >= 0 && T >= 0 && d > 0 && d % 4 == 0, VH_EINVAL,
               "vh_embed_sum_pe: bad dims B=%d T=%d d=%d (d must be a multiple of 4)", B, T, d);
    VH_REQUIRE(vh_aligned16(out) && (!pe || vh_aligned16(pe)) && out_bstride % 4 == 0, VH_EALIGN,
               "vh_embed_sum_pe: out/pe must be 16-byte aligned");
    DropArgs da;
    VH_REQUIRE(VH_DROP_OK(drop), VH_EINVAL, "vh_embed_sum_pe: dropout p must be in [0, 1)");
    if (vh_drop_args(drop, &da))
        VH_REQUIRE(out_bstride % d == 0 && (int64_t)B * (out_bstride / d) < (1ll << 32), VH_EINVAL,
                   "vh_embed_sum_pe: dropout needs out_bstride %% d == 0");
    if (B == 0 || T == 0) return VH_OK;
    TablePtrs tp{};
    for (int j = 0; j < n_tables; ++j) {
        VH_REQUIRE(tables[j] && vh_aligned16(tables[j]), VH_EALIGN,
                   "vh_embed_sum_pe: table %d null or unaligned", j);
        VH_REQUIRE(vocab[j] > 0, VH_EINVAL, "vh_embed_sum_pe: table %d has %d rows", j, vocab[j]);
        tp.t[j] = tables[j];
        tp.vocab[j] = vocab[j];
    }
    dim3 grid((T + 3) / 4, B);
    hipLaunchKernelGGL(embed_sum_pe_kernel, grid, dim3(256), 0, (hipStream_t)stream, ids,
                       ids_bstride, ids_tstride, ids_jstride, tp, n_tables, pe, pos0, lens, out,
                       out_bstride, out_t0, T, d, err_flag, row_pos0, row_t0, da);
    VH_CHECK_LAUNCH("vh_embed_sum_pe");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// out[b, t, :] = x[b, t, :] + pe[pos0 + t, :]: PositionalEncoding.forward called as a module (valle/models/modules.py:78-80);
// the model paths add the position inside embed_sum_pe_kernel.  One float4 per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_pe_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                     float* __restrict__ out, int64_t n4, int T, int d4, int pos0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int64_t row = i / d4;
    const int c = (int)(i - row * d4), t = (int)(row % T);
    st4(out + i * 4, ld4(x + i * 4) + ld4(pe + ((int64_t)(pos0 + t) * d4 + c) * 4));
}

extern "C" int vh_add_pe(const float* x, const float* pe, float* out, int B, int T, int d, int pos0, void* stream) {
    VH_REQUIRE(x && pe && out, VH_EINVAL, "vh_add_pe: null pointer");
    VH_REQUIRE(B >= 0 && T >= 0 && d > 0 && d % 4 == 0 && pos0 >= 0, VH_EINVAL, "vh_add_pe: bad dims B=%d T=%d d=%d pos0=%d", B, T,
               d, pos0);
    VH_REQUIRE(vh_aligned16(x) && vh_aligned16(pe) && vh_aligned16(out), VH_EALIGN, "vh_add_pe: pointers must be 16-byte aligned");
    const int64_t n4 = (int64_t)B * T * (d / 4);
    if (n4 == 0) return VH_OK;
    hipLaunchKernelGGL(add_pe_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, pe, out, n4, T,
                       d / 4, pos0);
    VH_CHECK_LAUNCH("vh_add_pe");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// Free-standing dropout (a module-level nn.Dropout outside the fused training step) and the field itself.
// One float4 per thread and Philox call; rows x cols/4 work items.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out,
                                                      int ldo, uint8_t* __restrict__ keep, int64_t rows, int c4n,
                                                      DropArgs drop) {
    const int64_t n = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c4n;
        const int c4 = (int)(i - r * c4n);
        const f32x4 m = vh_dropmul4(drop, (uint32_t)r, (uint32_t)c4);
        if (keep) {
            *reinterpret_cast<uint32_t*>(keep + i * 4) = (m.x != 0.f ? 1u : 0u) | (m.y != 0.f ? 0x100u : 0u) |
                                                         (m.z != 0.f ? 0x10000u : 0u) | (m.w != 0.f ? 0x1000000u : 0u);
        } else {
            st4(out + r * ldo + 4 * c4, ld4(x + r * ldx + 4 * c4) * m);
        }
    }
}

extern "C" int vh_dropout(const float* x, int ldx, float* out, int ldo, int64_t rows, int cols,
                          const vh_dropout_spec* spec, void* stream) {
    VH_REQUIRE(x && out && rows >= 0 && rows < (1ll << 32) && cols > 0 && cols % 4 == 0 && ldx >= cols && ldo >= cols &&
                   ldx % 4 == 0 && ldo % 4 == 0,
               VH_EINVAL, "vh_dropout: bad dims rows=%lld cols=%d (cols and strides multiples of 4)", (long long)rows, cols);
    VH_REQUIRE(vh_aligned16(x) && vh_aligned16(out), VH_EALIGN, "vh_dropout: pointers must be 16-byte aligned");
    VH_REQUIRE(VH_DROP_OK(spec), VH_EINVAL, "vh_dropout: p must be in [0, 1)");
    if (rows == 0) return VH_OK;
    DropArgs da;
    hipStream_t st = (hipStream_t)stream;
    if (!vh_drop_args(spec, &da)) {                      // p == 0: the identity
        if (x != out && hipMemcpy2DAsync(out, (size_t)ldo * 4, x, (size_t)ldx * 4, (size_t)cols * 4, (size_t)rows,
                                         hipMemcpyDeviceToDevice, st) != hipSuccess) {
            vh_set_error("vh_dropout: copy failed");
            return VH_ELAUNCH;
        }
        return VH_OK;
    }
    const int64_t n = rows * (cols / 4);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, out, ldo, (uint8_t*)nullptr, rows, cols / 4, da);
    VH_CHECK_LAUNCH("vh_dropout");
    return VH_OK;
}

extern "C" int vh_dropout_mask(uint8_t* keep, int64_t rows, int cols, const vh_dropout_spec* spec, void* stream) {
    VH_REQUIRE(keep && rows >= 0 && rows < (1ll << 32) && cols > 0 && cols % 4 == 0, VH_EINVAL,
               "vh_dropout_mask: bad dims rows=%lld cols=%d", (long long)rows, cols);
    VH_REQUIRE((reinterpret_cast<uintptr_t>(keep) & 3u) == 0, VH_EALIGN, "vh_dropout_mask: keep must be 4-byte aligned");
    VH_REQUIRE(VH_DROP_OK(spec), VH_EINVAL, "vh_dropout_mask: p must be in [0, 1)");
    if (rows == 0) return VH_OK;
    DropArgs da;
    hipStream_t st = (hipStream_t)stream;
    if (!vh_drop_args(spec, &da)) {
        if (hipMemsetAsync(keep, 1, (size_t)rows * cols, st) != hipSuccess) {
            vh_set_error("vh_dropout_mask: memset failed");
            return VH_ELAUNCH;
        }
        return VH_OK;
    }
    const int64_t n = rows * (cols / 4);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, st, (const float*)nullptr, 0, (float*)nullptr, 0, keep,
                       rows, cols / 4, da);
    VH_CHECK_LAUNCH("vh_dropout_mask");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// K3/K4  LayerNorm (two-pass in registers: mean, then sum of squared deviations) + AdaLN affine
// NV = float4 chunks per lane (d <= 256*NV).  grid: ceil(rows/4), block 256.
// ---------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ ada_scale, const float* __restrict__ ada_shift,
    float* __restrict__ out, int rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * d;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = c < d ? ld4(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            f32x4 t = v[i] - mean;
            ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    float* orow = out + (int64_t)row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            f32x4 y = (v[i] - mean) * rstd * ld4(gamma + c) + ld4(beta + c);
            if (ada_scale) y = ld4(ada_scale + c) * y + ld4(ada_shift + c);
            st4(orow + c, y);
        }
    }
}

extern "C" int vh_layernorm(const float* x, const float* gamma, const float* beta,
                            const float* ada_scale, const float* ada_shift, float* out, int rows,
                            int d, float eps, void* stream) {
    VH_REQUIRE(x && gamma && beta && out, VH_EINVAL, "vh_layernorm: null pointer");
    VH_REQUIRE((ada_scale == nullptr) == (ada_shift == nullptr), VH_EINVAL,
               "vh_layernorm: ada_scale and ada_shift must be given together");
    VH_REQUIRE(rows >= 0 && d > 0 && d % 4 == 0 && d <= 4096, VH_EINVAL,
               "vh_layernorm: bad dims rows=%d d=%d (d multiple of 4, <= 4096)", rows, d);
    VH_REQUIRE(vh_aligned16(x) && vh_aligned16(out) && vh_aligned16(gamma) && vh_aligned16(beta) &&
                   vh_aligned16(ada_scale) && vh_aligned16(ada_shift),
               VH_EALIGN, "vh_layernorm: pointers must be 16-byte aligned");
    if (rows == 0) return VH_OK;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LN_LAUNCH(NV)                                                                            \
    hipLaunchKernelGGL(layernorm_kernel<NV>, grid, block, 0, s, x, gamma, beta, ada_scale,       \
                       ada_shift, out, rows, d, eps)
    if (d <= 256) LN_LAUNCH(1);
    else if (d <= 512) LN_LAUNCH(2);
    else if (d <= 1024) LN_LAUNCH(4);
    else if (d <= 2048) LN_LAUNCH(8);
    else LN_LAUNCH(16);
#undef LN_LAUNCH
    VH_CHECK_LAUNCH("vh_layernorm");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// K12/K13  greedy step: argmax (lowest index on ties, as torch.topk / multinomial over a one-hot),
// EOS bookkeeping, token append, next-token embedding + position.  One 256-thread block per row.
// The step index is derived from the row's own audio position, so the kernel carries no global
// counter and a captured graph can be replayed as is.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void greedy_step_kernel(
    const float* __restrict__ logits, int ldl, int V, int eos, int64_t* __restrict__ codes,
    int64_t codes_stride, int32_t* __restrict__ eos_count, const int32_t* __restrict__ pos_base,
    const float* __restrict__ audio_emb, const float* __restrict__ pe,
    int32_t* __restrict__ audio_pos, int32_t* __restrict__ cache_len, float* __restrict__ x_next, int d) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ int s_tok;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* lr = logits + (int64_t)b * ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float v = lr[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_val[w] = best; s_idx[w] = bi; }
    __syncthreads();
    const int pos = audio_pos[b];  // index of the token being produced in row b's audio stream
    if (tid == 0) {
        for (int k = 1; k < 4; ++k)
            if (s_val[k] > best || (s_val[k] == best && s_idx[k] < bi)) { best = s_val[k]; bi = s_idx[k]; }
        int64_t* row = codes + (int64_t)b * codes_stride;
        int tok = bi;
        if (row[pos - 1] == (int64_t)eos) tok = eos;  // valle_ar.py:168: finished rows keep EOS
        row[pos] = tok;                               // valle_ar.py:171
        if (tok == eos) atomicAdd(&eos_count[pos - (pos_base ? pos_base[b] : 0)], 1);
        s_tok = tok;
    }
    __syncthreads();
    const int tok = s_tok;
    const float* er = audio_emb + (int64_t)tok * d;   // valle_ar.py:143-144 for the next step,
    const float* pr = pe + (int64_t)pos * d;          // only the new row (modules.py:337 keeps it)
    for (int c = tid * 4; c < d; c += 1024) {
        const f32x4 e = ld4(er + c) + ld4(pr + c);
        st4(x_next + (int64_t)b * d + c, e);
    }
    if (tid == 0) {
        audio_pos[b] = pos + 1;
        cache_len[b] += 1;
    }
}

extern "C" int vh_greedy_step(const float* logits, int ldl, int V, int eos, int64_t* codes,
                              int64_t codes_stride, int32_t* eos_count, const int32_t* pos_base,
                              const float* audio_emb, const float* pe, int32_t* audio_pos,
                              int32_t* cache_len, float* x_next, int B, int d, void* stream) {
    VH_REQUIRE(logits && codes && eos_count && audio_emb && pe && audio_pos && cache_len && x_next,
               VH_EINVAL, "vh_greedy_step: null pointer");
    VH_REQUIRE(B > 0 && V > 0 && ldl >= V && d > 0 && d % 4 == 0, VH_EINVAL,
               "vh_greedy_step: bad dims B=%d V=%d ldl=%d d=%d", B, V, ldl, d);
    VH_REQUIRE(vh_aligned16(audio_emb) && vh_aligned16(pe) && vh_aligned16(x_next), VH_EALIGN,
               "vh_greedy_step: audio_emb/pe/x_next must be 16-byte aligned");
    hipLaunchKernelGGL(greedy_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                       V, eos, codes, codes_stride, eos_count, pos_base, audio_emb, pe, audio_pos,
                       cache_len, x_next, d);
    VH_CHECK_LAUNCH("vh_greedy_step");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// K12 (stochastic)  top-k / top-p / temperature sampling (valle/models/utils.py:46-68 with the
// published transformers==4.38.2 top_k_top_p_filtering semantics), fused with the same decode-state
// update as greedy_step_kernel.  One 256-thread block per row:
//   1. scores = logits / temperature into LDS (padded to a power of two with -inf), bitonic sort
//      descending (ties: lower index first);
//   2. top-k: keep every score >= the k-th largest (ties kept; top_k <= 0 keeps all);
//   3. top-p: walking from the smallest kept score, drop entries whose cumulative probability is
//      <= 1 - top_p, always keeping the largest one;
//   4. draw from the renormalised kept set with a counter-based RNG keyed on (seed, row, position)
//      — a captured graph replays with fresh randomness because the position advances;
//   5. logprob = log p(token) under the filtered distribution; sum_logprobs[b] += logprob while the
//      row has not finished (valle_ar.py:167).
// The RNG stream differs from torch.multinomial's (CPU mt19937 / Philox): parity with the reference
// is distributional, not sample-exact (SURVEY.md §8c "parity unpinned" for top_k > 1).
// ---------------------------------------------------------------------------------------------
#define SAMPLE_MAXV 2048

__device__ __forceinline__ float uniform01(uint64_t seed, uint32_t row, uint32_t pos) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t)row << 32) | (uint64_t)(pos + 1u));
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;     // splitmix64 finaliser
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 random bits → [0, 1)
}

// order-preserving key of a float: a > b  <=>  okey(a) > okey(b)  (-inf lowest; no NaNs expected)
__device__ __forceinline__ uint32_t okey(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void sample_step_kernel(
    const float* __restrict__ logits, int ldl, int V, int eos, int top_k, float top_p, float inv_temp,
    uint64_t seed, int64_t* __restrict__ codes, int64_t codes_stride, int32_t* __restrict__ eos_count,
    const int32_t* __restrict__ pos_base, float* __restrict__ sum_logprobs,
    const float* __restrict__ audio_emb, const float* __restrict__ pe, int32_t* __restrict__ audio_pos,
    int32_t* __restrict__ cache_len, float* __restrict__ x_next, int d, int npow2, const uint64_t* __restrict__ seed_dev) {
    // a captured graph freezes `seed`; a decoder that outlives one generate() keeps the call's seed in device memory instead
    if (seed_dev) seed += *seed_dev;
    __shared__ float s_val[SAMPLE_MAXV];
    __shared__ int s_idx[SAMPLE_MAXV];
    __shared__ int s_hist[256];
    __shared__ int s_sel[4];          // [0] chosen bin, [1] rank still wanted inside it, [2] survivors, [3] scratch
    __shared__ float s_red[8];
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* lr = logits + (int64_t)b * ldl;
    const int pos = audio_pos[b];
    int pick_tok = 0;
    float pick_logprob = 0.f;

    // ---- fast path: top-k only (top_p == 1, the reference's default top_k = 50, tok_p = 1.0) ---------------
    // No sort: the k-th largest score is found by a 4 x 8-bit radix select on order-preserving keys,
    // the survivors (>= k-th, ties kept) are compacted in index order and the draw walks their
    // cumulative probabilities.  (The sorted path below costs 66 barrier-separated bitonic stages and
    // serial exp loops on one thread: 89 us per launch against the 5 us of a greedy step.)
    const bool fast = top_k > 0 && top_k < V && top_p == 1.0f;
    if (fast) {
        float vmax = -INFINITY;
        for (int i = tid; i < V; i += 256) {
            const float x = lr[i] * inv_temp;
            s_val[i] = x;
            vmax = fmaxf(vmax, x);
        }
        vmax = wave_max(vmax);
        if (lane == 0) s_red[wv] = vmax;
        uint32_t prefix = 0, mask = 0;
        int want = top_k;                               // rank (from the top) of the k-th value among candidates
        for (int shift = 24; shift >= 0; shift -= 8) {
            s_hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < V; i += 256) {
                const uint32_t key = okey(s_val[i]);
                if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            if (wv == 0) {                              // lane l owns bins 255-4l .. 252-4l (descending)
                int c[4], tot = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { c[j] = s_hist[255 - 4 * lane - j]; tot += c[j]; }
                int incl = tot;                          // inclusive prefix over lanes 0..lane
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                const int before = incl - tot;           // candidates in higher bins than this lane's
                if (before < want && incl >= want) {     // exactly one lane
                    int acc = before;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (acc < want && acc + c[j] >= want) { s_sel[0] = 255 - 4 * lane - j; s_sel[1] = want - acc; }
                        acc += c[j];
                    }
                }
            }
            __syncthreads();
            prefix |= (uint32_t)s_sel[0] << shift;
            mask |= 255u << shift;
            want = s_sel[1];
        }
        // prefix == key of the k-th largest score.  Survivors: key >= prefix, kept in index order.
        const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        int mine = 0;                                    // each thread owns a contiguous index range
        const int per = (V + 255) / 256, i0 = tid * per, i1 = min(V, i0 + per);
        for (int i = i0; i < i1; ++i) mine += okey(s_val[i]) >= prefix;
        int incl = mine;                                 // block-wide exclusive prefix of `mine`
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_hist[wv] = incl;
        __syncthreads();
        int base = incl - mine;
        for (int ww = 0; ww < wv; ++ww) base += s_hist[ww];
        const int n_keep = s_hist[0] + s_hist[1] + s_hist[2] + s_hist[3];
        __syncthreads();                                 // s_hist / s_val are re-used below
        // compact (value, index); the survivor count is small (top_k plus ties) but may be anything <= V
        float kv[8];
        int ki[8], nk = 0;
        for (int i = i0; i < i1; ++i)
            if (okey(s_val[i]) >= prefix) { kv[nk & 7] = s_val[i]; ki[nk & 7] = i; ++nk; }   // per <= 8 (V <= 2048)
        __syncthreads();
        for (int j = 0; j < nk; ++j) { s_val[base + j] = kv[j]; s_idx[base + j] = ki[j]; }
        __syncthreads();
        if (wv == 0) {
            // total = sum of exp(v - m) over the survivors, 64 at a time
            float total = 0.f;
            for (int j0 = 0; j0 < n_keep; j0 += 64) total += (j0 + lane < n_keep) ? expf(s_val[j0 + lane] - m) : 0.f;
            total = wave_sum(total);
            const float u = uniform01(seed, (uint32_t)b, (uint32_t)pos) * total;
            float run = 0.f;
            int pick = n_keep - 1;
            bool found = false;
            for (int j0 = 0; j0 < n_keep && !found; j0 += 64) {
                const float e = (j0 + lane < n_keep) ? expf(s_val[j0 + lane] - m) : 0.f;
                float inc = e;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const float up = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += up;
                }
                const unsigned long long hit = __ballot(j0 + lane < n_keep && run + inc > u);
                if (hit) { pick = j0 + __ffsll((long long)hit) - 1; found = true; }
                run += __shfl(inc, 63, 64);
            }
            if (lane == 0) {
                s_sel[2] = s_idx[pick];
                s_red[4] = (s_val[pick] - m) - logf(total);
            }
        }
        __syncthreads();
        pick_tok = s_sel[2];
        pick_logprob = s_red[4];
    } else {
        // ---- general path: full descending sort, then top-k / top-p on the sorted scores -------------------
        for (int i = tid; i < npow2; i += 256) {
            s_val[i] = i < V ? lr[i] * inv_temp : -INFINITY;
            s_idx[i] = i;
        }
        __syncthreads();
        for (int k = 2; k <= npow2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npow2; i += 256) {
                    const int p = i ^ j;
                    if (p > i) {
                        const bool desc = (i & k) == 0;
                        const float a = s_val[i], c = s_val[p];
                        const int ia = s_idx[i], ic = s_idx[p];
                        const bool a_first = a > c || (a == c && ia < ic);   // a belongs before c
                        if (desc ? !a_first : a_first) {
                            s_val[i] = c; s_val[p] = a; s_idx[i] = ic; s_idx[p] = ia;
                        }
                    }
                }
                __syncthreads();
            }
        if (tid == 0) {
            int n_keep = V;
            if (top_k > 0) {
                const float kth = s_val[min(top_k, V) - 1];
                n_keep = min(top_k, V);
                while (n_keep < V && s_val[n_keep] >= kth) ++n_keep;         // ties kept
            }
            const float m = s_val[0];
            float total = 0.f;
            for (int i = 0; i < n_keep; ++i) total += expf(s_val[i] - m);
            if (top_p >= 0.f && top_p <= 1.f) {                              // drop the low-probability tail
                const float limit = (1.0f - top_p) * total;
                float tail = 0.f;
                int n2 = n_keep;
                while (n2 > 1) {
                    tail += expf(s_val[n2 - 1] - m);
                    if (tail > limit) break;
                    --n2;
                }
                n_keep = n2;
                total = 0.f;
                for (int i = 0; i < n_keep; ++i) total += expf(s_val[i] - m);
            }
            const float u = uniform01(seed, (uint32_t)b, (uint32_t)pos) * total;
            float acc = 0.f;
            int pick = n_keep - 1;
            for (int i = 0; i < n_keep; ++i) {
                acc += expf(s_val[i] - m);
                if (acc > u) { pick = i; break; }
            }
            s_sel[2] = s_idx[pick];
            s_red[4] = (s_val[pick] - m) - logf(total);
        }
        __syncthreads();
        pick_tok = s_sel[2];
        pick_logprob = s_red[4];
    }
    if (tid == 0) {
        int64_t* row = codes + (int64_t)b * codes_stride;
        int tok = pick_tok;
        const bool finished = row[pos - 1] == (int64_t)eos;
        if (sum_logprobs && !finished) sum_logprobs[b] += pick_logprob;      // valle_ar.py:167
        if (finished) tok = eos;                                             // valle_ar.py:168
        row[pos] = tok;
        if (tok == eos) atomicAdd(&eos_count[pos - (pos_base ? pos_base[b] : 0)], 1);
        s_tok = tok;
    }
    __syncthreads();
    const int tok = s_tok;
    const float* er = audio_emb + (int64_t)tok * d;
    const float* pr = pe + (int64_t)pos * d;
    for (int c = tid * 4; c < d; c += 1024) {
        const f32x4 e = ld4(er + c) + ld4(pr + c);
        st4(x_next + (int64_t)b * d + c, e);
    }
    if (tid == 0) {
        audio_pos[b] = pos + 1;
        cache_len[b] += 1;
    }
}

// vh_sample_step with the seed = seed + *seed_dev (seed_dev may be NULL): the decoder plan's form (plan.hip)
int vh_internal_sample_step(const float* logits, int ldl, int V, int eos, int top_k, float top_p,
                            float temperature, uint64_t seed, const uint64_t* seed_dev, int64_t* codes, int64_t codes_stride,
                            int32_t* eos_count, const int32_t* pos_base, float* sum_logprobs,
                            const float* audio_emb, const float* pe, int32_t* audio_pos,
                            int32_t* cache_len, float* x_next, int B, int d, void* stream) {
    VH_REQUIRE(logits && codes && eos_count && audio_emb && pe && audio_pos && cache_len && x_next,
               VH_EINVAL, "vh_sample_step: null pointer");
    VH_REQUIRE(B > 0 && V > 0 && V <= SAMPLE_MAXV && ldl >= V && d > 0 && d % 4 == 0, VH_EINVAL,
               "vh_sample_step: bad dims B=%d V=%d (<= %d) ldl=%d d=%d", B, V, SAMPLE_MAXV, ldl, d);
    VH_REQUIRE(temperature > 0.f, VH_EINVAL, "vh_sample_step: temperature must be positive");
    VH_REQUIRE(vh_aligned16(audio_emb) && vh_aligned16(pe) && vh_aligned16(x_next), VH_EALIGN,
               "vh_sample_step: audio_emb/pe/x_next must be 16-byte aligned");
    int npow2 = 2;
    while (npow2 < V) npow2 <<= 1;
    hipLaunchKernelGGL(sample_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ldl, V,
                       eos, top_k, top_p, 1.0f / temperature, seed, codes, codes_stride, eos_count,
                       pos_base, sum_logprobs, audio_emb, pe, audio_pos, cache_len, x_next, d, npow2, seed_dev);
    VH_CHECK_LAUNCH("vh_sample_step");
    return VH_OK;
}

extern "C" int vh_sample_step(const float* logits, int ldl, int V, int eos, int top_k, float top_p,
                              float temperature, uint64_t seed, int64_t* codes, int64_t codes_stride,
                              int32_t* eos_count, const int32_t* pos_base, float* sum_logprobs,
                              const float* audio_emb, const float* pe, int32_t* audio_pos,
                              int32_t* cache_len, float* x_next, int B, int d, void* stream) {
    return vh_internal_sample_step(logits, ldl, V, eos, top_k, top_p, temperature, seed, nullptr, codes, codes_stride, eos_count,
                                   pos_base, sum_logprobs, audio_emb, pe, audio_pos, cache_len, x_next, B, d, stream);
}


// ---------------------------------------------------------------------------------------------
// Categorical(logits / temperature).sample() per row — the NAR stage sampler (valle/models/valle_nar.py:160)
// — or the arg-max (lowest index on ties) when `greedy`.  One wave per row, 4 rows per workgroup:
//   lane l owns the contiguous index range [l*chunk, (l+1)*chunk): row maximum by wave reduction, the
//   lane's sum of exp(x - max), an inclusive scan over the 64 lane sums (DPP-free shuffles: 6 steps),
//   u = uniform(seed, row, stream_id) * total picks the lane whose range holds the draw, and that lane walks
//   its range (inverse CDF in index order).  Also returns log p(token) (tests; optional).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void categorical_rows_kernel(
    const float* __restrict__ logits, int ld, int V, int rows, float inv_temp, int greedy, uint64_t seed,
    uint32_t stream_id, int64_t* __restrict__ tokens, int64_t tokens_stride, float* __restrict__ logprob) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* lr = logits + (int64_t)row * ld;
    const int chunk = (V + 63) / 64;
    const int i0 = min(V, lane * chunk), i1 = min(V, i0 + chunk);
    float m = -INFINITY;
    int am = V;                                         // first index of the lane's maximum
    for (int i = i0; i < i1; ++i) {
        const float x = lr[i] * inv_temp;
        if (x > m) { m = x; am = i; }
    }
    const float wm = wave_max(m);
    if (greedy) {
        // lowest index among the lanes holding the row maximum (torch.argmax's tie rule)
        int cand = (m == wm) ? am : V;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
        if (lane == 0) {
            // a row of NaN / -inf only has no maximum: emit the last valid id, never V (outside every table)
            tokens[(int64_t)row * tokens_stride] = min(cand, V - 1);
            if (logprob) logprob[row] = 0.f;
        }
        return;
    }
    float s = 0.f;
    for (int i = i0; i < i1; ++i) s += expf(lr[i] * inv_temp - wm);
    float incl = s;                                     // inclusive scan over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const float total = __shfl(incl, 63);
    const float u = uniform01(seed, (uint32_t)row, stream_id) * total;
    // the first lane whose inclusive sum exceeds u holds the draw (a lane with an empty range never does
    // unless every later lane is empty too: the last non-empty lane is the fallback for u == total rounding)
    const unsigned long long hit = __ballot(incl > u && i1 > i0);
    const unsigned long long nonempty = __ballot(i1 > i0);
    const int owner = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)nonempty);
    if (lane == owner) {
        float acc = incl - s;
        int pick = i1 - 1;
        for (int i = i0; i < i1; ++i) {
            acc += expf(lr[i] * inv_temp - wm);
            if (acc > u) { pick = i; break; }
        }
        tokens[(int64_t)row * tokens_stride] = pick;
        if (logprob) logprob[row] = (lr[pick] * inv_temp - wm) - logf(total);
    }
}

extern "C" int vh_categorical_rows(const float* logits, int ld, int V, int rows, float temperature, int greedy,
                                   uint64_t seed, uint32_t stream_id, int64_t* tokens, int64_t tokens_stride,
                                   float* logprob, void* stream) {
    VH_REQUIRE(logits && tokens && V > 0 && ld >= V && rows >= 0 && tokens_stride >= 1, VH_EINVAL,
               "vh_categorical_rows: bad args rows=%d V=%d ld=%d", rows, V, ld);
    VH_REQUIRE(greedy || temperature > 0.f, VH_EINVAL, "vh_categorical_rows: temperature=%g must be > 0",
               temperature);
    if (rows == 0) return VH_OK;
    hipLaunchKernelGGL(categorical_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld,
                       V, rows, greedy ? 1.0f : 1.0f / temperature, greedy, seed, stream_id, tokens, tokens_stride,
                       logprob);
    VH_CHECK_LAUNCH("vh_categorical_rows");
    return VH_OK;
}
