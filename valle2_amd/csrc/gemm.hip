// fp32 GEMMs of the path:  out = act(LN?(A) · Wᵀ + bias) + residual   (A (M,K), W (N,K))
//
// All on the exact-f32 matrix cores (no TF32 on gfx950):
//   * gemm_tile_kernel    M > 64 (prefill / NAR / training forward): 128x128x32 LDS-tiled,
//                         v_mfma_f32_32x32x2_f32, 4 waves x (2x2 tiles of 32x32), register-staged double
//                         buffer, branch-free hand-pipelined main loop, LDS-transposed float4 epilogue
//                         (bias / GELU / residual / QKV scatter / split-K slab), XCD-aware tile order.
//                         MFMA-bound (157 TF peak).
//   * gemm_skinny_fast    M <= 64 (decode step, heads): weight-streaming, v_mfma_f32_16x16x4_f32 with W
//                         as the A operand so a lane's 4 results are 4 consecutive output columns; K over
//                         the waves of a workgroup, reduced through LDS in fixed order; one workgroup per
//                         (16 columns, 8..16 rows); LayerNorm in the operand load or folded into the
//                         weights (vh_ln_fold); split-K slabs + splitk_reduce_kernel for K > 1024.
//                         Latency-bound.
//   * ffn_decode_kernel   FeedForward of the decode step as ONE launch split over dim_feedforward (ffn.hip).
//   * gemm_skinny_kernel  the guarded generic version for K off the fast shapes.
//
// MFMA operand maps used (cdna_hip_programming.md §3):
//   32x32x2 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D reg r: row i=(r&3)+8(r>>2)+4(l>>5), col j=l&31
//   16x16x4 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r: row i=4(l>>4)+r,          col j=l&15
// The k index of a product is only summed over, so each lane feeds 4 consecutive k (one float4)
// per operand and the 4 MFMAs of a group consume element j of both operands: same k on both sides.
#include <type_traits>

#include "vh_common.h"

struct GemmArgs {
    const float* A;
    int lda;
    const float* W;
    const float* bias;
    const float* res;
    int ldr;
    float* out;
    int ldo;
    int M, N, K, act;
    // QKV epilogue (EPI_QKV): columns [0,d) → out (q), [d,2d) → kcache, [2d,3d) → vcache
    float* kc;
    float* vc;
    const int32_t* cache_len;
    int T, S_max, d_model, n_heads;
    float* aux;   // training forward (EPI_PLAIN, tile kernels): pre-activation acc + bias stored here too
    int ldx;
    float* colsum;  // training backward (EPI_PLAIN, tile kernels): += column sums of the stored output (a bias gradient)
    int k_len;  // K range of one workgroup column (split-K: K / gridDim.y; otherwise K)
    int rg_rows;  // rows per row group when gridDim.z > 1 (16, or 8: half of the MFMA's 16 rows idle)
    int w16;      // decode GEMMs of perf mode: W points at an h16 (N,K) matrix (vh_common.h vh_h16), read as 8 bytes per fragment
    DropArgs drop;  // training forward (EPI_PLAIN, LDS-DMA tile kernel): out = dropout(act(acc + bias)) + residual
#ifdef VH_STAMPS
    long long* dbg;  // diagnostic build only (tools/probe_skinny.hip): per-wave phase stamps
#endif
};

#ifdef VH_STAMPS
#define STAMP(k)                                                                         \
    do {                                                                                 \
        if (a.dbg && (threadIdx.x & 63) == 0)                                            \
            a.dbg[((int64_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define STAMP(k)
#endif

enum { EPI_PLAIN = 0, EPI_QKV = 1, EPI_PARTIAL = 2,
       EPI_QKV16 = 3 /* EPI_QKV with the K / V rows appended as bf16 (perf mode of the decode step; skinny kernels only) */,
       EPI_HEAD = 4  /* the AR head with the greedy step in the same launch (vh_head_greedy; gemm_skinny_fast, MT = 1 only) */ };
#define IS_QKV(E) ((E) == EPI_QKV || (E) == EPI_QKV16)

// What the greedy step needs beside the head product (vh_head_greedy): the operands of greedy_step_kernel
// (elementwise.hip) + the hand-over area of the launch — one (logit, column) candidate per (row, 16-column block)
// and one arrival counter per row group.
struct HeadStep {
    uint64_t* cand;          // [rows][gridDim.x] (float bits << 32 | column), written and read at agent scope
    int* counters;           // [gridDim.z], zero between launches (the last arriver of a group resets its own)
    int V, eos;
    int64_t* codes;
    int64_t codes_stride;
    int32_t* eos_count;
    const int32_t* pos_base;
    const float* audio_emb;
    const float* pe;
    int32_t* audio_pos;
    int32_t* cache_len;
    float* x_next;
    int d;
};

// fp32 -> bf16, round to nearest even (finite inputs)
__device__ __forceinline__ uint32_t vh_bf16_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// column group of 4 consecutive output columns starting at n (n % 4 == 0) for row m
template <int EPI>
__device__ __forceinline__ void store4(const GemmArgs& a, int m, int n, f32x4 v) {
    if (m >= a.M || n >= a.N) return;
    if (EPI == EPI_PARTIAL) {  // raw partial sums of K-slice blockIdx.y → slab [split][M][ldo]
        st4(a.out + ((int64_t)blockIdx.y * a.M + m) * a.ldo + n, v);
        return;
    }
    if (EPI == EPI_QKV) {
        const int which = n / a.d_model, c = n - which * a.d_model;
        if (which == 0) {
            st4(a.out + (int64_t)m * a.ldo + c, v);
        } else {
            const int head = c / VH_HEAD_DIM, e = c - head * VH_HEAD_DIM;
            const int b = m / a.T, t = m - b * a.T;
            const int pos = (a.cache_len ? a.cache_len[b] : 0) + t;
            float* base = which == 1 ? a.kc : a.vc;
            st4(base + (((int64_t)b * a.n_heads + head) * a.S_max + pos) * VH_HEAD_DIM + e, v);
        }
        return;
    }
    const bool full = n + 3 < a.N;
    if (full) {
        if (a.bias) v += ld4(a.bias + n);
        if (a.act == VH_ACT_GELU_ERF) {
            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
        }
        if (a.res) v += ld4(a.res + (int64_t)m * a.ldr + n);
        st4(a.out + (int64_t)m * a.ldo + n, v);
    } else {  // ragged last column group (e.g. N = 1025 logits)
        for (int j = 0; j < 4 && n + j < a.N; ++j) {
            float s = v[j];
            if (a.bias) s += a.bias[n + j];
            if (a.act == VH_ACT_GELU_ERF) s = gelu_erf(s);
            if (a.res) s += a.res[(int64_t)m * a.ldr + n + j];
            a.out[(int64_t)m * a.ldo + n + j] = s;
        }
    }
}

template <int EPI>
__device__ __forceinline__ void store1(const GemmArgs& a, int m, int n, float s) {
    if (m >= a.M || n >= a.N) return;
    if (EPI == EPI_QKV) {
        const int which = n / a.d_model, c = n - which * a.d_model;
        if (which == 0) {
            a.out[(int64_t)m * a.ldo + c] = s;
        } else {
            const int head = c / VH_HEAD_DIM, e = c - head * VH_HEAD_DIM;
            const int b = m / a.T, t = m - b * a.T;
            const int pos = (a.cache_len ? a.cache_len[b] : 0) + t;
            float* base = which == 1 ? a.kc : a.vc;
            base[(((int64_t)b * a.n_heads + head) * a.S_max + pos) * VH_HEAD_DIM + e] = s;
        }
        return;
    }
    if (a.bias) s += a.bias[n];
    if (a.act == VH_ACT_GELU_ERF) s = gelu_erf(s);
    if (a.res) s += a.res[(int64_t)m * a.ldr + n];
    a.out[(int64_t)m * a.ldo + n] = s;
}

// =============================================================================================
// Large-M tile kernel.  Block tile 128(M) x 128(N), K step 32.  LDS rows padded to 36 floats so
// the ds_read_b128 fragment reads (16 distinct rows mod 16 per lane group) are conflict-free.
// =============================================================================================
#define TM 128
#define TN 128
#define TK 32
#define LDS_LD 36

// =============================================================================================
// Epilogue of the tile kernels (shared by the register-staged and the LDS-DMA kernel).
//
// The accumulators are transposed through LDS so that a lane owns 4 consecutive columns of one row: thread tid
// handles column group c4 = tid & 31 of rows (tid >> 5) + 8*it.  A wave then stores (and reads the residual as)
// two whole 512-B rows per instruction, a quarter of the memory instructions of the accumulator layout (lane =
// column, 4 B per lane).  That count is what matters: the CU's memory pipeline is shared with the other
// workgroup's operand loads, and with dword stores the K = 512 shapes lost 16 % to the stores and 11 % to the
// residual loads.  Residual and bias are fetched before the main loop, so the epilogue waits for nothing.
//
// Every vector instruction here also competes with the MFMA stream of the CU's other workgroup for the same
// issue port, so the interior-tile path is written to need almost none: addresses are a wave-uniform base
// (scalar adds per row group) plus one per-lane 32-bit offset computed once, there are no per-row bounds checks,
// and the K/V scatter walks (batch row, position) incrementally instead of dividing by T per row.
// =============================================================================================
struct TileEpi {
    f32x4 resv[16];
    f32x4 bias4;
};

// The epilogue operands of a tile are fetched in two parts: bias + residual rows 0..95 (EPI_EARLY of the 16 row groups)
// right after the first operand slab is requested, the last four row groups once the K loop has issued its last MFMA
// (tile_prefetch_late).  All 16 at the start is 68 registers held through the whole K loop on top of 64 accumulators
// and 32 fragment registers: the allocator spilled three of them to scratch, each spill waiting for ITS load — three
// serial memory round trips in front of every tile.
#define EPI_EARLY 12

template <int EPI>
__device__ __forceinline__ void tile_prefetch_rows(const GemmArgs& a, int m0, int n0, int tid, TileEpi& e, int it0, int it1) {
    const int ec4 = tid & 31, erow = tid >> 5;
    const int en = n0 + 4 * ec4;
    const bool interior = m0 + TM <= a.M && n0 + TN <= a.N;         // wave-uniform
    if (interior) {
        if (a.res) {
            const char* base = (const char*)(a.res + (int64_t)m0 * a.ldr + n0);
            const uint32_t voff = (uint32_t)(erow * a.ldr + 4 * ec4) * 4u;
#pragma unroll
            for (int it = it0; it < it1; ++it)
                e.resv[it] = ld4((const float*)(base + (int64_t)it * 8 * a.ldr * 4 + voff));
        } else {
#pragma unroll
            for (int it = it0; it < it1; ++it) e.resv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else {
        const bool ecol_full = en + 3 < a.N;
#pragma unroll
        for (int it = it0; it < it1; ++it) {
            const int m = m0 + erow + 8 * it;
            e.resv[it] = (a.res && ecol_full && m < a.M) ? ld4(a.res + (int64_t)m * a.ldr + en)
                                                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int EPI>
__device__ __forceinline__ void tile_prefetch(const GemmArgs& a, int m0, int n0, int tid, TileEpi& e) {
    const int en = n0 + 4 * (tid & 31);
    e.bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI != EPI_PLAIN) return;
    if (a.bias && en + 3 < a.N) e.bias4 = ld4(a.bias + en);
    tile_prefetch_rows<EPI>(a, m0, n0, tid, e, 0, EPI_EARLY);
}

template <int EPI>
__device__ __forceinline__ void tile_prefetch_late(const GemmArgs& a, int m0, int n0, int tid, TileEpi& e) {
    if (EPI != EPI_PLAIN) return;
    tile_prefetch_rows<EPI>(a, m0, n0, tid, e, EPI_EARLY, 16);
}

// Column sums of the tile's stored output, added to a.colsum[n0 ..]: thread (erow = tid >> 5, ec4 = tid & 31) holds the
// sum over its 16 rows of 4 columns; the 8 row groups meet in LDS (ct is free once every thread has read its rows),
// then one atomic per column and workgroup.
__device__ __forceinline__ void tile_colsum(const GemmArgs& a, float* ct, f32x4 csum, int n0, int tid) {
    __syncthreads();                                   // every thread is done reading ct
    st4(ct + (tid >> 5) * TN + 4 * (tid & 31), csum);
    __syncthreads();
    if (tid < TN) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += ct[r * TN + tid];
        if (n0 + tid < a.N) atomicAdd(a.colsum + n0 + tid, t);
    }
}

// ct: 128 x 132 floats of LDS, free (the main loop ended on a barrier).  D reg e of tile (mt,nt) holds row
// (e&3)+8(e>>2)+4h, column r: 32 lanes write 32 consecutive floats (conflict-free).
// DROP (EPI_PLAIN, act NONE / GELU_ERF / GELU_ERF_D): the dropout field of a.drop multiplies act(acc + bias) — and the
// stored GELU derivative — before the residual is added.  A template parameter: the dropout-off kernels stay as they were.
template <int EPI, bool DROP = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, const f32x16 (&acc)[2][2], float* ct, int m0, int n0,
                                              int tid, const TileEpi& e) {
    constexpr int LDC = TN + 4;                       // 132 floats: rows stay 16-byte aligned
    const int lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5, wm = w >> 1, wn = w & 1;
    float* cw = ct + (wm * 64 + 4 * h) * LDC + wn * 64 + r;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int x = 0; x < 16; ++x)
                cw[(mt * 32 + (x & 3) + 8 * (x >> 2)) * LDC + nt * 32] = acc[mt][nt][x];
    __syncthreads();
    const int ec4 = tid & 31, erow = tid >> 5;
    const int en = n0 + 4 * ec4;
    const float* cr = ct + erow * LDC + 4 * ec4;
    const bool interior = m0 + TM <= a.M && n0 + TN <= a.N;         // wave-uniform
    // QKV: a tile lies inside one of q / k / v when d_model is a multiple of the tile width
    const bool qkv_tile = EPI == EPI_QKV && a.d_model % TN == 0;
    if (interior && (EPI == EPI_PLAIN || EPI == EPI_PARTIAL || (qkv_tile && n0 < a.d_model))) {
        // rows of `out`: base + it * (8 rows) + per-lane offset
        const int64_t row0 = EPI == EPI_PARTIAL ? (int64_t)blockIdx.y * a.M + m0 : m0;
        char* base = (char*)(a.out + row0 * a.ldo + n0);
        const uint32_t voff = (uint32_t)(erow * a.ldo + 4 * ec4) * 4u;
        char* xbase = (EPI == EPI_PLAIN && a.aux) ? (char*)(a.aux + (int64_t)m0 * a.ldx + n0) : nullptr;
        const uint32_t xoff = (uint32_t)(erow * a.ldx + 4 * ec4) * 4u;
        f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            f32x4 v = ld4(cr + it * 8 * LDC);
            if (EPI == EPI_PLAIN) {
                v += e.bias4;
                f32x4 dm;
                if (DROP) dm = vh_dropmul4(a.drop, (uint32_t)(m0 + erow + 8 * it), (uint32_t)(en >> 2));
                if (a.act == VH_ACT_GELU_ERF_D) {               // GELU out, its derivative to aux (one erf for both)
                    f32x4 dv;
                    v = gelu_and_grad4(v, dv);
                    if (DROP) dv = dv * dm;
                    st4((float*)(xbase + (int64_t)it * 8 * a.ldx * 4 + xoff), dv);
                } else if (xbase) {
                    st4((float*)(xbase + (int64_t)it * 8 * a.ldx * 4 + xoff), v);           // pre-activation
                }
                if (a.act == VH_ACT_GELU_ERF) {                 // two per instruction (packed fp32)
                    const vh_f32x2 g0 = gelu_erf2(vh_f32x2{v.x, v.y}), g1 = gelu_erf2(vh_f32x2{v.z, v.w});
                    v = f32x4{g0.x, g0.y, g1.x, g1.y};
                }
                if (DROP) v = v * dm;
                if (a.act == VH_ACT_GELU_BWD) {                 // dY through the activation: acc * gelu'(pre)
                    const f32x4 p = e.resv[it];
                    v = f32x4{v.x * gelu_grad(p.x), v.y * gelu_grad(p.y), v.z * gelu_grad(p.z), v.w * gelu_grad(p.w)};
                } else if (a.act == VH_ACT_MUL) {               // ... with the derivative saved by the forward
                    v = v * e.resv[it];
                } else {
                    v += e.resv[it];
                }
            }
            st4((float*)(base + (int64_t)it * 8 * a.ldo * 4 + voff), v);
            if (EPI == EPI_PLAIN) csum += v;
        }
        if (EPI == EPI_PLAIN && a.colsum) tile_colsum(a, ct, csum, n0, tid);
        return;
    }
    if (interior && qkv_tile) {                        // a K or V tile: scatter rows into cache[b][head][pos][64]
        const int which = n0 >= 2 * a.d_model ? 2 : 1;
        const int c = en - which * a.d_model;
        float* cache = (which == 1 ? a.kc : a.vc) + (int64_t)(c / VH_HEAD_DIM) * a.S_max * VH_HEAD_DIM + (c % VH_HEAD_DIM);
        const int m = m0 + erow;
        int b = m / a.T, t = m - b * a.T;
        int cl = a.cache_len ? a.cache_len[b] : 0;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const f32x4 v = ld4(cr + it * 8 * LDC);
            st4(cache + ((int64_t)b * a.n_heads * a.S_max + cl + t) * VH_HEAD_DIM, v);
            t += 8;
            if (t >= a.T) {                           // next batch row (several at once only when T < 8)
                do { t -= a.T; ++b; } while (t >= a.T);
                if (it < 15 && a.cache_len) cl = a.cache_len[b];
            }
        }
        return;
    }
    // ---- edge tiles (ragged M or N) and QKV with d_model not a multiple of 128: guarded general form
    if (en >= a.N) return;                             // (never with a.colsum: the host requires N % 128 == 0 then)
    const bool ecol_full = en + 3 < a.N;
    float* qdst = nullptr;
    bool qcache = false;
    if (EPI == EPI_QKV) {
        const int which = en >= 2 * a.d_model ? 2 : (en >= a.d_model ? 1 : 0);
        const int c = en - which * a.d_model;
        if (which == 0) {
            qdst = a.out + c;
        } else {
            qdst = (which == 1 ? a.kc : a.vc) + (int64_t)(c / VH_HEAD_DIM) * a.S_max * VH_HEAD_DIM + (c % VH_HEAD_DIM);
            qcache = true;
        }
    }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int row = erow + 8 * it, m = m0 + row;
        if (m >= a.M) break;
        f32x4 v = ld4(cr + it * 8 * LDC);
        if (EPI == EPI_PARTIAL) {                      // raw partial sums of K slice blockIdx.y -> slab [split][M][ldo]
            float* dst = a.out + ((int64_t)blockIdx.y * a.M + m) * a.ldo + en;
            if (ecol_full) st4(dst, v);
            else
                for (int j = 0; j < 4 && en + j < a.N; ++j) dst[j] = v[j];
        } else if (EPI == EPI_QKV) {                   // N = 3 d_model, a multiple of 4: groups are whole
            if (qcache) {
                const int b = m / a.T, t = m - b * a.T;
                const int pos = (a.cache_len ? a.cache_len[b] : 0) + t;
                st4(qdst + ((int64_t)b * a.n_heads * a.S_max + pos) * VH_HEAD_DIM, v);
            } else {
                st4(qdst + (int64_t)m * a.ldo, v);
            }
        } else if (ecol_full) {
            v += e.bias4;
            f32x4 dm;
            if (DROP) dm = vh_dropmul4(a.drop, (uint32_t)m, (uint32_t)(en >> 2));
            if (a.act == VH_ACT_GELU_ERF_D) {
                f32x4 dv;
                v = gelu_and_grad4(v, dv);
                if (DROP) dv = dv * dm;
                st4(a.aux + (int64_t)m * a.ldx + en, dv);
            } else if (a.aux) {
                st4(a.aux + (int64_t)m * a.ldx + en, v);
            }
            if (a.act == VH_ACT_GELU_ERF) {
                v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
            }
            if (DROP) v = v * dm;
            if (a.act == VH_ACT_GELU_BWD) {
                const f32x4 p = e.resv[it];
                v = f32x4{v.x * gelu_grad(p.x), v.y * gelu_grad(p.y), v.z * gelu_grad(p.z), v.w * gelu_grad(p.w)};
            } else if (a.act == VH_ACT_MUL) {
                v = v * e.resv[it];
            } else {
                v += e.resv[it];
            }
            st4(a.out + (int64_t)m * a.ldo + en, v);
            csum += v;
        } else {                                       // ragged last column group (e.g. N = 1025 logits)
            for (int j = 0; j < 4 && en + j < a.N; ++j) {
                float sv = v[j] + (a.bias ? a.bias[en + j] : 0.f);
                if (a.act == VH_ACT_GELU_ERF_D) {
                    float dv;
                    sv = gelu_and_grad(sv, dv);
                    a.aux[(int64_t)m * a.ldx + en + j] = dv;
                } else if (a.aux) {
                    a.aux[(int64_t)m * a.ldx + en + j] = sv;
                }
                if (a.act == VH_ACT_GELU_ERF) sv = gelu_erf(sv);
                if (a.act == VH_ACT_GELU_BWD) sv *= gelu_grad(a.res[(int64_t)m * a.ldr + en + j]);
                else if (a.act == VH_ACT_MUL) sv *= a.res[(int64_t)m * a.ldr + en + j];
                else if (a.res) sv += a.res[(int64_t)m * a.ldr + en + j];
                a.out[(int64_t)m * a.ldo + en + j] = sv;
            }
        }
    }
    // (the column-sum epilogue serves whole column groups only: the host refuses it for N % 4 != 0; every thread of a
    // tile that reaches this point — en < N is uniform per column group, not per wave — must take part in the barrier,
    // so edge tiles whose column groups end inside the tile are refused by the host as well: N % 128 == 0)
    if (EPI == EPI_PLAIN && a.colsum) tile_colsum(a, ct, csum, n0, tid);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TM * LDS_LD];  // [buf][A|W][row][k]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD
    // a contiguous run of tiles; tiles are numbered n-fastest so a run shares its A row panel in L2.
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;

    // staging: thread loads 4 float4 of A and 4 of W per K step: rows (tid>>3)+32i, k-quad tid&7
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    f32x4 ra[4], rw[4];
    // row pointers and bounds of this thread's 4 + 4 staging rows, computed once
    const float* pa[4];
    const float* pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        // split-K (EPI_PARTIAL): this workgroup's K range starts at blockIdx.y * k_len
        pa[i] = a.A + (int64_t)min(m0 + row, a.M - 1) * a.lda + skq + blockIdx.y * a.k_len;
        pw[i] = a.W + (int64_t)min(n0 + row, a.N - 1) * a.K + skq + blockIdx.y * a.k_len;
    }
    // Loads are unconditional (no branches, no use of the loaded value inside the main loop): rows
    // beyond M / N are clamped to the last valid row — their products only reach output rows / columns
    // that are never stored — and a K tail (K % 32 = 16) is clamped here and zeroed when the slab is
    // written to LDS.
    auto gload1 = [&](int i, int k0) {           // one A row + one W row of the K slab at k0
        const int kk = k0 + skq < a.k_len ? k0 : 0;  // K % 4 == 0 is guaranteed by the host check
        ra[i] = ld4(pa[i] + kk);
        rw[i] = ld4(pw[i] + kk);
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gload1(i, k0);
    };
    auto lstore = [&](int buf, int k0) {
        const bool kin = k0 + skq < a.k_len;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            st4(&lds[buf][0][row * LDS_LD + skq], kin ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f});
            st4(&lds[buf][1][row * LDS_LD + skq], kin ? rw[i] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (a.k_len + TK - 1) / TK;
    gload(0);
    lstore(0, 0);

    TileEpi epi;                                          // residual and bias, fetched now (see tile_epilogue)
    tile_prefetch<EPI>(a, m0, n0, tid, epi);
    __syncthreads();
    // Main loop, software-pipelined by hand: the operand fragments of k-sub-step t+1 are read from LDS
    // while the 16 MFMAs of sub-step t run (two register sets), and the first fragments of the next
    // k-step are read right after the barrier, behind the last 16 MFMAs of this one.  Left to the
    // compiler the loop was "8 ds_read, wait, 32 MFMA" twice per k-step.  (Worth about 1 %: with two
    // workgroups per CU the other wave of the SIMD already covered most of that wait.)
    f32x4 fa[2][2], fw[2][2];
    auto fload = [&](int set, int buf, int t) {
        const float* As = &lds[buf][0][(wm * 64 + r) * LDS_LD + 4 * h + 8 * t];
        const float* Ws = &lds[buf][1][(wn * 64 + r) * LDS_LD + 4 * h + 8 * t];
        fa[set][0] = ld4(As);
        fa[set][1] = ld4(As + 32 * LDS_LD);
        fw[set][0] = ld4(Ws);
        fw[set][1] = ld4(Ws + 32 * LDS_LD);
    };
    fload(0, 0, 0);
    // one K step; PF = there is a next slab to fetch (every step but the last): no branch inside the step
    auto kstep = [&](int kt, auto pf) {
        constexpr bool PF = decltype(pf)::value;
        const int cur = kt & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // the next K slab is fetched a quarter per sub-step: two loads and their address math fit in
            // the shadow of the MFMAs already issued, a burst of eight at the top of the step does not
            if constexpr (PF) gload1(t, (kt + 1) * TK);
            if (t < 3) fload((t + 1) & 1, cur, t + 1);
            __builtin_amdgcn_sched_barrier(0);       // reads first: they fly under the 16 MFMAs below
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][0][j], fw[t & 1][0][j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][0][j], fw[t & 1][1][j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][1][j], fw[t & 1][0][j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][1][j], fw[t & 1][1][j], acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PF) lstore(cur ^ 1, (kt + 1) * TK);
        __syncthreads();
        if constexpr (PF) fload(0, cur ^ 1, 0);
    };
    for (int kt = 0; kt + 1 < nk; ++kt) kstep(kt, std::true_type{});
    kstep(nk - 1, std::false_type{});

    tile_prefetch_late<EPI>(a, m0, n0, tid, epi);
    tile_epilogue<EPI>(a, acc, &lds[0][0][0], m0, n0, tid, epi);
}

#ifdef VH_TILE_PROBE
// Timeline instrumentation (tools/probe_tile_timeline.hip only): every wave stamps clock64() at the start of
// each K step, before and after the step's barrier, into the 2 KB of slack behind its LDS region; the stamps
// and the hardware id of the workgroup go to global memory before the epilogue.
__device__ long long vh_tile_probe[8192 * 4 * 200];
__device__ unsigned vh_tile_hwid[8192 * 2];
#define VH_TP(ev, step) do { if (lane == 0) tp[(ev) * 64 + (step)] = clock64(); } while (0)
#else
#define VH_TP(ev, step) do { } while (0)
#endif

// =============================================================================================
// LDS-DMA variant of the tile kernel (k_len % 32 == 0): the operand slabs go global -> LDS directly
// (global_load_lds_dwordx4, 1 KB = 8 rows x 128 B per wave-instruction), no staging registers and no
// ds_write.  The LDS image is lane-linear, so rows are unpadded (32 floats) and bank conflicts are
// avoided by an XOR swizzle of the 16-byte chunk index with (row >> 1) & 7 — applied to the per-lane
// SOURCE address of the DMA and to the fragment reads alike (cdna_hip_programming.md rule 21).
// =============================================================================================
// Tail split.  The grid's tiles are dealt to 256 CUs, so T tiles cost ceil(T / 256) tile-times whatever T is: 320 tiles
// (16 x 640 training positions through a 512-wide projection) run as long as 512 do.  With a TailSplit the first
// n_whole tiles (a multiple of 256) are computed whole and each of the remaining tiles as `split` K slices of k_len,
// one workgroup per slice, whose raw sums go to ws[(tail tile * split + slice)][128][128]; tile_tail_fixup_kernel adds
// the slices in slice order and applies the epilogue.  n_whole == number of tiles: no tail, the kernel as before.
struct TailSplit {
    int n_whole, split, k_len;
    float* ws;
};

template <int EPI, bool DROP = false>
__global__ __launch_bounds__(256, 2) void gemm_tile_dma_kernel(GemmArgs a, int tiles_m, int tiles_n, TailSplit ts) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TM * LDS_LD];  // [buf][A|W][row][32] (+ slack for the epilogue)
    // Outside the main loop the wave runs at raised priority: its scalar/vector bookkeeping competes for issue slots
    // with the MFMA stream of the CU's other workgroup, and a short prologue / epilogue is worth more than the few
    // MFMA slots it displaces.
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef VH_TILE_PROBE
    const long long t_entry = clock64();
#endif
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD
    // a contiguous run of tiles; tiles are numbered n-fastest so a run shares its A row panel in L2.
    const int nwg = ts.n_whole;                           // (the whole tiles; the K slices of the tail tiles follow them)
    const int bid = blockIdx.x;
    const bool tail = bid >= nwg;                         // workgroup-uniform
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int unit = bid - nwg;
    const int tile = tail ? nwg + unit / ts.split
                          : (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    const int k_len = tail ? ts.k_len : a.k_len;
    const int k_ofs = tail ? (unit % ts.split) * ts.k_len : blockIdx.y * a.k_len;

    // DMA staging: wave w issues pieces q = 8w .. 8w+7 of a slab (16 pieces of A, then 16 of W); lane L of a
    // piece fills LDS slot L = (row in the 8-row group, chunk position p): it must fetch chunk p ^ swz(row).
    // The address is split the way the instruction wants it: a wave-uniform 64-bit base per operand (advanced by
    // scalar adds, K step by K step) plus one 32-bit per-lane byte offset per piece, computed once — the main
    // loop then holds no vector address arithmetic at all, which matters because ordinary VALU instructions and
    // MFMAs share an issue port (every one of them is a bubble in the MFMA stream of a CU's other workgroup too).
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const char* baseA = (const char*)(a.A + (int64_t)m0 * a.lda + k_ofs);
    const char* baseW = (const char*)(a.W + (int64_t)n0 * a.K + k_ofs);
    uint32_t voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = ws * 8 + i, row = (q & 15) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = q < 16 ? (uint32_t)(min(row, a.M - 1 - m0) * a.lda + 4 * c) * 4u
                         : (uint32_t)(min(row, a.N - 1 - n0) * a.K + 4 * c) * 4u;
    }
    auto dma1 = [&](int i, int buf, int k0) {
        const int q = ws * 8 + i;
        const char* base = (q < 16 ? baseA : baseW) + (int64_t)k0 * 4;
        const uint32_t dst = (uint32_t)(uintptr_t)&lds[buf][q >> 4][(q & 15) * 8 * 32];   // low half of the flat address = LDS offset
        // scalar-base form by hand: the compiler's selection of the LDS-DMA intrinsic only produces 64-bit vector addresses
        // (M0 holds the LDS destination.  It is a reserved register the compiler does not track through a clobber
        // list; on gfx9 it has no other reader in this kernel — LDS instructions do not need it and nothing here
        // indexes registers dynamically — which tools/check_isa.py asserts on the compiled code.)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst), "v"(voff[i]), "s"(base) : "memory");
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = k_len / TK;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma1(i, 0, 0);

    TileEpi epi;                                          // residual and bias, fetched now (see tile_epilogue)
    if (!tail) tile_prefetch<EPI>(a, m0, n0, tid, epi);
    // Slab 0 only: loads return in issue order, so with the epilogue operands requested AFTER the slab the wave may go
    // on with those still in flight — they are 48 KB per tile against the slab's 32 KB, and waiting for them here cost
    // the K = 512 shapes 7-12 % (bias + residual against plain).  The first K step's vmcnt(0) meets them.
    // (the count must be exact — a larger one would let the slab's youngest load stay outstanding too)
    const bool res_early = EPI == EPI_PLAIN && !tail && a.res && m0 + TM <= a.M && n0 + TN <= a.N;
    if (res_early && a.bias)
        __builtin_amdgcn_s_waitcnt(0x0F70 | (EPI_EARLY + 1));   // vmcnt(13): bias4 + 12 residual row groups
    else if (res_early)
        __builtin_amdgcn_s_waitcnt(0x0F70 | EPI_EARLY);         // vmcnt(12)
    else
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __syncthreads();
    // Main loop.  Each K step is four groups of 16 MFMAs on one register set of fragments; the other set is
    // read from LDS four MFMAs into the group, so its latency sits under the remaining twelve.  The next
    // slab's DMA is requested during groups 0 and 1 into the other buffer (free since the previous step's
    // barrier); the step's barrier sits inside group 3, after this wave's last read of the current buffer,
    // and is followed by the first fragment read of the next buffer.
#ifdef VH_TILE_PROBE
    long long* tp = (long long*)&lds[w >> 1][w & 1][4096];   // 256 stamps per wave
    if (lane == 0) { tp[192] = clock64(); tp[193] = wall_clock64(); }
#endif
    const int swz = (r >> 1) & 7;
    f32x4 fa[2][2], fw[2][2];
    const float* fA[4];                                   // per-lane fragment addresses in buffer 0, one per k-quad pair t
    const float* fW[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fA[t] = &lds[0][0][(wm * 64 + r) * 32 + (((2 * t + h) ^ swz) << 2)];
        fW[t] = &lds[0][1][(wn * 64 + r) * 32 + (((2 * t + h) ^ swz) << 2)];
    }
    auto fload = [&](int set, int buf, int t) {           // buf is a compile-time constant at every call site
        fa[set][0] = ld4(fA[t] + buf * 2 * TM * LDS_LD);
        fa[set][1] = ld4(fA[t] + buf * 2 * TM * LDS_LD + 32 * 32);
        fw[set][0] = ld4(fW[t] + buf * 2 * TM * LDS_LD);
        fw[set][1] = ld4(fW[t] + buf * 2 * TM * LDS_LD + 32 * 32);
    };
    auto mfma4 = [&](int set, int j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][0][j], fw[set][0][j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][0][j], fw[set][1][j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][1][j], fw[set][0][j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][1][j], fw[set][1][j], acc[1][1], 0, 0, 0);
    };
    __builtin_amdgcn_s_setprio(0);
    fload(0, 0, 0);
    auto kstep = [&](int kt, auto cur_c, auto pf) {       // cur_c: buffer holding slab kt (compile-time)
        constexpr bool PF = decltype(pf)::value;
        constexpr int cur = decltype(cur_c)::value;
        VH_TP(0, kt);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            mfma4(t & 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF) {
                if (t < 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dma1(4 * t + i, cur ^ 1, (kt + 1) * TK);
                }
            }
            if (t < 3) {
                fload((t + 1) & 1, cur, t + 1);
            } else if constexpr (PF) {
                VH_TP(1, kt);
                __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this wave's DMA pieces (issued from asm) have landed
                __syncthreads();
                VH_TP(2, kt);
                fload(0, cur ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma4(t & 1, 1);
            mfma4(t & 1, 2);
            mfma4(t & 1, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        kstep(kt, B0{}, std::true_type{});
        kstep(kt + 1, B1{}, std::true_type{});
    }
    if (kt + 2 == nk) {
        kstep(kt, B0{}, std::true_type{});
        kstep(kt + 1, B1{}, std::false_type{});
    } else {
        kstep(kt, B0{}, std::false_type{});
    }
    if (!tail) tile_prefetch_late<EPI>(a, m0, n0, tid, epi);   // under the last MFMAs' drain and the epilogue's LDS transpose
    __syncthreads();
#ifdef VH_TILE_PROBE
    if (lane == 0) { tp[194] = clock64(); tp[195] = wall_clock64(); tp[196] = t_entry; }
    if (blockIdx.x < 8192) {
        for (int i = lane; i < 100; i += 64)                                  // 200 stamps = 100 x 16 B
            *(f32x4*)((float*)&vh_tile_probe[(blockIdx.x * 4 + w) * 200] + 4 * i) = ld4((const float*)tp + 4 * i);
    }
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        vh_tile_hwid[blockIdx.x * 2] = hw; vh_tile_hwid[blockIdx.x * 2 + 1] = xcc;
    }
    __syncthreads();
#endif

    __builtin_amdgcn_s_setprio(3);
    if (tail) {                                           // raw sums of this K slice -> its 128 x 128 slab
        GemmArgs p = a;
        p.ldo = TN;
        p.out = ts.ws + (int64_t)unit * TM * TN - ((int64_t)m0 * TN + n0);
        tile_epilogue<EPI_PARTIAL>(p, acc, &lds[0][0][0], m0, n0, tid, epi);
    } else {
        tile_epilogue<EPI, DROP>(a, acc, &lds[0][0][0], m0, n0, tid, epi);
    }
#ifdef VH_TILE_PROBE
    if ((tid & 63) == 0 && blockIdx.x < 8192) vh_tile_probe[(blockIdx.x * 4 + (tid >> 6)) * 200 + 197] = clock64();
#endif
}

// =============================================================================================
// Skinny kernel: M <= 16*MT rows.  Block = NW waves, owns 16 output columns n0..n0+15 over all K.
// wave w owns a contiguous run of k-steps of 16; lane: i = l&15 (W row / activation row), g = l>>4.
//
// The kernel is latency-bound (a few KB per wave), so it is written around ONE memory round trip:
//   1. issue every operand load of the wave's first chunk (CH k-steps: W from HBM, x/gamma/beta
//      from L2) before anything else;
//   2. while those are in flight compute the LayerNorm row statistics from a register-resident
//      copy of the rows (single read, two-pass mean / centred variance), publish through LDS;
//   3. normalise the activation fragments in registers, run the MFMAs, reduce over waves in LDS
//      (fixed order → bitwise reproducible), fused epilogue.
// =============================================================================================
template <int MT, int NW, int EPI, int CH, bool LN>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs a, LnFuse ln) {
    __shared__ __attribute__((aligned(16))) float red[NW][MT][64][4];
    __shared__ float s_mean[16 * MT], s_rstd[16 * MT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nks = a.K / 16;
    const int per_wave = (nks + NW - 1) / NW;
    const int ks0 = w * per_wave, ks1 = min(nks, ks0 + per_wave);
    const bool nin = n0 + i < a.N;
    const float* wrow = a.W + (int64_t)(n0 + i) * a.K + 4 * g;
    const float* xrow[MT];
    bool min_[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        min_[mt] = mt * 16 + i < a.M;
        xrow[mt] = a.A + (int64_t)(mt * 16 + i) * a.lda + 4 * g;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 wf[CH], xf[CH][MT], gm[CH], bt[CH];
    auto issue = [&](int base) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ks = base + c;
            const bool kin = ks < ks1;
            const int k = ks * 16;
            wf[c] = (kin && nin) ? ld4(wrow + k) : zero4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[c][mt] = (kin && min_[mt]) ? ld4(xrow[mt] + k) : zero4;
            if (LN) {
                gm[c] = kin ? ld4(ln.gamma + k + 4 * g) : zero4;
                bt[c] = kin ? ld4(ln.beta + k + 4 * g) : zero4;
            }
        }
    };
    STAMP(0);
    issue(ks0);
    STAMP(1);

    float mean[MT], rstd[MT];
    if (LN) {
        // rows w, w+NW, ...: each row is held in registers (K <= 1024) so it is read once; rows
        // are taken RG at a time so their loads overlap
        constexpr int RPW = (16 * MT + NW - 1) / NW;
        constexpr int RG = RPW < 4 ? RPW : 4;
#pragma unroll 1
        for (int r0 = 0; r0 < RPW; r0 += RG) {
            f32x4 xr[RG][4];
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                const int row = w + (r0 + rr) * NW;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = (lane + 64 * c) * 4;
                    xr[rr][c] = (row < a.M && col < a.K) ? ld4(a.A + (int64_t)row * a.lda + col) : zero4;
                }
            }
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                const int row = w + (r0 + rr) * NW;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) s += (xr[rr][c].x + xr[rr][c].y) + (xr[rr][c].z + xr[rr][c].w);
                const float mu = wave_sum(s) / (float)a.K;
                float ss = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if ((lane + 64 * c) * 4 < a.K) {
                        const f32x4 t = xr[rr][c] - mu;
                        ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
                    }
                }
                const float var = wave_sum(ss) / (float)a.K;
                if (lane == 0 && row < a.M) { s_mean[row] = mu; s_rstd[row] = rsqrtf(var + ln.eps); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            mean[mt] = min_[mt] ? s_mean[mt * 16 + i] : 0.f;
            rstd[mt] = min_[mt] ? s_rstd[mt * 16 + i] : 0.f;
        }
    }

    STAMP(2);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = zero4;
    for (int base = ks0; base < ks1; base += CH) {
        if (base != ks0) issue(base);
        if (LN) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 y = (xf[c][mt] - mean[mt]) * rstd[mt] * gm[c] + bt[c];
                    if (ln.ada_scale) {
                        const int k = (base + c) * 16 + 4 * g;
                        if (base + c < ks1) y = ld4(ln.ada_scale + k) * y + ld4(ln.ada_shift + k);
                    }
                    xf[c][mt] = (min_[mt] && base + c < ks1) ? y : zero4;
                }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][j], xf[c][mt][j], acc[mt], 0, 0, 0);
    }
    // D[i=n][j=m]: lane holds m = mt*16 + (l&15), n = n0 + 4g + {0..3}
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st4(&red[w][mt][lane][0], acc[mt]);
    STAMP(3);
    __syncthreads();
    STAMP(4);
    for (int o = tid; o < MT * 64; o += NW * 64) {
        const int mt = o >> 6, l = o & 63;
        f32x4 s = ld4(&red[0][mt][l][0]);
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) s += ld4(&red[ww][mt][l][0]);
        store4<EPI>(a, mt * 16 + (l & 15), n0 + 4 * (l >> 4), s);
    }
    STAMP(5);
}

// =============================================================================================
// Compact fast path of the skinny kernel for K = 16*NW*PW*passes (every real model shape).
// Every launch starts with a cold instruction cache and these kernels do only a few KB of work per
// wave, so executed code bytes are time: no predication (out-of-range rows are CLAMPED to a valid
// row — their products only reach output rows/columns that are never stored), immediate-offset
// loads, DPP reductions, LayerNorm statistics for 4 rows per wave at once (one DPP row per
// activation row, NJ float4 per lane) and a single LDS reduction.
// =============================================================================================
// LN: 0 = none; 1 = rows normalised in the operand load (statistics first, then the products);
//     2 = folded (vh_ln_fold): the products run on the raw rows against W∘gamma while the statistics
//         are computed beside them, and the epilogue applies rstd·(acc − mean·c1) + c2 — no
//         statistics → normalise → MFMA dependency, one workgroup barrier instead of two;
//     3 = folded, statistics taken from the OPERAND FRAGMENTS (MT = 1): no second read of the rows — the rows are
//         fresh data of the previous launch, written on other XCDs, and every KB a workgroup pulls of them costs
//         (profiles/r3_probe_launch_floor.log: ~16 GB/s per CU); one-pass sums about the row's first element.
//   W16: the weights are h16 (perf mode of the decode step, round 6): a fragment is 8 bytes per lane instead of 16 — half the
//   bytes of the stream that bounds these kernels — widened to fp32 in registers right before its four MFMAs.
template <int MT, int NW, int EPI, int PW, int LN, int NJ, bool W16 = false>
__device__ __forceinline__ void skinny_body(GemmArgs a, LnFuse ln, const HeadStep* hs = nullptr) {
    const int hr0 = (EPI == EPI_HEAD && gridDim.z > 1) ? (int)blockIdx.z * a.rg_rows : 0;   // first row of this group
    // Row groups (gridDim.z > 1): this workgroup owns rows [rg_rows·z, rg_rows·(z+1)) of the problem — it
    // pulls the same weights but only its share of the activation rows through its L1 (the rows are two
    // thirds of the bytes a 16-column workgroup loads).  Implemented by rebasing the row pointers.
    if (gridDim.z > 1) {
        const int r0 = blockIdx.z * a.rg_rows;
        a.A += (int64_t)r0 * a.lda;
        if (a.res) a.res += (int64_t)r0 * a.ldr;
        a.out += (int64_t)r0 * a.ldo;
        if (IS_QKV(EPI)) {                 // T == 1 (host check): row = batch index
            const int64_t off = (int64_t)r0 * a.n_heads * a.S_max * VH_HEAD_DIM / (EPI == EPI_QKV16 ? 2 : 1);   // bf16: half
            a.kc += off;
            a.vc += off;
            if (a.cache_len) a.cache_len += r0;
        }
        a.M = min(a.M - r0, a.rg_rows);
    }
    __shared__ __attribute__((aligned(16))) float red[NW][MT][64][4];
    __shared__ float s_mean[16 * MT], s_rstd[16 * MT];
    __shared__ __attribute__((aligned(16))) float pst[LN == 3 ? 16 : 1][2 * NW];   // LN == 3: (sum, sum of squares) per wave and row
    static_assert(LN != 3 || (MT == 1 && NW == 8), "fragment statistics: one row tile, eight waves");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int koff = blockIdx.y * a.k_len + w * (PW * 16) + 4 * g;   // this lane's first k in a pass
    const float* wp = a.W + (W16 ? 0 : (int64_t)min(n0 + i, a.N - 1) * a.K + koff);
    const uint16_t* wp16 = reinterpret_cast<const uint16_t*>(a.W) + (int64_t)min(n0 + i, a.N - 1) * a.K + koff;
    const float* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = a.A + (int64_t)min(mt * 16 + i, a.M - 1) * a.lda + koff;
    const float* gp = LN == 1 ? ln.gamma + koff : nullptr;
    const float* bp = LN == 1 ? ln.beta + koff : nullptr;

    f32x4 wf[PW], xf[PW][MT], gm[PW], bt[PW];
    uint2 wraw[PW];                                        // W16: the raw fragments (4 h16 each)
    auto issue = [&](int kbase) {
#pragma unroll
        for (int c = 0; c < PW; ++c) {
            if constexpr (W16) wraw[c] = *reinterpret_cast<const uint2*>(wp16 + kbase + 16 * c);
            else wf[c] = ld4(wp + kbase + 16 * c);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[c][mt] = ld4(xp[mt] + kbase + 16 * c);
            if (LN == 1) {
                gm[c] = ld4(gp + kbase + 16 * c);
                bt[c] = ld4(bp + kbase + 16 * c);
            }
        }
    };
    // Issue order = return order for s_waitcnt: the LayerNorm rows (L2 hits) go first so their
    // reduction overlaps the weight loads (HBM) issued right behind them; the sched_barrier keeps
    // hipcc from sinking loads between the MFMAs (it would trade round trips for registers).
    constexpr bool ROWSTATS = LN == 1 || LN == 2;           // statistics from their own loads of whole rows
    constexpr int ROUNDS = ROWSTATS ? (16 * MT + 4 * NW - 1) / (4 * NW) : 0;
    f32x4 v[NJ];
    auto ln_load = [&](int r0) {
        const int row = (r0 * NW + w) * 4 + g;
        const float* xr = a.A + (int64_t)min(row, a.M - 1) * a.lda + 4 * i;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) v[jj] = ld4(xr + 64 * jj);
    };
    auto ln_reduce = [&](int r0) {
        // DPP row g of the wave reduces activation row 4*w + g (two-pass mean / centred variance)
        const int row = (r0 * NW + w) * 4 + g;
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) sum += (v[jj].x + v[jj].y) + (v[jj].z + v[jj].w);
        const float mu = row16_sum(sum) / (float)a.K;
        float ss = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const f32x4 t = v[jj] - mu;
            ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        }
        const float var = row16_sum(ss) / (float)a.K;
        if (i == 0 && row < a.M) { s_mean[row] = mu; s_rstd[row] = rsqrtf(var + ln.eps); }
    };
    STAMP(0);
    if (ROWSTATS) ln_load(0);
    float shift = 0.f, fsa = 0.f, fsb = 0.f;               // LN == 3: statistics about the row's first element
    if (LN == 3) shift = a.A[(int64_t)min(i, a.M - 1) * a.lda];
    issue(0);
    // Epilogue operands (bias / residual / cache position) are fetched NOW by the lanes that will
    // finalise (wave w finalises m-tile w): loaded in the epilogue they would add one more
    // dependent memory round trip to a kernel that is nothing but round trips.
    const int em = w * 16 + i, en = n0 + 4 * g;
    const bool fin = w < MT && em < a.M && en + 3 < a.N;
    f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_res = {0.f, 0.f, 0.f, 0.f}, e_c1 = {0.f, 0.f, 0.f, 0.f};
    int e_pos = 0;
    if (fin) {
        if (LN >= 2) {                       // c2 carries the bias
            e_c1 = ld4(ln.c1 + en);
            e_bias = ld4(ln.c2 + en);
        } else if (EPI == EPI_PLAIN && a.bias) e_bias = ld4(a.bias + en);
        if (EPI == EPI_PLAIN && a.res) e_res = ld4(a.res + (int64_t)em * a.ldr + en);
        if (IS_QKV(EPI) && a.cache_len) e_pos = a.cache_len[em / a.T];
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);

    float mean[MT], rstd[MT];
    if (ROWSTATS) {
        ln_reduce(0);
#pragma unroll 1
        for (int r0 = 1; r0 < ROUNDS; ++r0) {
            ln_load(r0);
            ln_reduce(r0);
        }
    }
    if (LN == 1) {
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = min(mt * 16 + i, a.M - 1);
            mean[mt] = s_mean[row];
            rstd[mt] = s_rstd[row];
        }
    }

    STAMP(2);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pass_stride = NW * PW * 16;
#pragma unroll 1
    for (int kbase = 0;;) {
        if (LN == 1) {
#pragma unroll
            for (int c = 0; c < PW; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 sc = gm[c] * rstd[mt];
                    xf[c][mt] = xf[c][mt] * sc + (bt[c] - sc * mean[mt]);
                    if (ln.ada_scale)
                        xf[c][mt] = ld4(ln.ada_scale + koff + kbase + 16 * c) * xf[c][mt] +
                                    ld4(ln.ada_shift + koff + kbase + 16 * c);
                }
        }
        if (LN == 3) {
#pragma unroll
            for (int c = 0; c < PW; ++c) {
                const f32x4 t = xf[c][0] - shift;
                fsa += (t.x + t.y) + (t.z + t.w);
                fsb += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
            }
        }
#pragma unroll
        for (int c = 0; c < PW; ++c) {
            if constexpr (W16) wf[c] = f32x4{vh_h16_lo(wraw[c].x), vh_h16_hi(wraw[c].x), vh_h16_lo(wraw[c].y), vh_h16_hi(wraw[c].y)};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][jj], xf[c][mt][jj], acc[mt], 0, 0, 0);
        }
        kbase += pass_stride;
        if (kbase >= a.k_len) break;
        issue(kbase);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st4(&red[w][mt][lane][0], acc[mt]);
    if (LN == 3) {      // fold the wave's four k groups (lanes i, i+16, i+32, i+48), one pair per wave and row
        fsa += __shfl_xor(fsa, 16, 64); fsb += __shfl_xor(fsb, 16, 64);
        fsa += __shfl_xor(fsa, 32, 64); fsb += __shfl_xor(fsb, 32, 64);
        if (g == 0) { pst[i][2 * w] = fsa; pst[i][2 * w + 1] = fsb; }
    }
    STAMP(3);
    __syncthreads();
    STAMP(4);
    if (tid < MT * 64) {
        const int mt = tid >> 6;
        f32x4 sacc = ld4(&red[0][mt][lane][0]);
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) sacc += ld4(&red[ww][mt][lane][0]);
        if (LN == 3 && fin) {                // the 8 waves' partial sums, added in wave order
            f32x4 p[NW / 2];
#pragma unroll
            for (int q4 = 0; q4 < NW / 2; ++q4) p[q4] = ld4(&pst[i][4 * q4]);
            float sa = p[0].x, sb = p[0].y;
            sa += p[0].z; sb += p[0].w;
#pragma unroll
            for (int q4 = 1; q4 < NW / 2; ++q4) { sa += p[q4].x; sb += p[q4].y; sa += p[q4].z; sb += p[q4].w; }
            const float dm = sa / (float)a.K;
            const float var = fmaxf(sb / (float)a.K - dm * dm, 0.f);
            sacc = (sacc - (shift + dm) * e_c1) * rsqrtf(var + ln.eps);
            if (IS_QKV(EPI)) sacc += e_bias;
        }
        if (LN == 2 && fin) {                // statistics were published before the barrier above
            const float mu = s_mean[em], rs = s_rstd[em];
            sacc = (sacc - mu * e_c1) * rs;  // e_bias = c2 is added below
            if (IS_QKV(EPI)) sacc += e_bias;
        }
        if constexpr (EPI == EPI_HEAD) {
            // the logits (no bias, activation or residual: valle_ar.py:29,158) and this workgroup's candidate per row:
            // the largest of its 16 columns, lowest column on ties (greedy_step_kernel's rule); -0 counts as +0
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            if (em < a.M) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (en + jj < a.N) {
                        a.out[(int64_t)em * a.ldo + en + jj] = sacc[jj];
                        if (sacc[jj] > bv) { bv = sacc[jj]; bi = en + jj; }
                    }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {               // the four column quads of row i sit in lanes i + 16 g
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (g == 0 && em < a.M)
                __hip_atomic_store(hs->cand + (int64_t)(hr0 + em) * gridDim.x + blockIdx.x,
                                   ((uint64_t)__float_as_uint(bv) << 32) | (uint32_t)bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!fin || EPI == EPI_PARTIAL) {
            store4<EPI>(a, em, en, sacc);          // ragged last column group / raw partial
        } else if (EPI == EPI_PLAIN) {
            sacc += e_bias;
            if (a.act == VH_ACT_GELU_ERF) {
                sacc.x = gelu_erf(sacc.x); sacc.y = gelu_erf(sacc.y);
                sacc.z = gelu_erf(sacc.z); sacc.w = gelu_erf(sacc.w);
            }
            st4(a.out + (int64_t)em * a.ldo + en, sacc + e_res);
        } else {  // EPI_QKV with the cache position already in a register
            const int which = en / a.d_model, c = en - which * a.d_model;
            if (which == 0) {
                st4(a.out + (int64_t)em * a.ldo + c, sacc);
            } else {
                const int head = c / VH_HEAD_DIM, e = c - head * VH_HEAD_DIM;
                const int b = em / a.T, t = em - b * a.T;
                float* base = which == 1 ? a.kc : a.vc;
                const int64_t at = (((int64_t)b * a.n_heads + head) * a.S_max + e_pos + t) * VH_HEAD_DIM + e;
                if (EPI == EPI_QKV16) {           // four bf16 = one 8-byte store
                    const uint64_t lo = vh_pack_h16(sacc.x, sacc.y);
                    const uint64_t hi = vh_pack_h16(sacc.z, sacc.w);
                    *reinterpret_cast<uint64_t*>(reinterpret_cast<uint16_t*>(base) + at) = lo | (hi << 32);
                } else {
                    st4(base + at, sacc);
                }
            }
        }
    }
    STAMP(5);
    if constexpr (EPI == EPI_HEAD) {
        // The greedy step without a second launch: every workgroup has published its candidates (write-through stores,
        // counted in vmcnt) and takes a ticket on its row group's counter; the LAST one to arrive picks every row's token
        // from the gridDim.x candidates, does the EOS bookkeeping and the append, and builds the next step's input rows
        // (what greedy_step_kernel does).  Nobody waits for anybody.  x_next may be the operand A itself: a row group's
        // rows are read by its own workgroups only, and all of them are past their loads once the last ticket is taken.
        static_assert(MT == 1 && NW == 8, "head + greedy step: 16 rows x 32 threads");
        __shared__ int s_ticket, s_tok[16], s_pos[16];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the ticket is the hand-over: release (this workgroup's candidates) and acquire (everybody else's) at agent scope, so
        // that the last arriver's view rests on the memory model and not on write-through stores + vmcnt alone (one fence
        // pair per workgroup; this opt-in form is the slower one either way)
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(hs->counters + blockIdx.z, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != (int)gridDim.x - 1) return;
        if (tid == 0) hs->counters[blockIdx.z] = 0;          // ready for the next launch (graph replay)
        const int row = tid >> 5, j = tid & 31;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        if (row < a.M)
            for (int c = j; c < (int)gridDim.x; c += 32) {
                const uint64_t p = __hip_atomic_load(hs->cand + (int64_t)(hr0 + row) * gridDim.x + c, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                const float v = __uint_as_float((uint32_t)(p >> 32));
                const int idx = (int)(uint32_t)p;
                if (v > bv || (v == bv && idx < bi)) { bv = v; bi = idx; }
            }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (j == 0 && row < a.M) {
            const int b = hr0 + row;
            const int pos = hs->audio_pos[b];
            int64_t* crow = hs->codes + (int64_t)b * hs->codes_stride;
            int tok = min(bi, hs->V - 1);                     // a row of NaNs has no candidate: stay inside the table
            if (crow[pos - 1] == (int64_t)hs->eos) tok = hs->eos;   // valle_ar.py:168
            crow[pos] = tok;                                  // valle_ar.py:171
            if (tok == hs->eos) atomicAdd(&hs->eos_count[pos - (hs->pos_base ? hs->pos_base[b] : 0)], 1);
            s_tok[row] = tok;
            s_pos[row] = pos;
            hs->audio_pos[b] = pos + 1;
            hs->cache_len[b] += 1;
        }
        __syncthreads();
        const int q4 = hs->d >> 2;
        for (int c = tid; c < a.M * q4; c += NW * 64) {       // valle_ar.py:143-144 for the next step (modules.py:337)
            const int r = c / q4, cc = (c - r * q4) * 4;
            st4(hs->x_next + (int64_t)(hr0 + r) * hs->d + cc,
                ld4(hs->audio_emb + (int64_t)s_tok[r] * hs->d + cc) + ld4(hs->pe + (int64_t)s_pos[r] * hs->d + cc));
        }
    }
}


template <int MT, int NW, int EPI, int PW, int LN, int NJ, bool W16 = false>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_fast(const float* hA, const float* hW, int h_lda, int hK, int h_klen,
                                                            int hM, int hN, GemmArgs a, LnFuse ln) {
    // The operands every wave needs for its first loads travel as leading scalar arguments: with
    // -mllvm -amdgpu-kernarg-preload-count the command processor places them in SGPRs at dispatch, so the
    // weight / activation loads are issued without first waiting for a kernarg s_load round trip.
    a.A = hA; a.W = hW; a.lda = h_lda; a.K = hK; a.k_len = h_klen; a.M = hM; a.N = hN;
    skinny_body<MT, NW, EPI, PW, LN, NJ, W16>(a, ln);
}

// The AR head + greedy step (EPI_HEAD): same body, the step's operands as one more by-value argument.
template <int PW>
__global__ __launch_bounds__(512) void gemm_head_greedy_kernel(const float* hA, const float* hW, int h_lda, int hK, int h_klen,
                                                               int hM, int hN, GemmArgs a, HeadStep hs) {
    a.A = hA; a.W = hW; a.lda = h_lda; a.K = hK; a.k_len = h_klen; a.M = hM; a.N = hN;
    LnFuse none{};
    skinny_body<1, 8, EPI_HEAD, PW, 0, 1>(a, none, &hs);
}
// =============================================================================================
// Split-K for the wide-K skinny GEMM (linear_2: K = dff).  One workgroup can pull only ~20-30 GB/s
// through its L1, and with K = 2048 a 16-column workgroup needs the whole (M, K) activation block
// (256 KB) plus 128 KB of weights: the launch is bound by per-CU fill rate, not by HBM.  So the K
// range is cut into gridDim.y slices (≈256 workgroups in all, each W 16 KB + x 32 KB), the slices
// leave raw partial sums in a workspace and a second tiny kernel adds them IN FIXED ORDER (bitwise
// reproducible, no atomics) and applies the bias / activation / residual epilogue.
// =============================================================================================
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int n_split,
                                                            GemmArgs a, int lds_) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ngroups = lds_ / 4;
    const int m = idx / ngroups, n = (idx - m * ngroups) * 4;
    if (m >= a.M || n >= a.N) return;
    const float* p = slabs + (int64_t)m * lds_ + n;
    const int64_t stride = (int64_t)a.M * lds_;
    f32x4 part[16];                     // all slices in flight at once, summed in slice order
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx)
        part[sidx] = sidx < n_split ? ld4(p + sidx * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
    const bool full = n + 3 < a.N;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, res4 = {0.f, 0.f, 0.f, 0.f};
    if (full && a.bias) bias4 = ld4(a.bias + n);
    if (full && a.res) res4 = ld4(a.res + (int64_t)m * a.ldr + n);
    f32x4 acc = part[0];
#pragma unroll
    for (int sidx = 1; sidx < 16; ++sidx) acc += part[sidx];
    if (!full) { store4<EPI_PLAIN>(a, m, n, acc); return; }
    acc += bias4;
    if (a.act == VH_ACT_GELU_ERF) {
        acc.x = gelu_erf(acc.x); acc.y = gelu_erf(acc.y); acc.z = gelu_erf(acc.z); acc.w = gelu_erf(acc.w);
    }
    st4(a.out + (int64_t)m * a.ldo + n, acc + res4);
}

// The tail tiles of a TailSplit launch: sum of the K slices in slice order + the EPI_PLAIN epilogue (bias, pre-activation
// copy, GELU / GELU', residual, column sums).  One workgroup per (tail tile, band of 32 rows); thread (erow = tid >> 5,
// ec4 = tid & 31) owns 4 columns of rows erow + 8 it.  N % 128 == 0 (the host only splits then), M may end inside a tile.
__global__ __launch_bounds__(256) void tile_tail_fixup_kernel(GemmArgs a, TailSplit ts, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float ct[8 * TN];
    const int tid = threadIdx.x, ec4 = tid & 31, erow = tid >> 5;
    const int t = blockIdx.x >> 2, band = blockIdx.x & 3;
    const int tile = ts.n_whole + t;
    const int m0 = (tile / tiles_n) * TM + band * 32, n0 = (tile % tiles_n) * TN, en = n0 + 4 * ec4;
    const float* slab = ts.ws + (int64_t)t * ts.split * TM * TN + (band * 32 + erow) * TN + 4 * ec4;
    const f32x4 bias4 = a.bias ? ld4(a.bias + en) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int m = m0 + erow + 8 * it;
        if (m >= a.M) break;
        f32x4 v = ld4(slab + it * 8 * TN);
        for (int c = 1; c < ts.split; ++c) v += ld4(slab + (int64_t)c * TM * TN + it * 8 * TN);
        v += bias4;
        f32x4 dm = {1.f, 1.f, 1.f, 1.f};
        if (a.drop.thresh) dm = vh_dropmul4(a.drop, (uint32_t)m, (uint32_t)(en >> 2));
        if (a.act == VH_ACT_GELU_ERF_D) {
            f32x4 dv;
            v = gelu_and_grad4(v, dv);
            st4(a.aux + (int64_t)m * a.ldx + en, dv * dm);
        } else if (a.aux) {
            st4(a.aux + (int64_t)m * a.ldx + en, v);
        }
        if (a.act == VH_ACT_GELU_ERF) {
            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
        }
        if (a.drop.thresh) v = v * dm;
        if (a.act == VH_ACT_GELU_BWD) {
            const f32x4 p = ld4(a.res + (int64_t)m * a.ldr + en);
            v = f32x4{v.x * gelu_grad(p.x), v.y * gelu_grad(p.y), v.z * gelu_grad(p.z), v.w * gelu_grad(p.w)};
        } else if (a.act == VH_ACT_MUL) {
            v = v * ld4(a.res + (int64_t)m * a.ldr + en);
        } else if (a.res) {
            v += ld4(a.res + (int64_t)m * a.ldr + en);
        }
        st4(a.out + (int64_t)m * a.ldo + en, v);
        csum += v;
    }
    if (a.colsum) {
        st4(ct + erow * TN + 4 * ec4, csum);
        __syncthreads();
        if (tid < TN) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += ct[r * TN + tid];
            atomicAdd(a.colsum + n0 + tid, s);
        }
    }
}

// K slices per tail tile (0 = no tail split) and the number of whole tiles in front of them
static int tail_plan(int M, int N, int K, int* n_whole) {
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    const int r = tiles % 256;
    *n_whole = tiles;
    if (tiles < 256 || r == 0 || r > 128 || N % TN != 0 || K % TK != 0 || vh_tuning(VH_TUNE_TAIL_SPLIT) == 1) return 0;
    int split = min(8, 256 / r);
    while (split >= 2 && (K % (split * TK) != 0 || K / split < 128)) --split;
    if (split < 2) return 0;
    *n_whole = tiles - r;
    return split;
}

extern "C" size_t vh_linear_ex_ws_bytes(int M, int N, int K) {
    int n_whole;
    const int split = tail_plan(M, N, K, &n_whole);
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    return split ? (size_t)(tiles - n_whole) * split * TM * TN * sizeof(float) : 0;
}

static int splitk_plan(int M, int N, int K) {
    // number of K slices (0 = do not split)
    if (M > 64) {
        // tile kernel: too few 128x128 tiles for the 256 CUs (a single utterance through the NAR stack, a short
        // prefill) and a long K: cut K so that the grid approaches two workgroups per CU
        const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
        if (tiles > 256 || K < 1024) return 0;
        int splits = 1;
        while (splits < 8 && tiles * splits * 2 <= 512 && K % (splits * 2 * TK) == 0 && K / (splits * 2) >= 256) splits *= 2;
        return splits >= 2 ? splits : 0;
    }
    if (K <= 1024 || K % 256 != 0) return 0;
    const int tiles = (N + 15) / 16;
    int splits = min(16, 256 / tiles);
    while (splits > 1 && (K % (splits * 256) != 0)) --splits;
    return splits >= 2 ? splits : 0;
}

extern "C" size_t vh_linear_ws_bytes(int M, int N, int K) {
    const int splits = splitk_plan(M, N, K);
    if (splits) return (size_t)splits * M * ((N + 3) / 4 * 4) * sizeof(float);
    return M > 64 ? vh_linear_ex_ws_bytes(M, N, K) : 0;      // the tile kernel's tail split (see TailSplit)
}

// =============================================================================================
// host dispatch
// =============================================================================================
static int check_gemm(const char* name, const GemmArgs& a, const LnFuse& ln) {
    VH_REQUIRE(a.A && a.W && a.out, VH_EINVAL, "%s: null pointer", name);
    VH_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, VH_EINVAL, "%s: bad dims M=%d N=%d K=%d", name, a.M,
               a.N, a.K);
    VH_REQUIRE(a.K % 16 == 0, VH_EUNSUPPORTED, "%s: K=%d must be a multiple of 16", name, a.K);
    VH_REQUIRE(a.lda % 4 == 0 && a.lda >= a.K, VH_EALIGN, "%s: lda=%d", name, a.lda);
    VH_REQUIRE(vh_aligned16(a.A) && vh_aligned16(a.W) && vh_aligned16(a.out) &&
                   vh_aligned16(a.bias) && vh_aligned16(a.res),
               VH_EALIGN, "%s: pointers must be 16-byte aligned", name);
    VH_REQUIRE(a.ldo % 4 == 0 && (!a.res || a.ldr % 4 == 0), VH_EALIGN, "%s: ldo/ldr", name);
    VH_REQUIRE((ln.gamma == nullptr) == (ln.beta == nullptr), VH_EINVAL,
               "%s: ln_gamma and ln_beta must be given together", name);
    VH_REQUIRE((ln.ada_scale == nullptr) == (ln.ada_shift == nullptr), VH_EINVAL,
               "%s: ada_scale and ada_shift must be given together", name);
    VH_REQUIRE(!ln.ada_scale || ln.gamma, VH_EINVAL, "%s: ada_* needs ln_gamma/ln_beta", name);
    VH_REQUIRE(!ln.gamma || (a.M <= 64 && a.K <= 1024), VH_EUNSUPPORTED,
               "%s: fused LayerNorm only for M <= 64 and K <= 1024 (M=%d K=%d); run vh_layernorm first",
               name, a.M, a.K);
    VH_REQUIRE(vh_aligned16(ln.gamma) && vh_aligned16(ln.beta) && vh_aligned16(ln.ada_scale) &&
                   vh_aligned16(ln.ada_shift),
               VH_EALIGN, "%s: LayerNorm vectors must be 16-byte aligned", name);
    return VH_OK;
}

#define SKINNY_ARGS(g) (g).A, (g).W, (g).lda, (g).K, (g).k_len, (g).M, (g).N

template <int EPI>
static int launch_gemm(const char* name, const GemmArgs& a, const LnFuse& ln, hipStream_t s, float* tail_ws = nullptr,
                       size_t tail_ws_bytes = 0) {
    if (a.M == 0) return VH_OK;
    const bool train_epi = a.aux != nullptr || a.colsum != nullptr || a.act >= VH_ACT_GELU_BWD || a.drop.thresh != 0;   // tile-kernel epilogues only
    if (a.M <= 64 && !train_epi) {
        const int mt = (a.M + 15) / 16;
        dim3 grid((a.N + 15) / 16);
        const bool wide = a.K > 1024;
        const bool has_ln = ln.gamma != nullptr || ln.c1 != nullptr;
        const bool fold = ln.c1 != nullptr;
        // one workgroup per (16 columns, 16 rows) instead of (16 columns, all rows): see the kernel
        const bool rowgroups = vh_tuning(VH_TUNE_ROW_GROUPS) != 2 && mt >= 2 && (!wide || a.K % 2048 == 0) &&
                               EPI != EPI_PARTIAL && (!IS_QKV(EPI) || a.T == 1);
        GemmArgs ag = a;                  // groups of 16 rows, or of 8 while that keeps the grid within the CUs
        if (rowgroups) {
            ag.rg_rows = (vh_tuning(VH_TUNE_ROW_GROUPS) != 3 && (int)grid.x * ((a.M + 7) / 8) <= 256) ? 8 : 16;
            grid.z = (a.M + ag.rg_rows - 1) / ag.rg_rows;
        }
        // ---- compact fast path: K = 16*NW*PW*passes
#define SF(MT, NW, PW, LN, NJ)                                                                                                   \
    do {                                                                                                                        \
        if constexpr ((EPI == EPI_PLAIN && (LN) == 0 && (NW) == 8) || (EPI == EPI_QKV16 && (LN) >= 2)) {                          \
            if (ag.w16) {                                                                                                       \
                hipLaunchKernelGGL((gemm_skinny_fast<MT, NW, EPI, PW, LN, NJ, true>), grid, dim3(NW * 64), 0, s, SKINNY_ARGS(ag), ag, ln); \
                break;                                                                                                          \
            }                                                                                                                   \
        }                                                                                                                       \
        if (ag.w16) { vh_set_error("%s: no 16-bit-weight form of this shape", name); return VH_EUNSUPPORTED; }                   \
        hipLaunchKernelGGL((gemm_skinny_fast<MT, NW, EPI, PW, LN, NJ>), grid, dim3(NW * 64), 0, s, SKINNY_ARGS(ag), ag, ln);        \
    } while (0)
#define SF_MT(NW, PW, LN, NJ)                                                  \
    do {                                                                       \
        if (rowgroups) SF(1, NW, PW, LN, NJ);                                  \
        else if (mt == 1) SF(1, NW, PW, LN, NJ);                               \
        else if (mt == 2) SF(2, NW, PW, LN, NJ);                               \
        else SF(4, NW, PW, LN, NJ);                                            \
        VH_CHECK_LAUNCH(name);                                                 \
        return VH_OK;                                                          \
    } while (0)
        if (has_ln) {  // K <= 1024 (check_gemm); statistics need K = 64*NJ
            if (fold && (rowgroups || mt == 1) && vh_tuning(VH_TUNE_LN_STATS) != 1) {   // statistics from the fragments
#define SF3(NW, PW) do { SF(1, NW, PW, 3, 1); VH_CHECK_LAUNCH(name); return VH_OK; } while (0)
                if (a.K == 128) SF3(8, 1);
                if (a.K == 256) SF3(8, 2);
                if (a.K == 512) SF3(8, 4);
                if (a.K == 1024) SF3(8, 4);
#undef SF3
            }
            if (fold) {
                if (a.K == 128) SF_MT(8, 1, 2, 2);
                if (a.K == 256) SF_MT(8, 2, 2, 4);
                if (a.K == 512) SF_MT(8, 4, 2, 8);
                if (a.K == 1024) SF_MT(8, 4, 2, 16);
                vh_set_error("%s: folded LayerNorm needs K in {128,256,512,1024} (K=%d)", name, a.K);
                return VH_EUNSUPPORTED;
            }
            if (a.K == 128) SF_MT(8, 1, 1, 2);
            if (a.K == 256) SF_MT(8, 2, 1, 4);
            if (a.K == 512) SF_MT(8, 4, 1, 8);
            if (a.K == 1024) SF_MT(8, 4, 1, 16);
        } else if (!wide) {
            if (a.K % 512 == 0) SF_MT(8, 4, 0, 1);
            if (a.K % 256 == 0) SF_MT(8, 2, 0, 1);
            if (a.K % 128 == 0) SF_MT(8, 1, 0, 1);
        } else {
            if (a.K % 2048 == 0 && (mt <= 2 || rowgroups)) {
                if (mt == 1 || rowgroups) SF(1, 16, 8, 0, 1); else SF(2, 16, 8, 0, 1);
                VH_CHECK_LAUNCH(name);
                return VH_OK;
            }
            if (a.K % 1024 == 0) SF_MT(16, 4, 0, 1);
        }
#undef SF_MT
#undef SF
        if constexpr (EPI == EPI_QKV16) {
            vh_set_error("%s: bf16 K/V append needs the folded LayerNorm shapes (K in {128,256,512,1024})", name);
            return VH_EUNSUPPORTED;
        } else {
        // ---- generic guarded kernel for every other K (multiple of 16)
        if (a.w16) { vh_set_error("%s: no 16-bit-weight form of this shape (K=%d)", name, a.K); return VH_EUNSUPPORTED; }
#define SK(MT, NW, CH, LN) \
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NW, EPI, CH, LN>), grid, dim3(NW * 64), 0, s, a, ln)
        if (has_ln) {
            if (mt == 1) SK(1, 8, 4, true); else if (mt == 2) SK(2, 8, 4, true); else SK(4, 8, 4, true);
        } else if (wide) {
            if (mt == 1) SK(1, 16, 8, false); else if (mt == 2) SK(2, 16, 8, false); else SK(4, 16, 4, false);
        } else {
            if (mt == 1) SK(1, 8, 4, false); else if (mt == 2) SK(2, 8, 4, false); else SK(4, 8, 4, false);
        }
#undef SK
        }
    } else if constexpr (EPI == EPI_QKV16) {
        vh_set_error("%s: bf16 K/V append is the decode path (M <= 64)", name);
        return VH_EUNSUPPORTED;
    } else {
        if (a.w16) { vh_set_error("%s: 16-bit weights are the decode step's form (M <= 64)", name); return VH_EUNSUPPORTED; }
        const int tm = (a.M + TM - 1) / TM, tn = (a.N + TN - 1) / TN;
        // The LDS-DMA kernel needs whole 32-wide K slabs; a ragged K goes to the register-staged kernel.
        if (a.k_len % TK == 0 && vh_tuning(VH_TUNE_TILE_DMA) != 1) {
            TailSplit ts{tm * tn, 1, a.K, nullptr};
            if constexpr (EPI == EPI_PLAIN) {
                int n_whole;
                const int split = tail_ws ? tail_plan(a.M, a.N, a.K, &n_whole) : 0;
                if (split && tail_ws_bytes >= (size_t)(tm * tn - n_whole) * split * TM * TN * sizeof(float))
                    ts = TailSplit{n_whole, split, a.K / split, tail_ws};
            }
            const int tail = tm * tn - ts.n_whole;
            if (EPI == EPI_PLAIN && a.drop.thresh)
                hipLaunchKernelGGL((gemm_tile_dma_kernel<EPI, EPI == EPI_PLAIN>), dim3(ts.n_whole + tail * ts.split), dim3(256), 0,
                                   s, a, tm, tn, ts);
            else
                hipLaunchKernelGGL((gemm_tile_dma_kernel<EPI>), dim3(ts.n_whole + tail * ts.split), dim3(256), 0, s, a, tm, tn,
                                   ts);
            if (tail) hipLaunchKernelGGL(tile_tail_fixup_kernel, dim3(tail * 4), dim3(256), 0, s, a, ts, tn);
        } else {
            hipLaunchKernelGGL((gemm_tile_kernel<EPI>), dim3(tm * tn), dim3(256), 0, s, a, tm, tn);
        }
    }
    VH_CHECK_LAUNCH(name);
    return VH_OK;
}

extern "C" int vh_linear(const float* A, int lda, const float* W, const float* bias,
                         const float* residual, int ldr, float* out, int ldo, int M, int N, int K,
                         int act, const float* ln_gamma, const float* ln_beta,
                         const float* ada_scale, const float* ada_shift, float ln_eps,
                         void* stream) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.bias = bias; a.res = residual; a.ldr = ldr; a.out = out;
    a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.act = act; a.k_len = K;
    LnFuse ln{ln_gamma, ln_beta, ada_scale, ada_shift, ln_eps};
    VH_REQUIRE(act == VH_ACT_NONE || act == VH_ACT_GELU_ERF, VH_EINVAL, "vh_linear: act=%d", act);
    VH_REQUIRE(ldo >= N && (!residual || ldr >= N), VH_EINVAL, "vh_linear: ldo/ldr < N");
    if (int rc = check_gemm("vh_linear", a, ln)) return rc;
    return launch_gemm<EPI_PLAIN>("vh_linear", a, ln, (hipStream_t)stream);
}

extern "C" int vh_linear_ex(const float* A, int lda, const float* W, const float* bias, const float* residual,
                            int ldr, float* out, int ldo, float* pre_out, int ldp, float* dcolsum, int M, int N, int K,
                            int act, const vh_dropout_spec* drop, void* workspace, size_t workspace_bytes, void* stream) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.bias = bias; a.res = residual; a.ldr = ldr; a.out = out;
    a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.act = act; a.k_len = K; a.aux = pre_out; a.ldx = ldp;
    a.colsum = dcolsum;
    VH_REQUIRE(!dcolsum || N % 128 == 0, VH_EUNSUPPORTED, "vh_linear_ex: dcolsum needs N %% 128 == 0 (N=%d)", N);
    VH_REQUIRE(VH_DROP_OK(drop), VH_EINVAL, "vh_linear_ex: dropout p must be in [0, 1)");
    if (vh_drop_args(drop, &a.drop)) {
        VH_REQUIRE(act == VH_ACT_NONE || act == VH_ACT_GELU_ERF || act == VH_ACT_GELU_ERF_D, VH_EINVAL,
                   "vh_linear_ex: dropout goes with act NONE / GELU_ERF / GELU_ERF_D (act=%d)", act);
        VH_REQUIRE(N % 4 == 0 && K % TK == 0 && vh_tuning(VH_TUNE_TILE_DMA) != 1, VH_EUNSUPPORTED,
                   "vh_linear_ex: dropout needs N %% 4 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
    }
    LnFuse ln{};
    VH_REQUIRE(act >= VH_ACT_NONE && act <= VH_ACT_MUL, VH_EINVAL, "vh_linear_ex: act=%d", act);
    VH_REQUIRE((act != VH_ACT_GELU_BWD && act != VH_ACT_MUL) || (residual && !bias && !pre_out), VH_EINVAL,
               "vh_linear_ex: VH_ACT_GELU_BWD / VH_ACT_MUL take the saved pre-activation / derivative as `residual`, no bias, "
               "no pre_out");
    VH_REQUIRE(act != VH_ACT_GELU_ERF_D || (pre_out && !residual), VH_EINVAL,
               "vh_linear_ex: VH_ACT_GELU_ERF_D stores the derivative in pre_out (required) and takes no residual");
    VH_REQUIRE(ldo >= N && (!residual || ldr >= N) && (!pre_out || (ldp >= N && ldp % 4 == 0 && vh_aligned16(pre_out))),
               VH_EINVAL, "vh_linear_ex: ldo/ldr/ldp");
    VH_REQUIRE(vh_aligned16(workspace), VH_EALIGN, "vh_linear_ex: workspace must be 16-byte aligned");
    if (int rc = check_gemm("vh_linear_ex", a, ln)) return rc;
    return launch_gemm<EPI_PLAIN>("vh_linear_ex", a, ln, (hipStream_t)stream, (float*)workspace, workspace_bytes);
}

extern "C" int vh_linear_qkv(const float* A, int lda, const float* Wqkv, float* q_out, int ldq,
                             float* kcache, float* vcache, const int32_t* cache_len, int B, int T,
                             int d_model, int n_heads, int S_max, const float* ln_gamma,
                             const float* ln_beta, const float* ada_scale, const float* ada_shift,
                             float ln_eps, void* stream) {
    VH_REQUIRE(kcache && vcache, VH_EINVAL, "vh_linear_qkv: null cache");
    VH_REQUIRE(B >= 0 && T >= 0 && n_heads > 0 && d_model == n_heads * VH_HEAD_DIM, VH_EUNSUPPORTED,
               "vh_linear_qkv: d_model=%d must equal n_heads=%d x %d", d_model, n_heads, VH_HEAD_DIM);
    VH_REQUIRE(S_max >= T && ldq >= d_model, VH_EINVAL, "vh_linear_qkv: S_max=%d < T=%d or ldq", S_max, T);
    VH_REQUIRE(vh_aligned16(kcache) && vh_aligned16(vcache), VH_EALIGN, "vh_linear_qkv: cache alignment");
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = Wqkv; a.out = q_out; a.ldo = ldq; a.M = B * T; a.N = 3 * d_model;
    a.K = d_model; a.k_len = d_model; a.act = VH_ACT_NONE; a.kc = kcache; a.vc = vcache; a.cache_len = cache_len;
    a.T = T > 0 ? T : 1; a.S_max = S_max; a.d_model = d_model; a.n_heads = n_heads;
    LnFuse ln{ln_gamma, ln_beta, ada_scale, ada_shift, ln_eps};
    if (int rc = check_gemm("vh_linear_qkv", a, ln)) return rc;
    return launch_gemm<EPI_QKV>("vh_linear_qkv", a, ln, (hipStream_t)stream);
}

// ---- LayerNorm folded into the weight operand (decode path) -----------------------------------
// LN(x)·Wᵀ + b = rstd·(x·(W∘γ)ᵀ − mean·c1) + c2 with c1[n] = Σ_k γ_k W[n,k], c2[n] = Σ_k β_k W[n,k] + b[n].
// One wave per output row n; the two sums in double (one-off preparation, rounded once).
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      float* __restrict__ Wf, float* __restrict__ c1,
                                                      float* __restrict__ c2, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 4 * lane; k < K; k += 256) {
        const f32x4 w = ld4(W + (int64_t)n * K + k), g = ld4(gamma + k), b = ld4(beta + k);
        const f32x4 wf = w * g;
        st4(Wf + (int64_t)n * K + k, wf);
        s1 += ((double)wf.x + (double)wf.y) + ((double)wf.z + (double)wf.w);
        s2 += ((double)w.x * b.x + (double)w.y * b.y) + ((double)w.z * b.z + (double)w.w * b.w);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) {
        c1[n] = (float)s1;
        c2[n] = (float)(s2 + (bias ? (double)bias[n] : 0.0));
    }
}

extern "C" int vh_ln_fold(const float* W, const float* gamma, const float* beta, const float* bias,
                          float* Wf, float* c1, float* c2, int N, int K, void* stream) {
    VH_REQUIRE(W && gamma && beta && Wf && c1 && c2, VH_EINVAL, "vh_ln_fold: null pointer");
    VH_REQUIRE(N > 0 && K > 0 && K % 4 == 0, VH_EINVAL, "vh_ln_fold: N=%d K=%d (K must be a multiple of 4)", N, K);
    VH_REQUIRE(vh_aligned16(W) && vh_aligned16(gamma) && vh_aligned16(beta) && vh_aligned16(Wf), VH_EALIGN,
               "vh_ln_fold: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(ln_fold_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, gamma, beta,
                       bias, Wf, c1, c2, N, K);
    VH_CHECK_LAUNCH("vh_ln_fold");
    return VH_OK;
}

static int check_folded(const char* name, const GemmArgs& a, const LnFuse& ln) {
    VH_REQUIRE(ln.c1 && ln.c2, VH_EINVAL, "%s: null c1/c2", name);
    VH_REQUIRE(a.M <= 64 && a.N % 16 == 0 && (a.K == 128 || a.K == 256 || a.K == 512 || a.K == 1024),
               VH_EUNSUPPORTED,
               "%s: folded LayerNorm is the decode path: M <= 64, N %% 16 == 0, K in {128,256,512,1024} "
               "(M=%d N=%d K=%d)", name, a.M, a.N, a.K);
    VH_REQUIRE(vh_aligned16(ln.c1) && vh_aligned16(ln.c2), VH_EALIGN, "%s: c1/c2 must be 16-byte aligned", name);
    return VH_OK;
}

// ---- the AR head with the greedy step in the same launch (EPI_HEAD) ----
#define HEAD_WS_COUNTERS 64               // ints in front of the candidates (one per row group)
extern "C" size_t vh_head_greedy_ws_bytes(int B, int V) {
    if (B <= 0 || V <= 0) return 0;
    return HEAD_WS_COUNTERS * sizeof(int) + (size_t)B * ((V + 15) / 16) * sizeof(uint64_t);
}

extern "C" int vh_head_greedy(const float* x, int ldx, const float* proj_w, float* logits, int ldl, int V, int eos,
                              int64_t* codes, int64_t codes_stride, int32_t* eos_count, const int32_t* pos_base,
                              const float* audio_emb, const float* pe, int32_t* audio_pos, int32_t* cache_len, float* x_next,
                              int B, int d, void* workspace, size_t workspace_bytes, void* stream) {
    VH_REQUIRE(x && proj_w && logits && codes && eos_count && audio_emb && pe && audio_pos && cache_len && x_next && workspace,
               VH_EINVAL, "vh_head_greedy: null pointer");
    VH_REQUIRE(B > 0 && V > 0 && ldl >= V && ldx >= d && ldx % 4 == 0, VH_EINVAL,
               "vh_head_greedy: bad dims B=%d V=%d ldl=%d d=%d ldx=%d", B, V, ldl, d, ldx);
    VH_REQUIRE(B <= 64 && (d == 128 || d == 256 || d == 512 || d == 1024), VH_EUNSUPPORTED,
               "vh_head_greedy: B <= 64 rows and d_model in {128, 256, 512, 1024} (B=%d d=%d); use vh_linear + vh_greedy_step",
               B, d);
    VH_REQUIRE(vh_aligned16(x) && vh_aligned16(proj_w) && vh_aligned16(audio_emb) && vh_aligned16(pe) && vh_aligned16(x_next) &&
                   vh_aligned16(workspace),
               VH_EALIGN, "vh_head_greedy: x/proj_w/audio_emb/pe/x_next and the workspace must be 16-byte aligned");
    VH_REQUIRE(workspace_bytes >= vh_head_greedy_ws_bytes(B, V), VH_EINVAL, "vh_head_greedy: workspace of %zu bytes, %zu needed",
               workspace_bytes, vh_head_greedy_ws_bytes(B, V));
    GemmArgs a{};
    a.A = x; a.lda = ldx; a.W = proj_w; a.out = logits; a.ldo = ldl; a.M = B; a.N = V; a.K = d; a.act = VH_ACT_NONE; a.k_len = d;
    dim3 grid((V + 15) / 16);
    if (B > 16) {                          // row groups as launch_gemm chooses them: 8 rows while the grid stays within the CUs
        a.rg_rows = ((int)grid.x * ((B + 7) / 8) <= 256) ? 8 : 16;
        grid.z = (B + a.rg_rows - 1) / a.rg_rows;
    }
    HeadStep hs{};
    hs.counters = (int*)workspace;
    hs.cand = (uint64_t*)((char*)workspace + HEAD_WS_COUNTERS * sizeof(int));
    hs.V = V; hs.eos = eos; hs.codes = codes; hs.codes_stride = codes_stride; hs.eos_count = eos_count; hs.pos_base = pos_base;
    hs.audio_emb = audio_emb; hs.pe = pe; hs.audio_pos = audio_pos; hs.cache_len = cache_len; hs.x_next = x_next; hs.d = d;
    hipStream_t s = (hipStream_t)stream;
    if (d % 512 == 0) hipLaunchKernelGGL(gemm_head_greedy_kernel<4>, grid, dim3(512), 0, s, SKINNY_ARGS(a), a, hs);
    else if (d == 256) hipLaunchKernelGGL(gemm_head_greedy_kernel<2>, grid, dim3(512), 0, s, SKINNY_ARGS(a), a, hs);
    else hipLaunchKernelGGL(gemm_head_greedy_kernel<1>, grid, dim3(512), 0, s, SKINNY_ARGS(a), a, hs);
    VH_CHECK_LAUNCH("vh_head_greedy");
    return VH_OK;
}

extern "C" int vh_linear_folded(const float* A, int lda, const float* Wf, const float* c1, const float* c2,
                                const float* residual, int ldr, float* out, int ldo, int M, int N, int K,
                                int act, float ln_eps, void* stream) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = Wf; a.res = residual; a.ldr = ldr; a.out = out;
    a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.act = act; a.k_len = K;
    LnFuse ln{nullptr, nullptr, nullptr, nullptr, ln_eps, c1, c2};
    VH_REQUIRE(act == VH_ACT_NONE || act == VH_ACT_GELU_ERF, VH_EINVAL, "vh_linear_folded: act=%d", act);
    VH_REQUIRE(ldo >= N && (!residual || ldr >= N), VH_EINVAL, "vh_linear_folded: ldo/ldr < N");
    if (int rc = check_gemm("vh_linear_folded", a, ln)) return rc;
    if (int rc = check_folded("vh_linear_folded", a, ln)) return rc;
    return launch_gemm<EPI_PLAIN>("vh_linear_folded", a, ln, (hipStream_t)stream);
}

extern "C" int vh_linear_qkv_folded(const float* A, int lda, const float* Wf, const float* c1, const float* c2,
                                    float* q_out, int ldq, float* kcache, float* vcache,
                                    const int32_t* cache_len, int B, int T, int d_model, int n_heads,
                                    int S_max, float ln_eps, void* stream) {
    VH_REQUIRE(kcache && vcache, VH_EINVAL, "vh_linear_qkv_folded: null cache");
    VH_REQUIRE(B >= 0 && T >= 0 && n_heads > 0 && d_model == n_heads * VH_HEAD_DIM, VH_EUNSUPPORTED,
               "vh_linear_qkv_folded: d_model=%d must equal n_heads=%d x %d", d_model, n_heads, VH_HEAD_DIM);
    VH_REQUIRE(S_max >= T && ldq >= d_model, VH_EINVAL, "vh_linear_qkv_folded: S_max=%d < T=%d or ldq", S_max, T);
    VH_REQUIRE(vh_aligned16(kcache) && vh_aligned16(vcache), VH_EALIGN, "vh_linear_qkv_folded: cache alignment");
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = Wf; a.out = q_out; a.ldo = ldq;
    a.M = B * T; a.N = 3 * d_model;
    a.K = d_model; a.k_len = d_model; a.act = VH_ACT_NONE; a.kc = kcache; a.vc = vcache; a.cache_len = cache_len;
    a.T = T > 0 ? T : 1; a.S_max = S_max; a.d_model = d_model; a.n_heads = n_heads;
    LnFuse ln{nullptr, nullptr, nullptr, nullptr, ln_eps, c1, c2};
    if (int rc = check_gemm("vh_linear_qkv_folded", a, ln)) return rc;
    if (int rc = check_folded("vh_linear_qkv_folded", a, ln)) return rc;
    return launch_gemm<EPI_QKV>("vh_linear_qkv_folded", a, ln, (hipStream_t)stream);
}

// perf mode of the decode step: the same launch with the K / V rows appended to a bf16 cache (B,h,S_max,64)
extern "C" int vh_linear_qkv_folded_kv16(const float* A, int lda, const float* Wf, const float* c1, const float* c2,
                                         float* q_out, int ldq, uint16_t* kcache16, uint16_t* vcache16,
                                         const int32_t* cache_len, int B, int d_model, int n_heads, int S_max,
                                         float ln_eps, void* stream) {
    VH_REQUIRE(kcache16 && vcache16, VH_EINVAL, "vh_linear_qkv_folded_kv16: null cache");
    VH_REQUIRE(B >= 0 && B <= 64 && n_heads > 0 && d_model == n_heads * VH_HEAD_DIM, VH_EUNSUPPORTED,
               "vh_linear_qkv_folded_kv16: B=%d d_model=%d n_heads=%d (decode rows, d_model = 64 n_heads)", B, d_model, n_heads);
    VH_REQUIRE(S_max >= 1 && ldq >= d_model, VH_EINVAL, "vh_linear_qkv_folded_kv16: S_max / ldq");
    VH_REQUIRE(vh_aligned16(kcache16) && vh_aligned16(vcache16), VH_EALIGN, "vh_linear_qkv_folded_kv16: cache alignment");
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = Wf; a.out = q_out; a.ldo = ldq;
    a.M = B; a.N = 3 * d_model;
    a.K = d_model; a.k_len = d_model; a.act = VH_ACT_NONE;
    a.kc = reinterpret_cast<float*>(kcache16); a.vc = reinterpret_cast<float*>(vcache16); a.cache_len = cache_len;
    a.T = 1; a.S_max = S_max; a.d_model = d_model; a.n_heads = n_heads;
    LnFuse ln{nullptr, nullptr, nullptr, nullptr, ln_eps, c1, c2};
    if (int rc = check_gemm("vh_linear_qkv_folded_kv16", a, ln)) return rc;
    if (int rc = check_folded("vh_linear_qkv_folded_kv16", a, ln)) return rc;
    return launch_gemm<EPI_QKV16>("vh_linear_qkv_folded_kv16", a, ln, (hipStream_t)stream);
}

// ---- the decode step of perf mode with h16 weights (plan.hip): vh_linear (no LayerNorm, M <= 64) and vh_linear_qkv_folded_kv16
int vh_internal_linear_w16(const float* A, int lda, const uint16_t* W16, const float* bias, const float* residual, int ldr,
                           float* out, int ldo, int M, int N, int K, void* stream) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = reinterpret_cast<const float*>(W16); a.bias = bias; a.res = residual; a.ldr = ldr; a.out = out;
    a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.act = VH_ACT_NONE; a.k_len = K; a.w16 = 1;
    LnFuse ln{};
    VH_REQUIRE(M <= 64 && K % 128 == 0 && K <= 1024 && ldo >= N && (!residual || ldr >= N), VH_EUNSUPPORTED,
               "vh_linear (16-bit weights): M=%d K=%d (decode rows, K a multiple of 128 up to 1024)", M, K);
    VH_REQUIRE(((uintptr_t)W16 & 7u) == 0, VH_EALIGN, "vh_linear (16-bit weights): W must be 8-byte aligned");
    if (int rc = check_gemm("vh_linear_w16", a, ln)) return rc;
    return launch_gemm<EPI_PLAIN>("vh_linear_w16", a, ln, (hipStream_t)stream);
}

int vh_internal_linear_qkv_folded_kv16_w16(const float* A, int lda, const uint16_t* Wf16, const float* c1, const float* c2,
                                           float* q_out, int ldq, uint16_t* kcache16, uint16_t* vcache16, const int32_t* cache_len,
                                           int B, int d_model, int n_heads, int S_max, float ln_eps, void* stream) {
    VH_REQUIRE(kcache16 && vcache16 && Wf16, VH_EINVAL, "vh_linear_qkv_folded_kv16 (16-bit weights): null pointer");
    VH_REQUIRE(B >= 0 && B <= 64 && n_heads > 0 && d_model == n_heads * VH_HEAD_DIM, VH_EUNSUPPORTED,
               "vh_linear_qkv_folded_kv16 (16-bit weights): B=%d d_model=%d n_heads=%d", B, d_model, n_heads);
    VH_REQUIRE(S_max >= 1 && ldq >= d_model, VH_EINVAL, "vh_linear_qkv_folded_kv16 (16-bit weights): S_max / ldq");
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = reinterpret_cast<const float*>(Wf16); a.out = q_out; a.ldo = ldq;
    a.M = B; a.N = 3 * d_model;
    a.K = d_model; a.k_len = d_model; a.act = VH_ACT_NONE; a.w16 = 1;
    a.kc = reinterpret_cast<float*>(kcache16); a.vc = reinterpret_cast<float*>(vcache16); a.cache_len = cache_len;
    a.T = 1; a.S_max = S_max; a.d_model = d_model; a.n_heads = n_heads;
    LnFuse ln{nullptr, nullptr, nullptr, nullptr, ln_eps, c1, c2};
    if (int rc = check_gemm("vh_linear_qkv_folded_kv16_w16", a, ln)) return rc;
    if (int rc = check_folded("vh_linear_qkv_folded_kv16_w16", a, ln)) return rc;
    return launch_gemm<EPI_QKV16>("vh_linear_qkv_folded_kv16_w16", a, ln, (hipStream_t)stream);
}

extern "C" int vh_linear_ws(const float* A, int lda, const float* W, const float* bias,
                            const float* residual, int ldr, float* out, int ldo, int M, int N, int K,
                            int act, void* workspace, size_t workspace_bytes, void* stream) {
    const int splits = splitk_plan(M, N, K);
    if (!splits && workspace && M > 64 && vh_linear_ex_ws_bytes(M, N, K))   // many tiles with a short tail: split the tail
        return vh_linear_ex(A, lda, W, bias, residual, ldr, out, ldo, nullptr, 0, nullptr, M, N, K, act, nullptr, workspace,
                            workspace_bytes, stream);
    if (!splits || !workspace)
        return vh_linear(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, act, nullptr, nullptr,
                         nullptr, nullptr, 0.f, stream);
    VH_REQUIRE(workspace_bytes >= vh_linear_ws_bytes(M, N, K) && vh_aligned16(workspace), VH_EINVAL,
               "vh_linear_ws: workspace too small or unaligned (%zu < %zu)", workspace_bytes,
               vh_linear_ws_bytes(M, N, K));
    VH_REQUIRE(act == VH_ACT_NONE || act == VH_ACT_GELU_ERF, VH_EINVAL, "vh_linear_ws: act=%d", act);
    VH_REQUIRE(ldo >= N && (!residual || ldr >= N), VH_EINVAL, "vh_linear_ws: ldo/ldr < N");
    if (M == 0) return VH_OK;
    const int lds_ = (N + 3) / 4 * 4;
    GemmArgs part{};
    float* slabs = (float*)workspace;
    part.A = A; part.lda = lda; part.W = W; part.out = slabs; part.ldo = lds_; part.M = M;
    part.N = N; part.K = K; part.act = VH_ACT_NONE; part.k_len = K / splits;
    LnFuse none{nullptr, nullptr, nullptr, nullptr, 0.f};
    if (int rc = check_gemm("vh_linear_ws", part, none)) return rc;
    GemmArgs fin{};
    fin.bias = bias; fin.res = residual; fin.ldr = ldr; fin.out = out; fin.ldo = ldo; fin.M = M; fin.N = N;
    fin.K = K; fin.act = act;
    VH_REQUIRE(vh_aligned16(out) && vh_aligned16(bias) && vh_aligned16(residual) && ldo % 4 == 0 &&
                   (!residual || ldr % 4 == 0),
               VH_EALIGN, "vh_linear_ws: out/bias/residual alignment");
    hipStream_t s = (hipStream_t)stream;
    if (M > 64) {                                 // tile kernel, K slices in gridDim.y
        const int tm = (M + TM - 1) / TM, tn = (N + TN - 1) / TN;
        if (part.k_len % TK == 0 && vh_tuning(VH_TUNE_TILE_DMA) != 1)
            hipLaunchKernelGGL((gemm_tile_dma_kernel<EPI_PARTIAL>), dim3(tm * tn, splits), dim3(256), 0, s, part, tm, tn,
                               TailSplit{tm * tn, 1, part.k_len, nullptr});
        else
            hipLaunchKernelGGL((gemm_tile_kernel<EPI_PARTIAL>), dim3(tm * tn, splits), dim3(256), 0, s, part, tm, tn);
        const int items_t = M * (lds_ / 4);
        const int rbt = 256;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((items_t + rbt - 1) / rbt), dim3(rbt), 0, s,
                           (const float*)slabs, splits, fin, lds_);
        VH_CHECK_LAUNCH("vh_linear_ws");
        return VH_OK;
    }
    dim3 grid((N + 15) / 16, splits);
    const int mt = (M + 15) / 16;
    // each slice: 4 waves x 4 k-steps of 16 = 256 k per pass
    if (mt == 1) hipLaunchKernelGGL((gemm_skinny_fast<1, 4, EPI_PARTIAL, 4, 0, 1>), grid, dim3(256), 0, s, SKINNY_ARGS(part), part, none);
    else if (mt == 2) hipLaunchKernelGGL((gemm_skinny_fast<2, 4, EPI_PARTIAL, 4, 0, 1>), grid, dim3(256), 0, s, SKINNY_ARGS(part), part, none);
    else hipLaunchKernelGGL((gemm_skinny_fast<4, 4, EPI_PARTIAL, 4, 0, 1>), grid, dim3(256), 0, s, SKINNY_ARGS(part), part, none);
    const int items = M * (lds_ / 4);
    // small workgroups: the slabs (splits x 64 KB) are pulled through as many CUs as possible
    const int rb = vh_tuning(VH_TUNE_REDUCE_BLOCK) > 0 ? vh_tuning(VH_TUNE_REDUCE_BLOCK) : 128;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((items + rb - 1) / rb), dim3(rb), 0, s,
                       (const float*)slabs, splits, fin, lds_);
    VH_CHECK_LAUNCH("vh_linear_ws");
    return VH_OK;
}

// =============================================================================================
// Weight-gradient GEMM of the training backward:  C (NI, NJ) = Aᵀ · B,  A (M, NI), B (M, NJ), both stored
// with the CONTRACTION index m as the row (dW = dYᵀ · X: A = dY, B = X; valle_ar.py:86 loss.backward()).
//
// Same machine as gemm_tile_dma_kernel (128x128 output tile, 4 waves x 2x2 v_mfma_f32_32x32x2_f32, LDS-DMA
// double buffer, two workgroups per CU, pinned schedule, no vector ALU in the K loop), with the operand
// image in LDS being what lies in memory: a K slab is 32 contraction rows x 128 tile columns = 32 rows of
// 512 B per operand, DMA'd as they are (one wave-instruction = 2 whole rows, perfectly coalesced, LDS
// written lane-linearly: [k][128], no padding, no swizzle).  The MFMA wants A[i = lane&31][k = lane>>5]:
// lane (r, h) reads LDS[k = 2j + h][tile column r] — the 32 lanes of a half read 32 consecutive dwords
// (ds_read_b32 banks are (a/4) % 32 per half: conflict-free), one ds_read_b32 per operand register, all
// addressed as one per-lane base + 16-bit immediate offsets.  64 ds_read_b32 + 64 MFMAs per wave and slab.
//
// Few output tiles (NI x NJ is a weight: 16..64 tiles) against a long contraction (every token of the
// batch): the contraction is cut into gridDim.y slices, each slice leaves its partial tile in a slab of
// the workspace (tile_epilogue<EPI_PARTIAL>) and tn_reduce_kernel adds the slabs in slice order — bitwise
// reproducible, no atomics.  One slice writes straight into C.
// =============================================================================================
#define TN_OPSZ 4224            // floats per operand buffer: a 32 x 128 slab (4096) + the epilogue's 128 x 132 needs 4 x 4224

struct TnArgs {
    const float* A; int lda;
    const float* B; int ldb;
    int M, NI, NJ;
    int m_chunk;                // contraction rows per slice (multiple of 32)
};

__global__ __launch_bounds__(256, 2) void gemm_tile_tn_kernel(TnArgs g, GemmArgs a, int tiles_i, int tiles_j) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TN_OPSZ];   // [buf][A|B][k][128]
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nwg = tiles_i * tiles_j;
    const int bid = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int i0 = (tile / tiles_j) * TM, j0 = (tile % tiles_j) * TN;
    const int mbeg = blockIdx.y * g.m_chunk;
    const int mend = min(g.M, mbeg + g.m_chunk);
    const int rows_total = mend - mbeg;                   // > 0 by construction of the grid
    const int nk = (rows_total + TK - 1) / TK;

    // DMA: wave w issues pieces q = 8w..8w+7 of a slab (16 of A, then 16 of B); piece q covers contraction
    // rows 2(q&15), 2(q&15)+1; lane L fetches the 16-B chunk L&31 of row (L>>5).  Columns beyond the operand's
    // padded width are clamped (their products land in output rows/columns that are never stored).
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const char* baseA = (const char*)(g.A + (int64_t)mbeg * g.lda + i0);
    const char* baseB = (const char*)(g.B + (int64_t)mbeg * g.ldb + j0);
    const int wa = ((g.NI + 3) & ~3) - i0 - 4, wb = ((g.NJ + 3) & ~3) - j0 - 4;   // last whole chunk of the tile's row
    uint32_t voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = ws * 8 + i, row = (q & 15) * 2 + (lane >> 5), c4 = 4 * (lane & 31);
        voff[i] = q < 16 ? (uint32_t)(row * g.lda + min(c4, wa)) * 4u : (uint32_t)(row * g.ldb + min(c4, wb)) * 4u;
    }
    const int64_t stepA = (int64_t)TK * g.lda * 4, stepB = (int64_t)TK * g.ldb * 4;   // bytes per K slab
    auto dma1 = [&](int i, int buf, int kt, auto tail_c) {
        const int q = ws * 8 + i;
        const char* base = q < 16 ? baseA + kt * stepA : baseB + kt * stepB;
        const uint32_t dst = (uint32_t)(uintptr_t)&lds[buf][q >> 4][(q & 15) * 256];
        uint32_t vo = voff[i];
        if constexpr (decltype(tail_c)::value) {          // last slab of the slice: rows beyond the end re-read the last row
            const int row = (q & 15) * 2 + (lane >> 5), last = rows_total - 1 - kt * TK;
            if (row > last) vo -= (uint32_t)((row - last) * (q < 16 ? g.lda : g.ldb)) * 4u;
        }
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst), "v"(vo), "s"(base) : "memory");
    };
    const bool ragged = (rows_total % TK) != 0;           // wave-uniform
    // rows of the A image beyond the slice's end must multiply as zeros (B's re-read rows are finite)
    auto zero_tail = [&](int buf, int kt) {
        const int valid = rows_total - kt * TK;           // < 32
        for (int idx = tid; idx < (TK - valid) * 32; idx += 256)
            st4(&lds[buf][0][(valid + idx / 32) * 128 + 4 * (idx % 32)], f32x4{0.f, 0.f, 0.f, 0.f});
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (ragged && nk == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma1(i, 0, 0, std::true_type{});
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma1(i, 0, 0, std::false_type{});
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (ragged && nk == 1) { zero_tail(0, 0); __syncthreads(); }

    // fragment addresses: per-lane base (k row h of the slab, tile column) + immediates
    const float* pA = &lds[0][0][h * 128 + wm * 64 + r];
    const float* pB = &lds[0][1][h * 128 + wn * 64 + r];
    float fa[2][2][4], fb[2][2][4];                       // [set][tile][k pair of the group]
    auto fload = [&](int set, int buf, int t) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int o = buf * 2 * TN_OPSZ + (4 * t + jj) * 256;
            fa[set][0][jj] = pA[o];
            fa[set][1][jj] = pA[o + 32];
            fb[set][0][jj] = pB[o];
            fb[set][1][jj] = pB[o + 32];
        }
    };
    auto mfma4 = [&](int set, int jj) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][0][jj], fb[set][0][jj], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][0][jj], fb[set][1][jj], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][1][jj], fb[set][0][jj], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][1][jj], fb[set][1][jj], acc[1][1], 0, 0, 0);
    };
    __builtin_amdgcn_s_setprio(0);
    fload(0, 0, 0);
    // PF: 0 = nothing to prefetch (last slab), 1 = prefetch a whole slab (the steady state: no branch, no vector
    // ALU besides the MFMAs), 2 = prefetch the slice's last slab, which may be partial
    auto kstep = [&](int kt, auto cur_c, auto pf) {
        constexpr int PF = decltype(pf)::value;
        constexpr int cur = decltype(cur_c)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            mfma4(t & 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF == 1) {
                if (t < 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dma1(4 * t + i, cur ^ 1, kt + 1, std::false_type{});
                }
            }
            if constexpr (PF == 2) {
                if (t < 2) {
                    if (ragged) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) dma1(4 * t + i, cur ^ 1, kt + 1, std::true_type{});
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) dma1(4 * t + i, cur ^ 1, kt + 1, std::false_type{});
                    }
                }
            }
            if (t < 3) {
                fload((t + 1) & 1, cur, t + 1);
            } else if constexpr (PF != 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
                if constexpr (PF == 2) {
                    if (ragged) {                         // wave-uniform, once per workgroup
                        zero_tail(cur ^ 1, kt + 1);
                        __syncthreads();
                    }
                }
                fload(0, cur ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma4(t & 1, 1);
            mfma4(t & 1, 2);
            mfma4(t & 1, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {                        // prefetches slabs <= nk - 2: always whole
        kstep(kt, B0{}, P1{});
        kstep(kt + 1, B1{}, P1{});
    }
    const int rem = nk - kt;                              // 1, 2 or 3 slabs left, the current one in buffer 0
    if (rem == 3) {
        kstep(kt, B0{}, P1{});
        kstep(kt + 1, B1{}, P2{});
        kstep(kt + 2, B0{}, P0{});
    } else if (rem == 2) {
        kstep(kt, B0{}, P2{});
        kstep(kt + 1, B1{}, P0{});
    } else {
        kstep(kt, B0{}, P0{});
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);
    TileEpi epi;
    epi.bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    tile_epilogue<EPI_PARTIAL>(a, acc, &lds[0][0][0], i0, j0, tid, epi);   // slab blockIdx.y of a.out (or C itself)
}

// C[i][j] = sum over slices of slab[s][i][j], in slice order (fixed → reproducible)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slabs, int n_split, int rows, int lds_,
                                                        int ncols, float* __restrict__ C, int ldc) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int groups = lds_ / 4;
    const int i = idx / groups, j = (idx - i * groups) * 4;
    if (i >= rows || j >= ncols) return;
    const float* p = slabs + (int64_t)i * lds_ + j;
    const int64_t stride = (int64_t)rows * lds_;
    f32x4 acc = ld4(p);
    for (int s0 = 1; s0 < n_split; s0 += 8) {
        f32x4 part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            part[u] = s0 + u < n_split ? ld4(p + (s0 + u) * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += part[u];
    }
    if (j + 3 < ncols) {
        st4(C + (int64_t)i * ldc + j, acc);
    } else {
        for (int u = 0; u < 4 && j + u < ncols; ++u) C[(int64_t)i * ldc + j + u] = acc[u];
    }
}

static int tn_plan(int M, int NI, int NJ) {
    const int tiles = ((NI + TM - 1) / TM) * ((NJ + TN - 1) / TN);
    const int slabs = (M + TK - 1) / TK;
    // one workgroup per CU: a lone workgroup already runs its MFMA stream at 94 % of what two resident ones reach
    // together, and half the slices are half the slab traffic (tools/ab_tn_split.py: 512 -> 256 workgroups is equal on
    // the 64-tile weights and 5-6 % faster on the 16- and 48-tile ones at 16 k contraction rows)
    const int wgs = vh_tuning(VH_TUNE_TN_WGS) > 0 ? vh_tuning(VH_TUNE_TN_WGS) : 256;
    int splits = wgs / tiles;
    if (splits > slabs / 8) splits = slabs / 8;            // at least 8 slabs (256 rows) per slice
    if (splits > 64) splits = 64;
    return splits < 1 ? 1 : splits;
}

extern "C" size_t vh_gemm_tn_ws_bytes(int M, int NI, int NJ) {
    const int splits = tn_plan(M, NI, NJ);
    return splits > 1 ? (size_t)splits * NI * ((NJ + 3) / 4 * 4) * sizeof(float) : 0;
}

extern "C" int vh_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int NI, int NJ,
                          void* workspace, size_t workspace_bytes, void* stream) {
    VH_REQUIRE(A && B && C, VH_EINVAL, "vh_gemm_tn: null pointer");
    VH_REQUIRE(M > 0 && NI > 0 && NJ > 0, VH_EINVAL, "vh_gemm_tn: bad dims M=%d NI=%d NJ=%d", M, NI, NJ);
    VH_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && lda >= ((NI + 3) & ~3) && ldb >= ((NJ + 3) & ~3) &&
                   ldc >= NJ,
               VH_EALIGN, "vh_gemm_tn: leading dimensions must be multiples of 4 and cover the padded row "
               "(lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    VH_REQUIRE(vh_aligned16(A) && vh_aligned16(B) && vh_aligned16(C) && vh_aligned16(workspace), VH_EALIGN,
               "vh_gemm_tn: pointers must be 16-byte aligned");
    const int splits = tn_plan(M, NI, NJ);
    const int lds_ = (NJ + 3) / 4 * 4;
    VH_REQUIRE(splits == 1 || (workspace && workspace_bytes >= vh_gemm_tn_ws_bytes(M, NI, NJ)), VH_EINVAL,
               "vh_gemm_tn: workspace of %zu B needed (vh_gemm_tn_ws_bytes)", vh_gemm_tn_ws_bytes(M, NI, NJ));
    const int slabs = (M + TK - 1) / TK;
    const int chunk = (slabs + splits - 1) / splits * TK;
    const int ny = (M + chunk - 1) / chunk;               // slices that really hold rows
    TnArgs g{A, lda, B, ldb, M, NI, NJ, chunk};
    GemmArgs a{};
    a.M = NI; a.N = NJ; a.K = M; a.k_len = chunk;
    a.out = ny > 1 ? (float*)workspace : C;
    a.ldo = ny > 1 ? lds_ : ldc;
    const int ti = (NI + TM - 1) / TM, tj = (NJ + TN - 1) / TN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tile_tn_kernel, dim3(ti * tj, ny), dim3(256), 0, s, g, a, ti, tj);
    if (ny > 1) {
        const int n = NI * (lds_ / 4);
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)workspace, ny, NI,
                           lds_, NJ, C, ldc);
    }
    VH_CHECK_LAUNCH("vh_gemm_tn");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// out (cols, rows) = in (rows, cols)ᵀ through 32x33 LDS tiles — the weights of the backward's dX = dY · W
// products are handed to the NT tile kernel as Wᵀ (a few MB per layer and step).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int ldi, int rows, int cols,
                                                        float* __restrict__ out, int ldo) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 + ty + 8 * i, cc = c0 + tx;
        t[ty + 8 * i][tx] = (rr < rows && cc < cols) ? in[(int64_t)rr * ldi + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ty + 8 * i, rr = r0 + tx;       // out row = in column; zero fill up to ldo (k padding)
        if (cc < cols && rr < ldo) out[(int64_t)cc * ldo + rr] = t[tx][ty + 8 * i];
    }
}

extern "C" int vh_transpose(const float* in, int ldi, int rows, int cols, float* out, int ldo, void* stream) {
    VH_REQUIRE(in && out && rows > 0 && cols > 0 && ldi >= cols && ldo >= rows, VH_EINVAL,
               "vh_transpose: bad args rows=%d cols=%d ldi=%d ldo=%d", rows, cols, ldi, ldo);
    // the tile grid covers ldo input rows, so out rows are zero-filled to the full ldo (a padded contraction width)
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (ldo + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                       in, ldi, rows, cols, out, ldo);
    VH_CHECK_LAUNCH("vh_transpose");
    return VH_OK;
}

// Many transposes in one launch: workgroup = one 32x32 tile; the item is found by a scan of the (<= a few hundred)
// descriptors' tile offsets by the first wave.
__global__ __launch_bounds__(256) void transpose_many_kernel(const vh_transpose_item* __restrict__ items, int n) {
    __shared__ float t[32][33];
    __shared__ int s_item;
    const int tile = blockIdx.x;
    if (threadIdx.x < 64) {
        int found = 0;
        for (int i = threadIdx.x; i < n; i += 64)
            if (items[i].tile0 <= tile) found = max(found, i);       // tile0 ascending: the last item starting at or before
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) found = max(found, __shfl_xor(found, o));
        if (threadIdx.x == 0) s_item = found;
    }
    __syncthreads();
    const vh_transpose_item it = items[s_item];
    const int local = tile - it.tile0, tiles_x = (it.cols + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = (local / tiles_x) * 32, c0 = (local % tiles_x) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 + ty + 8 * i, cc = c0 + tx;
        t[ty + 8 * i][tx] = (rr < it.rows && cc < it.cols) ? it.in[(int64_t)rr * it.ldi + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ty + 8 * i, rr = r0 + tx;
        if (cc < it.cols && rr < it.ldo) it.out[(int64_t)cc * it.ldo + rr] = t[tx][ty + 8 * i];
    }
}

extern "C" int vh_transpose_many(const vh_transpose_item* items, int n, int total_tiles, void* stream) {
    VH_REQUIRE(items && n > 0 && n <= 4096 && total_tiles > 0, VH_EINVAL, "vh_transpose_many: n=%d total_tiles=%d", n,
               total_tiles);
    hipLaunchKernelGGL(transpose_many_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, items, n);
    VH_CHECK_LAUNCH("vh_transpose_many");
    return VH_OK;
}
