// General batched fp32 GEMM for the backward pass:  C[b] = op(A[b]) · op(B[b])
//
// Same machine as gemm_tile_kernel (128x128x32 LDS tiles, 4 waves x 2x2 v_mfma_f32_32x32x2_f32,
// register-staged double buffer) with the operand storage made a template parameter:
//   A_KMAJOR = false : A is stored (M, K), k contiguous        (activations / dY as the left operand)
//   A_KMAJOR = true  : A is stored (K, M), m contiguous        (a transposed left operand: dYᵀ, Pᵀ, dSᵀ)
//   B_KMAJOR = false : B is stored (N, K), k contiguous        (nn.Linear weight as stored, Kᵀ, Vᵀ)
//   B_KMAJOR = true  : B is stored (K, N), n contiguous        (W for dX = dY·W, X for dW, K/Q/dO rows)
// A k-major tile is staged as it lies in memory ([k][128+4] in LDS, coalesced 16-B global loads
// along m) and its MFMA fragments are read with four ds_read_b32 (consecutive lanes → consecutive
// m: conflict-free) instead of being transposed on the way in.
// Two-level batch (b, h) with element strides so attention operands can be read in place from the
// (B,T,h,64) / (B,h,T,64) / (B,h,T,T) tensors and gradients written straight into (B*T, 3d).
// Leading dimensions must be multiples of 4 (16-B loads); M, N, K are free (guards + zero fill).
#include "vh_common.h"

#define GT 128
#define GK 32
#define LD_KC 36     // k-contiguous operand: [128 rows][36]
#define LD_KM 132    // k-major operand:      [32 k][132]
#define OPSZ (GT * LD_KC)  // 4608 floats >= 32*132 = 4224

struct GenArgs {
    const float* A; int lda; int64_t sAb, sAh;
    const float* B; int ldb; int64_t sBb, sBh;
    float* C; int ldc; int64_t sCb, sCh;
    int M, N, K, H;
    int k_chunk;   // K range of one gridDim.z slice (multiple of 32); == K when not split
};

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_general_kernel(GenArgs g, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][OPSZ];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int m0 = (tile / tiles_n) * GT, n0 = (tile % tiles_n) * GT;
    const int bb = blockIdx.y / g.H, hh = blockIdx.y - bb * g.H;
    const float* A = g.A + bb * g.sAb + hh * g.sAh;
    const float* B = g.B + bb * g.sBb + hh * g.sBh;
    float* C = g.C + bb * g.sCb + hh * g.sCh;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};

    f32x4 ra[4], rb[4];
    auto gload_one = [&](const float* P, int ld, int lim_rows, int base, int k0, bool kmajor, f32x4 (&rg)[4]) {
        if (!kmajor) {   // rows = tile rows (m or n), float4 along k
            const int row = tid >> 3, kq = (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = base + row + 32 * i, k = k0 + kq;
                if (rr < lim_rows && k < g.K) {
                    f32x4 v = ld4(P + (int64_t)rr * ld + k);
                    if (k + 3 >= g.K) {   // ragged K: zero the tail (buffers are padded to a multiple of 4)
                        if (k + 1 >= g.K) v.y = 0.f;
                        if (k + 2 >= g.K) v.z = 0.f;
                        v.w = 0.f;
                    }
                    rg[i] = v;
                } else {
                    rg[i] = z;
                }
            }
        } else {         // rows = k, float4 along the tile dimension
            const int kr = tid >> 5, mq = (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + kr + 8 * i, mm = base + mq;
                rg[i] = (k < g.K && mm < lim_rows) ? ld4(P + (int64_t)k * ld + mm) : z;
            }
        }
    };
    auto lstore_one = [&](float* dst, bool kmajor, const f32x4 (&rg)[4]) {
        if (!kmajor) {
            const int row = tid >> 3, kq = (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) st4(dst + (row + 32 * i) * LD_KC + kq, rg[i]);
        } else {
            const int kr = tid >> 5, mq = (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) st4(dst + (kr + 8 * i) * LD_KM + mq, rg[i]);
        }
    };
    // fragment of 4 consecutive k (8t+4h .. +3) for tile row `row`
    auto frag = [&](const float* src, bool kmajor, int row, int t) -> f32x4 {
        if (!kmajor) return ld4(src + row * LD_KC + 8 * t + 4 * h);
        const float* p = src + (8 * t + 4 * h) * LD_KM + row;
        return f32x4{p[0], p[LD_KM], p[2 * LD_KM], p[3 * LD_KM]};
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // split-K: gridDim.z slices of k_chunk; partial tiles meet in C through fp32 atomics (C zeroed by
    // the caller).  Used for the weight gradients dW = dYᵀ·X: few output tiles, K = all tokens.
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int nk = (kend - kbeg + GK - 1) / GK;
    if (nk <= 0) return;
    gload_one(A, g.lda, g.M, m0, kbeg, A_KMAJOR, ra);
    gload_one(B, g.ldb, g.N, n0, kbeg, B_KMAJOR, rb);
    lstore_one(&lds[0][0][0], A_KMAJOR, ra);
    lstore_one(&lds[0][1][0], B_KMAJOR, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            gload_one(A, g.lda, g.M, m0, kbeg + (kt + 1) * GK, A_KMAJOR, ra);
            gload_one(B, g.ldb, g.N, n0, kbeg + (kt + 1) * GK, B_KMAJOR, rb);
        }
        const float* As = &lds[cur][0][0];
        const float* Bs = &lds[cur][1][0];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a0 = frag(As, A_KMAJOR, wm * 64 + r, t), a1 = frag(As, A_KMAJOR, wm * 64 + 32 + r, t);
            const f32x4 b0 = frag(Bs, B_KMAJOR, wn * 64 + r, t), b1 = frag(Bs, B_KMAJOR, wn * 64 + 32 + r, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            lstore_one(&lds[cur ^ 1][0][0], A_KMAJOR, ra);
            lstore_one(&lds[cur ^ 1][1][0], B_KMAJOR, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int n = n0 + wn * 64 + nt * 32 + r;
                if (m < g.M && n < g.N) {
                    if (gridDim.z > 1) atomicAdd(&C[(int64_t)m * g.ldc + n], acc[mt][nt][e]);
                    else C[(int64_t)m * g.ldc + n] = acc[mt][nt][e];
                }
            }
}

extern "C" int vh_gemm_batched(const float* A, int lda, int64_t sAb, int64_t sAh, int a_kmajor,
                               const float* B, int ldb, int64_t sBb, int64_t sBh, int b_kmajor, float* C,
                               int ldc, int64_t sCb, int64_t sCh, int M, int N, int K, int batch, int H,
                               int k_splits, void* stream) {
    VH_REQUIRE(A && B && C, VH_EINVAL, "vh_gemm_batched: null pointer");
    VH_REQUIRE(k_splits >= 1 && k_splits <= 64, VH_EINVAL, "vh_gemm_batched: k_splits=%d (1..64)", k_splits);
    VH_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 1 && H >= 1, VH_EINVAL,
               "vh_gemm_batched: bad dims M=%d N=%d K=%d batch=%d H=%d", M, N, K, batch, H);
    VH_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && sAb % 4 == 0 && sAh % 4 == 0 && sBb % 4 == 0 && sBh % 4 == 0,
               VH_EALIGN, "vh_gemm_batched: leading dimensions and batch strides must be multiples of 4");
    VH_REQUIRE(vh_aligned16(A) && vh_aligned16(B), VH_EALIGN, "vh_gemm_batched: A/B must be 16-byte aligned");
    VH_REQUIRE(lda >= (a_kmajor ? M : K) - 3 && ldb >= (b_kmajor ? N : K) - 3 && ldc >= N, VH_EINVAL,
               "vh_gemm_batched: leading dimension smaller than the row length");
    if (M == 0 || N == 0) return VH_OK;
    const int k_chunk = k_splits == 1 ? (K > 0 ? K : 1) : ((K + k_splits - 1) / k_splits + GK - 1) / GK * GK;
    GenArgs g{A, lda, sAb, sAh, B, ldb, sBb, sBh, C, ldc, sCb, sCh, M, N, K, H, k_chunk};
    const int tm = (M + GT - 1) / GT, tn = (N + GT - 1) / GT;
    dim3 grid(tm * tn, batch * H, k_splits);
    hipStream_t s = (hipStream_t)stream;
#define GG(AK, BK) hipLaunchKernelGGL((gemm_general_kernel<AK, BK>), grid, dim3(256), 0, s, g, tm, tn)
    if (a_kmajor) { if (b_kmajor) GG(true, true); else GG(true, false); }
    else { if (b_kmajor) GG(false, true); else GG(false, false); }
#undef GG
    VH_CHECK_LAUNCH("vh_gemm_batched");
    return VH_OK;
}
