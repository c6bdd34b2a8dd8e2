// FeedForward + residual of the decode step (M <= 64 rows) as ONE launch split over dim_feedforward:
//     out = x + b2 + GELU(LN2(x) · W1ᵀ + b1) · W2ᵀ          (valle/models/modules.py:215-221, :278-279)
//
// Why: the decode step is a chain of dependent launches, each of which costs 4.6-5 us in the replayed graph
// whatever it does; linear_1 -> linear_2 -> split-K reduce were three of the five
// GEMM-side launches of a layer.  The only dependency between linear_1 and linear_2 is per hidden column: a
// workgroup that owns SW consecutive hidden columns (a "slice") for a group of <= 16 rows can compute that
// tile of the hidden activation, keep it in LDS and multiply it by the SAME columns of W2 — a K slice of
// linear_2 — without seeing any other workgroup's data.  What crosses a kernel boundary is then only the
// split-K sum over slices, which the old form paid for anyway.  The (M, dff) hidden activation never exists
// in memory.
//
// ffn_decode_kernel<D, SW>: 8 waves, workgroup = (slice of SW hidden columns, row group of 8 or 16 rows)
//   phase 1  hid[rows][SW] = GELU(rstd · (x · W1fᵀ − mean · c1) + c2)   K = D over the 8 waves (as
//            gemm_skinny_fast with the folded LayerNorm: the products run on the raw rows while four DPP rows per
//            wave compute the row statistics), partials reduced through LDS in wave order;
//   phase 2  partial[rows][D] = hid · W2[:, slice]ᵀ   K = SW, the D output columns over the 8 waves (no
//            cross-wave reduction), stored raw into slab `slice` of the workspace.
//   Every global load of both phases (statistics rows, W1f / x fragments, W2 fragments, c1 / c2) is issued
//   before the first use: one memory round trip per workgroup.
// ffn_reduce_kernel: out = x + b2 + Σ_slices slab, slices added in a fixed order (bitwise reproducible, no
//   atomics): workgroup = 64 columns of one row, 16 column groups x 16 parts; part p adds its run of slices,
//   the 16 parts are added in part order through LDS.
//
// MFMA operand maps as in gemm.hip (16x16x4: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r: i = 4(l>>4)+r,
// j = l&15) with the WEIGHT as the A operand, so a lane's 4 results are 4 consecutive output columns of
// activation row l&15.
#include "vh_common.h"

struct FfnArgs {
    const float* c1;
    const float* c2;
    float* slabs;          // [n_slices][M][D]
    int n_slices, n_rg, rg_rows;
    float eps;
};

// W16 (perf mode of the decode step, round 6): hw1 / hw2 point at h16 matrices (vh_common.h vh_h16) of the same logical
// shape; a fragment is 8 bytes per lane, widened to fp32 in registers before its MFMAs — half the weight bytes per launch.
template <int D, int SW, bool W16 = false>
__global__ __launch_bounds__(512) void ffn_decode_kernel(const float* hx, const float* hw1, const float* hw2, int h_ldx,
                                                         int hM, int h_dff, FfnArgs a) {
    constexpr int NW = 8, PW = D / 128, CB1 = SW / 16, CB2 = D / 128, KC2 = SW / 16, HLD = SW + 4;
    static_assert(D % 128 == 0 && SW % 16 == 0 && CB1 <= 2, "shape");
    __shared__ __attribute__((aligned(16))) float red[NW][CB1][64][4];
    __shared__ __attribute__((aligned(16))) float hid[16][HLD];
    __shared__ __attribute__((aligned(16))) float pst[16][2 * NW];   // per row: (sum, sum of squares) of each wave's K range
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    // grid = (slices, row groups).  Workgroups that share a slice (its W1f rows and W2 columns) differ in blockIdx.y
    // only: their linear ids differ by multiples of n_slices, so with n_slices % 8 == 0 they run on the same XCD
    // (round-robin dispatch) and the slice's 2 x SW x D x 4 bytes of weights are fetched into ONE L2.
    const int slice = blockIdx.x, rg = blockIdx.y;
    const int row0 = rg * a.rg_rows;
    const int rows = min(hM - row0, a.rg_rows);                 // 1..16 valid rows in this group
    const float* X = hx + (int64_t)row0 * h_ldx;
    const int n1 = slice * SW;                                  // first hidden column of the slice

    // ---- every load this workgroup will ever need, in the order their data is used.  STRAIGHT-LINE code: a load or a
    // reduction under a (wave-uniform) branch lets the compiler merge the branch bodies and wait for the first loads
    // before it has issued the rest — three dependent round trips instead of one (seen in the first version's ISA).
    // phase 1: x rows i (B operand), W1f rows n1 + 16 cb + i (A operand), k = w * 16 PW + 16 c + 4 g + {0..3}
    const int koff = w * (PW * 16) + 4 * g;
    const float* xrow = X + (int64_t)min(i, rows - 1) * h_ldx;  // rows beyond the group repeat its last row
    const float shift = xrow[0];                                // statistics are taken about the row's first element
    f32x4 wf1[CB1][PW], xf[PW];
    uint2 wr1[CB1][PW];                                         // W16: raw fragments (4 h16)
#pragma unroll
    for (int c = 0; c < PW; ++c) xf[c] = ld4(xrow + koff + 16 * c);
#pragma unroll
    for (int cb = 0; cb < CB1; ++cb) {
        if constexpr (W16) {
            const uint16_t* wp = reinterpret_cast<const uint16_t*>(hw1) + (int64_t)(n1 + 16 * cb + i) * D + koff;
#pragma unroll
            for (int c = 0; c < PW; ++c) wr1[cb][c] = *reinterpret_cast<const uint2*>(wp + 16 * c);
        } else {
            const float* wp = hw1 + (int64_t)(n1 + 16 * cb + i) * D + koff;
#pragma unroll
            for (int c = 0; c < PW; ++c) wf1[cb][c] = ld4(wp + 16 * c);
        }
    }
    // phase-1 epilogue operands: wave w finalises component (w >> 1) & 3 of hidden column block w % CB1 … see below
    const int fcb = w % CB1, fcomp = (w / CB1) & 3;
    const float e_c1 = a.c1[n1 + 16 * fcb + 4 * g + fcomp];
    const float e_c2 = a.c2[n1 + 16 * fcb + 4 * g + fcomp];
    // phase 2: W2 rows (= output columns) 16 (w CB2 + cb) + i, k = n1 + 16 c + 4 g + {0..3}: one 128-B line per row at SW = 32
    f32x4 wf2[CB2][KC2];
    uint2 wr2[CB2][KC2];
#pragma unroll
    for (int cb = 0; cb < CB2; ++cb) {
        if constexpr (W16) {
            const uint16_t* wp = reinterpret_cast<const uint16_t*>(hw2) + (int64_t)(16 * (w * CB2 + cb) + i) * h_dff + n1 + 4 * g;
#pragma unroll
            for (int c = 0; c < KC2; ++c) wr2[cb][c] = *reinterpret_cast<const uint2*>(wp + 16 * c);
        } else {
            const float* wp = hw2 + (int64_t)(16 * (w * CB2 + cb) + i) * h_dff + n1 + 4 * g;
#pragma unroll
            for (int c = 0; c < KC2; ++c) wf2[cb][c] = ld4(wp + 16 * c);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    auto widen = [](uint2 r) __attribute__((always_inline)) {
        return f32x4{vh_h16_lo(r.x), vh_h16_hi(r.x), vh_h16_lo(r.y), vh_h16_hi(r.y)};
    };

    // ---- row statistics from the operand fragments (no second read of the rows): this lane holds 4 PW elements of
    // row i; sums of (x - shift) and (x - shift)^2, folded over the wave's four k groups (lanes i, i+16, i+32, i+48),
    // one (sum, sum of squares) pair per wave and row to LDS; the finalising lanes add the 8 pairs in wave order.
    // The shift (an element of the row) keeps the one-pass variance free of cancellation: |shift - mean| is a few
    // standard deviations at most, so the subtraction below loses a few bits of 24, not most of them.
    {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int c = 0; c < PW; ++c) {
            const f32x4 t = xf[c] - shift;
            sa += (t.x + t.y) + (t.z + t.w);
            sb += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        }
        sa += __shfl_xor(sa, 16, 64); sb += __shfl_xor(sb, 16, 64);
        sa += __shfl_xor(sa, 32, 64); sb += __shfl_xor(sb, 32, 64);
        if (g == 0) { pst[i][2 * w] = sa; pst[i][2 * w + 1] = sb; }
    }

    // ---- phase 1: this wave's K range of x · W1fᵀ for the SW hidden columns
    f32x4 acc1[CB1];
#pragma unroll
    for (int cb = 0; cb < CB1; ++cb) acc1[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        if constexpr (W16) {
#pragma unroll
            for (int cb = 0; cb < CB1; ++cb) wf1[cb][c] = widen(wr1[cb][c]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int cb = 0; cb < CB1; ++cb)
                acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[cb][c][jj], xf[c][jj], acc1[cb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < CB1; ++cb) st4(&red[w][cb][lane][0], acc1[cb]);
    __syncthreads();
    // Finalise, spread over the waves so that each lane evaluates ONE GELU (the code every launch has to fetch cold is
    // what these kernels wait for): wave w takes component fcomp of column block fcb — lane (i, g): activation row
    // i, hidden column n1 + 16 fcb + 4 g + fcomp — partial sums added in wave order.  (CB1 * 4 <= 8 waves take part.)
    if (w < 4 * CB1) {
        float s = red[0][fcb][lane][fcomp];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) s += red[ww][fcb][lane][fcomp];
        f32x4 p[NW / 2];
#pragma unroll
        for (int q = 0; q < NW / 2; ++q) p[q] = ld4(&pst[i][4 * q]);
        float sa = p[0].x, sb = p[0].y;
        sa += p[0].z; sb += p[0].w;
#pragma unroll
        for (int q = 1; q < NW / 2; ++q) { sa += p[q].x; sb += p[q].y; sa += p[q].z; sb += p[q].w; }
        const float dm = sa * (1.0f / D);                       // mean - shift
        const float var = fmaxf(sb * (1.0f / D) - dm * dm, 0.f);
        const float mu = shift + dm, rs = rsqrtf(var + a.eps);
        hid[i][16 * fcb + 4 * g + fcomp] = gelu_erf((s - mu * e_c1) * rs + e_c2);
    }
    __syncthreads();

    // ---- phase 2: hid (rows x SW) · W2[:, slice]ᵀ for this wave's 16 CB2 output columns
    f32x4 hf[KC2];
#pragma unroll
    for (int c = 0; c < KC2; ++c) hf[c] = ld4(&hid[i][16 * c + 4 * g]);
    f32x4 acc2[CB2];
#pragma unroll
    for (int cb = 0; cb < CB2; ++cb) acc2[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KC2; ++c) {
        if constexpr (W16) {
#pragma unroll
            for (int cb = 0; cb < CB2; ++cb) wf2[cb][c] = widen(wr2[cb][c]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int cb = 0; cb < CB2; ++cb)
                acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf2[cb][c][jj], hf[c][jj], acc2[cb], 0, 0, 0);
    }
    // non-temporal stores: the slabs are read once, by another launch, from whatever XCD — written through they do
    // not wait in this XCD's L2 for the write-back at the end of the kernel (9.35 -> 8.66 us per launch)
    if (i < rows) {
        float* dst = a.slabs + ((int64_t)slice * hM + row0 + i) * D + 16 * (w * CB2) + 4 * g;
#pragma unroll
        for (int cb = 0; cb < CB2; ++cb) __builtin_nontemporal_store(acc2[cb], reinterpret_cast<f32x4*>(dst + 16 * cb));
    }
}

// out[m][n..n+3] = x[m][n..] + b2[n..] + sum over slices of slab[s][m][n..]
__global__ __launch_bounds__(256) void ffn_reduce_kernel(const float* __restrict__ slabs, int n_slices,
                                                         const float* x, int ldx, const float* __restrict__ b2,
                                                         float* out, int ldo, int M, int D) {
    __shared__ __attribute__((aligned(16))) float part[16][16][4];
    const int tid = threadIdx.x, c = tid & 15, p = tid >> 4;
    const int chunks = D >> 6;
    const int m = blockIdx.x / chunks, n = (blockIdx.x - m * chunks) * 64 + 4 * c;
    const int per = (n_slices + 15) >> 4;                       // slices per part
    const int s0 = p * per, s1 = min(n_slices, s0 + per);
    const float* src = slabs + (int64_t)m * D + n;
    const int64_t stride = (int64_t)M * D;
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    if (p == 0) {                                               // epilogue operands requested with the slabs
        e = ld4(x + (int64_t)m * ldx + n);
        if (b2) e += ld4(b2 + n);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; s += 8) {                          // eight slices in flight, added in slice order
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld4(src + (int64_t)min(s + u, s1 - 1) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s + u < s1) acc += t[u];
    }
    st4(&part[p][c][0], acc);
    __syncthreads();
    if (p == 0) {
        f32x4 tot = ld4(&part[0][c][0]);
#pragma unroll
        for (int q = 1; q < 16; ++q) tot += ld4(&part[q][c][0]);
        st4(out + (int64_t)m * ldo + n, tot + e);
    }
}

// hidden columns per workgroup (0 = unsupported shape) and rows per workgroup
static int ffn_plan(int M, int d_model, int dff, int* rg_rows) {
    if (M < 1 || M > 64 || !(d_model == 128 || d_model == 256 || d_model == 512 || d_model == 1024) || dff % 16 != 0 ||
        dff / 16 > 1024)
        return 0;
    const int rg8 = (M + 7) / 8, rg16 = (M + 15) / 16;
    int sw = vh_tuning(VH_TUNE_FFN_SLICE);
    if (sw != 16 && sw != 32) {
        // as many workgroups as the chip has CUs, as few slabs as that allows
        sw = (dff % 32 == 0 && (dff / 32) * rg8 >= 192) ? 32 : 16;
    }
    if (dff % sw != 0 || d_model == 1024) sw = 16;     // d_model = 1024: wider slices would spill registers
    int rows = vh_tuning(VH_TUNE_FFN_ROWS);
    if (rows != 8 && rows != 16) rows = (dff / sw) * rg8 <= 256 ? 8 : 16;
    (void)rg16;
    *rg_rows = rows;
    return sw;
}

extern "C" size_t vh_ffn_decode_ws_bytes(int M, int d_model, int dff) {
    // sized for the narrowest slice, so a tuning change never outgrows a caller's workspace
    if (M < 1 || M > 64 || dff % 16 != 0) return 0;
    return (size_t)(dff / 16) * M * d_model * sizeof(float);
}

static int ffn_decode_launch(const float* x, int ldx, const float* w1f, const float* c1, const float* c2,
                             const float* w2, const float* b2, float* out, int ldo, int M, int d_model, int dff,
                             float ln_eps, void* workspace, size_t workspace_bytes, void* stream, bool w16) {
    VH_REQUIRE(x && w1f && c1 && c2 && w2 && out && workspace, VH_EINVAL, "vh_ffn_decode: null pointer");
    int rg_rows = 0;
    const int sw = ffn_plan(M, d_model, dff, &rg_rows);
    VH_REQUIRE(sw != 0, VH_EUNSUPPORTED,
               "vh_ffn_decode: M=%d d_model=%d dff=%d (1 <= M <= 64, d_model in {128,256,512,1024}, dff %% 16 == 0, "
               "dff <= 16384)", M, d_model, dff);
    VH_REQUIRE(ldx >= d_model && ldo >= d_model && ldx % 4 == 0 && ldo % 4 == 0, VH_EINVAL, "vh_ffn_decode: ldx=%d ldo=%d",
               ldx, ldo);
    VH_REQUIRE(vh_aligned16(x) && vh_aligned16(w1f) && vh_aligned16(c1) && vh_aligned16(c2) && vh_aligned16(w2) &&
                   vh_aligned16(b2) && vh_aligned16(out) && vh_aligned16(workspace),
               VH_EALIGN, "vh_ffn_decode: pointers must be 16-byte aligned");
    const int n_slices = dff / sw;
    VH_REQUIRE(workspace_bytes >= (size_t)n_slices * M * d_model * sizeof(float), VH_EINVAL,
               "vh_ffn_decode: workspace of %zu B < %zu B (vh_ffn_decode_ws_bytes)", workspace_bytes,
               (size_t)n_slices * M * d_model * sizeof(float));
    FfnArgs a{c1, c2, (float*)workspace, n_slices, (M + rg_rows - 1) / rg_rows, rg_rows, ln_eps};
    const dim3 grid(n_slices, a.n_rg);
    hipStream_t s = (hipStream_t)stream;
#define FFN(DD, SS)                                                                                                  \
    do {                                                                                                            \
        if (w16) hipLaunchKernelGGL((ffn_decode_kernel<DD, SS, true>), grid, dim3(512), 0, s, x, w1f, w2, ldx, M, dff, a); \
        else hipLaunchKernelGGL((ffn_decode_kernel<DD, SS>), grid, dim3(512), 0, s, x, w1f, w2, ldx, M, dff, a);           \
    } while (0)
#define FFN_D(DD)                              \
    do {                                       \
        if (sw == 16) FFN(DD, 16);             \
        else FFN(DD, 32);                      \
    } while (0)
    if (d_model == 128) FFN_D(128);
    else if (d_model == 256) FFN_D(256);
    else if (d_model == 512) FFN_D(512);
    else FFN(1024, 16);
#undef FFN_D
#undef FFN
    hipLaunchKernelGGL(ffn_reduce_kernel, dim3(M * (d_model / 64)), dim3(256), 0, s, (const float*)workspace, n_slices, x,
                       ldx, b2, out, ldo, M, d_model);
    VH_CHECK_LAUNCH("vh_ffn_decode");
    return VH_OK;
}

extern "C" int vh_ffn_decode(const float* x, int ldx, const float* w1f, const float* c1, const float* c2,
                             const float* w2, const float* b2, float* out, int ldo, int M, int d_model, int dff,
                             float ln_eps, void* workspace, size_t workspace_bytes, void* stream) {
    return ffn_decode_launch(x, ldx, w1f, c1, c2, w2, b2, out, ldo, M, d_model, dff, ln_eps, workspace, workspace_bytes, stream, false);
}

// the same launch pair over h16 weights (w1f16 (dff, d), w2_16 (d, dff): the decoder plan's perf mode, plan.hip)
int vh_internal_ffn_decode_w16(const float* x, int ldx, const uint16_t* w1f16, const float* c1, const float* c2,
                               const uint16_t* w2_16, const float* b2, float* out, int ldo, int M, int d_model, int dff,
                               float ln_eps, void* workspace, size_t workspace_bytes, void* stream) {
    return ffn_decode_launch(x, ldx, reinterpret_cast<const float*>(w1f16), c1, c2, reinterpret_cast<const float*>(w2_16), b2, out, ldo,
                             M, d_model, dff, ln_eps, workspace, workspace_bytes, stream, true);
}
