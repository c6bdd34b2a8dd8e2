// fp32 GEMMs of the path:  out = act(LN?(A) · Wᵀ + bias) + residual   (A (M,K), W (N,K))
//
// Two kernels, both on the exact-f32 matrix cores (no TF32 on gfx950):
//   * gemm_tile_kernel    M large (prefill / NAR / training): 128x128x32 LDS-tiled,
//                         v_mfma_f32_32x32x2_f32, 4 waves x (2x2 tiles of 32x32), register-staged
//                         double buffer, XCD-aware tile order.  MFMA-bound (157 TF peak).
//   * gemm_skinny_kernel  M <= 64 (decode step, heads, AdaLN vectors): weight-streaming,
//                         v_mfma_f32_16x16x4_f32 with W as the A operand so a lane's 4 results are
//                         4 consecutive output columns; K split over the waves of a block and
//                         reduced through LDS in fixed order (deterministic); optional LayerNorm
//                         fused into the activation operand load.  HBM/latency-bound.
//
// MFMA operand maps used (cdna_hip_programming.md §3):
//   32x32x2 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D reg r: row i=(r&3)+8(r>>2)+4(l>>5), col j=l&31
//   16x16x4 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r: row i=4(l>>4)+r,          col j=l&15
// The k index of a product is only summed over, so each lane feeds 4 consecutive k (one float4)
// per operand and the 4 MFMAs of a group consume element j of both operands: same k on both sides.
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
};

enum { EPI_PLAIN = 0, EPI_QKV = 1 };

// column group of 4 consecutive output columns starting at n (n % 4 == 0) for row m
template <int EPI>
__device__ __forceinline__ void store4(const GemmArgs& a, int m, int n, f32x4 v) {
    if (m >= a.M || n >= a.N) return;
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
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int k = k0 + skq;
            const bool kin = k < a.K;  // K % 4 == 0 is guaranteed by the host check
            ra[i] = (m0 + row < a.M && kin) ? ld4(a.A + (int64_t)(m0 + row) * a.lda + k)
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
            rw[i] = (n0 + row < a.N && kin) ? ld4(a.W + (int64_t)(n0 + row) * a.K + k)
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            st4(&lds[buf][0][row * LDS_LD + skq], ra[i]);
            st4(&lds[buf][1][row * LDS_LD + skq], rw[i]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (a.K + TK - 1) / TK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * TK);
        const float* As = &lds[cur][0][(wm * 64 + r) * LDS_LD + 4 * h];
        const float* Ws = &lds[cur][1][(wn * 64 + r) * LDS_LD + 4 * h];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a0 = ld4(As + 8 * t), a1 = ld4(As + 32 * LDS_LD + 8 * t);
            const f32x4 w0 = ld4(Ws + 8 * t), w1 = ld4(Ws + 32 * LDS_LD + 8 * t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], w0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], w1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], w0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], w1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: D reg e of tile (mt,nt): row = (e&3)+8(e>>2)+4h, col = r.  32 lanes of a half
    // write 32 consecutive floats of one row (128 B), the other half another row.
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int n = n0 + wn * 64 + nt * 32 + r;
                store1<EPI>(a, m, n, acc[mt][nt][e]);
            }
}

// =============================================================================================
// Skinny kernel: M <= 16*MT rows.  Block = NW waves, owns 16 output columns n0..n0+15 over all K.
// wave w takes k-steps of 16 (w, w+NW, ...).  lane: i = l&15 (W row / activation row), g = l>>4.
// =============================================================================================
template <int MT, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs a, LnFuse ln) {
    __shared__ __attribute__((aligned(16))) float red[NW][MT][64][4];
    __shared__ float s_mean[16 * MT], s_rstd[16 * MT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;

    if (ln.gamma) {  // row statistics of the M activation rows (two-pass, fp32)
        for (int row = w; row < a.M; row += NW) {
            const float* xr = a.A + (int64_t)row * a.lda;
            float s = 0.f;
            for (int c = lane * 4; c < a.K; c += 256) {
                const f32x4 v = ld4(xr + c);
                s += (v.x + v.y) + (v.z + v.w);
            }
            const float mean = wave_sum(s) / (float)a.K;
            float ss = 0.f;
            for (int c = lane * 4; c < a.K; c += 256) {
                const f32x4 v = ld4(xr + c) - mean;
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            const float var = wave_sum(ss) / (float)a.K;
            if (lane == 0) { s_mean[row] = mean; s_rstd[row] = rsqrtf(var + ln.eps); }
        }
        __syncthreads();
    }

    f32x4 acc[MT];
    float mean[MT], rstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = mt * 16 + i;
        mean[mt] = (ln.gamma && row < a.M) ? s_mean[row] : 0.f;
        rstd[mt] = (ln.gamma && row < a.M) ? s_rstd[row] : 0.f;
    }
    const bool nin = n0 + i < a.N;
    const float* wrow = a.W + (int64_t)(n0 + i) * a.K + 4 * g;
    const int nks = a.K / 16;
    for (int ks = w; ks < nks; ks += NW) {
        const int k = ks * 16;
        const f32x4 wf = nin ? ld4(wrow + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xf[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + i;
            xf[mt] = row < a.M ? ld4(a.A + (int64_t)row * a.lda + k + 4 * g)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (ln.gamma) {
            const f32x4 gm = ld4(ln.gamma + k + 4 * g), bt = ld4(ln.beta + k + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                xf[mt] = (xf[mt] - mean[mt]) * rstd[mt] * gm + bt;
                if (ln.ada_scale)
                    xf[mt] = ld4(ln.ada_scale + k + 4 * g) * xf[mt] + ld4(ln.ada_shift + k + 4 * g);
                if (mt * 16 + i >= a.M) xf[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[mt][j], acc[mt], 0, 0, 0);
    }
    // D[i=n][j=m]: lane holds m = mt*16 + (l&15), n = n0 + 4g + {0..3}
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st4(&red[w][mt][lane][0], acc[mt]);
    __syncthreads();
    for (int o = tid; o < MT * 64; o += NW * 64) {
        const int mt = o >> 6, l = o & 63;
        f32x4 s = ld4(&red[0][mt][l][0]);
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) s += ld4(&red[ww][mt][l][0]);
        store4<EPI>(a, mt * 16 + (l & 15), n0 + 4 * (l >> 4), s);
    }
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
    VH_REQUIRE(!ln.gamma || a.M <= 64, VH_EUNSUPPORTED,
               "%s: fused LayerNorm only for M <= 64 (M=%d); run vh_layernorm first", name, a.M);
    VH_REQUIRE(vh_aligned16(ln.gamma) && vh_aligned16(ln.beta) && vh_aligned16(ln.ada_scale) &&
                   vh_aligned16(ln.ada_shift),
               VH_EALIGN, "%s: LayerNorm vectors must be 16-byte aligned", name);
    return VH_OK;
}

template <int EPI>
static int launch_gemm(const char* name, const GemmArgs& a, const LnFuse& ln, hipStream_t s) {
    if (a.M == 0) return VH_OK;
    if (a.M <= 64) {
        const int mt = (a.M + 15) / 16;
        dim3 grid((a.N + 15) / 16);
        // K split over 8 waves up to K=1024, 16 waves beyond (e.g. linear_2: K = dff = 2048)
        const bool wide = a.K > 1024;
#define SK(MT, NW) hipLaunchKernelGGL((gemm_skinny_kernel<MT, NW, EPI>), grid, dim3(NW * 64), 0, s, a, ln)
        if (mt == 1) { if (wide) SK(1, 16); else SK(1, 8); }
        else if (mt == 2) { if (wide) SK(2, 16); else SK(2, 8); }
        else { if (wide) SK(4, 16); else SK(4, 8); }
#undef SK
    } else {
        const int tm = (a.M + TM - 1) / TM, tn = (a.N + TN - 1) / TN;
        hipLaunchKernelGGL((gemm_tile_kernel<EPI>), dim3(tm * tn), dim3(256), 0, s, a, tm, tn);
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
    a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.act = act;
    LnFuse ln{ln_gamma, ln_beta, ada_scale, ada_shift, ln_eps};
    VH_REQUIRE(act == VH_ACT_NONE || act == VH_ACT_GELU_ERF, VH_EINVAL, "vh_linear: act=%d", act);
    VH_REQUIRE(ldo >= N && (!residual || ldr >= N), VH_EINVAL, "vh_linear: ldo/ldr < N");
    if (int rc = check_gemm("vh_linear", a, ln)) return rc;
    return launch_gemm<EPI_PLAIN>("vh_linear", a, ln, (hipStream_t)stream);
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
    a.K = d_model; a.act = VH_ACT_NONE; a.kc = kcache; a.vc = vcache; a.cache_len = cache_len;
    a.T = T > 0 ? T : 1; a.S_max = S_max; a.d_model = d_model; a.n_heads = n_heads;
    LnFuse ln{ln_gamma, ln_beta, ada_scale, ada_shift, ln_eps};
    if (int rc = check_gemm("vh_linear_qkv", a, ln)) return rc;
    return launch_gemm<EPI_QKV>("vh_linear_qkv", a, ln, (hipStream_t)stream);
}
