// Perf mode of the MFMA-bound legs (prompt pass, NAR stage forward): bf16 operands on v_mfma_f32_32x32x16_bf16, fp32
// accumulators, fp32 residual stream.  SECONDARY by construction (include/valle_hip.h "perf mode"): the parity path is
// the fp32 one; teacher-forced logits of this path agree with the reference to atol 5e-2.
//
//   gemm16_tile_kernel   out = act(A W^T + bias) + residual with A (M,K) and W (N,K) bf16: the LDS-DMA tile machine of
//                        gemm.hip in bytes — 128 x 128 tile, K step 64 (a row of a slab is 128 B, as the fp32 kernel's
//                        32 floats), 4 waves x (2 x 2) 32x32x16 MFMAs, XOR-swizzled lane-linear LDS image, 2 workgroups
//                        per CU; epilogue through LDS: fp32 out (+ fp32 residual) or bf16 out, or the QKV scatter (q and
//                        the K / V cache rows as bf16).
//   attn16_kernel        flash attention over bf16 q / K / V: S^T = K Q^T (a query is a lane), online softmax in fp32 on
//                        the accumulator registers, P^T narrowed in registers into the B operand of O^T = V^T P^T (keys of
//                        a k-step in accumulator order), V^T fragments by ds_read_b64_tr_b16 from the row-major V tile.
//   layernorm16_kernel   LayerNorm / AdaptiveLayerNorm of the fp32 residual stream written as bf16.
//
// MFMA operand maps (cdna_hip_programming.md section 3), lane l: r = l & 31, h = l >> 5:
//   32x32x16 bf16: A[i = r][k = 8h + j], B[k = 8h + j][col = r], j = 0..7 (one 16-byte fragment each);
//   D reg x: row (x & 3) + 8 (x >> 2) + 4 h, col r.
#include <type_traits>

#include "bf16_common.h"

// ---------------------------------------------------------------------------------------------
// fp32 -> bf16 of a (rows, cols) matrix (weights once per weights epoch; cols % 8 == 0)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ src, int64_t lds_, uint16_t* __restrict__ dst,
                                                      int64_t ldd, int64_t rows, int c8n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8n) return;
    const int64_t r = i / c8n;
    const int c = (int)(i - r * c8n) * 8;
    const f32x4 a = ld4(src + r * lds_ + c), b = ld4(src + r * lds_ + c + 4);
    stq(dst + r * ldd + c, u32x4{pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w)});
}

extern "C" int vh_to_bf16(const float* src, int64_t lds_, uint16_t* dst, int64_t ldd, int64_t rows, int cols, void* stream) {
    VH_REQUIRE(src && dst, VH_EINVAL, "vh_to_bf16: null pointer");
    VH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && lds_ % 4 == 0 && ldd % 8 == 0 && lds_ >= cols && ldd >= cols, VH_EINVAL,
               "vh_to_bf16: rows=%lld cols=%d (cols %% 8 == 0, leading dimensions multiples of 4 / 8)", (long long)rows, cols);
    VH_REQUIRE(vh_aligned16(src) && vh_aligned16(dst), VH_EALIGN, "vh_to_bf16: pointers must be 16-byte aligned");
    if (rows == 0) return VH_OK;
    const int64_t n = rows * (cols / 8);
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, lds_, dst, ldd,
                       rows, cols / 8);
    VH_CHECK_LAUNCH("vh_to_bf16");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm / AdaptiveLayerNorm, fp32 in, bf16 out: one wave per row, two-pass statistics in registers (as
// layernorm_kernel of elementwise.hip), 8 columns per lane and chunk.
// ---------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ ada_scale,
                                                          const float* __restrict__ ada_shift, uint16_t* __restrict__ out,
                                                          int rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * d;
    f32x4 v[NV][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 8;
        v[i][0] = c < d ? ld4(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        v[i][1] = c < d ? ld4(xr + c + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((lane + 64 * i) * 8 < d) {
            const f32x4 t0 = v[i][0] - mean, t1 = v[i][1] - mean;
            ss += ((t0.x * t0.x + t0.y * t0.y) + (t0.z * t0.z + t0.w * t0.w)) +
                  ((t1.x * t1.x + t1.y * t1.y) + (t1.z * t1.z + t1.w * t1.w));
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    uint16_t* orow = out + (int64_t)row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
            f32x4 y0 = (v[i][0] - mean) * rstd * ld4(gamma + c) + ld4(beta + c);
            f32x4 y1 = (v[i][1] - mean) * rstd * ld4(gamma + c + 4) + ld4(beta + c + 4);
            if (ada_scale) {
                y0 = ld4(ada_scale + c) * y0 + ld4(ada_shift + c);
                y1 = ld4(ada_scale + c + 4) * y1 + ld4(ada_shift + c + 4);
            }
            stq(orow + c, u32x4{pack_bf16(y0.x, y0.y), pack_bf16(y0.z, y0.w), pack_bf16(y1.x, y1.y), pack_bf16(y1.z, y1.w)});
        }
    }
}

extern "C" int vh_layernorm_bf16(const float* x, const float* gamma, const float* beta, const float* ada_scale,
                                 const float* ada_shift, uint16_t* out, int rows, int d, float eps, void* stream) {
    VH_REQUIRE(x && gamma && beta && out, VH_EINVAL, "vh_layernorm_bf16: null pointer");
    VH_REQUIRE((ada_scale == nullptr) == (ada_shift == nullptr), VH_EINVAL,
               "vh_layernorm_bf16: ada_scale and ada_shift must be given together");
    VH_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0 && d <= 4096, VH_EINVAL,
               "vh_layernorm_bf16: bad dims rows=%d d=%d (d multiple of 8, <= 4096)", rows, d);
    VH_REQUIRE(vh_aligned16(x) && vh_aligned16(out) && vh_aligned16(gamma) && vh_aligned16(beta) && vh_aligned16(ada_scale) &&
                   vh_aligned16(ada_shift), VH_EALIGN, "vh_layernorm_bf16: pointers must be 16-byte aligned");
    if (rows == 0) return VH_OK;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LN16(NV) hipLaunchKernelGGL(layernorm16_kernel<NV>, grid, block, 0, s, x, gamma, beta, ada_scale, ada_shift, out, rows, d, eps)
    if (d <= 512) LN16(1);
    else if (d <= 1024) LN16(2);
    else if (d <= 2048) LN16(4);
    else LN16(8);
#undef LN16
    VH_CHECK_LAUNCH("vh_layernorm_bf16");
    return VH_OK;
}

// =============================================================================================
// bf16 tile GEMM
// =============================================================================================
#define T16M 128
#define T16N 128
#define T16K 64                       // bf16 elements per K step: 128 B per row of a slab
#define SLAB16 (T16M * 128)           // bytes of one operand slab (128 rows x 128 B)
#define EPI16_LD 132                  // floats per row of the epilogue's transposition image

#ifdef VH_TILE_PROBE16
// Phase stamps (tools/probe_tile16.hip only): 100 MHz wall clock at entry | first slab landed | main loop done | image
// written | stores issued | stores acknowledged, + the hardware id, per workgroup.
__device__ long long vh_probe16[16384 * 8];
__device__ unsigned vh_hwid16[16384 * 2];
#define VH_P16(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) vh_probe16[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define VH_P16_END() do { VH_P16(4); __builtin_amdgcn_s_waitcnt(0x0F70); VH_P16(5); \
        if (threadIdx.x == 0 && blockIdx.x < 16384) { vh_hwid16[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); \
                                                       vh_hwid16[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); } } while (0)
#else
#define VH_P16(k) do { } while (0)
#define VH_P16_END() do { } while (0)
#endif

template <int OUT>
__global__ __launch_bounds__(256, 2) void gemm16_tile_kernel(Gemm16Args a, int tiles_m, int tiles_n) {
    // [buf][A | W][128 rows][128 B] for the main loop (64 KB); the epilogue re-uses it as a 128 x 132 fp32 image
    __shared__ __attribute__((aligned(16))) char lds[T16M * EPI16_LD * 4];
    __builtin_amdgcn_s_setprio(3);
    VH_P16(0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;

    // XCD-aware tile order (gemm.hip): blocks b and b + 8 share an XCD; each XCD gets a contiguous run of tiles,
    // numbered n-fastest so a run shares its A row panel in L2
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int m0 = (tile / tiles_n) * T16M, n0 = (tile % tiles_n) * T16N;

    // LDS-DMA staging: wave w issues pieces q = 8w .. 8w+7 of a slab pair (16 pieces of A, then 16 of W; a piece = 8 rows
    // x 128 B = 1 KiB per wave-instruction).  Lane L fills LDS slot (row L >> 3 of the group, 16-byte chunk L & 7): it must
    // fetch chunk (L & 7) ^ swz(row) of that row — the swizzle lives in the SOURCE address, the image stays lane-linear.
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const char* baseA = (const char*)(a.A + (int64_t)m0 * a.lda);
    const char* baseW = (const char*)(a.W + (int64_t)n0 * a.K);
    uint32_t voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = ws * 8 + i, row = (q & 15) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = q < 16 ? (uint32_t)(min(row, a.M - 1 - m0) * a.lda + 8 * c) * 2u
                         : (uint32_t)(min(row, a.N - 1 - n0) * a.K + 8 * c) * 2u;
    }
    auto dma1 = [&](int i, int buf, int k0) {
        const int q = ws * 8 + i;
        const char* base = (q < 16 ? baseA : baseW) + (int64_t)k0 * 2;
        const uint32_t dst = (uint32_t)(uintptr_t)(lds + buf * 2 * SLAB16 + (q >> 4) * SLAB16 + (q & 15) * 1024);
        // scalar-base form (see gemm.hip: M0 = LDS destination, one 32-bit per-lane offset, a wave-uniform 64-bit base)
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

    const int nk = a.K / T16K;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma1(i, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __syncthreads();
    VH_P16(1);

    // fragment addresses in buffer 0: row (wm | wn) * 64 + r, chunk (2 t + h) ^ swz — one ds_read_b128 per fragment
    const int swz = (r >> 1) & 7;
    const char* fA[4];
    const char* fW[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fA[t] = lds + (wm * 64 + r) * 128 + (((2 * t + h) ^ swz) << 4);
        fW[t] = lds + SLAB16 + (wn * 64 + r) * 128 + (((2 * t + h) ^ swz) << 4);
    }
    bf16x8 fa[2][2], fw[2][2];
    auto fload = [&](int set, int buf, int t) {
        fa[set][0] = __builtin_bit_cast(bf16x8, ldq(fA[t] + buf * 2 * SLAB16));
        fa[set][1] = __builtin_bit_cast(bf16x8, ldq(fA[t] + buf * 2 * SLAB16 + 32 * 128));
        fw[set][0] = __builtin_bit_cast(bf16x8, ldq(fW[t] + buf * 2 * SLAB16));
        fw[set][1] = __builtin_bit_cast(bf16x8, ldq(fW[t] + buf * 2 * SLAB16 + 32 * 128));
    };
    __builtin_amdgcn_s_setprio(0);
    fload(0, 0, 0);
    // One K step = four groups of 4 MFMAs (one per 16-wide k slice).  The other register set is read from LDS one MFMA
    // into the group; the next slab's DMA is requested in groups 0 and 1 (four pieces each) into the other buffer (free
    // since the previous step's barrier); the step's barrier sits inside group 3 after this wave's last read of the buffer.
    auto kstep = [&](int kt, auto cur_c, auto pf) {
        constexpr bool PF = decltype(pf)::value;
        constexpr int cur = decltype(cur_c)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int s = t & 1;
            acc[0][0] = VH_MFMA16(fa[s][0], fw[s][0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF) {                           // all eight pieces in the first two groups: a piece requested in
                if (t < 2) {                              // the last group would be waited for the moment it is issued
#pragma unroll
                    for (int i = 0; i < 4; ++i) dma1(4 * t + i, cur ^ 1, (kt + 1) * T16K);
                }
            }
            if (t < 3) {
                fload(s ^ 1, cur, t + 1);
            } else if constexpr (PF) {
                __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this wave's pieces of the next slab have landed
                __syncthreads();
                fload(0, cur ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = VH_MFMA16(fa[s][0], fw[s][1], acc[0][1]);
            acc[1][0] = VH_MFMA16(fa[s][1], fw[s][0], acc[1][0]);
            acc[1][1] = VH_MFMA16(fa[s][1], fw[s][1], acc[1][1]);
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
    __syncthreads();                                        // every wave has read its last fragments: LDS is free
    __builtin_amdgcn_s_setprio(3);
    VH_P16(2);

    // ---- epilogue: accumulators transposed through LDS so that a lane owns consecutive columns of one row ----
    float* ct = (float*)lds;
    // fp32 output with a residual: its 16 row groups are requested NOW, so that they fly under the transposition (the
    // fragment registers are dead; interior tiles only — a ragged tile fetches in its store loop)
    f32x4 resv[16];
    const bool res_early = OUT == G16_F32 && a.res && m0 + T16M <= a.M;
    if (OUT == G16_F32 && res_early) {
        const float* rb = a.res + (int64_t)(m0 + (tid >> 5)) * a.ldr + n0 + 4 * (tid & 31);
#pragma unroll
        for (int it = 0; it < 16; ++it) resv[it] = ld4(rb + (int64_t)it * 8 * a.ldr);
    }
    {
        float* cw = ct + (wm * 64 + 4 * h) * EPI16_LD + wn * 64 + r;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int x = 0; x < 16; ++x)
                    cw[(mt * 32 + (x & 3) + 8 * (x >> 2)) * EPI16_LD + nt * 32] = acc[mt][nt][x];
    }
    __syncthreads();
    VH_P16(3);
    if (OUT == G16_F32) {
        // thread = (column group of 4: tid & 31, row tid >> 5 + 8 it): whole 512-B rows per wave-instruction
        const int ec4 = tid & 31, erow = tid >> 5, en = n0 + 4 * ec4;
        const f32x4 bias4 = a.bias ? ld4(a.bias + en) : f32x4{0.f, 0.f, 0.f, 0.f};
        float* out = (float*)a.out;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = m0 + erow + 8 * it;
            if (m >= a.M) break;
            f32x4 v = ld4(ct + (erow + 8 * it) * EPI16_LD + 4 * ec4) + bias4;
            if (a.act == VH_ACT_GELU_ERF) {
                const vh_f32x2 g0 = gelu_erf2(vh_f32x2{v.x, v.y}), g1 = gelu_erf2(vh_f32x2{v.z, v.w});
                v = f32x4{g0.x, g0.y, g1.x, g1.y};
            }
            if (res_early) v += resv[it];
            else if (a.res) v += ld4(a.res + (int64_t)m * a.ldr + en);
            st4(out + (int64_t)m * a.ldo + en, v);
        }
        VH_P16_END();
        return;
    }
    // bf16 outputs: thread = (column group of 8: tid & 15, row tid >> 4 + 16 it): 256-B row segments, 16 B per lane
    const int ec8 = tid & 15, erow = tid >> 4, en = n0 + 8 * ec8;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (OUT == G16_BF16 && a.bias) { b0 = ld4(a.bias + en); b1 = ld4(a.bias + en + 4); }
    uint16_t* dst = (uint16_t*)a.out;
    int64_t dstride = a.ldo;                                // elements between consecutive rows (q / plain output)
    bool cache = false;
    int b = 0, t = 0, cl = 0;
    float qs = 1.f;                                         // q leaves pre-scaled (VH_Q16_PRESCALE)
    if (OUT == G16_QKV) {
        const int which = en / a.d_model, c = en - which * a.d_model;     // d_model % 128 == 0: a tile lies in one of q | K | V
        if (which == 0) {
            qs = VH_Q16_PRESCALE;
            dst = (uint16_t*)a.out + c;
        } else {
            cache = true;
            dst = (which == 1 ? a.kc : a.vc) + (int64_t)(c / VH_HEAD_DIM) * a.S_max * VH_HEAD_DIM + (c % VH_HEAD_DIM);
            const int m = m0 + erow;
            b = m / a.T;
            t = m - b * a.T;
            cl = (a.cache_len && m < a.M) ? a.cache_len[b] : 0;
        }
    } else {
        dst += en;
    }
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int m = m0 + erow + 16 * it;
        if (m >= a.M) break;
        const float* cr = ct + (erow + 16 * it) * EPI16_LD + 8 * ec8;
        f32x4 v0 = ld4(cr) + b0, v1 = ld4(cr + 4) + b1;
        if (OUT == G16_BF16 && a.act == VH_ACT_GELU_ERF) {
            const vh_f32x2 g0 = gelu16_2(vh_f32x2{v0.x, v0.y}), g1 = gelu16_2(vh_f32x2{v0.z, v0.w});
            const vh_f32x2 g2 = gelu16_2(vh_f32x2{v1.x, v1.y}), g3 = gelu16_2(vh_f32x2{v1.z, v1.w});
            v0 = f32x4{g0.x, g0.y, g1.x, g1.y};
            v1 = f32x4{g2.x, g2.y, g3.x, g3.y};
        }
        if (OUT == G16_QKV) { v0 = v0 * qs; v1 = v1 * qs; }
        const u32x4 pk = {pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
        if (cache) {
            stq(dst + ((int64_t)b * a.n_heads * a.S_max + cl + t) * VH_HEAD_DIM, pk);
            t += 16;
            if (t >= a.T) {                                 // next batch row (several at once only when T < 16)
                do { t -= a.T; ++b; } while (t >= a.T);
                if (a.cache_len && m + 16 < a.M) cl = a.cache_len[b];
            }
        } else {
            stq(dst + (int64_t)m * dstride, pk);
        }
    }
    VH_P16_END();
}

// =============================================================================================
// Occupancy variant of the bf16 tile GEMM (round 5, third form): ONE slab pair of K step 64 (32 KB of LDS), no software
// pipeline at all — request the slab, wait, barrier, 16 MFMAs per wave, barrier — and FOUR workgroups per CU (<= 128 VGPRs)
// whose phases interleave by themselves: while one workgroup waits for its slab the other three multiply.
// Why (tools/probe_tile16.hip, profiles/r5_probe_tile16.log): in the two-slab form a K step lasts 0.8 - 1.15 us against
// 0.21 us of MFMA work — the step is the latency of the ONE slab it has in flight — and a CU holds two such workgroups
// (64 KB in flight); here it holds four slabs in flight (128 KB) behind half the LDS, and the prologue / epilogue of a
// tile hide under three neighbours instead of one.  (cdna_hip_programming.md section 5: the plain two-barrier 128^2
// structure at >= 3 blocks per CU is the best of its class; explicit double buffering does not beat it.)
// Epilogue: the 32 KB hold half of the tile's fp32 image — two passes of 64 rows (the waves of row half wm = pass hand
// over), rows of 128 floats, columns XOR-ed with 32 on every other group of four rows (the two row halves of a 32x32
// accumulator block, rows r and r + 4, would otherwise share their banks).
// =============================================================================================
template <int OUT>
__global__ __launch_bounds__(256, 4) void gemm16_occ_kernel(Gemm16Args a, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char lds[2 * SLAB16];    // [A | W][128 rows][128 B]; epilogue: 64 x 128 floats
    VH_P16(0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int m0 = (tile / tiles_n) * T16M, n0 = (tile % tiles_n) * T16N;

    const int ws = __builtin_amdgcn_readfirstlane(w);
    const char* baseA = (const char*)(a.A + (int64_t)m0 * a.lda);
    const char* baseW = (const char*)(a.W + (int64_t)n0 * a.K);
    uint32_t voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = ws * 8 + i, row = (q & 15) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = q < 16 ? (uint32_t)(min(row, a.M - 1 - m0) * a.lda + 8 * c) * 2u
                         : (uint32_t)(min(row, a.N - 1 - n0) * a.K + 8 * c) * 2u;
    }
    auto dma1 = [&](int i, int k0) {
        const int q = ws * 8 + i;
        const char* base = (q < 16 ? baseA : baseW) + (int64_t)k0 * 2;
        const uint32_t dst = (uint32_t)(uintptr_t)(lds + (q >> 4) * SLAB16 + (q & 15) * 1024);
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

    const int swz = (r >> 1) & 7;
    const char* fA = lds + (wm * 64 + r) * 128;
    const char* fW = lds + SLAB16 + (wn * 64 + r) * 128;
    const int nk = a.K / T16K;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma1(i, kt * T16K);
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this wave's pieces have landed
        __syncthreads();
        if (kt == 0) VH_P16(1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = ((2 * t + h) ^ swz) << 4;
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, ldq(fA + off)), a1 = __builtin_bit_cast(bf16x8, ldq(fA + off + 32 * 128));
            const bf16x8 w0 = __builtin_bit_cast(bf16x8, ldq(fW + off)), w1 = __builtin_bit_cast(bf16x8, ldq(fW + off + 32 * 128));
            acc[0][0] = VH_MFMA16(a0, w0, acc[0][0]);
            acc[0][1] = VH_MFMA16(a0, w1, acc[0][1]);
            acc[1][0] = VH_MFMA16(a1, w0, acc[1][0]);
            acc[1][1] = VH_MFMA16(a1, w1, acc[1][1]);
        }
        __syncthreads();                                    // every wave has read the slab: it may be overwritten
    }

    VH_P16(2);
    // ---- epilogue: two passes of 64 rows through the 32 KB ----
    float* ct = (float*)lds;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // per-thread constants of the store side
    const int ec4 = tid & 31, erow4 = tid >> 5;             // fp32 output: column group of 4, rows erow4 + 8 it
    const int ec8 = tid & 15, erow8 = tid >> 4;             // bf16 outputs: column group of 8, rows erow8 + 16 it
    f32x4 b0 = zero4, b1 = zero4;
    if (OUT == G16_F32 && a.bias) b0 = ld4(a.bias + n0 + 4 * ec4);
    if (OUT == G16_BF16 && a.bias) { b0 = ld4(a.bias + n0 + 8 * ec8); b1 = ld4(a.bias + n0 + 8 * ec8 + 4); }
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int mp = m0 + 64 * pass;
        if (wm == pass) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int x = 0; x < 16; ++x) {
                        const int lrow = mt * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;      // (lrow >> 2) & 1 == h
                        ct[lrow * 128 + ((wn * 64 + nt * 32 + r) ^ (h << 5))] = acc[mt][nt][x];
                    }
        }
        __syncthreads();
        if (pass == 0) VH_P16(3);
        if (OUT == G16_F32) {
            float* out = (float*)a.out;
            const int en = n0 + 4 * ec4;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int lrow = erow4 + 8 * it, m = mp + lrow;
                if (m >= a.M) break;
                f32x4 v = ld4(ct + lrow * 128 + ((4 * ec4) ^ (((lrow >> 2) & 1) << 5))) + b0;
                if (a.act == VH_ACT_GELU_ERF) {
                    const vh_f32x2 g0 = gelu_erf2(vh_f32x2{v.x, v.y}), g1 = gelu_erf2(vh_f32x2{v.z, v.w});
                    v = f32x4{g0.x, g0.y, g1.x, g1.y};
                }
                if (a.res) v += ld4(a.res + (int64_t)m * a.ldr + en);      // (no early request: three neighbours hide it)
                st4(out + (int64_t)m * a.ldo + en, v);
            }
        } else {
            const int en = n0 + 8 * ec8;
            const int which = OUT == G16_QKV ? en / a.d_model : 0, c = en - which * a.d_model;   // a tile lies in one of q | K | V
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int lrow = erow8 + 16 * it, m = mp + lrow;
                if (m >= a.M) break;
                const float* cr = ct + lrow * 128 + ((8 * ec8) ^ (((lrow >> 2) & 1) << 5));
                f32x4 v0 = ld4(cr) + b0, v1 = ld4(cr + 4) + b1;
                if (OUT == G16_BF16 && a.act == VH_ACT_GELU_ERF) {
                    const vh_f32x2 g0 = gelu16_2(vh_f32x2{v0.x, v0.y}), g1 = gelu16_2(vh_f32x2{v0.z, v0.w});
                    const vh_f32x2 g2 = gelu16_2(vh_f32x2{v1.x, v1.y}), g3 = gelu16_2(vh_f32x2{v1.z, v1.w});
                    v0 = f32x4{g0.x, g0.y, g1.x, g1.y};
                    v1 = f32x4{g2.x, g2.y, g3.x, g3.y};
                }
                if (OUT == G16_QKV && which == 0) { v0 = v0 * VH_Q16_PRESCALE; v1 = v1 * VH_Q16_PRESCALE; }      // q leaves pre-scaled
                const u32x4 pk = {pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
                if (OUT == G16_QKV && which != 0) {
                    const int b = m / a.T, t = m - b * a.T;
                    const int cl = a.cache_len ? a.cache_len[b] : 0;
                    uint16_t* base = (which == 1 ? a.kc : a.vc) + (int64_t)(c / VH_HEAD_DIM) * a.S_max * VH_HEAD_DIM + (c % VH_HEAD_DIM);
                    stq(base + ((int64_t)b * a.n_heads * a.S_max + cl + t) * VH_HEAD_DIM, pk);
                } else {
                    stq((uint16_t*)a.out + (int64_t)m * a.ldo + (OUT == G16_QKV ? c : en), pk);
                }
            }
        }
        if (pass == 0) __syncthreads();                     // the image is read before the second half overwrites it
    }
    VH_P16_END();
}

// (round 5's third machine — a ring of three slabs of 32 k with counted waits, three workgroups per CU — was slower than the two-slab
// form on 7 of 8 shapes and is gone in round 6: docs/history.md; VH_TUNE_BF16_GEMM = 2 now means the default)
// default (set by measurement, tools/bench_bf16.py + tools/probe_tile16.hip): the one-slab / four-workgroup form for the
// bf16 outputs (QKV scatter, linear_1 + GELU: -10 ... -25 %), the two-slab form with its early residual request for the
// fp32 output + residual GEMMs (HBM-bound at 3.7 - 4.8 TB/s in both forms)
static bool use_occ16(bool bf16_out) {
    const int knob = vh_tuning(VH_TUNE_BF16_GEMM);
    return (knob == 0 || knob == 2) ? bf16_out : knob == 3;
}
// round 6: the persistent 256^2 / 8-wave form (gemm16p.hip) where the shape allows it (N % 256 == 0, K % 128 == 0) and there
// are enough rows to fill the part; VH_TUNE_BF16_GEMM = 4 forces it for every shape it takes, 1 / 2 / 3 keep the 128^2 forms
static bool use_p256(const Gemm16Args& a, int out_kind) {
    const int knob = vh_tuning(VH_TUNE_BF16_GEMM);
    if (knob != 0 && knob != 2 && knob != 4) return false;
    if (!vh_gemm16_p256_ok(a, out_kind)) return false;
    return knob == 4 || (int64_t)((a.M + 255) / 256) * (a.N / 256) >= 128;
}

static int check_gemm16(const char* name, const Gemm16Args& a) {
    VH_REQUIRE(a.A && a.W && a.out, VH_EINVAL, "%s: null pointer", name);
    VH_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.N % T16N == 0 && a.K % T16K == 0, VH_EUNSUPPORTED,
               "%s: M=%d N=%d K=%d (the bf16 tile kernel needs N %% 128 == 0 and K %% 64 == 0)", name, a.M, a.N, a.K);
    VH_REQUIRE(a.lda % 8 == 0 && a.lda >= a.K && a.ldo % 4 == 0 && (!a.res || (a.ldr % 4 == 0 && a.ldr >= a.N)), VH_EINVAL,
               "%s: leading dimensions lda=%d ldo=%d ldr=%d", name, a.lda, a.ldo, a.ldr);
    VH_REQUIRE((int64_t)128 * a.lda * 2 < (1ll << 31) && (int64_t)128 * a.K * 2 < (1ll << 31), VH_EUNSUPPORTED,
               "%s: rows too long for 32-bit tile offsets", name);
    VH_REQUIRE(vh_aligned16(a.A) && vh_aligned16(a.W) && vh_aligned16(a.out) && vh_aligned16(a.bias) && vh_aligned16(a.res),
               VH_EALIGN, "%s: pointers must be 16-byte aligned", name);
    VH_REQUIRE(a.act == VH_ACT_NONE || a.act == VH_ACT_GELU_ERF, VH_EUNSUPPORTED, "%s: act=%d", name, a.act);
    return VH_OK;
}

extern "C" int vh_linear_bf16(const uint16_t* A, int lda, const uint16_t* W, const float* bias, const float* residual,
                              int ldr, void* out, int ldo, int out_bf16, int M, int N, int K, int act, void* stream) {
    Gemm16Args a{};
    a.A = A; a.lda = lda; a.W = W; a.bias = bias; a.res = residual; a.ldr = ldr; a.out = out; a.ldo = ldo;
    a.M = M; a.N = N; a.K = K; a.act = act;
    if (int rc = check_gemm16("vh_linear_bf16", a)) return rc;
    VH_REQUIRE(ldo >= N && (!out_bf16 || (ldo % 8 == 0 && !residual)), VH_EINVAL,
               "vh_linear_bf16: ldo=%d (bf16 output: ldo %% 8 == 0, no residual)", ldo);
    const int tm = (M + T16M - 1) / T16M, tn = N / T16N;
    hipStream_t s = (hipStream_t)stream;
    if (use_p256(a, out_bf16 ? G16_BF16 : G16_F32)) return vh_gemm16_p256_launch(a, out_bf16 ? G16_BF16 : G16_F32, s);
    if (use_occ16(out_bf16 != 0)) {
        if (out_bf16) hipLaunchKernelGGL(gemm16_occ_kernel<G16_BF16>, dim3(tm * tn), dim3(256), 0, s, a, tm, tn);
        else hipLaunchKernelGGL(gemm16_occ_kernel<G16_F32>, dim3(tm * tn), dim3(256), 0, s, a, tm, tn);
    } else {
        if (out_bf16) hipLaunchKernelGGL(gemm16_tile_kernel<G16_BF16>, dim3(tm * tn), dim3(256), 0, s, a, tm, tn);
        else hipLaunchKernelGGL(gemm16_tile_kernel<G16_F32>, dim3(tm * tn), dim3(256), 0, s, a, tm, tn);
    }
    VH_CHECK_LAUNCH("vh_linear_bf16");
    return VH_OK;
}

extern "C" int vh_linear_qkv_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv, uint16_t* q_out, int ldq,
                                  uint16_t* kcache16, uint16_t* vcache16, const int32_t* cache_len, int B, int T,
                                  int d_model, int n_heads, int S_max, void* stream) {
    Gemm16Args a{};
    a.A = A; a.lda = lda; a.W = Wqkv; a.out = q_out; a.ldo = ldq;
    a.M = B * T; a.N = 3 * d_model; a.K = d_model; a.act = VH_ACT_NONE;
    a.kc = kcache16; a.vc = vcache16; a.cache_len = cache_len; a.T = T; a.S_max = S_max; a.d_model = d_model; a.n_heads = n_heads;
    VH_REQUIRE(kcache16 && vcache16 && B > 0 && T > 0 && T <= S_max, VH_EINVAL, "vh_linear_qkv_bf16: bad cache / dims");
    VH_REQUIRE(d_model == n_heads * VH_HEAD_DIM && d_model % T16N == 0, VH_EUNSUPPORTED,
               "vh_linear_qkv_bf16: d_model=%d (n_heads x 64, a multiple of 128)", d_model);
    if (int rc = check_gemm16("vh_linear_qkv_bf16", a)) return rc;
    VH_REQUIRE(ldq % 8 == 0 && ldq >= d_model && vh_aligned16(kcache16) && vh_aligned16(vcache16), VH_EALIGN,
               "vh_linear_qkv_bf16: ldq=%d / cache alignment", ldq);
    const int tm = (a.M + T16M - 1) / T16M, tn = a.N / T16N;
    if (use_p256(a, G16_QKV)) return vh_gemm16_p256_launch(a, G16_QKV, (hipStream_t)stream);
    if (use_occ16(true)) hipLaunchKernelGGL(gemm16_occ_kernel<G16_QKV>, dim3(tm * tn), dim3(256), 0, (hipStream_t)stream, a, tm, tn);
    else hipLaunchKernelGGL(gemm16_tile_kernel<G16_QKV>, dim3(tm * tn), dim3(256), 0, (hipStream_t)stream, a, tm, tn);
    VH_CHECK_LAUNCH("vh_linear_qkv_bf16");
    return VH_OK;
}

// =============================================================================================
// bf16 flash attention (many rows): prompt pass / NAR stage forward of perf mode
// =============================================================================================
// Workgroup = 4 waves x 32 queries of one (batch row, head); 64-key K / V tiles, register-staged into a double-buffered
// LDS image (one barrier per tile).  Images: rows of 128 B (64 bf16), 16-byte chunk c of key row k stored at chunk
//   K: c ^ ((k >> 1) & 7)      (ds_read_b128 of the S = K Q^T A operand: conflict-free, as the GEMM's slabs)
//   V: c ^ (((k >> 1) & 1) << 2) (ds_read_b64_tr_b16 of the O^T = V^T P^T A operand: the four key rows of a block land on
//                                 disjoint bank groups)
#define A16_QB 128
#define A16_KT 64
#define A16_RING 3
#define A16_LAZY 8.0f      // a row's reference point moves when a tile's maximum exceeds it by more than this (base-2 exponent)
#define A16_NEG (-1e30f)

struct Attn16Args {
    const uint16_t* q;
    int ldq;
    const uint16_t* kc;
    const uint16_t* vc;
    uint16_t* out;
    int ldo;
    int B, n_heads, Tq, Tk, S_max, mode, x_len;
    const int32_t* x_len_dev;
    const int32_t* kv_len;
    int n_qb;
};

__global__ __launch_bounds__(256, 2) void attn16_kernel(Attn16Args a) {
    __shared__ __attribute__((aligned(16))) char lds[A16_RING * 2 * A16_KT * 128];   // ring of [K | V][64 keys][128 B] = 48 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // Workgroup ids go round-robin over the 8 XCDs, each with its own L2; the grid is padded to 8 x ceil(n_bh / 8) (batch row,
    // head) pairs x n_qb query blocks.  FULL mask (every block of a pair costs the same): XCD x takes the pairs x, x + 8, ... and
    // runs the query blocks of ONE pair back to back, so the pair's K / V (2 x Tk x 128 B) is fetched once into that L2 and
    // re-read from there by the other blocks (NAR-stage shape: 0.93x the time of the order below).  PREFIX mask: the last
    // query block of a pair sees up to 4x the keys of the first, and the launch wants the heaviest blocks of ALL pairs
    // first (grouped by pair it took 1.4x the time at the prompt-pass shape); a pair's blocks still share an XCD.
    const int n_bh8 = (a.B * a.n_heads + 7) & ~7;
    int bh, qb;
    // (grouped only when the pairs deal evenly over the 8 XCDs, or are many: 4 pairs would leave half the chip without work)
    if (a.mode == VH_MASK_FULL && (a.B * a.n_heads == n_bh8 || a.B * a.n_heads >= 64)) {
        bh = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / a.n_qb);
        qb = (int)((blockIdx.x >> 3) % a.n_qb);
    } else {
        bh = (int)(blockIdx.x % n_bh8);
        qb = a.n_qb - 1 - (int)(blockIdx.x / n_bh8);
    }
    if (bh >= a.B * a.n_heads) return;                       // (the whole workgroup)
    const int b = bh / a.n_heads, head = bh - b * a.n_heads;
    const int q_off = a.Tk - a.Tq;
    const int kvl = a.kv_len ? min(a.kv_len[b], a.Tk) : a.Tk;
    const int xl = a.mode == VH_MASK_PREFIX ? (a.x_len_dev ? a.x_len_dev[b] : a.x_len) : 0;
    const bool prefix = a.mode == VH_MASK_PREFIX;

    const int q0 = qb * A16_QB + 32 * w;                     // this wave's first query
    const int qi = min(q0 + r, a.Tq - 1);                    // this lane's query (clamped: rows beyond Tq are never stored)
    const int qpos = q_off + qi;
    // this lane's query sees exactly the keys [0, klim): key < kv_len and (key < x_len or (qpos >= x_len and key <= qpos))
    const int klim = prefix ? min(kvl, qpos >= xl ? qpos + 1 : xl) : kvl;
    // keys any query of the WORKGROUP can see: [0, kend)
    const int wg_qlast = q_off + min(qb * A16_QB + A16_QB, a.Tq) - 1;
    int kend = kvl;
    if (prefix) kend = min(kvl, wg_qlast >= xl ? max(xl, wg_qlast + 1) : xl);
    const int n_tiles = (kend + A16_KT - 1) / A16_KT;
    // ... and of this WAVE (wave-uniform): tiles from wave_kend on are skipped by the wave (it still stages and syncs)
    const int wv_qfirst = q_off + min(q0, a.Tq - 1), wv_qlast = q_off + min(q0 + 31, a.Tq - 1);
    int wave_kend = kvl;
    if (prefix) wave_kend = min(kvl, wv_qlast >= xl ? max(xl, wv_qlast + 1) : xl);

    // staging by LDS-DMA (global_load_lds_dwordx4: one 1-KiB piece = 8 key rows x 128 B per wave-instruction, no staging
    // registers, no ds_write): a tile is 8 pieces of K + 8 of V; waves 0 / 1 bring the K rows 0..31 / 32..63, waves 2 / 3
    // the V rows.  Lane L fills slot (row L >> 3 of the piece, 16-byte chunk L & 7) of the lane-linear image, so it FETCHES
    // chunk (L & 7) ^ swz(row): the swizzle lives in the source address.  Keys beyond Tk re-read the last written row
    // (a masked weight is 0, but 0 x the NaN an unwritten cache row may hold is NaN).  Three images in a ring: tile t + 2 is
    // requested at the top of tile t (two tiles of flight time); the barrier at the end of tile t orders tile t + 1.
    const bool stage_v = w >= 2;
    const char* sbase = (const char*)((stage_v ? a.vc : a.kc) + ((int64_t)b * a.n_heads + head) * a.S_max * VH_HEAD_DIM);
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds + (stage_v ? A16_KT * 128 : 0) + (w & 1) * 4096;
    int srow[4];
    uint32_t schunk[4], soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        srow[i] = 32 * (w & 1) + 8 * i + (lane >> 3);
        const int swz = stage_v ? ((srow[i] >> 1) & 1) << 2 : (srow[i] >> 1) & 7;
        schunk[i] = (uint32_t)(((lane & 7) ^ swz) << 4);
        soff[i] = (uint32_t)srow[i] * 128u + schunk[i];      // byte offset inside a tile that lies wholly below Tk
    }
    auto dma_tile = [&](int k0, int buf) {
        const bool whole = k0 + A16_KT <= a.Tk;               // (wave-uniform) no row of the tile needs the clamp
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t voff = whole ? soff[i] + (uint32_t)k0 * 128u : (uint32_t)min(k0 + srow[i], a.Tk - 1) * 128u + schunk[i];
            const uint32_t dst = lds0 + buf * (2 * A16_KT * 128) + i * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sbase) : "memory");
        }
    };
    if (n_tiles > 0) dma_tile(0, 0);
    if (n_tiles > 1) dma_tile(A16_KT, 1);

    // Q^T fragments (B operand of S^T = K Q^T): lane (r, h) holds Q'[query r][d = 16 s + 8 h + j].  Q' = c Q with
    // c = 1 / sqrt(64) * log2(e) comes PRE-SCALED from the producer (vh_linear_qkv_bf16 scales in fp32 before the one
    // narrowing: no second rounding), so the score accumulators are base-2 exponents.
    bf16x8 qf[4];
    {
        const uint16_t* qr = a.q + ((int64_t)b * a.Tq + qi) * a.ldq + head * VH_HEAD_DIM + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(bf16x8, ldq(qr + 16 * s));
    }
    // Softmax without a per-element subtraction or sum (the kernel is bound by vector ISSUE slots, ~200 per 32 x 64 scores
    // before this; profiles/r6_pmc_attn16.md): the reference point m_ref of a row moves only when a tile's maximum exceeds it
    // by more than A16_LAZY (weights then reach at most 2^A16_LAZY: exact in the 16-bit formats, fp32 sums), and -m_ref sits
    // in sixteen registers that are the INITIAL ACCUMULATOR of the score MFMAs — the chain delivers c S - m_ref, the weight
    // is one v_exp_f32 of that.  The row sum comes out of the matrix pipe as well: a third accumulator block whose A operand
    // is 1 on rows 0 and 4 (lacc[0] of every lane = its query's sum over both key halves, taken from the ROUNDED weights
    // the products see).  Moving m_ref rescales O, the sums and the pending scores by 2^-delta, all of them (tile 0 always moves).
    f32x16 oacc[2], lacc, minit;
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; lacc[e] = 0.f; minit[e] = 0.f; }
    float m_ref = 0.f;
    bf16x8 ones;
    {
        const uint32_t one2 = vh_pack_h16(1.f, 1.f), w1 = (r == 0 || r == 4) ? one2 : 0u;
        ones = __builtin_bit_cast(bf16x8, u32x4{w1, w1, w1, w1});
    }

    // K fragment addresses (A operand of S^T): key row 32 u + r, chunk (2 s + h) ^ ((r >> 1) & 7)
    const int kswz = (r >> 1) & 7;
    // V^T fragment addresses: group g = lane >> 4 (h = g >> 1), lane 4 q4 + p of the group supplies key row kb + q4,
    // columns d = 32 db + 16 (g & 1) + 4 p .. + 3 -> chunk 4 db + 2 (g & 1) + (p >> 1), byte 8 (p & 1) in it
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;

    // vmcnt(0), unconditionally and BEFORE the loop: the first two tiles and the Q fragments have landed.  (The Q fragments
    // are first used inside the loop; left to the compiler's wait insertion, vmcnt(3..0) lands in front of the first four
    // score MFMAs of EVERY tile — a wait for the tile's own prefetches.  Seen in the ISA in round 6.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    int buf = 0;
    for (int it = 0; it < n_tiles; ++it) {
        const int k0 = it * A16_KT;
        const bool ahead = it + 2 < n_tiles;                 // (wave-uniform)
        if (ahead) dma_tile(k0 + 2 * A16_KT, buf >= 1 ? buf - 1 : 2);      // into image (it + 2) % 3, free since the last barrier
        if (k0 < wave_kend) {
            const char* kl = lds + buf * 2 * A16_KT * 128;
            const char* vl = kl + A16_KT * 128;
            // ---- c S^T - m_ref = K (c Q)^T + (-m_ref) for the two 32-key sub-tiles
            f32x16 sacc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 kf = __builtin_bit_cast(bf16x8, ldq(kl + (32 * u + r) * 128 + (((2 * s + h) ^ kswz) << 4)));
                    sacc[u] = VH_MFMA16(kf, qf[s], s ? sacc[u] : minit);
                }
            // ---- mask (only on tiles that need it: wave-uniform test): register x of sub-tile u holds key
            // k0 + 32 u + (x & 3) + 8 (x >> 2) + 4 h, visible iff it is below the lane's bound — one compare + select each
            const bool need_mask = k0 + A16_KT > kvl || (prefix && k0 + A16_KT > xl && (wv_qfirst < xl || k0 + A16_KT - 1 > wv_qfirst));
            if (need_mask) {
                const int t = klim - k0 - 4 * h;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int x = 0; x < 16; ++x) sacc[u][x] = 32 * u + (x & 3) + 8 * (x >> 2) < t ? sacc[u][x] : A16_NEG;
            }
            // ---- the tile's row maximum relative to m_ref (the query's other key half sits in lane ^ 32)
            float mx = A16_NEG;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int x = 0; x < 16; ++x) mx = fmaxf(mx, sacc[u][x]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, mx), __builtin_bit_cast(uint32_t, mx), false, false);
                mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
            }
            if (it == 0 || __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(mx > A16_LAZY) != 0)) {
                // move the reference point: to the row's maximum on the first tile, up to it (never down) later
                // (a row without any visible key in tile 0 — none under the analytic masks while kv_len >= 1 — stays at 0)
                const float delta = it == 0 ? (mx > 0.5f * A16_NEG ? mx : 0.f) : fmaxf(mx, 0.f);
                m_ref += delta;
                const float alpha = vh_exp2(-delta);
#pragma unroll
                for (int e = 0; e < 16; ++e) { minit[e] = -m_ref; sacc[0][e] -= delta; sacc[1][e] -= delta; }
                if (it) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
                    lacc[0] *= alpha;
                }
            }
            // ---- P^T = 2^(c S - m_ref) narrowed into the B operands (keys of a k-step in accumulator order)
            bf16x8 pf[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = vh_pack_h16_unit(vh_exp2(sacc[u][8 * s + 2 * j]), vh_exp2(sacc[u][8 * s + 2 * j + 1]));
                    pf[u][s] = __builtin_bit_cast(bf16x8, pk);
                }
            // ---- O^T += V^T P^T: A operand element j of lane half h = V[key 32 u + 16 s + 8 (j >> 2) + 4 h + (j & 3)][d = 32 db + r];
            // and the row sums, l += 1^T P^T
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        u32x2 lo, hi;
                        {
                            const int key = 32 * u + 16 * s + 4 * (g >> 1) + q4;
                            const int c = 4 * db + 2 * (g & 1) + (p >> 1);
                            const char* ad = vl + key * 128 + ((c ^ (((key >> 1) & 1) << 2)) << 4) + 8 * (p & 1);
                            lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                               (s16x4 __attribute__((address_space(3)))*)ad));
                            const int key2 = key + 8;
                            const char* ad2 = vl + key2 * 128 + ((c ^ (((key2 >> 1) & 1) << 2)) << 4) + 8 * (p & 1);
                            hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                               (s16x4 __attribute__((address_space(3)))*)ad2));
                        }
                        const bf16x8 vf = __builtin_bit_cast(bf16x8, u32x4{lo.x, lo.y, hi.x, hi.y});
                        oacc[db] = VH_MFMA16(vf, pf[u][s], oacc[db]);
                    }
                    lacc = VH_MFMA16(ones, pf[u][s], lacc);
                }
        }
        // tile it + 1 (this wave's pieces of it: the four requests BEFORE the ones issued above) has landed; after the barrier
        // every wave's pieces have, and nobody reads image `buf` any more
        if (ahead) __builtin_amdgcn_s_waitcnt(0x0F74);       // vmcnt(4)
        else __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
        __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
    }
    // ---- epilogue: O / l, narrowed, transposed through LDS (per wave: 32 queries x 64 d) and stored as whole 128-B rows
    const float l_tot = lacc[0];
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    char* ow = lds + w * 32 * 144;                           // 144-B rows (128 + 16 of padding)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const u32x2 pk = {pack_bf16(oacc[db][4 * g4] * inv, oacc[db][4 * g4 + 1] * inv),
                              pack_bf16(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv)};
            *reinterpret_cast<u32x2*>(ow + r * 144 + (32 * db + 8 * g4 + 4 * h) * 2) = pk;
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): the wave's own writes are in LDS (wave-private region)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i, row = id >> 3, c = id & 7;
        const int qrow = q0 + row;
        if (qrow < a.Tq && qrow < qb * A16_QB + A16_QB)
            stq(a.out + ((int64_t)b * a.Tq + qrow) * a.ldo + head * VH_HEAD_DIM + 8 * c, ldq(ow + row * 144 + 16 * c));
    }
}

extern "C" int vh_attn_rows_bf16(const uint16_t* q, int ldq_, const uint16_t* kcache16, const uint16_t* vcache16, uint16_t* out,
                                 int ldo, int B, int n_heads, int Tq, int Tk, int S_max, int mode, int x_len,
                                 const int32_t* x_len_dev, const int32_t* kv_len, void* stream) {
    VH_REQUIRE(q && kcache16 && vcache16 && out, VH_EINVAL, "vh_attn_rows_bf16: null pointer");
    VH_REQUIRE(B > 0 && n_heads > 0 && Tq > 0 && Tk >= Tq && Tk <= S_max, VH_EINVAL,
               "vh_attn_rows_bf16: bad dims B=%d h=%d Tq=%d Tk=%d S_max=%d", B, n_heads, Tq, Tk, S_max);
    VH_REQUIRE(mode == VH_MASK_FULL || mode == VH_MASK_PREFIX, VH_EUNSUPPORTED,
               "vh_attn_rows_bf16: mode=%d (analytic masks only: FULL / PREFIX)", mode);
    VH_REQUIRE(ldq_ % 8 == 0 && ldo % 8 == 0 && ldq_ >= n_heads * VH_HEAD_DIM && ldo >= n_heads * VH_HEAD_DIM, VH_EINVAL,
               "vh_attn_rows_bf16: ldq=%d ldo=%d", ldq_, ldo);
    VH_REQUIRE(vh_aligned16(q) && vh_aligned16(kcache16) && vh_aligned16(vcache16) && vh_aligned16(out), VH_EALIGN,
               "vh_attn_rows_bf16: pointers must be 16-byte aligned");
    Attn16Args a{q, ldq_, kcache16, vcache16, out, ldo, B, n_heads, Tq, Tk, S_max, mode, x_len, x_len_dev, kv_len,
                 (Tq + A16_QB - 1) / A16_QB};
    const int64_t grid = (int64_t)a.n_qb * 8 * (((int64_t)B * n_heads + 7) / 8);     // (see the kernel: 8 XCDs x pairs x query blocks)
    VH_REQUIRE(grid < (1ll << 31), VH_EUNSUPPORTED, "vh_attn_rows_bf16: grid too large");
    hipLaunchKernelGGL(attn16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    VH_CHECK_LAUNCH("vh_attn_rows_bf16");
    return VH_OK;
}

// =============================================================================================
// composite: the full-sequence forward of vh_transformer_forward with bf16 operands
// =============================================================================================
#define TRY16(call) do { int rc_ = (call); if (rc_ != VH_OK) return rc_; } while (0)

extern "C" int vh_transformer_forward_bf16(const vh_forward16_desc* f, void* stream) {
    VH_REQUIRE(f && f->layers && f->layers16 && f->x && f->xn16 && f->q16 && f->attn16 && f->hidden16, VH_EINVAL,
               "vh_transformer_forward_bf16: null pointer in desc");
    VH_REQUIRE(f->B > 0 && f->T > 0 && f->n_layers > 0 && f->S_max >= f->T, VH_EINVAL,
               "vh_transformer_forward_bf16: bad dims B=%d T=%d L=%d S_max=%d", f->B, f->T, f->n_layers, f->S_max);
    VH_REQUIRE(f->d_model == f->n_heads * VH_HEAD_DIM && f->d_model % 128 == 0 && f->dff % 128 == 0, VH_EUNSUPPORTED,
               "vh_transformer_forward_bf16: d_model=%d dff=%d (n_heads x 64, multiples of 128)", f->d_model, f->dff);
    VH_REQUIRE(f->mode == VH_MASK_FULL || f->mode == VH_MASK_PREFIX, VH_EUNSUPPORTED,
               "vh_transformer_forward_bf16: analytic masks only (mode=%d)", f->mode);
    const int B = f->B, T = f->T, D = f->d_model, M = B * T;
    for (int i = 0; i < f->n_layers; ++i) {
        const vh_layer& L = f->layers[i];
        const vh_layer16& H = f->layers16[i];
        VH_REQUIRE(H.wqkv && H.wo && H.w1 && H.w2 && H.kcache16 && H.vcache16, VH_EINVAL,
                   "vh_transformer_forward_bf16: layer %d has a null bf16 pointer", i);
        const float* ada = f->ada ? f->ada + (int64_t)i * 4 * D : nullptr;
        const float* src = (i == 0 && f->x_in) ? f->x_in : f->x;
        TRY16(vh_layernorm_bf16(src, L.ln1_g, L.ln1_b, ada, ada ? ada + D : nullptr, f->xn16, M, D, f->ln_eps, stream));
        TRY16(vh_linear_qkv_bf16(f->xn16, D, H.wqkv, f->q16, D, H.kcache16, H.vcache16, nullptr, B, T, D, f->n_heads,
                                 f->S_max, stream));
        TRY16(vh_attn_rows_bf16(f->q16, D, H.kcache16, H.vcache16, f->attn16, D, B, f->n_heads, T, T, f->S_max, f->mode,
                                f->x_len, f->x_len_dev, f->kv_len, stream));
        TRY16(vh_linear_bf16(f->attn16, D, H.wo, L.bo, src, D, f->x, D, 0, M, D, D, VH_ACT_NONE, stream));
        TRY16(vh_layernorm_bf16(f->x, L.ln2_g, L.ln2_b, ada ? ada + 2 * D : nullptr, ada ? ada + 3 * D : nullptr, f->xn16, M, D,
                                f->ln_eps, stream));
        TRY16(vh_linear_bf16(f->xn16, D, H.w1, L.b1, nullptr, 0, f->hidden16, f->dff, 1, M, f->dff, D, VH_ACT_GELU_ERF, stream));
        TRY16(vh_linear_bf16(f->hidden16, f->dff, H.w2, L.b2, f->x, D, f->x, D, 0, M, D, f->dff, VH_ACT_NONE, stream));
    }
    return VH_OK;
}
