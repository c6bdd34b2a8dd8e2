// Backward / training-only kernels of the path (valle/models/valle_ar.py:43-90 + loss.backward()):
// LayerNorm backward (with the adaptive scale/shift), exact-erf GELU forward/backward on a saved
// pre-activation, row softmax forward/backward for the recomputed attention probabilities,
// fused cross-entropy forward+backward, embedding scatter-add, column sums for bias gradients.
// All HBM-bound row kernels: one wave64 per row, float4 per lane, DPP reductions.  The backward's
// matrix products are this library's own MFMA kernels: dX = dY·W and dW = dYᵀ·X in gemm.hip
// (vh_linear_ex / vh_gemm_tn), the five attention products in attention.hip (vh_attn_rows_bwd_ws).
// Column-sum style gradients use fp32 atomics (order not fixed: last-bit run-to-run variation).
#include "vh_common.h"

#define NEG_INF (-INFINITY)

__device__ __forceinline__ float hsum4(f32x4 v) { return (v.x + v.y) + (v.z + v.w); }

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  Forward: xhat=(x-mu)*rstd; u=gamma*xhat+beta; y = s*u + t (s,t optional).
//   du = dy*s;  dgamma += sum_rows du*xhat;  dbeta += sum_rows du;  ds += sum_rows dy*u;  dt += sum_rows dy
//   g = du*gamma;  dx = rstd*(g - mean(g) - xhat*mean(g*xhat))
// Waves stride over rows and keep their column partial sums in registers; one atomicAdd per
// column per WORKGROUP at the end.
// ---------------------------------------------------------------------------------------------
#define LNB_WAVES 8
// DROP: the dropped copy of dx (the gradient of the branch under dropout1 / dropout2, modules.py:277-278) leaves in the
// same pass and the column sums are taken of IT; a template parameter so that the dropout-off kernel keeps its registers.
template <int NV, bool DROP>
__global__ __launch_bounds__(LNB_WAVES * 64) void layernorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ s, const float* __restrict__ dy, float* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ds,
    float* __restrict__ dt, const float* __restrict__ dres, float* __restrict__ dcol, int rows, int d, float eps,
    float* __restrict__ dx_drop, DropArgs drop) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * LNB_WAVES + (threadIdx.x >> 6), nw = gridDim.x * LNB_WAVES;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 gm[NV], bt[NV], sc[NV], ag[NV], ab[NV], as[NV], at[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        const bool in = c < d;
        gm[i] = in ? ld4(gamma + c) : z;
        bt[i] = (in && s) ? ld4(beta + c) : z;
        sc[i] = (in && s) ? ld4(s + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        ag[i] = ab[i] = as[i] = at[i] = z;
    }
    f32x4 ac[NV];                                   // column sums of the rows written to dx (dcol)
#pragma unroll
    for (int i = 0; i < NV; ++i) ac[i] = z;
    // A wave walks its rows one after the other, and a row is a chain of four wave reductions: the next row's three
    // operand rows (x, dy, the residual branch's gradient) are requested BEFORE this row is reduced, so the memory
    // round trip hides behind the chain instead of being paid per row (45.7 -> see profiles/r3_train_kernel_stats.md).
    f32x4 vn[NV], gn[NV], rn[NV];
    auto load_row = [&](int row) {
        const int64_t o = (int64_t)row * d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            vn[i] = c < d ? ld4(x + o + c) : z;
            gn[i] = c < d ? ld4(dy + o + c) : z;
            rn[i] = (dres && c < d) ? ld4(dres + o + c) : z;
        }
    };
    // (d_model <= 512 only: at NV >= 4 the second set of three rows does not fit beside the nine per-column accumulators
    // — the allocator spilled 16..700 registers to scratch — so the wide shapes fetch each row when they get to it)
    constexpr bool AHEAD = NV <= 2;
    if (AHEAD && wid < rows) load_row(wid);
    for (int row = wid; row < rows; row += nw) {
        f32x4 v[NV], g[NV], r[NV];
        float sum = 0.f;
        if (!AHEAD) load_row(row);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = vn[i]; g[i] = gn[i]; r[i] = rn[i];
            sum += hsum4(v[i]);
        }
        if (AHEAD && row + nw < rows) load_row(row + nw);
        const float mu = wave_sum(sum) / (float)d;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if ((lane + 64 * i) * 4 < d) {
                v[i] = v[i] - mu;
                ss += hsum4(v[i] * v[i]);
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 xhat = v[i] * rstd;
            const f32x4 dyv = g[i];
            const f32x4 du = dyv * sc[i];
            if (s) {
                as[i] += dyv * (gm[i] * xhat + bt[i]);
                at[i] += dyv;
            }
            ag[i] += du * xhat;
            ab[i] += du;
            g[i] = du * gm[i];
            v[i] = xhat;
            m1 += hsum4(g[i]);
            m2 += hsum4(g[i] * xhat);
        }
        m1 = wave_sum(m1) / (float)d;
        m2 = wave_sum(m2) / (float)d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                // the pre-norm block's residual branch: the gradient that bypasses the norm is added here instead of
                // by a separate elementwise launch (x feeds the norm AND the residual add, modules.py:271-279)
                const f32x4 o = (g[i] - m1 - v[i] * m2) * rstd + r[i];
                st4(dx + (int64_t)row * d + c, o);
                if (DROP) {
                    const f32x4 od = o * vh_dropmul4(drop, (uint32_t)row, (uint32_t)(lane + 64 * i));
                    st4(dx_drop + (int64_t)row * d + c, od);
                    if (dcol) ac[i] += od;
                } else if (dcol) {
                    ac[i] += o;
                }
            }
        }
    }
    // Column sums: the waves of the workgroup meet in LDS first, so each column receives ONE atomic
    // per workgroup and quantity.  (Per-wave atomics — 2M adds onto the same 1024 addresses at 16k rows —
    // ran at the contended-atomic rate of the chip and took 250 us per launch, 8x the row work.)
    __shared__ __attribute__((aligned(16))) float colbuf[LNB_WAVES][NV * 256];
    const int w = threadIdx.x >> 6;
    auto reduce_to = [&](f32x4 (&part)[NV], float* __restrict__ dst) {
#pragma unroll
        for (int i = 0; i < NV; ++i) st4(&colbuf[w][(lane + 64 * i) * 4], part[i]);
        __syncthreads();
        for (int c = threadIdx.x; c < d; c += LNB_WAVES * 64) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < LNB_WAVES; ++k) t += colbuf[k][c];
            atomicAdd(dst + c, t);
        }
        __syncthreads();
    };
    reduce_to(ag, dgamma);
    reduce_to(ab, dbeta);
    if (s) {
        reduce_to(as, ds);
        reduce_to(at, dt);
    }
    if (dcol) reduce_to(ac, dcol);
}

extern "C" int vh_layernorm_bwd(const float* x, const float* gamma, const float* beta,
                                const float* ada_scale, const float* dy, float* dx, float* dgamma,
                                float* dbeta, float* dscale, float* dshift, const float* dres, float* dcolsum,
                                float* dx_drop, const vh_dropout_spec* drop, int rows, int d, float eps, void* stream) {
    VH_REQUIRE(x && gamma && dy && dx && dgamma && dbeta, VH_EINVAL, "vh_layernorm_bwd: null pointer");
    VH_REQUIRE(!ada_scale || (beta && dscale && dshift), VH_EINVAL,
               "vh_layernorm_bwd: adaptive form needs beta, dscale, dshift");
    VH_REQUIRE(rows >= 0 && d > 0 && d % 4 == 0 && d <= 2048, VH_EINVAL,
               "vh_layernorm_bwd: bad dims rows=%d d=%d (d multiple of 4, <= 2048)", rows, d);
    VH_REQUIRE(vh_aligned16(x) && vh_aligned16(dy) && vh_aligned16(dx) && vh_aligned16(gamma) &&
                   vh_aligned16(beta) && vh_aligned16(ada_scale) && vh_aligned16(dres),
               VH_EALIGN, "vh_layernorm_bwd: pointers must be 16-byte aligned");
    VH_REQUIRE(VH_DROP_OK(drop) && vh_aligned16(dx_drop), VH_EINVAL, "vh_layernorm_bwd: dropout p in [0, 1), dx_drop aligned");
    DropArgs da;
    const bool dropped = vh_drop_args(drop, &da);
    VH_REQUIRE(dropped == (dx_drop != nullptr), VH_EINVAL, "vh_layernorm_bwd: dx_drop and a dropout spec with p > 0 go together");
    if (rows == 0) return VH_OK;
    const int blocks = rows < LNB_WAVES * 256 ? (rows + LNB_WAVES - 1) / LNB_WAVES : 256;
    hipStream_t st = (hipStream_t)stream;
#define LNB(NV)                                                                                                        \
    do {                                                                                                               \
        if (dropped)                                                                                                   \
            hipLaunchKernelGGL((layernorm_bwd_kernel<NV, true>), dim3(blocks), dim3(LNB_WAVES * 64), 0, st, x, gamma,  \
                               beta, ada_scale, dy, dx, dgamma, dbeta, dscale, dshift, dres, dcolsum, rows, d, eps,    \
                               dx_drop, da);                                                                           \
        else                                                                                                           \
            hipLaunchKernelGGL((layernorm_bwd_kernel<NV, false>), dim3(blocks), dim3(LNB_WAVES * 64), 0, st, x, gamma, \
                               beta, ada_scale, dy, dx, dgamma, dbeta, dscale, dshift, dres, dcolsum, rows, d, eps,    \
                               dx_drop, da);                                                                           \
    } while (0)
    if (d <= 256) LNB(1);
    else if (d <= 512) LNB(2);
    else if (d <= 1024) LNB(4);
    else LNB(8);
#undef LNB
    VH_CHECK_LAUNCH("vh_layernorm_bwd");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// GELU (exact erf) forward on a saved pre-activation, and its backward.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ pre,
                                                   const float* __restrict__ dh,
                                                   float* __restrict__ out, int64_t n4) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 x = ld4(pre + 4 * i);
        f32x4 r;
        if (dh) {
            const f32x4 g = ld4(dh + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float cdf = 0.5f * (1.0f + vh_erf(x[j] * 0.70710678118654752440f));
                const float pdf = 0.39894228040143267794f * expf(-0.5f * x[j] * x[j]);
                r[j] = g[j] * (cdf + x[j] * pdf);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = gelu_erf(x[j]);
        }
        st4(out + 4 * i, r);
    }
}

extern "C" int vh_gelu(const float* pre, const float* dh, float* out, int64_t n, void* stream) {
    VH_REQUIRE(pre && out && n >= 0 && n % 4 == 0, VH_EINVAL, "vh_gelu: n=%lld must be a multiple of 4",
               (long long)n);
    VH_REQUIRE(vh_aligned16(pre) && vh_aligned16(dh) && vh_aligned16(out), VH_EALIGN, "vh_gelu: alignment");
    if (n == 0) return VH_OK;
    const int64_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(gelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pre, dh, out, n4);
    VH_CHECK_LAUNCH("vh_gelu");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// Row softmax over materialised scores (training backward recomputes P = softmax(S*scale + mask)),
// in place, with the same analytic masks as vh_attn_rows; and its backward
// dS = scale * P o (dP - rowsum(dP o P)), in place on dP.   rows = B*h*Tq, one wave per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(
    float* __restrict__ S, int n_heads, int Tq, int Tk, float scale, int mode, int x_len,
    const int32_t* __restrict__ x_len_dev, const int32_t* __restrict__ kv_len,
    const uint8_t* __restrict__ mask, const uint8_t* __restrict__ pad, int64_t rows, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % Tq);
    const int b = (int)(row / ((int64_t)Tq * n_heads));
    const int kvl = kv_len ? min(kv_len[b], Tk) : Tk;
    const int xl = x_len_dev ? x_len_dev[b] : x_len;
    const int qpos = Tk - Tq + i;
    float* sr = S + row * ld;
    auto visible = [&](int j) {
        if (mode == VH_MASK_EXPLICIT) return !mask[(int64_t)i * Tk + j] && !(pad && pad[(int64_t)b * Tk + j]);
        bool v = j < kvl;
        if (mode == VH_MASK_PREFIX) v = v && (j < xl || (qpos >= xl && j <= qpos));
        return v;
    };
    float m = NEG_INF;
    for (int j = lane; j < Tk; j += 64) {
        const float v = visible(j) ? sr[j] * scale : NEG_INF;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < Tk; j += 64) {
        const float v = visible(j) ? expf(sr[j] * scale - m) : 0.f;
        sr[j] = v;
        l += v;
    }
    const float inv = 1.0f / wave_sum(l);
    for (int j = lane; j < Tk; j += 64) sr[j] *= inv;
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P,
                                                          float* __restrict__ dP, int Tk, float scale,
                                                          int64_t rows, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = P + row * ld;
    float* dr = dP + row * ld;
    float acc = 0.f;
    for (int j = lane; j < Tk; j += 64) acc += pr[j] * dr[j];
    acc = wave_sum(acc);
    for (int j = lane; j < Tk; j += 64) dr[j] = scale * pr[j] * (dr[j] - acc);
}

extern "C" int vh_softmax_rows(float* S, int ld, int B, int n_heads, int Tq, int Tk, float scale, int mode,
                               int x_len, const int32_t* x_len_dev, const int32_t* kv_len,
                               const uint8_t* mask, const uint8_t* pad, void* stream) {
    VH_REQUIRE(S && B >= 0 && n_heads > 0 && Tq >= 0 && Tk >= Tq && ld >= Tk, VH_EINVAL,
               "vh_softmax_rows: bad args");
    VH_REQUIRE(mode == VH_MASK_FULL || mode == VH_MASK_PREFIX || (mode == VH_MASK_EXPLICIT && mask),
               VH_EINVAL, "vh_softmax_rows: mode=%d", mode);
    const int64_t rows = (int64_t)B * n_heads * Tq;
    if (rows == 0) return VH_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, S, n_heads, Tq, Tk, scale, mode, x_len, x_len_dev, kv_len,
                       mask, pad, rows, ld);
    VH_CHECK_LAUNCH("vh_softmax_rows");
    return VH_OK;
}

extern "C" int vh_softmax_bwd(const float* P, float* dP, int ld, int64_t rows, int Tk, float scale,
                              void* stream) {
    VH_REQUIRE(P && dP && rows >= 0 && Tk > 0 && ld >= Tk, VH_EINVAL, "vh_softmax_bwd: bad args");
    if (rows == 0) return VH_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, P, dP, Tk, scale, rows, ld);
    VH_CHECK_LAUNCH("vh_softmax_bwd");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// Cross entropy (F.cross_entropy with mean reduction over ALL rows, valle_ar.py:86), fused with its
// gradient: loss += (lse - logit[target]) / R ;  dlogits = (softmax - onehot) * grad_scale.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_entropy_kernel(
    const float* __restrict__ logits, int ld, int V, const int64_t* __restrict__ target,
    float* __restrict__ loss, float* __restrict__ dlogits, int ldd, float inv_rows, int rows,
    int32_t* __restrict__ err_flag) {
    __shared__ float s_part[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float part = 0.f;                               // this wave's share of the mean loss
    bool bad = false;
    // a wave walks rows; the loss leaves the workgroup as ONE atomic (one per row — 14 400 adds onto one address —
    // ran at the contended-atomic rate of the memory side and was most of the launch: 195 us for 118 MB of traffic)
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const float* lr = logits + (int64_t)row * ld;
        float m = NEG_INF;
        for (int j = lane; j < V; j += 64) m = fmaxf(m, lr[j]);
        m = wave_max(m);
        float l = 0.f;
        for (int j = lane; j < V; j += 64) l += expf(lr[j] - m);
        l = wave_sum(l);
        const float lse = m + logf(l);
        const int64_t t64 = target[row];
        const bool ok = t64 >= 0 && t64 < V;        // F.cross_entropy raises for a class outside [0, V)
        const int tgt = ok ? (int)t64 : -1;
        part += (lse - (ok ? lr[tgt] : 0.f)) * inv_rows;
        bad |= !ok;
        if (dlogits) {
            float* dr = dlogits + (int64_t)row * ldd;
            for (int j = lane; j < V; j += 64) dr[j] = (expf(lr[j] - lse) - (j == tgt ? 1.f : 0.f)) * inv_rows;
        }
    }
    if (lane == 0) {
        s_part[w] = part;
        if (bad && err_flag) atomicOr(err_flag, VH_DEVERR_TARGET);
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

extern "C" int vh_cross_entropy(const float* logits, int ld, int V, const int64_t* target, float* loss,
                                float* dlogits, int ldd, int rows, int32_t* err_flag, void* stream) {
    VH_REQUIRE(logits && target && loss && rows > 0 && V > 0 && ld >= V && (!dlogits || ldd >= V),
               VH_EINVAL, "vh_cross_entropy: bad args rows=%d V=%d ld=%d", rows, V, ld);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) {
        vh_set_error("vh_cross_entropy: hipMemsetAsync failed");
        return VH_ELAUNCH;
    }
    const int ce_blocks = (rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(ce_blocks), dim3(256), 0, st, logits, ld, V, target,
                       loss, dlogits, ldd, 1.0f / (float)rows, rows, err_flag);
    VH_CHECK_LAUNCH("vh_cross_entropy");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// Embedding backward: dtable[ids[b,t]] += dout[b, t0+t, :]   (scatter-add, fp32 atomics)
// ---------------------------------------------------------------------------------------------
// A workgroup owns EMB_RUN consecutive positions of one batch row and a thread owns columns: it walks the positions and
// adds up consecutive ones that hit the SAME table row before issuing the atomic.  A padded batch ends every row in a
// run of one id (a third of configs[3]'s positions): per-position atomics put thousands of adds on that one table row
// and ran at the contended-atomic rate (157 us per AR step for 8.4 M adds); runs collapse 16:1 here.
#define EMB_RUN 16
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, int64_t ids_bs,
                                                        int64_t ids_ts, const float* __restrict__ dout,
                                                        int64_t dout_bs, int t0, float* __restrict__ dtable,
                                                        int vocab, int T, int d,
                                                        int32_t* __restrict__ err_flag, DropArgs drop) {
    __shared__ int sid[EMB_RUN];                                     // the run's table rows, -1 = skip (bad id / beyond T)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int t_begin = blockIdx.x * EMB_RUN;
    if (tid < EMB_RUN) {
        const int t = t_begin + tid;
        int64_t id = -1;
        if (t < T) {
            id = ids[b * ids_bs + t * ids_ts];
            if (id < 0 || id >= vocab) {                             // never scatter outside the table
                if (err_flag) atomicOr(err_flag, VH_DEVERR_EMBED_ID);
                id = -1;
            }
        }
        sid[tid] = (int)id;
    }
    __syncthreads();
    const float* src = dout + b * dout_bs + (int64_t)(t0 + t_begin) * d;
    const uint32_t frow0 = (uint32_t)((b * dout_bs) / d + t0 + t_begin);     // the forward's dropout field: row = row of dout
    for (int c = tid; c < d; c += 256) {
        float v[EMB_RUN];
#pragma unroll
        for (int i = 0; i < EMB_RUN; ++i) v[i] = sid[i] >= 0 ? src[(int64_t)i * d + c] : 0.f;   // all in flight at once
        if (drop.thresh) {                               // (a thread owns a column: one of the four words of a Philox call)
#pragma unroll
            for (int i = 0; i < EMB_RUN; ++i) v[i] *= vh_dropmul4(drop, frow0 + i, (uint32_t)c >> 2)[c & 3];
        }
        int cur = -1;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < EMB_RUN; ++i) {
            const int id = sid[i];
            if (id != cur) {
                if (cur >= 0) atomicAdd(dtable + (int64_t)cur * d + c, acc);
                cur = id;
                acc = v[i];
            } else {
                acc += v[i];
            }
        }
        if (cur >= 0) atomicAdd(dtable + (int64_t)cur * d + c, acc);
    }
}

extern "C" int vh_embed_bwd(const int64_t* ids, int64_t ids_bstride, int64_t ids_tstride,
                            const float* dout, int64_t dout_bstride, int out_t0, float* dtable, int vocab,
                            int B, int T, int d, int32_t* err_flag, const vh_dropout_spec* drop, void* stream) {
    VH_REQUIRE(ids && dout && dtable && vocab > 0 && B >= 0 && T >= 0 && d > 0, VH_EINVAL,
               "vh_embed_bwd: bad args");
    DropArgs da;
    VH_REQUIRE(VH_DROP_OK(drop), VH_EINVAL, "vh_embed_bwd: dropout p must be in [0, 1)");
    if (vh_drop_args(drop, &da))
        VH_REQUIRE(dout_bstride % d == 0 && d % 4 == 0, VH_EINVAL, "vh_embed_bwd: dropout needs dout_bstride %% d == 0");
    if (B == 0 || T == 0) return VH_OK;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((T + EMB_RUN - 1) / EMB_RUN, B), dim3(256), 0, (hipStream_t)stream, ids,
                       ids_bstride, ids_tstride, dout, dout_bstride, out_t0, dtable, vocab, T, d, err_flag, da);
    VH_CHECK_LAUNCH("vh_embed_bwd");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// Column sums (bias gradients): out[c] += sum_r x[r, c].   Thread per column, row slabs per block.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld,
                                                     float* __restrict__ out, int rows, int cols,
                                                     int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += x[(int64_t)r * ld + c];
    atomicAdd(out + c, acc);
}

extern "C" int vh_colsum(const float* x, int ld, float* out, int rows, int cols, void* stream) {
    VH_REQUIRE(x && out && rows >= 0 && cols > 0 && ld >= cols, VH_EINVAL, "vh_colsum: bad args");
    if (rows == 0) return VH_OK;
    const int rpb = 64;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256, (rows + rpb - 1) / rpb), dim3(256), 0,
                       (hipStream_t)stream, x, ld, out, rows, cols, rpb);
    VH_CHECK_LAUNCH("vh_colsum");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// Optimizer step over ONE flat fp32 buffer (valle/models/valle_ar.py:182-194 AdamW; global-norm clip
// of valle/train_model.py:31-32; the 1/world scale of the gradient all-reduce).  Two launches, no
// host read: (1) per-workgroup partial sums of g^2 in double, (2) every workgroup adds the partials
// in the same fixed order (reproducible), derives the clip coefficient and updates its slice.
// ---------------------------------------------------------------------------------------------
#define VH_SUMSQ_BLOCKS 1024

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ g, int64_t n4,
                                                             double* __restrict__ partials) {
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = ld4(g + 4 * i);
        acc += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         int64_t n4, const double* __restrict__ partials,
                                                         float lr, float beta1, float beta2, float eps,
                                                         float weight_decay, float bc1, float sqrt_bc2,
                                                         float grad_scale, float max_norm, int zero_grad,
                                                         float* __restrict__ norm_out,
                                                         const int32_t* __restrict__ block_slot,
                                                         const int32_t* __restrict__ slot_step,
                                                         const int32_t* __restrict__ guard) {
    // a device-side error flag raised earlier in this step (an id outside its table, ...): parameters and moments stay
    // as they are — the host reads the flag later, without a synchronisation per step, and raises — but the gradient
    // is still cleared when the caller asked for that: the host releases its gradient views after this launch either
    // way, and the next backward's accumulate-type kernels (bias column sums, LayerNorm weight gradients, embedding
    // scatter) rely on a zero buffer
    if (guard && *guard) {
        if (zero_grad)
            for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
                st4(g + 4 * i, f32x4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    // total gradient norm: 1024 partials, 4 per thread, fixed order
    __shared__ double s[4];
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < VH_SUMSQ_BLOCKS / 256; ++i) acc += partials[threadIdx.x * (VH_SUMSQ_BLOCKS / 256) + i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float total_norm = (float)sqrt((s[0] + s[1]) + (s[2] + s[3])) * grad_scale;
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total_norm;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    const float coef = max_norm > 0.f ? fminf(max_norm / (total_norm + 1e-6f), 1.0f) : 1.0f;
    const float gs = grad_scale * coef;
    const float decay = 1.0f - lr * weight_decay;
    float step_size = lr / bc1;
    const float lnb1 = logf(beta1), lnb2 = logf(beta2);      // 1 - beta^st = -expm1(st ln beta): accurate for beta -> 1
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        if (block_slot) {
            // per-parameter step counts (torch.optim.AdamW keeps `step` per parameter and skips a parameter
            // whose grad is None: no decay, no moment update): 64-float blocks map to parameter slots
            const int st = slot_step[block_slot[i >> 4]];
            if (st == 0) continue;                         // this parameter received no gradient this step
            step_size = lr / -expm1f((float)st * lnb1);
            sqrt_bc2 = sqrtf(-expm1f((float)st * lnb2));
        }
        const f32x4 gg = ld4(g + 4 * i) * gs;
        f32x4 pp = ld4(p + 4 * i) * decay;
        f32x4 mm = ld4(m + 4 * i), vv = ld4(v + 4 * i);
        mm = mm + (gg - mm) * (1.0f - beta1);                 // lerp, weight < 0.5
        vv = vv * beta2 + gg * gg * (1.0f - beta2);
#pragma unroll
        for (int j = 0; j < 4; ++j) pp[j] -= step_size * (mm[j] / (sqrtf(vv[j]) / sqrt_bc2 + eps));
        st4(p + 4 * i, pp);
        st4(m + 4 * i, mm);
        st4(v + 4 * i, vv);
        if (zero_grad) st4(g + 4 * i, f32x4{0.f, 0.f, 0.f, 0.f});
    }
}

extern "C" size_t vh_adamw_ws_bytes(void) { return VH_SUMSQ_BLOCKS * sizeof(double); }

extern "C" int vh_adamw_flat(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             float grad_scale, float max_norm, int zero_grad, void* workspace,
                             float* norm_out, const int32_t* block_slot, const int32_t* slot_step,
                             const int32_t* guard, void* stream) {
    VH_REQUIRE((block_slot == nullptr) == (slot_step == nullptr), VH_EINVAL,
               "vh_adamw_flat: block_slot and slot_step go together");
    VH_REQUIRE(!block_slot || n % 64 == 0, VH_EINVAL, "vh_adamw_flat: per-slot steps need n %% 64 == 0");
    VH_REQUIRE(param && grad && exp_avg && exp_avg_sq && workspace, VH_EINVAL, "vh_adamw_flat: null pointer");
    VH_REQUIRE(n >= 0 && n % 4 == 0, VH_EINVAL, "vh_adamw_flat: n=%lld must be a multiple of 4", (long long)n);
    VH_REQUIRE(step >= 1 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f,
               VH_EINVAL, "vh_adamw_flat: step=%d lr=%g betas=(%g,%g) eps=%g", step, lr, beta1, beta2, eps);
    VH_REQUIRE(vh_aligned16(param) && vh_aligned16(grad) && vh_aligned16(exp_avg) && vh_aligned16(exp_avg_sq) &&
                   vh_aligned16(workspace),
               VH_EALIGN, "vh_adamw_flat: pointers must be 16-byte aligned");
    if (n == 0) return VH_OK;
    const int64_t n4 = n / 4;
    hipStream_t s = (hipStream_t)stream;
    double* partials = (double*)workspace;
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(VH_SUMSQ_BLOCKS), dim3(256), 0, s, grad, n4, partials);
    // bias corrections in double on the host, as torch.optim does in Python floats
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n4,
                       partials, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale,
                       max_norm, zero_grad, norm_out, block_slot, slot_step, guard);
    VH_CHECK_LAUNCH("vh_adamw_flat");
    return VH_OK;
}

// ---------------------------------------------------------------------------------------------
// AdaptiveLayerNorm projections of a whole stack in one launch each way: out[i] = emb W_i^T + b_i for the n = 2 L
// project_layer Linears (valle/models/modules.py:94-96), one (1, d) stage embedding for all of them.  As separate
// Linears their backward was ~10 tiny launches each (transpose, two skinny GEMMs, an M = 1 weight-gradient GEMM, a
// column sum, fills and adds): 300 of the 617 launches of a NAR training step.  The weight gradient of a Linear with
// ONE input row is an outer product: dW_i = dout_i^T emb, db_i = dout_i; demb = sum_i dout_i W_i.
// One wave per 8 output rows of one item; K = d <= 2048 (NV float4 per lane).
// ---------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void adaproj_fwd_kernel(const vh_adaproj_item* __restrict__ items,
                                                          const float* __restrict__ emb, float* __restrict__ out, int N,
                                                          int K) {
    const int lane = threadIdx.x & 63, item = blockIdx.y;
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    if (n0 >= N) return;
    const vh_adaproj_item it = items[item];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 e[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) e[i] = (lane + 64 * i) * 4 < K ? ld4(emb + (lane + 64 * i) * 4) : z;
    for (int n = n0; n < min(N, n0 + 8); ++n) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < K) acc += hsum4(ld4(it.w + (int64_t)n * K + c) * e[i]);
        }
        acc = wave_sum(acc);
        if (lane == 0) out[(int64_t)item * N + n] = acc + (it.b ? it.b[n] : 0.f);
    }
}

#define ADA_ROWS 16     // output rows per wave
template <int NV>
__global__ __launch_bounds__(256) void adaproj_bwd_kernel(const vh_adaproj_item* __restrict__ items,
                                                          const float* __restrict__ emb, const float* __restrict__ dout,
                                                          float* __restrict__ demb, int N, int K) {
    __shared__ __attribute__((aligned(16))) float red[4][NV * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, item = blockIdx.y;
    const int n0 = (blockIdx.x * 4 + w) * ADA_ROWS;
    const vh_adaproj_item it = items[item];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 e[NV], acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        e[i] = (lane + 64 * i) * 4 < K ? ld4(emb + (lane + 64 * i) * 4) : z;
        acc[i] = z;
    }
    for (int n = n0; n < min(N, n0 + ADA_ROWS); ++n) {
        const float g = dout[(int64_t)item * N + n];
        if (lane == 0 && it.db) it.db[n] = g;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < K) {
                if (it.dw) st4(it.dw + (int64_t)n * K + c, e[i] * g);
                acc[i] += ld4(it.w + (int64_t)n * K + c) * g;
            }
        }
    }
    // d emb: every (item, row block) adds a K-vector onto the SAME K addresses.  Per-wave atomics were 1.5 M adds onto
    // 512 addresses at configs[3] (24 norms x 1024 rows) and ran at the contended-atomic rate: 317 us for a launch whose
    // 50 MB of weight gradients need 10.  The four waves meet in LDS first and a wave covers 16 rows: 8 x fewer adds.
    if (demb) {                                          // (uniform: every thread reaches the barrier)
#pragma unroll
        for (int i = 0; i < NV; ++i) st4(&red[w][(lane + 64 * i) * 4], acc[i]);
        __syncthreads();
        for (int c = threadIdx.x; c < K; c += 256) atomicAdd(demb + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    }
}

static int adaproj_check(const char* name, const vh_adaproj_item* items, int n, const float* emb, int N, int K) {
    VH_REQUIRE(items && emb && n > 0 && n <= 4096, VH_EINVAL, "%s: null pointer or n=%d", name, n);
    VH_REQUIRE(N > 0 && K > 0 && K % 4 == 0 && K <= 2048, VH_EUNSUPPORTED, "%s: N=%d K=%d (K %% 4 == 0, K <= 2048)", name, N, K);
    VH_REQUIRE(vh_aligned16(emb), VH_EALIGN, "%s: emb must be 16-byte aligned", name);
    return VH_OK;
}

extern "C" int vh_adaproj_fwd(const vh_adaproj_item* items, int n, const float* emb, float* out, int N, int K, void* stream) {
    if (int rc = adaproj_check("vh_adaproj_fwd", items, n, emb, N, K)) return rc;
    VH_REQUIRE(out, VH_EINVAL, "vh_adaproj_fwd: null out");
    const dim3 grid((N + 31) / 32, n);
    hipStream_t st = (hipStream_t)stream;
    if (K <= 256) hipLaunchKernelGGL(adaproj_fwd_kernel<1>, grid, dim3(256), 0, st, items, emb, out, N, K);
    else if (K <= 512) hipLaunchKernelGGL(adaproj_fwd_kernel<2>, grid, dim3(256), 0, st, items, emb, out, N, K);
    else if (K <= 1024) hipLaunchKernelGGL(adaproj_fwd_kernel<4>, grid, dim3(256), 0, st, items, emb, out, N, K);
    else hipLaunchKernelGGL(adaproj_fwd_kernel<8>, grid, dim3(256), 0, st, items, emb, out, N, K);
    VH_CHECK_LAUNCH("vh_adaproj_fwd");
    return VH_OK;
}

extern "C" int vh_adaproj_bwd(const vh_adaproj_item* items, int n, const float* emb, const float* dout, float* demb,
                              int N, int K, void* stream) {
    if (int rc = adaproj_check("vh_adaproj_bwd", items, n, emb, N, K)) return rc;
    VH_REQUIRE(dout, VH_EINVAL, "vh_adaproj_bwd: null dout");
    const dim3 grid((N + 4 * ADA_ROWS - 1) / (4 * ADA_ROWS), n);
    hipStream_t st = (hipStream_t)stream;
    if (K <= 256) hipLaunchKernelGGL(adaproj_bwd_kernel<1>, grid, dim3(256), 0, st, items, emb, dout, demb, N, K);
    else if (K <= 512) hipLaunchKernelGGL(adaproj_bwd_kernel<2>, grid, dim3(256), 0, st, items, emb, dout, demb, N, K);
    else if (K <= 1024) hipLaunchKernelGGL(adaproj_bwd_kernel<4>, grid, dim3(256), 0, st, items, emb, dout, demb, N, K);
    else hipLaunchKernelGGL(adaproj_bwd_kernel<8>, grid, dim3(256), 0, st, items, emb, dout, demb, N, K);
    VH_CHECK_LAUNCH("vh_adaproj_bwd");
    return VH_OK;
}

