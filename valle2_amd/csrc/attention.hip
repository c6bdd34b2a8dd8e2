// Self-attention kernels over the (B, h, S_max, 64) fp32 KV cache.
//
//  attn_rows_kernel   : many query rows (prefill, NAR stage, training forward). Flash-style online
//                       softmax, fp32 MFMA 32x32x2 for both products, computed TRANSPOSED
//                       (Sᵀ = K·Qᵀ, Oᵀ = Vᵀ·Pᵀ) so a query is a lane: row max/sum are in-lane
//                       plus one cross-half shuffle, the P accumulator registers ARE the next
//                       MFMA's B operand, and the O rescale is lane-local.  K/V tiles of 32 keys
//                       are register-staged into double-buffered LDS.  Masks are analytic
//                       (prefix-LM / full / key length), never materialised.   MFMA-bound.
//  attn_decode_kernel : one query row per (batch, head) against the whole cache — the dominant
//                       kernel of AR decoding.  Pure HBM streaming: every K/V byte is read once
//                       with 16-B loads, 1 KiB per wave-instruction, 16 KiB in flight per wave;
//                       dot products reduce over the 16 lanes of a DPP row; online softmax per
//                       32-key chunk; waves combine through LDS, key splits through a small
//                       workspace.   HBM-bound (algorithmic bytes = 2·len·64·4 per (b,head)).
#include "vh_common.h"

#define HD VH_HEAD_DIM
#define LOG2E 1.44269504088896340736f
#define NEG_INF (-INFINITY)

// =============================================================================================
// decode
// =============================================================================================
#define PART_LD 72  // floats per partial record: o[64], m, l, pad

template <int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ kc,
    const float* __restrict__ vc, float* __restrict__ out, int ldo,
    const int32_t* __restrict__ cache_len, int len_bias, int n_heads, int S_max, int n_split,
    float* __restrict__ partial) {
    __shared__ float s_m[NW], s_l[NW];
    __shared__ __attribute__((aligned(16))) float s_o[NW][HD];
    const int bh = blockIdx.y, b = bh / n_heads, head = bh - b * n_heads;
    const int split = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int len = cache_len[b] + len_bias;
    const int nchunks = (len + 31) >> 5;
    const int cps = (nchunks + n_split - 1) / n_split;
    const int c_begin = split * cps;
    const int c_end = min(nchunks, c_begin + cps);

    const float qscale = 0.125f * LOG2E;  // 1/sqrt(64), folded with log2(e) for exp2
    const f32x4 q4 = ld4(q + (int64_t)b * ldq + head * HD + 4 * c16) * qscale;
    const float* kb = kc + (int64_t)bh * S_max * HD + 4 * c16;
    const float* vb = vc + (int64_t)bh * S_max * HD + 4 * c16;

    float m = NEG_INF, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int c = c_begin + w; c < c_end; c += NW) {
        const int key0 = c * 32 + g;
        f32x4 kf[8], vf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = key0 + 4 * i;
            const bool in = key < len;
            if (NT) {
                kf[i] = in ? ld4_stream(kb + (int64_t)key * HD) : f32x4{0.f, 0.f, 0.f, 0.f};
                vf[i] = in ? ld4_stream(vb + (int64_t)key * HD) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                kf[i] = in ? ld4(kb + (int64_t)key * HD) : f32x4{0.f, 0.f, 0.f, 0.f};
                vf[i] = in ? ld4(vb + (int64_t)key * HD) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        float s[8];
        float cmax = NEG_INF;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 t = kf[i] * q4;
            float d = row16_sum((t.x + t.y) + (t.z + t.w));
            s[i] = (key0 + 4 * i < len) ? d : NEG_INF;
            cmax = fmaxf(cmax, s[i]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m, cmax);  // finite: chunk c < nchunks holds >= 1 valid key
        const float alpha = exp2f(m - m_new);
        o *= alpha;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p = exp2f(s[i] - m_new);
            l += p;
            o += vf[i] * p;
        }
        m = m_new;
    }
    // fold the 4 key groups of the wave (lanes l, l^16, l^32, l^48 hold the same dims)
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        o.x += __shfl_xor(o.x, sh, 64); o.y += __shfl_xor(o.y, sh, 64);
        o.z += __shfl_xor(o.z, sh, 64); o.w += __shfl_xor(o.w, sh, 64);
        l += __shfl_xor(l, sh, 64);
    }
    if (lane < 16) st4(&s_o[w][4 * c16], o);
    if (lane == 0) { s_m[w] = m; s_l[w] = l; }
    __syncthreads();
    if (tid < HD) {
        float M = s_m[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) M = fmaxf(M, s_m[k]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const float wgt = s_m[k] == NEG_INF ? 0.f : exp2f(s_m[k] - M);
            L += s_l[k] * wgt;
            O += s_o[k][tid] * wgt;
        }
        if (n_split == 1) {
            out[(int64_t)b * ldo + head * HD + tid] = O / L;
        } else {
            float* pr = partial + ((int64_t)bh * n_split + split) * PART_LD;
            pr[tid] = O;
            if (tid == 0) { pr[HD] = M; pr[HD + 1] = L; }
        }
    }
}

// Pipelined variant: 16-key chunks, two named register sets (A/B): the loads of chunk c+NW are in
// flight while chunk c is reduced, so the wave never idles between a burst and its use.
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_pipe_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ kc,
    const float* __restrict__ vc, float* __restrict__ out, int ldo,
    const int32_t* __restrict__ cache_len, int len_bias, int n_heads, int S_max, int n_split,
    float* __restrict__ partial) {
    __shared__ float s_m[NW], s_l[NW];
    __shared__ __attribute__((aligned(16))) float s_o[NW][HD];
    const int bh = blockIdx.y, b = bh / n_heads, head = bh - b * n_heads;
    const int split = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int len = cache_len[b] + len_bias;
    const int nchunks = (len + 15) >> 4;
    const int cps = (nchunks + n_split - 1) / n_split;
    const int c_begin = split * cps;
    const int c_end = min(nchunks, c_begin + cps);

    const float qscale = 0.125f * LOG2E;
    const f32x4 q4 = ld4(q + (int64_t)b * ldq + head * HD + 4 * c16) * qscale;
    const float* kb = kc + (int64_t)bh * S_max * HD + 4 * c16;
    const float* vb = vc + (int64_t)bh * S_max * HD + 4 * c16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    float m = NEG_INF, l = 0.f;
    f32x4 o = zero4;
    f32x4 kA[4], vA[4], kB[4], vB[4];
    auto load = [&](int c, f32x4 (&kf)[4], f32x4 (&vf)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = c * 16 + g + 4 * i;
            const bool in = c < c_end && key < len;
            kf[i] = in ? ld4(kb + (int64_t)key * HD) : zero4;
            vf[i] = in ? ld4(vb + (int64_t)key * HD) : zero4;
        }
    };
    auto reduce = [&](int c, const f32x4 (&kf)[4], const f32x4 (&vf)[4]) {
        float s[4];
        float cmax = NEG_INF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 t = kf[i] * q4;
            const float d = row16_sum((t.x + t.y) + (t.z + t.w));
            s[i] = (c * 16 + g + 4 * i < len) ? d : NEG_INF;
            cmax = fmaxf(cmax, s[i]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m, cmax);
        const float alpha = exp2f(m - m_new);
        o *= alpha;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float p = exp2f(s[i] - m_new);
            l += p;
            o += vf[i] * p;
        }
        m = m_new;
    };
    int c = c_begin + w;
    load(c, kA, vA);
    while (c < c_end) {
        load(c + NW, kB, vB);
        reduce(c, kA, vA);
        c += NW;
        if (c >= c_end) break;
        load(c + NW, kA, vA);
        reduce(c, kB, vB);
        c += NW;
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        o.x += __shfl_xor(o.x, sh, 64); o.y += __shfl_xor(o.y, sh, 64);
        o.z += __shfl_xor(o.z, sh, 64); o.w += __shfl_xor(o.w, sh, 64);
        l += __shfl_xor(l, sh, 64);
    }
    if (lane < 16) st4(&s_o[w][4 * c16], o);
    if (lane == 0) { s_m[w] = m; s_l[w] = l; }
    __syncthreads();
    if (tid < HD) {
        float M = s_m[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) M = fmaxf(M, s_m[k]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const float wgt = s_m[k] == NEG_INF ? 0.f : exp2f(s_m[k] - M);
            L += s_l[k] * wgt;
            O += s_o[k][tid] * wgt;
        }
        if (n_split == 1) {
            out[(int64_t)b * ldo + head * HD + tid] = O / L;
        } else {
            float* pr = partial + ((int64_t)bh * n_split + split) * PART_LD;
            pr[tid] = O;
            if (tid == 0) { pr[HD] = M; pr[HD + 1] = L; }
        }
    }
}

__global__ __launch_bounds__(64) void attn_decode_combine_kernel(
    const float* __restrict__ partial, float* __restrict__ out, int ldo, int n_heads, int n_split) {
    const int bh = blockIdx.x, b = bh / n_heads, head = bh - b * n_heads;
    const int tid = threadIdx.x;
    const float* pr = partial + (int64_t)bh * n_split * PART_LD;
    float M = NEG_INF;
    for (int s = 0; s < n_split; ++s) M = fmaxf(M, pr[s * PART_LD + HD]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < n_split; ++s) {
        const float ms = pr[s * PART_LD + HD];
        const float wgt = ms == NEG_INF ? 0.f : exp2f(ms - M);
        L += pr[s * PART_LD + HD + 1] * wgt;
        O += pr[s * PART_LD + tid] * wgt;
    }
    out[(int64_t)b * ldo + head * HD + tid] = O / L;
}

extern "C" size_t vh_attn_decode_ws_bytes(int B, int n_heads, int n_split) {
    if (n_split <= 1) return 0;
    return (size_t)B * n_heads * n_split * PART_LD * sizeof(float);
}

extern "C" int vh_attn_decode(const float* q, int ldq, const float* kcache, const float* vcache,
                              float* out, int ldo, const int32_t* cache_len, int len_bias, int B,
                              int n_heads, int S_max, int n_split, void* partial, void* stream) {
    VH_REQUIRE(q && kcache && vcache && out && cache_len, VH_EINVAL, "vh_attn_decode: null pointer");
    VH_REQUIRE(B > 0 && n_heads > 0 && S_max > 0 && n_split >= 1 && n_split <= 64, VH_EINVAL,
               "vh_attn_decode: bad dims B=%d h=%d S_max=%d n_split=%d", B, n_heads, S_max, n_split);
    VH_REQUIRE(len_bias == 0 || len_bias == 1, VH_EINVAL, "vh_attn_decode: len_bias=%d", len_bias);
    VH_REQUIRE(ldq % 4 == 0 && ldq >= n_heads * HD && ldo >= n_heads * HD, VH_EINVAL,
               "vh_attn_decode: ldq=%d ldo=%d", ldq, ldo);
    VH_REQUIRE(vh_aligned16(q) && vh_aligned16(kcache) && vh_aligned16(vcache), VH_EALIGN,
               "vh_attn_decode: pointers must be 16-byte aligned");
    VH_REQUIRE(n_split == 1 || partial, VH_EINVAL, "vh_attn_decode: n_split>1 needs a workspace");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(n_split, B * n_heads);
    // few workgroups → 16 waves each (one (b,head) can own a whole CU); many → 4 waves each
    const bool big = (int64_t)B * n_heads * n_split < 1024;
#define AD(KERN, ...)                                                                              \
    hipLaunchKernelGGL((KERN<__VA_ARGS__>), grid, dim3(waves * 64), 0, s, q, ldq, kcache, vcache, out, ldo, \
                       cache_len, len_bias, n_heads, S_max, n_split, (float*)partial)
    const int variant = vh_tuning(VH_TUNE_DECODE_VARIANT);
    const int nw = vh_tuning(VH_TUNE_DECODE_WAVES);
    const int waves = nw ? nw : (big ? 16 : 4);
    if (variant == 3) {          // burst kernel with plain (temporal) loads, for A/B runs
        if (waves == 16) AD(attn_decode_kernel, 16, false); else if (waves == 8) AD(attn_decode_kernel, 8, false);
        else AD(attn_decode_kernel, 4, false);
    } else if (variant != 2) {   // default: burst kernel, non-temporal K/V loads
        if (waves == 16) AD(attn_decode_kernel, 16, true); else if (waves == 8) AD(attn_decode_kernel, 8, true);
        else AD(attn_decode_kernel, 4, true);
    } else {
        if (waves == 16) AD(attn_decode_pipe_kernel, 16); else if (waves == 8) AD(attn_decode_pipe_kernel, 8);
        else AD(attn_decode_pipe_kernel, 4);
    }
#undef AD
    if (n_split > 1)
        hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(B * n_heads), dim3(64), 0, s,
                           (const float*)partial, out, ldo, n_heads, n_split);
    VH_CHECK_LAUNCH("vh_attn_decode");
    return VH_OK;
}

// =============================================================================================
// many-row attention
// =============================================================================================
#define QB 128   // queries per block (4 waves x 32)
#define KT 32    // keys per tile
#define KLD 68   // LDS row stride (floats): 17 x 16-B slots → conflict-free ds_read_b128 by row

struct RowsArgs {
    const float* q; int ldq;
    const float* kc; const float* vc;
    float* out; int ldo;
    int n_heads, Tq, Tk, S_max, mode, x_len;
    const int32_t* x_len_dev; const int32_t* kv_len;
    const uint8_t* mask; const uint8_t* pad;
    int n_qblocks;
};

__global__ __launch_bounds__(256, 2) void attn_rows_kernel(RowsArgs a) {
    // [buf][K|V][key][KLD]; reused at the end as the [128][KLD] output transpose buffer
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * KT * KLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // 1-D grid, heaviest query blocks first: under the causal/prefix mask the last query block of
    // a head visits up to 4x the key tiles of the first, and with only ~4 workgroups per CU slot a
    // heavy block dispatched late becomes the tail.  id % (B*h) is the (b,head), so workgroups that
    // share an XCD (ids b, b+8, ...) still see every query block: no XCD gets only heavy ones.
    const int n_bh = gridDim.x / a.n_qblocks;
    const int bh = blockIdx.x % n_bh, b = bh / a.n_heads, head = bh - b * a.n_heads;
    const int q0 = (a.n_qblocks - 1 - (int)(blockIdx.x / n_bh)) * QB;
    const int q_off = a.Tk - a.Tq;
    const int kvl = a.kv_len ? min(a.kv_len[b], a.Tk) : a.Tk;
    const int xl = a.x_len_dev ? a.x_len_dev[b] : a.x_len;

    // number of key tiles this block has to visit
    const int last_pos = q_off + min(q0 + QB, a.Tq) - 1;
    int kmax = kvl;
    if (a.mode == VH_MASK_PREFIX) kmax = min(kvl, max(xl, last_pos >= xl ? last_pos + 1 : 0));
    if (a.mode == VH_MASK_EXPLICIT) kmax = a.Tk;
    const int n_tiles = (kmax + KT - 1) / KT;

    // Q fragment (B operand of Sᵀ = K·Qᵀ): lane holds Q[qi][8t+4h+j], pre-scaled
    const int qi = q0 + w * 32 + r;            // this lane's query row
    const int qpos = q_off + qi;               // its key position
    const float qscale = 0.125f * LOG2E;
    f32x4 qf[8];
    {
        const float* qp = a.q + ((int64_t)b * a.Tq + qi) * a.ldq + head * HD + 4 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            qf[t] = qi < a.Tq ? ld4(qp + 8 * t) * qscale : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const float* kb = a.kc + (int64_t)bh * a.S_max * HD;
    const float* vb = a.vc + (int64_t)bh * a.S_max * HD;
    const int skey = tid >> 4, squad = (tid & 15) * 4;  // staging: 2 keys per thread per operand
    f32x4 rk[2], rv[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k0 + skey + 16 * i;
            const bool in = key < a.Tk;  // rows < Tk were all written by this forward's QKV GEMM
            rk[i] = in ? ld4(kb + (int64_t)key * HD + squad) : f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = in ? ld4(vb + (int64_t)key * HD + squad) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
        float* kd = lds + buf * (2 * KT * KLD);
        float* vd = kd + KT * KLD;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            st4(kd + (skey + 16 * i) * KLD + squad, rk[i]);
            st4(vd + (skey + 16 * i) * KLD + squad, rv[i]);
        }
    };

    f32x16 O0, O1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { O0[e] = 0.f; O1[e] = 0.f; }
    float m = NEG_INF, l = 0.f;

    // positions covered by this wave's 32 queries (for tile skipping / fast path)
    const int wpos_min = q_off + q0 + w * 32, wpos_max = wpos_min + 31;

    if (n_tiles > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int cur = kt & 1, k0 = kt * KT;
        if (kt + 1 < n_tiles) gload(k0 + KT);

        bool any = k0 < kvl, all = k0 + KT <= kvl;
        if (a.mode == VH_MASK_PREFIX) {
            any = any && (k0 < xl || (wpos_max >= xl && k0 <= wpos_max));
            all = all && (k0 + KT <= xl || (wpos_min >= xl && k0 + KT - 1 <= wpos_min));
        } else if (a.mode == VH_MASK_EXPLICIT) {
            any = true; all = false;
        }
        if (any) {
            const float* ks = lds + cur * (2 * KT * KLD);
            const float* vs = ks + KT * KLD;
            // ---- Sᵀ[key][q] = sum_d K[key][d] Q[q][d]
            f32x16 S;
#pragma unroll
            for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 kf = ld4(ks + r * KLD + 8 * t + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[t][j], S, 0, 0, 0);
            }
            // ---- mask: reg e ↔ key k0 + (e&3) + 8(e>>2) + 4h ; lane ↔ query qi
            if (!all) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    bool vis;
                    if (a.mode == VH_MASK_EXPLICIT) {
                        vis = key < a.Tk && qi < a.Tq && !a.mask[(int64_t)qi * a.Tk + key] &&
                              !(a.pad && a.pad[(int64_t)b * a.Tk + key]);
                    } else {
                        vis = key < kvl;
                        if (a.mode == VH_MASK_PREFIX) vis = vis && (key < xl || (qpos >= xl && key <= qpos));
                    }
                    if (!vis) S[e] = NEG_INF;
                }
            }
            // ---- online softmax (per lane = per query; the other 16 keys sit in lane^32)
            float tmax = S[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, S[e]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m, tmax);
            const float m_use = m_new == NEG_INF ? 0.f : m_new;  // fully masked so far: p = 0
            const float alpha = exp2f(m - m_use);
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                S[e] = exp2f(S[e] - m_use);
                psum += S[e];
            }
            l = l * alpha + psum;
            m = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) { O0[e] *= alpha; O1[e] *= alpha; }
            // ---- Oᵀ[d][q] += sum_key V[key][d] P[key][q]; k-step e covers keys (e,h=0),(e,h=1)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float* vrow = vs + ((e & 3) + 8 * (e >> 2) + 4 * h) * KLD + r;
                O0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], S[e], O0, 0, 0, 0);
                O1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], S[e], O1, 0, 0, 0);
            }
        }
        if (kt + 1 < n_tiles) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise, transpose through LDS, store coalesced rows
    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = 1.0f / ltot;  // fully masked row → 0/0 = NaN, as SDPA gives
    float* ob = lds;                // [128][KLD]; all K/V reads finished at the last barrier
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        // regs 4g4..4g4+3 ↔ d = 8*g4 + 4h + {0..3} (+32 for O1)
        f32x4 v0 = {O0[4 * g4], O0[4 * g4 + 1], O0[4 * g4 + 2], O0[4 * g4 + 3]};
        f32x4 v1 = {O1[4 * g4], O1[4 * g4 + 1], O1[4 * g4 + 2], O1[4 * g4 + 3]};
        st4(ob + (w * 32 + r) * KLD + 8 * g4 + 4 * h, v0 * inv);
        st4(ob + (w * 32 + r) * KLD + 32 + 8 * g4 + 4 * h, v1 * inv);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = skey + 16 * i;
        if (q0 + row < a.Tq)
            st4(a.out + ((int64_t)b * a.Tq + q0 + row) * a.ldo + head * HD + squad,
                ld4(ob + row * KLD + squad));
    }
}

extern "C" int vh_attn_rows(const float* q, int ldq, const float* kcache, const float* vcache,
                            float* out, int ldo, int B, int n_heads, int Tq, int Tk, int S_max,
                            int mode, int x_len, const int32_t* x_len_dev, const int32_t* kv_len,
                            const uint8_t* mask, const uint8_t* pad, void* stream) {
    VH_REQUIRE(q && kcache && vcache && out, VH_EINVAL, "vh_attn_rows: null pointer");
    VH_REQUIRE(B >= 0 && n_heads > 0 && Tq >= 0 && Tk >= Tq && S_max >= Tk, VH_EINVAL,
               "vh_attn_rows: bad dims B=%d h=%d Tq=%d Tk=%d S_max=%d", B, n_heads, Tq, Tk, S_max);
    VH_REQUIRE(mode == VH_MASK_FULL || mode == VH_MASK_PREFIX || mode == VH_MASK_EXPLICIT, VH_EINVAL,
               "vh_attn_rows: mode=%d", mode);
    VH_REQUIRE(mode != VH_MASK_EXPLICIT || mask, VH_EINVAL, "vh_attn_rows: explicit mode needs mask");
    VH_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ldq >= n_heads * HD && ldo >= n_heads * HD, VH_EINVAL,
               "vh_attn_rows: ldq=%d ldo=%d", ldq, ldo);
    VH_REQUIRE(vh_aligned16(q) && vh_aligned16(kcache) && vh_aligned16(vcache) && vh_aligned16(out),
               VH_EALIGN, "vh_attn_rows: pointers must be 16-byte aligned");
    if (B == 0 || Tq == 0) return VH_OK;
    const int nqb = (Tq + QB - 1) / QB;
    RowsArgs a{q, ldq, kcache, vcache, out, ldo, n_heads, Tq, Tk, S_max, mode, x_len,
               x_len_dev, kv_len, mask, pad, nqb};
    dim3 grid(nqb * B * n_heads);
    hipLaunchKernelGGL(attn_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    VH_CHECK_LAUNCH("vh_attn_rows");
    return VH_OK;
}
