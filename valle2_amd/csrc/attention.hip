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
#include <hip/hip_ext.h>
#include "vh_common.h"
#include <mutex>
#include <queue>
#include <type_traits>
#include <unordered_map>
#include <vector>

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
    float* __restrict__ partial, unsigned* __restrict__ arrived) {
    __shared__ float s_m[NW], s_l[NW];
    __shared__ __attribute__((aligned(16))) float s_o[NW][HD];
    __shared__ int s_last;
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
        const float alpha = vh_exp2(m - m_new);
        o *= alpha;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p = vh_exp2(s[i] - m_new);
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
            const float wgt = s_m[k] == NEG_INF ? 0.f : vh_exp2(s_m[k] - M);
            L += s_l[k] * wgt;
            O += s_o[k][tid] * wgt;
        }
        if (n_split == 1) {
            out[(int64_t)b * ldo + head * HD + tid] = O / L;
        } else if (arrived == nullptr) {                     // two-launch form: plain stores, the next launch reads them
            float* pr = partial + ((int64_t)bh * n_split + split) * PART_LD;
            pr[tid] = O;
            if (tid == 0) { pr[HD] = M; pr[HD + 1] = L; }
        } else {                                             // handed to another workgroup of THIS launch: write-through
            float* pr = partial + ((int64_t)bh * n_split + split) * PART_LD;
            __hip_atomic_store(pr + tid, O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                __hip_atomic_store(pr + HD, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pr + HD + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // Key splits, combined by the LAST workgroup of the (b, head) to arrive instead of by a second launch (opt-in,
    // VH_TUNE_DECODE_COMBINE = 1: measured slower than the launch it removes, see vh_attn_decode).  The 8 XCDs have
    // private L2s, so the record leaves as agent-scope (sc1, write-through) stores that the storing wave drains before
    // one lane takes a ticket with a relaxed agent-scope add — no release fence: an agent-scope release writes back the
    // whole L2 and made this form 240 us per step SLOWER than the second launch it replaces
    // (cdna_hip_programming.md Guideline 16, recipe R1).  The workgroup that draws the last ticket reads every record with
    // agent-scope loads (they bypass its L1: no acquire either) and adds them IN SPLIT ORDER — the result does not depend
    // on who came last — then re-arms the ticket word for the next launch of the stream (words start at zero:
    // vh_attn_decode_ws_bytes).
    if (arrived == nullptr) return;                          // (uniform: n_split == 1, or the two-launch form)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave: its record has left
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(arrived + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == (unsigned)(n_split - 1);
        if (s_last) __hip_atomic_store(arrived + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the ticket)
    if (tid < HD) {
        const float* pr = partial + (int64_t)bh * n_split * PART_LD;
        auto ldg = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        float M = NEG_INF;
        for (int k = 0; k < n_split; ++k) M = fmaxf(M, ldg(pr + k * PART_LD + HD));
        float L = 0.f, O = 0.f;
        for (int k = 0; k < n_split; ++k) {
            const float ms = ldg(pr + k * PART_LD + HD);
            const float wgt = ms == NEG_INF ? 0.f : vh_exp2(ms - M);
            L += ldg(pr + k * PART_LD + HD + 1) * wgt;
            O += ldg(pr + k * PART_LD + tid) * wgt;
        }
        out[(int64_t)b * ldo + head * HD + tid] = O / L;
    }
}

// Ring variant (round 2).  The burst kernel has its 256 KB per CU in flight only at the moment its 16 waves have
// all just issued; each wave then waits for its whole burst, reduces it, and only then asks again, and its very first
// request waits for a dependent load of cache_len[b].  Here a wave owns a ring of D register sets of 32 keys each:
// D-1 bursts are always outstanding while one is reduced, and the first D-1 bursts are issued BEFORE the context
// length is known (addresses clamped inside the cache allocation; what lies beyond the row's length is masked once
// the length has arrived), so the stream starts with the kernel.  The read ceiling of this part is 6.3-6.5 TB/s
// for any read-only kernel (tools/probe_read_bw.hip, corrected: its first version dropped loop remainders and
// reported 8 TB/s); the 8 x 2 ring reaches 6.0-6.1 TB/s inside the decode step.
template <int NW, int D, int CK = 32>
__global__ __launch_bounds__(NW * 64) void attn_decode_ring_kernel(
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
    const float* kb = kc + (int64_t)bh * S_max * HD + 4 * c16;
    const float* vb = vc + (int64_t)bh * S_max * HD + 4 * c16;

    constexpr int LPS = CK / 4;                                    // loads per set and operand: 4 keys per wave instruction
    f32x4 kf[D][LPS], vf[D][LPS];
    int key_limit = S_max - 1;                                     // before the length is known: stay inside the allocation
    auto load = [&](int c, f32x4 (&kq)[LPS], f32x4 (&vq)[LPS]) {  // unconditional loads, clamped row index
        const int key0 = c * CK + g;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int key = min(key0 + 4 * i, key_limit);
            kq[i] = ld4_stream(kb + (int64_t)key * HD);
            vq[i] = ld4_stream(vb + (int64_t)key * HD);
        }
    };
    // speculative start (n_split == 1: the wave's chunks are w, w + NW, ... whatever the length is)
    const bool spec = n_split == 1;
    if (spec) {
#pragma unroll
        for (int j = 0; j < D - 1; ++j) load(w + j * NW, kf[j], vf[j]);
    }
    const int len = cache_len[b] + len_bias;
    const int nchunks = (len + CK - 1) / CK;
    const int cps = (nchunks + n_split - 1) / n_split;
    const int c_begin = split * cps;
    const int c_end = min(nchunks, c_begin + cps);
    key_limit = len - 1;                 // from here on the tail re-reads the row's last key instead of rows beyond it
    const float qscale = 0.125f * LOG2E;
    const f32x4 q4 = ld4(q + (int64_t)b * ldq + head * HD + 4 * c16) * qscale;
    if (!spec) {
#pragma unroll
        for (int j = 0; j < D - 1; ++j)
            if (c_begin + w + j * NW < c_end) load(c_begin + w + j * NW, kf[j], vf[j]);
    }

    float m = NEG_INF, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    auto reduce = [&](int c, const f32x4 (&kq)[LPS], const f32x4 (&vq)[LPS]) {
        const int key0 = c * CK + g;
        const bool whole = c * CK + CK <= len;                    // wave-uniform: no masking for interior chunks
        float sc[LPS];
        float cmax = NEG_INF;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const f32x4 t = kq[i] * q4;
            const float d = row16_sum((t.x + t.y) + (t.z + t.w));
            sc[i] = (whole || key0 + 4 * i < len) ? d : NEG_INF;
            cmax = fmaxf(cmax, sc[i]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m, cmax);  // finite: chunk c < nchunks holds >= 1 valid key
        const float alpha = vh_exp2(m - m_new);
        o *= alpha;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const float p = vh_exp2(sc[i] - m_new);
            l += p;
            // rows beyond the length hold whatever the allocation held (possibly NaN): select, do not multiply by 0
            const f32x4 vv = (whole || key0 + 4 * i < len) ? vq[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            o += vv * p;
        }
        m = m_new;
    };
    for (int c0 = c_begin + w; c0 < c_end; c0 += D * NW) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int c = c0 + j * NW;
            if (c < c_end) {
                const int cn = c + (D - 1) * NW;                 // the burst that keeps D - 1 outstanding
                if (cn < c_end) load(cn, kf[(j + D - 1) % D], vf[(j + D - 1) % D]);
                reduce(c, kf[j], vf[j]);
            }
        }
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
            const float wgt = s_m[k] == NEG_INF ? 0.f : vh_exp2(s_m[k] - M);
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

// =============================================================================================
// perf mode: the same single-row attention over a bf16 K/V cache (B, h, S_max, 64) — half the bytes of the stream
// that bounds the decode step; q, the softmax and the accumulators stay fp32.  A key row is 64 x 2 B = 128 B = 8
// lanes x 16 B: lane l holds dimensions 8 (l & 7) .. +7 of key (l >> 3) of a group of 8 keys, so one wave-instruction
// still moves 1 KiB.  Ring of D register sets of 32 keys (4 loads per operand and set), speculative first bursts
// before the row's length is known, non-temporal loads — the structure of attn_decode_ring_kernel.
// =============================================================================================
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld16_stream(const uint16_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
// sum over the 8 lanes of a half DPP row (lanes 8j .. 8j+7)
__device__ __forceinline__ float row8_sum(float v) {
    v += dpp_get<0xB1, 0xF>(v, 0.f);
    v += dpp_get<0x4E, 0xF>(v, 0.f);
    v += dpp_get<0x141, 0xF>(v, 0.f);     // row_half_mirror: lane i <-> 7 - i inside each half row
    return v;
}
#define BF_LO(w) vh_h16_lo(w)
#define BF_HI(w) vh_h16_hi(w)

template <int NW, int D>
__global__ __launch_bounds__(NW * 64) void attn_decode_ring16_kernel(
    const float* __restrict__ q, int ldq, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    float* __restrict__ out, int ldo, const int32_t* __restrict__ cache_len, int len_bias, int n_heads, int S_max) {
    __shared__ float s_m[NW], s_l[NW];
    __shared__ __attribute__((aligned(16))) float s_o[NW][HD];
    const int bh = blockIdx.y, b = bh / n_heads, head = bh - b * n_heads;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c8 = lane & 7, kg = lane >> 3;
    const uint16_t* kb = kc + (int64_t)bh * S_max * HD + 8 * c8;
    const uint16_t* vb = vc + (int64_t)bh * S_max * HD + 8 * c8;
    constexpr int LPS = 4;                                         // 8 keys per wave-instruction
    u32x4 kf[D][LPS], vf[D][LPS];
    int key_limit = S_max - 1;
    auto load = [&](int c, u32x4 (&kq)[LPS], u32x4 (&vq)[LPS]) {
        const int key0 = c * 32 + kg;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int key = min(key0 + 8 * i, key_limit);
            kq[i] = ld16_stream(kb + (int64_t)key * HD);
            vq[i] = ld16_stream(vb + (int64_t)key * HD);
        }
    };
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load(w + j * NW, kf[j], vf[j]);
    const int len = cache_len[b] + len_bias;
    const int c_end = (len + 31) / 32;
    key_limit = len - 1;
    const float qscale = 0.125f * LOG2E;
    const float* qp = q + (int64_t)b * ldq + head * HD + 8 * c8;
    const f32x4 qa = ld4(qp) * qscale, qb = ld4(qp + 4) * qscale;

    float m = NEG_INF, l = 0.f;
    f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
    auto reduce = [&](int c, const u32x4 (&kq)[LPS], const u32x4 (&vq)[LPS]) {
        const int key0 = c * 32 + kg;
        const bool whole = c * 32 + 32 <= len;
        float sc[LPS];
        float cmax = NEG_INF;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const u32x4 kk = kq[i];
            float d = BF_LO(kk.x) * qa.x + BF_HI(kk.x) * qa.y + BF_LO(kk.y) * qa.z + BF_HI(kk.y) * qa.w;
            d += BF_LO(kk.z) * qb.x + BF_HI(kk.z) * qb.y + BF_LO(kk.w) * qb.z + BF_HI(kk.w) * qb.w;
            d = row8_sum(d);
            sc[i] = (whole || key0 + 8 * i < len) ? d : NEG_INF;
            cmax = fmaxf(cmax, sc[i]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 8, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m, cmax);
        const float alpha = vh_exp2(m - m_new);
        oa *= alpha; ob *= alpha;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const float p = vh_exp2(sc[i] - m_new);
            l += p;
            u32x4 vv = vq[i];
            if (!(whole || key0 + 8 * i < len)) vv = u32x4{0u, 0u, 0u, 0u};   // rows beyond the length: select, not * 0
            oa += f32x4{BF_LO(vv.x), BF_HI(vv.x), BF_LO(vv.y), BF_HI(vv.y)} * p;
            ob += f32x4{BF_LO(vv.z), BF_HI(vv.z), BF_LO(vv.w), BF_HI(vv.w)} * p;
        }
        m = m_new;
    };
    for (int c0 = w; c0 < c_end; c0 += D * NW) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int c = c0 + j * NW;
            if (c < c_end) {
                const int cn = c + (D - 1) * NW;
                if (cn < c_end) load(cn, kf[(j + D - 1) % D], vf[(j + D - 1) % D]);
                reduce(c, kf[j], vf[j]);
            }
        }
    }
    // fold the 8 key groups of the wave (lanes l, l^8, l^16, l^32 hold the same dimensions)
#pragma unroll
    for (int sh = 8; sh <= 32; sh <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            oa[j] += __shfl_xor(oa[j], sh, 64);
            ob[j] += __shfl_xor(ob[j], sh, 64);
        }
        l += __shfl_xor(l, sh, 64);
    }
    if (lane < 8) { st4(&s_o[w][8 * c8], oa); st4(&s_o[w][8 * c8 + 4], ob); }
    if (lane == 0) { s_m[w] = m; s_l[w] = l; }
    __syncthreads();
    if (tid < HD) {
        float M = s_m[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) M = fmaxf(M, s_m[k]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const float wgt = s_m[k] == NEG_INF ? 0.f : vh_exp2(s_m[k] - M);
            L += s_l[k] * wgt;
            O += s_o[k][tid] * wgt;
        }
        out[(int64_t)b * ldo + head * HD + tid] = O / L;
    }
}

// fp32 cache rows -> bf16 cache rows (round to nearest even): the prompt pass runs in fp32, its K/V are narrowed once
__global__ __launch_bounds__(256) void kv_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                         int rows, int64_t src_stride, int64_t dst_stride) {
    const int stream = blockIdx.y;
    const float* s = src + (int64_t)stream * src_stride;
    uint16_t* d = dst + (int64_t)stream * dst_stride;
    const int64_t n8 = (int64_t)rows * HD / 8;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f32x4 a = ld4(s + 8 * i), c = ld4(s + 8 * i + 4);
        u32x4 o;
        o.x = vh_pack_h16(a.x, a.y); o.y = vh_pack_h16(a.z, a.w);
        o.z = vh_pack_h16(c.x, c.y); o.w = vh_pack_h16(c.z, c.w);
        *reinterpret_cast<u32x4*>(d + 8 * i) = o;
    }
}

extern "C" int vh_kv_to_bf16(const float* src, uint16_t* dst, int n_streams, int rows, int S_src, int S_dst, void* stream) {
    VH_REQUIRE(src && dst && n_streams > 0 && rows >= 0 && rows <= S_src && rows <= S_dst, VH_EINVAL,
               "vh_kv_to_bf16: n_streams=%d rows=%d S_src=%d S_dst=%d", n_streams, rows, S_src, S_dst);
    VH_REQUIRE(vh_aligned16(src) && vh_aligned16(dst), VH_EALIGN, "vh_kv_to_bf16: pointers must be 16-byte aligned");
    if (rows == 0) return VH_OK;
    const int bx = (int)min((int64_t)64, ((int64_t)rows * HD / 8 + 255) / 256);
    hipLaunchKernelGGL(kv_to_bf16_kernel, dim3(bx, n_streams), dim3(256), 0, (hipStream_t)stream, src, dst, rows,
                       (int64_t)S_src * HD, (int64_t)S_dst * HD);
    VH_CHECK_LAUNCH("vh_kv_to_bf16");
    return VH_OK;
}

static thread_local hipEvent_t g_attn_ev[2] = {nullptr, nullptr};
void vh_internal_attn_decode_events(hipEvent_t start, hipEvent_t stop) { g_attn_ev[0] = start; g_attn_ev[1] = stop; }

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
        const float wgt = ms == NEG_INF ? 0.f : vh_exp2(ms - M);
        L += pr[s * PART_LD + HD + 1] * wgt;
        O += pr[s * PART_LD + tid] * wgt;
    }
    out[(int64_t)b * ldo + head * HD + tid] = O / L;
}

extern "C" int vh_attn_decode_kv16(const float* q, int ldq, const uint16_t* kcache16, const uint16_t* vcache16,
                                   float* out, int ldo, const int32_t* cache_len, int len_bias, int B, int n_heads,
                                   int S_max, void* stream) {
    VH_REQUIRE(q && kcache16 && vcache16 && out && cache_len, VH_EINVAL, "vh_attn_decode_kv16: null pointer");
    VH_REQUIRE(B > 0 && n_heads > 0 && S_max > 0 && (len_bias == 0 || len_bias == 1), VH_EINVAL,
               "vh_attn_decode_kv16: bad dims B=%d h=%d S_max=%d len_bias=%d", B, n_heads, S_max, len_bias);
    VH_REQUIRE(ldq % 4 == 0 && ldq >= n_heads * HD && ldo >= n_heads * HD, VH_EINVAL, "vh_attn_decode_kv16: ldq=%d ldo=%d",
               ldq, ldo);
    VH_REQUIRE(vh_aligned16(q) && vh_aligned16(kcache16) && vh_aligned16(vcache16), VH_EALIGN,
               "vh_attn_decode_kv16: pointers must be 16-byte aligned");
#define AD16(NW, D)                                                                                                  \
    hipExtLaunchKernelGGL((attn_decode_ring16_kernel<NW, D>), dim3(1, B * n_heads), dim3(NW * 64), 0, (hipStream_t)stream, \
                          g_attn_ev[0], g_attn_ev[1], 0, q, ldq, kcache16, vcache16, out, ldo, cache_len, len_bias, n_heads, \
                          S_max)
    // ring shape (waves x register sets of 32 keys): 8 x 2 as the fp32 kernel; 16 x 1, 16 x 2, 8 x 3, 8 x 4 and 4 x 4
    // measured within 3 % of it (461-477 us per decode step, profiles/r3_ab_decode_perf_mode.log)
    AD16(8, 2);
#undef AD16
    VH_CHECK_LAUNCH("vh_attn_decode_kv16");
    return VH_OK;
}

// workspace of a key-split launch: [B h][n_split] partial records, then B h ticket words (one per (b, head), padded to 16 B)
static size_t decode_records_bytes(int B, int n_heads, int n_split) {
    return (size_t)B * n_heads * n_split * PART_LD * sizeof(float);
}
extern "C" size_t vh_attn_decode_ws_bytes(int B, int n_heads, int n_split) {
    if (n_split <= 1) return 0;
    return decode_records_bytes(B, n_heads, n_split) + ((size_t)B * n_heads * sizeof(unsigned) + 15) / 16 * 16;
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
    // g_attn_ev (set only by vh_ar_decoder_profile_attn): start/stop events attached to the kernel's own dispatch
    // packet (hipExtLaunchKernelGGL), i.e. the kernel's begin/end timestamps rather than a marker bracket.
#define AD(KERN, ...)                                                                              \
    hipExtLaunchKernelGGL((KERN<__VA_ARGS__>), grid, dim3(waves * 64), 0, s, g_attn_ev[0], g_attn_ev[1], 0, q, ldq, \
                          kcache, vcache, out, ldo, cache_len, len_bias, n_heads, S_max, n_split, (float*)partial, arrived)
    // default: one (b, head) per CU and no key split -> the ring kernel (8 waves x 2 register sets of 32 keys, speculative
    // start: 605.9 vs 615.2 us per decode step, profiles/r2_ab_decode_ring.log); otherwise — key splits, more (b, head)
    // pairs than CUs — the burst kernel.  VH_TUNE_DECODE_VARIANT = 1 forces the burst kernel (A/B); the other ring
    // shapes round 2 measured (8x3, 16x1, 12x2, 16x2x16-key, 8x4x16-key; profiles/r2_attn_ab.log) were slower and are gone.
    const int variant = vh_tuning(VH_TUNE_DECODE_VARIANT);
    const int nw = vh_tuning(VH_TUNE_DECODE_WAVES);
    // key splits: combined by a second launch (default) or by the last workgroup to arrive (VH_TUNE_DECODE_COMBINE = 1).
    // Measured on generate() at the reference defaults (4 beams, 8 splits, 12L/512d; profiles/r4_ab_default_generate_combine.log):
    // second launch 364.6 us per step; in-launch with an agent-scope release + acquire 609.7 (the release writes back the
    // L2); in-launch with write-through stores and agent-scope loads, no fences, 377.6 — the hand-off (drain, ticket,
    // dependent loads from beyond the L2) still costs 1 us per layer more than the 1.75 us dependent-launch floor it saves.
    // the split records are combined by the last arriver of a (row, head) inside this launch (knob 1), or by a second launch
    // (knob 2); default: inside at TWO splits (configs[4], 8 rows x 16 heads: 1483.5 vs 1496.6 us per step at context 2.7 k, 1027.2 vs
    // 1028.3 at 0.6 k, profiles/r6_ab_config5_combine.log), the second launch at more (4 beams x 8 heads, 8 splits: DESIGN 3.1)
    const int combine_knob = vh_tuning(VH_TUNE_DECODE_COMBINE);
    const bool fused_combine = n_split > 1 && (combine_knob == 1 || (combine_knob == 0 && n_split == 2));
    unsigned* arrived = fused_combine ? (unsigned*)((char*)partial + decode_records_bytes(B, n_heads, n_split)) : nullptr;
    if (variant != 1 && big && n_split == 1 && nw == 0) {
        const int waves = 8;
        hipExtLaunchKernelGGL((attn_decode_ring_kernel<8, 2>), grid, dim3(waves * 64), 0, s, g_attn_ev[0], g_attn_ev[1], 0, q,
                              ldq, kcache, vcache, out, ldo, cache_len, len_bias, n_heads, S_max, n_split, (float*)partial);
    } else {
        const int waves = nw ? nw : (big ? 16 : 4);
        if (waves == 16) AD(attn_decode_kernel, 16, true); else if (waves == 8) AD(attn_decode_kernel, 8, true);
        else AD(attn_decode_kernel, 4, true);
    }
#undef AD
    if (n_split > 1 && !fused_combine)
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
    float* lse2;   // optional (B, h, Tq): log2-sum-exp of the scaled scores, for the backward kernels
    int64_t mask_bstride;   // explicit mode: elements between the masks of consecutive batch rows (0: one (Tq,Tk) mask for all)
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
    // Keys beyond Tk are clamped to the last written row (rows < Tk were all written by this forward's QKV GEMM):
    // their scores are masked to -inf, so their (finite) values meet p = 0.  Address = wave-uniform head base +
    // one 32-bit per-lane byte offset: no 64-bit vector arithmetic in the loop (VALU slots are MFMA slots).
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t off = (uint32_t)(min(k0 + skey + 16 * i, a.Tk - 1) * HD + squad) * 4u;
            rk[i] = ld4((const float*)((const char*)kb + off));
            rv[i] = ld4((const float*)((const char*)vb + off));
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
            for (int t = 0; t < 8; ++t) {
                const f32x4 kf = ld4(ks + r * KLD + 8 * t + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j)        // the first product starts from the constant-zero C operand
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[t][j], (t | j) ? S : f32x16{}, 0, 0, 0);
            }
            // ---- mask: reg e ↔ key k0 + (e&3) + 8(e>>2) + 4h ; lane ↔ query qi
            if (!all) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    bool vis;
                    if (a.mode == VH_MASK_EXPLICIT) {
                        vis = key < a.Tk && qi < a.Tq && !a.mask[b * a.mask_bstride + (int64_t)qi * a.Tk + key] &&
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
            const float alpha = vh_exp2(m - m_use);
            // subtract and sum two scores per instruction (v_pk_add_f32); the exponentials stay scalar
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 psum2 = {0.f, 0.f};
            const f32x2 nm2 = {-m_use, -m_use};
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                f32x2 d = f32x2{S[e], S[e + 1]} + nm2;
                d.x = vh_exp2(d.x);
                d.y = vh_exp2(d.y);
                psum2 += d;
                S[e] = d.x;
                S[e + 1] = d.y;
            }
            const float psum = psum2.x + psum2.y;
            l = l * alpha + psum;
            m = m_new;
            if (__any(alpha != 1.0f)) {                // no query of this wave raised its maximum: nothing to rescale
#pragma unroll
                for (int e = 0; e < 16; ++e) { O0[e] *= alpha; O1[e] *= alpha; }
            }
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
    if (a.lse2 && h == 0 && qi < a.Tq) a.lse2[(int64_t)bh * a.Tq + qi] = m + log2f(ltot);
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

static int attn_rows_launch(const float* q, int ldq, const float* kcache, const float* vcache,
                            float* out, int ldo, int B, int n_heads, int Tq, int Tk, int S_max,
                            int mode, int x_len, const int32_t* x_len_dev, const int32_t* kv_len,
                            const uint8_t* mask, const uint8_t* pad, float* lse2, void* stream, int64_t mask_bstride = 0) {
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
               x_len_dev, kv_len, mask, pad, nqb, lse2, mask_bstride};
    dim3 grid(nqb * B * n_heads);
    hipLaunchKernelGGL(attn_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    VH_CHECK_LAUNCH("vh_attn_rows");
    return VH_OK;
}

extern "C" int vh_attn_rows(const float* q, int ldq, const float* kcache, const float* vcache,
                            float* out, int ldo, int B, int n_heads, int Tq, int Tk, int S_max,
                            int mode, int x_len, const int32_t* x_len_dev, const int32_t* kv_len,
                            const uint8_t* mask, const uint8_t* pad, void* stream) {
    return attn_rows_launch(q, ldq, kcache, vcache, out, ldo, B, n_heads, Tq, Tk, S_max, mode, x_len,
                            x_len_dev, kv_len, mask, pad, nullptr, stream);
}

extern "C" int vh_attn_rows_bmask(const float* q, int ldq, const float* kcache, const float* vcache, float* out, int ldo, int B,
                                  int n_heads, int Tq, int Tk, int S_max, const uint8_t* mask, int64_t mask_batch_stride,
                                  const uint8_t* pad, void* stream) {
    VH_REQUIRE(mask && (mask_batch_stride == 0 || mask_batch_stride >= (int64_t)Tq * Tk), VH_EINVAL,
               "vh_attn_rows_bmask: a (B,Tq,Tk) mask with batch stride >= Tq*Tk (or 0: one mask for every row), got %lld",
               (long long)mask_batch_stride);
    return attn_rows_launch(q, ldq, kcache, vcache, out, ldo, B, n_heads, Tq, Tk, S_max, VH_MASK_EXPLICIT, 0, nullptr, nullptr,
                            mask, pad, nullptr, stream, mask_batch_stride);
}

extern "C" int vh_attn_rows_lse(const float* q, int ldq, const float* kcache, const float* vcache,
                                float* out, int ldo, int B, int n_heads, int Tq, int Tk, int S_max,
                                int mode, int x_len, const int32_t* x_len_dev, const int32_t* kv_len,
                                const uint8_t* mask, const uint8_t* pad, float* lse2, void* stream) {
    VH_REQUIRE(lse2, VH_EINVAL, "vh_attn_rows_lse: null lse2");
    return attn_rows_launch(q, ldq, kcache, vcache, out, ldo, B, n_heads, Tq, Tk, S_max, mode, x_len,
                            x_len_dev, kv_len, mask, pad, lse2, stream);
}

// =============================================================================================
// many-row attention, backward (training: Tq == Tk).  Two launches, no atomics, P never materialised:
//   dq kernel : one workgroup per 128 queries of a (b,head), queries are lanes (the forward's layout);
//               per 32-key tile   Sᵀ = K·Qᵀ,  P = exp2(Sᵀ − lse),  dPᵀ = V·dOᵀ,  dSᵀ = P∘(dPᵀ − D),
//               dQᵀ += Kᵀ·dSᵀ   (D[q] = Σ_d dO[q][d]·O[q][d], lane-local, also written out for the next kernel)
//   dkv kernel: one workgroup per 128 keys, keys are lanes; per 32-query tile
//               S = Q·Kᵀ, P, dP = dO·Vᵀ, dS as above (lse, D per register now),  dVᵀ += dOᵀ·P,  dKᵀ += Qᵀ·dS
// In both, the accumulator of the first product IS the B operand of the last (no shuffle, no LDS trip).
// =============================================================================================
struct BwdArgs {
    const float* q; int ldq;          // (B*T, ldq), head h at columns h*64
    const float* kc; const float* vc; // (B, h, S_max, 64)
    const float* o; int ldo;          // forward output (B*T, ldo)
    const float* dout; int lddo;      // its gradient
    const float* lse2;                // (B, h, T) from vh_attn_rows_lse
    float* dsum;                      // (B, h, T): D, written by the dq kernel, read by the dkv kernel
    float* dq; float* dk; float* dv; int ldg;   // gradients, each (B*T, ldg) with head h at columns h*64
    int n_heads, T, S_max, mode, x_len;
    const int32_t* x_len_dev; const int32_t* kv_len;
    const uint8_t* mask; const uint8_t* pad;
    int n_blocks;
    float* slab;                      // fused form: dQ partial sums, (n_blocks, B*h, T, 64); null = one key chunk, dq written directly
    int chunk_keys;                   // fused form: keys per workgroup (a multiple of 32, <= 256)
};

__device__ __forceinline__ bool bwd_visible(const BwdArgs& a, int b, int qi, int key, int kvl, int xl) {
    if (a.mode == VH_MASK_EXPLICIT)
        return key < a.T && qi < a.T && !a.mask[(int64_t)qi * a.T + key] && !(a.pad && a.pad[(int64_t)b * a.T + key]);
    bool vis = key < kvl;
    if (a.mode == VH_MASK_PREFIX) vis = vis && (key < xl || (qi >= xl && key <= qi));
    return vis;
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * KT * KLD];   // [buf][K|V][key][KLD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_bh = gridDim.x / a.n_blocks;
    const int bh = blockIdx.x % n_bh, b = bh / a.n_heads, head = bh - b * a.n_heads;
    const int q0 = (a.n_blocks - 1 - (int)(blockIdx.x / n_bh)) * QB;      // heaviest query blocks first
    const int kvl = a.kv_len ? min(a.kv_len[b], a.T) : a.T;
    const int xl = a.x_len_dev ? a.x_len_dev[b] : a.x_len;
    const int last_pos = min(q0 + QB, a.T) - 1;
    int kmax = kvl;
    if (a.mode == VH_MASK_PREFIX) kmax = min(kvl, max(xl, last_pos >= xl ? last_pos + 1 : 0));
    if (a.mode == VH_MASK_EXPLICIT) kmax = a.T;
    const int n_tiles = (kmax + KT - 1) / KT;

    const int qi = q0 + w * 32 + r;
    const bool qin = qi < a.T;
    const float qscale = 0.125f * LOG2E;
    f32x4 qf[8], df[8];
    float dsum = 0.f;
    {
        const int64_t row = (int64_t)b * a.T + min(qi, a.T - 1);
        const float* qp = a.q + row * a.ldq + head * HD + 4 * h;
        const float* dp = a.dout + row * a.lddo + head * HD + 4 * h;
        const float* op = a.o + row * a.ldo + head * HD + 4 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            qf[t] = ld4(qp + 8 * t) * qscale;
            df[t] = ld4(dp + 8 * t);
            const f32x4 ov = ld4(op + 8 * t);
            dsum += (df[t].x * ov.x + df[t].y * ov.y) + (df[t].z * ov.z + df[t].w * ov.w);
        }
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float lse = a.lse2[(int64_t)bh * a.T + min(qi, a.T - 1)];
    if (h == 0 && qin) a.dsum[(int64_t)bh * a.T + qi] = dsum;

    const float* kb = a.kc + (int64_t)bh * a.S_max * HD;
    const float* vb = a.vc + (int64_t)bh * a.S_max * HD;
    const int skey = tid >> 4, squad = (tid & 15) * 4;
    f32x4 rk[2], rv[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                          // keys clamped (masked anyway); uniform base + 32-bit lane offset
            const uint32_t off = (uint32_t)(min(k0 + skey + 16 * i, a.T - 1) * HD + squad) * 4u;
            rk[i] = ld4((const float*)((const char*)kb + off));
            rv[i] = ld4((const float*)((const char*)vb + off));
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
    f32x16 G0, G1;                      // dQᵀ halves: d = r and r + 32
#pragma unroll
    for (int e = 0; e < 16; ++e) { G0[e] = 0.f; G1[e] = 0.f; }
    const int wpos_min = q0 + w * 32, wpos_max = wpos_min + 31;
    if (n_tiles > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int cur = kt & 1, k0 = kt * KT;
        if (kt + 1 < n_tiles) gload(k0 + KT);
        bool any = k0 < kvl;
        if (a.mode == VH_MASK_PREFIX) any = any && (k0 < xl || (wpos_max >= xl && k0 <= wpos_max));
        else if (a.mode == VH_MASK_EXPLICIT) any = true;
        if (any) {
            const float* ks = lds + cur * (2 * KT * KLD);
            const float* vs = ks + KT * KLD;
            f32x16 S, P;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 kf = ld4(ks + r * KLD + 8 * t + 4 * h);
                const f32x4 vf = ld4(vs + r * KLD + 8 * t + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {      // the first products start from the constant-zero C operand
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[t][j], (t | j) ? S : f32x16{}, 0, 0, 0);   // Sᵀ[key][q]
                    P = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j], df[t][j], (t | j) ? P : f32x16{}, 0, 0, 0);   // dPᵀ[key][q]
                }
            }
            // whole tile visible to every query of this wave: no per-element mask (vector compares and selects
            // are MFMA issue slots)
            bool all = wpos_max < a.T && k0 + KT <= kvl;
            if (a.mode == VH_MASK_PREFIX) all = all && (k0 + KT <= xl || (wpos_min >= xl && k0 + KT - 1 <= wpos_min));
            else if (a.mode == VH_MASK_EXPLICIT) all = false;
            if (all) {
#pragma unroll
                for (int e = 0; e < 16; ++e) S[e] = vh_exp2(S[e] - lse) * (P[e] - dsum);   // dSᵀ[key][q]
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool vis = qin && bwd_visible(a, b, qi, key, kvl, xl);
                    const float p = vis ? vh_exp2(S[e] - lse) : 0.f;
                    S[e] = p * (P[e] - dsum);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float* krow = ks + ((e & 3) + 8 * (e >> 2) + 4 * h) * KLD + r;
                G0 = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[0], S[e], G0, 0, 0, 0);
                G1 = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[32], S[e], G1, 0, 0, 0);
            }
        }
        if (kt + 1 < n_tiles) lstore(cur ^ 1);
        __syncthreads();
    }
    float* ob = lds;                // [128][KLD] transpose buffer
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v0 = {G0[4 * g4], G0[4 * g4 + 1], G0[4 * g4 + 2], G0[4 * g4 + 3]};
        f32x4 v1 = {G1[4 * g4], G1[4 * g4 + 1], G1[4 * g4 + 2], G1[4 * g4 + 3]};
        st4(ob + (w * 32 + r) * KLD + 8 * g4 + 4 * h, v0 * 0.125f);
        st4(ob + (w * 32 + r) * KLD + 32 + 8 * g4 + 4 * h, v1 * 0.125f);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = skey + 16 * i;
        if (q0 + row < a.T)
            st4(a.dq + ((int64_t)b * a.T + q0 + row) * a.ldg + head * HD + squad, ld4(ob + row * KLD + squad));
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(BwdArgs a) {
    // [buf][Q|dO][query][KLD] + [buf][lse|D][32]
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * KT * KLD];
    __shared__ __attribute__((aligned(16))) float stat[2][2][KT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_bh = gridDim.x / a.n_blocks;
    const int bh = blockIdx.x % n_bh, b = bh / a.n_heads, head = bh - b * a.n_heads;
    const int kb0 = (int)(blockIdx.x / n_bh) * QB;                          // first key of this block
    const int kvl = a.kv_len ? min(a.kv_len[b], a.T) : a.T;
    const int xl = a.x_len_dev ? a.x_len_dev[b] : a.x_len;
    // queries that can see a key of this block: all of them for keys < xl (or FULL / EXPLICIT), else q >= key
    int qmin = 0;
    if (a.mode == VH_MASK_PREFIX && kb0 >= xl) qmin = kb0;
    const int t_first = qmin / KT, n_tiles = (a.T + KT - 1) / KT;

    const int kj = kb0 + w * 32 + r;            // this lane's key
    const bool kin = kj < a.T;
    const float qscale = 0.125f * LOG2E;
    f32x4 kf[8], vf[8];
    {
        const float* kp = a.kc + ((int64_t)bh * a.S_max + min(kj, a.T - 1)) * HD + 4 * h;
        const float* vp = a.vc + ((int64_t)bh * a.S_max + min(kj, a.T - 1)) * HD + 4 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            kf[t] = ld4(kp + 8 * t) * qscale;
            vf[t] = ld4(vp + 8 * t);
        }
    }
    const int sq = tid >> 4, squad = (tid & 15) * 4;   // staging: 2 query rows per thread per operand
    f32x4 rq[2], rd[2];
    float rstat = 0.f;
    const float* qbase = a.q + (int64_t)b * a.T * a.ldq + head * HD;
    const float* dbase = a.dout + (int64_t)b * a.T * a.lddo + head * HD;
    auto gload = [&](int q0t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                          // uniform (batch row, head) base + 32-bit lane offset
            const int qrow = min(q0t + sq + 16 * i, a.T - 1);
            rq[i] = ld4((const float*)((const char*)qbase + (uint32_t)(qrow * a.ldq + squad) * 4u));
            rd[i] = ld4((const float*)((const char*)dbase + (uint32_t)(qrow * a.lddo + squad) * 4u));
        }
        if (tid < 2 * KT) {
            const int qq = min(q0t + (tid & 31), a.T - 1);
            rstat = (tid < KT ? a.lse2 : a.dsum)[(int64_t)bh * a.T + qq];
        }
    };
    auto lstore = [&](int buf) {
        float* qd = lds + buf * (2 * KT * KLD);
        float* dd = qd + KT * KLD;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            st4(qd + (sq + 16 * i) * KLD + squad, rq[i]);
            st4(dd + (sq + 16 * i) * KLD + squad, rd[i]);
        }
        if (tid < 2 * KT) stat[buf][tid >> 5][tid & 31] = rstat;
    };
    f32x16 GK0, GK1, GV0, GV1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { GK0[e] = 0.f; GK1[e] = 0.f; GV0[e] = 0.f; GV1[e] = 0.f; }
    const int wk_min = kb0 + w * 32;
    if (t_first < n_tiles) {
        gload(t_first * KT);
        lstore(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): K / V fragments landed on every path (see attn_bwd_fused_kernel)
    __syncthreads();
    for (int qt = t_first; qt < n_tiles; ++qt) {
        const int cur = (qt - t_first) & 1, q0t = qt * KT;
        if (qt + 1 < n_tiles) gload(q0t + KT);
        bool any = wk_min < kvl;
        if (a.mode == VH_MASK_PREFIX) any = any && (wk_min < xl || q0t + KT - 1 >= wk_min);
        else if (a.mode == VH_MASK_EXPLICIT) any = true;
        if (any) {
            const float* qs = lds + cur * (2 * KT * KLD);
            const float* ds = qs + KT * KLD;
            f32x16 S, P;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 qv = ld4(qs + r * KLD + 8 * t + 4 * h);
                const f32x4 dv = ld4(ds + r * KLD + 8 * t + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[j], kf[t][j], (t | j) ? S : f32x16{}, 0, 0, 0);   // S[q][key]
                    P = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[j], vf[t][j], (t | j) ? P : f32x16{}, 0, 0, 0);   // dP[q][key]
                }
            }
            // every (query of the tile, key of this wave) pair visible: no per-element mask
            const int wk_max = wk_min + 31;
            bool all = wk_max < kvl && q0t + KT <= a.T;
            if (a.mode == VH_MASK_PREFIX) all = all && (wk_max < xl || (q0t >= xl && wk_max <= q0t));
            else if (a.mode == VH_MASK_EXPLICIT) all = false;
            if (all) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 lse4 = ld4(&stat[cur][0][8 * g4 + 4 * h]);
                    const f32x4 d4 = ld4(&stat[cur][1][8 * g4 + 4 * h]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g4 + j;
                        const float p = vh_exp2(S[e] - lse4[j]);
                        S[e] = p;                                   // P[q][key]
                        P[e] = p * (P[e] - d4[j]);                  // dS[q][key]
                    }
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 lse4 = ld4(&stat[cur][0][8 * g4 + 4 * h]);
                    const f32x4 d4 = ld4(&stat[cur][1][8 * g4 + 4 * h]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g4 + j;
                        const int qi = q0t + 8 * g4 + 4 * h + j;
                        const bool vis = kin && qi < a.T && bwd_visible(a, b, qi, kj, kvl, xl);
                        const float p = vis ? vh_exp2(S[e] - lse4[j]) : 0.f;
                        S[e] = p;
                        P[e] = p * (P[e] - d4[j]);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qrow = (e & 3) + 8 * (e >> 2) + 4 * h;
                const float* dorow = ds + qrow * KLD + r;
                const float* qrowp = qs + qrow * KLD + r;
                GV0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow[0], S[e], GV0, 0, 0, 0);    // dVᵀ[d][key]
                GV1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow[32], S[e], GV1, 0, 0, 0);
                GK0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qrowp[0], P[e], GK0, 0, 0, 0);    // dKᵀ[d][key]
                GK1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qrowp[32], P[e], GK1, 0, 0, 0);
            }
        }
        if (qt + 1 < n_tiles) lstore(cur ^ 1);
        __syncthreads();
    }
    float* ob = lds;
    for (int which = 0; which < 2; ++which) {
        const f32x16& A0 = which ? GV0 : GK0;
        const f32x16& A1 = which ? GV1 : GK1;
        const float sc = which ? 1.0f : 0.125f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 v0 = {A0[4 * g4], A0[4 * g4 + 1], A0[4 * g4 + 2], A0[4 * g4 + 3]};
            f32x4 v1 = {A1[4 * g4], A1[4 * g4 + 1], A1[4 * g4 + 2], A1[4 * g4 + 3]};
            st4(ob + (w * 32 + r) * KLD + 8 * g4 + 4 * h, v0 * sc);
            st4(ob + (w * 32 + r) * KLD + 32 + 8 * g4 + 4 * h, v1 * sc);
        }
        __syncthreads();
        float* dst = which ? a.dv : a.dk;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = sq + 16 * i;
            if (kb0 + row < a.T)
                st4(dst + ((int64_t)b * a.T + kb0 + row) * a.ldg + head * HD + squad, ld4(ob + row * KLD + squad));
        }
        __syncthreads();
    }
}

// =============================================================================================
// many-row attention, backward, FIVE products in one kernel (round 4).  The keys-as-lanes kernel above already holds
// dS[q][key] in registers; dQ = dS·K contracts over the key, which is the LANE index of those registers, so the
// two-kernel form recomputes S and dP in a second kernel with queries as lanes (7 products).  Here dS makes one trip
// through LDS instead: a workgroup of 8 waves owns a chunk of up to 256 keys (K transposed into LDS once per
// workgroup), every wave writes its 32 x 32 block of dS into a shared [32 q][256 keys] tile, and after a barrier each
// wave computes ONE 16 x 16 block of the tile's dQ (4 d-blocks x 2 q-blocks = 8 waves) over ALL the chunk's keys on
// mfma_f32_16x16x4 — no cross-wave reduction.  Both operands are read with ds_read_b128: four consecutive keys per
// lane (the contraction index of four consecutive MFMAs is permuted accordingly: key(u, g, j) = 16u + 4g + j), rows
// XOR-swizzled by (row & 15) in 16-byte slots, which is conflict-free for the instruction's four 16-lane groups.
// The dQ partial of a (query tile, key chunk) pair goes to slab `chunk` — (n_chunks, B*h, T, 64), T/256 = 4 slabs at the
// training shape — and attn_dq_reduce_kernel adds the slabs in chunk order: bitwise reproducible, no atomics.
// D = rowsum(dO∘O) is computed while the dO / O rows are staged (the dq kernel used to do it).
// =============================================================================================
#define FKMAX 256       // keys per workgroup at most (8 waves x 32)

// FW waves per workgroup = FW x 32 keys per chunk at most.  FW = 8: one workgroup per CU (113 KB of LDS); FW = 4: two per
// CU (65 KB each) — the same eight waves per CU in half-size work items.  The Q / dO tile is single-buffered: every read
// of tile i happens before the tile's first barrier and the staged rows of tile i + 1 are stored after it, so the two
// barriers the dS hand-over needs anyway also order the tile buffer.
template <int FW>
__global__ __launch_bounds__(FW * 64, FW == 8 ? 1 : 2) void attn_bwd_fused_kernel(BwdArgs a) {
    constexpr int KEYS = FW * 32;                    // row length of Kt and dS
    constexpr int RPT = 8 / FW;                      // staged tile rows per thread (32 rows x 16 threads over FW x 64 threads)
    constexpr int QB_W = 8 / FW;                     // 16-query blocks of the dQ tile per wave (FW = 4: both, one d-block per wave)
    // [Q|dO][query][KLD] | [lse|D][32] | Kt [64 d][KEYS] | dS [32 q][KEYS]
    __shared__ __attribute__((aligned(16))) float lds[2 * KT * KLD + 2 * KT + HD * KEYS + KT * KEYS];
    float* const stat = lds + 2 * KT * KLD;
    float* const kt = stat + 2 * KT;
    float* const dsb = kt + HD * KEYS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_bh = gridDim.x / a.n_blocks;
    const int bh = blockIdx.x % n_bh, b = bh / a.n_heads, head = bh - b * a.n_heads;
    const int chunk = (int)(blockIdx.x / n_bh);                 // chunk 0 first: under the prefix mask it sees every query
    const int kb0 = chunk * a.chunk_keys;
    const int kend = min(kb0 + a.chunk_keys, a.T);              // keys [kb0, kend) belong to this workgroup
    const int kvl = a.kv_len ? min(a.kv_len[b], a.T) : a.T;
    const int xl = a.x_len_dev ? a.x_len_dev[b] : a.x_len;
    const int sq = tid >> 4, squad = (tid & 15) * 4;            // staging: row sq (+ 16 with four waves), one 16-byte chunk

    if (a.mode != VH_MASK_EXPLICIT && kb0 >= kvl) {
        // every key of the chunk is padding: dK = dV = 0, no dQ contribution (the reduce kernel skips this chunk too)
        for (int row = kb0 + sq; row < kend; row += FW * 4) {
            const int64_t o = ((int64_t)b * a.T + row) * a.ldg + head * HD + squad;
            st4(a.dk + o, f32x4{0.f, 0.f, 0.f, 0.f});
            st4(a.dv + o, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        // a single chunk writes dQ itself (no slab, no reduce launch): a row without any valid key (kv_len = 0, T <= 256)
        // must leave zeros there, not the caller's uninitialised buffer (round-4 advisor finding)
        if (a.slab == nullptr)
            for (int row = sq; row < a.T; row += FW * 4)
                st4(a.dq + ((int64_t)b * a.T + row) * a.ldg + head * HD + squad, f32x4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    int qmin = 0;
    if (a.mode == VH_MASK_PREFIX && kb0 >= xl) qmin = kb0;
    const int t_first = qmin / KT, n_tiles = (a.T + KT - 1) / KT;

    const int wk_min = kb0 + w * 32, wk_max = wk_min + 31;
    const int kj = wk_min + r;                   // this lane's key
    const bool kin = kj < kend;
    const float qscale = 0.125f * LOG2E;
    f32x4 kf[8], vf[8];
    {
        const int64_t krow = (int64_t)bh * a.S_max + min(kj, a.T - 1);
        const float* kp = a.kc + krow * HD + 4 * h;
        const float* vp = a.vc + krow * HD + 4 * h;
        const int kl = w * 32 + r;               // key inside the chunk
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            kf[t] = ld4(kp + 8 * t);
            vf[t] = ld4(vp + 8 * t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {        // Kt[d][key], 16-byte slot (key >> 2) XOR-swizzled by d & 15
                const int d = 8 * t + 4 * h + j;
                kt[d * KEYS + ((((kl >> 2) ^ (d & 15)) << 2) | (kl & 3))] = kf[t][j];
            }
            kf[t] = kf[t] * qscale;
        }
    }
    f32x4 rq[RPT], rd[RPT], ro[RPT];
    float rlse = 0.f;
    const float* qbase = a.q + (int64_t)b * a.T * a.ldq + head * HD;
    const float* dbase = a.dout + (int64_t)b * a.T * a.lddo + head * HD;
    const float* obase = a.o + (int64_t)b * a.T * a.ldo + head * HD;
    // loads only: nothing here may wait for them.  Q and dO rows are requested at the top of a tile and land under its
    // MFMAs; the O row (needed only for D) and lse are requested after the tile's first barrier, when the S / dP
    // registers are dead (the kernel sits at the register limit), and land under the dQ product.
    // (four waves stage two rows per thread: the second row's Q / dO are requested late as well — eight registers less
    // across the S / dP / dV / dK phase, where the count peaks)
    auto gload = [&](int q0t) {
        const int qrow = min(q0t + sq, a.T - 1);
        rq[0] = ld4((const float*)((const char*)qbase + (uint32_t)(qrow * a.ldq + squad) * 4u));
        rd[0] = ld4((const float*)((const char*)dbase + (uint32_t)(qrow * a.lddo + squad) * 4u));
    };
    auto gload_late = [&](int q0t) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int qrow = min(q0t + sq + 16 * i, a.T - 1);
            if (i > 0) {
                rq[i] = ld4((const float*)((const char*)qbase + (uint32_t)(qrow * a.ldq + squad) * 4u));
                rd[i] = ld4((const float*)((const char*)dbase + (uint32_t)(qrow * a.lddo + squad) * 4u));
            }
            ro[i] = ld4((const float*)((const char*)obase + (uint32_t)(qrow * a.ldo + squad) * 4u));
        }
        if (tid < KT) rlse = a.lse2[(int64_t)bh * a.T + min(q0t + tid, a.T - 1)];
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            st4(lds + (sq + 16 * i) * KLD + squad, rq[i]);
            st4(lds + KT * KLD + (sq + 16 * i) * KLD + squad, rd[i]);
            const float dsum = row16_sum((rd[i].x * ro[i].x + rd[i].y * ro[i].y) + (rd[i].z * ro[i].z + rd[i].w * ro[i].w));
            if ((tid & 15) == 0) stat[KT + sq + 16 * i] = dsum;                  // D of the staged row
        }
        if (tid < KT) stat[tid] = rlse;
    };
    f32x16 GK0, GK1, GV0, GV1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { GK0[e] = 0.f; GK1[e] = 0.f; GV0[e] = 0.f; GV1[e] = 0.f; }
    // the dQ block(s) of this wave: d0 = 16 (w & 3); q0 = 16 (w >> 2) with eight waves, both q-blocks with four;
    // lane (i, g) reads row i of both operands
    const int mi = lane & 15, mg = lane >> 4;
    const float* ap = kt + (16 * (w & 3) + mi) * KEYS;
    const float* bp = dsb + (FW == 8 ? 16 * (w >> 2) + mi : mi) * KEYS;
    const int n_waves_keys = (kend - kb0 + 31) >> 5;            // waves that own at least one key of the chunk
    if (t_first < n_tiles) {
        gload(t_first * KT);
        gload_late(t_first * KT);
        lstore();
    }
    // vmcnt(0), unconditionally: the K / V fragments are first USED inside the loop and are still outstanding on the path
    // around the conditional above — without this the wait insertion puts vmcnt(3..0) in front of the first score MFMAs
    // of EVERY tile, which makes the tile wait for its own prefetch of the next Q / dO rows (seen in the ISA, round 6)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int qt = t_first; qt < n_tiles; ++qt) {
        const int q0t = qt * KT;
        if (qt + 1 < n_tiles) gload(q0t + KT);
        // waves of this workgroup that see a query of the tile: a prefix of the waves (every term falls with the key index)
        int n_act = n_waves_keys;
        if (a.mode != VH_MASK_EXPLICIT) {
            n_act = min(n_act, (kvl - kb0 + 31) >> 5);
            if (a.mode == VH_MASK_PREFIX) {
                // wave v is active iff kb0 + 32 v < xl  or  kb0 + 32 v <= q0t + 31
                const int lim = max(xl - 1, q0t + KT - 1) - kb0;   // largest admissible 32 v
                n_act = min(n_act, lim < 0 ? 0 : (lim >> 5) + 1);
            }
        }
        const bool any = w < n_act;
        const float* qs = lds;
        const float* ds = qs + KT * KLD;
        const float* st = stat;
        if (any) {
            f32x16 S, P;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 qv = ld4(qs + r * KLD + 8 * t + 4 * h);
                const f32x4 dv = ld4(ds + r * KLD + 8 * t + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[j], kf[t][j], (t | j) ? S : f32x16{}, 0, 0, 0);   // S[q][key]
                    P = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[j], vf[t][j], (t | j) ? P : f32x16{}, 0, 0, 0);   // dP[q][key]
                }
            }
            bool all = wk_max < min(kvl, kend) && q0t + KT <= a.T;
            if (a.mode == VH_MASK_PREFIX) all = all && (wk_max < xl || (q0t >= xl && wk_max <= q0t));
            else if (a.mode == VH_MASK_EXPLICIT) all = false;
            if (all) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 lse4 = ld4(st + 8 * g4 + 4 * h);
                    const f32x4 d4 = ld4(st + KT + 8 * g4 + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g4 + j;
                        const float p = vh_exp2(S[e] - lse4[j]);
                        S[e] = p;                                   // P[q][key]
                        P[e] = p * (P[e] - d4[j]);                  // dS[q][key]
                    }
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 lse4 = ld4(st + 8 * g4 + 4 * h);
                    const f32x4 d4 = ld4(st + KT + 8 * g4 + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g4 + j;
                        const int qi = q0t + 8 * g4 + 4 * h + j;
                        const bool vis = kin && qi < a.T && bwd_visible(a, b, qi, kj, kvl, xl);
                        const float p = vis ? vh_exp2(S[e] - lse4[j]) : 0.f;
                        S[e] = p;
                        P[e] = p * (P[e] - d4[j]);
                    }
                }
            }
            // dS → the shared tile, [q][key] with the slot swizzle: one ds_write_b32 per register, 32 consecutive keys per half-wave
            {
                const int kl = w * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int qrow = (e & 3) + 8 * (e >> 2) + 4 * h;
                    dsb[qrow * KEYS + ((((kl >> 2) ^ (qrow & 15)) << 2) | (kl & 3))] = P[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qrow = (e & 3) + 8 * (e >> 2) + 4 * h;
                const float* dorow = ds + qrow * KLD + r;
                const float* qrowp = qs + qrow * KLD + r;
                GV0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow[0], S[e], GV0, 0, 0, 0);    // dVᵀ[d][key]
                GV1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow[32], S[e], GV1, 0, 0, 0);
                GK0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qrowp[0], P[e], GK0, 0, 0, 0);    // dKᵀ[d][key]
                GK1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qrowp[32], P[e], GK1, 0, 0, 0);
            }
        }
        __syncthreads();                     // the dS blocks of the n_act active waves are in LDS; the Q / dO tile is free
        if (qt + 1 < n_tiles) gload_late(q0t + KT);
        {
            // dQ[q][d] block = Σ_key dS[q][key]·K[key][d] over the active waves' keys: A = Kᵀ rows (d), B = dS rows (q)
            f32x4 c[QB_W][4];
#pragma unroll
            for (int qb = 0; qb < QB_W; ++qb)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[qb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int nu = 2 * n_act;                    // 16-key groups of the active waves (n_act >= 1 on every visited tile)
            f32x4 a4 = ld4(ap + ((mg ^ mi) << 2)), b4[QB_W];
#pragma unroll
            for (int qb = 0; qb < QB_W; ++qb) b4[qb] = ld4(bp + qb * 16 * KEYS + ((mg ^ mi) << 2));
            for (int u = 0; u < nu; ++u) {               // operands of group u + 1 requested BEFORE group u's MFMAs
                const int slot = ((4 * min(u + 1, nu - 1) + mg) ^ mi) << 2;
                const f32x4 an = ld4(ap + slot);
                f32x4 bn[QB_W];
#pragma unroll
                for (int qb = 0; qb < QB_W; ++qb) bn[qb] = ld4(bp + qb * 16 * KEYS + slot);
                __builtin_amdgcn_sched_barrier(0);       // (left alone the compiler sinks the reads below the MFMAs)
#pragma unroll
                for (int qb = 0; qb < QB_W; ++qb)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        c[qb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[qb][j], c[qb][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a4 = an;
#pragma unroll
                for (int qb = 0; qb < QB_W; ++qb) b4[qb] = bn[qb];
            }
            const uint32_t dcol = 16 * (w & 3) + 4 * mg;
#pragma unroll
            for (int qb = 0; qb < QB_W; ++qb) {
                const f32x4 cc = (c[qb][0] + c[qb][1]) + (c[qb][2] + c[qb][3]);   // lane (q = mi, d = 4 mg + e)
                const int qi = q0t + (FW == 8 ? 16 * (w >> 2) : 16 * qb) + mi;
                if (qi < a.T) {                          // wave-uniform 64-bit base + one 32-bit lane offset
                    if (a.slab) st4((float*)((char*)(a.slab + ((int64_t)chunk * n_bh + bh) * a.T * HD) + (uint32_t)(qi * HD + dcol) * 4u), cc);
                    else st4((float*)((char*)(a.dq + (int64_t)b * a.T * a.ldg + head * HD) + (uint32_t)(qi * a.ldg + dcol) * 4u), cc * 0.125f);
                }
            }
        }
        if (qt + 1 < n_tiles) lstore();
        __syncthreads();                     // next tile staged; every wave is done with this tile's dS
    }
    // ---- dK, dV: transposed through LDS (everything above is dead), whole rows out
    float* ob = lds;                         // [KEYS][KLD]
    for (int which = 0; which < 2; ++which) {
        const f32x16& A0 = which ? GV0 : GK0;
        const f32x16& A1 = which ? GV1 : GK1;
        const float sc = which ? 1.0f : 0.125f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 v0 = {A0[4 * g4], A0[4 * g4 + 1], A0[4 * g4 + 2], A0[4 * g4 + 3]};
            f32x4 v1 = {A1[4 * g4], A1[4 * g4 + 1], A1[4 * g4 + 2], A1[4 * g4 + 3]};
            st4(ob + (w * 32 + r) * KLD + 8 * g4 + 4 * h, v0 * sc);
            st4(ob + (w * 32 + r) * KLD + 32 + 8 * g4 + 4 * h, v1 * sc);
        }
        __syncthreads();
        float* dst = which ? a.dv : a.dk;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = sq + FW * 4 * i;
            if (kb0 + row < kend)
                st4(dst + ((int64_t)b * a.T + kb0 + row) * a.ldg + head * HD + squad, ld4(ob + row * KLD + squad));
        }
        __syncthreads();
    }
}

// dq[row][head*64 + d] = 1/8 · Σ_chunks slab[chunk][b,head][q][d], chunks in index order; a chunk contributes to a query
// exactly when the fused kernel visited the pair (same rule, so no slab element is read that was not written).
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(BwdArgs a, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    const int d4 = (int)(idx & 15);
    const int64_t rh = idx >> 4;
    const int head = (int)(rh % a.n_heads);
    const int64_t row = rh / a.n_heads;
    const int b = (int)(row / a.T), q = (int)(row - (int64_t)b * a.T);
    const int kvl = a.kv_len ? min(a.kv_len[b], a.T) : a.T;
    const int xl = a.x_len_dev ? a.x_len_dev[b] : a.x_len;
    const int64_t n_bh = (int64_t)(n4 / 16 / a.T / a.n_heads) * a.n_heads;   // B * h
    const int bh = b * a.n_heads + head;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.n_blocks; ++c) {
        const int kb0 = c * a.chunk_keys;
        if (a.mode != VH_MASK_EXPLICIT && kb0 >= kvl) break;
        if (a.mode == VH_MASK_PREFIX && kb0 >= xl && q < kb0) break;
        acc = acc + ld4(a.slab + (((int64_t)c * n_bh + bh) * a.T + q) * HD + 4 * d4);
    }
    st4(a.dq + row * a.ldg + head * HD + 4 * d4, acc * 0.125f);
}

extern "C" int vh_attn_rows_bwd(const float* q, int ldq, const float* kcache, const float* vcache,
                                const float* out, int ldo, const float* dout, int lddo, const float* lse2,
                                float* dsum, float* dq, float* dk, float* dv, int ldg, int B, int n_heads,
                                int T, int S_max, int mode, int x_len, const int32_t* x_len_dev,
                                const int32_t* kv_len, const uint8_t* mask, const uint8_t* pad, void* stream) {
    VH_REQUIRE(q && kcache && vcache && out && dout && lse2 && dsum && dq && dk && dv, VH_EINVAL,
               "vh_attn_rows_bwd: null pointer");
    VH_REQUIRE(B >= 0 && n_heads > 0 && T >= 0 && S_max >= T, VH_EINVAL,
               "vh_attn_rows_bwd: bad dims B=%d h=%d T=%d S_max=%d", B, n_heads, T, S_max);
    VH_REQUIRE(mode == VH_MASK_FULL || mode == VH_MASK_PREFIX || (mode == VH_MASK_EXPLICIT && mask), VH_EINVAL,
               "vh_attn_rows_bwd: mode=%d", mode);
    const int wd = n_heads * HD;
    VH_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0 && ldg % 4 == 0 && ldq >= wd && ldo >= wd &&
                   lddo >= wd && ldg >= wd,
               VH_EINVAL, "vh_attn_rows_bwd: leading dimensions");
    VH_REQUIRE(vh_aligned16(q) && vh_aligned16(kcache) && vh_aligned16(vcache) && vh_aligned16(out) &&
                   vh_aligned16(dout) && vh_aligned16(dq) && vh_aligned16(dk) && vh_aligned16(dv),
               VH_EALIGN, "vh_attn_rows_bwd: pointers must be 16-byte aligned");
    if (B == 0 || T == 0) return VH_OK;
    const int nb = (T + QB - 1) / QB;
    BwdArgs a{q, ldq, kcache, vcache, out, ldo, dout, lddo, lse2, dsum, dq, dk, dv, ldg,
              n_heads, T, S_max, mode, x_len, x_len_dev, kv_len, mask, pad, nb, nullptr, 0};
    dim3 grid(nb * B * n_heads);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    VH_CHECK_LAUNCH("vh_attn_rows_bwd");
    return VH_OK;
}

// ---- the same gradients with a caller-provided workspace: the five-product kernel + slab reduce (default), or the
// two-kernel form above with its D scratch inside the workspace (VH_TUNE_ATTN_BWD = 1)
// Key chunks per (batch row, head): at least ceil(T / 256) (a workgroup holds 256 keys), more when that evens out the
// launch.  Model, calibrated with tools/sweep_attn_bwd_chunks.py (profiles/r4_sweep_attn_bwd_chunks.log): workgroups are
// dealt one per CU in index order (chunk 0 first) to whichever CU frees up; a workgroup's time is its query tiles x
// 1.0 with 5..8 active 32-key waves (two per SIMD) or x 0.6 with <= 4 (one per SIMD: 10.7 vs 6.5 us per tile); under the
// prefix mask chunk c only sees the query tiles from its first key on; every chunk adds a slab of dQ partials
// (B*h*T*512 bytes written and read at ~5 TB/s).  The launch is simulated for every admissible count (chunks of 97..256
// keys) and the cheapest taken; results are cached per shape.
static int bwd_chunks_model(int bh, int T, bool causal) {
    const int nc_min = (T + FKMAX - 1) / FKMAX, nc_max = max(nc_min, (T + 127) / 128);
    int best = nc_min;
    double best_cost = 1e300;
    const int n_tiles = (T + KT - 1) / KT;
    for (int nc = nc_min; nc <= nc_max; ++nc) {
        const int keys = ((T + nc - 1) / nc + 31) / 32 * 32;
        if ((int64_t)keys * (nc - 1) >= T) continue;                     // the last chunk would be empty
        const double unit = keys > 128 ? 1.0 : (causal ? 0.6 : 0.55);   // <= 128 keys: four active waves (full / explicit masks: the 4-wave kernel, two per CU)
        std::priority_queue<double, std::vector<double>, std::greater<double>> cu;
        for (int i = 0; i < 256; ++i) cu.push(0.0);
        double end = 0.0;
        for (int c = 0; c < nc; ++c) {
            const double work = unit * (causal ? n_tiles - c * keys / KT : n_tiles);
            for (int j = 0; j < bh; ++j) {
                const double t = cu.top() + work;
                cu.pop();
                cu.push(t);
                end = t > end ? t : end;
            }
        }
        const double cost = end + (nc > 1 ? 1e-5 * nc * bh * (double)T : 0.0);
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = nc; }
    }
    return best;
}

static int bwd_chunks(int B, int n_heads, int T, bool causal) {
    const int nc_min = (T + FKMAX - 1) / FKMAX;
    const int forced = vh_tuning(VH_TUNE_ATTN_BWD_CHUNKS);
    if (forced > 0) {
        int nc = max(nc_min, min(forced, (T + 31) / 32));
        while (nc > nc_min && (int64_t)(((T + nc - 1) / nc + 31) / 32 * 32) * (nc - 1) >= T) --nc;   // no empty last chunk
        return nc;
    }
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> cache;
    const uint64_t key = ((uint64_t)(B * n_heads) << 33) | ((uint64_t)T << 1) | (causal ? 1u : 0u);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (cache.size() > 4096) cache.clear();
    return cache[key] = bwd_chunks_model(B * n_heads, T, causal);
}

extern "C" int vh_attn_rows_bwd_chunks(int B, int n_heads, int T, int mode) {
    if (B <= 0 || n_heads <= 0 || T <= 0) return 0;
    return bwd_chunks(B, n_heads, T, mode == VH_MASK_PREFIX);
}

extern "C" size_t vh_attn_rows_bwd_ws_bytes(int B, int n_heads, int T) {
    if (B <= 0 || n_heads <= 0 || T <= 0) return 0;
    const int nc = max(bwd_chunks(B, n_heads, T, false), bwd_chunks(B, n_heads, T, true));   // either mask family fits
    const size_t slabs = nc > 1 ? (size_t)nc * B * n_heads * T * HD * sizeof(float) : 0;
    const size_t dsum = ((size_t)B * n_heads * T * sizeof(float) + 15) & ~(size_t)15;
    return slabs > dsum ? slabs : dsum;
}

extern "C" int vh_attn_rows_bwd_ws(const float* q, int ldq, const float* kcache, const float* vcache,
                                   const float* out, int ldo, const float* dout, int lddo, const float* lse2,
                                   float* dq, float* dk, float* dv, int ldg, int B, int n_heads, int T, int S_max,
                                   int mode, int x_len, const int32_t* x_len_dev, const int32_t* kv_len,
                                   const uint8_t* mask, const uint8_t* pad, void* ws, size_t ws_bytes, void* stream) {
    VH_REQUIRE(B < 0 || T < 0 || n_heads <= 0 || ws_bytes >= vh_attn_rows_bwd_ws_bytes(B, n_heads, T), VH_EINVAL,
               "vh_attn_rows_bwd_ws: workspace of %zu bytes, need %zu", ws_bytes, vh_attn_rows_bwd_ws_bytes(B, n_heads, T));
    VH_REQUIRE(ws && vh_aligned16(ws), VH_EALIGN, "vh_attn_rows_bwd_ws: workspace must be a 16-byte aligned device pointer");
    if (vh_tuning(VH_TUNE_ATTN_BWD) == 1)
        return vh_attn_rows_bwd(q, ldq, kcache, vcache, out, ldo, dout, lddo, lse2, (float*)ws, dq, dk, dv, ldg, B,
                                n_heads, T, S_max, mode, x_len, x_len_dev, kv_len, mask, pad, stream);
    VH_REQUIRE(q && kcache && vcache && out && dout && lse2 && dq && dk && dv, VH_EINVAL,
               "vh_attn_rows_bwd_ws: null pointer");
    VH_REQUIRE(B >= 0 && n_heads > 0 && T >= 0 && S_max >= T, VH_EINVAL,
               "vh_attn_rows_bwd_ws: bad dims B=%d h=%d T=%d S_max=%d", B, n_heads, T, S_max);
    VH_REQUIRE(mode == VH_MASK_FULL || mode == VH_MASK_PREFIX || (mode == VH_MASK_EXPLICIT && mask), VH_EINVAL,
               "vh_attn_rows_bwd_ws: mode=%d", mode);
    const int wd = n_heads * HD;
    VH_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0 && ldg % 4 == 0 && ldq >= wd && ldo >= wd &&
                   lddo >= wd && ldg >= wd,
               VH_EINVAL, "vh_attn_rows_bwd_ws: leading dimensions");
    VH_REQUIRE(vh_aligned16(q) && vh_aligned16(kcache) && vh_aligned16(vcache) && vh_aligned16(out) &&
                   vh_aligned16(dout) && vh_aligned16(dq) && vh_aligned16(dk) && vh_aligned16(dv),
               VH_EALIGN, "vh_attn_rows_bwd_ws: pointers must be 16-byte aligned");
    if (B == 0 || T == 0) return VH_OK;
    const int nc = bwd_chunks(B, n_heads, T, mode == VH_MASK_PREFIX);
    const int chunk_keys = ((T + nc - 1) / nc + 31) / 32 * 32;       // equal chunks of whole 32-key wave blocks
    BwdArgs a{q, ldq, kcache, vcache, out, ldo, dout, lddo, lse2, nullptr, dq, dk, dv, ldg,
              n_heads, T, S_max, mode, x_len, x_len_dev, kv_len, mask, pad, nc, nc > 1 ? (float*)ws : nullptr, chunk_keys};
    // chunks of <= 128 keys under a uniform mask: the four-wave kernel, two workgroups per CU (NAR step 391 -> 375 us per
    // layer).  Not under the prefix mask: both slots of every CU are filled at once in index order, which pairs the heavy
    // low chunks with each other's neighbours instead of with the light high ones (B = 8: 269 -> 305 us).
    if (chunk_keys <= 128 && mode != VH_MASK_PREFIX)
        hipLaunchKernelGGL(attn_bwd_fused_kernel<4>, dim3(nc * B * n_heads), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attn_bwd_fused_kernel<8>, dim3(nc * B * n_heads), dim3(512), 0, (hipStream_t)stream, a);
    if (nc > 1) {
        const int64_t n4 = (int64_t)B * T * n_heads * 16;
        hipLaunchKernelGGL(attn_dq_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, n4);
    }
    VH_CHECK_LAUNCH("vh_attn_rows_bwd_ws");
    return VH_OK;
}

// =============================================================================================
// Decode attention over a SHARED PROMPT (round 5): the reference's own generate() replicates ONE utterance over
// num_beams rows (valle/models/valle_ar.py:135-138), so every beam's prompt K/V is the same bits.  Here the prompt K/V
// ("prefix": (1, h, prefix_S, 64) per layer, written once by a one-row prompt pass) is read ONCE per step for all beams
// and every beam keeps only its own generated rows ("suffix": (B, h, S_suf, 64)).  Bytes per step and layer:
// 2 (S0 + B t) d instead of 2 B (S0 + t) d.  Two launches:
//   attn_shared_kernel   workgroups of 4 waves in two roles.  PREFIX role (blocks of 4 x 32 prompt keys, per head): a wave
//                        takes one 32-key block and computes S^T = K Q^T for ALL beams at once — beams are the 32 lanes of a
//                        32x32x2 MFMA tile (a second pass for beams 32..63) — the block's softmax statistics and unnormalised
//                        O^T = V^T P^T, one split record per (beam, head, key block).  SUFFIX role ((beam, head), key split):
//                        the burst kernel over the beam's own rows, one record per split.  The roles share nothing and run
//                        side by side.
//   attn_records_merge_kernel   per (beam, head): all records merged — weights by one parallel pass over the (m, l) pairs,
//                        the o vectors summed by four thread groups over interleaved records, partials added in group
//                        order (deterministic).  (The serial loop of attn_decode_combine_kernel is fine for <= 16 splits;
//                        here there are up to 157 + 64 records, each load a trip beyond the L1.)
// Record format and units as the key-split decode kernels' (PART_LD floats: o[64], m, l; scores scaled by log2 e / 8).
// =============================================================================================
struct SharedArgs {
    const float* q; int ldq;
    const float* kp; const float* vp;            // prefix (1, h, prefix_S, 64)
    int prefix_len, prefix_S;
    const float* ks; const float* vs;            // suffix (B, h, S_suf, 64)
    const int32_t* suffix_len; int len_bias;
    int S_suf, n_split;
    float* partial;
    int n_heads, B, n_pb, n_tot, prefix_wgs;     // n_pb: 32-key blocks of the prefix; prefix_wgs = ceil(n_pb / 4) * n_heads
};

__device__ __forceinline__ void shared_prefix_role(const SharedArgs& a, int wg, int lane, int w) {
    const int head = wg % a.n_heads, blk = (wg / a.n_heads) * 4 + w;
    if (blk >= a.n_pb) return;                               // (wave-uniform; no barrier in this role)
    const int r = lane & 31, hh = lane >> 5;
    const float* kb = a.kp + (int64_t)head * a.prefix_S * HD;
    const float* vb = a.vp + (int64_t)head * a.prefix_S * HD;
    const int k0 = blk * 32;
    const float qscale = 0.125f * LOG2E;
    // K fragment (A operand): key k0 + r, d = 32 hh + j; rows beyond the prompt repeat its last row (masked below).  The k
    // index of the product is only summed over, so both operands simply use the same d per (hh, j).
    f32x4 kf[8];
    {
        const float* kr = kb + (int64_t)min(k0 + r, a.prefix_len - 1) * HD + 32 * hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = ld4(kr + 4 * j);
    }
    // V^T operand values: product x pairs key base(x) (lanes hh = 0) with base(x) + 4 (lanes hh = 1), base(x) = (x & 3) +
    // 8 (x >> 2) — exactly the keys whose weights sit in accumulator register x of the two lane halves
    float vf[2][16];
#pragma unroll
    for (int x = 0; x < 16; ++x) {
        const int key = min(k0 + (x & 3) + 8 * (x >> 2) + 4 * hh, a.prefix_len - 1);
        vf[0][x] = vb[(int64_t)key * HD + r];
        vf[1][x] = vb[(int64_t)key * HD + 32 + r];
    }
    const bool whole = k0 + 32 <= a.prefix_len;              // wave-uniform
    for (int qb = 0; qb * 32 < a.B; ++qb) {
        const int b = min(qb * 32 + r, a.B - 1);             // lanes beyond B repeat the last beam (never stored)
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        {
            const float* qr = a.q + (int64_t)b * a.ldq + head * HD + 32 * hh;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 qf = ld4(qr + 4 * j) * qscale;
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].x, qf.x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].y, qf.y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].z, qf.z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j].w, qf.w, s, 0, 0, 0);
            }
        }
        // D reg x of lane (r = beam, hh): key k0 + (x & 3) + 8 (x >> 2) + 4 hh
        float m = NEG_INF;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            if (!whole && k0 + (x & 3) + 8 * (x >> 2) + 4 * hh >= a.prefix_len) s[x] = NEG_INF;
            m = fmaxf(m, s[x]);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));                 // the beam's other key half; finite: key k0 < prefix_len
        float l = 0.f;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            s[x] = vh_exp2(s[x] - m);
            l += s[x];
        }
        l += __shfl_xor(l, 32, 64);
        f32x16 o0, o1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
#pragma unroll
        for (int x = 0; x < 16; ++x) {                       // O^T += V^T P^T: P^T is the B operand as it lies
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[0][x], s[x], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[1][x], s[x], o1, 0, 0, 0);
        }
        if (qb * 32 + r < a.B) {
            float* pr = a.partial + (((int64_t)b * a.n_heads + head) * a.n_tot + blk) * PART_LD;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                st4(pr + 8 * g4 + 4 * hh, f32x4{o0[4 * g4], o0[4 * g4 + 1], o0[4 * g4 + 2], o0[4 * g4 + 3]});
                st4(pr + 32 + 8 * g4 + 4 * hh, f32x4{o1[4 * g4], o1[4 * g4 + 1], o1[4 * g4 + 2], o1[4 * g4 + 3]});
            }
            if (hh == 0) { pr[HD] = m; pr[HD + 1] = l; }
        }
    }
}

__global__ __launch_bounds__(256) void attn_shared_kernel(SharedArgs a) {
    constexpr int NW = 4;
    __shared__ float s_m[NW], s_l[NW];
    __shared__ __attribute__((aligned(16))) float s_o[NW][HD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if ((int)blockIdx.x < a.prefix_wgs) {                    // workgroup-uniform
        shared_prefix_role(a, blockIdx.x, lane, w);
        return;
    }
    // ---- suffix role: (beam, head) x key split over the beam's own rows (the burst kernel's body)
    const int unit = blockIdx.x - a.prefix_wgs;
    const int split = unit % a.n_split, bh = unit / a.n_split;
    const int b = bh / a.n_heads, head = bh - b * a.n_heads;
    const int c16 = lane & 15, g = lane >> 4;
    const int len = a.suffix_len[b] + a.len_bias;
    const int nchunks = (len + 31) >> 5;
    const int cps = (nchunks + a.n_split - 1) / a.n_split;
    const int c_begin = split * cps, c_end = min(nchunks, c_begin + cps);
    const float qscale = 0.125f * LOG2E;
    const f32x4 q4 = ld4(a.q + (int64_t)b * a.ldq + head * HD + 4 * c16) * qscale;
    const float* kb = a.ks + (int64_t)bh * a.S_suf * HD + 4 * c16;
    const float* vb = a.vs + (int64_t)bh * a.S_suf * HD + 4 * c16;
    float m = NEG_INF, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    // two register sets: the next chunk of the wave is requested before the current one is reduced (rows beyond the length
    // are clamped to the row's last key and masked below: no load is predicated)
    f32x4 kf[2][8], vf[2][8];
    auto load = [&](int c, f32x4 (&kq)[8], f32x4 (&vq)[8]) {
        const int key0 = c * 32 + g;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = min(key0 + 4 * i, len - 1);
            kq[i] = ld4_stream(kb + (int64_t)key * HD);      // (non-temporal: the beams' rows must not evict the weights)
            vq[i] = ld4_stream(vb + (int64_t)key * HD);
        }
    };
    auto reduce = [&](int c, const f32x4 (&kq)[8], const f32x4 (&vq)[8]) {
        const int key0 = c * 32 + g;
        float sc[8];
        float cmax = NEG_INF;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 t = kq[i] * q4;
            const float d = row16_sum((t.x + t.y) + (t.z + t.w));
            sc[i] = (key0 + 4 * i < len) ? d : NEG_INF;
            cmax = fmaxf(cmax, sc[i]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m, cmax);                  // finite: chunk c < nchunks holds >= 1 valid key
        const float alpha = vh_exp2(m - m_new);
        o *= alpha;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p = vh_exp2(sc[i] - m_new);          // 0 for a masked key (its clamped V row is finite)
            l += p;
            o += vq[i] * p;
        }
        m = m_new;
    };
    if (c_begin + w < c_end) load(c_begin + w, kf[0], vf[0]);
    for (int c = c_begin + w; c < c_end; c += 2 * NW) {
        if (c + NW < c_end) load(c + NW, kf[1], vf[1]);
        reduce(c, kf[0], vf[0]);
        if (c + NW < c_end) {
            if (c + 2 * NW < c_end) load(c + 2 * NW, kf[0], vf[0]);
            reduce(c + NW, kf[1], vf[1]);
        }
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
    if (tid >= HD) return;
    float M = s_m[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) M = fmaxf(M, s_m[k]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const float wgt = s_m[k] == NEG_INF ? 0.f : vh_exp2(s_m[k] - M);      // (a wave without a chunk; M itself is -inf
        L += s_l[k] * wgt;                                                    //  only for an empty split)
        O += s_o[k][tid] * wgt;
    }
    float* pr = a.partial + ((int64_t)bh * a.n_tot + a.n_pb + split) * PART_LD;
    pr[tid] = O;
    if (tid == 0) { pr[HD] = M; pr[HD + 1] = L; }
}

// out[b, head] = merge of the n_tot records of (b, head).  256 threads: one parallel pass over the (m, l) pairs gives the
// weights and the denominator; thread group g = tid >> 6 then sums records g, g + 4, ... of column tid & 63 (independent
// loads, issued back to back) and the four partial sums are added in group order.
__global__ __launch_bounds__(256) void attn_records_merge_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                 int ldo, int n_heads, int n_tot) {
    __shared__ float s_w[256], s_red[8], s_part[4][HD];
    const int bh = blockIdx.x, b = bh / n_heads, head = bh - b * n_heads;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* pr = partial + (int64_t)bh * n_tot * PART_LD;
    // ONE memory round trip: the (m, l) pair of record tid and the first eight o values of this thread's column (records
    // w, w + 4, ..., w + 28) are requested together — the o values do not depend on the weights
    const bool have = tid < n_tot;
    const float mk = have ? pr[tid * PART_LD + HD] : NEG_INF;
    const float lk = have ? pr[tid * PART_LD + HD + 1] : 0.f;
    float a8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a8[j] = pr[min(w + 4 * j, n_tot - 1) * PART_LD + lane];
    const float wm = wave_max(mk);
    if (lane == 0) s_red[w] = wm;
    __syncthreads();
    const float M = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));   // finite: the prefix holds >= 1 key
    const float wk = mk == NEG_INF ? 0.f : vh_exp2(mk - M);
    s_w[tid] = wk;                                           // (0 beyond n_tot: the clamped loads above weigh nothing)
    const float ws = wave_sum(lk * wk);
    if (lane == 0) s_red[4 + w] = ws;
    __syncthreads();
    const float L = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a8[j] * s_w[w + 4 * j];
    for (int k = w + 32; k < n_tot; k += 16) {               // longer prompts: four more independent loads per pass
        const float a0 = pr[k * PART_LD + lane], a1 = pr[min(k + 4, n_tot - 1) * PART_LD + lane];
        const float a2 = pr[min(k + 8, n_tot - 1) * PART_LD + lane], a3 = pr[min(k + 12, n_tot - 1) * PART_LD + lane];
        acc += a0 * s_w[k];
        acc += a1 * s_w[min(k + 4, 255)];
        acc += a2 * s_w[min(k + 8, 255)];
        acc += a3 * s_w[min(k + 12, 255)];
    }
    s_part[w][lane] = acc;
    __syncthreads();
    if (tid < HD) out[(int64_t)b * ldo + head * HD + tid] = ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) / L;
}

extern "C" size_t vh_attn_decode_shared_ws_bytes(int B, int n_heads, int prefix_len, int n_split_suffix) {
    if (B <= 0 || n_heads <= 0 || prefix_len <= 0 || n_split_suffix < 1) return 0;
    const int n_tot = (prefix_len + 31) / 32 + n_split_suffix;
    return (size_t)B * n_heads * n_tot * PART_LD * sizeof(float);
}

extern "C" int vh_attn_decode_shared(const float* q, int ldq, const float* kprefix, const float* vprefix, int prefix_len,
                                     int prefix_S, const float* ksuffix, const float* vsuffix, float* out, int ldo,
                                     const int32_t* suffix_len, int len_bias, int B, int n_heads, int S_suf,
                                     int n_split_suffix, void* partial, size_t partial_bytes, void* stream) {
    VH_REQUIRE(q && kprefix && vprefix && ksuffix && vsuffix && out && suffix_len && partial, VH_EINVAL,
               "vh_attn_decode_shared: null pointer");
    VH_REQUIRE(B > 0 && B <= 64 && n_heads > 0 && prefix_len > 0 && prefix_len <= prefix_S && S_suf > 0 &&
                   n_split_suffix >= 1 && n_split_suffix <= 64, VH_EINVAL,
               "vh_attn_decode_shared: bad dims B=%d h=%d prefix=%d/%d S_suf=%d n_split=%d", B, n_heads, prefix_len, prefix_S,
               S_suf, n_split_suffix);
    const int n_pb = (prefix_len + 31) / 32, n_tot = n_pb + n_split_suffix;
    VH_REQUIRE(n_tot <= 256, VH_EUNSUPPORTED, "vh_attn_decode_shared: %d prefix blocks + %d splits exceed the 256 records one merge serves",
               n_pb, n_split_suffix);
    VH_REQUIRE(len_bias == 0 || len_bias == 1, VH_EINVAL, "vh_attn_decode_shared: len_bias=%d", len_bias);
    VH_REQUIRE(ldq % 4 == 0 && ldq >= n_heads * HD && ldo >= n_heads * HD, VH_EINVAL, "vh_attn_decode_shared: ldq=%d ldo=%d", ldq, ldo);
    VH_REQUIRE(vh_aligned16(q) && vh_aligned16(kprefix) && vh_aligned16(vprefix) && vh_aligned16(ksuffix) &&
                   vh_aligned16(vsuffix) && vh_aligned16(partial), VH_EALIGN, "vh_attn_decode_shared: pointers must be 16-byte aligned");
    VH_REQUIRE(partial_bytes >= vh_attn_decode_shared_ws_bytes(B, n_heads, prefix_len, n_split_suffix), VH_EINVAL,
               "vh_attn_decode_shared: workspace of %zu bytes, need %zu", partial_bytes,
               vh_attn_decode_shared_ws_bytes(B, n_heads, prefix_len, n_split_suffix));
    hipStream_t s = (hipStream_t)stream;
    const int prefix_wgs = (n_pb + 3) / 4 * n_heads;
    SharedArgs a{q, ldq, kprefix, vprefix, prefix_len, prefix_S, ksuffix, vsuffix, suffix_len, len_bias, S_suf, n_split_suffix,
                 (float*)partial, n_heads, B, n_pb, n_tot, prefix_wgs};
    // (the events of vh_ar_decoder_profile_attn bracket this launch: the step's attention kernel in this form)
    hipExtLaunchKernelGGL(attn_shared_kernel, dim3(prefix_wgs + n_split_suffix * B * n_heads), dim3(256), 0, s, g_attn_ev[0],
                          g_attn_ev[1], 0, a);
    hipLaunchKernelGGL(attn_records_merge_kernel, dim3(B * n_heads), dim3(256), 0, s, (const float*)partial, out, ldo, n_heads,
                       n_tot);
    VH_CHECK_LAUNCH("vh_attn_decode_shared");
    return VH_OK;
}
