// Shared device/host helpers for libvalle_hip.so (gfx950 only: wave64, MFMA f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/valle_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- the 16-bit operand format of PERF MODE ("h16": K/V cache of the decode step, operands of the bf16.hip / gemm16p.hip
// kernels).  Default: IEEE fp16 — the MFMA rate of the two formats is the same (v_mfma_f32_32x32x16_f16 / _bf16), but fp16 keeps 11
// bits of significand against bf16's 8: teacher-forced logits of the 24-layer / 1024-d stack against the REAL reference are
// 5.5e-3 off with fp16 and 4.2e-2 with bf16 (SURVEY 8c's bound: 5e-2; profiles/r6_probe_precision.log: the WEIGHTS' rounding
// alone is 3.1e-2 in bf16).  Every narrowed quantity of this path is a normalised activation, a projection of one, a softmax
// weight or a weight: |x| < 7 on the goldens against fp16's 65504.  The narrowing is IEEE: a value beyond the range becomes an
// infinity and a NaN stays a NaN — a broken input shows in the output instead of being clamped away.
// -DVH_PERF_BF16 builds the round-5 format.  vh_h16_format() tells the host side which one a library carries; entry points
// and file names keep their round-5 "bf16" names.
#ifdef VH_PERF_BF16
#define VH_H16_IS_BF16 1
typedef __bf16 vh_h16;
#else
#define VH_H16_IS_BF16 0
typedef _Float16 vh_h16;
#endif
typedef vh_h16 vh_h16x2 __attribute__((ext_vector_type(2)));
typedef vh_h16 vh_h16x8 __attribute__((ext_vector_type(8)));
// two fp32 -> one dword of two h16, round to nearest even (v_cvt_pk_bf16_f32 | 2 x v_cvt_f16_f32 + v_pack_b32_f16)
__device__ __forceinline__ uint32_t vh_pack_h16(float lo, float hi) {
#if VH_H16_IS_BF16
    const vh_h16x2 r = {(__bf16)lo, (__bf16)hi};
#else
    const vh_h16x2 r = {(_Float16)lo, (_Float16)hi};
#endif
    return __builtin_bit_cast(uint32_t, r);
}
// the same for values known to lie in [0, 1] (softmax weights): fp16 by truncation in ONE instruction (v_cvt_pkrtz_f16_f32) —
// a relative 2^-11 low, uniformly, which the row sum (taken from the fp32 weights) turns into a 5e-4 scale of the
// attention output; bf16 has the packed round-to-nearest instruction anyway
__device__ __forceinline__ uint32_t vh_pack_h16_unit(float lo, float hi) {
#if VH_H16_IS_BF16
    return vh_pack_h16(lo, hi);
#else
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lo, hi));
#endif
}
// one dword of two h16 -> fp32
__device__ __forceinline__ float vh_h16_lo(uint32_t w) {
#if VH_H16_IS_BF16
    return __uint_as_float(w << 16);
#else
    return (float)__builtin_bit_cast(vh_h16x2, w)[0];
#endif
}
__device__ __forceinline__ float vh_h16_hi(uint32_t w) {
#if VH_H16_IS_BF16
    return __uint_as_float(w & 0xffff0000u);
#else
    return (float)__builtin_bit_cast(vh_h16x2, w)[1];
#endif
}
#if VH_H16_IS_BF16
#define VH_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#else
#define VH_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

// perf mode: the QKV product writes q PRE-SCALED by 1 / sqrt(64) * log2(e) (in fp32, before the one narrowing), so that the score
// accumulators of the many-row attention are base-2 exponents (include/valle_hip.h, vh_linear_qkv_bf16 / vh_attn_rows_bf16)
#define VH_Q16_PRESCALE 0.18033688011112042f

#define VH_WAVE 64

// ---- host side: argument checking ------------------------------------------------------------
void vh_set_error(const char* fmt, ...);

#define VH_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            vh_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)

static inline bool vh_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define VH_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            vh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
            return VH_ELAUNCH;                                                     \
        }                                                                          \
    } while (0)

// tuning knobs (vh_set_tuning); 0 = built-in default
int vh_tuning(int knob);

// ---- device side -----------------------------------------------------------------------------
// Cross-lane reductions on DPP (data-parallel primitives): one VALU instruction per step and no
// LDS crossbar traffic (ds_bpermute), which matters twice here: the decode kernels are latency
// bound, and every launch starts with a cold instruction cache, so compact code is faster code.
//   quad_perm(1,0,3,2)=0xB1  quad_perm(2,3,0,1)=0x4E  row_mirror=0x140  row_half_mirror=0x141
//   row_bcast15=0x142 (lane 15 of each row → next row)  row_bcast31=0x143 (lane 31 → rows 2,3)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v, float identity) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL,
                                                      ROW_MASK, 0xF, false));
}
// sum over the 16 lanes of one DPP row (lanes 16g..16g+15); every lane of the row gets the sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_get<0xB1, 0xF>(v, 0.f);
    v += dpp_get<0x4E, 0xF>(v, 0.f);
    v += dpp_get<0x141, 0xF>(v, 0.f);
    v += dpp_get<0x140, 0xF>(v, 0.f);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_get<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_get<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_get<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_get<0x140, 0xF>(v, v));
    return v;
}
// combine the 4 row values of a wave (each row already uniform): result uniform over the wave
__device__ __forceinline__ float rows4_sum(float v) {
    v += dpp_get<0x142, 0xA>(v, 0.f);
    v += dpp_get<0x143, 0xC>(v, 0.f);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float rows4_max(float v) {
    v = fmaxf(v, dpp_get<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_get<0x143, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum(float v) { return rows4_sum(row16_sum(v)); }
__device__ __forceinline__ float wave_max(float v) { return rows4_max(row16_max(v)); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// streaming (read-once) load: non-temporal hint, so the KV stream does not evict the weights that
// every decode step re-reads from L2 / Infinity Cache
__device__ __forceinline__ f32x4 ld4_stream(const float* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// 2^x as the bare v_exp_f32 (1 ulp; results below 2^-126 flush to zero, -inf -> 0): exp2f() wraps the same
// instruction in a denormal-range fix-up (compare, select, add, ldexp — five instructions), which the softmax
// kernels do not need: their arguments are score - running max <= 0 and a flushed tail weight is exactly what
// the sum would round away.
__device__ __forceinline__ float vh_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf(x) to 1.2e-7 absolute (about one ulp of the result near 1) in ~20 vector instructions, a third of the
// library erff: the GEMM epilogues evaluate it 64 times per lane and per tile, and every one of those instructions
// is a bubble in the MFMA stream of the CU's other workgroup.  |x| < 1: x * P5(x^2).  Otherwise
// 1 - 2^(R7(t) - log2(e) t^2) with t = min(|x|, 4) and R7 ~ log2(erfcx(t)) (erf(4) rounds to 1 in fp32).
// Coefficients: Chebyshev fits of the two branches, rounded to fp32 (tests/test_kernels_gpu.py checks the
// function against torch.erf on a dense grid).
__device__ __forceinline__ float vh_erf(float x) {
    const float s = x * x;
    float p = -5.654105917e-04f;
    p = fmaf(p, s, 4.923277535e-03f);
    p = fmaf(p, s, -2.671638504e-02f);
    p = fmaf(p, s, 1.128036454e-01f);
    p = fmaf(p, s, -3.761234879e-01f);
    p = fmaf(p, s, 1.128379107e+00f);
    const float small = p * x;
    const float t = fminf(fabsf(x), 4.0f);
    float q = -1.920139221e-05f;
    q = fmaf(q, t, 4.581596004e-04f);
    q = fmaf(q, t, -4.958351608e-03f);
    q = fmaf(q, t, 3.263662755e-02f);
    q = fmaf(q, t, -1.490577757e-01f);
    q = fmaf(q, t, 5.206782818e-01f);
    q = fmaf(q, t, -1.624367833e+00f);
    q = fmaf(q, t, -1.092074905e-03f);
    q = fmaf(-1.4426950408889634f * t, t, q);
    const float large = copysignf(1.0f - __builtin_amdgcn_exp2f(q), x);   // v_exp_f32; q >= -26, no denormals
    return fabsf(x) < 1.0f ? small : large;
}

// Two GELUs at once: the polynomial work of vh_erf on packed pairs (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32
// process two fp32 per instruction), the two exponentials and the selects scalar.  Same formulas and coefficients
// as vh_erf / gelu_erf, so a pair gives bit-identical results to two scalar calls (fma is fma either way).
typedef float vh_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vh_f32x2 vh_splat2(float v) { return vh_f32x2{v, v}; }
__device__ __forceinline__ vh_f32x2 gelu_erf2(vh_f32x2 x) {
    const vh_f32x2 z = x * vh_splat2(0.70710678118654752440f);
    const vh_f32x2 s = z * z;
    vh_f32x2 p = vh_splat2(-5.654105917e-04f);
    p = __builtin_elementwise_fma(p, s, vh_splat2(4.923277535e-03f));
    p = __builtin_elementwise_fma(p, s, vh_splat2(-2.671638504e-02f));
    p = __builtin_elementwise_fma(p, s, vh_splat2(1.128036454e-01f));
    p = __builtin_elementwise_fma(p, s, vh_splat2(-3.761234879e-01f));
    p = __builtin_elementwise_fma(p, s, vh_splat2(1.128379107e+00f));
    const vh_f32x2 small = p * z;
    const vh_f32x2 t = {fminf(fabsf(z.x), 4.0f), fminf(fabsf(z.y), 4.0f)};
    vh_f32x2 q = vh_splat2(-1.920139221e-05f);
    q = __builtin_elementwise_fma(q, t, vh_splat2(4.581596004e-04f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(-4.958351608e-03f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(3.263662755e-02f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(-1.490577757e-01f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(5.206782818e-01f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(-1.624367833e+00f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(-1.092074905e-03f));
    q = __builtin_elementwise_fma(vh_splat2(-1.4426950408889634f) * t, t, q);
    vh_f32x2 e;
    e.x = fabsf(z.x) < 1.0f ? small.x : copysignf(1.0f - __builtin_amdgcn_exp2f(q.x), z.x);
    e.y = fabsf(z.y) < 1.0f ? small.y : copysignf(1.0f - __builtin_amdgcn_exp2f(q.y), z.y);
    return vh_splat2(0.5f) * x * (vh_splat2(1.0f) + e);
}

__device__ __forceinline__ float gelu_erf(float x) {
    // nn.GELU() default = exact erf form (valle/models/modules.py:216)
    return 0.5f * x * (1.0f + vh_erf(x * 0.70710678118654752440f));
}

// d/dx of the exact-erf GELU: Phi(x) + x * phi(x)  (the backward of modules.py:216)
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + vh_erf(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return cdf + x * pdf;
}

// GELU and its derivative from ONE evaluation of erf (the training forward stores the derivative instead of the
// pre-activation, so the backward's epilogue is a multiply): gelu = x Phi(x), d = Phi(x) + x phi(x).
__device__ __forceinline__ float gelu_and_grad(float x, float& d) {
    const float cdf = 0.5f * (1.0f + vh_erf(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    d = cdf + x * pdf;
    return x * cdf;
}

__device__ __forceinline__ f32x4 gelu_and_grad4(f32x4 v, f32x4& d) {
    float d0, d1, d2, d3;
    const f32x4 g = {gelu_and_grad(v.x, d0), gelu_and_grad(v.y, d1), gelu_and_grad(v.z, d2), gelu_and_grad(v.w, d3)};
    d = f32x4{d0, d1, d2, d3};
    return g;
}

// ---- dropout field (include/valle_hip.h, vh_dropout_spec) ---------------------------------------------------------
// Philox4x32 with 7 rounds (the fewest that pass BigCrush in Salmon et al., SC'11 table 2; 10 is the library default
// with its safety margin — a dropout mask needs neither cryptographic nor Monte-Carlo-grade streams, and every round is
// two 32x32->64 multiplies in epilogues that share their issue port with the other workgroup's MFMA stream).
// One call covers 4 consecutive columns of one row: exactly the float4 a lane of the tile epilogue / the row kernels
// owns, so no lane ever computes bits it throws away.
struct DropArgs {
    uint32_t k0, k1;      // key   = seed
    uint32_t s0, s1;      // counter words 2, 3 = site
    uint32_t thresh;      // keep iff bits >= thresh (= round(p * 2^32)); 0 = dropout off
    float scale;          // 1 / (1 - p)
};

__device__ __forceinline__ void vh_philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                                uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
}
// (written out round by round: a loop here, even a fully unrollable one, kept the 16-row epilogue loop around it from
// being unrolled, and the epilogue's per-row registers went to scratch)
__device__ __forceinline__ void vh_philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1, uint32_t (&o)[4]) {
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    vh_philox_round(c0, c1, c2, c3, k0, k1);
    vh_philox_round(c0, c1, c2, c3, k0 + W0, k1 + W1);
    vh_philox_round(c0, c1, c2, c3, k0 + 2 * W0, k1 + 2 * W1);
    vh_philox_round(c0, c1, c2, c3, k0 + 3 * W0, k1 + 3 * W1);
    vh_philox_round(c0, c1, c2, c3, k0 + 4 * W0, k1 + 4 * W1);
    vh_philox_round(c0, c1, c2, c3, k0 + 5 * W0, k1 + 5 * W1);
    vh_philox_round(c0, c1, c2, c3, k0 + 6 * W0, k1 + 6 * W1);
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// keep / (1 - p) factors of columns 4 c4 .. 4 c4 + 3 of row `row`
__device__ __forceinline__ f32x4 vh_dropmul4(const DropArgs& d, uint32_t row, uint32_t c4) {
    uint32_t b[4];
    vh_philox4x32_7(c4, row, d.s0, d.s1, d.k0, d.k1, b);
    return f32x4{b[0] >= d.thresh ? d.scale : 0.f, b[1] >= d.thresh ? d.scale : 0.f,
                 b[2] >= d.thresh ? d.scale : 0.f, b[3] >= d.thresh ? d.scale : 0.f};
}

// host: spec -> kernel arguments; returns false (and leaves `a` off) for a NULL spec or p == 0
static inline bool vh_drop_args(const vh_dropout_spec* spec, DropArgs* a) {
    *a = DropArgs{0, 0, 0, 0, 0, 1.0f};
    if (!spec || !(spec->p > 0.f)) return false;
    a->k0 = (uint32_t)spec->seed; a->k1 = (uint32_t)(spec->seed >> 32);
    a->s0 = (uint32_t)spec->site; a->s1 = (uint32_t)(spec->site >> 32);
    double t = (double)spec->p * 4294967296.0 + 0.5;
    a->thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    if (a->thresh == 0) a->thresh = 1;               // a p below 2^-33 still drops something rather than turning off
    a->scale = (float)(1.0 / (1.0 - (double)spec->p));
    return true;
}
#define VH_DROP_OK(spec) (!(spec) || ((spec)->p >= 0.f && (spec)->p < 1.f))

// LayerNorm parameters that may be fused into an operand load
struct LnFuse {
    const float* gamma;  // (K) or nullptr → no fused LN
    const float* beta;
    const float* ada_scale;  // (K) or nullptr
    const float* ada_shift;
    float eps;
    // folded form (vh_ln_fold): the weight operand is W∘gamma and LN(x)·Wᵀ + bias is rebuilt in the
    // epilogue as rstd·(x·Wfᵀ − mean·c1) + c2; gamma/beta are unused when c1 != nullptr
    const float* c1;
    const float* c2;
};
