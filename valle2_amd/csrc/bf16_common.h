// Shared pieces of the perf-mode (bf16 MFMA) kernels: bf16.hip (128^2 tile machines, attention, LayerNorm) and
// gemm16p.hip (the persistent 256^2 / 8-wave tile machine).
#pragma once
#include "vh_common.h"

typedef vh_h16x8 bf16x8;                 // (round-5 name: eight h16 = one 32x32x16 operand fragment)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) { return vh_pack_h16(lo, hi); }      // (round-5 name)
__device__ __forceinline__ u32x4 ldq(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void stq(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// GELU (exact-erf form, nn.GELU() of valle/models/modules.py:216) for a result that is ROUNDED TO bf16 right after: erf as
// 1 - 2^(R5(t) - log2(e) t^2), t = min(|z|, 4), R5 ~ log2(erfcx(t)) (degree 5, least squares on Chebyshev nodes of [0, 4]),
// one branch — the cancellation near z = 0 is an ABSOLUTE error of 4e-5 in gelu, a sixth of bf16's half ulp wherever
// |gelu| > 0.03, and below SURVEY 8(c)'s 5e-2 by three orders.  12 vector instructions per pair against gelu_erf2's 25:
// at bf16 MFMA rates the epilogue's GELU costs as much issue time as the tile's products.
__device__ __forceinline__ vh_f32x2 gelu16_2(vh_f32x2 x) {
    const vh_f32x2 z = x * vh_splat2(0.70710678118654752440f);
    const vh_f32x2 t = {fminf(fabsf(z.x), 4.0f), fminf(fabsf(z.y), 4.0f)};
    vh_f32x2 q = vh_splat2(-1.2391665950417519e-03f);
    q = __builtin_elementwise_fma(q, t, vh_splat2(1.875305362045765e-02f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(-1.2387025356292725e-01f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(5.004633665084839e-01f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(-1.6198172569274902e+00f));
    q = __builtin_elementwise_fma(q, t, vh_splat2(-4.4126855209469795e-04f));
    q = __builtin_elementwise_fma(vh_splat2(-1.4426950408889634f) * t, t, q);
    const vh_f32x2 e = {copysignf(1.0f - __builtin_amdgcn_exp2f(q.x), z.x), copysignf(1.0f - __builtin_amdgcn_exp2f(q.y), z.y)};
    return vh_splat2(0.5f) * x * (vh_splat2(1.0f) + e);
}

enum { G16_F32 = 0, G16_BF16 = 1, G16_QKV = 2 };

struct Gemm16Args {
    const uint16_t* A;
    int lda;
    const uint16_t* W;
    const float* bias;
    const float* res;
    int ldr;
    void* out;
    int ldo;
    int M, N, K, act;
    uint16_t* kc;
    uint16_t* vc;
    const int32_t* cache_len;
    int T, S_max, d_model, n_heads;
};


// gemm16p.hip: the persistent 256 x 256 / 8-wave form; returns false when the shape is not one it takes
bool vh_gemm16_p256_ok(const Gemm16Args& a, int out_kind);
int vh_gemm16_p256_launch(const Gemm16Args& a, int out_kind, hipStream_t stream);
