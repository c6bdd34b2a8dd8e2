// Shared device/host helpers for libvalle_hip.so (gfx950 only: wave64, MFMA f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/valle_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// ---- device side -----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce over the 16 lanes of one DPP row (lanes 16g..16g+15); every lane gets the sum
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ float gelu_erf(float x) {
    // nn.GELU() default = exact erf form (valle/models/modules.py:216)
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// LayerNorm parameters that may be fused into an operand load
struct LnFuse {
    const float* gamma;  // (K) or nullptr → no fused LN
    const float* beta;
    const float* ada_scale;  // (K) or nullptr
    const float* ada_shift;
    float eps;
};
