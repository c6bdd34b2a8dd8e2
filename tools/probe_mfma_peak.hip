// Diagnostic: what does the chip sustain on fp32 MFMA alone (no memory traffic in the loop), with random
// (non-zero) operands — the clock under matrix load is data-dependent (MI355X_MICROARCH "DVFS give-back")?
// hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_peak.hip -o tools/probe_mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>   // 0: 32x32x2, 4 accumulators; 1: 16x16x4, 16 accumulators
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters, int prio) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    // prio 1: the second half of the grid runs at wave priority 1; prio 2: odd workgroups do
    if (prio == 1 && (blockIdx.x >> 8)) __builtin_amdgcn_s_setprio(1);
    if (prio == 2 && (blockIdx.x & 1)) __builtin_amdgcn_s_setprio(1);
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + i + 4096) & 65535]; }
    if (KIND == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 1) & 7], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 1) & 7], b[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 1) & 7], b[(j + 1) & 7], acc[3], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        out[t] = s;
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 16; ++x)
                    acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + (x >> 2)) & 7], b[(j + x) & 7], acc[x], 0, 0, 0);
        }
        f32x4 s = acc[0];
        for (int i = 1; i < 16; ++i) s += acc[i];
        out[t] = s.x + s.y + s.z + s.w;
    }
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 65536 * 4)); CK(hipMalloc(&out, 256 * 2048 * 4));
    for (int rnd = 0; rnd < 2; ++rnd) {
        std::vector<float> h(65536);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = rnd ? ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f : 0.f; }
        CK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
        for (int kind = 0; kind < 2; ++kind)
            for (int prio = 0; prio < 3; ++prio)
            for (int wgs = (prio ? 512 : 256); wgs <= 512; wgs *= 2) {       // 1 or 2 waves per SIMD
                const int iters = 20000;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0, 0);
                    if (kind == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(wgs), dim3(256), 0, 0, in, out, iters, prio);
                    else hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(256), 0, 0, in, out, iters, prio);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)wgs * 4 * iters * (kind == 0 ? 32.0 * 4096 : 64.0 * 2048);
                printf("%s operands, %s, prio mode %d, %d wave(s)/SIMD: %.1f ms -> %.1f TFLOP/s\n", rnd ? "random" : "zero  ",
                       kind == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", prio, wgs / 256, ms, flops / ms / 1e9);
            }
    }
    return 0;
}
