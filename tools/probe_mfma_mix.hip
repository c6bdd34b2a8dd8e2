// Diagnostic: which companion instructions take matrix throughput away from a wave's MFMA stream?
// One wave per SIMD (256 workgroups x 256 threads), v_mfma_f32_32x32x2_f32 on 4 accumulators, 64 MFMAs per
// "step" as in gemm_tile_kernel, plus per step: R x ds_read_b128 feeding the operands, W x ds_write_b128,
// G x global_load_dwordx4, B barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, int W, int G, int B>
__global__ __launch_bounds__(256, 1) void mix(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36 * 2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 2 * 128 * 36 * 2; i += 256) lds[i] = in[i & 65535];
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 f[16];
    for (int i = 0; i < 16; ++i) f[i] = *reinterpret_cast<const f32x4*>(in + ((tid * 4 + i * 1024) & 65532));
    f32x4 gl[8];
    for (int i = 0; i < 8; ++i) gl[i] = f[i];
    const float* lp = lds + ((w >> 1) * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
    const float* gp = in + ((blockIdx.x * 256 + tid) * 4 & 65532);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (R) {
#pragma unroll
                for (int x = 0; x < R / 4; ++x) f[(t & 1) * 8 + x] = *reinterpret_cast<const f32x4*>(lp + x * 32 * 36 + 8 * t + (it & 1) * 9216);
            }
            if (G) {
#pragma unroll
                for (int x = 0; x < G / 4; ++x) gl[t * (G / 4) + x] = *reinterpret_cast<const f32x4*>(gp + ((it * 4 + t) * 64 + x * 8192 & 65532));
            }
            __builtin_amdgcn_sched_barrier(0);
            const int s = ((t + 1) & 1) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[s + 0][j], f[s + 2][j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[s + 0][j], f[s + 3][j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[s + 1][j], f[s + 2][j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[s + 1][j], f[s + 3][j], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (W) {
#pragma unroll
            for (int x = 0; x < W; ++x) *reinterpret_cast<f32x4*>(lds + ((it + 1) & 1) * 9216 + ((tid >> 3) + 32 * (x & 3)) * 36 + (tid & 7) * 4 + (x >> 2) * 4608) = gl[x & 7];
        }
        if (B) __syncthreads();
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) sum += acc[i][e];
    for (int i = 0; i < 8; ++i) sum += gl[i].x;
    out[blockIdx.x * 256 + tid] = sum;
}

template <int R, int W, int G, int B> int run(const float* in, float* out, const char* name) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mix<R, W, G, B>), dim3(256), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    CK(hipGetLastError());
    printf("%-58s %.1f TFLOP/s\n", name, 256.0 * 4 * iters * 64 * 4096 / ms / 1e9);
    return 0;
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 65536 * 4 + 64)); CK(hipMalloc(&out, 256 * 256 * 4));
    float* h = new float[65536];
    unsigned s = 1; for (int i = 0; i < 65536; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    CK(hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice));
    run<0, 0, 0, 0>(in, out, "64 MFMA per step only");
    run<16, 0, 0, 0>(in, out, "+ 16 ds_read_b128 (operands)");
    run<16, 8, 0, 0>(in, out, "+ 16 ds_read_b128 + 8 ds_write_b128");
    run<16, 8, 0, 1>(in, out, "+ 16 ds_read + 8 ds_write + barrier");
    run<0, 0, 8, 0>(in, out, "+ 8 global_load_dwordx4");
    run<16, 8, 8, 1>(in, out, "+ 16 ds_read + 8 ds_write + 8 global_load + barrier");
    run<0, 0, 0, 1>(in, out, "+ barrier only");
    return 0;
}
