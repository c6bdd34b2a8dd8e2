// Diagnostic (never shipped; its first version dropped the loop remainders and so over-reported by 25 %): the read-only
// streaming ceiling of this part, to judge the decode-attention kernel
// (5.9-6.2 TB/s) against what ANY kernel can read: grid-stride float4 loads (plain / non-temporal), U loads in
// flight per thread, 256 workgroups x 1024 threads (one per CU, as the attention kernel) or more.
// hipcc --offload-arch=gfx950 -O3 tools/probe_read_bw.hip -o tools/probe_read_bw.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(1024) void read_kernel(const f32x4* __restrict__ p, size_t n4, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += U * stride) {                      // every element exactly once: the tail re-reads the last one
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride < n4 ? i + u * stride : n4 - 1;
            v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// contiguous per-workgroup streams, as the attention kernel reads them: workgroup b walks its own 1/gridDim of the buffer
template <int U, bool NT>
__global__ __launch_bounds__(1024) void read_streams_kernel(const f32x4* __restrict__ p, size_t n4, float* out) {
    const size_t per = n4 / gridDim.x;
    const f32x4* q = p + (size_t)blockIdx.x * per;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per; i += U * 1024) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * 1024 < per ? i + u * 1024 : per - 1;
            v[u] = NT ? __builtin_nontemporal_load(q + j) : q[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

int main() {
    const size_t bytes = (size_t)168 << 20;          // one layer's K + V at S = 1280, B = 32
    const int NBUF = 12;                             // rotate over 12 buffers (2 GB) as the layers do
    f32x4* buf[NBUF];
    float* out;
    for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], 0x3c, bytes)); }
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1, k0, k1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    const size_t n4 = bytes / 16;
#define RUN(name, kern, grid)                                                                       \
    do {                                                                                            \
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, buf[w], n4, out); \
        CK(hipEventRecord(e0, 0));                                                                  \
        for (int it = 0; it < 48; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, buf[it % NBUF], n4, out); \
        CK(hipEventRecord(e1, 0));                                                                  \
        CK(hipEventSynchronize(e1));                                                                \
        float ms;                                                                                   \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                       \
        /* per-dispatch begin/end timestamps (no overlap between neighbours counted) */            \
        double ksum = 0;                                                                            \
        for (int it = 0; it < 24; ++it) {                                                           \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, k0, k1, 0, buf[it % NBUF], n4, out); \
            CK(hipEventSynchronize(k1));                                                            \
            float kms;                                                                              \
            CK(hipEventElapsedTime(&kms, k0, k1));                                                  \
            ksum += kms;                                                                            \
        }                                                                                           \
        printf("%-52s grid %4d: back-to-back %6.2f us = %5.2f TB/s; per dispatch %6.2f us = %5.2f TB/s\n", name, grid, \
               ms / 48 * 1e3, bytes / (ms / 48 * 1e-3) / 1e12, ksum / 24 * 1e3, bytes / (ksum / 24 * 1e-3) / 1e12); \
    } while (0)
    RUN("grid-stride, plain, 4 in flight", (read_kernel<4, false>), 256);
    RUN("grid-stride, nontemporal, 4 in flight", (read_kernel<4, true>), 256);
    RUN("grid-stride, nontemporal, 8 in flight", (read_kernel<8, true>), 256);
    RUN("grid-stride, nontemporal, 16 in flight", (read_kernel<16, true>), 256);
    RUN("grid-stride, nontemporal, 8 in flight", (read_kernel<8, true>), 512);
    RUN("grid-stride, nontemporal, 8 in flight", (read_kernel<8, true>), 1024);
    RUN("per-workgroup streams, nontemporal, 8 in flight", (read_streams_kernel<8, true>), 256);
    RUN("per-workgroup streams, nontemporal, 16 in flight", (read_streams_kernel<16, true>), 256);
    RUN("per-workgroup streams, plain, 8 in flight", (read_streams_kernel<8, false>), 256);
    RUN("per-workgroup streams, nontemporal, 8 in flight", (read_streams_kernel<8, true>), 512);
    return 0;
}
