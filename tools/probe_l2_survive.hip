// Do L2 lines survive a kernel boundary on this part?  (developer probe)
// Kernel A touches a buffer (each workgroup its own 16 KB slice); kernel B — a dependent launch on the same stream,
// same grid, so the same workgroup index lands on the same XCD — times one dependent pointer-chase-free load per lane
// of the SAME slice with the shader clock.  Compared: B after A (warm), B after a kernel that touched another buffer
// (cold), and the warm case with 256 MB streamed in between.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_l2_survive.hip -o gpurun_out/probe_l2_survive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void touch(const float4* buf, float* sink, int per_wg) {
    const float4* p = buf + (size_t)blockIdx.x * per_wg;
    float4 acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < per_wg; i += blockDim.x) { float4 v = p[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
}

__global__ void timed(const float4* buf, float* sink, int per_wg, long long* out) {
    const float4* p = buf + (size_t)blockIdx.x * per_wg;
    const long long t0 = wall_clock64();
    float4 v = p[threadIdx.x];                       // one 16-byte load per lane, first thing the kernel does
    if (v.x == 123.456f) sink[1] = v.y;
    __syncthreads();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

__global__ void stream_other(const float4* big, float* sink, size_t n) {
    float4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(big) + i); acc.x += v.x; acc.y += v.w;
    }
    if (acc.x == 123.456f) sink[2] = acc.y;
}

int main() {
    const int wgs = 256, threads = 512, per_wg = 1024;          // 16 KB per workgroup, 4 MB in all
    float4 *a, *b, *big; float* sink; long long* out;
    const size_t nbig = (size_t)256 << 20 >> 4;
    CK(hipMalloc(&a, (size_t)wgs * per_wg * 16)); CK(hipMalloc(&b, (size_t)wgs * per_wg * 16)); CK(hipMalloc(&big, nbig * 16));
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, wgs * 8));
    CK(hipMemset(a, 0, (size_t)wgs * per_wg * 16)); CK(hipMemset(b, 0, (size_t)wgs * per_wg * 16)); CK(hipMemset(big, 0, nbig * 16));
    auto report = [&](const char* name) -> int {
        std::vector<long long> h(wgs);
        CK(hipMemcpy(h.data(), out, wgs * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("%-46s first-load latency: median %6.0f ns  p10 %6.0f  p90 %6.0f\n", name, h[wgs / 2] * 10.0, h[wgs / 10] * 10.0, h[wgs * 9 / 10] * 10.0);
        return 0;
    };
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(threads), 0, 0, b, sink, per_wg);
        hipLaunchKernelGGL(timed, dim3(wgs), dim3(threads), 0, 0, a, sink, per_wg, out);
        CK(hipDeviceSynchronize()); report("cold (previous kernel touched another buffer)");
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(threads), 0, 0, a, sink, per_wg);
        hipLaunchKernelGGL(timed, dim3(wgs), dim3(threads), 0, 0, a, sink, per_wg, out);
        CK(hipDeviceSynchronize()); report("warm (previous kernel touched the same slice)");
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(threads), 0, 0, a, sink, per_wg);
        hipLaunchKernelGGL(stream_other, dim3(1024), dim3(256), 0, 0, big, sink, nbig);
        hipLaunchKernelGGL(timed, dim3(wgs), dim3(threads), 0, 0, a, sink, per_wg, out);
        CK(hipDeviceSynchronize()); report("warm, then 256 MB streamed (nt) in between");
    }
    return 0;
}
