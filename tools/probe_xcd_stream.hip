// How fast can the CUs of ONE XCD pull a weight set?  (developer probe for DESIGN.md section 8: the decode chain on one XCD.)
// A launch of 256 workgroups x 1024 threads in which only the workgroups of `nx` XCDs work (workgroup i runs on XCD i % 8):
// each working workgroup streams its contiguous share of a 12.6 MB buffer (one layer's matmul weights) with 8 x 16 bytes
// per lane in flight; reported: the time from the first working workgroup's start to the last one's end.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_xcd_stream.hip -o gpurun_out/probe_xcd_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void stream_kernel(const f4* buf, size_t n16, int nx, float* sink, long long* t) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;         // 32 workgroups per XCD
    if (xcd >= nx) return;
    const int nwork = 32 * nx, me = xcd * 32 + slot;
    const size_t per = n16 / nwork;
    const f4* p = buf + (size_t)me * per;
    const long long t0 = wall_clock64();
    f4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i + 7 * 1024 < per; i += 8 * 1024) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[i + j * 1024];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
    __syncthreads();
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64(); }
}

int main() {
    const size_t bytes = (size_t)12 << 20;                           // one layer: 3 145 728 matmul weights = 12.6 MB; 12 MB here
    const size_t n16 = bytes / 16;
    f4* buf; float* sink; long long* t;
    CK(hipMalloc(&buf, bytes * 8)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&t, 512 * 8));
    CK(hipMemset(buf, 0, bytes * 8));
    for (int nx : {1, 2, 4, 8}) {
        std::vector<double> us;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemset(t, 0, 512 * 8));
            // a different 12 MB region each time: cold in L2
            hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(1024), 0, 0, buf + (size_t)(rep % 8) * n16, n16, nx, sink, t);
            CK(hipDeviceSynchronize());
            long long h[512];
            CK(hipMemcpy(h, t, 512 * 8, hipMemcpyDeviceToHost));
            long long lo = 0, hi = 0;
            for (int i = 0; i < 256; ++i) if (h[2 * i]) { if (!lo || h[2 * i] < lo) lo = h[2 * i]; if (h[2 * i + 1] > hi) hi = h[2 * i + 1]; }
            us.push_back((hi - lo) / 100.0);
        }
        std::sort(us.begin(), us.end());
        printf("%d XCD(s), %3d workgroups: 12 MB in %6.2f us (median of 6) = %5.2f TB/s, %5.1f GB/s per CU\n", nx, 32 * nx, us[3],
               bytes / us[3] / 1e6, bytes / us[3] / 1e3 / (32 * nx));
    }
    return 0;
}
