// Does hipGraph replay run forked branches concurrently on this runtime?  (diagnostic only)
// hipcc --offload-arch=gfx950 -O3 tools/probe_graph_fork.hip -o tools/probe_graph_fork.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 1024) *sink = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    hipEvent_t fork, join, e0, e1;
    CK(hipEventCreate(&fork)); CK(hipEventCreate(&join)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long ticks = 2000;  // 20 us at 100 MHz
    for (int chain = 1; chain <= 4; chain *= 4) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
        for (int i = 0; i < chain; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s0, ticks, nullptr);
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, ticks, nullptr);
        }
        CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fork/join graph, 2 branches x %d spin kernels of 20 us: %.1f us per replay (serial would be %d us, concurrent %d us)\n",
               chain, ms * 1e3 / reps, 40 * chain, 20 * chain);
    }
    return 0;
}
