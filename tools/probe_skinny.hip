// Diagnostic build (never shipped): the skinny GEMM with wall-clock stamps per phase, plus
// launch-floor probes.  hipcc --offload-arch=gfx950 -O3 -DVH_STAMPS tools/probe_skinny.hip -o /tmp/probe
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../valle2_amd/csrc/gemm.hip"

void vh_set_error(const char* fmt, ...) {}
int vh_tuning(int) { return 0; }

__global__ void empty_kernel() {}
__global__ void touch_kernel(const float* x, float* y) { y[threadIdx.x + blockIdx.x * blockDim.x] = x[threadIdx.x + blockIdx.x * blockDim.x] + 1.f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class F> float time_us(F f, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}

int main() {
    float *x, *y; CK(hipMalloc(&x, 1 << 24)); CK(hipMalloc(&y, 1 << 24)); CK(hipMemset(x, 0, 1 << 24));
    printf("empty kernel 1 block            : %6.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }, 2000));
    printf("empty kernel 256 blocks x 1024  : %6.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, 0); }, 2000));
    printf("touch kernel 32 blocks x 512    : %6.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(touch_kernel, dim3(32), dim3(512), 0, 0, x, y); }, 2000));
    printf("touch kernel 1024 blocks x 512  : %6.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(512), 0, 0, x, y); }, 2000));

    const int M = 32;
    struct Case { const char* name; int N, K; bool ln; } cases[] = {
        {"qkv+ln", 1536, 512, true}, {"qkv-noln", 1536, 512, false}, {"ffn1-noln", 2048, 512, false}, {"out-proj", 512, 512, false}, {"ffn1+ln", 2048, 512, true}, {"ffn2", 512, 2048, false}};
    for (auto& c : cases) {
        float *A, *W, *O, *g, *b; long long* dbg;
        CK(hipMalloc(&A, M * c.K * 4)); CK(hipMalloc(&W, (size_t)c.N * c.K * 4)); CK(hipMalloc(&O, M * c.N * 4));
        CK(hipMalloc(&g, c.K * 4)); CK(hipMalloc(&b, c.K * 4));
        const int nblk = (c.N + 15) / 16;
        CK(hipMalloc(&dbg, nblk * 16 * 8 * 8)); CK(hipMemset(dbg, 0, nblk * 16 * 8 * 8));
        std::vector<float> h(std::max((size_t)c.N * c.K, (size_t)M * c.K), 0.01f);
        CK(hipMemcpy(A, h.data(), M * c.K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, h.data(), (size_t)c.N * c.K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(g, h.data(), c.K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), c.K * 4, hipMemcpyHostToDevice));
        GemmArgs a{}; a.A = A; a.lda = c.K; a.W = W; a.out = O; a.ldo = c.N; a.M = M; a.N = c.N; a.K = c.K; a.dbg = nullptr;
        LnFuse ln{c.ln ? g : nullptr, c.ln ? b : nullptr, nullptr, nullptr, 1e-5f};
        float us = time_us([&] { launch_gemm<EPI_PLAIN>("p", a, ln, 0); }, 500);
        a.dbg = dbg;
        launch_gemm<EPI_PLAIN>("p", a, ln, 0);
        CK(hipDeviceSynchronize());
        const int NW = c.K > 1024 ? 16 : 8;
        std::vector<long long> t(nblk * 16 * 8);
        CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = 1LL << 62, tend = 0;
        for (int blk = 0; blk < nblk; ++blk) for (int w = 0; w < NW; ++w) { t0 = std::min(t0, t[(blk * 16 + w) * 8]); tend = std::max(tend, t[(blk * 16 + w) * 8 + 5]); }
        printf("%-9s N=%4d K=%4d: %6.2f us/launch (warm, back-to-back); stamped span first-start..last-end %.2f us (100 MHz clock)\n",
               c.name, c.N, c.K, us, (tend - t0) * 0.01);
        // per-phase medians over waves (relative to the wave's own start) and start skew
        const char* ph[] = {"start", "loads issued", "stats done", "mfma+store LDS", "barrier", "epilogue"};
        for (int k = 0; k < 6; ++k) {
            std::vector<double> v;
            for (int blk = 0; blk < nblk; ++blk) for (int w = 0; w < NW; ++w) v.push_back((t[(blk * 16 + w) * 8 + k] - (k ? t[(blk * 16 + w) * 8] : t0)) * 0.01);
            std::sort(v.begin(), v.end());
            printf("    %-16s median %6.2f us   max %6.2f us%s\n", ph[k], v[v.size() / 2], v.back(), k ? "" : "  (wave start after first wave)");
        }
    }
    return 0;
}
