// Diagnostic (never shipped): does a dependent launch cost less when its code is the SAME code as its
// predecessor's (instruction cache kept warm across a kernel boundary) than when every launch of the chain is a
// different kernel?  If it does, a decode step whose launches were all entries of ONE kernel would be cheaper.
//
// A chain of small dependent kernels (32 workgroups x 256 threads: load a 64 KB activation, reduce, store) with
// ~PAD KB of straight-line code in front of the work, replayed from a hipGraph:
//   (a) the same kernel 64 times;  (b) 8 distinct instantiations (distinct code addresses) in rotation;
//   (c) the same ONE kernel with a runtime `path` argument selecting one of 8 code regions in rotation.
// hipcc --offload-arch=gfx950 -O3 tools/probe_icache.hip -o tools/probe_icache.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ROWS = 32, D = 512;

// straight-line filler the compiler cannot fold away: ~N dependent FMAs on a value that ends up in the result
template <int N>
__device__ __forceinline__ float filler(float v, float k) {
#pragma unroll
    for (int i = 0; i < N; ++i) v = __builtin_fmaf(v, k, 1.0f + (float)i * 1e-7f);
    return v;
}

__device__ __forceinline__ void body(const float* in, float* out, float seed) {
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const float4 a = *reinterpret_cast<const float4*>(in + (size_t)row * D + 4 * (tid & 127));
    float s = (a.x + a.y) + (a.z + a.w) + seed * 1e-30f;
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) out[(size_t)row * D + (tid >> 6)] = s;
    if (tid >= 4) out[(size_t)row * D + tid] = a.x;
}

template <int ID, int PAD>
__global__ __launch_bounds__(256) void k_distinct(const float* in, float* out, float k) {
    const float f = filler<PAD>((float)ID + k, k);          // PAD x 8 bytes of code executed before the work
    body(in, out, f);
}

template <int PAD>
__global__ __launch_bounds__(256) void k_paths(const float* in, float* out, float k, int path) {
    float f;
    switch (path) {                                          // eight code regions inside one kernel
        case 0: f = filler<PAD>(0.f + k, k); break;
        case 1: f = filler<PAD>(1.f + k, k * 1.0001f); break;
        case 2: f = filler<PAD>(2.f + k, k * 1.0002f); break;
        case 3: f = filler<PAD>(3.f + k, k * 1.0003f); break;
        case 4: f = filler<PAD>(4.f + k, k * 1.0004f); break;
        case 5: f = filler<PAD>(5.f + k, k * 1.0005f); break;
        case 6: f = filler<PAD>(6.f + k, k * 1.0006f); break;
        default: f = filler<PAD>(7.f + k, k * 1.0007f); break;
    }
    body(in, out, f);
}

template <int PAD>
static int run(const char* label) {
    float *a, *b;
    CK(hipMalloc(&a, ROWS * D * 4));
    CK(hipMalloc(&b, ROWS * D * 4));
    CK(hipMemset(a, 0, ROWS * D * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int N = 64;
    for (int mode = 0; mode < 3; ++mode) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            const float* in = (i & 1) ? b : a;
            float* out = (i & 1) ? a : b;
            if (mode == 0) {
                hipLaunchKernelGGL((k_distinct<0, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f);
            } else if (mode == 1) {
                switch (i & 7) {
                    case 0: hipLaunchKernelGGL((k_distinct<0, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                    case 1: hipLaunchKernelGGL((k_distinct<1, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                    case 2: hipLaunchKernelGGL((k_distinct<2, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                    case 3: hipLaunchKernelGGL((k_distinct<3, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                    case 4: hipLaunchKernelGGL((k_distinct<4, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                    case 5: hipLaunchKernelGGL((k_distinct<5, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                    case 6: hipLaunchKernelGGL((k_distinct<6, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                    default: hipLaunchKernelGGL((k_distinct<7, PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f); break;
                }
            } else {
                hipLaunchKernelGGL((k_paths<PAD>), dim3(ROWS), dim3(256), 0, s, in, out, 0.5f, i & 7);
            }
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float best = 1e9f;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const char* names[3] = {"same kernel 64x", "8 distinct kernels in rotation", "one kernel, 8 code paths in rotation"};
        printf("%s: %-38s %6.2f us per launch\n", label, names[mode], best * 1e3f / (8 * N));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    CK(hipFree(a));
    CK(hipFree(b));
    return 0;
}

// Second question: what does a dependent launch cost when every workgroup consumes data that workgroups on OTHER XCDs
// wrote in the previous launch (row r is written by workgroup r = XCD r % 8; reading row r + shift crosses XCDs unless
// shift % 8 == 0), and with a grid that fills the chip?
__global__ __launch_bounds__(256) void k_shift(const float* in, float* out, int shift, int rows_mask, int nload) {
    const int tid = threadIdx.x;
    const int row = blockIdx.x & rows_mask;
    float s = 0.f;
    for (int j = 0; j < nload; ++j) {                       // nload dependent-free row reads, all in flight
        const int src = (row + shift + j * 8 * (shift & 7 ? 1 : 1)) & rows_mask;
        const float4 a = *reinterpret_cast<const float4*>(in + (size_t)src * D + 4 * (tid & 127));
        s += (a.x + a.y) + (a.z + a.w);
    }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    out[(size_t)row * D + tid] = s;
    out[(size_t)row * D + 256 + tid] = s;
}

static int run_shift() {
    const int R = 256;
    float *a, *b;
    CK(hipMalloc(&a, R * D * 4));
    CK(hipMalloc(&b, R * D * 4));
    CK(hipMemset(a, 0, R * D * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int N = 64;
    struct Case { const char* name; int grid, shift, nload; } cases[] = {
        {"32 wg, same-XCD rows, 1 row ", 32, 0, 1},   {"32 wg, cross-XCD rows, 1 row", 32, 3, 1},
        {"32 wg, cross-XCD, 8 rows    ", 32, 3, 8},   {"256 wg, same-XCD rows, 1 row", 256, 0, 1},
        {"256 wg, cross-XCD, 1 row    ", 256, 3, 1},  {"256 wg, cross-XCD, 8 rows   ", 256, 3, 8},
        {"256 wg, cross-XCD, 32 rows  ", 256, 3, 32}};
    for (const Case& c : cases) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i)
            hipLaunchKernelGGL(k_shift, dim3(c.grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, c.shift, c.grid - 1,
                               c.nload);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float best = 1e9f;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("dependent chain, %s: %6.2f us per launch\n", c.name, best * 1e3f / (8 * N));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}

int main() {
    if (run_shift()) return 1;
    if (run<64>("0.5 KB of code ")) return 1;
    if (run<512>("4 KB of code   ")) return 1;
    if (run<1024>("8 KB of code   ")) return 1;
    return 0;
}
