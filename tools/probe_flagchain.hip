// Diagnostic (never shipped): does overlapping a kernel's ramp (dispatch, cold instruction fetch,
// kernarg fetch, weight loads) with its predecessor pay on MI355X?  A chain of small "GEMM-like"
// kernels, each consuming the whole 64 KB output of its predecessor, is replayed from a hipGraph
//   (a) on one stream, dependencies = kernel boundaries (what the decoder does today);
//   (b) alternating over two streams with no edge between neighbours: kernel k+1 starts while k runs,
//       prefetches its private weights, then waits on k's arrival counters (release/acquire at agent
//       scope, cdna_hip_programming.md Guideline 16), so at most two kernels are in flight.
// hipcc --offload-arch=gfx950 -O3 tools/probe_flagchain.hip -o tools/probe_flagchain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ROWS = 32, D = 512;          // activation: 32 x 512 floats = 64 KB
constexpr int NSHARD = 8;                  // arrival counters per kernel (blockIdx % 8)
constexpr int SPIN_LIMIT = 200000;

struct ChainArgs {
    const float* in;        // [ROWS][D] from the predecessor
    float* out;             // [ROWS][D]
    const float* w;         // private weights: nblk * 16 cols * D floats
    unsigned* my_cnt;       // [NSHARD] arrival counters of this kernel
    const unsigned* prev_cnt;  // [NSHARD] of the predecessor (nullptr: no flag wait)
    const unsigned* epoch;  // replay number (1-based), bumped once per replay
    unsigned* err;
    int prev_blocks;        // predecessor grid size
    int use_flags;
    float bias;
};

// 32 blocks x 512 threads; block b produces columns 16b..16b+15 of all 32 rows:
// out[r][c] = sum_k in[r][k] * w[c][k] (scaled) — FMA form, the point here is the dependency chain.
__global__ __launch_bounds__(512) void chain_kernel(ChainArgs a) {
    __shared__ float part[8][ROWS][16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, kq = lane >> 4;                 // column, k quarter of the wave's slice
    // ---- prologue that needs no predecessor data: this wave's weights, K slice [64*wave, 64*wave+64)
    const float* wp = a.w + ((size_t)(blockIdx.x * 16 + c) * D + wave * 64 + kq * 16);
    float4 wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = *reinterpret_cast<const float4*>(wp + 4 * i);

    if (a.use_flags && a.prev_cnt) {
        if (tid < NSHARD) {
            const unsigned ep = *a.epoch;
            // blocks b with b % 8 == tid arrive on shard tid
            const unsigned target = ep * (unsigned)((a.prev_blocks - tid + NSHARD - 1) / NSHARD);
            int spins = 0;
            while (__hip_atomic_load(a.prev_cnt + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { *a.err = 1; break; }
            }
            if (a.use_flags & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // ---- body: consume the predecessor's whole output (8 float4 per thread = 64 KB per block)
    float s = 0.f;
    {
        float4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = reinterpret_cast<const float4*>(a.in)[tid + 512 * i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 ww = wv[i & 3];
            s += x[i].x * ww.x + x[i].y * ww.y + x[i].z * ww.z + x[i].w * ww.w;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) part[0][0][wave] = s;
    __syncthreads();
    {
        float tot = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) tot += part[0][0][w8];
        const int r = tid >> 4, cc = tid & 15;
        a.out[(size_t)r * D + (blockIdx.x * 16 + cc) % D] = 0.5f * tot / ROWS + a.bias;
    }
    if (a.use_flags) {
        __syncthreads();   // every wave's stores issued and waited (vmcnt(0)) before the release
        if (tid == 0) {
            if (a.use_flags & 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(a.my_cnt + (blockIdx.x % NSHARD), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void bump_epoch(unsigned* epoch) { *epoch += 1; }
__global__ void fill_kernel(float* p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }

int main() {
    const int NK = 60;                      // kernels per replay
    const int grids[5] = {96, 32, 128, 128, 32};
    float *act[2], *w; unsigned *cnt, *epoch, *err;
    CK(hipMalloc(&act[0], ROWS * D * 4)); CK(hipMalloc(&act[1], ROWS * D * 4));
    CK(hipMemset(act[0], 0, ROWS * D * 4)); CK(hipMemset(act[1], 0, ROWS * D * 4));
    const size_t wbytes = (size_t)128 * 16 * D * 4;   // 4 MB per kernel
    CK(hipMalloc(&w, wbytes * NK));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, w, wbytes / 4 * NK, 1.0f / D);
    CK(hipDeviceSynchronize());
    CK(hipMalloc(&cnt, NK * NSHARD * 4)); CK(hipMalloc(&epoch, 4)); CK(hipMalloc(&err, 4));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    hipEvent_t fork, join, e0, e1;
    CK(hipEventCreate(&fork)); CK(hipEventCreate(&join)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    const int streams_of[4] = {1, 1, 2, 2}, flags_of[4] = {0, 3, 1, 3};
    const char* names[4] = {"1 stream, boundaries only        ", "1 stream, counters + fences      ", "2 streams, counters, NO fences   ", "2 streams, counters + fences     "};
    for (int mode = 0; mode < 4; ++mode) {
        const int two = streams_of[mode] == 2;
        CK(hipMemset(act[0], 0, ROWS * D * 4)); CK(hipMemset(act[1], 0, ROWS * D * 4));
        CK(hipMemset(cnt, 0, NK * NSHARD * 4)); CK(hipMemset(epoch, 0, 4)); CK(hipMemset(err, 0, 4));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(bump_epoch, dim3(1), dim3(1), 0, sa, epoch);
        if (two) { CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0)); }
        for (int k = 0; k < NK; ++k) {
            ChainArgs a{};
            a.in = act[k & 1]; a.out = act[(k + 1) & 1]; a.w = w + (wbytes / 4) * k;
            a.my_cnt = cnt + k * NSHARD; a.prev_cnt = k ? cnt + (k - 1) * NSHARD : nullptr;
            a.epoch = epoch; a.err = err; a.prev_blocks = k ? grids[(k - 1) % 5] : 0; a.use_flags = flags_of[mode]; a.bias = (float)(k + 1);
            hipStream_t s = (two && (k & 1)) ? sb : sa;
            hipLaunchKernelGGL(chain_kernel, dim3(grids[k % 5]), dim3(512), 0, s, a);
        }
        if (two) { CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0)); }
        CK(hipStreamEndCapture(sa, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, sa));
        CK(hipStreamSynchronize(sa));
        const int reps = 50;
        CK(hipEventRecord(e0, sa));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, sa));
        CK(hipEventRecord(e1, sa));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        std::vector<float> h(ROWS * D);
        CK(hipMemcpy(h.data(), act[NK & 1], ROWS * D * 4, hipMemcpyDeviceToHost));
        double sum = 0; for (float v : h) sum += v;
        printf("%s: %.2f us per replay of %d kernels = %.2f us per kernel   (spin-timeout flag %u, checksum %.6f)\n",
               names[mode], ms * 1e3 / reps, NK,
               ms * 1e3 / reps / NK, herr, sum);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
