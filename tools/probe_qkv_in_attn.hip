// Diagnostic (never shipped): can the decode attention launch absorb the QKV projection of its (row, head)?
// Each of 256 workgroups (512 threads, one per CU; workgroup i runs on XCD i % 8 = its head) would have to read its
// head's slices of the folded QKV weight — 3 x 64 rows x 512 floats = 384 KB, the SAME bytes for the 32 workgroups of
// an XCD (L2 hits after the first touch) — before it can score a single key, while the K/V stream (656 KB per
// workgroup at context 1280, distinct bytes from HBM) can only run ahead by its register ring.
//   mode 0: the K/V stream alone (ring of 2 sets, like attn_decode_ring_kernel)        -> the attention launch today
//   mode 1: weight slices alone (W_BYTES per workgroup, shared per XCD)                -> the prologue's cost
//   mode 2: ring prefetch issued, then the weight slices, then the stream              -> the fused launch
// hipcc --offload-arch=gfx950 -O3 tools/probe_qkv_in_attn.hip -o tools/probe_qkv_in_attn.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ldnt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }

// kv: [wg][keys][128 floats] (K row + V row), w: [8 heads][w_floats]
__global__ __launch_bounds__(512) void k(const float* __restrict__ kv, int keys, const float* __restrict__ w, int w_floats,
                                         float* __restrict__ out, int mode, long long* stamps) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wg = blockIdx.x;
    const float* base = kv + (size_t)wg * keys * 128;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int chunks = keys / 32;                       // a chunk = 32 keys = 16 KB; a wave-instruction = 4 keys of K or V
    f32x4 ring[2][16];
    auto load = [&](int c, f32x4 (&r)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = ldnt(base + ((size_t)c * 32 * 128) + i * 256 + 4 * lane);
    };
    auto use = [&](const f32x4 (&r)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += r[i];
    };
    if (mode != 1) load(wv, ring[0]);                   // speculative first burst
    const long long t0 = wall_clock64();
    if (mode != 0) {                                    // the weight slices: wave wv reads its eighth, 16 loads in flight
        const float* ws = w + (size_t)(wg % 8) * w_floats + (size_t)wv * (w_floats / 8);
        const int n16 = w_floats / 8 / (64 * 4) / 16;
        for (int it = 0; it < n16; ++it) {
            f32x4 t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = *reinterpret_cast<const f32x4*>(ws + ((size_t)it * 16 + i) * 256 + 4 * lane);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += t[i];
        }
        __syncthreads();                                // q published through LDS
    }
    const long long t1 = wall_clock64();
    if (mode != 1) {
        for (int c0 = wv; c0 < chunks; c0 += 16) {
            if (c0 + 8 < chunks) load(c0 + 8, ring[1]);
            use(ring[0]);
            if (c0 + 8 < chunks) {
                if (c0 + 16 < chunks) load(c0 + 16, ring[0]);
                use(ring[1]);
            }
        }
    }
    const long long t2 = wall_clock64();
    out[(size_t)wg * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
    if (tid == 0) { stamps[wg * 3] = t0; stamps[wg * 3 + 1] = t1; stamps[wg * 3 + 2] = t2; }
}

int main() {
    const int WG = 256, keys = 1280;
    float *kv, *w, *out; long long* st;
    const size_t kvf = (size_t)WG * keys * 128;
    CK(hipMalloc(&kv, kvf * 4 * 2));                      // two copies, alternated, so no launch re-reads cached bytes
    CK(hipMemset(kv, 0, kvf * 4 * 2));
    CK(hipMalloc(&w, 8 * (size_t)(3 * 64 * 512) * 4 * 4));
    CK(hipMemset(w, 0, 8 * (size_t)(3 * 64 * 512) * 4 * 4));
    CK(hipMalloc(&out, WG * 512 * 4));
    CK(hipMalloc(&st, WG * 3 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int mode, w_floats; } cases[] = {
        {"K/V stream alone                      ", 0, 3 * 64 * 512},
        {"weight slices alone, 384 KB (q, k, v)  ", 1, 3 * 64 * 512},
        {"weight slices alone, 128 KB (q only)   ", 1, 64 * 512},
        {"fused: ring + 384 KB slices + stream   ", 2, 3 * 64 * 512},
        {"fused: ring + 128 KB slice + stream    ", 2, 64 * 512}};
    for (const Case& c : cases) {
        float best = 1e9f; std::vector<long long> h(WG * 3);
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k, dim3(WG), dim3(512), 0, s, kv + (rep & 1) * kvf, keys, w + (size_t)(rep & 3) * 8 * (3 * 64 * 512),
                               c.w_floats, out, c.mode, st);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) { best = ms; CK(hipMemcpy(h.data(), st, WG * 3 * 8, hipMemcpyDeviceToHost)); }
        }
        double pa = 0, pb = 0; long long tmin = h[0], tmax = h[2];
        for (int i = 0; i < WG; ++i) { pa += h[i * 3 + 1] - h[i * 3]; pb += h[i * 3 + 2] - h[i * 3 + 1];
            if (h[i * 3] < tmin) tmin = h[i * 3]; if (h[i * 3 + 2] > tmax) tmax = h[i * 3 + 2]; }
        printf("%s: event bracket %6.2f us | per workgroup: prologue %5.2f us, stream %5.2f us | first start -> last end %5.2f us\n",
               c.name, best * 1e3f, pa / WG / 100.0, pb / WG / 100.0, (tmax - tmin) / 100.0);
    }
    return 0;
}
