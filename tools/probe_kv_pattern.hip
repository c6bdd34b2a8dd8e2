// Diagnostic (never shipped): which property of the decode-attention access pattern costs bandwidth?  256 workgroups x
// 1024 threads, each workgroup reads its own K and V streams of S x 256 B inside (S_max x 256 B)-strided caches, 8 + 8
// non-temporal 16-byte loads per lane in flight, no arithmetic but adds.  Patterns:
//   0  as attn_decode_kernel: wave w owns 32-key chunks w, w+16, ...; a lane's 8 loads are 1 KB apart (4 keys)
//   1  a lane's 8 loads 16 KB apart: the 16 waves together sweep 64 keys per load instruction
//   2  K then V of the same chunk issued as 16 loads over one 16-key range each ... (K and V bursts of 8 KB)
//   3  pattern 1 with V skipped (one stream per workgroup, half the bytes)
// hipcc --offload-arch=gfx950 -O3 tools/probe_kv_pattern.hip -o tools/probe_kv_pattern.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(1024) void kv_kernel(const float* __restrict__ kc, const float* __restrict__ vc, int S, int S_max,
                                                  float* out) {
    const int bh = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, g = lane >> 4;
    const float* kb = kc + (size_t)bh * S_max * 64 + 4 * c16;
    const float* vb = vc + (size_t)bh * S_max * 64 + 4 * c16;
    f32x4 acc = {0, 0, 0, 0};
    if (P == 0) {
        for (int c = w; c * 32 < S; c += 16) {
            f32x4 kf[8], vf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int key = min(c * 32 + g + 4 * i, S - 1);
                kf[i] = __builtin_nontemporal_load((const f32x4*)(kb + (size_t)key * 64));
                vf[i] = __builtin_nontemporal_load((const f32x4*)(vb + (size_t)key * 64));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += kf[i] + vf[i];
        }
    } else if (P == 1 || P == 3) {
        for (int j = 0; j * 512 < S; ++j) {
            f32x4 kf[8], vf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int key = min(j * 512 + i * 64 + w * 4 + g, S - 1);
                kf[i] = __builtin_nontemporal_load((const f32x4*)(kb + (size_t)key * 64));
                if (P == 1) vf[i] = __builtin_nontemporal_load((const f32x4*)(vb + (size_t)key * 64));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += P == 1 ? kf[i] + vf[i] : kf[i];
        }
    } else {
        // 2: 16 loads over K only for a 1024-key super-chunk, then 16 over V (bursts alternate between the two streams)
        for (int j = 0; j * 1024 < S; ++j) {
            f32x4 kf[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = min(j * 1024 + i * 64 + w * 4 + g, S - 1);
                kf[i] = __builtin_nontemporal_load((const f32x4*)(kb + (size_t)key * 64));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += kf[i];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = min(j * 1024 + i * 64 + w * 4 + g, S - 1);
                kf[i] = __builtin_nontemporal_load((const f32x4*)(vb + (size_t)key * 64));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += kf[i];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

int main(int argc, char** argv) {
    const int BH = 256, NL = 12;
    const int S_max = argc > 1 ? atoi(argv[1]) : 1536;
    const size_t bytes = (size_t)BH * S_max * 256;
    printf("S_max = %d: stream stride %zu B = 0x%zx\n", S_max, (size_t)S_max * 256, (size_t)S_max * 256);
    float *kc[NL], *vc[NL], *out;
    for (int i = 0; i < NL; ++i) {
        CK(hipMalloc(&kc[i], bytes)); CK(hipMalloc(&vc[i], bytes));
        CK(hipMemset(kc[i], 0x3c, bytes)); CK(hipMemset(vc[i], 0x3c, bytes));
    }
    CK(hipMalloc(&out, 4));
    hipEvent_t k0, k1;
    CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    for (int S : {1024, 1280}) {
#define RUN(P, name)                                                                                               \
    do {                                                                                                           \
        double sum = 0;                                                                                            \
        for (int it = 0; it < 28; ++it) {                                                                          \
            hipExtLaunchKernelGGL((kv_kernel<P>), dim3(BH), dim3(1024), 0, 0, k0, k1, 0, kc[it % NL], vc[it % NL], S, S_max, out); \
            CK(hipEventSynchronize(k1));                                                                           \
            float ms;                                                                                              \
            CK(hipEventElapsedTime(&ms, k0, k1));                                                                  \
            if (it >= 4) sum += ms;                                                                                \
        }                                                                                                          \
        const double us = sum / 24 * 1e3, mb = (P == 3 ? 1.0 : 2.0) * BH * S * 256 / 1e6;                           \
        printf("S=%4d %-58s %6.2f us  %6.1f MB  %5.2f TB/s\n", S, name, us, mb, mb / us);  \
    } while (0)
        RUN(0, "0 attention-like (lane loads 1 KB apart, K+V)");
        RUN(1, "1 lane loads 16 KB apart, K+V");
        RUN(2, "2 16-load bursts alternating K / V");
        RUN(3, "3 pattern 1, K only");
    }
    return 0;
}
