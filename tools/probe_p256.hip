// Diagnostic: where does a wave of the persistent 256^2 bf16 GEMM (csrc/gemm16p.hip) spend its cycles?  Builds the kernel with
// VH_P256_PROBE (shader-clock sums per phase segment and wave, wall-clock stamps of entry / first phase / exit) and prints, per
// shape, the means over the workgroups for the two wave groups.  The stamps' own waits (lgkmcnt(0) at every stamp) forbid overlaps
// the real kernel has: read the SHARES, not the length (cdna_hip_programming.md section 7, In-kernel stamps).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_p256.hip -o tools/probe_p256.bin
#define VH_P256_PROBE 1
#include "../valle2_amd/csrc/gemm16p.hip"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

void vh_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int vh_tuning(int) { return 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(int M, int N, int K, int mode) {      // mode 0: fp32 out + residual, 1: bf16 out, 2: bf16 out + GELU
    uint16_t *A, *W;
    float *R, *O, *bias;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2));
    CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&O, (size_t)M * N * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
    std::vector<uint16_t> h((size_t)std::max(M, N) * K);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3C00 + ((s >> 9) & 0x3FF) + ((s >> 3) & 0x8000)); }
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(R, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, (size_t)N * 4));
    Gemm16Args a{};
    a.A = A; a.lda = K; a.W = W; a.bias = bias; a.res = mode == 0 ? R : nullptr; a.ldr = N; a.out = O; a.ldo = N;
    a.M = M; a.N = N; a.K = K; a.act = mode == 2 ? VH_ACT_GELU_ERF : VH_ACT_NONE;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (vh_gemm16_p256_launch(a, mode == 0 ? G16_F32 : G16_BF16, nullptr) != 0) return 1;
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const int tiles = ((M + 255) / 256) * (N / 256), nb = std::min(tiles, 256), nk = K / 64;
    const char* names[] = {"fp32 out + residual", "bf16 out", "bf16 out + GELU"};
    printf("\nM=%d N=%d K=%d %s: %d tiles on %d workgroups, %.1f us, %.1f TFLOP/s (instrumented)\n", M, N, K, names[mode], tiles, nb,
           ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    std::vector<unsigned long long> tp((size_t)256 * 8 * 8);
    CK(hipMemcpyFromSymbol(tp.data(), HIP_SYMBOL(vh_p256_probe), tp.size() * 8));
    const double phases = 2.0 * nk * tiles / nb;       // per workgroup (mean)
    for (int grp = 0; grp < 2; ++grp) {
        double sm[6] = {0, 0, 0, 0, 0, 0}, pro = 0, tot = 0;
        int cnt = 0;
        for (int b = 0; b < nb; ++b)
            for (int w = 4 * grp; w < 4 * grp + 4; ++w) {
                const unsigned long long* q = &tp[((size_t)b * 8 + w) * 8];
                for (int i = 0; i < 6; ++i) sm[i] += (double)q[i];
                pro += q[6] / 100.0;
                tot += q[7] / 100.0;
                ++cnt;
            }
        printf("  waves %d-%d: cycles per phase: LDS reads + DMA requests (reads waited for) %.0f | counted vmcnt wait %.0f | wait at barrier A %.0f | MFMA section %.0f | wait at barrier B %.0f"
               " = %.0f (16 MFMAs = 512); epilogue %.0f cycles per tile; prologue %.2f us, whole kernel %.2f us\n",
               4 * grp, 4 * grp + 3, sm[5] / cnt / phases, sm[0] / cnt / phases, sm[1] / cnt / phases, sm[2] / cnt / phases, sm[3] / cnt / phases,
               (sm[0] + sm[1] + sm[2] + sm[3] + sm[5]) / cnt / phases, sm[4] / cnt / (phases / 2 / nk), pro / cnt, tot / cnt);
    }
    hipFree(A); hipFree(W); hipFree(R); hipFree(O); hipFree(bias);
    return 0;
}

int main() {
    if (run(65536, 1536, 512, 1)) return 1;     // qkv-like
    if (run(65536, 512, 512, 0)) return 1;      // out-projection
    if (run(65536, 2048, 512, 2)) return 1;     // linear_1 + GELU
    if (run(65536, 512, 2048, 0)) return 1;     // linear_2
    if (run(65536, 512, 2048, 1)) return 1;     // the same products with a bf16 result (half the epilogue bytes, no residual)
    return 0;
}
