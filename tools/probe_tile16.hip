// Diagnostic: where does a workgroup of the bf16 tile GEMM (gemm16_tile_kernel) spend its time?  Builds bf16.hip with
// VH_TILE_PROBE16 (six wall-clock stamps per workgroup), runs one launch per shape and prints the mean duration of
// every phase and, for one CU, the timeline of the workgroups that ran on it.  usage: probe_tile16.bin [form = 1 | 3]
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_tile16.hip -o tools/probe_tile16.bin
#define VH_TILE_PROBE16 1
#include "../valle2_amd/csrc/bf16.hip"
#include "../valle2_amd/csrc/gemm16p.hip"      // (bf16.hip's dispatcher refers to the 256^2 form; this probe forces forms 1 / 3)
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

void vh_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
static int g_form = 1;
int vh_tuning(int knob) { return knob == VH_TUNE_BF16_GEMM ? g_form : 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(int M, int N, int K, int mode) {      // mode 0: fp32 out + residual, 1: bf16 out, 2: bf16 out + GELU
    const int blocks = ((M + 127) / 128) * (N / 128);
    uint16_t *A, *W;
    float *R, *O, *bias;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2));
    CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&O, (size_t)M * N * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
    std::vector<uint16_t> h((size_t)std::max(M, N) * K);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3C00 + ((s >> 9) & 0x3FF) + ((s >> 3) & 0x8000)); }   // +-(0.0078 .. 0.0156)
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(R, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, (size_t)N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (vh_linear_bf16(A, K, W, bias, mode == 0 ? R : nullptr, N, O, N, mode != 0, M, N, K, mode == 2 ? VH_ACT_GELU_ERF : VH_ACT_NONE, nullptr) != 0) return 1;
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const char* names[] = {"fp32 out + residual", "bf16 out", "bf16 out + GELU"};
    printf("\nM=%d N=%d K=%d %s: %d workgroups, %.1f us, %.1f TFLOP/s (instrumented)\n", M, N, K, names[mode], blocks, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    const int nb = std::min(blocks, 16384);
    std::vector<long long> tp((size_t)nb * 8);
    std::vector<unsigned> hw(nb * 2);
    CK(hipMemcpyFromSymbol(tp.data(), HIP_SYMBOL(vh_probe16), tp.size() * 8));
    CK(hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(vh_hwid16), hw.size() * 4));
    double ph[5] = {0, 0, 0, 0, 0};
    long long t0 = 1LL << 62, t1 = 0;
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < nb; ++b) {
        const long long* q = &tp[(size_t)b * 8];
        for (int k = 0; k < 5; ++k) ph[k] += (q[k + 1] - q[k]) / 100.0;
        t0 = std::min(t0, q[0]); t1 = std::max(t1, q[5]);
        by_cu[((hw[2 * b + 1] & 15) << 16) | (hw[2 * b] & 0xFF00)].push_back(b);
    }
    printf("kernel spans %.1f us; mean per workgroup: entry->first slab %.2f us | main loop %.2f | accumulators -> LDS image %.2f | "
           "image -> stores issued %.2f | stores acknowledged %.2f | total %.2f us\n",
           (t1 - t0) / 100.0, ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb, ph[4] / nb, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / nb);
    size_t lo = 1 << 30, hi = 0;
    for (auto& kv : by_cu) { lo = std::min(lo, kv.second.size()); hi = std::max(hi, kv.second.size()); }
    printf("%zu distinct CUs, workgroups per CU: min %zu max %zu\n", by_cu.size(), lo, hi);
    auto it = by_cu.begin(); std::advance(it, by_cu.size() / 2);
    std::vector<int> wgs = it->second;
    std::sort(wgs.begin(), wgs.end(), [&](int x, int y) { return tp[(size_t)x * 8] < tp[(size_t)y * 8]; });
    printf("CU %#x (us since kernel start): entry | first slab | loop end | image | issued | acked\n", it->first);
    for (size_t i = 0; i < wgs.size() && i < 14; ++i) {
        const long long* q = &tp[(size_t)wgs[i] * 8];
        printf("  wg %5d:", wgs[i]);
        for (int k = 0; k < 6; ++k) printf(" %7.2f", (q[k] - t0) / 100.0);
        printf("\n");
    }
    // fraction of the kernel's span during which a CU's workgroups are in their main loops (sum over its workgroups / span)
    double inloop = 0;
    for (int b = 0; b < nb; ++b) inloop += (tp[(size_t)b * 8 + 2] - tp[(size_t)b * 8 + 1]) / 100.0;
    printf("sum of main-loop time / (CUs x span) = %.2f (2.0 = both resident workgroups always in their loops)\n",
           inloop / (by_cu.size() * ((t1 - t0) / 100.0)));
    hipFree(A); hipFree(W); hipFree(R); hipFree(O); hipFree(bias);
    return 0;
}

int main(int argc, char** argv) {
    g_form = argc > 1 ? atoi(argv[1]) : 1;      // VH_TUNE_BF16_GEMM: 1 = two slabs of 64 k, 3 = one slab, four workgroups per CU
    printf("VH_TUNE_BF16_GEMM = %d\n", g_form);
    if (run(65536, 1536, 512, 1)) return 1;     // qkv-like
    if (run(65536, 512, 512, 0)) return 1;      // out-projection
    if (run(65536, 2048, 512, 2)) return 1;     // linear_1 + GELU
    if (run(65536, 512, 2048, 0)) return 1;     // linear_2
    return 0;
}
