// Diagnostic: where does a workgroup of the LDS-DMA tile GEMM spend its time?  Builds gemm.hip with
// VH_TILE_PROBE (clock stamps per wave and K step), runs one launch and prints, for one CU, the timeline of every
// workgroup that ran on it: per step the MFMA phase and the time spent in the barrier.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_tile_timeline.hip -o tools/probe_tile_timeline.bin
#define VH_TILE_PROBE 1
#include "../valle2_amd/csrc/gemm.hip"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

void vh_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int vh_tuning(int knob) { return knob == VH_TUNE_TILE_DMA ? 2 : 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    const int nk = K / 32, blocks = (M / 128) * (N / 128);
    float *A, *W, *O;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&O, (size_t)M * N * 4));
    std::vector<float> h((size_t)M * K);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; }
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (vh_linear(A, K, W, nullptr, nullptr, 0, O, N, M, N, K, 0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr) != 0) return 1;
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("M=%d N=%d K=%d: %d workgroups, %.1f us, %.1f TFLOP/s (instrumented)\n", M, N, K, blocks, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    const int nb = std::min(blocks, 8192);
    std::vector<long long> tp((size_t)nb * 4 * 200);
    std::vector<unsigned> hw(nb * 2);
    CK(hipMemcpyFromSymbol(tp.data(), HIP_SYMBOL(vh_tile_probe), tp.size() * 8));
    CK(hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(vh_tile_hwid), hw.size() * 4));
    // HW_ID: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se ; XCC_ID [3:0]
    std::map<unsigned, std::vector<int>> by_cu;
    long long t_min = 1LL << 62, t_max = 0, w_min = 1LL << 62, w_max = 0;
    for (int b = 0; b < nb; ++b) {
        const unsigned id = ((hw[2 * b + 1] & 15) << 16) | (hw[2 * b] & 0xFF00);
        by_cu[id].push_back(b);
        const long long* q = &tp[(size_t)b * 4 * 200];
        if (q[196] > 0) t_min = std::min(t_min, q[196]); t_max = std::max(t_max, q[194]);
        w_min = std::min(w_min, q[193]); w_max = std::max(w_max, q[195]);
    }
    const double clk_per_wall = double(t_max - t_min) / double(w_max - w_min);
    printf("%zu distinct CUs; kernel spans %lld clock64 ticks = %lld wall ticks (100 MHz -> %.1f us; clock64 runs at %.1f MHz)\n",
           by_cu.size(), t_max - t_min, w_max - w_min, (w_max - w_min) / 100.0, clk_per_wall * 100.0);
    size_t lo = 1 << 30, hi = 0;
    for (auto& kv : by_cu) { lo = std::min(lo, kv.second.size()); hi = std::max(hi, kv.second.size()); }
    printf("workgroups per CU: min %zu max %zu\n", lo, hi);
    // aggregate over all recorded workgroups: per step MFMA phase (start -> barrier entry), barrier wait, and the
    // pre-loop / post-loop parts, in clock64 ticks, wave 0..3 averaged
    double mf = 0, bw = 0, st = 0; long long cnt = 0;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) {
            const long long* q = &tp[((size_t)b * 4 + w) * 200];
            for (int k = 0; k + 1 < nk; ++k) { mf += q[64 + k] - q[k]; bw += q[128 + k] - q[64 + k]; st += q[k + 1] - q[k]; ++cnt; }
        }
    printf("mean per step: %.0f ticks  (start->barrier %.0f, in barrier %.0f)\n", st / cnt, mf / cnt, bw / cnt);
    // one CU in detail
    auto it = by_cu.begin(); std::advance(it, by_cu.size() / 2);
    std::vector<int> wgs = it->second;
    std::sort(wgs.begin(), wgs.end(), [&](int x, int y) { return tp[(size_t)x * 800 + 196] < tp[(size_t)y * 800 + 196]; });
    printf("CU %#x ran %zu workgroups; times in ticks relative to kernel start\n", it->first, wgs.size());
    {   // lifetime breakdown over all workgroups, and the hand-over gap on this CU
        double pro = 0, loop = 0, epi = 0;
        for (int b = 0; b < nb; ++b) { const long long* q = &tp[(size_t)b * 800]; pro += q[0] - q[196]; loop += q[194] - q[0]; epi += q[197] - q[194]; }
        printf("mean per workgroup: entry->loop %.0f, loop %.0f, loop-end->exit %.0f ticks\n", pro / nb, loop / nb, epi / nb);
        std::vector<long long> exits;
        for (int b : wgs) { long long e = 0; for (int w = 0; w < 4; ++w) e = std::max(e, tp[((size_t)b * 4 + w) * 200 + 197]); exits.push_back(e); }
        std::sort(exits.begin(), exits.end());
        printf("hand-over on this CU (k-th entry minus (k-2)-th exit):");
        for (size_t k = 2; k < wgs.size(); ++k) printf(" %lld", tp[(size_t)wgs[k] * 800 + 196] - exits[k - 2]);
        printf("\n");
    }
    for (int b : wgs) {
        const long long* q = &tp[(size_t)b * 800];
        printf(" wg %4d: entry %8lld loop %8lld..%8lld exit %8lld (%.0f ticks/step)  barrier waits w0:", b,
               q[196] - t_min, q[0] - t_min, q[194] - t_min, q[197] - t_min, double(q[194] - q[0]) / nk);
        for (int k = 0; k + 1 < nk && k < 6; ++k) printf(" %lld", q[128 + k] - q[64 + k]);
        printf("\n");
    }
    return 0;
}
