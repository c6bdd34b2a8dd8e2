// The staged decoder's launch pattern in miniature (developer probe): main M, second GEMM stream H, attention stream L.
//   fork: record ev on M; L and H wait for it.
//   L: KL  waits f_q, sets f_a              (attention)
//   H: KH  waits f_1, sets f_2              (stage 2)
//   M: KM0 sets f_q; KM1 waits f_a, sets f_1 (stage 1); KM3 waits f_2 (stage 3)
// host enqueue order L, H, M — as the decoder does.  M = the NULL stream or a created one.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k(unsigned* flags, int wait_i, int set_i, long long* out, int slot, long long t_ref) {
    const long long t0 = wall_clock64();
    int ok = 1;
    if (wait_i >= 0)
        while (__hip_atomic_load(flags + wait_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (wall_clock64() - t0 > 2000000) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(16);
        }
    if (set_i >= 0) __hip_atomic_store(flags + set_i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[3 * slot] = ok; out[3 * slot + 1] = t0; out[3 * slot + 2] = wall_clock64();
}

int run(const char* name, hipStream_t M, int prioH, int prioL, int blocks) {
    hipStream_t H, L; hipEvent_t ev;
    CK(hipStreamCreateWithPriority(&H, hipStreamNonBlocking, prioH));
    CK(hipStreamCreateWithPriority(&L, hipStreamNonBlocking, prioL));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    unsigned* f; long long* out;
    CK(hipMalloc(&f, 64 * 4)); CK(hipMalloc(&out, 64 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(f, 0, 64 * 4)); CK(hipMemset(out, 0, 64 * 8));
        CK(hipDeviceSynchronize());
        enum { FQ, FA, F1, F2 };
        CK(hipEventRecord(ev, M));
        CK(hipStreamWaitEvent(L, ev, 0)); CK(hipStreamWaitEvent(H, ev, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, L, f, FQ, FA, out, 0, 0);   // KL
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, H, f, F1, F2, out, 1, 0);   // KH
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, M, f, -1, FQ, out, 2, 0);   // KM0
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, M, f, FA, F1, out, 3, 0);   // KM1
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, M, f, F2, -1, out, 4, 0);   // KM3
        CK(hipDeviceSynchronize());
        long long h[64];
        CK(hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost));
        long long t0 = h[1];
        for (int i = 1; i < 5; ++i) if (h[3 * i + 1] < t0) t0 = h[3 * i + 1];
        const char* nm[5] = {"KL", "KH", "KM0", "KM1", "KM3"};
        printf("%-40s", name);
        for (int i = 0; i < 5; ++i) printf(" %s:%s[%.0f..%.0f]", nm[i], h[3 * i] ? "ok" : "TIMEOUT", (h[3 * i + 1] - t0) / 100.0, (h[3 * i + 2] - t0) / 100.0);
        printf("\n");
    }
    CK(hipStreamDestroy(H)); CK(hipStreamDestroy(L)); CK(hipEventDestroy(ev)); CK(hipFree(f)); CK(hipFree(out));
    return 0;
}

int main() {
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t M;
    CK(hipStreamCreateWithPriority(&M, hipStreamNonBlocking, 0));
    run("M created, H high, L low, 1 block", M, greatest, least, 1);
    run("M null,    H high, L low, 1 block", nullptr, greatest, least, 1);
    run("M null,    H normal, L low, 1 block", nullptr, 0, least, 1);
    run("M created, H high, L low, 256 blocks", M, greatest, least, 256);
    run("M null,    H high, L low, 256 blocks", nullptr, greatest, least, 256);
    run("M null,    H normal, L low, 256 blocks", nullptr, 0, least, 256);
    return 0;
}
