// Can launches on N different HIP streams make progress while launches on the others spin?  (developer probe)
// Each stream i gets a kernel that waits (bounded, 20 ms) for flag[i] and then sets flag[i-1]; the LAST stream's kernel
// only sets its predecessor's flag.  Host enqueue order: stream 0 first — so every waiting kernel is enqueued before
// the one that releases it, and the chain resolves only if all N streams run concurrently.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_queues.hip -o gpurun_out/probe_queues
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void wait_then_set(unsigned* flags, int i, int n, long long* out) {
    const long long t0 = wall_clock64();
    int ok = 1;
    if (i + 1 < n) {                                   // wait for flag[i] (set by stream i + 1)
        while (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (wall_clock64() - t0 > 2000000) { ok = 0; break; }      // 20 ms at 100 MHz
            __builtin_amdgcn_s_sleep(16);
        }
    }
    if (i > 0) __hip_atomic_store(flags + i - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[2 * i] = ok;
    out[2 * i + 1] = wall_clock64() - t0;
}

int run(const char* name, std::vector<hipStream_t>& st) {
    const int n = (int)st.size();
    unsigned* flags; long long* out;
    CK(hipMalloc(&flags, 64 * 4)); CK(hipMalloc(&out, 64 * 8));
    CK(hipMemset(flags, 0, 64 * 4)); CK(hipMemset(out, 0, 64 * 8));
    CK(hipDeviceSynchronize());
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(wait_then_set, dim3(1), dim3(64), 0, st[i], flags, i, n, out);
    CK(hipDeviceSynchronize());
    long long h[64];
    CK(hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost));
    printf("%-44s", name);
    for (int i = 0; i < n; ++i) printf("  s%d:%s %7.1fus", i, h[2 * i] ? "ok" : "TIMEOUT", h[2 * i + 1] / 100.0);
    printf("\n");
    CK(hipFree(flags)); CK(hipFree(out));
    return 0;
}

int main() {
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("priority range: least %d greatest %d\n", least, greatest);
    for (int rep = 0; rep < 3; ++rep) {
        {   std::vector<hipStream_t> st(2);
            CK(hipStreamCreateWithPriority(&st[0], hipStreamNonBlocking, 0)); CK(hipStreamCreateWithPriority(&st[1], hipStreamNonBlocking, 0));
            run("2 streams, normal/normal", st); }
        {   std::vector<hipStream_t> st(2);
            CK(hipStreamCreateWithPriority(&st[0], hipStreamNonBlocking, least)); CK(hipStreamCreateWithPriority(&st[1], hipStreamNonBlocking, 0));
            run("2 streams, low waits for normal", st); }
        {   std::vector<hipStream_t> st(3);
            for (auto& s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0));
            run("3 streams, all normal", st); }
        {   std::vector<hipStream_t> st(3);
            CK(hipStreamCreateWithPriority(&st[0], hipStreamNonBlocking, least)); CK(hipStreamCreateWithPriority(&st[1], hipStreamNonBlocking, greatest));
            CK(hipStreamCreateWithPriority(&st[2], hipStreamNonBlocking, 0));
            run("3 streams, low <- high <- normal", st); }
        {   std::vector<hipStream_t> st(3);
            CK(hipStreamCreateWithPriority(&st[0], hipStreamNonBlocking, least)); CK(hipStreamCreateWithPriority(&st[1], hipStreamNonBlocking, 0));
            CK(hipStreamCreateWithPriority(&st[2], hipStreamNonBlocking, greatest));
            run("3 streams, low <- normal <- high", st); }
        {   std::vector<hipStream_t> st(4);
            for (auto& s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0));
            run("4 streams, all normal", st); }
        {   std::vector<hipStream_t> st(6);
            for (auto& s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0));
            run("6 streams, all normal", st); }
    }
    return 0;
}
