// The decode GEMM chain of one layer for 16 rows on the 32 CUs of ONE XCD (developer probe for DESIGN.md section 8).
// 256 workgroups are launched, the 32 on XCD 0 (workgroup i runs on XCD i % 8) work: they first touch all the weight
// rows they will need (LDS-DMA dump loads: the lines land in this XCD's L2), then run the stages with the library's own
// skinny_body — out-projection (1 column block per workgroup), LN2 + linear_1 + GELU (4), linear_2 over the whole
// K = 2048 (1, no split-K), LN1 + QKV of the next layer (3) — separated by intra-XCD counter barriers (+ an L1
// invalidate: the activations were written by other CUs of the same XCD and sit in its L2).  Results are compared with
// the library's ordinary launches; the time is taken inside the kernel (wall clock, first start to last end).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I valle2_amd/csrc tools/probe_chain_onexcd.hip -o gpurun_out/probe_chain_onexcd
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
void vh_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int vh_tuning(int) { return 0; }
#include "../valle2_amd/csrc/gemm.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct XArgs {
    const float* attn; float* x; float* q; float* xm; float* hid;
    const float *wo, *bo, *w1f, *w1c1, *w1c2, *w2, *b2, *wqf, *qc1, *qc2;
    float *kc, *vc; const int32_t* cache_len;
    int M, d, dff, n_heads, S_max; float eps;
    unsigned* bar; long long* stamps; int prefetch;
};

__device__ __forceinline__ void xcd_barrier(unsigned* ctr, unsigned* err) {
    __builtin_amdgcn_s_waitcnt(0);                       // this wave's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u) {
            if (++spins > 2000000) { atomicCAS(err, 0u, 0xB0000000u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");         // drop this CU's L1 lines: the inputs of the next stage are in the L2
}

__global__ __launch_bounds__(512) void chain_onexcd_kernel(XArgs c) {
    if (blockIdx.x & 7) return;
    const int j = blockIdx.x >> 3, tid = threadIdx.x, w = tid >> 6;
    __shared__ float dump[8 * 64];
    long long* st = c.stamps + j * 8;
    if (tid == 0) st[0] = wall_clock64();
    unsigned* err = c.bar + 8;
    if (c.prefetch) {                                    // every weight line of this workgroup's items, once, into the L2
        const char* base[4] = {(const char*)(c.wo + (size_t)16 * j * c.d), (const char*)(c.w1f + (size_t)64 * j * c.d),
                               (const char*)(c.w2 + (size_t)16 * j * c.dff), (const char*)(c.wqf + (size_t)48 * j * c.d)};
        const int lines[4] = {16 * c.d / 16, 64 * c.d / 16, 16 * c.dff / 16, 48 * c.d / 16};
#pragma unroll
        for (int m = 0; m < 4; ++m)
            for (int ln = tid; ln < lines[m]; ln += 512)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base[m] + (size_t)ln * 64),
                                                 (__attribute__((address_space(3))) void*)(dump + w * 64), 4, 0, 0);
    }
    const LnFuse none{};
    {   // S1
        GemmArgs a{};
        a.A = c.attn; a.lda = c.d; a.W = c.wo; a.bias = c.bo; a.res = c.x; a.ldr = c.d; a.out = c.xm; a.ldo = c.d;
        a.M = c.M; a.N = c.d; a.K = c.d; a.k_len = c.d; a.act = VH_ACT_NONE;
        skinny_body<1, 8, EPI_PLAIN, 4, 0, 1, 0, 0>(a, none, VB{j, 0, 0, 1, 1});
    }
    if (tid == 0) st[1] = wall_clock64();
    xcd_barrier(c.bar + 0, err);
    if (tid == 0) st[2] = wall_clock64();
    {   // S2
        GemmArgs a{};
        a.A = c.xm; a.lda = c.d; a.W = c.w1f; a.out = c.hid; a.ldo = c.dff;
        a.M = c.M; a.N = c.dff; a.K = c.d; a.k_len = c.d; a.act = VH_ACT_GELU_ERF;
        LnFuse ln{nullptr, nullptr, nullptr, nullptr, c.eps, c.w1c1, c.w1c2};
        for (int b = 0; b < 4; ++b) {
            skinny_body<1, 8, EPI_PLAIN, 4, 2, 8, 0, 0>(a, ln, VB{4 * j + b, 0, 0, 1, 1});
            __syncthreads();
        }
    }
    if (tid == 0) st[3] = wall_clock64();
    xcd_barrier(c.bar + 1, err);
    {   // S3: the whole K, no split
        GemmArgs a{};
        a.A = c.hid; a.lda = c.dff; a.W = c.w2; a.bias = c.b2; a.res = c.xm; a.ldr = c.d; a.out = c.x; a.ldo = c.d;
        a.M = c.M; a.N = c.d; a.K = c.dff; a.k_len = c.dff; a.act = VH_ACT_NONE;
        skinny_body<1, 8, EPI_PLAIN, 4, 0, 1, 0, 0>(a, none, VB{j, 0, 0, 1, 1});
    }
    if (tid == 0) st[4] = wall_clock64();
    xcd_barrier(c.bar + 2, err);
    {   // S5
        GemmArgs a{};
        a.A = c.x; a.lda = c.d; a.W = c.wqf; a.out = c.q; a.ldo = c.d;
        a.M = c.M; a.N = 3 * c.d; a.K = c.d; a.k_len = c.d; a.act = VH_ACT_NONE;
        a.kc = c.kc; a.vc = c.vc; a.cache_len = c.cache_len; a.T = 1; a.S_max = c.S_max; a.d_model = c.d; a.n_heads = c.n_heads;
        LnFuse ln{nullptr, nullptr, nullptr, nullptr, c.eps, c.qc1, c.qc2};
        for (int b = 0; b < 3; ++b) {
            skinny_body<1, 8, EPI_QKV, 4, 2, 8, 0, 0>(a, ln, VB{3 * j + b, 0, 0, 1, 1});
            __syncthreads();
        }
    }
    if (tid == 0) st[5] = wall_clock64();
}

static std::vector<float> rnd(size_t n, float s, unsigned seed) {
    std::vector<float> v(n); unsigned x = seed * 2654435761u + 12345u;
    for (auto& f : v) { x = x * 1664525u + 1013904223u; f = s * ((int)(x >> 9) / 4194304.0f - 1.0f); }
    return v;
}
template <class T> static T* dev(const std::vector<T>& h) { T* p; hipMalloc(&p, h.size() * sizeof(T)); hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return p; }

int main() {
    const int M = 16, d = 512, dff = 2048, h = 8, S = 64;
    float* attn = dev(rnd((size_t)M * d, 1.f, 1)); auto hx = rnd((size_t)M * d, 1.f, 2);
    float *x = dev(hx), *x_ref = dev(hx);
    float *q, *q_ref, *xm, *xm_ref, *hid, *hid_ref;
    CK(hipMalloc(&q, M * d * 4)); CK(hipMalloc(&q_ref, M * d * 4)); CK(hipMalloc(&xm, M * d * 4)); CK(hipMalloc(&xm_ref, M * d * 4));
    CK(hipMalloc(&hid, M * dff * 4)); CK(hipMalloc(&hid_ref, M * dff * 4));
    const int L = 12;                                     // rotate through 12 weight sets: nothing stays in a cache by luck
    std::vector<float*> wo(L), w1f(L), c11(L), c12(L), w2(L), wqf(L), cq1(L), cq2(L);
    float *bo = dev(rnd(d, .1f, 3)), *b1 = dev(rnd(dff, .1f, 4)), *b2 = dev(rnd(d, .1f, 5)), *g = dev(std::vector<float>(d, 1.f)), *be = dev(std::vector<float>(d, 0.f));
    for (int l = 0; l < L; ++l) {
        wo[l] = dev(rnd((size_t)d * d, .04f, 10 + l)); w2[l] = dev(rnd((size_t)d * dff, .02f, 30 + l));
        float* w1 = dev(rnd((size_t)dff * d, .04f, 50 + l)); float* wq = dev(rnd((size_t)3 * d * d, .04f, 70 + l));
        CK(hipMalloc(&w1f[l], (size_t)dff * d * 4)); CK(hipMalloc(&c11[l], dff * 4)); CK(hipMalloc(&c12[l], dff * 4));
        CK(hipMalloc(&wqf[l], (size_t)3 * d * d * 4)); CK(hipMalloc(&cq1[l], 3 * d * 4)); CK(hipMalloc(&cq2[l], 3 * d * 4));
        if (vh_ln_fold(w1, g, be, b1, w1f[l], c11[l], c12[l], dff, d, nullptr)) return 1;
        if (vh_ln_fold(wq, g, be, nullptr, wqf[l], cq1[l], cq2[l], 3 * d, d, nullptr)) return 1;
    }
    float *kc, *vc, *kc_ref, *vc_ref; int32_t* cl;
    CK(hipMalloc(&kc, (size_t)M * h * S * 64 * 4)); CK(hipMalloc(&vc, (size_t)M * h * S * 64 * 4));
    CK(hipMalloc(&kc_ref, (size_t)M * h * S * 64 * 4)); CK(hipMalloc(&vc_ref, (size_t)M * h * S * 64 * 4));
    CK(hipMemset(kc, 0, (size_t)M * h * S * 64 * 4)); CK(hipMemset(vc, 0, (size_t)M * h * S * 64 * 4));
    CK(hipMemset(kc_ref, 0, (size_t)M * h * S * 64 * 4)); CK(hipMemset(vc_ref, 0, (size_t)M * h * S * 64 * 4));
    cl = dev(std::vector<int32_t>(M, 3));
    unsigned* bar; long long* stamps; void* ws;
    CK(hipMalloc(&bar, 64)); CK(hipMalloc(&stamps, 32 * 8 * 8)); CK(hipMalloc(&ws, vh_linear_ws_bytes(M, d, dff) + 64));
    CK(hipMemset(ws, 0, vh_linear_ws_bytes(M, d, dff) + 64));
    const int IT = 96;
    unsigned* bars; long long* stampsN; float* x0 = dev(hx);
    CK(hipMalloc(&bars, IT * 64)); CK(hipMalloc(&stampsN, (size_t)IT * 32 * 8 * 8));
    for (int pf = 0; pf < 2; ++pf) {
        // all launches back to back, no host synchronisation in between (an idle GPU drops its clocks)
        CK(hipMemset(bars, 0, IT * 64)); CK(hipMemset(stampsN, 0, (size_t)IT * 32 * 8 * 8));
        CK(hipDeviceSynchronize());
        for (int it = 0; it < IT; ++it) {
            const int l = it % L;
            CK(hipMemcpyAsync(x, x0, M * d * 4, hipMemcpyDeviceToDevice, 0));
            XArgs c{attn, x, q, xm, hid, wo[l], bo, w1f[l], c11[l], c12[l], w2[l], b2, wqf[l], cq1[l], cq2[l], kc, vc, cl,
                    M, d, dff, h, S, 1e-5f, bars + 16 * it, stampsN + (size_t)it * 32 * 8, pf};
            hipLaunchKernelGGL(chain_onexcd_kernel, dim3(256), dim3(512), 0, 0, c);
        }
        CK(hipDeviceSynchronize());
        std::vector<long long> hs((size_t)IT * 32 * 8); std::vector<unsigned> hb(IT * 16);
        CK(hipMemcpy(hs.data(), stampsN, hs.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), bars, IT * 64, hipMemcpyDeviceToHost));
        std::vector<double> dur, s1, b0, s2, s3, s5;
        for (int it = 24; it < IT; ++it) {
            if (hb[16 * it + 8]) { printf("barrier timed out (%#x)\n", hb[16 * it + 8]); return 2; }
            const long long* t = hs.data() + (size_t)it * 32 * 8;
            long long t0 = t[0], t1 = 0;
            for (int j = 0; j < 32; ++j) { t0 = std::min(t0, t[8 * j]); t1 = std::max(t1, t[8 * j + 5]); }
            dur.push_back((t1 - t0) / 100.0);
            s1.push_back((t[1] - t[0]) / 100.0); b0.push_back((t[2] - t[1]) / 100.0); s2.push_back((t[3] - t[2]) / 100.0);
            s3.push_back((t[4] - t[3]) / 100.0); s5.push_back((t[5] - t[4]) / 100.0);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("%s: chain for %d rows on one XCD: %.2f us (median of %zu back-to-back launches); workgroup 0: S1 %.2f, barrier %.2f, "
               "S2 (4 items) %.2f, barrier + S3 %.2f, barrier + S5 (3 items) %.2f us\n",
               pf ? "with the L2 weight prefetch" : "without prefetch          ", M, med(dur), dur.size(), med(s1), med(b0), med(s2), med(s3), med(s5));
    }
    stamps = stampsN + (size_t)(IT - 1) * 32 * 8;
    // correctness of the last launch (weight set (48 - 1) % 12 = 11) against the library's ordinary launches
    const int l = (IT - 1) % L;
    CK(hipMemcpy(x_ref, dev(hx), M * d * 4, hipMemcpyDeviceToDevice));
    if (vh_linear(attn, d, wo[l], bo, x_ref, d, xm_ref, d, M, d, d, VH_ACT_NONE, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr)) return 3;
    if (vh_linear_folded(xm_ref, d, w1f[l], c11[l], c12[l], nullptr, 0, hid_ref, dff, M, dff, d, VH_ACT_GELU_ERF, 1e-5f, nullptr)) return 3;
    if (vh_linear_ws(hid_ref, dff, w2[l], b2, xm_ref, d, x_ref, d, M, d, dff, VH_ACT_NONE, ws, vh_linear_ws_bytes(M, d, dff), nullptr)) return 3;
    if (vh_linear_qkv_folded(x_ref, 0, d, wqf[l], cq1[l], cq2[l], q_ref, d, kc_ref, vc_ref, cl, M, 1, d, h, S, 1e-5f, 0, nullptr)) return 3;
    CK(hipDeviceSynchronize());
    auto maxdiff = [&](const float* a, const float* b, size_t n) { std::vector<float> ha(n), hb(n); hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost); double m = 0; for (size_t i = 0; i < n; ++i) m = std::max(m, (double)std::fabs(ha[i] - hb[i])); return m; };
    printf("max |difference| to the ordinary launches: x_mid %.2e  hidden %.2e  x %.2e  q %.2e  K cache %.2e\n", maxdiff(xm, xm_ref, M * d),
           maxdiff(hid, hid_ref, (size_t)M * dff), maxdiff(x, x_ref, M * d), maxdiff(q, q_ref, M * d), maxdiff(kc, kc_ref, (size_t)M * h * S * 64));
    return 0;
}
