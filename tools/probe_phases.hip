// Do phase-structured waves overlap on one SIMD of gfx950?  Every wave alternates a MATRIX phase (NM back-to-back MFMAs, two
// accumulator chains) and a VECTOR phase (NV dependent-on-nothing v_fma_f32), as the perf-mode attention does per key tile
// (8 + 12 MFMAs, ~150 vector instructions).  W waves per SIMD (a workgroup of 256 x W threads per CU, all CUs), no barriers, no
// memory.  If the waves overlap, a loop body costs max(NM x 32, NM x 8 + NV x 4) cycles per wave-slot; if they do not, the sum.
// -DPRIO: s_setprio 3 around the matrix phase.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_phases.hip -o tools/probe_phases.bin && tools/probe_phases.bin [GHz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NM, int NV, bool PRIO>
__global__ __launch_bounds__(1024) void body(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc0 = {}, acc1 = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    // de-phase the waves of a SIMD a little (as different workgroups would be)
    for (int k = 0; k < (int)(threadIdx.x >> 8) * 37; ++k) { v[0] = __builtin_fmaf(v[0], 1.0001f, 0.5f); asm volatile("" : "+v"(v[0])); }
    for (int it = 0; it < iters; ++it) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            v[k & 15] = __builtin_fmaf(v[k & 15], 1.0001f, 0.5f);
            asm volatile("" : "+v"(v[k & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, bool PRIO>
static void run(int waves_per_simd, double ghz) {
    const int iters = 4000, threads = 256 * waves_per_simd, blocks = 256;
    float* out;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    body<NM, NV, PRIO><<<blocks, threads>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    body<NM, NV, PRIO><<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9 / iters;
    printf("waves/SIMD %d  NM=%2d NV=%3d %s: %8.1f cycles per body per wave-slot | pipe %d, issue %d, sum %d\n", waves_per_simd, NM, NV,
           PRIO ? "prio " : "     ", cyc / waves_per_simd, NM * 32, NM * 8 + NV * 4, NM * 32 + NV * 4);
    (void)hipFree(out);
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.0;
    for (int w = 1; w <= 4; ++w) {
        run<20, 150, false>(w, ghz);
        run<20, 150, true>(w, ghz);
        run<20, 0, false>(w, ghz);
        run<0, 150, false>(w, ghz);
        run<8, 75, false>(w, ghz);
    }
    return 0;
}
