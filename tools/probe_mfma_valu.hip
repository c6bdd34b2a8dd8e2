// Does vector work hide behind a wave's own MFMAs on gfx950?  One loop body = 1 v_mfma_f32_32x32x16_f16 + N independent VALU
// instructions (v_fma_f32, or v_exp_f32 with -DTRANS), W waves per SIMD (a workgroup of 256 x W threads per CU, all CUs), timed
// with HIP events; printed as shader cycles per loop body at the clock given on the command line (default 2.0 GHz).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_valu.hip -o tools/probe_mfma_valu.bin && tools/probe_mfma_valu.bin [GHz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int N, bool TRANS, bool MFMA>
__global__ __launch_bounds__(1024) void body(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc0 = {}, acc1 = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MFMA) {
                if (rep & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (TRANS) v[k & 15] = __builtin_amdgcn_exp2f(v[k & 15]);
                else v[k & 15] = __builtin_fmaf(v[k & 15], 1.0001f, 0.5f);
                asm volatile("" : "+v"(v[k & 15]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, bool TRANS, bool MFMA>
static void run(int waves_per_simd, double ghz) {
    const int iters = 20000, threads = 256 * waves_per_simd, blocks = 256;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    body<N, TRANS, MFMA><<<blocks, threads>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    body<N, TRANS, MFMA><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9 / (iters * 4.0);
    printf("waves/SIMD %d  %s  N=%2d %-6s : %7.1f cycles per body per SIMD-slot (%.1f per wave)\n", waves_per_simd, MFMA ? "MFMA +" : "no MFMA", N,
           TRANS ? "v_exp" : "v_fma", cyc / waves_per_simd, cyc);
    hipFree(out);
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.0;
    for (int w = 1; w <= 2; ++w) {
        run<0, false, true>(w, ghz);
        run<4, false, true>(w, ghz);
        run<8, false, true>(w, ghz);
        run<16, false, true>(w, ghz);
        run<16, false, false>(w, ghz);
        run<4, true, true>(w, ghz);
        run<8, true, true>(w, ghz);
        run<8, true, false>(w, ghz);
    }
    return 0;
}
