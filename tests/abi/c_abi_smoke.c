/* The drop-in boundary used from plain C (no Python, no torch): allocate device buffers with the HIP
 * runtime, call libvalle_hip.so through include/valle_hip.h, compare with a CPU loop.
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/abi/c_abi_smoke.c \
 *       -Lvalle2_amd/csrc -lvalle_hip -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_smoke
 * Exit code 0 = every check passed.  `c_abi_smoke --no-gpu` only touches the entry points that need
 * no device (version, error string, argument rejection). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "valle_hip.h"

#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                 \
            fprintf(stderr, "\n");                        \
            return 1;                                     \
        }                                                 \
    } while (0)

static float frand(unsigned* s) {
    *s = *s * 1664525u + 1013904223u;
    return ((*s >> 8) & 0xFFFF) / 65536.0f - 0.5f;
}

int main(int argc, char** argv) {
    const int no_gpu = argc > 1 && strcmp(argv[1], "--no-gpu") == 0;
    CHECK(vh_version() == VH_VERSION, "vh_version() = %d, header says %d", vh_version(), VH_VERSION);
    /* a rejected call launches nothing and explains itself */
    int rc = vh_linear(NULL, 16, NULL, NULL, NULL, 0, NULL, 16, 4, 16, 16, VH_ACT_NONE, NULL, NULL, NULL, NULL, 0.f, NULL);
    CHECK(rc == VH_EINVAL, "null pointers must be rejected with VH_EINVAL, got %d", rc);
    CHECK(strstr(vh_last_error(), "null pointer") != NULL, "error string: '%s'", vh_last_error());
    CHECK(vh_linear_ws_bytes(32, 512, 2048) > 0 && vh_linear_ws_bytes(32, 512, 512) == 0, "workspace sizes");
    if (no_gpu) {
        printf("c_abi_smoke: host-only checks passed\n");
        return 0;
    }

    /* out = gelu?(LN(x) W^T + b) + residual at decode shape, three ways: plain LayerNorm kernel + linear,
     * LayerNorm in the operand load, LayerNorm folded into the weights */
    enum { M = 32, N = 512, K = 512 };
    unsigned seed = 7;
    float *hx = malloc(sizeof(float) * M * K), *hw = malloc(sizeof(float) * N * K), *hb = malloc(sizeof(float) * N);
    float *hg = malloc(sizeof(float) * K), *hbeta = malloc(sizeof(float) * K), *hres = malloc(sizeof(float) * M * N);
    float *ref = malloc(sizeof(float) * M * N), *got = malloc(sizeof(float) * M * N);
    for (int i = 0; i < M * K; ++i) hx[i] = 2.f * frand(&seed) + 0.3f;
    for (int i = 0; i < N * K; ++i) hw[i] = 0.1f * frand(&seed);
    for (int i = 0; i < N; ++i) hb[i] = frand(&seed);
    for (int i = 0; i < K; ++i) { hg[i] = 1.f + 0.2f * frand(&seed); hbeta[i] = 0.2f * frand(&seed); }
    for (int i = 0; i < M * N; ++i) hres[i] = frand(&seed);
    for (int m = 0; m < M; ++m) {                       /* CPU reference in double */
        double mu = 0, var = 0;
        for (int k = 0; k < K; ++k) mu += hx[m * K + k];
        mu /= K;
        for (int k = 0; k < K; ++k) var += (hx[m * K + k] - mu) * (hx[m * K + k] - mu);
        const double rstd = 1.0 / sqrt(var / K + 1e-5);
        for (int n = 0; n < N; ++n) {
            double acc = hb[n];
            for (int k = 0; k < K; ++k) acc += ((hx[m * K + k] - mu) * rstd * hg[k] + hbeta[k]) * hw[n * K + k];
            ref[m * N + n] = (float)(acc + hres[m * N + n]);
        }
    }
    float *x, *w, *b, *g, *beta, *res, *out, *xn, *wf, *c1, *c2;
#define DEV(p, n, src)                                                                       \
    CHECK(hipMalloc((void**)&p, sizeof(float) * (n)) == hipSuccess, "hipMalloc");            \
    if (src) CHECK(hipMemcpy(p, src, sizeof(float) * (n), hipMemcpyHostToDevice) == hipSuccess, "hipMemcpy")
    DEV(x, M * K, hx); DEV(w, N * K, hw); DEV(b, N, hb); DEV(g, K, hg); DEV(beta, K, hbeta);
    DEV(res, M * N, hres); DEV(out, M * N, (float*)NULL); DEV(xn, M * K, (float*)NULL);
    DEV(wf, N * K, (float*)NULL); DEV(c1, N, (float*)NULL); DEV(c2, N, (float*)NULL);
    hipStream_t s;
    CHECK(hipStreamCreate(&s) == hipSuccess, "hipStreamCreate");
    const char* names[3] = {"vh_layernorm + vh_linear", "vh_linear (LayerNorm in the operand load)", "vh_ln_fold + vh_linear_folded"};
    for (int variant = 0; variant < 3; ++variant) {
        rc = 0;
        if (variant == 0) {
            rc = vh_layernorm(x, g, beta, NULL, NULL, xn, M, K, 1e-5f, s);
            if (!rc) rc = vh_linear(xn, K, w, b, res, N, out, N, M, N, K, VH_ACT_NONE, NULL, NULL, NULL, NULL, 0.f, s);
        } else if (variant == 1) {
            rc = vh_linear(x, K, w, b, res, N, out, N, M, N, K, VH_ACT_NONE, g, beta, NULL, NULL, 1e-5f, s);
        } else {
            rc = vh_ln_fold(w, g, beta, b, wf, c1, c2, N, K, s);
            if (!rc) rc = vh_linear_folded(x, K, wf, c1, c2, res, N, out, N, M, N, K, VH_ACT_NONE, 1e-5f, s);
        }
        CHECK(rc == VH_OK, "%s: rc=%d (%s)", names[variant], rc, vh_last_error());
        CHECK(hipStreamSynchronize(s) == hipSuccess, "sync");
        CHECK(hipMemcpy(got, out, sizeof(float) * M * N, hipMemcpyDeviceToHost) == hipSuccess, "copy back");
        double worst = 0;
        for (int i = 0; i < M * N; ++i) worst = fmax(worst, fabs((double)got[i] - ref[i]));
        printf("c_abi_smoke: %-45s max |err| = %.2e\n", names[variant], worst);
        CHECK(worst < 5e-5, "%s differs from the CPU reference", names[variant]);
    }
    printf("c_abi_smoke: all checks passed\n");
    return 0;
}
