// Host side of libvalle_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer ("ASan/UBSan build
// of the C-ABI host shim").  CPU only: the library's HOST code is compiled with -fsanitize=address,undefined and
// -fno-gpu-sanitize (device code untouched: no GPU sanitizer, no XNACK) and this driver calls everything that needs no device:
//   * every entry point with null / misaligned / out-of-range arguments — each must refuse (negative code, a reason in
//     vh_last_error, nothing launched) without reading a byte behind a pointer it was given;
//   * every workspace planner (vh_*_ws_bytes, vh_attn_rows_bwd_chunks) over a grid of shapes, degenerate ones included;
//   * the decoder life-cycle on bad descriptors (create refuses, destroy(NULL), replay before capture, capture on the null
//     stream) and on a descriptor whose pointers are never dereferenced on the host;
//   * vh_set_tuning bounds, the thread-local error string from eight threads.
// Exit code 0 and "asan_host: N checks passed" = clean; a sanitizer report aborts the process (-fno-sanitize-recover).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "valle_hip.h"

static int n_checks = 0;
#define EXPECT(cond, ...)                                         \
    do {                                                          \
        ++n_checks;                                               \
        if (!(cond)) {                                            \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);  \
            fprintf(stderr, __VA_ARGS__);                         \
            fprintf(stderr, " (last error: '%s')\n", vh_last_error()); \
            exit(1);                                              \
        }                                                         \
    } while (0)
#define REFUSED(call) do { int rc_ = (call); EXPECT(rc_ < 0 && vh_last_error()[0], "%s must be refused, rc=%d", #call, rc_); } while (0)

// Pointers that pass the alignment test but must never be dereferenced on the host: inside a PROT_NONE-like poison — here
// simply addresses in a heap block the sanitizer has been told is off limits (a freed block is poisoned by ASan).
static float* poisoned() {
    static char* block = nullptr;
    if (!block) {
        block = (char*)aligned_alloc(64, 1 << 16);
        free(block);                       // any host read or write through it from now on is a use-after-free report
    }
    return (float*)block;
}

int main() {
    float* P = poisoned();
    float* MIS = (float*)((char*)P + 4);                   // misaligned for the 16-byte rule
    const uint16_t* P16 = (const uint16_t*)P;
    uint16_t* W16 = (uint16_t*)P;
    int32_t* I32 = (int32_t*)P;
    int64_t* I64 = (int64_t*)P;
    uint8_t* U8 = (uint8_t*)P;
    void* S = nullptr;

    EXPECT(vh_version() == VH_VERSION, "version");
    EXPECT(vh_h16_format() == 0 || vh_h16_format() == 1, "h16 format");
    EXPECT(vh_set_tuning(-1, 0) < 0 && vh_set_tuning(VH_TUNE_COUNT, 1) < 0 && vh_set_tuning(VH_TUNE_FFN_FUSED, 0) == VH_OK, "tuning bounds");

    // ---- forward primitives -------------------------------------------------------------------------------------
    const float* tabs[VH_MAX_TABLES + 1] = {P, P, P, P, P, P, P, P, P};
    const int32_t voc[VH_MAX_TABLES + 1] = {4, 4, 4, 4, 4, 4, 4, 4, 4};
    REFUSED(vh_embed_sum_pe(nullptr, 0, 0, 0, tabs, voc, 1, P, 0, nullptr, P, 0, 0, 1, 1, 64, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_embed_sum_pe(I64, 1, 1, 1, tabs, voc, VH_MAX_TABLES + 1, P, 0, nullptr, P, 64, 0, 1, 1, 64, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_embed_sum_pe(I64, 1, 1, 1, tabs, voc, 1, P, 0, nullptr, P, 64, 0, 1, 1, 62, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_embed_sum_pe(I64, 1, 1, 1, tabs, voc, 1, P, 0, nullptr, MIS, 64, 0, 1, 1, 64, nullptr, nullptr, nullptr, nullptr, S));
    const int32_t voc0[1] = {0};
    REFUSED(vh_embed_sum_pe(I64, 1, 1, 1, tabs, voc0, 1, P, 0, nullptr, P, 64, 0, 1, 1, 64, nullptr, nullptr, nullptr, nullptr, S));
    vh_dropout_spec bad_p = {1, 2, 1.5f}, ok_p = {1, 2, 0.1f};
    REFUSED(vh_embed_sum_pe(I64, 1, 1, 1, tabs, voc, 1, P, 0, nullptr, P, 64, 0, 1, 1, 64, nullptr, nullptr, nullptr, &bad_p, S));
    EXPECT(vh_embed_sum_pe(I64, 1, 1, 1, tabs, voc, 1, P, 0, nullptr, P, 64, 0, 0, 5, 64, nullptr, nullptr, nullptr, &ok_p, S) == VH_OK, "B = 0 is a no-op");
    REFUSED(vh_dropout(nullptr, 4, P, 4, 1, 4, &ok_p, S));
    REFUSED(vh_dropout(P, 4, P, 4, 1, 6, &ok_p, S));
    REFUSED(vh_dropout(P, 4, P, 4, 1, 4, &bad_p, S));
    REFUSED(vh_dropout_mask(nullptr, 1, 4, &ok_p, S));
    REFUSED(vh_dropout_mask(U8, 1, 5, &ok_p, S));
    REFUSED(vh_add_pe(nullptr, P, P, 1, 1, 64, 0, S));
    REFUSED(vh_add_pe(P, P, P, 1, 1, 63, 0, S));
    REFUSED(vh_add_pe(P, MIS, P, 1, 1, 64, 0, S));
    REFUSED(vh_add_pe(P, P, P, 1, 1, 64, -1, S));
    EXPECT(vh_add_pe(P, P, P, 0, 7, 64, 0, S) == VH_OK, "empty add_pe");
    REFUSED(vh_layernorm(nullptr, P, P, nullptr, nullptr, P, 1, 64, 1e-5f, S));
    REFUSED(vh_layernorm(P, P, P, P, nullptr, P, 1, 64, 1e-5f, S));
    REFUSED(vh_layernorm(P, P, P, nullptr, nullptr, P, 1, 66, 1e-5f, S));
    REFUSED(vh_layernorm(P, P, P, nullptr, nullptr, P, 1, 8192, 1e-5f, S));
    REFUSED(vh_layernorm(P, MIS, P, nullptr, nullptr, P, 1, 64, 1e-5f, S));
    EXPECT(vh_layernorm(P, P, P, nullptr, nullptr, P, 0, 64, 1e-5f, S) == VH_OK, "0 rows");
    REFUSED(vh_linear(nullptr, 16, nullptr, nullptr, nullptr, 0, nullptr, 16, 4, 16, 16, VH_ACT_NONE, nullptr, nullptr, nullptr, nullptr, 0.f, S));
    REFUSED(vh_linear(P, 16, P, nullptr, nullptr, 0, MIS, 16, 4, 16, 16, VH_ACT_NONE, nullptr, nullptr, nullptr, nullptr, 0.f, S));
    REFUSED(vh_linear(P, 14, P, nullptr, nullptr, 0, P, 16, 4, 16, 16, VH_ACT_NONE, nullptr, nullptr, nullptr, nullptr, 0.f, S));
    REFUSED(vh_linear(P, 16, P, nullptr, nullptr, 0, P, 16, -1, 16, 16, VH_ACT_NONE, nullptr, nullptr, nullptr, nullptr, 0.f, S));
    REFUSED(vh_linear(P, 16, P, nullptr, nullptr, 0, P, 16, 100, 16, 16, VH_ACT_NONE, P, P, nullptr, nullptr, 1e-5f, S));   // fused LN: M <= 64
    REFUSED(vh_linear_ws(nullptr, 16, P, nullptr, nullptr, 0, P, 16, 4, 16, 16, VH_ACT_NONE, nullptr, 0, S));
    REFUSED(vh_linear_qkv(nullptr, 128, P, P, 128, P, P, nullptr, 1, 1, 128, 2, 4, nullptr, nullptr, nullptr, nullptr, 0.f, S));
    REFUSED(vh_linear_qkv(P, 128, P, P, 128, P, P, nullptr, 1, 8, 128, 2, 4, nullptr, nullptr, nullptr, nullptr, 0.f, S));   // T > S_max
    REFUSED(vh_linear_qkv(P, 128, P, P, 128, P, P, nullptr, 1, 1, 128, 3, 4, nullptr, nullptr, nullptr, nullptr, 0.f, S));   // d != h * 64
    REFUSED(vh_ln_fold(nullptr, P, P, nullptr, P, P, P, 16, 128, S));
    REFUSED(vh_ln_fold(P, P, P, nullptr, P, P, P, 16, 100, S));
    REFUSED(vh_linear_folded(nullptr, 128, P, P, P, nullptr, 0, P, 16, 4, 16, 128, VH_ACT_NONE, 1e-5f, S));
    REFUSED(vh_linear_folded(P, 128, P, P, P, nullptr, 0, P, 16, 100, 16, 128, VH_ACT_NONE, 1e-5f, S));
    REFUSED(vh_linear_folded(P, 192, P, P, P, nullptr, 0, P, 16, 4, 16, 192, VH_ACT_NONE, 1e-5f, S));
    REFUSED(vh_linear_qkv_folded(nullptr, 128, P, P, P, P, 128, P, P, I32, 1, 1, 128, 2, 4, 1e-5f, S));
    REFUSED(vh_linear_qkv_folded_kv16(nullptr, 128, P, P, P, P, 128, W16, W16, I32, 1, 128, 2, 4, 1e-5f, S));
    REFUSED(vh_ffn_decode(nullptr, 128, P, P, P, P, nullptr, P, 128, 4, 128, 512, 1e-5f, P, 1 << 20, S));
    REFUSED(vh_ffn_decode(P, 128, P, P, P, P, nullptr, P, 128, 4, 192, 512, 1e-5f, P, 1 << 20, S));
    REFUSED(vh_ffn_decode(P, 128, P, P, P, P, nullptr, P, 128, 4, 128, 512, 1e-5f, P, 16, S));      // workspace too small
    REFUSED(vh_ffn_decode(P, 128, P, P, P, P, nullptr, P, 128, 65, 128, 512, 1e-5f, P, 1 << 24, S));
    // ---- attention -------------------------------------------------------------------------------------------------
    REFUSED(vh_attn_rows(nullptr, 128, P, P, P, 128, 1, 2, 4, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_attn_rows(P, 128, P, P, P, 128, 1, 2, 9, 9, 8, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, S));      // Tk > S_max
    REFUSED(vh_attn_rows(P, 128, P, P, P, 128, 1, 2, 4, 4, 8, VH_MASK_EXPLICIT, 0, nullptr, nullptr, nullptr, nullptr, S));  // no mask
    REFUSED(vh_attn_rows(P, 64, P, P, P, 128, 1, 2, 4, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, S));       // ldq < h * 64
    REFUSED(vh_attn_rows(P, 128, P, P, P, 128, 1, 2, 4, 4, 8, 7, 0, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_attn_rows_lse(P, 128, P, P, P, 128, 1, 2, 4, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_attn_rows_bmask(P, 128, P, P, P, 128, 1, 2, 4, 4, 8, nullptr, 16, nullptr, S));                                  // no mask
    REFUSED(vh_attn_rows_bmask(P, 128, P, P, P, 128, 2, 2, 4, 4, 8, (const uint8_t*)P, 15, nullptr, S));                        // stride < Tq * Tk
    REFUSED(vh_attn_rows_bmask(nullptr, 128, P, P, P, 128, 2, 2, 4, 4, 8, (const uint8_t*)P, 16, nullptr, S));
    REFUSED(vh_attn_rows_bwd(nullptr, 128, P, P, P, 128, P, 128, P, P, P, P, P, 128, 1, 2, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_attn_rows_bwd(P, 128, P, P, P, 128, P, 128, P, P, P, P, P, 126, 1, 2, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_attn_rows_bwd_ws(P, 128, P, P, P, 128, P, 128, P, P, P, P, 128, 1, 2, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1 << 20, S));
    REFUSED(vh_attn_rows_bwd_ws(P, 128, P, P, P, 128, P, 128, P, P, P, P, 128, 4, 8, 1000, 1000, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, P, 16, S));   // ws too small
    REFUSED(vh_attn_decode(nullptr, 128, P, P, P, 128, I32, 1, 1, 2, 8, 1, nullptr, S));
    REFUSED(vh_attn_decode(P, 128, P, P, P, 128, I32, 2, 1, 2, 8, 1, nullptr, S));                 // len_bias
    REFUSED(vh_attn_decode(P, 128, P, P, P, 128, I32, 1, 1, 2, 8, 2, nullptr, S));                 // split without workspace
    REFUSED(vh_attn_decode(P, 128, P, P, P, 128, I32, 1, 1, 2, 8, 65, P, S));
    REFUSED(vh_attn_decode(P, 128, MIS, P, P, 128, I32, 1, 1, 2, 8, 1, nullptr, S));
    REFUSED(vh_attn_decode_shared(nullptr, 128, P, P, 64, 64, P, P, P, 128, I32, 1, 4, 2, 32, 1, P, 1 << 20, S));
    REFUSED(vh_attn_decode_shared(P, 128, P, P, 65, 64, P, P, P, 128, I32, 1, 4, 2, 32, 1, P, 1 << 20, S));        // prefix_len > prefix_S
    REFUSED(vh_attn_decode_shared(P, 128, P, P, 64, 64, P, P, P, 128, I32, 1, 65, 2, 32, 1, P, 1 << 24, S));       // B > 64
    REFUSED(vh_attn_decode_shared(P, 128, P, P, 64, 64, P, P, P, 128, I32, 1, 4, 2, 32, 1, P, 64, S));             // workspace too small
    REFUSED(vh_attn_decode_shared(P, 128, P, P, 64, 64, P, P, P, 128, I32, 1, 4, 2, 32, 1, nullptr, 1 << 20, S));
    REFUSED(vh_attn_decode_shared(P, 128, P, MIS, 64, 64, P, P, P, 128, I32, 1, 4, 2, 32, 1, P, 1 << 20, S));
    REFUSED(vh_kv_to_bf16(nullptr, W16, 1, 1, 4, 4, S));
    REFUSED(vh_kv_to_bf16(P, W16, 1, 8, 4, 4, S));                                                 // rows > S
    REFUSED(vh_attn_decode_kv16(nullptr, 128, P16, P16, P, 128, I32, 1, 1, 2, 8, S));
    // ---- sampling --------------------------------------------------------------------------------------------------
    REFUSED(vh_greedy_step(nullptr, 8, 8, 7, I64, 4, I32, nullptr, P, P, I32, I32, P, 1, 64, S));
    REFUSED(vh_greedy_step(P, 4, 8, 7, I64, 4, I32, nullptr, P, P, I32, I32, P, 1, 64, S));       // ldl < V
    REFUSED(vh_greedy_step(P, 8, 8, 7, I64, 4, I32, nullptr, P, MIS, I32, I32, P, 1, 64, S));
    REFUSED(vh_head_greedy(nullptr, 128, P, P, 1028, 1025, 1024, I64, 80, I32, nullptr, P, P, I32, I32, P, 4, 128, P, 1 << 20, S));
    REFUSED(vh_head_greedy(P, 128, P, P, 1024, 1025, 1024, I64, 80, I32, nullptr, P, P, I32, I32, P, 4, 128, P, 1 << 20, S));   // ldl < V
    REFUSED(vh_head_greedy(P, 192, P, P, 1028, 1025, 1024, I64, 80, I32, nullptr, P, P, I32, I32, P, 4, 192, P, 1 << 20, S));   // d = 192
    REFUSED(vh_head_greedy(P, 128, P, P, 1028, 1025, 1024, I64, 80, I32, nullptr, P, P, I32, I32, P, 65, 128, P, 1 << 20, S));  // B > 64
    REFUSED(vh_head_greedy(P, 128, P, P, 1028, 1025, 1024, I64, 80, I32, nullptr, P, P, I32, I32, P, 4, 128, P, 256, S));       // workspace too small
    REFUSED(vh_head_greedy(P, 128, P, P, 1028, 1025, 1024, I64, 80, I32, nullptr, P, P, I32, I32, P, 4, 128, nullptr, 1 << 20, S));
    REFUSED(vh_head_greedy(P, 128, P, P, 1028, 1025, 1024, I64, 80, I32, nullptr, P, P, I32, I32, P, 4, 128, MIS, 1 << 20, S)); // workspace alignment
    for (int B : {-1, 0, 1, 32, 64})
        for (int V : {0, 1, 16, 17, 1025, 2048})
            EXPECT(vh_head_greedy_ws_bytes(B, V) == (B > 0 && V > 0 ? 256 + (size_t)B * ((V + 15) / 16) * 8 : 0), "head workspace B=%d V=%d", B, V);
    REFUSED(vh_sample_step(nullptr, 8, 8, 7, 5, 1.f, 1.f, 1, I64, 4, I32, nullptr, P, P, P, I32, I32, P, 1, 64, S));
    REFUSED(vh_sample_step(P, 8, 8, 7, 5, 1.f, 0.f, 1, I64, 4, I32, nullptr, P, P, P, I32, I32, P, 1, 64, S));      // temperature 0
    REFUSED(vh_sample_step(P, 4096, 4096, 7, 5, 1.f, 1.f, 1, I64, 4, I32, nullptr, P, P, P, I32, I32, P, 1, 64, S)); // V > 2048
    REFUSED(vh_categorical_rows(nullptr, 8, 8, 1, 1.f, 0, 1, 0, I64, 1, nullptr, S));
    REFUSED(vh_categorical_rows(P, 8, 8, 1, 0.f, 0, 1, 0, I64, 1, nullptr, S));
    // ---- training row kernels, optimizer, training products ---------------------------------------------------------
    REFUSED(vh_layernorm_bwd(nullptr, P, P, nullptr, P, P, P, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 64, 1e-5f, S));
    REFUSED(vh_layernorm_bwd(P, P, P, nullptr, P, P, P, P, nullptr, nullptr, nullptr, nullptr, P, nullptr, 1, 64, 1e-5f, S));   // dx_drop without drop
    REFUSED(vh_gelu(nullptr, nullptr, P, 4, S));
    REFUSED(vh_softmax_rows(nullptr, 8, 1, 1, 1, 8, 1.f, VH_MASK_FULL, 0, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_softmax_bwd(nullptr, P, 8, 1, 8, 1.f, S));
    REFUSED(vh_cross_entropy(nullptr, 8, 8, I64, P, nullptr, 0, 1, nullptr, S));
    REFUSED(vh_cross_entropy(P, 4, 8, I64, P, nullptr, 0, 1, nullptr, S));
    REFUSED(vh_embed_bwd(nullptr, 1, 1, P, 64, 0, P, 4, 1, 1, 64, nullptr, nullptr, S));
    REFUSED(vh_colsum(nullptr, 8, P, 1, 8, S));
    vh_adaproj_item* items = (vh_adaproj_item*)P;
    REFUSED(vh_adaproj_fwd(nullptr, 1, P, P, 128, 64, S));
    REFUSED(vh_adaproj_fwd(items, 1, P, P, 128, 63, S));
    REFUSED(vh_adaproj_fwd(items, 1, P, P, 8192, 4096, S));
    REFUSED(vh_adaproj_bwd(items, 1, P, nullptr, P, 128, 64, S));
    EXPECT(vh_adamw_ws_bytes() > 0, "adamw workspace");
    REFUSED(vh_adamw_flat(nullptr, P, P, P, 64, 1e-4f, .9f, .98f, 1e-8f, .1f, 1, 1.f, 1.f, 1, P, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_adamw_flat(P, P, P, P, 62, 1e-4f, .9f, .98f, 1e-8f, .1f, 1, 1.f, 1.f, 1, P, nullptr, nullptr, nullptr, nullptr, S));
    REFUSED(vh_adamw_flat(P, P, P, P, 64, 1e-4f, .9f, .98f, 1e-8f, .1f, 0, 1.f, 1.f, 1, P, nullptr, nullptr, nullptr, nullptr, S));   // step 0
    REFUSED(vh_adamw_flat(P, P, P, P, 64, 1e-4f, .9f, .98f, 1e-8f, .1f, 1, 1.f, 1.f, 1, P, nullptr, I32, nullptr, nullptr, S));       // block_slot without slot_step
    REFUSED(vh_linear_ex(nullptr, 32, P, nullptr, nullptr, 0, P, 128, nullptr, 0, nullptr, 4, 128, 32, VH_ACT_NONE, nullptr, nullptr, 0, S));
    REFUSED(vh_linear_ex(P, 32, P, nullptr, nullptr, 0, P, 128, nullptr, 0, nullptr, 4, 128, 32, VH_ACT_GELU_ERF_D, nullptr, nullptr, 0, S));   // needs pre_out
    REFUSED(vh_linear_ex(P, 32, P, P, nullptr, 0, P, 128, nullptr, 0, nullptr, 4, 128, 32, VH_ACT_MUL, nullptr, nullptr, 0, S));               // needs residual
    REFUSED(vh_linear_ex(P, 32, P, nullptr, nullptr, 0, P, 100, nullptr, 0, P, 4, 100, 32, VH_ACT_NONE, nullptr, nullptr, 0, S));             // colsum: N % 128
    REFUSED(vh_transpose(nullptr, 8, 4, 8, P, 4, S));
    REFUSED(vh_transpose(P, 8, 4, 8, P, 2, S));
    REFUSED(vh_transpose_many(nullptr, 1, 1, S));
    REFUSED(vh_gemm_tn(nullptr, 128, P, 128, P, 128, 64, 128, 128, P, 1 << 20, S));
    REFUSED(vh_gemm_tn(P, 128, P, 128, P, 128, 4096, 128, 128, nullptr, 0, S));
    REFUSED(vh_gemm_tn(P, 126, P, 128, P, 128, 64, 128, 128, P, 1 << 20, S));
    REFUSED(vh_gemm_batched(nullptr, 8, 0, 0, 0, P, 8, 0, 0, 0, P, 8, 0, 0, 8, 8, 8, 1, 1, 1, S));
    REFUSED(vh_gemm_batched(P, 6, 0, 0, 0, P, 8, 0, 0, 0, P, 8, 0, 0, 8, 8, 8, 1, 1, 1, S));
    // ---- perf mode (bf16) ------------------------------------------------------------------------------------------------
    REFUSED(vh_to_bf16(nullptr, 8, W16, 8, 1, 8, S));
    REFUSED(vh_to_bf16(P, 8, W16, 8, 1, 12, S));
    REFUSED(vh_to_bf16(P, 4, W16, 8, 1, 8, S));                                                       // lds < cols
    EXPECT(vh_to_bf16(P, 8, W16, 8, 0, 8, S) == VH_OK, "0 rows");
    REFUSED(vh_layernorm_bf16(nullptr, P, P, nullptr, nullptr, W16, 1, 64, 1e-5f, S));
    REFUSED(vh_layernorm_bf16(P, P, P, nullptr, nullptr, W16, 1, 68, 1e-5f, S));
    REFUSED(vh_linear_bf16(nullptr, 64, P16, nullptr, nullptr, 0, P, 128, 0, 4, 128, 64, VH_ACT_NONE, S));
    REFUSED(vh_linear_bf16(P16, 64, P16, nullptr, nullptr, 0, P, 100, 0, 4, 100, 64, VH_ACT_NONE, S));      // N % 128
    REFUSED(vh_linear_bf16(P16, 96, P16, nullptr, nullptr, 0, P, 128, 0, 4, 128, 96, VH_ACT_NONE, S));      // K % 64
    REFUSED(vh_linear_bf16(P16, 64, P16, nullptr, P, 128, P, 128, 1, 4, 128, 64, VH_ACT_NONE, S));          // bf16 out + residual
    REFUSED(vh_linear_bf16(P16, 64, P16, nullptr, nullptr, 0, P, 128, 0, 4, 128, 64, VH_ACT_MUL, S));
    REFUSED(vh_linear_qkv_bf16(P16, 128, P16, W16, 128, nullptr, W16, nullptr, 1, 1, 128, 2, 4, S));
    REFUSED(vh_linear_qkv_bf16(P16, 64, P16, W16, 64, W16, W16, nullptr, 1, 1, 64, 1, 4, S));                // d % 128
    REFUSED(vh_attn_rows_bf16(nullptr, 128, P16, P16, W16, 128, 1, 2, 4, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, S));
    REFUSED(vh_attn_rows_bf16(P16, 128, P16, P16, W16, 128, 1, 2, 4, 4, 8, VH_MASK_EXPLICIT, 0, nullptr, nullptr, S));
    REFUSED(vh_attn_rows_bf16(P16, 128, P16, P16, W16, 128, 1, 2, 5, 4, 8, VH_MASK_FULL, 0, nullptr, nullptr, S));   // Tq > Tk

    // ---- workspace planners: pure host arithmetic over a grid of shapes (overflow / division by zero / negative sizes) ----
    const int Ms[] = {0, 1, 7, 32, 64, 65, 1000, 5008, 16320, 65536}, Ns[] = {1, 16, 512, 1025, 1536, 4096}, Ks[] = {4, 64, 512, 1024, 2048, 4096};
    for (int M : Ms)
        for (int N : Ns)
            for (int K : Ks) {
                const size_t a = vh_linear_ws_bytes(M, N, K), b = vh_linear_ex_ws_bytes(M, N, K), c = vh_gemm_tn_ws_bytes(M, N, K);
                EXPECT(a < (1ull << 34) && b <= (16ull << 20) && c < (1ull << 34), "workspace plan M=%d N=%d K=%d: %zu %zu %zu", M, N, K, a, b, c);
            }
    for (int M : {0, 1, 8, 32, 64, 65})
        for (int d : {128, 192, 256, 512, 1024})
            for (int dff : {16, 512, 2048, 4096, 16400}) (void)vh_ffn_decode_ws_bytes(M, d, dff), ++n_checks;
    for (int B : {-1, 0, 1, 8, 16, 64})
        for (int h : {0, 1, 8, 16})
            for (int T : {-5, 0, 1, 31, 256, 257, 640, 1021, 2875, 5000}) {
                const int nc_full = vh_attn_rows_bwd_chunks(B, h, T, VH_MASK_FULL), nc_pre = vh_attn_rows_bwd_chunks(B, h, T, VH_MASK_PREFIX);
                const size_t ws = vh_attn_rows_bwd_ws_bytes(B, h, T);
                const bool live = B > 0 && h > 0 && T > 0;
                EXPECT(live ? (nc_full >= (T + 255) / 256 && nc_pre >= (T + 255) / 256 && ws > 0) : (nc_full == 0 && nc_pre == 0 && ws == 0),
                       "attention backward plan B=%d h=%d T=%d: %d %d %zu", B, h, T, nc_full, nc_pre, ws);
            }
    for (int k = 1; k <= 40; ++k) {                         // the forced chunk count never leaves an empty last chunk
        vh_set_tuning(VH_TUNE_ATTN_BWD_CHUNKS, k);
        for (int T : {1, 33, 300, 1021, 2875}) {
            const int nc = vh_attn_rows_bwd_chunks(4, 8, T, VH_MASK_FULL), keys = ((T + nc - 1) / nc + 31) / 32 * 32;
            EXPECT(nc >= 1 && (int64_t)keys * (nc - 1) < T, "forced chunks k=%d T=%d -> nc=%d keys=%d", k, T, nc, keys);
        }
    }
    vh_set_tuning(VH_TUNE_ATTN_BWD_CHUNKS, 0);
    for (int B : {0, 1, 4, 64})
        for (int h : {1, 8, 16})
            for (int ns : {0, 1, 2, 16, 64}) {
                const size_t w = vh_attn_decode_ws_bytes(B, h, ns);
                EXPECT(ns <= 1 ? w == 0 : w >= (size_t)B * h * 4, "decode workspace B=%d h=%d n_split=%d: %zu", B, h, ns, w);
            }

    for (int B : {0, 1, 4, 32, 64})
        for (int h : {1, 8, 16})
            for (int pl : {0, 1, 31, 626, 1024, 2907, 5000})
                for (int ns : {0, 1, 8}) {
                    const size_t w = vh_attn_decode_shared_ws_bytes(B, h, pl, ns);
                    const bool live = B > 0 && pl > 0 && ns >= 1;
                    // one PART_LD = 72-float record per 32-key block of the prefix and per suffix split
                    EXPECT(live ? w == (size_t)B * h * ((pl + 31) / 32 + ns) * 288 : w == 0,
                           "shared-prompt workspace B=%d h=%d prefix=%d n_split=%d: %zu", B, h, pl, ns, w);
                }

    // ---- composites: descriptors ------------------------------------------------------------------------------------
    REFUSED(vh_transformer_forward(nullptr, S));
    vh_forward_desc fd;
    memset(&fd, 0, sizeof fd);
    REFUSED(vh_transformer_forward(&fd, S));
    vh_layer layers[2];
    memset(layers, 0, sizeof layers);
    fd.layers = layers; fd.x = fd.xn = fd.q = fd.attn = fd.hidden = P;
    fd.B = 1; fd.T = 8; fd.n_layers = 2; fd.S_max = 4; fd.d_model = 128; fd.n_heads = 2; fd.dff = 256;
    REFUSED(vh_transformer_forward(&fd, S));               // S_max < T
    fd.S_max = 8; fd.n_heads = 3;
    REFUSED(vh_transformer_forward(&fd, S));               // d_model != n_heads x 64
    REFUSED(vh_transformer_forward_bf16(nullptr, S));
    vh_forward16_desc f16;
    memset(&f16, 0, sizeof f16);
    REFUSED(vh_transformer_forward_bf16(&f16, S));
    vh_layer16 l16[2];
    memset(l16, 0, sizeof l16);
    f16.layers = layers; f16.layers16 = l16; f16.x = P; f16.xn16 = f16.q16 = f16.attn16 = f16.hidden16 = W16;
    f16.B = 1; f16.T = 8; f16.n_layers = 2; f16.S_max = 8; f16.d_model = 64; f16.n_heads = 1; f16.dff = 256;
    REFUSED(vh_transformer_forward_bf16(&f16, S));         // d_model % 128
    f16.d_model = 128; f16.n_heads = 2; f16.mode = VH_MASK_EXPLICIT;
    REFUSED(vh_transformer_forward_bf16(&f16, S));
    f16.mode = VH_MASK_FULL;
    REFUSED(vh_transformer_forward_bf16(&f16, S));         // layer 0 holds null bf16 pointers

    EXPECT(vh_ar_decoder_create(nullptr) == nullptr, "null desc");
    vh_ar_decoder_desc dd;
    memset(&dd, 0, sizeof dd);
    EXPECT(vh_ar_decoder_create(&dd) == nullptr && strstr(vh_last_error(), "B=0"), "empty desc");
    dd.B = 4; dd.d_model = 128; dd.n_heads = 2; dd.dff = 256; dd.n_layers = 2; dd.S_max = 64; dd.V = 1025; dd.eos = 1024;
    dd.n_split = 1; dd.ln_eps = 1e-5f; dd.layers = layers; dd.proj_w = dd.audio_emb = dd.audio_pe = P;
    dd.x = dd.q = dd.attn = dd.hidden = dd.logits = P; dd.cache_len = dd.audio_pos = dd.eos_count = I32; dd.codes = I64;
    dd.codes_stride = 80; dd.top_k = 1; dd.temperature = 1.f;
    vh_ar_decoder_desc bad = dd;
    bad.B = 65;             EXPECT(vh_ar_decoder_create(&bad) == nullptr, "B = 65");
    bad = dd; bad.n_heads = 3;   EXPECT(vh_ar_decoder_create(&bad) == nullptr, "d != h x 64");
    bad = dd; bad.dff = 250;     EXPECT(vh_ar_decoder_create(&bad) == nullptr, "dff %% 16");
    bad = dd; bad.n_split = 0;   EXPECT(vh_ar_decoder_create(&bad) == nullptr, "n_split 0");
    bad = dd; bad.n_split = 4;   EXPECT(vh_ar_decoder_create(&bad) == nullptr, "split without workspace");
    bad = dd; bad.codes = nullptr; EXPECT(vh_ar_decoder_create(&bad) == nullptr, "null codes");
    bad = dd; bad.layers = nullptr; EXPECT(vh_ar_decoder_create(&bad) == nullptr, "null layers");
    bad = dd; bad.ffn_ws = P; bad.ffn_ws_bytes = 16; EXPECT(vh_ar_decoder_create(&bad) == nullptr, "ffn workspace too small");
    bad = dd; bad.ffn_ws = P; bad.ffn_ws_bytes = 1 << 26; EXPECT(vh_ar_decoder_create(&bad) == nullptr && strstr(vh_last_error(), "folded"), "ffn workspace without folded weights");
    bad = dd; bad.kv_bf16 = 1;   EXPECT(vh_ar_decoder_create(&bad) == nullptr && strstr(vh_last_error(), "folded"), "bf16 cache without folded weights");
    bad = dd; bad.top_k = 50; bad.temperature = 0.f; EXPECT(vh_ar_decoder_create(&bad) == nullptr, "sampling at temperature 0");
    bad = dd; bad.head_ws = P; bad.head_ws_bytes = 64; EXPECT(vh_ar_decoder_create(&bad) == nullptr && strstr(vh_last_error(), "head_ws"), "head workspace too small");
    bad = dd; bad.head_ws = P; bad.head_ws_bytes = 1 << 20; bad.top_k = 50; EXPECT(vh_ar_decoder_create(&bad) == nullptr && strstr(vh_last_error(), "head_ws"), "head workspace with sampling");
    bad = dd; bad.prefix_len = -3; EXPECT(vh_ar_decoder_create(&bad) == nullptr, "negative prefix");
    bad = dd; bad.prefix_len = 100; bad.prefix_S = 128; bad.attn_partial = P; bad.attn_partial_bytes = 1 << 24;
    EXPECT(vh_ar_decoder_create(&bad) == nullptr && strstr(vh_last_error(), "kprefix"), "shared prompt without prefix caches");
    {
        vh_layer pl[2];
        memcpy(pl, layers, sizeof pl);
        pl[0].kprefix = pl[0].vprefix = pl[1].kprefix = pl[1].vprefix = P;
        bad.layers = pl;
        bad.attn_partial_bytes = 16;
        EXPECT(vh_ar_decoder_create(&bad) == nullptr && strstr(vh_last_error(), "attn_partial"), "shared prompt with a small workspace");
        bad.attn_partial_bytes = 1 << 24; bad.prefix_S = 64;
        EXPECT(vh_ar_decoder_create(&bad) == nullptr, "prefix_len > prefix_S");
        bad.prefix_S = 128; bad.kv_bf16 = 1;
        EXPECT(vh_ar_decoder_create(&bad) == nullptr, "shared prompt with a bf16 cache");
        bad.kv_bf16 = 0;
        vh_ar_decoder* ok = vh_ar_decoder_create(&bad);
        EXPECT(ok != nullptr, "a well-formed shared-prompt descriptor");
        vh_ar_decoder_destroy(ok);
    }
    std::vector<vh_ar_decoder*> decs;
    for (int i = 0; i < 64; ++i) {                           // the descriptor and its layer array are COPIED: mutate / free the originals
        vh_layer* tmp = (vh_layer*)malloc(2 * sizeof(vh_layer));
        memset(tmp, 0, 2 * sizeof(vh_layer));
        vh_ar_decoder_desc d2 = dd;
        d2.layers = tmp;
        vh_ar_decoder* dec = vh_ar_decoder_create(&d2);
        free(tmp);
        EXPECT(dec != nullptr, "create");
        EXPECT(vh_ar_decoder_replay(dec, 1, S) == VH_ESTATE, "replay before capture");
        EXPECT(vh_ar_decoder_capture(dec, nullptr) == VH_EINVAL, "capture on the null stream");
        float ms = 0;
        EXPECT(vh_ar_decoder_profile_attn(dec, 0, S, &ms, nullptr, nullptr) == VH_EINVAL, "profile of 0 steps");
        EXPECT(vh_ar_decoder_profile_attn(dec, 1, S, nullptr, nullptr, nullptr) == VH_EINVAL, "profile without an output");
        decs.push_back(dec);
    }
    for (vh_ar_decoder* d : decs) vh_ar_decoder_destroy(d);
    vh_ar_decoder_destroy(nullptr);
    EXPECT(vh_ar_decoder_step(nullptr, S) == VH_EINVAL && vh_ar_decoder_capture(nullptr, (void*)16) == VH_EINVAL &&
               vh_ar_decoder_replay(nullptr, 1, S) == VH_ESTATE, "null decoder");

    // ---- the error string is thread-local ----------------------------------------------------------------------------------
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
        th.emplace_back([t, P] {
            for (int i = 0; i < 200; ++i) {
                if (t & 1) {
                    (void)vh_layernorm(nullptr, P, P, nullptr, nullptr, P, 1, 64, 1e-5f, nullptr);
                    if (!strstr(vh_last_error(), "vh_layernorm")) abort();
                } else {
                    (void)vh_colsum(nullptr, 8, P, 1, 8, nullptr);
                    if (!strstr(vh_last_error(), "vh_colsum")) abort();
                }
                (void)vh_attn_rows_bwd_chunks(1 + (i & 7), 8, 100 + 37 * i, i & 1);      // the planner's cache under contention
            }
        });
    for (auto& x : th) x.join();
    ++n_checks;
    printf("asan_host: %d checks passed\n", n_checks);
    return 0;
}
