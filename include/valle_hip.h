/*
 * valle_hip.h — C ABI of libvalle_hip.so: the MI355X (gfx950) kernels behind the AR + NAR
 * codec-token transformer path of KubiakJakub01/Valle2.
 *
 * The reference has no native layer and no FFI: its hot path is PyTorch ops
 * issued from valle/models/ (*.py).  Each entry point below therefore cites the reference
 * *call site* whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions (every entry point):
 *   - returns 0 when the work was enqueued, a negative VH_E* code when an argument is rejected
 *     (nothing is launched then); vh_last_error() gives the thread-local reason string;
 *   - no allocation, no synchronisation, no host read of device memory: every call may be
 *     captured in a hipGraph; the caller owns all buffers including workspaces;
 *   - all tensors fp32, contiguous in the stated layout, base pointers 16-byte aligned,
 *     leading dimensions multiples of 4 elements; token ids int64; lengths int32;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - one device per process; calls are thread-safe for distinct streams.
 *
 * Arithmetic is fp32 end to end (fp32 MFMA v_mfma_f32_32x32x2_f32 / 16x16x4_f32, exact f32
 * FMA chains) because greedy-token parity with the reference needs it.
 */
#ifndef VALLE_HIP_H
#define VALLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VH_VERSION 127            /* 0.2.3: perf-mode q is PRE-SCALED by 1/sqrt(64) log2(e) between vh_linear_qkv_bf16 and vh_attn_rows_bf16; 0.2.2: head + greedy step in one launch (vh_head_greedy, opt-in: vh_ar_decoder_desc.head_ws); 0.2.1: shared-prompt decode attention (vh_attn_decode_shared); 0.2.0: bf16-MFMA perf mode of the prompt pass / NAR stage (vh_*_bf16); 0.1.2: five-product attention backward (vh_attn_rows_bwd_ws); 0.1.1: dropout fields (vh_dropout_spec) */
#define VH_MAX_TABLES 8           /* EnCodec @6 kbps: 8 codebooks (valle/config.py:15-17) */
#define VH_HEAD_DIM 64            /* every configuration of the path has d_model/n_heads = 64 */

enum { VH_OK = 0, VH_EINVAL = -1, VH_EALIGN = -2, VH_EUNSUPPORTED = -3, VH_ELAUNCH = -4,
       VH_ESTATE = -5 };

/* activation fused in a GEMM epilogue */
enum { VH_ACT_NONE = 0, VH_ACT_GELU_ERF = 1,
       VH_ACT_GELU_BWD = 2,   /* vh_linear_ex only: out = acc * gelu'(residual) (backward through the activation) */
       VH_ACT_GELU_ERF_D = 3, /* vh_linear_ex only: out = gelu(acc + bias), pre_out = gelu'(acc + bias) — the forward keeps
                                 the DERIVATIVE instead of the pre-activation (erf is evaluated once for both) ... */
       VH_ACT_MUL = 4         /* ... so that the backward is out = acc * residual[m][n], no transcendental in its epilogue */ };

/* attention mask modes (analytic; no (B,h,T,T) tensor is ever materialised) */
enum {
    VH_MASK_FULL = 0,    /* every key < kv_len visible                (NAR, valle_nar.py:94-96) */
    VH_MASK_PREFIX = 1,  /* prefix-LM of build_attn_mask(x_len, y_len) (valle/models/utils.py:17-43):
                            keys < x_len visible to all rows; rows >= x_len also see keys <= row */
    VH_MASK_EXPLICIT = 2 /* caller-supplied u8 (Tq,Tk) mask + optional u8 (B,Tk) key padding,
                            nonzero = masked (API mask convention, valle/models/modules.py:160-164) */
};

int vh_version(void);
const char* vh_last_error(void);

/* Kernel-selection knobs for benchmarking A/B runs in one process (0 = built-in default).
 * Results are identical up to fp32 summation order whatever the setting. */
enum { VH_TUNE_DECODE_VARIANT = 0,  /* decode attention: 0 = default (the ring kernel, 8 waves x 2 register sets of 32 keys, when there
                                       is one (b, head) per CU and no key split; else the 32-key burst kernel), 1 = always the burst
                                       kernel */
       VH_TUNE_DECODE_WAVES = 1,    /* waves per workgroup of the burst kernel: 4, 8 or 16 (nonzero also selects the burst kernel) */
       VH_TUNE_ROW_GROUPS = 2,      /* decode GEMMs (16 < M <= 64): 1 (default) = one workgroup per 16 rows x 16 columns,
                                       (8 rows while the grid stays within the CUs), 2 = one workgroup per
                                       16 columns (all rows), 3 = groups of 16 rows only */
       VH_TUNE_TILE_DMA = 4,        /* large-M GEMM operand staging: 0 (default) = LDS-DMA when K % 32 == 0,
                                       1 = always through registers, 2 = LDS-DMA whenever eligible */
       VH_TUNE_REDUCE_BLOCK = 3,    /* threads per workgroup of the split-K reduce: 64, 128 (default), 256 */
       VH_TUNE_FFN_FUSED = 5,       /* decode step FeedForward: 0 (default) = vh_ffn_decode (one launch split over dim_feedforward
                                       + the slab reduce) when the weights are folded and d_model <= 512 (measured slower at
                                       1024), 1 = always linear_1 and linear_2 as separate launches (vh_linear_folded +
                                       vh_linear_ws), 2 = vh_ffn_decode wherever it is supported */
       VH_TUNE_GRAPH_STEPS = 6,     /* decode graph: 0 (default) = replay in graphs of 8 consecutive steps (+ single-step graphs
                                       for the remainder); 1 = one graph launch per step */
       VH_TUNE_FFN_SLICE = 7,       /* vh_ffn_decode: hidden columns per workgroup, 0 (default) = chosen from the shape, else 16 / 32 */
       VH_TUNE_FFN_ROWS = 8,        /* vh_ffn_decode: rows per workgroup, 0 (default) = chosen from the shape, else 8 / 16 */
       VH_TUNE_LN_STATS = 9,        /* folded-LayerNorm decode GEMMs on <= 16 rows per workgroup: 0 (default) = row statistics from
                                       the operand fragments (no second read of the rows), 1 = from their own row loads */
       VH_TUNE_TAIL_SPLIT = 10,     /* vh_linear_ex with a workspace: 0 (default) = the tiles beyond the last multiple of 256 are
                                       computed as K slices + a fix-up launch when they would fill <= half of the CUs, 1 = never */
       VH_TUNE_TN_WGS = 11,         /* vh_gemm_tn: workgroups the contraction split aims at, 0 (default) = 256 (one per CU) */
       VH_TUNE_DECODE_COMBINE = 12, /* decode attention with key splits: 1 = the last workgroup of a (b, head) to arrive adds the split
                                       records in the same launch, 2 = a second launch does (same bits: split order either way);
                                       0 (default) = as measured: in the same launch at two splits (configs[4], 8 rows x 16 heads:
                                       13 us per step faster at context 2.7 k), a second launch at more (13 us per step slower at
                                       4 beams x 8 splits) */
       VH_TUNE_ATTN_BWD = 13,       /* vh_attn_rows_bwd_ws: 0 (default) / 2 = the five-product kernel + slab reduce, 1 = the two-kernel,
                                       seven-product form of vh_attn_rows_bwd (its D scratch taken from the workspace) */
       VH_TUNE_ATTN_BWD_CHUNKS = 14, /* five-product attention backward: key chunks per (batch row, head); 0 (default) = chosen from the
                                       shape (bwd_chunks in csrc/attention.hip), else that many (at least ceil(T / 256)) */
       VH_TUNE_BF16_GEMM = 15,      /* perf-mode tile GEMM: 0 (default) = form 4 where the shape allows it (N % 256 == 0, K % 128 == 0,
                                       >= 128 tiles), else per output type (16-bit outputs: form 3; fp32 output + residual: form 1);
                                       1 = 128^2 tiles, two slabs of 64 k (two workgroups per CU), 3 = one slab of 64 k, no software
                                       pipeline, four workgroups per CU, 4 = one persistent workgroup per CU, 256^2 tiles, 8 waves
                                       taking turns on the matrix pipe, requests in flight across barriers and tiles
                                       (csrc/gemm16p.hip); 2 (round 5's ring of three slabs of 32 k, removed: slower on 7 of 8 shapes)
                                       = the default */
       VH_TUNE_COUNT = 16 };
int vh_set_tuning(int knob, int value);

/* ---- dropout field of the training path -------------------------------------------------------------
 * replaces nn.Dropout at its four call sites of a training step: PositionalEncoding (valle/models/modules.py:56-58,
 * 78-80: p = 0.1 whatever config.dropout says), FeedForward (:219), EncoderLayer.dropout1 / dropout2 (:277-278, p =
 * config.dropout, default 0.1 valle/config.py:26) and TokenEmbedding (:35).  The field is never stored: it is a pure
 * function of (seed, site, row, column) that every kernel regenerates where it needs it — forward epilogue and
 * backward alike — so a training step with dropout reads and writes no mask tensors.
 *   keep(row, col) = philox4x32_7(key = (seed lo, seed hi), counter = (col / 4, row, site lo, site hi))[col % 4]
 *                    >= round(p * 2^32)                                   (Philox: Salmon et al., SC'11; 7 rounds)
 *   dropped value  = keep ? value / (1 - p) : 0                           (torch.nn.functional.dropout's scaling)
 * `row` / `col` index the logical (rows, cols) matrix the site covers (cols % 4 == 0); `site` tells the fields of one
 * step apart (layer, call site), `seed` the steps.  A NULL spec or p == 0 means "no dropout" everywhere below; p must
 * be < 1.  The stream differs from torch's generators: parity with the reference is through the exported mask
 * (vh_dropout_mask) handed to the oracle, and distributional. */
typedef struct { uint64_t seed; uint64_t site; float p; } vh_dropout_spec;
/* out[r][c] = keep(r, c) ? x[r][c] / (1 - p) : 0 for a (rows, cols) matrix (row strides ldx / ldo, multiples of 4);
 * out may be x.  Forward and backward of a free-standing nn.Dropout are this same call. */
int vh_dropout(const float* x, int ldx, float* out, int ldo, int64_t rows, int cols, const vh_dropout_spec* spec,
               void* stream);
/* keep[r * cols + c] = 1 / 0: the field itself, for tests and for handing the very same mask to a CPU oracle */
int vh_dropout_mask(uint8_t* keep, int64_t rows, int cols, const vh_dropout_spec* spec, void* stream);

/* ---- K1/K2: embedding gather (sum over n_tables codebooks) + sinusoidal position add --------
 * replaces TokenEmbedding.forward + PositionalEncoding.forward (valle/models/modules.py:33-37,
 * 78-80) and the 8-codebook sum of ValleNAR._prepare_audio_codes (valle/models/valle_nar.py:
 * 179-185).  For b<B, t<T (t < lens[b] when lens != NULL):
 *   out[b*out_bstride + (out_t0 + t)*d + :] =
 *       sum_{j<n_tables} tables[j][ ids[b*ids_bstride + t*ids_tstride + j*ids_jstride] ][:]
 *       + pe[(pos0 + t)*d + :]                      (pe == NULL: no position term)
 * `tables` is a HOST array of n_tables device pointers, each (vocab[j], d); `vocab` is a HOST array of
 * the n_tables row counts.  An id outside [0, vocab[j]) reads row 0 instead (never out of bounds) and
 * ORs VH_DEVERR_EMBED_ID into *err_flag (device int32, may be NULL); the reference's nn.Embedding raises
 * IndexError there — the host side turns the flag into that exception at its next synchronisation.
 * row_pos0 / row_t0 (device int32 (B), each may be NULL): per-row overrides of pos0 / out_t0 — a ragged batch whose
 * rows sit at different offsets (text | prompt | target per utterance) is embedded by one launch.
 * drop (NULL: none): the dropout that follows the position add (modules.py:80) applied to the sum before it is stored;
 * the field's row index is the row's offset in `out` / d (= b * (out_bstride / d) + out_t0 + t; out_bstride % d == 0
 * then), its column the column — what vh_embed_bwd regenerates from the same spec. */
enum { VH_DEVERR_EMBED_ID = 1, VH_DEVERR_TARGET = 2 };
int vh_embed_sum_pe(const int64_t* ids, int64_t ids_bstride, int64_t ids_tstride,
                    int64_t ids_jstride, const float* const* tables, const int32_t* vocab, int n_tables,
                    const float* pe, int pos0, const int32_t* lens, float* out,
                    int64_t out_bstride, int out_t0, int B, int T, int d, int32_t* err_flag,
                    const int32_t* row_pos0, const int32_t* row_t0, const vh_dropout_spec* drop, void* stream);

/* PositionalEncoding.forward as a module call (valle/models/modules.py:78-80; its dropout is vh_dropout):
 *   out[b, t, :] = x[b, t, :] + pe[(pos0 + t) * d + :]   for (B, T, d) contiguous x / out (out may be x); d % 4 == 0;
 * the caller guarantees pos0 + T rows in `pe`.  (The model paths add the position inside vh_embed_sum_pe.) */
int vh_add_pe(const float* x, const float* pe, float* out, int B, int T, int d, int pos0, void* stream);

/* ---- K3/K4: LayerNorm (eps) with optional adaptive scale/shift ------------------------------
 * replaces nn.LayerNorm (valle/models/modules.py:284) and AdaptiveLayerNorm.forward (:93-99):
 *   out[r,:] = ada_scale[:] * (gamma * (x[r,:]-mean)/sqrt(var+eps) + beta) + ada_shift[:]
 * ada_scale/ada_shift (d) may both be NULL (plain LayerNorm). */
int vh_layernorm(const float* x, const float* gamma, const float* beta, const float* ada_scale,
                 const float* ada_shift, float* out, int rows, int d, float eps, void* stream);

/* ---- K5/K9/K10/K11: out = act(LN?(A) W^T + bias) + residual ---------------------------------
 * replaces nn.Linear at valle/models/modules.py:146 (qkv), :171 (out), :221 (linear_1 +
 * exact-erf GELU, linear_2) and the heads valle_ar.py:83,158 / valle_nar.py:100.
 * A (M,K) lda; W (N,K) row-major (the nn.Linear weight as stored); bias (N)|NULL;
 * residual (M,N) ldr |NULL (may alias out); out (M,N) ldo.
 * ln_gamma/ln_beta (K)|NULL: when given (M <= 64 only) A's rows are LayerNorm-ed (eps ln_eps,
 * then optional ada_scale/ada_shift (K)) on the fly in the operand load — the decode path. */
int vh_linear(const float* A, int lda, const float* W, const float* bias, const float* residual,
              int ldr, float* out, int ldo, int M, int N, int K, int act, const float* ln_gamma,
              const float* ln_beta, const float* ada_scale, const float* ada_shift, float ln_eps,
              void* stream);

/* Same contract as vh_linear without the fused LayerNorm, plus a caller-owned workspace of
 * vh_linear_ws_bytes(M,N,K) bytes: for M <= 64 and K > 1024 (linear_2, K = dim_feedforward) the K
 * range is split over ~256 workgroups whose partial sums meet in the workspace and are added in
 * slice order by a second small kernel (bitwise reproducible; no floating-point atomics).  The same
 * is done for M > 64 when the (M,N) grid has at most 256 tiles of 128x128 and K >= 1024 (one
 * utterance through the NAR stack, a short prefill): K slices in the second grid dimension of the
 * tile kernel.  With MORE than 256 tiles the workspace serves the tile kernel's tail split instead (the tiles
 * beyond the last multiple of 256 as K slices + a fix-up launch when they would fill at most half of the CUs:
 * vh_linear_ex below) — a prompt pass of 8 x 1100 positions has 276 tiles per 512-wide projection, two rounds for
 * the work of 1.08.
 * Falls back to vh_linear when the shape does not split or workspace == NULL. */
size_t vh_linear_ws_bytes(int M, int N, int K);
int vh_linear_ws(const float* A, int lda, const float* W, const float* bias, const float* residual,
                 int ldr, float* out, int ldo, int M, int N, int K, int act, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---- K5+K6: QKV projection with the K/V rows appended in place to the cache -----------------
 * replaces qkv Linear + chunk + rearrange + torch.cat cache growth (valle/models/modules.py:
 * 146-157).  A (M=B*T, K=d) lda; Wqkv (3d, d) rows [Q | K | V], head j = rows j*64..j*64+63.
 * q_out (M, d) ldq receives the Q columns; K and V columns of row (b,t) go to
 * kcache/vcache[(b*h + head)*S_max + pos0(b) + t][0..63] with pos0(b) = cache_len ? cache_len[b] : 0.
 * Optional fused LayerNorm on A's rows as in vh_linear (M <= 64 only). */
int vh_linear_qkv(const float* A, int lda, const float* Wqkv, float* q_out, int ldq, float* kcache,
                  float* vcache, const int32_t* cache_len, int B, int T, int d_model, int n_heads,
                  int S_max, const float* ln_gamma, const float* ln_beta, const float* ada_scale,
                  const float* ada_shift, float ln_eps, void* stream);

/* ---- LayerNorm folded into the weights (decode path, M <= 64) -------------------------------
 * The LayerNorm in front of a Linear (valle/models/modules.py:271 norm1 → qkv, :278 norm2 →
 * linear_1) is rewritten so the products do not wait for the row statistics:
 *   LN(x) W^T + b = rstd * (x Wf^T - mean * c1) + c2,
 *   Wf = W * gamma (per input column), c1[n] = sum_k Wf[n,k], c2[n] = sum_k beta[k] W[n,k] + b[n].
 * vh_ln_fold prepares Wf (N,K), c1 (N), c2 (N) once per weight set (bias may be NULL);
 * vh_linear_folded / vh_linear_qkv_folded are vh_linear / vh_linear_qkv with (Wf, c1, c2) in
 * place of (W, bias, ln_gamma, ln_beta): mean / rstd are computed beside the products and applied
 * in the epilogue.  Supported: M <= 64, N % 16 == 0, K in {128, 256, 512, 1024}. */
int vh_ln_fold(const float* W, const float* gamma, const float* beta, const float* bias, float* Wf,
               float* c1, float* c2, int N, int K, void* stream);
int vh_linear_folded(const float* A, int lda, const float* Wf, const float* c1, const float* c2,
                     const float* residual, int ldr, float* out, int ldo, int M, int N, int K,
                     int act, float ln_eps, void* stream);
int vh_linear_qkv_folded(const float* A, int lda, const float* Wf, const float* c1, const float* c2,
                         float* q_out, int ldq, float* kcache, float* vcache, const int32_t* cache_len, int B,
                         int T, int d_model, int n_heads, int S_max, float ln_eps, void* stream);

/* ---- FeedForward + residual of the decode step as one launch split over dim_feedforward --------
 * replaces FeedForward.forward + the residual add of EncoderLayer.forward (valle/models/modules.py:215-221,
 * :278-279) for M <= 64 rows:   out = x + b2 + GELU(LN2(x) W1^T + b1) W2^T
 * with (w1f, c1, c2) = vh_ln_fold(W1, ln2_gamma, ln2_beta, b1).  Workgroup (slice, row group) owns SW
 * consecutive hidden columns and up to 16 rows: it computes that (rows x SW) tile of the hidden activation
 * (folded LayerNorm, exact-erf GELU), keeps it in LDS, multiplies it by W2[:, slice]^T and leaves a raw
 * (rows x d_model) partial in slab `slice` of the workspace — linear_1 -> linear_2 never crosses a kernel
 * boundary and the (M, dff) hidden activation never exists in memory.  A second small launch adds the slabs
 * in slice order (bitwise reproducible, no atomics) with b2 and the residual x.  out may alias x.
 * x (M, d_model) ldx; w1f (dff, d_model); w2 (d_model, dff) as stored; c1, c2 (dff); b2 (d_model) | NULL.
 * Supported: M <= 64, d_model in {128, 256, 512, 1024}, dff % 16 == 0, dff / 16 <= 1024 slices.
 * workspace: vh_ffn_decode_ws_bytes(M, d_model, dff) bytes (no initialisation needed). */
size_t vh_ffn_decode_ws_bytes(int M, int d_model, int dff);
int vh_ffn_decode(const float* x, int ldx, const float* w1f, const float* c1, const float* c2, const float* w2,
                  const float* b2, float* out, int ldo, int M, int d_model, int dff, float ln_eps,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- K7+K8a: multi-row attention (prefill / NAR / training forward) -------------------------
 * replaces merge_masks + F.scaled_dot_product_attention (valle/models/modules.py:160-167,
 * 175-207).  q (B,Tq,ldq) heads at columns head*64; K/V from the cache layout (B,h,S_max,64);
 * out (B,Tq,ldo).  Query row i sits at key position q_off + i (q_off = Tk - Tq).
 *   VH_MASK_FULL   : key j visible iff j < kvl(b)
 *   VH_MASK_PREFIX : key j visible iff j < kvl(b) && (j < xl(b) || (q_off+i >= xl(b) && j <= q_off+i))
 *   VH_MASK_EXPLICIT: key j visible iff !mask[i*Tk+j] && !(pad && pad[b*Tk+j])
 * kvl(b) = kv_len ? kv_len[b] : Tk ; xl(b) = x_len_dev ? x_len_dev[b] : x_len.
 * softmax scale = 1/sqrt(64). */
int vh_attn_rows(const float* q, int ldq, const float* kcache, const float* vcache, float* out,
                 int ldo, int B, int n_heads, int Tq, int Tk, int S_max, int mode, int x_len,
                 const int32_t* x_len_dev, const int32_t* kv_len, const uint8_t* mask,
                 const uint8_t* pad, void* stream);

/* vh_attn_rows in VH_MASK_EXPLICIT mode with ONE MASK PER BATCH ROW: mask (B,Tq,Tk) u8, mask_batch_stride elements between the
 * rows' masks (0 = the same (Tq,Tk) mask for all) — the 3-D attn_mask MultiHeadAttention.forward accepts
 * (valle/models/modules.py:187-188: 'b t t -> b 1 t t'); key j of row b visible to query i iff
 * !mask[b*stride + i*Tk + j] && !(pad && pad[b*Tk + j]). */
int vh_attn_rows_bmask(const float* q, int ldq, const float* kcache, const float* vcache, float* out, int ldo, int B,
                       int n_heads, int Tq, int Tk, int S_max, const uint8_t* mask, int64_t mask_batch_stride,
                       const uint8_t* pad, void* stream);

/* vh_attn_rows that also writes lse2 (B, n_heads, Tq): log2 of the softmax denominator of every query row
 * in the scaled-score units the backward kernels use (lse2 = max + log2(sum 2^(s - max)), s = q.k/8*log2 e). */
int vh_attn_rows_lse(const float* q, int ldq, const float* kcache, const float* vcache, float* out,
                     int ldo, int B, int n_heads, int Tq, int Tk, int S_max, int mode, int x_len,
                     const int32_t* x_len_dev, const int32_t* kv_len, const uint8_t* mask,
                     const uint8_t* pad, float* lse2, void* stream);

/* ---- backward of vh_attn_rows (training, Tq == Tk == T): dQ, dK, dV without materialising P -----
 * replaces the autograd backward of F.scaled_dot_product_attention (valle/models/modules.py:167) in
 * loss.backward() of training_step.  q (B*T, ldq) / kcache, vcache (B,h,S_max,64) / out (B*T, ldo) /
 * lse2 as produced by vh_attn_rows_lse, dout (B*T, lddo) the gradient of out; same mask arguments.
 * dq, dk, dv: (B*T, ldg) each with head h at columns h*64 (three column blocks of one (B*T, 3d) buffer
 * work: ldg = 3d).  dsum (B, n_heads, T) is scratch (D = rowsum(dout * out)).  Two launches, no atomics:
 * results are bitwise reproducible. */
int vh_attn_rows_bwd(const float* q, int ldq, const float* kcache, const float* vcache, const float* out,
                     int ldo, const float* dout, int lddo, const float* lse2, float* dsum, float* dq,
                     float* dk, float* dv, int ldg, int B, int n_heads, int T, int S_max, int mode,
                     int x_len, const int32_t* x_len_dev, const int32_t* kv_len, const uint8_t* mask,
                     const uint8_t* pad, void* stream);

/* The same gradients from FIVE products (round 4): one kernel with keys as lanes computes S, dP, dV, dK per (32-query tile,
 * 256-key chunk) and sends dS through LDS into the chunk's dQ partial (mfma 16x16x4 over all the chunk's keys, no cross-wave
 * sum); the partials of the ceil(T / 256) chunks land in slabs inside `ws` and a second launch adds them in chunk order
 * (T <= 256: dq is written directly, one launch).  No atomics: bitwise reproducible.  ws: vh_attn_rows_bwd_ws_bytes bytes,
 * 16-byte aligned, scratch.  D = rowsum(dout * out) is computed in the kernel (no dsum argument). */
size_t vh_attn_rows_bwd_ws_bytes(int B, int n_heads, int T);
/* the number of key chunks (= dQ slabs) vh_attn_rows_bwd_ws uses for this shape and mask family (tools, tests) */
int vh_attn_rows_bwd_chunks(int B, int n_heads, int T, int mode);
int vh_attn_rows_bwd_ws(const float* q, int ldq, const float* kcache, const float* vcache, const float* out,
                        int ldo, const float* dout, int lddo, const float* lse2, float* dq, float* dk, float* dv,
                        int ldg, int B, int n_heads, int T, int S_max, int mode, int x_len,
                        const int32_t* x_len_dev, const int32_t* kv_len, const uint8_t* mask,
                        const uint8_t* pad, void* ws, size_t ws_bytes, void* stream);

/* ---- K8b: single-row decode attention over the KV cache (HBM-bound) -------------------------
 * replaces SDPA with q-len 1 (valle/models/modules.py:167 reached via :336-338).
 * q (B, ldq); keys 0 .. cache_len[b] + len_bias - 1 of row b are attended (len_bias = 1 when the
 * new K/V row was just appended by vh_linear_qkv and cache_len is not yet incremented).
 * n_split >= 1 splits the key range of one (b,head) over n_split workgroups; then `partial`
 * must hold vh_attn_decode_ws_bytes(B, n_heads, n_split) bytes: the split records followed by one ticket word per
 * (b, head).  The records are combined in split order by a second launch (default) or, with VH_TUNE_DECODE_COMBINE = 1,
 * inside the same launch by whichever workgroup finishes last; for that form the TICKET WORDS (the last B * n_heads * 4
 * bytes, rounded up to 16) MUST BE ZERO before the first call on a workspace; every call leaves them zero (calls on one
 * workspace must be stream-ordered).  out (B, ldo). */
size_t vh_attn_decode_ws_bytes(int B, int n_heads, int n_split);
int vh_attn_decode(const float* q, int ldq, const float* kcache, const float* vcache, float* out,
                   int ldo, const int32_t* cache_len, int len_bias, int B, int n_heads, int S_max,
                   int n_split, void* partial, void* stream);
/* ---- decode attention over a shared prompt (the beams of ONE utterance) ------------------------------------------
 * replaces the same SDPA (valle/models/modules.py:167) for generate()'s num_beams rows, whose first prefix_len keys are the
 * same bits in every row (valle_ar.py:135-138 replicates one utterance): kprefix / vprefix (1, h, prefix_S, 64) are read
 * once per step for all B beams (a beam is a lane of the score tile), ksuffix / vsuffix (B, h, S_suf, 64) hold each
 * beam's generated rows, suffix_len[b] + len_bias of them attended.  Bytes per call: 2 (prefix_len + sum_b suffix) 64 h 4
 * instead of 2 B (prefix_len + suffix) 64 h 4.  Two launches: one whose workgroups take either four 32-key blocks of the
 * prefix for all beams or one (beam, head, key split) of the suffixes — a split record each — and one that merges the
 * records of every (beam, head).  fp32, deterministic (fixed merge order); results differ from vh_attn_decode's by
 * summation order only.  ceil(prefix_len / 32) + n_split_suffix <= 256.
 * partial: vh_attn_decode_shared_ws_bytes() bytes, no initialisation needed. */
size_t vh_attn_decode_shared_ws_bytes(int B, int n_heads, int prefix_len, int n_split_suffix);
int vh_attn_decode_shared(const float* q, int ldq, const float* kprefix, const float* vprefix, int prefix_len,
                          int prefix_S, const float* ksuffix, const float* vsuffix, float* out, int ldo,
                          const int32_t* suffix_len, int len_bias, int B, int n_heads, int S_suf, int n_split_suffix,
                          void* partial, size_t partial_bytes, void* stream);

/* ---- perf mode of the decode step: bf16 K/V cache (opt-in; never the parity path) ---------------
 * SURVEY.md section 7's "perf mode": the K/V cache — 93 % of the bytes a decode step reads at configs[1] — stored
 * as bf16 (B, h, S_max, 64), everything else (weights, q, softmax, accumulators, residual stream) fp32 as in the parity
 * path.  Greedy tokens are NOT guaranteed to equal the reference's; teacher-forced logits agree to atol 5e-2.
 *   vh_kv_to_bf16            narrows the first `rows` rows of n_streams (b, head) streams of an fp32 cache (row stride
 *                            64, stream stride S_src * 64) into a bf16 cache (stream stride S_dst * 64), round to
 *                            nearest even: the prompt pass runs in fp32 and is narrowed once;
 *   vh_linear_qkv_folded_kv16  vh_linear_qkv_folded for one new row per sequence with the K / V rows appended as bf16;
 *   vh_attn_decode_kv16      vh_attn_decode (one (b, head) per workgroup, no key split) over the bf16 cache. */
int vh_kv_to_bf16(const float* src, uint16_t* dst, int n_streams, int rows, int S_src, int S_dst, void* stream);
int vh_linear_qkv_folded_kv16(const float* A, int lda, const float* Wf, const float* c1, const float* c2, float* q_out,
                              int ldq, uint16_t* kcache16, uint16_t* vcache16, const int32_t* cache_len, int B,
                              int d_model, int n_heads, int S_max, float ln_eps, void* stream);
int vh_attn_decode_kv16(const float* q, int ldq, const uint16_t* kcache16, const uint16_t* vcache16, float* out, int ldo,
                        const int32_t* cache_len, int len_bias, int B, int n_heads, int S_max, void* stream);

/* ---- K12/K13: greedy sampling + decode-state update + next-token embedding ------------------
 * replaces topk_sampling(top_k=1) (valle/models/utils.py:46-68: argmax, lowest index on ties),
 * the EOS bookkeeping valle/models/valle_ar.py:167-171 and the re-embedding :143-144 for the
 * next step.  `codes` (B, codes_stride) int64 is the growing `prompt_codes` of the reference
 * (BOS + prompt pre-filled); audio_pos[b] = index of the token produced now.  Per row b:
 *   p = audio_pos[b]; tok = (codes[b][p-1]==eos) ? eos : argmax(logits[b,:V]); codes[b][p] = tok;
 *   if (tok==eos) eos_count[p - (pos_base?pos_base[b]:0)] += 1;
 *   x_next[b,:] = audio_emb[tok,:] + pe[p*d + :]; audio_pos[b] = p+1; cache_len[b] += 1.
 * eos_count[s] == B means every row had finished at step s (the reference's break, :169-170);
 * the host polls it every few steps instead of synchronising every step. */
int vh_greedy_step(const float* logits, int ldl, int V, int eos, int64_t* codes,
                   int64_t codes_stride, int32_t* eos_count, const int32_t* pos_base,
                   const float* audio_emb, const float* pe, int32_t* audio_pos, int32_t* cache_len,
                   float* x_next, int B, int d, void* stream);

/* ---- K11 + K12/K13 in ONE launch (opt-in; DESIGN.md 3.20 has the A/B) -----------------------------
 * logits[b,:V] = x[b,:] . proj_w^T (the head: no bias, valle_ar.py:29,158) and, in the same launch, everything
 * vh_greedy_step does with them: every 16-column workgroup publishes its (largest logit, lowest column) per row, the
 * last workgroup to arrive per group of rows picks the tokens, does the EOS bookkeeping and builds x_next.  Same
 * results as vh_linear + vh_greedy_step bit for bit (the logits are the same sums, the arg-max has the same tie rule).
 * B <= 64, d in {128, 256, 512, 1024}; x_next may be x.  workspace: vh_head_greedy_ws_bytes(B, V) bytes, 16-byte
 * aligned, ZEROED once before the first call (the launches leave it zeroed where it matters). */
size_t vh_head_greedy_ws_bytes(int B, int V);
int vh_head_greedy(const float* x, int ldx, const float* proj_w, float* logits, int ldl, int V, int eos,
                   int64_t* codes, int64_t codes_stride, int32_t* eos_count, const int32_t* pos_base,
                   const float* audio_emb, const float* pe, int32_t* audio_pos, int32_t* cache_len, float* x_next,
                   int B, int d, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K12 (stochastic): temperature / top-k / top-p sampling + the same state update ----------
 * replaces topk_sampling (valle/models/utils.py:46-68) incl. the published semantics of
 * transformers==4.38.2 top_k_top_p_filtering (top-k keeps ties of the k-th score, top_k <= 0 keeps
 * all; top-p drops, from the smallest, entries whose cumulative probability is <= 1 - top_p and
 * always keeps the largest), torch.multinomial and the log-prob gather; then the bookkeeping of
 * vh_greedy_step plus sum_logprobs[b] += logprob while row b has not finished (valle_ar.py:167).
 * Randomness is a counter-based generator keyed on (seed, row, audio_pos[b]) — replaying a captured
 * graph draws fresh numbers; the stream is NOT torch's, so parity is distributional.  V <= 2048. */
int vh_sample_step(const float* logits, int ldl, int V, int eos, int top_k, float top_p,
                   float temperature, uint64_t seed, int64_t* codes, int64_t codes_stride,
                   int32_t* eos_count, const int32_t* pos_base, float* sum_logprobs,
                   const float* audio_emb, const float* pe, int32_t* audio_pos, int32_t* cache_len,
                   float* x_next, int B, int d, void* stream);

/* ---- composite: one AR decode step / hipGraph replay ----------------------------------------
 * The 5 launches per layer of one decode step (LN1+QKV+append, decode attention, out-proj+
 * residual, FeedForward slices, slab reduce + residual) plus head GEMM and vh_greedy_step, enqueued natively
 * (valle/models/valle_ar.py:141-171 with modules.py:336-352).  */
typedef struct {
    const float *ln1_g, *ln1_b, *wqkv, *wo, *bo, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;
    float *kcache, *vcache;           /* this layer's (B,h,S_max,64) caches */
    /* optional vh_ln_fold outputs for (ln1, wqkv) and (ln2, w1, b1); all NULL → LayerNorm applied in
     * the operand load from ln*_g / ln*_b.  Used by the decode step only. */
    const float *wqkv_f, *qkv_c1, *qkv_c2, *w1_f, *w1_c1, *w1_c2;
    /* shared-prompt decoding (vh_ar_decoder_desc.prefix_len > 0): this layer's prompt K / V, (1, h, prefix_S, 64), written
     * once by a one-row prompt pass and read by every beam; kcache / vcache then hold only the beams' generated rows */
    const float *kprefix, *vprefix;
    /* perf mode of the decode step, second half (round 6; with vh_ar_decoder_desc.kv_bf16): h16 copies (vh_h16_format) of the four
     * matrices a step streams — wqkv_f16 (3d, d) and w1_f16 (dff, d) of the FOLDED weights above, wo16 (d, d), w2_16 (d, dff).
     * All four non-NULL: the step's QKV, out-projection and FeedForward launches read them (half the weight bytes; fp32
     * accumulators, fp32 rows, c1 / c2 / biases fp32); any NULL: the fp32 matrices, as before. */
    const uint16_t *wqkv_f16, *wo16, *w1_f16, *w2_16;
} vh_layer;

typedef struct {
    int B, d_model, n_heads, dff, n_layers, S_max, V, eos, n_split;
    float ln_eps;
    const vh_layer* layers;           /* host array of n_layers */
    const float *proj_w;              /* (V, d) */
    const float *audio_emb, *audio_pe;
    float *x;                         /* (B,d) residual stream, holds the current token's embedding */
    float *q, *attn, *hidden, *logits;/* (B,d) (B,d) (B,dff) (B,ldl) scratch; ldl = round_up(V,4) */
    void *attn_partial;               /* vh_attn_decode_ws_bytes() or NULL when n_split == 1 */
    void *gemm_ws;                    /* vh_linear_ws_bytes(B, d_model, dff) bytes or NULL */
    size_t gemm_ws_bytes;
    int32_t *cache_len, *audio_pos, *eos_count;
    const int32_t *pos_base;          /* (B) or NULL */
    int64_t *codes;                   /* (B, codes_stride) growing code sequence */
    int64_t codes_stride;
    /* sampling (valle/config.py:48-51): top_k == 1 → vh_greedy_step, else vh_sample_step */
    int top_k;
    float top_p, temperature;
    uint64_t seed;
    float *sum_logprobs;              /* (B) or NULL */
    /* optional: workspace of vh_ffn_decode_ws_bytes(B, d_model, dff) bytes; with it and folded weights in every
     * layer the FeedForward of a layer is vh_ffn_decode (two launches instead of three). */
    void *ffn_ws;
    size_t ffn_ws_bytes;
    /* perf mode (opt-in): nonzero = every layer's kcache / vcache point at bf16 caches (B,h,S_max,64) and the step
     * runs vh_linear_qkv_folded_kv16 + vh_attn_decode_kv16; needs folded weights in every layer and n_split == 1. */
    int kv_bf16;
    /* shared prompt (the reference's generate(): ONE utterance replicated over num_beams rows, valle_ar.py:135-138 — every
     * beam's prompt K/V is the same bits).  prefix_len > 0: the first prefix_len keys of every row live ONCE in
     * layers[i].kprefix / vprefix ((1, h, prefix_S, 64)); layers[i].kcache / vcache are (B, h, S_max, 64) caches of the
     * GENERATED rows only, cache_len[b] counts those, and the step's attention is vh_attn_decode_shared (n_split = key
     * splits of the suffix part).  attn_partial must hold vh_attn_decode_shared_ws_bytes(B, n_heads, prefix_len, n_split)
     * bytes (attn_partial_bytes says how many it holds).  Not with kv_bf16. */
    int prefix_len, prefix_S;
    size_t attn_partial_bytes;
    /* optional (opt-in): zeroed workspace of vh_head_greedy_ws_bytes(B, V) bytes; with it and top_k == 1 the head and the
     * greedy step are ONE launch (vh_head_greedy) instead of vh_linear + vh_greedy_step. */
    void *head_ws;
    size_t head_ws_bytes;
    /* optional: a device uint64 ADDED to `seed` by every sampling step (top_k != 1).  A captured graph freezes `seed`; a decoder
     * that serves many generate() calls keeps the call's seed here and rewrites it between calls (valle2_amd/valle_ar.py keeps
     * a decoder per shape: graphs, caches and workspaces survive the call). */
    const uint64_t *seed_dev;
    /* optional (with kv_bf16): an h16 copy of proj_w (V, d) for the head GEMM of the step */
    const uint16_t *proj_w16;
} vh_ar_decoder_desc;

typedef struct vh_ar_decoder vh_ar_decoder;
vh_ar_decoder* vh_ar_decoder_create(const vh_ar_decoder_desc* desc);   /* NULL on bad desc */
void vh_ar_decoder_destroy(vh_ar_decoder* dec);
/* enqueue one step eagerly (no graph) */
int vh_ar_decoder_step(vh_ar_decoder* dec, void* stream);
/* capture one step into a hipGraph on `stream` (which must be a non-null, idle stream) */
int vh_ar_decoder_capture(vh_ar_decoder* dec, void* stream);
/* replay the captured step n_steps times on `stream` */
int vh_ar_decoder_replay(vh_ar_decoder* dec, int n_steps, void* stream);
/* eager steps with hipEvents on every decode-attention launch.  *kernel_ms (optional): mean elapsed time between
 * the start and stop events attached to the kernel's own dispatch (hipExtLaunchKernelGGL) — the kernel's duration
 * as the profiler sees it.  *mean_ms: mean event-to-event time of marker events recorded before and after the
 * launch (an upper bound: it carries event + dispatch overhead); *floor_ms (optional): the same marker bracket
 * with nothing inside it, once per step.  All in milliseconds.  Synchronises the stream; measurement only. */
int vh_ar_decoder_profile_attn(vh_ar_decoder* dec, int n_steps, void* stream, float* mean_ms,
                               float* floor_ms, float* kernel_ms);

/* ---- training path: backward of the row ops (loss.backward() of valle/models/valle_ar.py:86) -----
 * The plain backward GEMMs (dX = dY.W, dW = dY^T.X and the five attention products) are library
 * GEMMs issued by the host; these entry points are the hand-written non-GEMM halves.  Gradient
 * buffers marked "+=" are accumulated with fp32 atomics: the caller zeroes them first. */
/* LayerNorm / AdaptiveLayerNorm backward (modules.py:284, :93-99).  y = s*(gamma*xhat+beta)+t:
 * dx written; dgamma += , dbeta += , and when ada_scale != NULL dscale += , dshift += (all (d)).
 * dres (rows, d) | NULL: added to dx — the gradient of the pre-norm block's residual branch, which bypasses the norm
 * (x feeds norm AND residual add, modules.py:271-279): no separate elementwise add.  dx must not alias dres.
 * dcolsum (d) | NULL: += the column sums of the rows written to dx — the bias gradient of the Linear whose output x
 * was (out-projection / linear_2), without a column-sum launch.
 * dx_drop (rows, d) + drop (both or neither): x = residual + dropout(branch) (modules.py:277-278): the gradient of the
 * BRANCH is the dropped dx.  It is written to dx_drop in the same pass (field row = row, column = column) and dcolsum
 * then sums dx_drop — the branch's Linear bias sits under the dropout — while dx stays the residual stream's gradient. */
int vh_layernorm_bwd(const float* x, const float* gamma, const float* beta, const float* ada_scale,
                     const float* dy, float* dx, float* dgamma, float* dbeta, float* dscale,
                     float* dshift, const float* dres, float* dcolsum, float* dx_drop, const vh_dropout_spec* drop,
                     int rows, int d, float eps, void* stream);
/* exact-erf GELU on a saved pre-activation: dh == NULL → out = gelu(pre); else out = dh*gelu'(pre) */
int vh_gelu(const float* pre, const float* dh, float* out, int64_t n, void* stream);
/* P = softmax(S*scale + mask) in place over rows of (B,h,Tq,Tk) with row stride ld >= Tk; same mask
 * semantics as vh_attn_rows */
int vh_softmax_rows(float* S, int ld, int B, int n_heads, int Tq, int Tk, float scale, int mode,
                    int x_len, const int32_t* x_len_dev, const int32_t* kv_len, const uint8_t* mask,
                    const uint8_t* pad, void* stream);
/* dS = scale * P o (dP - rowsum(dP o P)), in place on dP (both with row stride ld) */
int vh_softmax_bwd(const float* P, float* dP, int ld, int64_t rows, int Tk, float scale, void* stream);
/* mean cross entropy over `rows` rows of (rows, V) logits (F.cross_entropy, valle_ar.py:86):
 * *loss = mean(lse - logit[target]); dlogits (may be NULL) = (softmax - onehot) / rows.
 * A target outside [0, V) (this includes torch's ignore_index -100, which the collate format never
 * emits: it pads with 0, valle/collate.py:35,65) contributes lse only, gets no one-hot term and ORs
 * VH_DEVERR_TARGET into *err_flag (device int32, may be NULL). */
int vh_cross_entropy(const float* logits, int ld, int V, const int64_t* target, float* loss,
                     float* dlogits, int ldd, int rows, int32_t* err_flag, void* stream);
/* embedding backward: dtable[ids[b,t], :] += dout[b, out_t0 + t, :]; ids outside [0, vocab) are skipped
 * (and flagged as in vh_embed_sum_pe).  drop (NULL: none): the forward's vh_embed_sum_pe dropout — dout is multiplied
 * by the regenerated field on the way in (dout_bstride % d == 0 then). */
int vh_embed_bwd(const int64_t* ids, int64_t ids_bstride, int64_t ids_tstride, const float* dout,
                 int64_t dout_bstride, int out_t0, float* dtable, int vocab, int B, int T, int d,
                 int32_t* err_flag, const vh_dropout_spec* drop, void* stream);
/* bias gradient: out[c] += sum_r x[r, c] */
int vh_colsum(const float* x, int ld, float* out, int rows, int cols, void* stream);

/* ---- AdaptiveLayerNorm projections of a whole stack (training) ------------------------------------
 * out[i, :] = emb W_i^T + b_i for the n project_layer Linears of a stack (valle/models/modules.py:94-96; n = 2 per
 * EncoderLayer), one (1, K) stage embedding for all, in ONE launch; and their backward in one launch:
 *   dw_i (N, K) = dout_i^T emb (an outer product: the Linear saw one input row), db_i = dout_i   — both WRITTEN,
 *   demb (K) += sum_i dout_i W_i   (fp32 atomics; the caller zeroes it; NULL: not wanted).
 * items: DEVICE array of n descriptors; dw / db / b may be NULL.  N = 2 K for the reference's modules; K % 4 == 0,
 * K <= 2048. */
typedef struct { const float* w; const float* b; float* dw; float* db; } vh_adaproj_item;
int vh_adaproj_fwd(const vh_adaproj_item* items, int n, const float* emb, float* out, int N, int K, void* stream);
int vh_adaproj_bwd(const vh_adaproj_item* items, int n, const float* emb, const float* dout, float* demb, int N, int K,
                   void* stream);

/* ---- NAR stage sampler -------------------------------------------------------------------------
 * replaces `Categorical(logits=logits / temperature).sample()` of ValleNAR.generate
 * (valle/models/valle_nar.py:160) for `rows` rows of (rows, V) logits (row stride ld):
 *   tokens[r * tokens_stride] ~ softmax(logits[r] / temperature)     (inverse CDF in index order, counter
 *   RNG keyed on (seed, r, stream_id): the stream differs from torch's, parity is distributional), or
 *   the arg-max with the lowest index on ties when greedy != 0 (what the parity tests pin).
 * logprob (rows floats, may be NULL) receives log p(token) (0 when greedy). */
int vh_categorical_rows(const float* logits, int ld, int V, int rows, float temperature, int greedy,
                        uint64_t seed, uint32_t stream_id, int64_t* tokens, int64_t tokens_stride,
                        float* logprob, void* stream);

/* ---- optimizer step over one flat fp32 buffer ------------------------------------------------
 * replaces optim.AdamW(fused) (valle/models/valle_ar.py:182-194), the global-norm clip
 * `gradient_clip_val` of the Trainer (valle/train_model.py:31-32) and the 1/world scale after the
 * gradient all-reduce, in two launches with no host read:
 *   norm = grad_scale * ||grad||_2 (double accumulation, fixed order: reproducible);
 *   coef = max_norm > 0 ? min(1, max_norm / (norm + 1e-6)) : 1;   g = grad * grad_scale * coef;
 *   p *= 1 - lr*wd;  m += (g - m)(1-b1);  v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)       (torch.optim.AdamW, amsgrad off).
 * param/grad/exp_avg/exp_avg_sq: n floats each (n % 4 == 0); zero_grad != 0 clears grad afterwards;
 * norm_out (1 float, optional) receives norm; workspace: vh_adamw_ws_bytes() bytes.
 * block_slot / slot_step (both NULL: every element updates with `step`): torch.optim.AdamW keeps a step
 * count per parameter and skips parameters whose grad is None (NAR trains one stage per step: the other
 * stages' heads and embeddings receive nothing, valle_nar.py:76).  block_slot (n / 64 int32, device) maps
 * each 64-float block to its parameter slot, slot_step (device) holds the slot's own step count for this
 * update, 0 = skip the slot entirely (no decay, no moments, no move).  Needs n % 64 == 0.
 * guard (optional, one int32 on the device): nonzero when the update launch starts = parameters and moments are not
 * touched (the error flag of the range-checking kernels: a step that saw a bad id never reaches the parameters, and
 * the host can read the flag a step later instead of synchronising every step); the gradient is still cleared when
 * zero_grad != 0, so the buffer the next backward accumulates into is zero either way. */
size_t vh_adamw_ws_bytes(void);
int vh_adamw_flat(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  float max_norm, int zero_grad, void* workspace, float* norm_out, const int32_t* block_slot,
                  const int32_t* slot_step, const int32_t* guard, void* stream);

/* ---- training forward / backward products on the LDS-DMA tile machine --------------------------
 * vh_linear_ex = vh_linear (same tile kernels, M of any size) with the two epilogues training needs:
 *   pre_out != NULL : the pre-activation acc + bias is ALSO stored to pre_out (row stride ldp) — the
 *                     forward of FeedForward.linear_1 keeps it for the GELU backward (modules.py:221);
 *   act == VH_ACT_GELU_ERF_D (pre_out required) / VH_ACT_MUL (residual required, no bias, no pre_out): the pair the
 *                     training step uses — forward stores gelu'(pre) in pre_out, backward multiplies by it; the
 *                     GELU'-epilogue of VH_ACT_GELU_BWD (erf + exp on 64 values per lane and tile) held the biggest
 *                     product of the backward at 75 % MFMA-busy against 84 % for a plain epilogue.
 *   act == VH_ACT_GELU_BWD : out = acc * gelu'(residual[m][n]) — the backward through nn.GELU fused into
 *                     the dX = dY . W product of linear_2 (residual = the saved pre-activation).
 *   dcolsum != NULL (N % 128 == 0): the column sums of `out` are ADDED to dcolsum (N floats, fp32 atomics, one per
 *                     column and tile) — the gradient of linear_1's bias falls out of the product that makes its
 *                     pre-activation's gradient, without a column-sum launch.
 * The backward's dX = dY . W products run on this NT form with W handed over transposed (vh_transpose),
 * so K here is the forward's N: pad it to a multiple of 32 with zero columns for the fast kernel.
 *   workspace (optional, vh_linear_ex_ws_bytes(M, N, K) bytes, 16-byte aligned): lets the product split its TAIL.
 *                     128 x 128 tiles are dealt to 256 CUs, so T tiles cost ceil(T / 256) rounds: 320 tiles (10240
 *                     training positions x a 512-wide projection) take as long as 512.  With a workspace the tiles
 *                     beyond the last multiple of 256 — when they would occupy at most half of the CUs — are computed
 *                     as 2..8 K slices each (one workgroup per slice, raw sums to the workspace) and a small second
 *                     launch adds the slices in slice order and applies the epilogue: deterministic, but those tiles'
 *                     sums are associated differently from an unsplit run.  vh_linear_ex_ws_bytes returns 0 for a
 *                     shape that is not split (<= 16 MiB otherwise); NULL or a smaller workspace = never split. */
/*   drop (NULL: none; act NONE / GELU_ERF / GELU_ERF_D, N % 4 == 0, K % 32 == 0): out = dropout(act(acc + bias)) +
 *                     residual with the field indexed by (m, n) — dropout1 / dropout2 inside the out-projection's /
 *                     linear_2's epilogue (modules.py:277-278) and FeedForward's dropout inside linear_1's (:219).
 *                     With VH_ACT_GELU_ERF_D the stored derivative is multiplied by the same keep / (1 - p) factor, so
 *                     the backward's VH_ACT_MUL epilogue already yields the gradient through dropout AND GELU. */
size_t vh_linear_ex_ws_bytes(int M, int N, int K);
int vh_linear_ex(const float* A, int lda, const float* W, const float* bias, const float* residual, int ldr,
                 float* out, int ldo, float* pre_out, int ldp, float* dcolsum, int M, int N, int K, int act,
                 const vh_dropout_spec* drop, void* workspace, size_t workspace_bytes, void* stream);

/* out (cols, ldo) = in (rows, cols)^T; out rows are zero-filled from `rows` up to ldo. */
int vh_transpose(const float* in, int ldi, int rows, int cols, float* out, int ldo, void* stream);
/* The same for n matrices in ONE launch (every Linear weight of the stack, once per optimizer step, for the dX
 * products of the backward): `items` is a DEVICE array of n descriptors; tile0 = number of 32x32 output tiles of the
 * items before this one (tiles of an item: ceil(cols / 32) * ceil(ldo / 32)), total_tiles their sum. */
typedef struct { const float* in; float* out; int32_t ldi, rows, cols, ldo, tile0, pad_; } vh_transpose_item;
int vh_transpose_many(const vh_transpose_item* items, int n, int total_tiles, void* stream);

/* Weight gradient dW = dY^T . X of `loss.backward()` (valle/models/valle_ar.py:86):
 *   C (NI, NJ) = A^T . B,  A (M, NI) row stride lda, B (M, NJ) row stride ldb — both stored with the
 * contraction index (the token) as the row, read in place (no transposed copies).  The contraction is cut
 * into slices whose partial tiles are summed in fixed order from `workspace` (vh_gemm_tn_ws_bytes; bitwise
 * reproducible, no atomics).  lda/ldb/ldc multiples of 4 covering the row padded to 4 floats. */
size_t vh_gemm_tn_ws_bytes(int M, int NI, int NJ);
int vh_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int NI, int NJ,
               void* workspace, size_t workspace_bytes, void* stream);

/* General batched fp32 GEMM of the backward pass: C[b,h] = op(A[b,h]) . op(B[b,h]) on the fp32 matrix
 * cores (128x128x32 tiles).  a_kmajor = 0: A stored (M,K) k contiguous; 1: stored (K,M) m contiguous.
 * b_kmajor = 0: B stored (N,K) (an nn.Linear weight as stored); 1: stored (K,N).  C is (M,N) ldc.
 * Two-level batch index (b < batch, h < H) with element strides, so attention operands are read in
 * place from (B,T,h,64) / (B,h,T,64) / (B,h,T,T) tensors.  Leading dimensions and strides must be
 * multiples of 4; ragged M/N/K are guarded (buffers padded to the leading dimension).
 * k_splits > 1 cuts K into that many slices handled by separate workgroups whose partial tiles are
 * ADDED to C with fp32 atomics (the caller zeroes C; summation order is not fixed) — for weight
 * gradients, where the output is a few tiles and K is the whole token count. */
int vh_gemm_batched(const float* A, int lda, int64_t sAb, int64_t sAh, int a_kmajor, const float* B,
                    int ldb, int64_t sBb, int64_t sBh, int b_kmajor, float* C, int ldc, int64_t sCb,
                    int64_t sCh, int M, int N, int K, int batch, int H, int k_splits, void* stream);

/* ---- composite: full-sequence transformer forward (prefill / NAR stage / training forward) --
 * valle/models/modules.py:305-352 without cache input: x (B*T, d) in/out in place, every
 * layer's K/V written to its cache at positions 0..T-1.  scratch: xn (B*T,d), q (B*T,d),
 * attn (B*T,d), hidden (B*T,dff).  ada_* per layer (L,2,2,d) = [layer][norm1|norm2][scale|shift]
 * or NULL for plain LayerNorm. */
typedef struct {
    int B, T, d_model, n_heads, dff, n_layers, S_max, mode, x_len;
    float ln_eps;
    const vh_layer* layers;
    const float* ada;                 /* (L,2,2,d) or NULL */
    const int32_t *x_len_dev, *kv_len;
    const uint8_t *mask, *pad;
    float *x, *xn, *q, *attn, *hidden;
    /* optional split-K workspace for the out-projection and linear_2 when B*T is small (few 128x128 tiles,
     * long K): the largest of vh_linear_ws_bytes(B*T, d_model, d_model), (B*T, d_model, dff), (B*T, dff, d_model) */
    void *gemm_ws;
    size_t gemm_ws_bytes;
    /* optional: the stack's INPUT rows (B*T, d) when they must stay untouched (the module API returns a new
     * tensor, modules.py:341-349): layer 0 reads its LayerNorm input and its residual from x_in and the stack
     * writes its output to x — no copy of the input.  NULL: x is input and output. */
    const float *x_in;
} vh_forward_desc;
int vh_transformer_forward(const vh_forward_desc* desc, void* stream);

/* ---- perf mode of the MFMA-bound legs: bf16 operands, fp32 accumulate (opt-in; never the parity path) ------------
 * SURVEY.md section 7's "perf mode" for the prompt pass (valle/models/valle_ar.py:141-158 at kv_cache=None) and the NAR
 * stage forward (valle_nar.py:87-100): every matrix product on v_mfma_f32_32x32x16_bf16 with bf16 operands and fp32
 * accumulators; the residual stream, LayerNorm statistics, biases, softmax and the heads stay fp32.  The reference itself
 * asks for reduced-precision matmuls on a GPU (valle/utils.py:11, set_float32_matmul_precision('high')).  Teacher-forced
 * logits agree with the reference to atol 5e-2; greedy tokens are NOT guaranteed.  bf16 = the upper 16 bits
 * of an fp32, round to nearest even, passed as uint16_t.
 *   vh_to_bf16              narrows a (rows, cols) fp32 matrix (weights: once per weight set); cols % 8 == 0.
 *   vh_layernorm_bf16       vh_layernorm with a bf16 output (the A operand of the product that follows); d % 8 == 0.
 *   vh_linear_bf16          out = act(A W^T + bias) + residual, A (M,K) lda and W (N,K) bf16, bias (N) / residual (M,N) ldr
 *                           fp32; out fp32 (M,N) ldo, or bf16 when out_bf16 != 0 (no residual then).  N % 128 == 0,
 *                           K % 64 == 0, act NONE / GELU_ERF.
 *   vh_linear_qkv_bf16      vh_linear_qkv with bf16 operands: q_out (M, d) ldq bf16, K / V rows appended to bf16 caches
 *                           (B,h,S_max,64) — the layout vh_attn_decode_kv16 streams, so a perf-mode generate needs no
 *                           narrowing pass.  d_model % 128 == 0.  q_out holds q' = q / sqrt(64) * log2(e), scaled in fp32
 *                           before the one narrowing (since 0.2.3): the form vh_attn_rows_bf16 consumes.
 *   vh_attn_rows_bf16       vh_attn_rows over bf16 q' / K / V with a bf16 output; analytic masks only (FULL / PREFIX).
 *                           q' is PRE-SCALED (above): the weights are 2^(q'.k - max), i.e. softmax(q.k / 8) of the unscaled q. */
/* The 16-bit operand format ("h16") of everything called *bf16 / *kv16 / *16 in this header — operands of the perf-mode kernels and
 * the narrow K/V cache of the decode step: 0 = IEEE fp16 (the default build: same MFMA rate as bf16, 11 bits of significand
 * against 8 — teacher-forced logits of the 24-layer stack 5.5e-3 from the reference against 4.2e-2), 1 = bf16 (the round-5
 * format, -DVH_PERF_BF16).  The entry points keep their round-5 names; the narrowing is IEEE (overflow -> infinity, NaN stays NaN). */
int vh_h16_format(void);
int vh_to_bf16(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t rows, int cols, void* stream);
int vh_layernorm_bf16(const float* x, const float* gamma, const float* beta, const float* ada_scale,
                      const float* ada_shift, uint16_t* out, int rows, int d, float eps, void* stream);
int vh_linear_bf16(const uint16_t* A, int lda, const uint16_t* W, const float* bias, const float* residual, int ldr,
                   void* out, int ldo, int out_bf16, int M, int N, int K, int act, void* stream);
int vh_linear_qkv_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv, uint16_t* q_out, int ldq, uint16_t* kcache16,
                       uint16_t* vcache16, const int32_t* cache_len, int B, int T, int d_model, int n_heads, int S_max,
                       void* stream);
int vh_attn_rows_bf16(const uint16_t* q, int ldq, const uint16_t* kcache16, const uint16_t* vcache16, uint16_t* out, int ldo,
                      int B, int n_heads, int Tq, int Tk, int S_max, int mode, int x_len, const int32_t* x_len_dev,
                      const int32_t* kv_len, void* stream);

/* vh_transformer_forward in perf mode.  `layers` supplies the fp32 LayerNorm parameters and biases (its weight matrices
 * and caches are not touched); layers16[i] the bf16 copies of the four matrices (vh_to_bf16) and the layer's bf16 caches.
 * x (B*T, d) fp32 in/out (x_in as in vh_forward_desc); scratch: xn16, q16, attn16 (B*T, d) and hidden16 (B*T, dff) bf16. */
typedef struct { const uint16_t *wqkv, *wo, *w1, *w2; uint16_t *kcache16, *vcache16; } vh_layer16;
typedef struct {
    int B, T, d_model, n_heads, dff, n_layers, S_max, mode, x_len;
    float ln_eps;
    const vh_layer* layers;
    const vh_layer16* layers16;
    const float* ada;                 /* (L,2,2,d) or NULL */
    const int32_t *x_len_dev, *kv_len;
    float* x;
    const float* x_in;
    uint16_t *xn16, *q16, *attn16, *hidden16;
} vh_forward16_desc;
int vh_transformer_forward_bf16(const vh_forward16_desc* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VALLE_HIP_H */
