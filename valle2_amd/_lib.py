"""ctypes binding of libvalle_hip.so (include/valle_hip.h).

There is no CPU fallback anywhere in this package: `lib()` raises when the shared library is
missing or a HIP device is not visible, and every wrapper raises `VhError` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch  # imported first so the process-wide HIP runtime is the one torch already loaded

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / 'csrc' / 'libvalle_hip.so'

c_f32p = C.c_void_p   # device pointers travel as integers
c_i32p = C.c_void_p
c_i64p = C.c_void_p
c_u8p = C.c_void_p


class VhError(RuntimeError):
    pass


class VhDropoutSpec(C.Structure):
    """include/valle_hip.h vh_dropout_spec: (seed, site, p) of one dropout field."""
    _fields_ = [('seed', C.c_uint64), ('site', C.c_uint64), ('p', C.c_float)]


c_dropp = C.POINTER(VhDropoutSpec)


class VhLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'ln1_g', 'ln1_b', 'wqkv', 'wo', 'bo', 'ln2_g', 'ln2_b', 'w1', 'b1', 'w2', 'b2',
        'kcache', 'vcache', 'wqkv_f', 'qkv_c1', 'qkv_c2', 'w1_f', 'w1_c1', 'w1_c2', 'kprefix', 'vprefix',
        'wqkv_f16', 'wo16', 'w1_f16', 'w2_16')]


class VhArDecoderDesc(C.Structure):
    _fields_ = [
        ('B', C.c_int), ('d_model', C.c_int), ('n_heads', C.c_int), ('dff', C.c_int),
        ('n_layers', C.c_int), ('S_max', C.c_int), ('V', C.c_int), ('eos', C.c_int),
        ('n_split', C.c_int), ('ln_eps', C.c_float),
        ('layers', C.POINTER(VhLayer)),
        ('proj_w', C.c_void_p), ('audio_emb', C.c_void_p), ('audio_pe', C.c_void_p),
        ('x', C.c_void_p), ('q', C.c_void_p), ('attn', C.c_void_p), ('hidden', C.c_void_p),
        ('logits', C.c_void_p), ('attn_partial', C.c_void_p), ('gemm_ws', C.c_void_p),
        ('gemm_ws_bytes', C.c_size_t),
        ('cache_len', C.c_void_p), ('audio_pos', C.c_void_p), ('eos_count', C.c_void_p),
        ('pos_base', C.c_void_p), ('codes', C.c_void_p), ('codes_stride', C.c_int64),
        ('top_k', C.c_int), ('top_p', C.c_float), ('temperature', C.c_float), ('seed', C.c_uint64),
        ('sum_logprobs', C.c_void_p), ('ffn_ws', C.c_void_p), ('ffn_ws_bytes', C.c_size_t), ('kv_bf16', C.c_int),
        ('prefix_len', C.c_int), ('prefix_S', C.c_int), ('attn_partial_bytes', C.c_size_t),
        ('head_ws', C.c_void_p), ('head_ws_bytes', C.c_size_t), ('seed_dev', C.c_void_p), ('proj_w16', C.c_void_p),
    ]


class VhForwardDesc(C.Structure):
    _fields_ = [
        ('B', C.c_int), ('T', C.c_int), ('d_model', C.c_int), ('n_heads', C.c_int),
        ('dff', C.c_int), ('n_layers', C.c_int), ('S_max', C.c_int), ('mode', C.c_int),
        ('x_len', C.c_int), ('ln_eps', C.c_float),
        ('layers', C.POINTER(VhLayer)), ('ada', C.c_void_p),
        ('x_len_dev', C.c_void_p), ('kv_len', C.c_void_p), ('mask', C.c_void_p), ('pad', C.c_void_p),
        ('x', C.c_void_p), ('xn', C.c_void_p), ('q', C.c_void_p), ('attn', C.c_void_p),
        ('hidden', C.c_void_p), ('gemm_ws', C.c_void_p), ('gemm_ws_bytes', C.c_size_t), ('x_in', C.c_void_p),
    ]


class VhLayer16(C.Structure):
    """include/valle_hip.h vh_layer16: bf16 copies of a layer's four matrices + its bf16 K / V caches (perf mode)."""
    _fields_ = [(n, C.c_void_p) for n in ('wqkv', 'wo', 'w1', 'w2', 'kcache16', 'vcache16')]


class VhForward16Desc(C.Structure):
    _fields_ = [
        ('B', C.c_int), ('T', C.c_int), ('d_model', C.c_int), ('n_heads', C.c_int),
        ('dff', C.c_int), ('n_layers', C.c_int), ('S_max', C.c_int), ('mode', C.c_int),
        ('x_len', C.c_int), ('ln_eps', C.c_float),
        ('layers', C.POINTER(VhLayer)), ('layers16', C.POINTER(VhLayer16)), ('ada', C.c_void_p),
        ('x_len_dev', C.c_void_p), ('kv_len', C.c_void_p), ('x', C.c_void_p), ('x_in', C.c_void_p),
        ('xn16', C.c_void_p), ('q16', C.c_void_p), ('attn16', C.c_void_p), ('hidden16', C.c_void_p),
    ]


# name → (restype, argtypes); must list every symbol include/valle_hip.h declares
SIGNATURES = {
    'vh_version': (C.c_int, []),
    'vh_last_error': (C.c_char_p, []),
    'vh_set_tuning': (C.c_int, [C.c_int, C.c_int]),
    'vh_embed_sum_pe': (C.c_int, [c_i64p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_int32), C.c_int, c_f32p, C.c_int, c_i32p, c_f32p, C.c_int64,
                                  C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, c_i32p, c_dropp, C.c_void_p]),
    'vh_dropout': (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int64, C.c_int, c_dropp, C.c_void_p]),
    'vh_dropout_mask': (C.c_int, [c_u8p, C.c_int64, C.c_int, c_dropp, C.c_void_p]),
    'vh_add_pe': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'vh_layernorm': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int,
                               C.c_float, C.c_void_p]),
    'vh_linear': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int,
                            C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                            C.c_float, C.c_void_p]),
    'vh_linear_ws_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'vh_linear_ws': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'vh_linear_qkv': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, c_f32p, c_i32p,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p,
                                c_f32p, C.c_float, C.c_void_p]),
    'vh_ln_fold': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int,
                             C.c_void_p]),
    'vh_linear_folded': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    'vh_linear_qkv_folded': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                       c_f32p, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_void_p]),
    'vh_ffn_decode_ws_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'vh_ffn_decode': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    'vh_attn_rows': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, c_u8p,
                               c_u8p, C.c_void_p]),
    'vh_attn_rows_bmask': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, c_u8p, C.c_int64, c_u8p, C.c_void_p]),
    'vh_attn_rows_lse': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, c_u8p,
                                   c_u8p, c_f32p, C.c_void_p]),
    'vh_attn_rows_bwd': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p,
                                   c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, c_i32p, c_i32p, c_u8p, c_u8p, C.c_void_p]),
    'vh_attn_rows_bwd_ws_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'vh_attn_rows_bwd_chunks': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'vh_attn_rows_bwd_ws': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p,
                                      c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, c_i32p, c_i32p, c_u8p, c_u8p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'vh_kv_to_bf16': (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'vh_linear_qkv_folded_kv16': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p,
                                            C.c_void_p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    'vh_attn_decode_kv16': (C.c_int, [c_f32p, C.c_int, C.c_void_p, C.c_void_p, c_f32p, C.c_int, c_i32p, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    'vh_attn_decode_ws_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'vh_attn_decode_shared_ws_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'vh_attn_decode_shared': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int,
                                        c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'vh_attn_decode': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_i32p, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'vh_greedy_step': (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_i64p, C.c_int64, c_i32p,
                                 c_i32p, c_f32p, c_f32p, c_i32p, c_i32p, c_f32p, C.c_int, C.c_int,
                                 C.c_void_p]),
    'vh_head_greedy_ws_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'vh_head_greedy': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_i64p, C.c_int64, c_i32p,
                                 c_i32p, c_f32p, c_f32p, c_i32p, c_i32p, c_f32p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    'vh_sample_step': (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64,
                                 c_i64p, C.c_int64, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p,
                                 c_f32p, C.c_int, C.c_int, C.c_void_p]),
    'vh_categorical_rows': (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_uint32,
                                      c_i64p, C.c_int64, c_f32p, C.c_void_p]),
    'vh_adamw_ws_bytes': (C.c_size_t, []),
    'vh_adamw_flat': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                c_f32p, c_i32p, c_i32p, c_i32p, C.c_void_p]),
    'vh_ar_decoder_create': (C.c_void_p, [C.POINTER(VhArDecoderDesc)]),
    'vh_ar_decoder_destroy': (None, [C.c_void_p]),
    'vh_ar_decoder_step': (C.c_int, [C.c_void_p, C.c_void_p]),
    'vh_ar_decoder_capture': (C.c_int, [C.c_void_p, C.c_void_p]),
    'vh_ar_decoder_replay': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'vh_ar_decoder_profile_attn': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p,
                                             C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'vh_layernorm_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                   c_f32p, c_f32p, c_f32p, c_f32p, c_dropp, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    'vh_gelu': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    'vh_softmax_rows': (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                  C.c_int, c_i32p, c_i32p, c_u8p, c_u8p, C.c_void_p]),
    'vh_softmax_bwd': (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    'vh_cross_entropy': (C.c_int, [c_f32p, C.c_int, C.c_int, c_i64p, c_f32p, c_f32p, C.c_int, C.c_int,
                                   c_i32p, C.c_void_p]),
    'vh_embed_bwd': (C.c_int, [c_i64p, C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_int, c_f32p, C.c_int,
                               C.c_int, C.c_int, C.c_int, c_i32p, c_dropp, C.c_void_p]),
    'vh_colsum': (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_void_p]),
    'vh_linear_ex': (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int,
                               c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_dropp, C.c_void_p, C.c_size_t, C.c_void_p]),
    'vh_linear_ex_ws_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'vh_transpose': (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_void_p]),
    'vh_transpose_many': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'vh_adaproj_fwd': (C.c_int, [C.c_void_p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, C.c_void_p]),
    'vh_adaproj_bwd': (C.c_int, [C.c_void_p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_void_p]),
    'vh_gemm_tn_ws_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'vh_gemm_tn': (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_void_p, C.c_size_t, C.c_void_p]),
    'vh_gemm_batched': (C.c_int, [c_f32p, C.c_int, C.c_int64, C.c_int64, C.c_int, c_f32p, C.c_int, C.c_int64,
                                  C.c_int64, C.c_int, c_f32p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'vh_transformer_forward': (C.c_int, [C.POINTER(VhForwardDesc), C.c_void_p]),
    'vh_h16_format': (C.c_int, []),
    'vh_to_bf16': (C.c_int, [c_f32p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    'vh_layernorm_bf16': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                    C.c_void_p]),
    'vh_linear_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'vh_linear_qkv_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, c_i32p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'vh_attn_rows_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, C.c_void_p]),
    'vh_transformer_forward_bf16': (C.c_int, [C.POINTER(VhForward16Desc), C.c_void_p]),
}

_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """dlopen the library and type every entry point.  Needs no GPU (the not-gpu tests use this to
    check that the ABI exports what the header declares)."""
    p = Path(path or LIB_PATH)
    if not p.exists():
        raise VhError(f'{p} not found: build it with `python __graft_entry__.py build` '
                      f'(make -C valle2_amd/csrc). There is no CPU fallback.')
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    # tuning knobs shape what a decoder captures: whoever keeps captured graphs across calls (valle_ar's decoder slots)
    # keys them on this count of vh_set_tuning calls
    raw_set = lib.vh_set_tuning

    def set_tuning(knob, value):
        global TUNING_EPOCH
        TUNING_EPOCH += 1
        return raw_set(knob, value)
    lib.vh_set_tuning = set_tuning
    return lib


TUNING_EPOCH = 0


def h16_dtype():
    """torch dtype of the library's 16-bit operand format ("h16": operands of the perf-mode kernels, the narrow K/V cache):
    float16 in the default build, bfloat16 in a -DVH_PERF_BF16 one (include/valle_hip.h, vh_h16_format).  Needs no GPU."""
    global _H16
    if _H16 is None:
        _H16 = torch.bfloat16 if load_library().vh_h16_format() else torch.float16
    return _H16


_H16 = None


def lib() -> C.CDLL:
    """The loaded library, for compute calls: requires a visible HIP device."""
    global _lib
    if _lib is None:
        if not torch.cuda.is_available():
            raise VhError('valle2_amd needs a HIP device (MI355X): torch.cuda.is_available() is '
                          'False and there is no CPU fallback. The CPU oracle lives in oracle/ and '
                          'is test infrastructure only.')
        _lib = load_library()
        # one HIP runtime per process: ours must be the copy torch loaded
        n = sum(1 for line in open('/proc/self/maps') if 'libamdhip64' in line and ' r-xp ' in line)
        if n > 1:
            raise VhError('two HIP runtimes are mapped in this process; import torch before '
                          'loading libvalle_hip.so')
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib().vh_last_error()
        raise VhError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')


def ptr(t: torch.Tensor | None) -> int | None:
    """Device pointer of a contiguous CUDA/HIP tensor (None passes a NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise VhError('tensor is not on a HIP device')
    if not t.is_contiguous():
        raise VhError('tensor must be contiguous')
    return t.data_ptr()


_PIN_RING: dict = {}            # device index -> [{'buf': pinned uint8 tensor, 'ev': event of its last copy or None}]
_PIN_RING_MAX = 16


def _pinned_slot(idx: int, nbytes: int):
    """A pinned staging buffer of >= nbytes whose last copy has left it.  A small ring per device: a step's handful of id /
    length tensors each take a slot, the copies of the step before have long finished when the slots come round again."""
    ring = _PIN_RING.setdefault(idx, [])
    for slot in ring:
        if slot['buf'].numel() >= nbytes and (slot['ev'] is None or slot['ev'].query()):
            return slot
    if len(ring) >= _PIN_RING_MAX:                      # every slot busy or too small: recycle the oldest one
        slot = ring.pop(0)
        if slot['ev'] is not None:
            slot['ev'].synchronize()
        if slot['buf'].numel() >= nbytes:
            ring.append(slot)
            return slot
    slot = {'buf': torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True), 'ev': None}
    ring.append(slot)
    return slot


def to_device_async(t: torch.Tensor, device, dtype=None) -> torch.Tensor:
    """Small host tensor (lengths, ids) -> device without stalling the host: through pinned memory with a
    stream-ordered copy.  A plain `.to(device)` of pageable memory waits until the stream has drained — in a training
    loop that is the whole previous step, so the host could never enqueue ahead of the GPU (5 ms per step at
    configs[3]).  A tensor that already lives on the device is only converted.

    The staging copy is numpy's, into a ring of pinned buffers kept here — not `Tensor.pin_memory()`: torch's CPU copy of
    a 70 k-element id tensor is a parallel region over every core `os.cpu_count()` reports, and on a box whose cgroup
    grants 16 of 256 cores the spinning thread pool gets the whole process throttled for tens of milliseconds, at random
    points of the step (round 4: NAR step 40-60 ms instead of 28 with host-resident batches)."""
    if t.is_cuda:
        return t.to(device=device, dtype=dtype or t.dtype)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    t = t.detach().contiguous()
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return torch.empty(t.shape, dtype=t.dtype, device=dev)
    slot = _pinned_slot(idx, nbytes)
    staged = slot['buf'][:nbytes].view(t.dtype).view(t.shape)
    staged.numpy()[...] = t.numpy()
    with torch.cuda.device(idx):
        out = staged.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    slot['ev'] = ev
    return out


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- device-side index errors ---------------------------------------------------------------------
# The gather / scatter / cross-entropy kernels range-check token ids and targets themselves (an id
# outside its table is read as row 0 or skipped — never an out-of-bounds access) and OR a code into a
# per-device int32 flag.  The flag is read where the host synchronises anyway (end of generate, the
# optimizer step, or explicitly) and becomes the IndexError the reference's nn.Embedding /
# F.cross_entropy would have raised.
DEVERR_EMBED_ID, DEVERR_TARGET = 1, 2
_err_flags: dict = {}


def err_flag(device) -> torch.Tensor:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _err_flags:
        with torch.inference_mode(False):       # a normal tensor even when first touched inside generate()
            _err_flags[idx] = torch.zeros(1, device=torch.device('cuda', idx), dtype=torch.int32)
    return _err_flags[idx]


def raise_device_errors(device=None, code=None):
    """Synchronises, then raises IndexError if a kernel saw an out-of-range id since the last call.  `code`: a flag
    value already read (asynchronously) from `device`'s flag — no synchronisation then."""
    flags = list(_err_flags.values()) if device is None else [err_flag(device)]
    for f in flags:
        c = int(f.item()) if code is None else int(code)
        if c:
            f.zero_()
            what = []
            if c & DEVERR_EMBED_ID:
                what.append('a token / codec id outside its embedding table (nn.Embedding raises IndexError; '
                            'e.g. EOS/BOS inside NAR `codes`, a text id >= vocab_size)')
            if c & DEVERR_TARGET:
                what.append('a cross-entropy target outside [0, V) (ignore_index is not supported: the collate '
                            'format pads with 0)')
            raise IndexError('index out of range on the HIP device: ' + '; '.join(what))
