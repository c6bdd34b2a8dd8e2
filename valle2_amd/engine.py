"""Device-side runtime of the path: KV cache, scratch buffers, the native forward composite and the
hipGraph-replayed AR decoder.  Python here only wires pointers; all arithmetic runs in
libvalle_hip.so (`_lib.lib()` raises without a HIP device — there is no CPU fallback).

Data layout in HBM (all fp32):
  residual stream x        (B*T, d) row-major, updated in place by the GEMM epilogues
  KV cache per layer       K (B, h, S_max, 64), V (B, h, S_max, 64): one (b,head) stream is
                           contiguous, so decode attention reads it as pure 16-B-per-lane bursts
                           and a new token's K/V row is appended in place (no torch.cat regrow,
                           valle/models/modules.py:151-157)
  weights                  the nn.Linear / nn.LayerNorm parameters exactly as stored ((N,K)
                           row-major): both GEMM operands are K-contiguous, no repacking
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import torch

from . import _lib, kernels
from ._lib import VhArDecoderDesc, VhForwardDesc, VhLayer, check, ptr, stream

HEAD_DIM = kernels.HEAD_DIM


class KVCache:
    """(L, 2, B, h, S_max, 64) in one allocation; `length` rows are valid for every batch row
    unless a per-row `cache_len` tensor says otherwise."""

    def __init__(self, n_layers, batch, n_heads, s_max, device, dtype=torch.float32):
        self.buf = torch.empty(n_layers, 2, batch, n_heads, s_max, HEAD_DIM, device=device, dtype=dtype)
        self.n_layers, self.batch, self.n_heads, self.s_max = n_layers, batch, n_heads, s_max

    @property
    def bf16(self):
        return self.buf.dtype == kernels.H16

    def narrowed(self, s_max):
        """A bf16 cache of `s_max` rows per (layer, K|V, row, head) stream holding this fp32 cache's rows (round to
        nearest even, vh_kv_to_bf16): the decode steps of perf mode read half the bytes."""
        if self.bf16 or s_max < self.s_max:
            raise _lib.VhError('KVCache.narrowed: an fp32 cache and s_max >= its length')
        out = KVCache(self.n_layers, self.batch, self.n_heads, s_max, self.buf.device, dtype=kernels.H16)
        check(_lib.lib().vh_kv_to_bf16(ptr(self.buf), ptr(out.buf), self.n_layers * 2 * self.batch * self.n_heads,
                                       self.s_max, self.s_max, s_max, stream()), 'vh_kv_to_bf16')
        return out

    def k(self, i):
        return self.buf[i, 0]

    def v(self, i):
        return self.buf[i, 1]


def _layer_params(layer):
    """The 11 parameter tensors of one EncoderLayer in VhLayer order."""
    at, ff = layer.self_attn, layer.ffn
    n1 = layer.norm1.norm if hasattr(layer.norm1, 'project_layer') else layer.norm1
    n2 = layer.norm2.norm if hasattr(layer.norm2, 'project_layer') else layer.norm2
    return (n1.weight, n1.bias, at.qkv.weight, at.out.weight, at.out.bias, n2.weight, n2.bias,
            ff.linear_1.weight, ff.linear_1.bias, ff.linear_2.weight, ff.linear_2.bias)


FOLD_LAYERNORM = os.environ.get('VALLE2_FOLD_LN', '1') != '0'   # LayerNorm folded into the QKV / linear_1
                                                                 # weights (vh_ln_fold); without it the decode step
                                                                 # normalises in the operand load and the FeedForward
                                                                 # runs as linear_1 + split-K linear_2 + reduce


_WEIGHTS_EPOCH = 0
# Derived weights (folded LayerNorm forms, AdaLN tables, bf16 copies) per Transformer module.  Kept HERE, not in the
# module's __dict__: the entries hold weak references and device tensors that torch.save(model) must not try to pickle, and
# an entry dies with its module.
_DERIVED = weakref.WeakKeyDictionary()


def _derived(module) -> dict:
    d = _DERIVED.get(module)
    if d is None:
        d = _DERIVED[module] = {}
    return d


def bump_weights_epoch():
    """Invalidate every derived weight form (folded LayerNorm weights, AdaLN tables, bf16 copies).  Called by code that
    rewrites parameters behind torch's back (the flat optimizer kernel) — and the hook for USER code that does: the caches
    are keyed on (tensor identity, `_version`, `data_ptr`), and a write through `p.data` (an EMA swap by
    `p.data.copy_(...)`) moves none of them.  After such a write call this (or `invalidate_derived`)."""
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


def invalidate_derived(module=None):
    """Drop the derived weights of `module` (a Transformer), or of every module when None."""
    if module is None:
        _DERIVED.clear()
        bump_weights_epoch()
    else:
        _DERIVED.pop(module, None)


def folded_layer_norms(transformer):
    """Per layer ((Wqkv∘γ1, c1, c2), (W1∘γ2, c1, c2)) for the decode step, or None when the shape is
    outside the folded kernels (d_model not in {128,256,512,1024}) or the norms are adaptive.  Cached
    on the module and rebuilt when any of the source parameters changed (optimizer step, load)."""
    layers = list(transformer.layers)
    d = transformer.hparams.d_model
    if (not FOLD_LAYERNORM or d not in (128, 256, 512, 1024) or transformer.hparams.dim_feedforward % 16
            or any(hasattr(l.norm1, 'project_layer') for l in layers)):
        return None
    srcs = [(l.norm1.weight, l.norm1.bias, l.self_attn.qkv.weight, l.norm2.weight, l.norm2.bias,
             l.ffn.linear_1.weight, l.ffn.linear_1.bias) for l in layers]
    key = (_WEIGHTS_EPOCH,) + tuple((t.data_ptr(), t._version) for ps in srcs for t in ps)
    cached = _derived(transformer).get('folded')
    if cached is not None and cached[0] == key:
        return cached[1]
    with torch.no_grad():
        folded = [(kernels.ln_fold(wq.detach(), g1.detach(), b1.detach()),
                   kernels.ln_fold(w1.detach(), g2.detach(), b2.detach(), bias1.detach()))
                  for g1, b1, wq, g2, b2, w1, bias1 in srcs]
    _derived(transformer)['folded'] = (key, folded)
    return folded


def layer_table(transformer, cache: KVCache, folded=None):
    """ctypes array of VhLayer for `transformer.layers` bound to `cache`."""
    layers = list(transformer.layers)
    arr = (VhLayer * len(layers))()
    for i, layer in enumerate(layers):
        ps = _layer_params(layer)
        for (name, _), t in zip(VhLayer._fields_[:11], ps):
            if t.dtype != torch.float32:
                raise _lib.VhError(f'parameter {name} of layer {i} is {t.dtype}; the path is fp32')
            setattr(arr[i], name, ptr(t.detach()))
        if cache is not None:
            arr[i].kcache = ptr(cache.k(i))
            arr[i].vcache = ptr(cache.v(i))
        if folded is not None:
            (arr[i].wqkv_f, arr[i].qkv_c1, arr[i].qkv_c2), (arr[i].w1_f, arr[i].w1_c1, arr[i].w1_c2) = (
                tuple(ptr(t) for t in folded[i][0]), tuple(ptr(t) for t in folded[i][1]))
    return arr


def adaln_table(transformer, embedding):
    """(L, 2, 2, d) [layer][norm1|norm2][scale|shift] = project_layer(embedding) for every AdaptiveLayerNorm of the
    stack (valle/models/modules.py:94-98) in ONE launch (`vh_adaproj_fwd`, the training path's kernel: a wave per 8
    output rows of one of the 2 L Linears), instead of 2 L single-row GEMV launches from Python.

    The table of a stage depends on the stage embedding and the projection weights only, so it is kept per embedding
    TENSOR OBJECT (a weak reference: a parameter such as stage_embs[n].weight lives as long as the model; an ad-hoc
    tensor's entry dies with it, and a recycled address can never alias a live entry) and reused while neither that
    tensor's version, nor any projection parameter's, nor the weights epoch (flat optimizer steps) has moved — the 7
    stages of ValleNAR.generate_batch build 7 tables once, not 7 per call."""
    import numpy as np
    layers = list(transformer.layers)
    d = embedding.shape[-1]
    projs = [(n.project_layer.weight, n.project_layer.bias) for l in layers for n in (l.norm1, l.norm2)]
    n = len(projs)
    state = _derived(transformer).setdefault('ada', {'items': None, 'tables': []})
    pkey = (_WEIGHTS_EPOCH,) + tuple(x for w, b in projs for x in (w.data_ptr(), w._version, b.data_ptr(), b._version))
    for ref, ver, key, table in state['tables']:
        if ref() is embedding and ver == embedding._version and key == pkey and table.device == embedding.device:
            return table
    state['tables'] = [e for e in state['tables'] if e[0]() is not None and e[0]() is not embedding][-15:]
    if any(tuple(w.shape) != (2 * d, d) or not w.is_contiguous() or w.dtype != torch.float32 for w, _ in projs):
        raise _lib.VhError('adaln_table: project_layer weights must be contiguous fp32 (2 d, d)')
    ptrs = tuple(x for w, b in projs for x in (w.data_ptr(), b.data_ptr()))
    if state['items'] is None or state['items'][0] != ptrs or state['items'][1].device != embedding.device:
        rec = np.zeros(n, dtype=[('w', 'u8'), ('b', 'u8'), ('dw', 'u8'), ('db', 'u8')])
        for i, (w, b) in enumerate(projs):
            rec[i] = (w.data_ptr(), b.data_ptr(), 0, 0)
        state['items'] = (ptrs, _lib.to_device_async(torch.from_numpy(rec.view(np.uint8).copy()), embedding.device))
    emb = embedding.detach().reshape(-1).contiguous()
    with torch.inference_mode(False):            # a normal tensor: the cached table outlives an inference_mode() caller
        out = torch.empty(n, 2 * d, device=emb.device, dtype=torch.float32)
    check(_lib.lib().vh_adaproj_fwd(ptr(state['items'][1]), n, ptr(emb), ptr(out), 2 * d, d, stream()), 'vh_adaproj_fwd')
    table = out.view(len(layers), 2, 2, d)
    try:
        state['tables'].append((weakref.ref(embedding), embedding._version, pkey, table))
    except TypeError:
        pass
    return table


def bf16_weights(transformer):
    """Per layer (Wqkv, Wo, W1, W2) as bf16 (vh_to_bf16, round to nearest even) for the perf-mode forward; built once per
    weight set (same cache rule as `folded_layer_norms`)."""
    layers = list(transformer.layers)
    srcs = [(l.self_attn.qkv.weight, l.self_attn.out.weight, l.ffn.linear_1.weight, l.ffn.linear_2.weight) for l in layers]
    key = (_WEIGHTS_EPOCH,) + tuple((t.data_ptr(), t._version) for ps in srcs for t in ps)
    cached = _derived(transformer).get('bf16')
    if cached is not None and cached[0] == key:
        return cached[1]
    with torch.no_grad(), torch.inference_mode(False):
        out = [tuple(kernels.to_bf16(t.detach()) for t in ps) for ps in srcs]
    _derived(transformer)['bf16'] = (key, out)
    return out


def head16(module, weight):
    """The h16 copy of a head matrix (V, d) for the perf-mode NAR stage (valle_nar.py: the stage's projection over every target
    frame is a large-M product like the stack's; fp32 it was 0.5 of a 10 ms stage) — None when the 16-bit tile GEMM does not
    take the shape.  Built once per weight version, cached on the owning module like the other derived forms."""
    V, d = weight.shape
    if V % 128 or d % 64:
        return None
    cache = _derived(module).setdefault('head16', {})
    key = (_WEIGHTS_EPOCH, weight.data_ptr(), weight._version)
    hit = cache.get(id(weight))
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad(), torch.inference_mode(False):
        w16 = kernels.to_bf16(weight.detach())
    cache[id(weight)] = (key, w16)
    return w16


# perf mode's decode step over h16 copies of its four matrices (+ the head): built in round 6, measured SLOWER (471.1 vs 463.9 us per
# step at 32 rows x 12L/512d, profiles/r6_ab_decode_w16.log — the chain's launches are round trips, not byte streams, and an 8-byte
# fragment per lane halves the useful part of every line it touches) — so it is opt-in: VALLE2_DECODE_W16=1
DECODE_W16 = os.environ.get('VALLE2_DECODE_W16', '0') == '1'


def decode_weights16(transformer, folded):
    """Per layer (Wqkv∘γ1, Wo, W1∘γ2, W2) as h16 for the PERF-MODE decode step (the second half of SURVEY section 7's perf mode:
    16-bit storage of what a step streams, fp32 accumulate — the folded matrices are narrowed AFTER the fold, so the
    epilogue's c1 / c2 stay the fp32 sums of the fold).  Built once per weight set, beside the folded weights."""
    layers = list(transformer.layers)
    key = (_WEIGHTS_EPOCH, id(folded)) + tuple((t.data_ptr(), t._version) for l in layers
                                               for t in (l.self_attn.out.weight, l.ffn.linear_2.weight))
    cached = _derived(transformer).get('decode16')
    if cached is not None and cached[0] == key:
        return cached[1]
    with torch.no_grad(), torch.inference_mode(False):
        out = [(kernels.to_bf16(folded[i][0][0]), kernels.to_bf16(l.self_attn.out.weight.detach()),
                kernels.to_bf16(folded[i][1][0]), kernels.to_bf16(l.ffn.linear_2.weight.detach())) for i, l in enumerate(layers)]
    _derived(transformer)['decode16'] = (key, out, folded)       # (the folded list is kept alive: its id is part of the key)
    return out


class ForwardScratch16:
    """bf16 activations of the perf-mode forward: xn, q, attn (rows, d) and hidden (rows, dff)."""

    def __init__(self, rows, d, dff, device):
        bf = dict(device=device, dtype=kernels.H16)
        self.xn = torch.empty(rows, d, **bf)
        self.q = torch.empty(rows, d, **bf)
        self.attn = torch.empty(rows, d, **bf)
        self.hidden = torch.empty(rows, dff, **bf)


def perf_forward_supported(cfg):
    """Shapes the bf16 tile kernels serve: d_model = n_heads x 64, d_model and dim_feedforward multiples of 128."""
    return cfg.d_model == cfg.n_heads * HEAD_DIM and cfg.d_model % 128 == 0 and cfg.dim_feedforward % 128 == 0


def transformer_forward_bf16(transformer, x, cache: 'KVCache', *, mode, x_len=0, x_len_dev=None, kv_len=None, embedding=None,
                             scratch=None, x_in=None):
    """`transformer_forward` in PERF MODE (SECONDARY): every product of the stack on the bf16 matrix
    cores (bf16 operands, fp32 accumulators), residual stream x fp32 in place, K/V written to a bf16 cache.  Analytic
    masks only.  Teacher-forced logits agree with the reference to 5e-2, not to the parity path's 2e-4."""
    cfg = transformer.hparams
    B, T, d = x.shape
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise _lib.VhError('x must be contiguous fp32')
    if not perf_forward_supported(cfg):
        raise _lib.VhError(f'perf mode needs d_model = n_heads x 64 and d_model, dim_feedforward multiples of 128 '
                           f'(got {cfg.d_model}, {cfg.n_heads}, {cfg.dim_feedforward})')
    if cache is None or not cache.bf16 or cache.batch != B or cache.s_max < T or cache.n_layers != cfg.num_layers:
        raise _lib.VhError('perf mode needs a bf16 KV cache that fits this forward')
    scratch = scratch or ForwardScratch16(B * T, d, cfg.dim_feedforward, x.device)
    w16 = bf16_weights(transformer)
    table = layer_table(transformer, None)
    t16 = (_lib.VhLayer16 * cfg.num_layers)()
    for i, (wq, wo, w1, w2) in enumerate(w16):
        t16[i].wqkv, t16[i].wo, t16[i].w1, t16[i].w2 = wq.data_ptr(), wo.data_ptr(), w1.data_ptr(), w2.data_ptr()
        t16[i].kcache16, t16[i].vcache16 = cache.k(i).data_ptr(), cache.v(i).data_ptr()
    ada = None
    if cfg.norm != 'LayerNorm':
        if embedding is None:
            raise TypeError('AdaptiveLayerNorm needs `embedding` (reference: Linear(None) TypeError)')
        ada = adaln_table(transformer, embedding)
    desc = _lib.VhForward16Desc(
        B=B, T=T, d_model=d, n_heads=cfg.n_heads, dff=cfg.dim_feedforward, n_layers=cfg.num_layers, S_max=cache.s_max,
        mode=mode, x_len=int(x_len), ln_eps=1e-5, layers=table, layers16=t16, ada=ptr(ada), x_len_dev=ptr(x_len_dev),
        kv_len=ptr(kv_len), x=ptr(x), x_in=ptr(x_in), xn16=scratch.xn.data_ptr(), q16=scratch.q.data_ptr(),
        attn16=scratch.attn.data_ptr(), hidden16=scratch.hidden.data_ptr())
    check(_lib.lib().vh_transformer_forward_bf16(C.byref(desc), stream()), 'vh_transformer_forward_bf16')
    return x


class ForwardScratch:
    def __init__(self, rows, d, dff, device):
        self.xn = torch.empty(rows, d, device=device, dtype=torch.float32)
        self.q = torch.empty(rows, d, device=device, dtype=torch.float32)
        self.attn = torch.empty(rows, d, device=device, dtype=torch.float32)
        self.hidden = torch.empty(rows, dff, device=device, dtype=torch.float32)
        L = _lib.lib()
        self.ws_bytes = max(L.vh_linear_ws_bytes(rows, d, d), L.vh_linear_ws_bytes(rows, d, dff),
                            L.vh_linear_ws_bytes(rows, dff, d))
        self.ws = torch.zeros(self.ws_bytes // 4, device=device, dtype=torch.float32) if self.ws_bytes else None


def transformer_forward(transformer, x, cache: KVCache, *, mode, x_len=0, x_len_dev=None,
                        kv_len=None, mask=None, pad=None, embedding=None, scratch=None, x_in=None):
    """Run all layers over x (B, T, d) IN PLACE (valle/models/modules.py:341-349), writing each
    layer's K/V to `cache` rows 0..T-1.  Returns x.  With `x_in` (same shape, contiguous) the input rows are
    read from there and left untouched, x is output only."""
    cfg = transformer.hparams
    B, T, d = x.shape
    if not x.is_contiguous():
        raise _lib.VhError('x must be contiguous')
    if x_in is not None and (tuple(x_in.shape) != tuple(x.shape) or x_in.dtype != torch.float32):
        raise _lib.VhError('x_in must match x')
    if d != cfg.n_heads * HEAD_DIM:                          # a head width the native composite is not built for
        return _transformer_forward_generic(transformer, x, mode=mode, x_len=x_len, x_len_dev=x_len_dev, kv_len=kv_len,
                                            mask=mask, pad=pad, embedding=embedding, x_in=x_in)
    if cache is None or cache.batch != B or cache.s_max < T or cache.n_layers != cfg.num_layers:
        raise _lib.VhError('KV cache does not fit this forward')
    scratch = scratch or ForwardScratch(B * T, d, cfg.dim_feedforward, x.device)
    table = layer_table(transformer, cache)
    ada = None
    if cfg.norm != 'LayerNorm':
        if embedding is None:
            raise TypeError('AdaptiveLayerNorm needs `embedding` (reference: Linear(None) TypeError)')
        ada = adaln_table(transformer, embedding)
    desc = VhForwardDesc(
        B=B, T=T, d_model=d, n_heads=cfg.n_heads, dff=cfg.dim_feedforward, n_layers=cfg.num_layers,
        S_max=cache.s_max, mode=mode, x_len=int(x_len), ln_eps=1e-5, layers=table, ada=ptr(ada),
        x_len_dev=ptr(x_len_dev), kv_len=ptr(kv_len), mask=ptr(mask), pad=ptr(pad),
        x=ptr(x), xn=ptr(scratch.xn), q=ptr(scratch.q), attn=ptr(scratch.attn),
        hidden=ptr(scratch.hidden), gemm_ws=ptr(scratch.ws), gemm_ws_bytes=scratch.ws_bytes, x_in=ptr(x_in))
    check(_lib.lib().vh_transformer_forward(C.byref(desc), stream()), 'vh_transformer_forward')
    return x


def _transformer_forward_generic(transformer, x, *, mode, x_len, x_len_dev, kv_len, mask, pad, embedding, x_in):
    """`transformer_forward` for a head width other than 64 (valle/models/modules.py:109-111 allows any divisor of d_model;
    every configuration of the path and of the reference's tests has 64, which is what the flash kernels and the native
    composite are built for).  The same pre-norm stack layer by layer on the general kernels — LayerNorm, the tile / skinny
    GEMMs, materialised attention (`kernels.attn_generic`) — correct, not tuned, and without a KV cache: a caller that
    decodes recomputes (ValleAR.generate_batch does)."""
    cfg = transformer.hparams
    B, T, d = x.shape
    h = cfg.n_heads
    hd = d // h
    if hd % 4:
        raise _lib.VhError(f'head_dim {hd}: the general attention path needs a multiple of 4')
    ada = None
    if cfg.norm != 'LayerNorm':
        if embedding is None:
            raise TypeError('AdaptiveLayerNorm needs `embedding` (reference: Linear(None) TypeError)')
        ada = adaln_table(transformer, embedding)
    cur = (x_in if x_in is not None else x).reshape(B * T, d)
    spec = dict(mode=mode, x_len=int(x_len), x_len_dev=x_len_dev, kv_len=kv_len, mask=mask, pad=pad)
    f32 = dict(device=x.device, dtype=torch.float32)
    for i, layer in enumerate(transformer.layers):
        g1, b1, wqkv, wo, bo, g2, b2, w1, bb1, w2, bb2 = (t.detach() for t in _layer_params(layer))
        sc1, sh1, sc2, sh2 = (ada[i, 0, 0], ada[i, 0, 1], ada[i, 1, 0], ada[i, 1, 1]) if ada is not None else (None,) * 4
        xn = kernels.layernorm(cur, g1, b1, ada_scale=sc1, ada_shift=sh1, eps=layer.norm1.eps)
        qkv = kernels.linear(xn, wqkv, out=torch.empty(B * T, 3 * d, **f32))
        q, k, v = (qkv.view(B, T, 3, h, hd)[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        attn = torch.empty(B * T, d, **f32)
        kernels.attn_generic(q, k, v, attn.view(B, T, h, hd).permute(0, 2, 1, 3), hd ** -0.5, **spec)
        nxt = kernels.linear(attn, wo, bo, residual=cur, out=torch.empty(B * T, d, **f32))
        xn = kernels.layernorm(nxt, g2, b2, ada_scale=sc2, ada_shift=sh2, eps=layer.norm2.eps)
        hid = kernels.linear(xn, w1, bb1, act=kernels.ACT_GELU, out=torch.empty(B * T, w1.shape[0], **f32))
        cur = kernels.linear(hid, w2, bb2, residual=nxt, out=torch.empty(B * T, d, **f32))
    x.reshape(B * T, d).copy_(cur)
    return x


SHARED_MAX_RECORDS = 256          # vh_attn_decode_shared: prefix blocks of 32 keys + suffix splits one merge launch serves


def shared_n_split(batch: int, n_heads: int) -> int:
    """Key splits of the beams' own rows under a shared prompt (two workgroups per CU keep twice the loads in flight:
    profiles/r5_ab_shared_prompt.log).  VALLE2_SHARED_SPLIT: the A/B knob."""
    return int(os.environ.get('VALLE2_SHARED_SPLIT') or min(16, -(-512 // (batch * n_heads))))


def shared_prompt_fits(batch: int, n_heads: int, prefix_len: int) -> bool:
    """Whether vh_attn_decode_shared takes a prompt of prefix_len keys for `batch` beams (its merge serves at most 256
    records per (row, head): ceil(prefix_len / 32) prefix blocks + the suffix splits) — 4 beams x 8 heads: up to 7680 keys."""
    return (prefix_len + 31) // 32 + shared_n_split(batch, n_heads) <= SHARED_MAX_RECORDS


def pick_n_split(rows_x_heads: int) -> int:
    """Key-range splits of decode attention so that the grid covers the 256 CUs."""
    if rows_x_heads >= 256:
        return 1
    return max(1, min(16, -(-256 // rows_x_heads)))


_CAPTURE_STREAMS = {}
_CAPTURE_LOCK = threading.Lock()      # one capture at a time on the shared capture stream (host threads with their own streams)


def _capture_stream(device_index):
    """ONE side stream per device for every graph capture of the process (capture enqueues nothing, so decoders can
    share it): a fresh torch.cuda.Stream() per decoder walks through torch's stream pool, and every new stream that
    is used becomes another hardware queue — which also decides where a later stream of another priority lands
    (pipelined attention: the placement of its queue)."""
    st = _CAPTURE_STREAMS.get(device_index)
    if st is None:
        st = _CAPTURE_STREAMS[device_index] = torch.cuda.Stream(device=device_index)
    return st


class StepSampler:
    """Head + sampling step on the last hidden row of a full forward (valle_ar.py:158-171: logits, top-k / top-p / greedy
    draw, EOS bookkeeping, token append) with the decode state on the device — what ArDecoder.sample_from does, without
    the native decoder behind it: the recompute path of ValleAR.generate_batch (use_kv_cache=False, or a head width the
    decoder is not built for) samples through this."""

    def __init__(self, model, batch, codes, cache_len, audio_pos, pos_base, seed=0):
        cfg = model.config
        dev = codes.device
        self.B, self.V, self.d = batch, cfg.num_audio_tokens + 1, cfg.d_model
        self.eos = cfg.num_audio_tokens
        self.logits = torch.zeros(batch, (self.V + 3) // 4 * 4, device=dev, dtype=torch.float32)
        self.x = torch.empty(batch, cfg.d_model, device=dev, dtype=torch.float32)       # the appended token's embedding (unused here)
        self.eos_count = torch.zeros(codes.shape[1] + 1, device=dev, dtype=torch.int32)
        self.sum_logprobs = torch.zeros(batch, device=dev, dtype=torch.float32)
        self.sampling = (int(cfg.top_k), float(cfg.tok_p), float(cfg.temperature), int(seed))
        self.codes, self.cache_len, self.audio_pos, self.pos_base = codes, cache_len, audio_pos, pos_base
        self._keep = (model.proj.weight.detach(), model.audio_emb.weight.detach(), model.audio_position_emb.pe)
        self.n_split, self.ffn_ws, self.kv_bf16, self.head_ws = 0, None, False, None

    def sample_from(self, hidden_last):
        m = self._keep
        kernels.linear(hidden_last, m[0], out=self.logits[:, : self.V])
        if self.sampling[0] == 1:
            kernels.greedy_step(self.logits, self.V, self.eos, self.codes, self.eos_count, m[1], m[2],
                                self.audio_pos, self.cache_len, self.x, pos_base=self.pos_base)
        else:
            top_k, top_p, temp, seed = self.sampling
            kernels.sample_step(self.logits, self.V, self.eos, top_k, top_p, temp, seed, self.codes,
                                self.eos_count, self.sum_logprobs, m[1], m[2], self.audio_pos,
                                self.cache_len, self.x, pos_base=self.pos_base)

    def capture(self):
        pass

    def close(self):
        pass


class ArDecoder:
    """The AR decode loop of valle/models/valle_ar.py:141-171 for B independent rows, one token
    per row per step, the whole step enqueued natively and replayed as a hipGraph."""

    def __init__(self, model, batch, s_max, codes, cache: KVCache, cache_len, audio_pos, pos_base,
                 n_split=None, use_graph=True, seed=0, prefix: KVCache | None = None, prefix_len=0):
        """prefix / prefix_len: SHARED-PROMPT decoding (the beams of ONE utterance, valle_ar.py:135-138): `prefix` is a
        one-row cache holding the prompt's K/V (its first prefix_len rows), `cache` then holds only the generated rows of
        every beam (s_max = its capacity) and cache_len counts those."""
        cfg = model.config
        dev = cache.buf.device
        d, dff, V = cfg.d_model, cfg.dim_feedforward, cfg.num_audio_tokens + 1
        self.B, self.V, self.d = batch, V, d
        self.ldl = (V + 3) // 4 * 4
        self.n_split = n_split or pick_n_split(batch * cfg.n_heads)
        if prefix is not None and n_split is None:
            # shared prompt: the beams' own rows are short streams — two workgroups per CU keep twice the loads in flight
            # (32 beams x 8 heads: 436.8 us per step with one split, 420.9 with two, 426.9 with four)
            self.n_split = shared_n_split(batch, cfg.n_heads)
        f32 = dict(device=dev, dtype=torch.float32)
        self.x = torch.empty(batch, d, **f32)
        self.q = torch.empty(batch, d, **f32)
        self.attn = torch.empty(batch, d, **f32)
        self.hidden = torch.empty(batch, dff, **f32)
        self.logits = torch.zeros(batch, self.ldl, **f32)
        self.prefix, self.prefix_len = prefix, int(prefix_len)
        if prefix is not None:
            if prefix.bf16 or cache.bf16 or prefix.batch != 1 or prefix.n_layers != cfg.num_layers or not 0 < prefix_len <= prefix.s_max:
                raise _lib.VhError('shared-prompt decoding: a one-row fp32 prefix cache holding prefix_len rows')
            self.partial = kernels.attn_decode_shared_ws(batch, cfg.n_heads, self.prefix_len, self.n_split, dev)
        else:
            self.partial = kernels.attn_decode_ws(batch, cfg.n_heads, self.n_split, dev)
        ws_bytes = _lib.lib().vh_linear_ws_bytes(batch, d, dff)
        self.gemm_ws = torch.empty(max(ws_bytes, 16) // 4, **f32) if ws_bytes else None
        self.eos_count = torch.zeros(codes.shape[1] + 1, device=dev, dtype=torch.int32)
        self.sum_logprobs = torch.zeros(batch, **f32)
        self.sampling = (int(cfg.top_k), float(cfg.tok_p), float(cfg.temperature), int(seed))
        # the sampling seed lives in device memory (desc.seed = 0 + *seed_dev): a captured graph would freeze a by-value
        # seed, and this decoder may serve many generate() calls (`reset`)
        self.seed_dev = torch.tensor([int(seed)], dtype=torch.int64).to(dev) if self.sampling[0] != 1 else None
        self.codes, self.cache, self.cache_len, self.audio_pos = codes, cache, cache_len, audio_pos
        self.pos_base = pos_base
        self._folded = folded_layer_norms(model.transformer)   # kept alive: the table holds raw pointers
        self.kv_bf16 = cache.bf16
        if self.kv_bf16 and (self._folded is None or self.n_split != 1):
            raise _lib.VhError('perf mode (bf16 K/V cache) needs the folded LayerNorm weights and rows x heads >= 256 '
                               f'(one (row, head) per workgroup: n_split = {self.n_split})')
        # FeedForward of a layer as one launch split over dim_feedforward + the slab reduce (vh_ffn_decode)
        ffn_bytes = _lib.lib().vh_ffn_decode_ws_bytes(batch, d, dff) if self._folded is not None else 0
        self.ffn_ws = torch.empty(ffn_bytes // 4, **f32) if ffn_bytes else None
        # opt-in (VALLE2_HEAD_FUSED=1): head + greedy step as one launch (vh_head_greedy; DESIGN.md 3.20 has the A/B)
        self.head_ws = None
        if os.environ.get('VALLE2_HEAD_FUSED') == '1' and self.sampling[0] == 1 and batch <= 64 and d in (128, 256, 512, 1024):
            self.head_ws = kernels.head_greedy_ws(batch, V, dev)
        self._table = layer_table(model.transformer, cache, self._folded)
        # perf mode, second half: the step's four matrices (and the head) as h16 — half the weight bytes per launch
        self._w16 = self._proj16 = None
        if self.kv_bf16 and DECODE_W16 and self.ffn_ws is not None and d <= 1024 and cfg.dim_feedforward % 16 == 0:
            self._w16 = decode_weights16(model.transformer, self._folded)
            for i, (wq, wo, w1, w2) in enumerate(self._w16):
                self._table[i].wqkv_f16, self._table[i].wo16 = ptr(wq), ptr(wo)
                self._table[i].w1_f16, self._table[i].w2_16 = ptr(w1), ptr(w2)
            with torch.inference_mode(False):
                self._proj16 = kernels.to_bf16(model.proj.weight.detach()) if d % 8 == 0 else None
        self.w16 = self._w16 is not None
        if prefix is not None:
            for i in range(cfg.num_layers):
                self._table[i].kprefix, self._table[i].vprefix = ptr(prefix.k(i)), ptr(prefix.v(i))
        self._keep = (model.proj.weight.detach(), model.audio_emb.weight.detach(),
                      model.audio_position_emb.pe)
        desc = VhArDecoderDesc(
            B=batch, d_model=d, n_heads=cfg.n_heads, dff=dff, n_layers=cfg.num_layers, S_max=s_max,
            V=V, eos=cfg.num_audio_tokens, n_split=self.n_split, ln_eps=1e-5, layers=self._table,
            proj_w=ptr(self._keep[0]), audio_emb=ptr(self._keep[1]), audio_pe=ptr(self._keep[2]),
            x=ptr(self.x), q=ptr(self.q), attn=ptr(self.attn), hidden=ptr(self.hidden),
            logits=ptr(self.logits), attn_partial=ptr(self.partial), gemm_ws=ptr(self.gemm_ws),
            gemm_ws_bytes=ws_bytes, cache_len=ptr(cache_len),
            audio_pos=ptr(audio_pos), eos_count=ptr(self.eos_count), pos_base=ptr(pos_base),
            codes=ptr(codes), codes_stride=codes.stride(0), top_k=self.sampling[0], top_p=self.sampling[1],
            temperature=self.sampling[2], seed=0 if self.seed_dev is not None else self.sampling[3] & (2 ** 64 - 1),
            seed_dev=ptr(self.seed_dev), proj_w16=ptr(self._proj16),
            sum_logprobs=ptr(self.sum_logprobs), ffn_ws=ptr(self.ffn_ws), ffn_ws_bytes=ffn_bytes,
            kv_bf16=int(self.kv_bf16), prefix_len=self.prefix_len if prefix is not None else 0,
            prefix_S=prefix.s_max if prefix is not None else 0,
            attn_partial_bytes=self.partial.numel() * 4 if self.partial is not None else 0,
            head_ws=ptr(self.head_ws), head_ws_bytes=self.head_ws.numel() * 4 if self.head_ws is not None else 0)
        self._desc = desc
        self._h = _lib.lib().vh_ar_decoder_create(C.byref(desc))
        if not self._h:
            msg = _lib.lib().vh_last_error()
            raise _lib.VhError(f'vh_ar_decoder_create: {msg.decode() if msg else "failed"}')
        self._captured = False
        self.use_graph = use_graph

    def close(self):
        if getattr(self, '_h', None) and _lib is not None:     # (a decoder kept in a slot may outlive the module at interpreter exit)
            _lib.lib().vh_ar_decoder_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self, seed=0):
        """Ready for another generate() over the SAME buffers (codes, caches, cache_len, audio_pos, pos_base are the
        caller's to refill): counters and scores cleared, the new call's seed written where the captured steps read it."""
        self.eos_count.zero_()
        self.sum_logprobs.zero_()
        self.sampling = self.sampling[:3] + (int(seed),)
        if self.seed_dev is not None:
            self.seed_dev.copy_(torch.tensor([int(seed)], dtype=torch.int64), non_blocking=True)

    def sample_from(self, hidden_last):
        """Head + greedy step on the last hidden row of a prefill (the tail of step 0)."""
        m = self._keep
        kernels.linear(hidden_last, m[0], out=self.logits[:, : self.V])
        if self.sampling[0] == 1:
            kernels.greedy_step(self.logits, self.V, self._desc.eos, self.codes, self.eos_count, m[1], m[2],
                                self.audio_pos, self.cache_len, self.x, pos_base=self.pos_base)
        else:
            top_k, top_p, temp, seed = self.sampling
            kernels.sample_step(self.logits, self.V, self._desc.eos, top_k, top_p, temp, seed, self.codes,
                                self.eos_count, self.sum_logprobs, m[1], m[2], self.audio_pos,
                                self.cache_len, self.x, pos_base=self.pos_base)

    def capture(self):
        """Record the step graphs now (host work only: capture enqueues nothing).  generate_batch calls it right after
        the prompt pass is enqueued, so the ~1.7 ms of capture + instantiate run on the host while the GPU is busy."""
        if self.use_graph and not self._captured:
            with _CAPTURE_LOCK:
                cap = _capture_stream(torch.cuda.current_device())
                check(_lib.lib().vh_ar_decoder_capture(self._h, cap.cuda_stream), 'vh_ar_decoder_capture')
            self._captured = True

    def run(self, n_steps):
        """Enqueue n_steps decode steps on the current stream (graph replay when enabled)."""
        if n_steps <= 0:
            return
        s = stream()
        L = _lib.lib()
        if self.use_graph:
            self.capture()
            check(L.vh_ar_decoder_replay(self._h, n_steps, s), 'vh_ar_decoder_replay')
        else:
            for _ in range(n_steps):
                check(L.vh_ar_decoder_step(self._h, s), 'vh_ar_decoder_step')

    def profile_attn(self, n_steps):
        """(bracket_ms, floor_ms, kernel_ms) over n_steps eager steps: mean event-to-event time of marker events
        around the decode-attention launches, the same bracket with nothing inside, and the mean time between the
        start/stop events attached to the kernel dispatch itself (advances the decode state by n_steps)."""
        ms, floor, kern = C.c_float(0), C.c_float(0), C.c_float(0)
        check(_lib.lib().vh_ar_decoder_profile_attn(self._h, n_steps, stream(), C.byref(ms), C.byref(floor),
                                                    C.byref(kern)), 'vh_ar_decoder_profile_attn')
        return ms.value, floor.value, kern.value
