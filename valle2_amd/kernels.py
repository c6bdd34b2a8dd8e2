"""Tensor-level wrappers of the C-ABI primitives (include/valle_hip.h).

Each function checks shapes on the host (a kernel that faults can reset every GPU of the box),
passes raw device pointers + the current torch HIP stream, and raises on a non-zero status.
Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, stream

MASK_FULL, MASK_PREFIX, MASK_EXPLICIT = 0, 1, 2
H16 = _lib.h16_dtype()      # the 16-bit operand format of perf mode (float16 by default; the *_bf16 names are round 5's)
ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_GELU_D, ACT_MUL = 0, 1, 2, 3, 4
HEAD_DIM = 64


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise _lib.VhError(f'{name} must be float32, got {t.dtype}')
    return t


def _drop_ref(drop):
    return None if drop is None else C.byref(drop)


def embed_sum_pe(ids, tables, pe, pos0, out, out_t0=0, lens=None, row_pos0=None, row_t0=None, max_pos=None, drop=None):
    """out[b, out_t0+t] = sum_j tables[j][ids[b,t,j]] + pe[pos0+t].  ids (B,T) or (B,T,J) int64
    (any strides); tables: list of (vocab,d); pe (P,1,d)|(P,d)|None; out (B,T_out,d).
    lens / row_pos0 / row_t0: device int32 (B) — valid ids per row and per-row overrides of pos0 / out_t0 (a ragged
    batch in one launch); with them the caller vouches, through `max_pos` (largest position + 1 any row reaches) and
    its own layout, that positions stay inside the table and rows inside `out`.
    drop: a dropout.spec — the dropout after the position add (modules.py:80), field row = the row's index in `out`."""
    if ids.dtype != torch.int64:
        raise _lib.VhError('ids must be int64')
    if ids.dim() == 2:
        ids = ids.unsqueeze(-1)
    B, T, J = ids.shape
    n = len(tables)
    if n > J and J != 1:
        raise _lib.VhError(f'{n} tables but ids has {J} codebooks')
    d = tables[0].shape[1]
    if out.dim() != 3 or out.shape[0] != B or out.shape[2] != d or (row_t0 is None and out_t0 + T > out.shape[1]):
        raise _lib.VhError(f'out shape {tuple(out.shape)} does not fit B={B} T={T} d={d} t0={out_t0}')
    if row_pos0 is not None or row_t0 is not None:
        for t_ in (row_pos0, row_t0, lens):
            if t_ is not None and (t_.dtype != torch.int32 or t_.numel() != B or not t_.is_cuda):
                raise _lib.VhError('embed_sum_pe: lens / row_pos0 / row_t0 must be device int32 tensors of B entries')
        if lens is None or max_pos is None:
            raise _lib.VhError('embed_sum_pe: per-row offsets need `lens` and `max_pos`')
        if pe is not None and max_pos > pe.shape[0]:
            raise _lib.VhError(f'position {max_pos} exceeds the table ({pe.shape[0]})')
    elif pe is not None and pos0 + T > pe.shape[0]:
        raise _lib.VhError(f'position {pos0 + T} exceeds the table ({pe.shape[0]})')
    if not ids.is_cuda:
        raise _lib.VhError('embed_sum_pe: ids must be on the HIP device')
    if any(t.dim() != 2 or t.shape[1] != d for t in tables):
        raise _lib.VhError('embed_sum_pe: every table must be (vocab, d)')
    arr = (C.c_void_p * n)(*[ptr(_f32(t, 'table')) for t in tables])
    vocab = (C.c_int32 * n)(*[int(t.shape[0]) for t in tables])    # ids are range-checked in the kernel
    check(_lib.lib().vh_embed_sum_pe(
        ids.data_ptr(), ids.stride(0), ids.stride(1), ids.stride(2), arr, vocab, n,
        ptr(_f32(pe, 'pe')), pos0, ptr(lens), ptr(_f32(out, 'out')), out.stride(0), out_t0,
        B, T, d, ptr(_lib.err_flag(out.device)), ptr(row_pos0), ptr(row_t0), _drop_ref(drop), stream()),
        'vh_embed_sum_pe')
    return out


def ids_to_device(ids, device, vocab, what):
    """Token ids → `device`.  Ids still on the host are range-checked here, before they travel (the
    reference's nn.Embedding raises IndexError); device-resident ids are checked by the kernels
    (`_lib.raise_device_errors`)."""
    if not ids.is_cuda and ids.numel():
        # numpy on the tensor's own memory, not torch's CPU reductions: those fan a 70 k-element min / max out over
        # every core `os.cpu_count()` reports (256 on a GPU box whose cgroup grants 16) and then cost 2-70 ms per
        # call, erratically — the NAR training step with host-resident batches ran at 40-60 ms instead of 28
        # (gpurun_out/nar_host.log, round 4); numpy's single-threaded pass over the same bytes takes ~30 us
        a = ids.detach().numpy()
        lo, hi = int(a.min()), int(a.max())
        if lo < 0 or hi >= vocab:
            raise IndexError(f'{what}: id {lo if lo < 0 else hi} is outside [0, {vocab}) (index out of range)')
    return _lib.to_device_async(ids, device)


def _dev_f32(t, name):
    """data_ptr of a row-major fp32 DEVICE tensor (a host pointer handed to a kernel is a fault)."""
    if not t.is_cuda:
        raise _lib.VhError(f'{name} is on {t.device}: kernels take HIP device tensors only')
    return _f32(t, name).data_ptr()


def add_pe(x, pe, pos0=0):
    """x (B, T, d) + pe[pos0 : pos0 + T] (pe (P, 1, d) or (P, d)) -> new tensor: PositionalEncoding.forward as a module call."""
    B, T, d = x.shape
    if pos0 + T > pe.shape[0] or pe.shape[-1] != d:
        raise _lib.VhError(f'add_pe: positions {pos0}..{pos0 + T} / width {d} outside the table {tuple(pe.shape)}')
    x = x.contiguous()
    out = torch.empty_like(x)
    check(_lib.lib().vh_add_pe(_dev_f32(x, 'x'), ptr(_f32(pe, 'pe')), ptr(out), B, T, d, pos0, stream()), 'vh_add_pe')
    return out


def layernorm(x, gamma, beta, out=None, ada_scale=None, ada_shift=None, eps=1e-5):
    d = x.shape[-1]
    rows = x.numel() // d
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().vh_layernorm(ptr(_f32(x, 'x')), ptr(gamma), ptr(beta), ptr(ada_scale),
                                  ptr(ada_shift), ptr(out), rows, d, eps, stream()), 'vh_layernorm')
    return out


def _ln_args(ln):
    if ln is None:
        return None, None, None, None, 0.0
    g, b, sc, sh, eps = ln
    return ptr(g), ptr(b), ptr(sc), ptr(sh), float(eps)


def linear(a, w, bias=None, residual=None, out=None, act=ACT_NONE, ln=None):
    """out = act(LN?(a) @ w.T + bias) + residual.  a (M,K) (row stride lda), w (N,K)."""
    M, K = a.shape
    N, K2 = w.shape
    if K != K2:
        raise _lib.VhError(f'linear: K mismatch {K} vs {K2}')
    if a.stride(1) != 1 or not w.is_contiguous():
        raise _lib.VhError('linear: operands must be row-major')
    if out is None:
        ldo = (N + 3) // 4 * 4
        out = torch.empty(M, ldo, device=a.device, dtype=torch.float32)[:, :N]
    if out.stride(1) != 1 or (residual is not None and residual.stride(1) != 1):
        raise _lib.VhError('linear: out/residual must be row-major')
    if bias is not None and bias.numel() != N:
        raise _lib.VhError('linear: bias size')
    if residual is not None and tuple(residual.shape) != (M, N):
        raise _lib.VhError('linear: residual shape')
    g, b, sc, sh, eps = _ln_args(ln)
    check(_lib.lib().vh_linear(
        _dev_f32(a, 'a'), a.stride(0), ptr(_f32(w, 'w')), ptr(bias),
        _dev_f32(residual, 'residual') if residual is not None else None,
        residual.stride(0) if residual is not None else 0,
        _dev_f32(out, 'out'), out.stride(0), M, N, K, act, g, b, sc, sh, eps, stream()),
        'vh_linear')
    return out


def linear_ws(a, w, bias=None, residual=None, out=None, act=ACT_NONE, workspace=None):
    """vh_linear_ws: split-K path for wide K (see include/valle_hip.h)."""
    M, K = a.shape
    N, K2 = w.shape
    if K != K2:
        raise _lib.VhError(f'linear_ws: K mismatch {K} vs {K2}')
    if a.stride(1) != 1 or not w.is_contiguous():
        raise _lib.VhError('linear_ws: operands must be row-major')
    if out is None:
        out = torch.empty(M, (N + 3) // 4 * 4, device=a.device, dtype=torch.float32)[:, :N]
    if tuple(out.shape) != (M, N) or out.stride(1) != 1:
        raise _lib.VhError(f'linear_ws: out must be a row-major ({M},{N}) tensor')
    if bias is not None and bias.numel() != N:
        raise _lib.VhError('linear_ws: bias size')
    if residual is not None and (tuple(residual.shape) != (M, N) or residual.stride(1) != 1):
        raise _lib.VhError('linear_ws: residual shape')
    need = _lib.lib().vh_linear_ws_bytes(M, N, K)
    if workspace is None and need:
        workspace = torch.empty(need // 4, device=a.device, dtype=torch.float32)
    if need and workspace.numel() * 4 < need:
        raise _lib.VhError(f'linear_ws: workspace of {workspace.numel() * 4} B < {need} B')
    check(_lib.lib().vh_linear_ws(
        _dev_f32(a, 'a'), a.stride(0), ptr(_f32(w, 'w')), ptr(bias),
        _dev_f32(residual, 'residual') if residual is not None else None,
        residual.stride(0) if residual is not None else 0,
        _dev_f32(out, 'out'), out.stride(0), M, N, K, act, ptr(workspace),
        workspace.numel() * 4 if workspace is not None else 0, stream()), 'vh_linear_ws')
    return out


def linear_qkv(a, wqkv, q_out, kcache, vcache, B, T, n_heads, cache_len=None, ln=None):
    """QKV projection; Q → q_out (B*T, d); K,V rows appended to the caches (B,h,S_max,64)."""
    M, d = a.shape
    if M != B * T or tuple(wqkv.shape) != (3 * d, d) or d != n_heads * HEAD_DIM:
        raise _lib.VhError(f'linear_qkv: shapes a={tuple(a.shape)} w={tuple(wqkv.shape)} B={B} T={T}')
    S_max = kcache.shape[2]
    if tuple(kcache.shape) != (B, n_heads, S_max, HEAD_DIM) or kcache.shape != vcache.shape:
        raise _lib.VhError(f'linear_qkv: cache shape {tuple(kcache.shape)}')
    if cache_len is None and T > S_max:
        raise _lib.VhError('linear_qkv: T > S_max')
    g, b, sc, sh, eps = _ln_args(ln)
    check(_lib.lib().vh_linear_qkv(
        _f32(a, 'a').data_ptr(), a.stride(0), ptr(wqkv), q_out.data_ptr(), q_out.stride(0),
        ptr(kcache), ptr(vcache), ptr(cache_len), B, T, d, n_heads, S_max, g, b, sc, sh, eps,
        stream()), 'vh_linear_qkv')
    return q_out


def ln_fold(w, gamma, beta, bias=None):
    """(Wf, c1, c2) of vh_ln_fold: LN(x) @ w.T + bias == rstd * (x @ Wf.T - mean * c1) + c2."""
    N, K = w.shape
    if not w.is_contiguous() or gamma.numel() != K or beta.numel() != K:
        raise _lib.VhError(f'ln_fold: w={tuple(w.shape)} gamma={tuple(gamma.shape)}')
    wf = torch.empty_like(w)
    c = torch.empty(2, (N + 3) // 4 * 4, device=w.device, dtype=torch.float32)
    check(_lib.lib().vh_ln_fold(_f32(w, 'w').data_ptr(), ptr(gamma), ptr(beta), ptr(bias), ptr(wf),
                                c[0].data_ptr(), c[1].data_ptr(), N, K, stream()), 'vh_ln_fold')
    return wf, c[0, :N], c[1, :N]


def linear_folded(a, folded, residual=None, out=None, act=ACT_NONE, eps=1e-5):
    """act(LN(a) @ w.T + bias) + residual from the folded triple of ln_fold (decode rows, M <= 64)."""
    wf, c1, c2 = folded
    M, K = a.shape
    N = wf.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    check(_lib.lib().vh_linear_folded(
        _f32(a, 'a').data_ptr(), a.stride(0), ptr(wf), ptr(c1), ptr(c2), ptr(residual),
        residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), M, N, K, act,
        eps, stream()), 'vh_linear_folded')
    return out


def linear_qkv_folded(a, folded, q_out, kcache, vcache, B, T, n_heads, cache_len=None, eps=1e-5):
    wf, c1, c2 = folded
    M, d = a.shape
    S_max = kcache.shape[2]
    if M != B * T or tuple(wf.shape) != (3 * d, d) or tuple(kcache.shape) != (B, n_heads, S_max, HEAD_DIM):
        raise _lib.VhError(f'linear_qkv_folded: shapes a={tuple(a.shape)} w={tuple(wf.shape)}')
    check(_lib.lib().vh_linear_qkv_folded(
        _dev_f32(a, 'a'), a.stride(0), ptr(wf), ptr(c1), ptr(c2),
        q_out.data_ptr(), q_out.stride(0), ptr(kcache), ptr(vcache), ptr(cache_len), B, T, d, n_heads, S_max,
        eps, stream()), 'vh_linear_qkv_folded')
    return q_out


def ffn_decode_ws(M, d, dff, device):
    n = _lib.lib().vh_ffn_decode_ws_bytes(M, d, dff)
    return torch.empty(max(n, 16) // 4, device=device, dtype=torch.float32) if n else None


def ffn_decode(x, folded, w2, b2=None, out=None, workspace=None, eps=1e-5):
    """out = x + b2 + GELU(LN2(x) @ W1.T + b1) @ w2.T for M <= 64 decode rows (vh_ffn_decode); `folded` =
    ln_fold(W1, ln2_gamma, ln2_beta, b1).  out may be x itself."""
    wf, c1, c2 = folded
    M, d = x.shape
    dff = wf.shape[0]
    if tuple(wf.shape) != (dff, d) or tuple(w2.shape) != (d, dff) or not w2.is_contiguous() or not wf.is_contiguous():
        raise _lib.VhError(f'ffn_decode: x {tuple(x.shape)} w1f {tuple(wf.shape)} w2 {tuple(w2.shape)}')
    if c1.numel() != dff or c2.numel() != dff or (b2 is not None and b2.numel() != d):
        raise _lib.VhError('ffn_decode: c1 / c2 / b2 sizes')
    if out is None:
        out = torch.empty(M, d, device=x.device, dtype=torch.float32)
    if tuple(out.shape) != (M, d) or out.stride(1) != 1 or x.stride(1) != 1:
        raise _lib.VhError('ffn_decode: x / out must be row-major (M, d)')
    if workspace is None:
        workspace = ffn_decode_ws(M, d, dff, x.device)
    if workspace is None:
        raise _lib.VhError(f'ffn_decode: unsupported shape M={M} d={d} dff={dff}')
    check(_lib.lib().vh_ffn_decode(
        _dev_f32(x, 'x'), x.stride(0), ptr(_f32(wf, 'w1f')), ptr(c1), ptr(c2), ptr(_f32(w2, 'w2')), ptr(b2),
        _dev_f32(out, 'out'), out.stride(0), M, d, dff, eps, ptr(workspace), workspace.numel() * 4, stream()),
        'vh_ffn_decode')
    return out


def attn_rows(q, kcache, vcache, out, B, n_heads, Tq, Tk, mode, x_len=0, x_len_dev=None,
              kv_len=None, mask=None, pad=None, lse2=None):
    S_max = kcache.shape[2]
    if tuple(kcache.shape) != (B, n_heads, S_max, HEAD_DIM) or Tk > S_max or Tq > Tk:
        raise _lib.VhError(f'attn_rows: cache {tuple(kcache.shape)} Tq={Tq} Tk={Tk}')
    if q.shape[0] != B * Tq or out.shape[0] != B * Tq:
        raise _lib.VhError('attn_rows: q/out rows')
    if mask is not None and mask.dim() == 3:       # one mask per batch row (inference only: no lse2)
        if mask.dtype != torch.uint8 or tuple(mask.shape) != (B, Tq, Tk) or not mask.is_contiguous() or lse2 is not None:
            raise _lib.VhError('attn_rows: a per-row mask must be contiguous uint8 (B,Tq,Tk); the training forward takes 2-D masks')
        if pad is not None and (pad.dtype != torch.uint8 or tuple(pad.shape) != (B, Tk)):
            raise _lib.VhError('attn_rows: pad must be uint8 (B,Tk)')
        check(_lib.lib().vh_attn_rows_bmask(
            _dev_f32(q, 'q'), q.stride(0), ptr(kcache), ptr(vcache), _dev_f32(out, 'out'), out.stride(0), B, n_heads, Tq, Tk,
            S_max, ptr(mask), Tq * Tk, ptr(pad), stream()), 'vh_attn_rows_bmask')
        return out
    if mask is not None and (mask.dtype != torch.uint8 or tuple(mask.shape) != (Tq, Tk)):
        raise _lib.VhError('attn_rows: mask must be uint8 (Tq,Tk)')
    if pad is not None and (pad.dtype != torch.uint8 or tuple(pad.shape) != (B, Tk)):
        raise _lib.VhError('attn_rows: pad must be uint8 (B,Tk)')
    if lse2 is not None:        # training forward: keep the row log-sum-exp for attn_rows_bwd
        if tuple(lse2.shape) != (B, n_heads, Tq) or lse2.dtype != torch.float32:
            raise _lib.VhError('attn_rows: lse2 must be float32 (B, n_heads, Tq)')
        check(_lib.lib().vh_attn_rows_lse(
            _dev_f32(q, 'q'), q.stride(0), ptr(kcache), ptr(vcache), _dev_f32(out, 'out'), out.stride(0), B,
            n_heads, Tq, Tk, S_max, mode, x_len, ptr(x_len_dev), ptr(kv_len), ptr(mask), ptr(pad),
            ptr(lse2), stream()), 'vh_attn_rows_lse')
        return out
    check(_lib.lib().vh_attn_rows(
        _dev_f32(q, 'q'), q.stride(0), ptr(kcache), ptr(vcache), _dev_f32(out, 'out'), out.stride(0), B,
        n_heads, Tq, Tk, S_max, mode, x_len, ptr(x_len_dev), ptr(kv_len), ptr(mask), ptr(pad),
        stream()), 'vh_attn_rows')
    return out


def attn_rows_bwd(q, kcache, vcache, out, dout, lse2, dq, dk, dv, B, n_heads, T, mode, x_len=0,
                  x_len_dev=None, kv_len=None, mask=None, pad=None):
    """dq, dk, dv (each (B*T, d) views with a common row stride, heads at columns h*64) of
    attn_rows(q, kcache, vcache) given dout; P is recomputed tile by tile, never materialised.  Five products in one
    kernel + the slab reduce (vh_attn_rows_bwd_ws); `_lib.lib().vh_set_tuning(VH_TUNE_ATTN_BWD = 13, 1)` selects the
    older two-kernel, seven-product form."""
    S_max = kcache.shape[2]
    if tuple(kcache.shape) != (B, n_heads, S_max, HEAD_DIM) or kcache.shape != vcache.shape or T > S_max:
        raise _lib.VhError(f'attn_rows_bwd: cache {tuple(kcache.shape)} T={T}')
    if not (dq.stride(0) == dk.stride(0) == dv.stride(0)) or dq.stride(1) != 1:
        raise _lib.VhError('attn_rows_bwd: dq/dk/dv must share one row stride')
    # scratch: the dQ partial slabs of the five-product kernel (chunks x (B*h, T, 64) floats: 134 MB at the configs[3] AR
    # step) or, under VH_TUNE_ATTN_BWD = 1, the two-kernel form's D vector.  One growing buffer per (device, stream) —
    # launches of a stream run in order, so its 12 layers share it — instead of 12 allocations of that size per step
    # whose sizes change with every ragged batch.
    need = _lib.lib().vh_attn_rows_bwd_ws_bytes(B, n_heads, T)
    ws = _stream_ws(_BWD_WS, q.device, need)
    check(_lib.lib().vh_attn_rows_bwd_ws(
        q.data_ptr(), q.stride(0), ptr(kcache), ptr(vcache), out.data_ptr(), out.stride(0),
        dout.data_ptr(), dout.stride(0), ptr(lse2), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
        dq.stride(0), B, n_heads, T, S_max, mode, x_len, ptr(x_len_dev), ptr(kv_len), ptr(mask), ptr(pad),
        ptr(ws), ws.numel() * 4, stream()), 'vh_attn_rows_bwd_ws')
    return dq, dk, dv


def softmax_rows(S, ld, B, n_heads, Tq, Tk, scale, mode=MASK_FULL, x_len=0, x_len_dev=None, kv_len=None, mask=None,
                 pad=None):
    """P = softmax(S * scale + mask) in place over the rows of the (B, h, Tq, Tk) maps inside S (row stride ld); the mask
    arguments are those of attn_rows."""
    check(_lib.lib().vh_softmax_rows(S.data_ptr(), ld, B, n_heads, Tq, Tk, float(scale), mode, x_len, ptr(x_len_dev),
                                     ptr(kv_len), ptr(mask), ptr(pad), stream()), 'vh_softmax_rows')
    return S


def attn_generic(q, k, v, out, scale, **spec):
    """Attention for a head width the flash kernels are not built for (they serve 64 = every configuration of the path;
    valle/models/modules.py:109-111 allows any divisor of d_model): the probabilities are MATERIALISED — S = Q K^T on the
    batched MFMA GEMM, the masked row softmax, O = P V on the same GEMM.  Correct for any head width that is a multiple
    of 4, not tuned: O(T^2) memory per (batch row, head).  q (B, h, Tq, hd), k / v (B, h, Tk, hd), out (B, h, Tq, hd):
    views with unit stride in the last dimension (the per-head column blocks of a (rows, 3 d) projection are read in
    place).  Returns P (B, h, Tq, Tk) for a backward that wants it."""
    B, h, Tq, hd = q.shape
    Tk = k.shape[2]
    tp = (Tk + 3) // 4 * 4
    P = torch.empty(B, h, Tq, tp, device=q.device, dtype=torch.float32)[..., :Tk]
    gemm(q, k, P)                                               # raw scores
    mask = spec.get('mask')
    if mask is not None and mask.dim() == 3:                    # one mask per batch row: the row softmax row by row
        pad = spec.get('pad')
        for b in range(B):
            softmax_rows(P[b:b + 1], tp, 1, h, Tq, Tk, scale, mode=spec['mode'], mask=mask[b],
                         pad=None if pad is None else pad[b:b + 1])
    else:
        softmax_rows(P, tp, B, h, Tq, Tk, scale, **spec)
    gemm(P, v, out, b_kmajor=True)
    return P


def attn_decode_ws(B, n_heads, n_split, device):
    """Workspace of a key-split decode attention: split records + the per-(b, head) ticket words, which must start at
    zero (the kernel re-arms them itself) — hence zeros, not empty."""
    n = _lib.lib().vh_attn_decode_ws_bytes(B, n_heads, n_split)
    return torch.zeros(max(n, 16) // 4, device=device, dtype=torch.float32) if n else None


def attn_decode(q, kcache, vcache, out, cache_len, len_bias, n_split=1, partial=None):
    B, n_heads, S_max, hd = kcache.shape
    if hd != HEAD_DIM or q.shape[0] != B or out.shape[0] != B:
        raise _lib.VhError('attn_decode: shapes')
    if cache_len.dtype != torch.int32 or cache_len.numel() != B:
        raise _lib.VhError('attn_decode: cache_len must be int32 (B)')
    check(_lib.lib().vh_attn_decode(
        q.data_ptr(), q.stride(0), ptr(kcache), ptr(vcache), out.data_ptr(), out.stride(0),
        ptr(cache_len), len_bias, B, n_heads, S_max, n_split, ptr(partial), stream()),
        'vh_attn_decode')
    return out


def attn_decode_shared_ws(B, n_heads, prefix_len, n_split, device):
    """Record workspace of `attn_decode_shared` (vh_attn_decode_shared_ws_bytes), as a float32 tensor."""
    n = _lib.lib().vh_attn_decode_shared_ws_bytes(B, n_heads, prefix_len, n_split)
    return torch.empty(max(n, 16) // 4, device=device, dtype=torch.float32)


def attn_decode_shared(q, kprefix, vprefix, prefix_len, ksuffix, vsuffix, out, suffix_len, len_bias, n_split=1, partial=None):
    """One-query attention of B beams over a SHARED prompt (kprefix / vprefix (1, h, prefix_S, 64), first `prefix_len` rows)
    followed by each beam's own rows (ksuffix / vsuffix (B, h, S_suf, 64), suffix_len[b] + len_bias of them)."""
    B, n_heads, S_suf, hd = ksuffix.shape
    if hd != HEAD_DIM or tuple(kprefix.shape[:2]) != (1, n_heads) or kprefix.shape[3] != HEAD_DIM or q.shape[0] != B:
        raise _lib.VhError(f'attn_decode_shared: prefix {tuple(kprefix.shape)} suffix {tuple(ksuffix.shape)} q {tuple(q.shape)}')
    if suffix_len.dtype != torch.int32 or suffix_len.numel() != B:
        raise _lib.VhError('attn_decode_shared: suffix_len must be int32 (B)')
    if partial is None:
        partial = attn_decode_shared_ws(B, n_heads, prefix_len, n_split, q.device)
    check(_lib.lib().vh_attn_decode_shared(
        _dev_f32(q, 'q'), q.stride(0), ptr(kprefix), ptr(vprefix), prefix_len, kprefix.shape[2], ptr(ksuffix), ptr(vsuffix),
        _dev_f32(out, 'out'), out.stride(0), ptr(suffix_len), len_bias, B, n_heads, S_suf, n_split, ptr(partial),
        partial.numel() * 4, stream()), 'vh_attn_decode_shared')
    return out


def attn_decode_kv16(q, kcache16, vcache16, out, cache_len, len_bias):
    """attn_decode over a bf16 K/V cache (perf mode); one (row, head) per workgroup."""
    B, n_heads, S_max, hd = kcache16.shape
    if hd != HEAD_DIM or kcache16.dtype != H16 or vcache16.dtype != H16 or q.shape[0] != B:
        raise _lib.VhError('attn_decode_kv16: bf16 caches (B, h, S_max, 64)')
    if cache_len.dtype != torch.int32 or cache_len.numel() != B:
        raise _lib.VhError('attn_decode_kv16: cache_len must be int32 (B)')
    check(_lib.lib().vh_attn_decode_kv16(_dev_f32(q, 'q'), q.stride(0), ptr(kcache16), ptr(vcache16), _dev_f32(out, 'out'),
                                         out.stride(0), ptr(cache_len), len_bias, B, n_heads, S_max, stream()),
          'vh_attn_decode_kv16')
    return out


def linear_qkv_folded_kv16(a, folded, q_out, kcache16, vcache16, n_heads, cache_len, eps=1e-5):
    wf, c1, c2 = folded
    B, d = a.shape
    S_max = kcache16.shape[2]
    if tuple(wf.shape) != (3 * d, d) or tuple(kcache16.shape) != (B, n_heads, S_max, HEAD_DIM) or \
            kcache16.dtype != H16 or vcache16.dtype != H16:
        raise _lib.VhError('linear_qkv_folded_kv16: shapes / dtypes')
    check(_lib.lib().vh_linear_qkv_folded_kv16(_dev_f32(a, 'a'), a.stride(0), ptr(wf), ptr(c1), ptr(c2), q_out.data_ptr(),
                                               q_out.stride(0), ptr(kcache16), ptr(vcache16), ptr(cache_len), B, d,
                                               n_heads, S_max, eps, stream()), 'vh_linear_qkv_folded_kv16')
    return q_out


# ---- perf mode of the MFMA-bound legs: bf16 operands, fp32 accumulate (include/valle_hip.h; SECONDARY, never the parity path)
def _bf16(t, name):
    if t.dtype != H16 or not t.is_cuda or t.stride(-1) != 1:
        raise _lib.VhError(f'{name} must be a row-major {H16} HIP tensor, got {t.dtype} on {t.device}')
    return t.data_ptr()


def to_bf16(x, out=None):
    """Round-to-nearest-even narrowing of a (rows, cols) fp32 matrix (vh_to_bf16): weights once per weight set."""
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=H16)
    check(_lib.lib().vh_to_bf16(_dev_f32(x, 'x'), x.stride(0), _bf16(out, 'out'), out.stride(0), rows, cols, stream()),
          'vh_to_bf16')
    return out


def layernorm_bf16(x, gamma, beta, out=None, ada_scale=None, ada_shift=None, eps=1e-5):
    d = x.shape[-1]
    rows = x.numel() // d
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=H16)
    check(_lib.lib().vh_layernorm_bf16(ptr(_f32(x, 'x')), ptr(gamma), ptr(beta), ptr(ada_scale), ptr(ada_shift),
                                       _bf16(out, 'out'), rows, d, eps, stream()), 'vh_layernorm_bf16')
    return out


def linear_bf16(a, w, bias=None, residual=None, out=None, act=ACT_NONE, out_bf16=False):
    """out = act(a @ w.T + bias) + residual with a (M,K), w (N,K) bf16; fp32 accumulate; out fp32 or bf16."""
    M, K = a.shape
    N, K2 = w.shape
    if K != K2 or not w.is_contiguous():
        raise _lib.VhError(f'linear_bf16: K mismatch {K} vs {K2} / w must be contiguous')
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=H16 if out_bf16 else torch.float32)
    if out.dtype != (H16 if out_bf16 else torch.float32) or out.stride(1) != 1:
        raise _lib.VhError('linear_bf16: out dtype / layout')
    check(_lib.lib().vh_linear_bf16(_bf16(a, 'a'), a.stride(0), _bf16(w, 'w'), ptr(bias),
                                    _dev_f32(residual, 'residual') if residual is not None else None,
                                    residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0),
                                    int(out_bf16), M, N, K, act, stream()), 'vh_linear_bf16')
    return out


Q16_PRESCALE = 0.125 * 1.4426950408889634     # perf mode: q leaves linear_qkv_bf16 scaled by 1/sqrt(64) * log2(e); attn_rows_bf16 expects it


def linear_qkv_bf16(a, wqkv, q_out, kcache16, vcache16, B, T, n_heads, cache_len=None):
    d = a.shape[1]
    S_max = kcache16.shape[2]
    if tuple(wqkv.shape) != (3 * d, d) or tuple(kcache16.shape) != (B, n_heads, S_max, HEAD_DIM) or a.shape[0] != B * T:
        raise _lib.VhError('linear_qkv_bf16: shapes')
    check(_lib.lib().vh_linear_qkv_bf16(_bf16(a, 'a'), a.stride(0), _bf16(wqkv, 'wqkv'), _bf16(q_out, 'q_out'),
                                        q_out.stride(0), _bf16(kcache16, 'kcache'), _bf16(vcache16, 'vcache'), ptr(cache_len),
                                        B, T, d, n_heads, S_max, stream()), 'vh_linear_qkv_bf16')
    return q_out


def attn_rows_bf16(q, kcache16, vcache16, out, B, n_heads, Tq, Tk, mode, x_len=0, x_len_dev=None, kv_len=None):
    S_max = kcache16.shape[2]
    if tuple(kcache16.shape) != (B, n_heads, S_max, HEAD_DIM) or Tk > S_max or Tq > Tk or q.shape[0] != B * Tq:
        raise _lib.VhError(f'attn_rows_bf16: cache {tuple(kcache16.shape)} Tq={Tq} Tk={Tk}')
    check(_lib.lib().vh_attn_rows_bf16(_bf16(q, 'q'), q.stride(0), _bf16(kcache16, 'kcache'), _bf16(vcache16, 'vcache'),
                                       _bf16(out, 'out'), out.stride(0), B, n_heads, Tq, Tk, S_max, mode, x_len,
                                       ptr(x_len_dev), ptr(kv_len), stream()), 'vh_attn_rows_bf16')
    return out


def greedy_step(logits, V, eos, codes, eos_count, audio_emb, pe, audio_pos, cache_len, x_next,
                pos_base=None):
    B = logits.shape[0]
    d = x_next.shape[1]
    check(_lib.lib().vh_greedy_step(
        logits.data_ptr(), logits.stride(0), V, eos, ptr(codes), codes.stride(0), ptr(eos_count),
        ptr(pos_base), ptr(audio_emb), ptr(pe), ptr(audio_pos), ptr(cache_len), ptr(_f32(x_next, 'x_next')), B, d,
        stream()), 'vh_greedy_step')


def head_greedy_ws(B, V, device):
    """Zeroed workspace of vh_head_greedy (arrival counters + one candidate per row and 16-column block)."""
    return torch.zeros(_lib.lib().vh_head_greedy_ws_bytes(B, V) // 4, device=device, dtype=torch.int32)


def head_greedy(x, proj_w, logits, V, eos, codes, eos_count, audio_emb, pe, audio_pos, cache_len, x_next, ws, pos_base=None):
    """The AR head and the greedy step in ONE launch: logits = x @ proj_w.T, then what greedy_step does with them."""
    B, d = x.shape
    check(_lib.lib().vh_head_greedy(
        ptr(_f32(x, 'x')), x.stride(0), ptr(_f32(proj_w, 'proj_w')), ptr(_f32(logits, 'logits')), logits.stride(0), V, eos,
        ptr(codes), codes.stride(0), ptr(eos_count), ptr(pos_base), ptr(audio_emb), ptr(pe), ptr(audio_pos), ptr(cache_len),
        ptr(_f32(x_next, 'x_next')), B, d, ptr(ws), ws.numel() * 4, stream()), 'vh_head_greedy')


def sample_step(logits, V, eos, top_k, top_p, temperature, seed, codes, eos_count, sum_logprobs, audio_emb,
                pe, audio_pos, cache_len, x_next, pos_base=None):
    B = logits.shape[0]
    d = x_next.shape[1]
    check(_lib.lib().vh_sample_step(
        logits.data_ptr(), logits.stride(0), V, eos, int(top_k), float(top_p), float(temperature),
        int(seed) & (2 ** 64 - 1), ptr(codes), codes.stride(0), ptr(eos_count), ptr(pos_base),
        ptr(sum_logprobs), ptr(audio_emb), ptr(pe), ptr(audio_pos), ptr(cache_len), ptr(_f32(x_next, 'x_next')),
        B, d, stream()), 'vh_sample_step')


def pad32(n):
    return (n + 31) // 32 * 32


_TAIL_WS = {}
_BWD_WS = {}


def _stream_ws(pool, device, need, floor=1 << 20):
    """A float32 workspace of >= need bytes owned by (device, current stream): grown when too small, kept otherwise."""
    key = (device.index, stream())
    ws = pool.get(key)
    if ws is None or ws.numel() * 4 < need:
        if len(pool) >= 8 and key not in pool:       # streams come and go: keep the most recent few
            torch.cuda.synchronize(device)           # (rare: nothing in flight may still be using the one let go)
            pool.pop(next(iter(pool)))
        # (the buffer let go was allocated on this stream and only ever used by launches of this stream: torch's caching
        # allocator hands its block to later work of the same stream only, so no wait is needed before it is replaced)
        ws = pool[key] = torch.empty(max(need, floor) // 4, device=device, dtype=torch.float32)
    return ws


def _tail_ws(device, stream_handle, M, N, K):
    """Workspace of vh_linear_ex's tail split (K slices of the tiles beyond the last multiple of 256; <= 16 MiB), one per
    (device, stream): launches of one stream run in order, so they can share it."""
    need = _lib.lib().vh_linear_ex_ws_bytes(M, N, K)
    if not need:
        return None
    key = (device.index, stream_handle)
    ws = _TAIL_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        if len(_TAIL_WS) >= 8:                       # streams come and go: keep the most recent few (16 MiB each)
            torch.cuda.synchronize(device)           # (rare: nothing in flight may still be writing the one let go)
            _TAIL_WS.pop(next(iter(_TAIL_WS)))
        ws = _TAIL_WS[key] = torch.empty(max(need, 1 << 24) // 4, device=device, dtype=torch.float32)
    return ws


def linear_ex(a, w, bias=None, residual=None, out=None, pre_out=None, act=ACT_NONE, K=None, colsum=None, drop=None):
    """vh_linear_ex: out = act(a @ w.T + bias) + residual on the tile kernels whatever M is, with the training
    epilogues (pre_out: also store the pre-activation; ACT_GELU_BWD: out = (a @ w.T) * gelu'(residual)).
    `K` overrides the contraction width (operands whose rows are zero-padded to a multiple of 32).
    colsum (N,) fp32, N % 128 == 0: += the column sums of `out` (a bias gradient) in the same launch.
    drop: a dropout.spec — out = dropout(act(a @ w.T + bias)) + residual, field indexed by (row, column) of `out`."""
    M = a.shape[0]
    N = w.shape[0]
    K = K or a.shape[1]
    if a.stride(1) != 1 or w.stride(1) != 1 or a.stride(0) < K:
        raise _lib.VhError('linear_ex: operands must be row-major and cover K')
    if w.stride(0) != K:
        raise _lib.VhError(f'linear_ex: the weight rows must be exactly K={K} wide (stride {w.stride(0)})')
    if out is None:
        out = torch.empty(M, (N + 3) // 4 * 4, device=a.device, dtype=torch.float32)[:, :N]
    for t, name in ((residual, 'residual'), (pre_out, 'pre_out'), (out, 'out')):
        if t is not None and (tuple(t.shape) != (M, N) or t.stride(1) != 1):
            raise _lib.VhError(f'linear_ex: {name} must be a row-major ({M},{N}) tensor')
    st = stream()
    ws = _tail_ws(a.device, st, M, N, K)
    check(_lib.lib().vh_linear_ex(
        _dev_f32(a, 'a'), a.stride(0), _dev_f32(w, 'w'), ptr(bias),
        _dev_f32(residual, 'residual') if residual is not None else None,
        residual.stride(0) if residual is not None else 0, _dev_f32(out, 'out'), out.stride(0),
        _dev_f32(pre_out, 'pre_out') if pre_out is not None else None,
        pre_out.stride(0) if pre_out is not None else 0, ptr(colsum), M, N, K, act, _drop_ref(drop), ptr(ws),
        ws.numel() * 4 if ws is not None else 0, st), 'vh_linear_ex')
    return out


def transpose(w, ldo=None, out=None):
    """out (cols, ldo) = w (rows, cols)^T, rows zero-padded up to ldo (default: rows rounded up to 32)."""
    rows, cols = w.shape
    ldo = ldo or pad32(rows)
    if out is None:
        out = torch.empty(cols, ldo, device=w.device, dtype=torch.float32)
    if w.stride(1) != 1 or tuple(out.shape) != (cols, ldo) or not out.is_contiguous():
        raise _lib.VhError('transpose: w must be row-major and out a contiguous (cols, ldo) tensor')
    check(_lib.lib().vh_transpose(_dev_f32(w, 'w'), w.stride(0), rows, cols, _dev_f32(out, 'out'), ldo, stream()),
          'vh_transpose')
    return out


class TransposePlan:
    """Transposed copies (cols, ldo) of a fixed set of weight matrices, refreshed by ONE launch (vh_transpose_many).
    The weights must stay where they are (parameters re-homed into FlatAdamW's flat buffer do; the plan is rebuilt by
    its owner when a data_ptr changes)."""

    def __init__(self, weights, ldos):
        import numpy as np
        dev = weights[0].device
        self.outs = [torch.empty(w.shape[1], ldo, device=dev, dtype=torch.float32) for w, ldo in zip(weights, ldos)]
        rec = np.zeros(len(weights), dtype=[('in', 'u8'), ('out', 'u8'), ('ldi', 'i4'), ('rows', 'i4'), ('cols', 'i4'),
                                            ('ldo', 'i4'), ('tile0', 'i4'), ('pad', 'i4')])
        tile = 0
        for i, (w, ldo, o) in enumerate(zip(weights, ldos, self.outs)):
            if w.dim() != 2 or w.stride(1) != 1 or w.dtype != torch.float32 or not w.is_cuda or ldo < w.shape[0]:
                raise _lib.VhError('TransposePlan: row-major fp32 device matrices, ldo >= rows')
            rec[i] = (w.data_ptr(), o.data_ptr(), w.stride(0), w.shape[0], w.shape[1], ldo, tile, 0)
            tile += -(-w.shape[1] // 32) * -(-ldo // 32)
        self.n, self.tiles = len(weights), tile
        self.key = tuple(w.data_ptr() for w in weights)
        self.items = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)

    def run(self):
        check(_lib.lib().vh_transpose_many(self.items.data_ptr(), self.n, self.tiles, stream()), 'vh_transpose_many')
        return self.outs


_tn_ws = {}


def gemm_tn(a, b, out=None):
    """out (NI, NJ) = a.T @ b for a (M, NI), b (M, NJ) stored token-major (dW = dY^T X), read in place."""
    M, NI = a.shape
    M2, NJ = b.shape
    if M != M2 or a.stride(1) != 1 or b.stride(1) != 1:
        raise _lib.VhError(f'gemm_tn: a {tuple(a.shape)} b {tuple(b.shape)}')
    if out is None:
        out = torch.empty(NI, (NJ + 3) // 4 * 4, device=a.device, dtype=torch.float32)[:, :NJ]
    if tuple(out.shape) != (NI, NJ) or out.stride(1) != 1:
        raise _lib.VhError('gemm_tn: out shape')
    need = _lib.lib().vh_gemm_tn_ws_bytes(M, NI, NJ)
    ws = _stream_ws(_tn_ws, a.device, need) if need else None     # one growing workspace per (device, stream)
    check(_lib.lib().vh_gemm_tn(_dev_f32(a, 'a'), a.stride(0), _dev_f32(b, 'b'), b.stride(0), _dev_f32(out, 'out'),
                                out.stride(0), M, NI, NJ, ptr(ws), ws.numel() * 4 if ws is not None else 0, stream()),
          'vh_gemm_tn')
    return out


def categorical_rows(logits, tokens, temperature=1.0, greedy=False, seed=0, stream_id=0, logprob=None):
    """tokens[r] ~ Categorical(logits[r] / temperature) (valle_nar.py:160) or arg-max when greedy.
    logits (R, V) row-major (any row stride); tokens: int64 (R,) view with any element stride."""
    R, V = logits.shape
    if logits.stride(1) != 1 or tokens.dtype != torch.int64 or tokens.dim() != 1 or tokens.shape[0] != R:
        raise _lib.VhError(f'categorical_rows: logits {tuple(logits.shape)} / tokens {tuple(tokens.shape)} {tokens.dtype}')
    if not tokens.is_cuda or (logprob is not None and (logprob.numel() != R or not logprob.is_contiguous())):
        raise _lib.VhError('categorical_rows: tokens / logprob must be device tensors of R entries')
    check(_lib.lib().vh_categorical_rows(_dev_f32(logits, 'logits'), logits.stride(0), V, R, float(temperature),
                                         int(bool(greedy)), int(seed) & (2 ** 64 - 1), int(stream_id) & 0xFFFFFFFF,
                                         tokens.data_ptr(), tokens.stride(0) if R > 1 else 1, ptr(logprob),
                                         stream()), 'vh_categorical_rows')
    return tokens


def gemm(a, b, out, a_kmajor=False, b_kmajor=False):
    """out = op(a) @ op(b) through vh_gemm_batched.  a, b, out are 2-D (rows, cols) or 4-D
    (batch, heads, rows, cols) tensors / views with unit stride in the last dimension:
      a_kmajor=False: a is (M, K);  True: a is stored (K, M)   (the left operand transposed)
      b_kmajor=False: b is (N, K) as an nn.Linear weight;  True: b is stored (K, N)
    out is (M, N).  Strides are taken from the tensors, so permuted views are read in place."""
    def spec(t):
        if t.dim() == 2:
            return t.shape[0], t.shape[1], t.stride(0), 0, 0, 1, 1
        if t.dim() == 4:
            return t.shape[2], t.shape[3], t.stride(2), t.stride(0), t.stride(1), t.shape[0], t.shape[1]
        raise _lib.VhError('gemm: operands must be 2-D or 4-D')
    for t in (a, b, out):
        if t.stride(-1) != 1 or t.dtype != torch.float32:
            raise _lib.VhError('gemm: fp32 operands with unit stride in the last dimension required')
    ar, ac, lda, sab, sah, nb, nh = spec(a)
    br, bc, ldb, sbb, sbh, nb2, nh2 = spec(b)
    cr, cc, ldc, scb, sch, nb3, nh3 = spec(out)
    M, K = (ac, ar) if a_kmajor else (ar, ac)
    N, K2 = (bc, br) if b_kmajor else (br, bc)
    if K != K2 or (cr, cc) != (M, N) or (nb, nh) != (nb2, nh2) or (nb, nh) != (nb3, nh3):
        raise _lib.VhError(f'gemm: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} out{tuple(out.shape)}')
    # few output tiles but a long K (weight gradients: K = every token of the batch): split K over
    # more workgroups; their partial tiles are summed into a zeroed `out` with fp32 atomics
    tiles = -(-M // 128) * -(-N // 128) * nb * nh
    splits = 1
    if tiles < 256 and K >= 2048:
        splits = max(1, min(32, 512 // tiles, K // 512))
    if splits > 1:
        out.zero_()
    check(_lib.lib().vh_gemm_batched(a.data_ptr(), lda, sab, sah, int(a_kmajor), b.data_ptr(), ldb, sbb, sbh,
                                     int(b_kmajor), out.data_ptr(), ldc, scb, sch, M, N, K, nb, nh, splits,
                                     stream()), 'vh_gemm_batched')
    return out
