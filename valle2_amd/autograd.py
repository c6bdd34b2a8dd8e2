"""Autograd wrappers: the training path (`training_step` → `loss.backward()`, valle_ar.py:43-90).

Forward passes run the same HIP kernels as inference.  Backward passes are hand-written HIP throughout:
  * dX = dY · W          → the NT LDS-DMA tile kernel (`vh_linear_ex`) on Wᵀ (`vh_transpose`, a few MB per layer),
                           with the GELU backward fused into linear_2's product (the forward keeps gelu'(pre): `VH_ACT_GELU_ERF_D` / `VH_ACT_MUL`);
  * dW = dYᵀ · X         → `vh_gemm_tn`: both operands read token-major as they lie, contraction split over the
                           chip, slabs summed in fixed order (bitwise reproducible);
  * attention            → `vh_attn_rows_bwd` (flash-style, P recomputed tile by tile);
  * everything row-wise  → csrc/train.hip (LayerNorm/AdaLN, cross entropy, embedding scatter, bias column sums).
There is no library GEMM and no CPU arithmetic on this path.
"""
from __future__ import annotations

import torch

import ctypes as C

from . import _lib, dropout, kernels, optim
from ._lib import check, ptr, stream

HEAD_DIM = kernels.HEAD_DIM

# Backward of the attention core:
#   'flash'        — vh_attn_rows_bwd: two hand-written kernels that recompute P tile by tile from the saved
#                    row log-sum-exp (no (T,T) maps in memory, no atomics);
#   'materialized' — S, P, dP as (B,h,T,T) tensors through vh_gemm_batched + row softmax kernels (kept as a
#                    second, independently written derivation the gradient tests compare against).
ATTENTION_BACKWARD = 'flash'


def _mm(a, b, out, a_kmajor=False, b_kmajor=False):
    """out = op(a) @ op(b) on the hand-written batched GEMM (strided attention views read in place)."""
    return kernels.gemm(a, b, out, a_kmajor=a_kmajor, b_kmajor=b_kmajor)


def _kpad(t):
    """`t` (rows, n) as an operand whose contraction width is a multiple of 32: itself when n already is,
    else a zero-padded copy (only the 1025-wide logit gradients of the AR head)."""
    rows, n = t.shape
    if n % 32 == 0 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t, n
    kp = kernels.pad32(n)
    buf = torch.zeros(rows, kp, device=t.device, dtype=torch.float32)
    buf[:, :n] = t
    return buf[:, :n], kp


def _dx(dy, w, kp, residual=None, act=kernels.ACT_NONE):
    """dX = dY · W for a forward y = x @ W.T: the NT tile kernel on Wᵀ (K x N, rows zero-padded to kp)."""
    wt = kernels.transpose(w.detach(), ldo=kp)
    m, k = dy.shape[0], w.shape[1]
    out = torch.empty(m, k, device=dy.device, dtype=torch.float32)
    return kernels.linear_ex(dy, wt, residual=residual, out=out, act=act, K=kp)


def _colsum(dy, bias):
    """Column sums of dy, accumulated into the bias' gradient buffer (its flat-gradient slice on the first gradient of a
    step, optim.grad_out)."""
    db = optim.grad_out(bias, zero=True)
    check(_lib.lib().vh_colsum(ptr(dy) if dy.is_contiguous() else dy.data_ptr(), dy.stride(0), ptr(db), dy.shape[0],
                               db.numel(), stream()), 'vh_colsum')
    return db


def _zeros_like(t):
    return torch.zeros_like(t, memory_format=torch.contiguous_format)


class LinearFn(torch.autograd.Function):
    """y = x @ W.T + b (+ residual).  x (M,K), W (N,K)."""

    @staticmethod
    def forward(ctx, x, w, b, residual):
        x = x.contiguous()
        n = w.shape[0]
        out = torch.empty(x.shape[0], (n + 3) // 4 * 4, device=x.device, dtype=torch.float32)[:, :n]
        kernels.linear(x, w.detach(), None if b is None else b.detach(), residual=residual, out=out)
        if n % 4:
            out = out.contiguous()        # ragged head (N = 1025): hand autograd a dense tensor
        ctx.save_for_backward(x, w, b)
        ctx.has_b, ctx.has_res = b is not None, residual is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        dyp, kp = _kpad(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:      # dX = dY . W
            dx = _dx(dyp, w, kp)
        if ctx.needs_input_grad[1]:      # dW = dY^T . X      (both operands stored with the token index as rows)
            dw = optim.grad_out(w)
            kernels.gemm_tn(dyp, x, out=dw)
        db = None
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = _colsum(dyp, b)
        return dx, dw, db, (dy if ctx.has_res else None)


# FeedForward's activation pair: the forward keeps gelu'(pre) where it would keep pre (one erf for GELU and derivative),
# the backward's epilogue multiplies by it.  (tools/ab_train_step.py flips this to (ACT_GELU, ACT_GELU_BWD) for the A/B.)
_FFN_FWD_ACT = [kernels.ACT_GELU_D, kernels.ACT_MUL]


class FfnFn(torch.autograd.Function):
    """out = linear_2(gelu(linear_1(xn))) + residual  (modules.py:215-221,278) with dropout off.
    Forward: linear_1's epilogue stores the GELU and its DERIVATIVE (one erf for both) in one pass.  Backward: the
    epilogue of linear_2's dX product multiplies by that derivative — no activation pass, no transcendental there."""

    @staticmethod
    def forward(ctx, xn, w1, b1, w2, b2, residual):
        xn = xn.contiguous()
        m, dff = xn.shape[0], w1.shape[0]
        pre = torch.empty(m, dff, device=xn.device, dtype=torch.float32)
        hid = torch.empty_like(pre)
        kernels.linear_ex(xn, w1.detach(), bias=b1.detach(), out=hid, pre_out=pre, act=_FFN_FWD_ACT[0])   # pre := gelu'
        out = torch.empty(m, w2.shape[0], device=xn.device, dtype=torch.float32)
        kernels.linear_ex(hid, w2.detach(), bias=b2.detach(), residual=residual, out=out)
        ctx.save_for_backward(xn, w1, w2, pre, hid, b1, b2)
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        xn, w1, w2, pre, hid, b1, b2 = ctx.saved_tensors
        dy = dy.contiguous()
        d, dff = w2.shape
        dw2 = optim.grad_out(w2)
        kernels.gemm_tn(dy, hid, out=dw2)                                   # dW2 = dY^T . hid
        db2 = _colsum(dy, b2)
        dpre = _dx(dy, w2, kernels.pad32(d), residual=pre, act=_FFN_FWD_ACT[1])   # (dY . W2) * gelu'(pre)
        dw1 = optim.grad_out(w1)
        kernels.gemm_tn(dpre, xn, out=dw1)
        db1 = _colsum(dpre, b1)
        dxn = _dx(dpre, w1, kernels.pad32(dff)) if ctx.needs_input_grad[0] else None
        return dxn, dw1, db1, dw2, db2, (dy if ctx.has_res else None)


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre):
        pre = pre.contiguous()
        out = torch.empty_like(pre)
        check(_lib.lib().vh_gelu(ptr(pre), None, ptr(out), pre.numel(), stream()), 'vh_gelu')
        ctx.save_for_backward(pre)
        return out

    @staticmethod
    def backward(ctx, dh):
        (pre,) = ctx.saved_tensors
        dh = dh.contiguous()
        out = torch.empty_like(pre)
        check(_lib.lib().vh_gelu(ptr(pre), ptr(dh), ptr(out), pre.numel(), stream()), 'vh_gelu')
        return out


class LayerNormFn(torch.autograd.Function):
    """y = s * (gamma * xhat + beta) + t   (s, t optional: AdaptiveLayerNorm, modules.py:93-99)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, s, t, eps):
        x = x.contiguous()
        out = kernels.layernorm(x, gamma.detach(), beta.detach(),
                                ada_scale=None if s is None else s.detach().contiguous(),
                                ada_shift=None if t is None else t.detach().contiguous(), eps=eps)
        ctx.save_for_backward(x, gamma, beta, s)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, s = ctx.saved_tensors
        dy = dy.contiguous()
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        dg, db = optim.grad_out(gamma, zero=True), optim.grad_out(beta, zero=True)
        ds = dt = None
        if s is not None:
            s = s.contiguous()
            ds, dt = torch.zeros(d, device=x.device), torch.zeros(d, device=x.device)
        check(_lib.lib().vh_layernorm_bwd(ptr(x), ptr(gamma.detach()), ptr(beta.detach()), ptr(s), ptr(dy),
                                          ptr(dx), ptr(dg), ptr(db), ptr(ds), ptr(dt), None, None, None, None, rows, d,
                                          ctx.eps, stream()), 'vh_layernorm_bwd')
        if ds is not None:
            ds, dt = ds.view_as(s), dt.view_as(s)
        return dx, dg, db, ds, dt, None


class QkvAttentionFn(torch.autograd.Function):
    """out (B*T, d) = SDPA(split_heads(x @ Wqkv.T)) with the analytic / explicit masks of
    vh_attn_rows (modules.py:146-170).  Backward recomputes P from the saved q, k, v."""

    @staticmethod
    def forward(ctx, x, wqkv, B, T, n_heads, spec):
        x = x.contiguous()
        d = x.shape[1]
        dev = x.device
        q = torch.empty(B * T, d, device=dev, dtype=torch.float32)
        k = torch.empty(B, n_heads, T, HEAD_DIM, device=dev, dtype=torch.float32)
        v = torch.empty_like(k)
        kernels.linear_qkv(x, wqkv.detach(), q, k, v, B, T, n_heads)
        out = torch.empty(B * T, d, device=dev, dtype=torch.float32)
        lse2 = torch.empty(B, n_heads, T, device=dev, dtype=torch.float32)
        kernels.attn_rows(q, k, v, out, B, n_heads, T, T, lse2=lse2, **spec)
        ctx.save_for_backward(x, wqkv, q, k, v, out, lse2)
        ctx.dims, ctx.spec = (B, T, n_heads), spec
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wqkv, q, k, v, out, lse2 = ctx.saved_tensors
        B, T, h = ctx.dims
        d = h * HEAD_DIM
        spec = ctx.spec
        if ATTENTION_BACKWARD == 'flash':
            dqkv = torch.empty(B * T, 3 * d, device=x.device, dtype=torch.float32)
            kernels.attn_rows_bwd(q, k, v, out, dout.contiguous(), lse2, dqkv[:, :d], dqkv[:, d:2 * d],
                                  dqkv[:, 2 * d:], B, h, T, **spec)
            dx = dw = None
            if ctx.needs_input_grad[0]:
                dx = _dx(dqkv, wqkv, 3 * d)
            if ctx.needs_input_grad[1]:
                dw = optim.grad_out(wqkv)
                kernels.gemm_tn(dqkv, x, out=dw)
            return dx, dw, None, None, None, None
        scale = HEAD_DIM ** -0.5
        L = _lib.lib()
        tp = (T + 3) // 4 * 4                                              # row stride of the (T,T) maps
        qh = q.view(B, T, h, HEAD_DIM).permute(0, 2, 1, 3)               # (B,h,T,64) views, read in place
        do = dout.contiguous().view(B, T, h, HEAD_DIM).permute(0, 2, 1, 3)
        dqkv = torch.empty(B * T, 3 * d, device=x.device, dtype=torch.float32)
        dq, dk, dv = (dqkv.view(B, T, 3, h, HEAD_DIM)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        P = torch.empty(B, h, T, tp, device=x.device, dtype=torch.float32)[..., :T]
        dP = torch.empty(B, h, T, tp, device=x.device, dtype=torch.float32)[..., :T]
        _mm(qh, k, P)                                                     # S = Q K^T (raw scores)
        check(L.vh_softmax_rows(P.data_ptr(), tp, B, h, T, T, scale, spec['mode'], spec.get('x_len', 0),
                                ptr(spec.get('x_len_dev')), ptr(spec.get('kv_len')), ptr(spec.get('mask')),
                                ptr(spec.get('pad')), stream()), 'vh_softmax_rows')
        _mm(P, do, dv, a_kmajor=True, b_kmajor=True)                      # dV = P^T dO   → dqkv[:, 2d:]
        _mm(do, v, dP)                                                    # dP = dO V^T
        check(L.vh_softmax_bwd(P.data_ptr(), dP.data_ptr(), tp, B * h * T, T, scale, stream()), 'vh_softmax_bwd')
        _mm(dP, k, dq, b_kmajor=True)                                     # dQ = dS K     → dqkv[:, :d]
        _mm(dP, qh, dk, a_kmajor=True, b_kmajor=True)                     # dK = dS^T Q   → dqkv[:, d:2d]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _dx(dqkv, wqkv, 3 * d)
        if ctx.needs_input_grad[1]:
            dw = optim.grad_out(wqkv)
            kernels.gemm_tn(dqkv, x, out=dw)
        return dx, dw, None, None, None, None


class QkvAttentionAnyHeadDimFn(torch.autograd.Function):
    """QkvAttentionFn for a head width other than 64 (modules.py:109-111 allows any divisor of d_model): the whole node
    on the general kernels with the probabilities materialised — qkv = x Wqkv^T, P = softmax(Q K^T / sqrt(hd) + mask),
    out = P V forward; dV = P^T dO, dP = dO V^T, dS, dQ = dS K, dK = dS^T Q backward — every operand a strided view of
    the (rows, 3 d) projection or its gradient, read and written in place.  Correct, not tuned (P is kept: O(T^2) memory
    per (batch row, head))."""

    @staticmethod
    def forward(ctx, x, wqkv, B, T, n_heads, spec):
        x = x.contiguous()
        d = x.shape[1]
        hd = d // n_heads
        if hd % 4:
            raise _lib.VhError(f'head_dim {hd}: the general attention path needs a multiple of 4')
        qkv = kernels.linear(x, wqkv.detach(), out=torch.empty(B * T, 3 * d, device=x.device, dtype=torch.float32))
        q, k, v = (qkv.view(B, T, 3, n_heads, hd)[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        out = torch.empty(B * T, d, device=x.device, dtype=torch.float32)
        P = kernels.attn_generic(q, k, v, out.view(B, T, n_heads, hd).permute(0, 2, 1, 3), hd ** -0.5, **spec)
        ctx.save_for_backward(x, wqkv, qkv, P)
        ctx.dims = (B, T, n_heads, hd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wqkv, qkv, P = ctx.saved_tensors
        B, T, h, hd = ctx.dims
        d = h * hd
        q, k, v = (qkv.view(B, T, 3, h, hd)[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        do = dout.contiguous().view(B, T, h, hd).permute(0, 2, 1, 3)
        dqkv = torch.empty(B * T, 3 * d, device=x.device, dtype=torch.float32)
        dq, dk, dv = (dqkv.view(B, T, 3, h, hd)[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        tp = P.stride(2)
        dP = torch.empty(B, h, T, tp, device=x.device, dtype=torch.float32)[..., :T]
        _mm(P, do, dv, a_kmajor=True, b_kmajor=True)                      # dV = P^T dO
        _mm(do, v, dP)                                                    # dP = dO V^T
        check(_lib.lib().vh_softmax_bwd(P.data_ptr(), dP.data_ptr(), tp, B * h * T, T, hd ** -0.5, stream()), 'vh_softmax_bwd')
        _mm(dP, k, dq, b_kmajor=True)                                     # dQ = dS K
        _mm(dP, q, dk, a_kmajor=True, b_kmajor=True)                      # dK = dS^T Q
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _dx(dqkv, wqkv, 3 * d)
        if ctx.needs_input_grad[1]:
            dw = optim.grad_out(wqkv)
            kernels.gemm_tn(dqkv, x, out=dw)
        return dx, dw, None, None, None, None


def _drop_ref(sp):
    return None if sp is None else C.byref(sp)


class EmbedSumPeFn(torch.autograd.Function):
    """x[:, t0:t0+T] = dropout(sum_j tables[j][ids[..., j]] + pe[pos0:pos0+T])  written into a fresh (B,T,d);
    drop: a dropout.spec or None (modules.py:35,80)."""

    @staticmethod
    def forward(ctx, ids, pe, pos0, drop, *tables):
        if ids.dim() == 2:
            ids = ids.unsqueeze(-1)
        B, T, _ = ids.shape
        d = tables[0].shape[1]
        out = torch.empty(B, T, d, device=ids.device, dtype=torch.float32)
        kernels.embed_sum_pe(ids, [t.detach() for t in tables], pe, pos0, out, drop=drop)
        ctx.ids, ctx.shapes, ctx.tables, ctx.drop = ids, [t.shape for t in tables], tables, drop
        return out

    @staticmethod
    def backward(ctx, dout):
        ids = ctx.ids
        dout = dout.contiguous()
        B, T, d = dout.shape
        grads = []
        for j, shp in enumerate(ctx.shapes):
            if not ctx.needs_input_grad[4 + j]:
                grads.append(None)
                continue
            g = optim.grad_out(ctx.tables[j], zero=True)
            col = ids[..., j]
            check(_lib.lib().vh_embed_bwd(col.data_ptr(), col.stride(0), col.stride(1), ptr(dout),
                                          dout.stride(0), 0, ptr(g), int(shp[0]), B, T, d,
                                          ptr(_lib.err_flag(dout.device)), _drop_ref(ctx.drop), stream()), 'vh_embed_bwd')
            grads.append(g)
        return (None, None, None, None, *grads)


class EmbedConcatFn(torch.autograd.Function):
    """The model input x (B, sum T_i, d) of a training step in one buffer: part i = sum_j tables[idx][ids_i[..., j]] +
    pe_i[pos0_i : pos0_i + T_i] is written at its row offset (what EmbedSumPeFn per part + torch.cat made with a copy
    forward and two strided copies backward), and every table receives ONE gradient however many parts read it.
    parts: [(ids (B, T_i) | (B, T_i, J_i), pe, pos0, [indices into `tables`, one per codebook column], drop)] —
    drop: the dropout.spec of the part's PositionalEncoding dropout (modules.py:80) or None; its field is indexed by the
    row of the joint buffer, so parts never share an element whatever their specs."""

    @staticmethod
    def forward(ctx, parts, *tables):
        B = parts[0][0].shape[0]
        d = tables[0].shape[1]
        total = sum(p[0].shape[1] for p in parts)
        out = torch.empty(B, total, d, device=tables[0].device, dtype=torch.float32)
        t0, laid = 0, []
        for ids, pe, pos0, idx, drop in parts:
            if ids.dim() == 2:
                ids = ids.unsqueeze(-1)
            kernels.embed_sum_pe(ids, [tables[j].detach() for j in idx], pe, pos0, out, out_t0=t0, drop=drop)
            laid.append((ids, t0, idx, drop))
            t0 += ids.shape[1]
        ctx.laid, ctx.tables = laid, tables
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        B, _, d = dout.shape
        grads = [optim.grad_out(t, zero=True) if ctx.needs_input_grad[1 + j] else None for j, t in enumerate(ctx.tables)]
        for ids, t0, idx, drop in ctx.laid:
            for col_j, j in enumerate(idx):
                if grads[j] is None:
                    continue
                col = ids[..., col_j]
                check(_lib.lib().vh_embed_bwd(col.data_ptr(), col.stride(0), col.stride(1), ptr(dout), dout.stride(0), t0,
                                              ptr(grads[j]), int(ctx.tables[j].shape[0]), B, ids.shape[1], d,
                                              ptr(_lib.err_flag(dout.device)), _drop_ref(drop), stream()), 'vh_embed_bwd')
        return (None, *grads)


class CrossEntropyFn(torch.autograd.Function):
    """Mean cross entropy over every row of (R, V) logits against (R,) int64 targets."""

    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous() if logits.stride(1) != 1 else logits
        R, V = logits.shape
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dl = torch.empty(R, V, device=logits.device, dtype=torch.float32)
        target = target.contiguous()
        check(_lib.lib().vh_cross_entropy(logits.data_ptr(), logits.stride(0), V, ptr(target), ptr(loss),
                                          ptr(dl), V, R, ptr(_lib.err_flag(logits.device)), stream()),
              'vh_cross_entropy')
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def _ln_bwd(x, gamma, beta, s, dy, dres, dcol, eps, dst=None, drop=None):
    """LayerNorm / AdaLN backward with the residual-branch gradient added in the same pass and (optionally) the column
    sums of the result accumulated into `dcol`.  `dst` (2, d), zero on entry: where the AdaLN scale / shift gradients
    are accumulated.  drop (a dropout.spec): x was residual + dropout(branch) — the branch's gradient, dx under that
    dropout's field, is written as well and `dcol` sums IT.  Returns (dx, dgamma, dbeta, dx_dropped | None)."""
    d = x.shape[-1]
    dx = torch.empty_like(x)
    dxd = torch.empty_like(x) if drop is not None else None
    dg, db = optim.grad_out(gamma, zero=True), optim.grad_out(beta, zero=True)
    ds = dt = None
    if s is not None:
        ds, dt = dst[0], dst[1]
    check(_lib.lib().vh_layernorm_bwd(ptr(x), ptr(gamma.detach()), ptr(beta.detach()), ptr(s), ptr(dy), ptr(dx),
                                      ptr(dg), ptr(db), ptr(ds), ptr(dt), ptr(dres), ptr(dcol), ptr(dxd), _drop_ref(drop),
                                      x.numel() // d, d, eps, stream()), 'vh_layernorm_bwd')
    return dx, dg, db, dxd


class AdaProjFn(torch.autograd.Function):
    """(n, 2d) = [emb @ W_i.T + b_i for the n AdaptiveLayerNorm project_layer Linears of a stack] in one launch, and
    their whole backward in one launch (outer-product weight gradients, vh_adaproj_fwd / _bwd)."""

    @staticmethod
    def _items(ws, bs, dws, dbs, dev):
        import numpy as np
        rec = np.zeros(len(ws), dtype=[('w', 'u8'), ('b', 'u8'), ('dw', 'u8'), ('db', 'u8')])
        for i, (w, b) in enumerate(zip(ws, bs)):
            rec[i] = (w.data_ptr(), b.data_ptr(), dws[i].data_ptr() if dws else 0, dbs[i].data_ptr() if dbs else 0)
        return _lib.to_device_async(torch.from_numpy(rec.view(np.uint8).copy()), dev)

    @staticmethod
    def forward(ctx, emb, *params):
        ws, bs = params[0::2], params[1::2]
        n, (N, K) = len(ws), ws[0].shape
        if any(tuple(w.shape) != (N, K) or not w.is_contiguous() for w in ws) or any(b.numel() != N for b in bs):
            raise _lib.VhError('AdaProjFn: equal-shaped contiguous project_layer weights')
        emb = emb.detach().reshape(-1).contiguous()
        out = torch.empty(n, N, device=emb.device, dtype=torch.float32)
        items = AdaProjFn._items([w.detach() for w in ws], [b.detach() for b in bs], None, None, emb.device)
        check(_lib.lib().vh_adaproj_fwd(ptr(items), n, ptr(emb), ptr(out), N, K, stream()), 'vh_adaproj_fwd')
        ctx.save_for_backward(emb, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        emb, *params = ctx.saved_tensors
        ws, bs = params[0::2], params[1::2]
        n, (N, K) = len(ws), ws[0].shape
        dout = dout.contiguous()
        dws = [optim.grad_out(w) for w in ws]
        dbs = [optim.grad_out(b) for b in bs]
        items = AdaProjFn._items([w.detach() for w in ws], [b.detach() for b in bs], dws, dbs, emb.device)
        demb = torch.zeros(K, device=emb.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        check(_lib.lib().vh_adaproj_bwd(ptr(items), n, ptr(emb), ptr(dout), ptr(demb), N, K, stream()), 'vh_adaproj_bwd')
        grads = [g for pair in zip(dws, dbs) for g in pair]
        return (None if demb is None else demb.view(1, K), *grads)


class _StackPass:
    """What the layers of ONE transformer_train call hand each other during its backward (a fresh object per forward,
    carried in every layer's `meta`: nothing outlives the pass it belongs to).
      bias_ahead[l]  layer l+1's norm1 backward writes dx — the gradient of layer l's output — and accumulates its column
                     sums, which ARE the gradient of layer l's linear_2 bias, into that bias' gradient slot; layer l's
                     backward picks the tensor up here instead of launching a column-sum kernel.
      dy_drop[l]     (dx, dx under layer l's dropout2 field or None, dx's version counter) from the same launch: layer l's
                     backward uses the dropped copy and the column sums above only when the gradient it receives IS that
                     dx, unmodified (same storage, same version); otherwise it recomputes both."""

    def __init__(self, drops):
        self.drops = drops                 # dropout.StackDropout
        self.bias_ahead = {}
        self.dy_drop = {}


class EncoderLayerFn(torch.autograd.Function):
    """One pre-norm block (modules.py:240-280) as ONE autograd node, dropout included:
        xm = x + drop1(out(attn(qkv(norm1(x)))));  y = xm + drop2(linear_2(dropf(gelu(linear_1(norm2(xm)))))).
    The three dropouts are fields regenerated in the epilogues (dropout.py): drop1 / drop2 inside the out-projection's /
    linear_2's bias + residual epilogue, dropf inside linear_1's GELU epilogue, which also multiplies the stored GELU
    derivative by the field — so the backward's (dY . W2) * saved epilogue needs nothing new.  The gradients of the two
    residual branches are the dropped copies the LayerNorm backward launches write next to the residual stream's.
    The forward is the kernel sequence of the separate Functions; the backward runs the whole block in one Python
    call with the residual-gradient adds folded into the two LayerNorm backward launches, the out-projection's bias
    gradient (and the linear_2 bias gradient of the layer BELOW) taken from those launches' column sums, and the
    dX products reading weights transposed once per optimizer step for the whole stack (kernels.TransposePlan)."""

    @staticmethod
    def forward(ctx, x, meta, wqkv, wo, bo, g1, be1, g2, be2, w1, b1, w2, b2, ada_all):
        # AdaLN: meta carries this layer's (scale, shift) pairs as views of the stack's projections (ada_all, made by
        # AdaProjFn) and the slices of the ONE gradient buffer every layer accumulates into; only the first layer
        # receives ada_all as a differentiable input and hands that buffer back as its gradient (it runs last)
        B, T, n_heads, spec, eps, wts, below_b2, ada, ada_grad, sp, idx = meta
        s1, t1, s2, t2 = ada if ada is not None else (None, None, None, None)
        dr1, drf, dr2 = sp.drops.layer(idx)
        x = x.contiguous()
        dev, d = x.device, x.shape[1]
        det = lambda p: None if p is None else p.detach().contiguous()   # noqa: E731
        xn1 = kernels.layernorm(x, g1.detach(), be1.detach(), ada_scale=det(s1), ada_shift=det(t1), eps=eps)
        q = torch.empty(B * T, d, device=dev, dtype=torch.float32)
        k = torch.empty(B, n_heads, T, HEAD_DIM, device=dev, dtype=torch.float32)
        v = torch.empty_like(k)
        kernels.linear_qkv(xn1, wqkv.detach(), q, k, v, B, T, n_heads)
        a = torch.empty(B * T, d, device=dev, dtype=torch.float32)
        lse2 = torch.empty(B, n_heads, T, device=dev, dtype=torch.float32)
        kernels.attn_rows(q, k, v, a, B, n_heads, T, T, lse2=lse2, **spec)
        xm = torch.empty_like(x)
        kernels.linear_ex(a, wo.detach(), bias=bo.detach(), residual=x, out=xm, drop=dr1)    # (tile kernel with the tail split)
        xn2 = kernels.layernorm(xm, g2.detach(), be2.detach(), ada_scale=det(s2), ada_shift=det(t2), eps=eps)
        pre = torch.empty(B * T, w1.shape[0], device=dev, dtype=torch.float32)
        hid = torch.empty_like(pre)
        if drf is not None and _FFN_FWD_ACT[0] != kernels.ACT_GELU_D:
            raise _lib.VhError('FeedForward dropout rides on the (ACT_GELU_D, ACT_MUL) activation pair')
        kernels.linear_ex(xn2, w1.detach(), bias=b1.detach(), out=hid, pre_out=pre, act=_FFN_FWD_ACT[0], drop=drf)  # pre := gelu' (x field)
        y = torch.empty_like(x)
        kernels.linear_ex(hid, w2.detach(), bias=b2.detach(), residual=xm, out=y, drop=dr2)
        M = B * T
        dropout.record(f'layer{idx}.dropout1', dr1, M, d)
        dropout.record(f'layer{idx}.ffn.dropout', drf, M, w1.shape[0])
        dropout.record(f'layer{idx}.dropout2', dr2, M, d)
        ctx.save_for_backward(x, xn1, q, k, v, a, lse2, xm, xn2, pre, hid, wqkv, wo, bo, g1, be1, g2, be2, w1, b1, w2, b2)
        ctx.meta = meta
        ctx.returns_ada = ada_all is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xn1, q, k, v, a, lse2, xm, xn2, pre, hid, wqkv, wo, bo, g1, be1, g2, be2, w1, b1, w2, b2 = ctx.saved_tensors
        B, T, h, spec, eps, wts, below_b2, ada, ada_grad, sp, idx = ctx.meta
        s1, _, s2, _ = ada if ada is not None else (None, None, None, None)
        dst1, dst2, ada_grads = ada_grad if ada_grad is not None else (None, None, None)
        dr1, _, dr2 = sp.drops.layer(idx)
        below_dr2 = sp.drops.layer(idx - 1)[2] if idx > 0 else None
        d = x.shape[1]
        dy = dy.contiguous()
        wqkv_t, wo_t, w1_t, w2_t = wts
        det = lambda p: None if p is None else p.detach().contiguous()   # noqa: E731
        # ---- FeedForward: its branch sits under dropout2, so its products read dY under that field (written by the layer
        # above's norm1 backward next to dY itself; the top layer — or a dY autograd re-made — gets it from one launch here)
        ahead_dy = sp.dy_drop.pop(idx, None)
        db2 = sp.bias_ahead.pop(idx, None)                                 # the layer above summed (dropped) dY's columns already
        # the hand-over holds only while the gradient received IS the dx the layer above wrote, UNCHANGED: same storage
        # and same version counter — a tensor hook that scales or clips the layer output's gradient in place keeps the
        # pointer but bumps the version (round-4 advisor finding), one that returns a new tensor changes the pointer
        same = ahead_dy is not None and ahead_dy[0].data_ptr() == dy.data_ptr() and ahead_dy[0]._version == ahead_dy[2]
        if ahead_dy is not None and not same:
            db2 = None                                                     # (recomputed below, overwriting the stale sums)
        dyd = dy
        if dr2 is not None:
            if same and ahead_dy[1] is not None:
                dyd = ahead_dy[1]
            else:
                dyd = dropout.apply_raw(dy, dr2)
                db2 = None
        dw2 = optim.grad_out(w2)
        kernels.gemm_tn(dyd, hid, out=dw2)                                 # dW2 = dY^T . hid
        if db2 is None:
            db2 = _colsum(dyd, b2)
        dpre = torch.empty_like(pre)                                       # (dY . W2) * gelu'(pre), + its column sums = db1
        fold = pre.shape[1] % 128 == 0                                     # (whole 128-column tiles)
        db1 = optim.grad_out(b1, zero=True) if fold else None
        kernels.linear_ex(dyd, w2_t, residual=pre, out=dpre, act=_FFN_FWD_ACT[1], K=w2_t.shape[1], colsum=db1)
        if not fold:
            db1 = _colsum(dpre, b1)
        dw1 = optim.grad_out(w1)
        kernels.gemm_tn(dpre, xn2, out=dw1)
        dxn2 = torch.empty_like(x)
        kernels.linear_ex(dpre, w1_t, out=dxn2, K=w1_t.shape[1])
        # ---- norm2 (+ the residual branch's dY, + the out-projection's bias gradient = column sums of the result)
        dbo = optim.grad_out(bo, zero=True)
        dxm, dg2, dbe2, dxm_d = _ln_bwd(xm, g2, be2, det(s2), dxn2, dy, dbo, eps, dst2, drop=dr1)
        # ---- out-projection (under dropout1: the dropped copy of the same launch)
        dbr = dxm if dxm_d is None else dxm_d
        dwo = optim.grad_out(wo)
        kernels.gemm_tn(dbr, a, out=dwo)
        da = torch.empty_like(x)
        kernels.linear_ex(dbr, wo_t, out=da, K=wo_t.shape[1])
        # ---- attention core + QKV projection
        dqkv = torch.empty(B * T, 3 * d, device=x.device, dtype=torch.float32)
        kernels.attn_rows_bwd(q, k, v, a, da, lse2, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, T, **spec)
        dwqkv = optim.grad_out(wqkv)
        kernels.gemm_tn(dqkv, xn1, out=dwqkv)
        dxn1 = torch.empty_like(x)
        kernels.linear_ex(dqkv, wqkv_t, out=dxn1, K=wqkv_t.shape[1])
        # ---- norm1 (+ the residual branch's gradient; + the bias gradient of the layer below's linear_2)
        ahead = optim.grad_out(below_b2, zero=True) if below_b2 is not None else None
        dx, dg1, dbe1, dx_d = _ln_bwd(x, g1, be1, det(s1), dxn1, dxm, ahead, eps, dst1, drop=below_dr2)
        if ahead is not None:
            sp.bias_ahead[idx - 1] = ahead
        if ahead is not None or dx_d is not None:
            sp.dy_drop[idx - 1] = (dx, dx_d, dx._version)              # (dx, its dropped copy or None, dx's version now)
        dada = ada_grads if ctx.returns_ada else None            # the whole stack's buffer, complete once layer 0 is done
        return (dx, None, dwqkv, dwo, dbo, dg1, dbe1, dg2, dbe2, dw1, db1, dw2, db2, dada)


def linear(x, w, b=None, residual=None):
    return LinearFn.apply(x, w, b, residual)


def layer_norm(x, gamma, beta, s=None, t=None, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, s, t, eps)


def encoder_layer_train(layer, x, B, T, spec, embedding=None):
    """One pre-norm block on x (B*T, d) with a full autograd graph of separate nodes (modules.py:240-280): the form the
    shapes the fused node does not serve take (and the 'materialized' attention backward).  Dropout (training mode,
    p > 0): free-standing fields through vh_dropout (dropout.apply)."""
    cfg = layer.config
    at, ff = layer.self_attn, layer.ffn

    def norm(n, inp):
        if cfg.norm == 'LayerNorm':
            return layer_norm(inp, n.weight, n.bias, eps=n.eps)
        wb = linear(embedding.reshape(1, -1), n.project_layer.weight, n.project_layer.bias).view(2, -1)
        return layer_norm(inp, n.norm.weight, n.norm.bias, wb[0], wb[1], eps=n.eps)

    attention = QkvAttentionFn if at.head_dim == HEAD_DIM else QkvAttentionAnyHeadDimFn
    a = attention.apply(norm(layer.norm1, x), at.qkv.weight, B, T, at.n_heads, spec)
    d1, d2 = layer.dropout1, layer.dropout2
    if d1.training and d1.p > 0:
        x = x + dropout.apply(d1, linear(a, at.out.weight, at.out.bias))
    else:
        x = linear(a, at.out.weight, at.out.bias, residual=x)
    ffn_drop = ff.dropout.training and ff.dropout.p > 0
    res_drop = d2.training and d2.p > 0
    if not ffn_drop and not res_drop and cfg.dim_feedforward % 32 == 0:
        return FfnFn.apply(norm(layer.norm2, x), ff.linear_1.weight, ff.linear_1.bias, ff.linear_2.weight,
                           ff.linear_2.bias, x)
    hid = GeluFn.apply(linear(norm(layer.norm2, x), ff.linear_1.weight, ff.linear_1.bias))
    if ffn_drop:
        hid = dropout.apply(ff.dropout, hid)
    if res_drop:
        x = x + dropout.apply(d2, linear(hid, ff.linear_2.weight, ff.linear_2.bias))
    else:
        x = linear(hid, ff.linear_2.weight, ff.linear_2.bias, residual=x)
    return x


def _stack_transposes(transformer):
    """Wᵀ of every Linear of the stack for the backward's dX products, refreshed by one launch whenever a weight has
    changed since the last refresh (an optimizer step, a load): [(Wqkvᵀ, Woᵀ, W1ᵀ, W2ᵀ)] per layer."""
    from . import engine
    layers = list(transformer.layers)
    ws = [w for l in layers for w in (l.self_attn.qkv.weight, l.self_attn.out.weight, l.ffn.linear_1.weight,
                                      l.ffn.linear_2.weight)]
    ptrs = tuple(w.data_ptr() for w in ws)
    state = getattr(transformer, '_vh_wt', None)
    if state is None or state[0].key != ptrs:
        plan = kernels.TransposePlan([w.detach() for w in ws], [kernels.pad32(w.shape[0]) for w in ws])
        state = [plan, None]
        transformer._vh_wt = state
    version = (engine._WEIGHTS_EPOCH,) + tuple(w._version for w in ws)
    if state[1] != version:
        state[0].run()
        state[1] = version
    outs = state[0].outs
    return [tuple(outs[4 * i:4 * i + 4]) for i in range(len(layers))]


# test hook: a callable (layer index, grad) -> grad | None registered as a tensor hook on every layer's output of the fused
# stack (what a user's gradient-clipping / scaling hook on an intermediate activation would be)
LAYER_OUTPUT_HOOK = None


def transformer_train(transformer, x, B, T, spec, embedding=None):
    layers = list(transformer.layers)
    cfg = transformer.hparams
    fused = (ATTENTION_BACKWARD == 'flash' and cfg.dim_feedforward % 32 == 0 and cfg.d_model % 32 == 0 and
             cfg.d_model == cfg.n_heads * HEAD_DIM)      # (another head width: node by node on the general kernels)
    if not fused:
        for layer in layers:
            x = encoder_layer_train(layer, x, B, T, spec, embedding)
        return x
    sp = _StackPass(dropout.StackDropout(layers))           # one seed draw per forward; the fields of every layer
    wts = _stack_transposes(transformer)
    ada_all = ada_vals = ada_grads = None
    d = cfg.d_model
    if cfg.norm != 'LayerNorm':
        # every AdaLN (scale, shift) of the stack from ONE launch; their gradients meet in one zeroed buffer that the
        # first layer's node hands back to autograd
        projs = [p for l in layers for n in (l.norm1, l.norm2) for p in (n.project_layer.weight, n.project_layer.bias)]
        ada_all = AdaProjFn.apply(embedding, *projs)                          # (2 L, 2 d)
        ada_vals = ada_all.detach()
        ada_grads = torch.zeros_like(ada_vals)
    for i, layer in enumerate(layers):
        at, ff = layer.self_attn, layer.ffn
        below_b2 = layers[i - 1].ffn.linear_2.bias if i > 0 else None
        if ada_all is None:
            n1, n2, ada, ada_grad = layer.norm1, layer.norm2, None, None
        else:
            n1, n2 = layer.norm1.norm, layer.norm2.norm
            av, gv = ada_vals[2 * i:2 * i + 2].view(2, 2, d), ada_grads[2 * i:2 * i + 2].view(2, 2, d)
            ada = (av[0, 0], av[0, 1], av[1, 0], av[1, 1])              # (scale, shift) of norm1, of norm2
            ada_grad = (gv[0], gv[1], ada_grads)
        meta = (B, T, at.n_heads, spec, layer.norm1.eps, wts[i], below_b2, ada, ada_grad, sp, i)
        x = EncoderLayerFn.apply(x, meta, at.qkv.weight, at.out.weight, at.out.bias, n1.weight, n1.bias, n2.weight,
                                 n2.bias, ff.linear_1.weight, ff.linear_1.bias, ff.linear_2.weight, ff.linear_2.bias,
                                 ada_all if i == 0 else None)
        if LAYER_OUTPUT_HOOK is not None and x.requires_grad:
            x.register_hook(lambda g, i=i: LAYER_OUTPUT_HOOK(i, g))
    return x
