"""Autograd wrappers: the training path (`training_step` → `loss.backward()`, valle_ar.py:43-90).

Forward passes run the same HIP kernels as inference.  Backward passes are hand-written HIP throughout:
  * dX = dY · W          → the NT LDS-DMA tile kernel (`vh_linear_ex`) on Wᵀ (`vh_transpose`, a few MB per layer),
                           with the GELU backward fused into linear_2's product (`VH_ACT_GELU_BWD`);
  * dW = dYᵀ · X         → `vh_gemm_tn`: both operands read token-major as they lie, contraction split over the
                           chip, slabs summed in fixed order (bitwise reproducible);
  * attention            → `vh_attn_rows_bwd` (flash-style, P recomputed tile by tile);
  * everything row-wise  → csrc/train.hip (LayerNorm/AdaLN, cross entropy, embedding scatter, bias column sums).
There is no library GEMM and no CPU arithmetic on this path.
"""
from __future__ import annotations

import torch

from . import _lib, kernels, optim
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


class FfnFn(torch.autograd.Function):
    """out = linear_2(gelu(linear_1(xn))) + residual  (modules.py:215-221,278) with dropout off.
    Forward: linear_1's epilogue stores the pre-activation and the GELU in one pass.  Backward: the GELU
    derivative is applied in the epilogue of linear_2's dX product — no separate activation passes."""

    @staticmethod
    def forward(ctx, xn, w1, b1, w2, b2, residual):
        xn = xn.contiguous()
        m, dff = xn.shape[0], w1.shape[0]
        pre = torch.empty(m, dff, device=xn.device, dtype=torch.float32)
        hid = torch.empty_like(pre)
        kernels.linear_ex(xn, w1.detach(), bias=b1.detach(), out=hid, pre_out=pre, act=kernels.ACT_GELU)
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
        dpre = _dx(dy, w2, kernels.pad32(d), residual=pre, act=kernels.ACT_GELU_BWD)   # (dY . W2) * gelu'(pre)
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
                                          ptr(dx), ptr(dg), ptr(db), ptr(ds), ptr(dt), rows, d, ctx.eps,
                                          stream()), 'vh_layernorm_bwd')
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


class EmbedSumPeFn(torch.autograd.Function):
    """x[:, t0:t0+T] = sum_j tables[j][ids[..., j]] + pe[pos0:pos0+T]  written into a fresh (B,T,d)."""

    @staticmethod
    def forward(ctx, ids, pe, pos0, *tables):
        if ids.dim() == 2:
            ids = ids.unsqueeze(-1)
        B, T, _ = ids.shape
        d = tables[0].shape[1]
        out = torch.empty(B, T, d, device=ids.device, dtype=torch.float32)
        kernels.embed_sum_pe(ids, [t.detach() for t in tables], pe, pos0, out)
        ctx.ids, ctx.shapes, ctx.tables = ids, [t.shape for t in tables], tables
        return out

    @staticmethod
    def backward(ctx, dout):
        ids = ctx.ids
        dout = dout.contiguous()
        B, T, d = dout.shape
        grads = []
        for j, shp in enumerate(ctx.shapes):
            if not ctx.needs_input_grad[3 + j]:
                grads.append(None)
                continue
            g = optim.grad_out(ctx.tables[j], zero=True)
            col = ids[..., j]
            check(_lib.lib().vh_embed_bwd(col.data_ptr(), col.stride(0), col.stride(1), ptr(dout),
                                          dout.stride(0), 0, ptr(g), int(shp[0]), B, T, d,
                                          ptr(_lib.err_flag(dout.device)), stream()), 'vh_embed_bwd')
            grads.append(g)
        return (None, None, None, *grads)


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


def linear(x, w, b=None, residual=None):
    return LinearFn.apply(x, w, b, residual)


def layer_norm(x, gamma, beta, s=None, t=None, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, s, t, eps)


def encoder_layer_train(layer, x, B, T, spec, embedding=None):
    """One pre-norm block on x (B*T, d) with a full autograd graph (modules.py:240-280).
    Dropout (training mode, p > 0) uses torch's device RNG, as the reference does."""
    cfg = layer.config
    at, ff = layer.self_attn, layer.ffn

    def norm(n, inp):
        if cfg.norm == 'LayerNorm':
            return layer_norm(inp, n.weight, n.bias, eps=n.eps)
        wb = linear(embedding.reshape(1, -1), n.project_layer.weight, n.project_layer.bias).view(2, -1)
        return layer_norm(inp, n.norm.weight, n.norm.bias, wb[0], wb[1], eps=n.eps)

    a = QkvAttentionFn.apply(norm(layer.norm1, x), at.qkv.weight, B, T, at.n_heads, spec)
    d1, d2 = layer.dropout1, layer.dropout2
    if d1.training and d1.p > 0:
        x = x + d1(linear(a, at.out.weight, at.out.bias))
    else:
        x = linear(a, at.out.weight, at.out.bias, residual=x)
    ffn_drop = ff.dropout.training and ff.dropout.p > 0
    res_drop = d2.training and d2.p > 0
    if not ffn_drop and not res_drop and cfg.dim_feedforward % 32 == 0:
        return FfnFn.apply(norm(layer.norm2, x), ff.linear_1.weight, ff.linear_1.bias, ff.linear_2.weight,
                           ff.linear_2.bias, x)
    hid = GeluFn.apply(linear(norm(layer.norm2, x), ff.linear_1.weight, ff.linear_1.bias))
    if ffn_drop:
        hid = ff.dropout(hid)
    if res_drop:
        x = x + d2(linear(hid, ff.linear_2.weight, ff.linear_2.bias))
    else:
        x = linear(hid, ff.linear_2.weight, ff.linear_2.bias, residual=x)
    return x


def transformer_train(transformer, x, B, T, spec, embedding=None):
    for layer in transformer.layers:
        x = encoder_layer_train(layer, x, B, T, spec, embedding)
    return x
