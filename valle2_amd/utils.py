"""Mask builders, sampling and beam selection with the reference signatures
(valle/models/utils.py:8-88).  The mask builders are host-side integer/bool plumbing and work on
any device (the reference's own tests call them with device='cpu'); the tensors they return also
carry their defining lengths as attributes so the attention kernels evaluate them analytically.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


def build_pad_mask(lens: Tensor, device) -> Tensor:
    """(B, max_len) bool, True = padded (valle/models/utils.py:8-14)."""
    max_len = int(lens.max().item())
    steps = torch.arange(max_len, device=device)
    mask = steps.unsqueeze(0) >= lens.to(device).unsqueeze(1)
    mask._vh_lens_host = lens.detach().to('cpu', torch.int64)
    return mask


def build_attn_mask(x_len: int, y_len: int, device) -> Tensor:
    """Prefix-LM mask (x_len+y_len)^2 bool, True = masked (valle/models/utils.py:17-43): text rows
    see text only; audio rows see all text and audio up to themselves."""
    n = x_len + y_len
    row = torch.arange(n, device=device).unsqueeze(1)
    col = torch.arange(n, device=device).unsqueeze(0)
    mask = (col >= x_len) & ((row < x_len) | (col > row))
    mask._vh_prefix = (int(x_len), int(y_len))
    return mask


def pad_lens_for_kernel(padding_mask, x_pad: int, device):
    """kv_len (B) int32 on `device` for a padding mask built by build_pad_mask and left-padded
    with `x_pad` unmasked text columns; None when the mask carries no length tag."""
    lens = getattr(padding_mask, '_vh_lens_host', None)
    if lens is None:
        return None
    return (lens + x_pad).to(device=device, dtype=torch.int32)


def topk_sampling(logits: Tensor, top_k: int = 50, tok_p: float = 1.0, temperature: float | None = 1.0):
    """valle/models/utils.py:46-68 → (token (B,1) int64, logprob (B,)).

    On-device implementation of the greedy case only (top_k == 1: the filtered distribution is a
    one-hot, so the sample is the arg-max with the lowest index on ties and its log-prob is 0).
    Stochastic top-k/top-p sampling is a listed next row and raises."""
    if not logits.is_cuda:
        raise _lib.VhError('topk_sampling: logits must be on a HIP device (no CPU fallback)')
    if top_k != 1:
        raise NotImplementedError('valle2_amd: only greedy sampling (top_k=1) runs on device yet')
    if temperature is not None and temperature <= 0:
        raise ValueError('temperature must be positive')
    token = torch.argmax(logits, dim=-1, keepdim=True)
    return token, torch.zeros(logits.shape[0], device=logits.device, dtype=logits.dtype)


def get_best_beam(x, sum_logprobs, stop_token, length_penalty=1.0):
    """valle/models/utils.py:71-88 — the beam with the best length-normalised log-probability,
    stop tokens stripped.  O(B*S) integer bookkeeping after the decode loop."""
    length = torch.sum(x != stop_token, dim=-1)
    avg = sum_logprobs / length**length_penalty
    best = x[torch.argmax(avg), :]
    return best[best != stop_token]
