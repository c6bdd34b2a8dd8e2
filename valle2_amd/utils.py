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


def topk_sampling(logits: Tensor, top_k: int = 50, tok_p: float = 1.0, temperature: float | None = 1.0,
                  seed: int | None = None):
    """valle/models/utils.py:46-68 → (token (B,1) int64, logprob (B,)) on the device: temperature,
    top-k (ties kept), top-p, multinomial draw and the log-prob of the draw in one kernel
    (`vh_sample_step`; `top_k == 1` is the arg-max with the lowest index on ties, log-prob 0).
    The random stream is the kernel's counter-based generator seeded from torch's RNG (or `seed`),
    not torch.multinomial's, so agreement with the reference is distributional."""
    from . import kernels
    home = logits.device
    if not logits.is_cuda:
        # host logits hop to the HIP device and the draw comes back, as every module does for host inputs
        # (modules._on_device); without a device this raises VhError — there is no CPU sampler
        _lib.lib()
        logits = logits.to(torch.device('cuda', torch.cuda.current_device()))
    B, V = logits.shape
    dev = logits.device
    temperature = 1.0 if temperature is None else float(temperature)
    lg = logits.float().contiguous()
    codes = torch.zeros(B, 2, device=dev, dtype=torch.int64)
    state = dict(eos_count=torch.zeros(4, device=dev, dtype=torch.int32),
                 audio_pos=torch.ones(B, device=dev, dtype=torch.int32),
                 cache_len=torch.zeros(B, device=dev, dtype=torch.int32))
    emb, pe = torch.zeros(V + 1, 4, device=dev), torch.zeros(2, 4, device=dev)
    x = torch.empty(B, 4, device=dev)
    lp = torch.zeros(B, device=dev)
    if top_k == 1:
        kernels.greedy_step(lg, V, -1, codes, state['eos_count'], emb, pe, state['audio_pos'],
                            state['cache_len'], x)
    else:
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        kernels.sample_step(lg, V, -1, top_k, tok_p, temperature, seed, codes, state['eos_count'], lp, emb,
                            pe, state['audio_pos'], state['cache_len'], x)
    return codes[:, 1:2].clone().to(home), lp.to(logits.dtype).to(home)


def get_best_beam(x, sum_logprobs, stop_token, length_penalty=1.0):
    """valle/models/utils.py:71-88 — the beam with the best length-normalised log-probability,
    stop tokens stripped.  O(B*S) integer bookkeeping after the decode loop."""
    length = torch.sum(x != stop_token, dim=-1)
    avg = sum_logprobs / length**length_penalty
    best = x[torch.argmax(avg), :]
    return best[best != stop_token]
