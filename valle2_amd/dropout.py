"""Dropout of the training path as counter-based fields (include/valle_hip.h, `vh_dropout_spec`).

The reference calls nn.Dropout at four places of a training step — PositionalEncoding (valle/models/modules.py:56-58,80:
p = 0.1 whatever config.dropout says, D9), FeedForward (:219), EncoderLayer.dropout1 / dropout2 (:277-278; p =
config.dropout, default 0.1, valle/config.py:26) — and TokenEmbedding (:35, p = 0 by default).  Here a mask is never a
tensor: it is a pure function of (seed, site, row, column) (Philox4x32-7) that the GEMM epilogues, the LayerNorm backward
and the embedding kernels regenerate where they need it.

  seed  one 62-bit draw per forward from torch's CPU generator (so `torch.manual_seed` makes a run repeatable, as it
        does for the reference), or from the private generator installed by `manual_seed` below;
  site  (rank << 48) | (layer << 8) | kind — tells the fields of one forward apart; `set_rank` gives every data-parallel
        rank its own fields even when all ranks were seeded alike (Lightning's seed_everything does exactly that).

Nothing here computes on the CPU; `mask()` exports a field for the tests that hand it to the CPU oracle.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import VhDropoutSpec, check, stream

# site kinds
PE_TEXT, PE_AUDIO, EMB, ATTN_RES, FFN_HID, FFN_RES, MODULE = 1, 2, 3, 4, 5, 6, 7

_state = {'generator': None, 'rank': 0, 'module_calls': 0}


def manual_seed(seed: int):
    """Draw the per-forward seeds from a private generator seeded here (default: torch's global CPU generator)."""
    _state['generator'] = torch.Generator().manual_seed(int(seed))
    _state['module_calls'] = 0


def set_rank(rank: int):
    """Data-parallel rank mixed into every site id: ranks seeded alike still draw different fields."""
    _state['rank'] = int(rank) & 0xFFFF


def draw_seed() -> int:
    """One seed per forward pass (a host-side draw: no device work, no synchronisation)."""
    return int(torch.randint(0, 2 ** 62, (1,), generator=_state['generator']).item())


def seed_if(*ps) -> int:
    """A fresh seed when any of the probabilities `ps` is live, else 0 WITHOUT touching the generator: a forward with every
    dropout off (eval mode, p = 0) must not advance the caller's CPU random stream."""
    return draw_seed() if any(p > 0 for p in ps) else 0


def site(kind: int, layer: int = 0) -> int:
    return (_state['rank'] << 48) | ((int(layer) & 0xFFFFFF) << 8) | (int(kind) & 0xFF)


def live(module) -> float:
    """p of an nn.Dropout that is active (training mode, p > 0), else 0."""
    return float(module.p) if (module.training and module.p > 0) else 0.0


def spec(seed: int, site_id: int, p: float):
    """ctypes vh_dropout_spec, or None when p == 0 (no dropout).  p == 1 is nn.Dropout's all-zero corner: callers
    handle it before they get here (`apply`)."""
    if not p:
        return None
    if not 0.0 < p < 1.0:
        raise _lib.VhError(f'dropout p={p} must be in [0, 1) on the fused path')
    return VhDropoutSpec(int(seed) & (2 ** 64 - 1), int(site_id) & (2 ** 64 - 1), float(p))


def _ref(sp):
    import ctypes as C
    return None if sp is None else C.byref(sp)


def _rows_cols(x):
    cols = x.shape[-1]
    if cols % 4 or x.stride(-1) != 1:
        raise _lib.VhError('dropout: the last dimension must be contiguous and a multiple of 4')
    return x.numel() // cols, cols


def apply_raw(x, sp, out=None):
    """out = field(sp) * x / (1 - p) over x viewed as (rows, last dim) — forward and backward of one nn.Dropout."""
    x = x.contiguous()
    rows, cols = _rows_cols(x)
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().vh_dropout(x.data_ptr(), cols, out.data_ptr(), cols, rows, cols, _ref(sp), stream()), 'vh_dropout')
    return out


def mask(sp, rows: int, cols: int, device) -> torch.Tensor:
    """The keep field (rows, cols) as uint8 — test hook / export for the CPU oracle."""
    keep = torch.empty(rows, cols, device=device, dtype=torch.uint8)
    check(_lib.lib().vh_dropout_mask(keep.data_ptr(), rows, cols, _ref(sp), stream()), 'vh_dropout_mask')
    return keep


class DropoutFn(torch.autograd.Function):
    """A free-standing nn.Dropout (module-level forwards outside the fused training step)."""

    @staticmethod
    def forward(ctx, x, sp):
        ctx.sp = sp
        return apply_raw(x, sp)

    @staticmethod
    def backward(ctx, dy):
        return apply_raw(dy, ctx.sp), None


def apply(module, x):
    """`module(x)` for an nn.Dropout on the HIP kernels: identity in eval mode / p = 0, zeros at p = 1, else a fresh
    field (its own seed draw) through vh_dropout."""
    p = live(module)
    if not p:
        return x
    if p >= 1.0:
        return x * 0.0
    _state['module_calls'] += 1
    sp = spec(draw_seed(), site(MODULE, _state['module_calls']), p)
    return DropoutFn.apply(x, sp)


# ---- the fields of one fused training forward ------------------------------------------------------------------------
class StackDropout:
    """Seed + per-layer specs of one `transformer_train` call.  `record`: a dict the tests pass to receive every spec
    with its (rows, cols) so they can export the masks afterwards."""

    def __init__(self, layers, seed=None):
        self.p = [(live(l.dropout1), live(l.ffn.dropout), live(l.dropout2)) for l in layers]
        # a seed is drawn from torch's CPU generator only when some dropout is live: an eval-mode or p = 0 forward must not
        # advance the user's CPU random stream (DataLoader shuffling, random_split ...) — the reference consumes no CPU
        # randomness there (round-4 advisor finding)
        self.seed = int(seed) if seed is not None else (draw_seed() if self.any else 0)
        for ps in self.p:
            if any(q >= 1.0 for q in ps):
                raise _lib.VhError('dropout p = 1 is not supported on the fused training path')

    @property
    def any(self):
        return any(q > 0 for ps in self.p for q in ps)

    def layer(self, i):
        """(attn-residual spec, ffn-hidden spec, ffn-residual spec) of layer i; None where that dropout is off."""
        p1, pf, p2 = self.p[i]
        return (spec(self.seed, site(ATTN_RES, i), p1), spec(self.seed, site(FFN_HID, i), pf),
                spec(self.seed, site(FFN_RES, i), p2))


# test hook: when set to a list, every fused forward appends {'name', 'spec', 'rows', 'cols'} for each live field
RECORD = None


def record(name, sp, rows, cols):
    if RECORD is not None and sp is not None:
        RECORD.append({'name': name, 'seed': sp.seed, 'site': sp.site, 'p': sp.p, 'rows': int(rows), 'cols': int(cols)})
