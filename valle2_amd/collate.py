"""Batch wire format consumed by `training_step` (reference: valle/collate.py:19-66).

AR: first codebook only; input = BOS + codes, target = codes + EOS; zero padding; `*_lens`.
NAR: the reference's collate receives (Q, t) items and breaks on differing t (defect D7); here the
items are transposed to (t, Q) first, so `codes` is (B, t_max, Q) as `ValleNAR` expects.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn.utils.rnn import pad_sequence


def collate_list(x_list: list[Tensor]) -> tuple[Tensor, Tensor]:
    """valle/collate.py:63-66 — zero-pad along dim 0, lengths as int64."""
    lens = torch.tensor([len(x) for x in x_list], dtype=torch.int64)
    return pad_sequence(x_list, batch_first=True), lens


@dataclass
class ValleARCollate:
    config: object

    def __call__(self, batch: list[dict[str, Tensor]]) -> dict[str, Tensor]:
        first = [item['codes'][0] for item in batch]                      # (t,) first codebook
        codes, codes_lens = collate_list([F.pad(c, (1, 0), value=self.config.bos_token) for c in first])
        target, _ = collate_list([F.pad(c, (0, 1), value=self.config.eos_token) for c in first])
        tokens, tokens_lens = collate_list([item['tokens'] for item in batch])
        assert (codes_lens > tokens_lens).all(), 'Codes length must be greater than tokens length.'
        return {'codes': codes, 'codes_lens': codes_lens, 'target': target,
                'tokens': tokens, 'tokens_lens': tokens_lens}


@dataclass
class ValleNARCollate:
    config: object

    def __call__(self, batch: list[dict[str, Tensor]]) -> dict[str, Tensor]:
        codes, codes_lens = collate_list([item['codes'].transpose(0, 1) for item in batch])  # (t, Q)
        tokens, tokens_lens = collate_list([item['tokens'] for item in batch])
        assert (codes_lens > tokens_lens).all(), 'Codes length must be greater than tokens length.'
        return {'codes': codes, 'codes_lens': codes_lens, 'tokens': tokens, 'tokens_lens': tokens_lens}


def get_collate(model_name: str):
    return {'ValleAR': ValleARCollate, 'ValleNAR': ValleNARCollate}[model_name]
