"""Deterministic synthetic weights and token batches.

There is no network here for checkpoints or datasets, so tests, the golden-vector generator and
`bench.py` all draw weights and inputs from this module.  Everything comes from a seeded CPU
`torch.Generator`, so the same tensors are rebuilt bit-for-bit on any box (the GPU box never sees
`/root/reference`; it only needs the same seed).

State-dict key layout follows the reference's modules (checkpoint compatibility):
`valle/models/valle_ar.py:19-29`, `valle/models/valle_nar.py:24-47`,
`valle/models/modules.py:114-115,215-218,88-89,284`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

PE_MAX_LEN = 5000  # reference: PositionalEncoding(max_len=5000), valle/models/modules.py:56


def sinusoid_table(d_model: int, max_len: int = PE_MAX_LEN) -> torch.Tensor:
    """(max_len, 1, d_model) fp32 table, pe[p,0,2i]=sin(p*w_i), pe[p,0,2i+1]=cos(p*w_i),
    w_i = exp(-2i*ln(1e4)/d_model).  Same torch-CPU op sequence as the reference buffer
    (valle/models/modules.py:60-66) so the table is bit-identical to a loaded checkpoint's."""
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    table = torch.zeros(max_len, d_model)
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq)
    return table.unsqueeze(1).contiguous()


def state_dict_shapes(cfg, model: str) -> 'OrderedDict[str, tuple]':
    """Every persistent tensor of `model` ('ValleAR' | 'ValleNAR' | 'Transformer') → shape."""
    d, dff, L = cfg.d_model, cfg.dim_feedforward, cfg.num_layers
    va, vt, q = cfg.num_audio_tokens, cfg.vocab_size, cfg.num_quantizers
    shapes: 'OrderedDict[str, tuple]' = OrderedDict()
    pre = '' if model == 'Transformer' else 'transformer.'
    if model == 'ValleAR':
        shapes['tokens_emb.word_embeddings.weight'] = (vt, d)
        shapes['audio_emb.word_embeddings.weight'] = (va + 2, d)
        shapes['proj.weight'] = (va + 1, d)
    elif model == 'ValleNAR':
        shapes['tokens_emb.word_embeddings.weight'] = (vt, d)
        for j in range(q):
            shapes[f'codes_embs.{j}.word_embeddings.weight'] = (va, d)
        for j in range(q - 1):
            shapes[f'stage_embs.{j}.word_embeddings.weight'] = (1, d)
            shapes[f'proj_layers.{j}.weight'] = (va, d)
    elif model != 'Transformer':
        raise KeyError(model)
    if model != 'Transformer':
        shapes['tokens_position_emb.pe'] = (PE_MAX_LEN, 1, d)
        shapes['audio_position_emb.pe'] = (PE_MAX_LEN, 1, d)
    for i in range(L):
        p = f'{pre}layers.{i}.'
        shapes[p + 'self_attn.qkv.weight'] = (3 * d, d)
        shapes[p + 'self_attn.out.weight'] = (d, d)
        shapes[p + 'self_attn.out.bias'] = (d,)
        shapes[p + 'ffn.linear_1.weight'] = (dff, d)
        shapes[p + 'ffn.linear_1.bias'] = (dff,)
        shapes[p + 'ffn.linear_2.weight'] = (d, dff)
        shapes[p + 'ffn.linear_2.bias'] = (d,)
        for n in ('norm1', 'norm2'):
            if cfg.norm == 'LayerNorm':
                shapes[p + n + '.weight'] = (d,)
                shapes[p + n + '.bias'] = (d,)
            else:
                shapes[p + n + '.project_layer.weight'] = (2 * d, d)
                shapes[p + n + '.project_layer.bias'] = (2 * d,)
                shapes[p + n + '.norm.weight'] = (d,)
                shapes[p + n + '.norm.bias'] = (d,)
    return shapes


def make_state_dict(cfg, model: str, seed: int = 0, rich: bool = True, std: float = 0.02):
    """Seeded weights for `model`, filled in sorted-key order.

    rich=False: matrices/embeddings ~ N(0, std), LayerNorm gamma=1/beta=0, biases 0 (the plain
    init SURVEY.md §8d names for the bench).  rich=True additionally perturbs every bias and
    affine parameter so a kernel that drops one is caught by the parity tests; the AdaLN
    projection bias gets mean 1 on its scale half so the stage-conditioned scale is O(1).
    """
    g = torch.Generator(device='cpu').manual_seed(seed)
    shapes = state_dict_shapes(cfg, model)
    d = cfg.d_model
    out = {}
    for key in sorted(shapes):
        shape = shapes[key]
        if key.endswith('.pe'):
            out[key] = sinusoid_table(d)
            continue
        noise = torch.randn(shape, generator=g, dtype=torch.float32)
        is_ln_gamma = key.endswith(('norm1.weight', 'norm2.weight', '.norm.weight'))
        is_bias = key.endswith('.bias')
        if is_ln_gamma:
            out[key] = 1.0 + (0.1 * noise if rich else 0.0 * noise)
        elif key.endswith('project_layer.bias'):
            base = torch.cat([torch.ones(d), torch.zeros(d)])
            out[key] = base + (0.1 * noise if rich else 0.0 * noise)
        elif is_bias:
            out[key] = std * noise if rich else torch.zeros(shape)
        else:
            out[key] = std * noise
    return OrderedDict((k, out[k].contiguous()) for k in shapes)


def silence_eos(state_dict, cfg):
    """Zero the EOS row of the AR head so greedy decoding never stops early:
    `proj` has no bias (valle/models/valle_ar.py:29) so the EOS logit becomes exactly 0."""
    state_dict['proj.weight'][cfg.num_audio_tokens].zero_()
    return state_dict


def synth_utterance(cfg, n_prompt_tokens: int, n_target_tokens: int, n_prompt_frames: int,
                    seed: int = 1234):
    """One utterance in `generate()`'s input format: text ids ~U{0..V_t-1}, codec ids
    ~U{0..V_a-1} (no BOS/EOS inside prompts), int64 (valle/models/valle_ar.py:95-97)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    prompt_tokens = torch.randint(0, cfg.vocab_size, (n_prompt_tokens,), generator=g)
    target_tokens = torch.randint(0, cfg.vocab_size, (n_target_tokens,), generator=g)
    prompt_codes = torch.randint(0, cfg.num_audio_tokens,
                                 (n_prompt_frames, cfg.num_quantizers), generator=g)
    return prompt_tokens, prompt_codes, target_tokens


def synth_ar_batch(cfg, batch: int, tok_range=(40, 120), code_range=(225, 900), seed: int = 1234):
    """A padded AR training batch in the collate wire format (valle/collate.py:23-44):
    codes = BOS + first codebook, target = first codebook + EOS, zero padded, plus *_lens."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    tl = torch.randint(tok_range[0], tok_range[1] + 1, (batch,), generator=g)
    cl = torch.randint(code_range[0], code_range[1] + 1, (batch,), generator=g)
    tokens = torch.zeros(batch, int(tl.max()), dtype=torch.int64)
    codes = torch.zeros(batch, int(cl.max()) + 1, dtype=torch.int64)
    target = torch.zeros(batch, int(cl.max()) + 1, dtype=torch.int64)
    for b in range(batch):
        tokens[b, : tl[b]] = torch.randint(0, cfg.vocab_size, (int(tl[b]),), generator=g)
        first = torch.randint(0, cfg.num_audio_tokens, (int(cl[b]),), generator=g)
        codes[b, 0] = cfg.bos_token
        codes[b, 1 : cl[b] + 1] = first
        target[b, : cl[b]] = first
        target[b, cl[b]] = cfg.eos_token
    return {'codes': codes, 'codes_lens': cl + 1, 'target': target,
            'tokens': tokens, 'tokens_lens': tl}


def synth_nar_batch(cfg, batch: int, n_tokens: int, n_frames: int, seed: int = 1234):
    """A NAR batch in the (fixed, D7) collate layout: codes (B, t, Q) int64, equal lengths."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    tokens = torch.randint(0, cfg.vocab_size, (batch, n_tokens), generator=g)
    codes = torch.randint(0, cfg.num_audio_tokens, (batch, n_frames, cfg.num_quantizers),
                          generator=g)
    return {'codes': codes, 'codes_lens': torch.full((batch,), n_frames, dtype=torch.int64),
            'tokens': tokens, 'tokens_lens': torch.full((batch,), n_tokens, dtype=torch.int64)}
