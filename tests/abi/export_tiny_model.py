"""Write the flat model file tests/abi/c_abi_decode.c loads: the 2L/128d AR model and utterance of the `ar_generate_tiny`
golden case (BASELINE configs[0]: 128 text + BOS + 255 codec tokens, 4 beams, 64 greedy tokens) and the REAL reference's
tokens + margins from tests/golden/ar_generate_tiny.npz.  CPU only (torch as a tensor library); the C program that reads
the file has no Python and no torch in its process.

Layout (little endian): int32 header[12] = {magic 'VHM1', d_model, n_heads, dff, n_layers, V_text, V_audio, n_text,
n_prompt (BOS included), n_new, beams, n_pe}; then fp32: tokens_emb (V_text, d), audio_emb (V_audio + 2, d),
pe_text (n_pe, d), pe_audio (n_pe, d), proj (V_audio + 1, d), per layer ln1_g, ln1_b, wqkv (3d, d), wo (d, d), bo,
ln2_g, ln2_b, w1 (dff, d), b1, w2 (d, dff), b2; then int64: text ids (n_text), codes (n_prompt), golden tokens (n_new);
then fp32 margins (n_new).

Usage: python tests/abi/export_tiny_model.py OUT.bin
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

MAGIC = 0x314D4856      # 'VHM1'


def export(path):
    from tests.golden import cases as C
    from tests.oracle_runners import load_golden
    kw, sd, utt = C.ar_generate_inputs('tiny')
    cfg = C.cfg_of(kw)
    gold = load_golden('ar_generate_tiny')
    text = torch.cat([utt[0], utt[2]])
    codes = torch.cat([torch.tensor([cfg.bos_token]), utt[1][:, 0]])
    n_new = int(gold['steps'])
    n_pe = max(len(text), len(codes) + n_new) + 1
    hdr = np.array([MAGIC, cfg.d_model, cfg.n_heads, cfg.dim_feedforward, cfg.num_layers, cfg.vocab_size,
                    cfg.num_audio_tokens, len(text), len(codes), n_new, cfg.num_beams, n_pe], dtype='<i4')
    f32 = [sd['tokens_emb.word_embeddings.weight'], sd['audio_emb.word_embeddings.weight'],
           sd['tokens_position_emb.pe'][:n_pe, 0], sd['audio_position_emb.pe'][:n_pe, 0], sd['proj.weight']]
    for i in range(cfg.num_layers):
        p = f'transformer.layers.{i}.'
        f32 += [sd[p + k] for k in ('norm1.weight', 'norm1.bias', 'self_attn.qkv.weight', 'self_attn.out.weight',
                                    'self_attn.out.bias', 'norm2.weight', 'norm2.bias', 'ffn.linear_1.weight',
                                    'ffn.linear_1.bias', 'ffn.linear_2.weight', 'ffn.linear_2.bias')]
    with open(path, 'wb') as f:
        f.write(hdr.tobytes())
        for t in f32:
            f.write(t.contiguous().numpy().astype('<f4').tobytes())
        for t in (text, codes, gold['tokens']):
            f.write(t.numpy().astype('<i8').tobytes())
        f.write(gold['margin'].numpy().astype('<f4').tobytes())
    return path


if __name__ == '__main__':
    print(export(sys.argv[1]))
