"""A/B of the shared-prompt decode (generate() of one utterance, configs[1], 32 beams) against the independent-rows decode
and across the suffix key-split count (VALLE2_SHARED_SPLIT): alternating generates in ONE process.

    python tools/ab_shared_prompt.py [--reps 3]
"""
import argparse
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    os.chdir(tempfile.mkdtemp(prefix='vh_ab_'))
    from valle2_amd import ConfigValle, get_model_class, synth
    dev = 'cuda'
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm', num_beams=32,
                      top_k=1, max_audio_len=512)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    u = synth.synth_utterance(cfg, 128, 128, 767, seed=1234)
    text, first = torch.cat([u[0], u[2]]).to(dev), u[1][:, 0].to(dev)
    arms = [('independent', dict(shared_prompt=False), None), ('shared split 1', dict(shared_prompt=True), '1'),
            ('shared split 2', dict(shared_prompt=True), '2'), ('shared split 4', dict(shared_prompt=True), '4')]
    res = {n: [] for n, _, _ in arms}
    ref = None
    for rep in range(args.reps + 1):
        for name, kw, split in arms:
            if split is None:
                os.environ.pop('VALLE2_SHARED_SPLIT', None)
            else:
                os.environ['VALLE2_SHARED_SPLIT'] = split
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = m.generate_batch([text] * 32, [first] * 32, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = m.last_generate_stats
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), name
            if rep:
                res[name].append((dt * 1e3, st['decode_ms'] / 511 * 1e3, st['prefill_ms']))
    for name, v in res.items():
        ms = sorted(x[0] for x in v)[len(v) // 2]
        us = sorted(x[1] for x in v)[len(v) // 2]
        print(f'{name:16s} {ms:8.2f} ms per generate, {us:7.1f} us per decode step, prompt pass {v[-1][2]:.2f} ms')


if __name__ == '__main__':
    main()
