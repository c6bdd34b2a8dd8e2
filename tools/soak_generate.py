"""Soak: the same configs[1] generate N times; every run must return the first run's tokens (developer tool).
usage: soak_generate.py [n=60] [shared=0]     shared=1: generate() of ONE utterance with 32 beams (shared prompt K/V,
vh_attn_decode_shared) and with 4 sampled beams under a fixed torch seed, instead of 32 distinct rows"""
import os
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, get_model_class, synth  # noqa: E402


def main(n=60, shared=0):
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm',
                      top_k=1, num_beams=32, max_audio_len=512)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    utts = [synth.synth_utterance(cfg, 128, 128, 767, seed=1234 + u) for u in range(32)]
    texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
    firsts = [u[1][:, 0].cuda() for u in utts]
    if shared:
        u0 = [t.cuda() for t in utts[0]]
        for label, kw in (('32 greedy beams', {}), ('4 sampled beams (top_k 50, torch.manual_seed(7))', dict(num_beams=4, top_k=50))):
            for k, v in kw.items():
                setattr(m.config, k, v)
            ref, bad, times = None, 0, []
            for i in range(n):
                torch.manual_seed(7)
                out = m.generate(*u0)
                assert m.last_generate_stats['shared_prompt']
                times.append(m.last_generate_stats['decode_ms'] / max(1, m.last_generate_stats['steps_run'] - 1) * 1e3)
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(out, ref):
                    bad += 1
            print(f'generate(), {label}: {n} runs, {bad} differing from the first; decode step min {min(times):.1f} '
                  f'median {sorted(times)[n // 2]:.1f} max {max(times):.1f} us', flush=True)
        return
    ref = None
    for _ in (0,):
        bad, times = 0, []
        for i in range(n):
            out = m.generate_batch(texts, firsts)
            times.append(m.last_generate_stats['decode_ms'] / 511 * 1e3)
            if ref is None:
                ref = out.clone()
            elif not torch.equal(out, ref):
                bad += 1
        print(f'{n} generates, {bad} differing from the first; decode step min {min(times):.1f} '
              f'median {sorted(times)[n // 2]:.1f} max {max(times):.1f} us', flush=True)


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:]])
