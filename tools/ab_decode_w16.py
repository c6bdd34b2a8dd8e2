"""A/B of the perf-mode decode step with fp32 and with h16 weights (VALLE2_DECODE_W16 = 0 | 1): configs[1] (32 rows, 1024-token
prompt -> 512 new tokens, 12L/512d) and configs[4]'s AR leg (8 rows, 24L/1024d, 400 text + 225 prompt frames -> 2250 new tokens),
alternating arms in one process; tokens of the two arms compared.

    python tools/ab_decode_w16.py [--rounds 3]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
DEV = 'cuda'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=3)
    args = ap.parse_args()
    from valle2_amd import ConfigValle, engine, get_model_class, synth
    for name, kw, rows, text, frames, new in (
            ('configs[1] 12L/512d x 32 rows', dict(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12), 32, 256, 767, 512),
            ('24L/1024d x 16 rows (configs[4] model; perf mode wants rows x heads >= 256)', dict(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24), 16, 400, 225, 512)):
        cfg = ConfigValle(**kw, dropout=0.0, norm='LayerNorm', num_beams=rows, top_k=1, max_audio_len=new)
        sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
        m = get_model_class('ValleAR')(cfg)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        utts = [synth.synth_utterance(cfg, text // 2, text - text // 2, frames, seed=1234 + u) for u in range(rows)]
        texts = [torch.cat([u[0], u[2]]).to(DEV) for u in utts]
        firsts = [u[1][:, 0].to(DEV) for u in utts]
        best, toks = {}, {}
        for rnd in range(args.rounds):
            for w16 in (False, True):
                engine.DECODE_W16 = w16
                m.release_decoders()
                m.generate_batch(texts, firsts, perf_mode=True)          # builds the decoder of this arm
                out = m.generate_batch(texts, firsts, perf_mode=True)
                st = m.last_generate_stats
                assert st['decode_w16'] == w16, st
                us = st['decode_ms'] / (new - 1) * 1e3
                best[w16] = min(best.get(w16, 1e30), us)
                toks[w16] = out
        engine.DECODE_W16 = True
        f32 = m.generate_batch(texts, firsts)
        us32 = m.last_generate_stats['decode_ms'] / (new - 1) * 1e3
        print(f'{name}: perf-mode decode step {best[False]:7.1f} us with fp32 weights, {best[True]:7.1f} us with h16 weights '
              f'(x{best[False] / best[True]:.3f}); fp32 path {us32:7.1f} us; tokens equal between the arms: '
              f'{float((toks[False] == toks[True]).float().mean()):.4f}, h16-weights arm vs fp32 path: {float((toks[True] == f32).float().mean()):.4f}',
              flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
