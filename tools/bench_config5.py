"""Decode timing at BASELINE.json configs[4]: 24L/1024d/h16/dff4096 AR, 8 rows, 400 text + 226 prompt
frames, long context; `joint` chains AR generate -> NAR generate for one utterance
(developer tool; usage: python tools/bench_config5.py [joint] [new_tokens=256])."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import ConfigValle, get_model_class, synth  # noqa: E402


def main(new=256):
    cfg = ConfigValle(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0, norm='LayerNorm',
                      num_beams=8, max_audio_len=new)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.to('cuda').eval()
    for frames in (225, 2250):        # 3 s prompt; and the end of a 30 s utterance (context ~2876)
        utts = [synth.synth_utterance(cfg, 200, 200, frames, seed=7 + u) for u in range(8)]
        texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
        firsts = [u[1][:, 0].cuda() for u in utts]
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = m.generate_batch(texts, firsts)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        s0 = 400 + frames + 1
        kv = 2 * 24 * 8 * 1024 * 4 * (s0 + new / 2)
        w = 24 * (4 * 1024 * 1024 + 2 * 1024 * 4096) * 4
        print(f'context {s0}..{s0 + new}: {dt * 1e3:.1f} ms per generate of {new} tokens x 8 rows = '
              f'{8 * new / dt:.0f} tokens/s; decode bytes/step ~ {(kv + w) / 1e9:.2f} GB', flush=True)


def joint(new=256):
    """AR + NAR joint inference of ONE utterance, as a user of the reference would chain them:
    ValleAR.generate (8 beams, default sampling) -> best beam = first codebook -> ValleNAR.generate (7 stages)."""
    kw = dict(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0, num_beams=8, max_audio_len=new)
    cfg_ar = ConfigValle(norm='LayerNorm', **kw)
    cfg_nar = ConfigValle(norm='AdaptiveLayerNorm', **kw)
    ar = get_model_class('ValleAR')(cfg_ar)
    ar.load_state_dict(synth.silence_eos(synth.make_state_dict(cfg_ar, 'ValleAR', seed=0, rich=False), cfg_ar))
    nar = get_model_class('ValleNAR')(cfg_nar)
    nar.load_state_dict(synth.make_state_dict(cfg_nar, 'ValleNAR', seed=1, rich=False))
    ar, nar = ar.to('cuda').eval(), nar.to('cuda').eval()
    prompt_tokens, prompt_codes, target_tokens = [t.cuda() for t in synth.synth_utterance(cfg_ar, 200, 200, 225, seed=11)]
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = ar.generate(prompt_tokens, prompt_codes, target_tokens)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        codes = nar.generate(prompt_tokens, prompt_codes, target_tokens, first)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    frames = first.shape[0]
    print(f'joint AR+NAR, one utterance: AR {frames} frames x 8 beams in {(t1 - t0) * 1e3:.1f} ms, NAR 7 stages over '
          f'{400 + 225 + frames} positions in {(t2 - t1) * 1e3:.1f} ms -> {tuple(codes.shape)} codes; '
          f'{frames / 75:.1f} s of audio in {(t2 - t0):.2f} s', flush=True)


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if '=' in a]
    if 'joint' in sys.argv[1:]:
        joint(*[int(a.split('=')[1]) for a in args])
    else:
        main(*[int(a.split('=')[1]) for a in args])
