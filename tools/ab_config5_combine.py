"""the split-KV combine of the decode attention as its own launch (default) against the last-arriver combine
inside the attention launch (VH_TUNE_DECODE_COMBINE = 1), re-measured where the attention pair is long: configs[4]'s AR leg —
24L/1024d, 8 rows, contexts 626 and 2651 (+ 256 new tokens).  DESIGN 3.1 measured it slower at 4 beams x 8 heads.
Alternating arms in one process, best of three, same tokens checked."""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, _lib, get_model_class, synth  # noqa: E402

cfg = ConfigValle(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0, norm='LayerNorm', num_beams=8, top_k=1,
                  max_audio_len=256)
sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
m = get_model_class('ValleAR')(cfg)
m.load_state_dict(sd)
m = m.to('cuda').eval()
for frames in (225, 2250):
    utts = [synth.synth_utterance(cfg, 200, 200, frames, seed=7 + u) for u in range(8)]
    texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
    firsts = [u[1][:, 0].cuda() for u in utts]
    best, toks = {}, {}
    for rnd in range(3):
        for knob in (0, 1):
            _lib.lib().vh_set_tuning(12, knob)
            m.generate_batch(texts, firsts)                      # (a new tuning epoch: this call builds the arm's decoder)
            out = m.generate_batch(texts, firsts)
            torch.cuda.synchronize()
            us = m.last_generate_stats['decode_ms'] / 255 * 1e3
            best[knob] = min(best.get(knob, 1e30), us)
            toks[knob] = out
    print(f'context {400 + frames + 1}..{400 + frames + 257}: combine as its own launch {best[0]:.1f} us per step, inside the attention launch '
          f'{best[1]:.1f} us per step ({best[0] - best[1]:+.1f} us, {(best[0] - best[1]) / 24:+.2f} us per layer); same tokens '
          f'{bool(torch.equal(toks[0], toks[1]))}; n_split {m.last_generate_stats["n_split"]}', flush=True)
_lib.lib().vh_set_tuning(12, 0)
