"""In-process A/B on generate() at the reference's generation defaults (num_beams=4, top_k=50, max_audio_len=1024, 12L/512d):
key splits of the decode attention combined inside the launch by the last workgroup to arrive (knob 12 = 1) against the
separate combine launch (knob 12 = 0, default).  Alternating generates under one torch seed (same samples); decode
time from the HIP events inside generate_batch.   usage: python tools/ab_default_generate.py [rounds=6]"""
import os
import statistics
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, _lib, get_model_class, synth  # noqa: E402


def main(rounds=6):
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm')
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    utt = [u.cuda() for u in synth.synth_utterance(cfg, 128, 128, 767, seed=1234)]
    lib = _lib.lib()
    forms = {'combine in the launch (last arriver)': 1, 'combine as a second launch': 0}
    res, outs = {k: [] for k in forms}, {}
    for r in range(rounds + 1):
        for name, knob in forms.items():
            lib.vh_set_tuning(12, knob)
            torch.manual_seed(0)
            outs[name] = m.generate(*utt)
            torch.cuda.synchronize()
            st = m.last_generate_stats
            if r:
                res[name].append(st['decode_ms'] / (st['steps_run'] - 1) * 1e3)
    lib.vh_set_tuning(12, 0)
    for name, v in res.items():
        print(f'{name:40s} decode step {statistics.median(v):7.2f} us (min {min(v):7.2f}, max {max(v):7.2f}, n={len(v)})')
    a, b = outs.values()
    print('same tokens:', bool(a.shape == b.shape and torch.equal(a, b)), 'steps', st['steps_run'], 'n_split', st['n_split'])


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:]])
