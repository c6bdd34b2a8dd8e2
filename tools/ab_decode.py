"""In-process A/B of the decode-step forms at configs[1] (32 rows, 1024 -> 1536): alternating generates,
per-phase HIP-event times from generate_batch (developer tool)."""
import statistics
import sys
import tempfile
import os
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, get_model_class, synth  # noqa: E402


def main(rounds=8):
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm',
                      top_k=1, num_beams=32, max_audio_len=512)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    utts = [synth.synth_utterance(cfg, 128, 128, 767, seed=1234 + u) for u in range(32)]
    texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
    firsts = [u[1][:, 0].cuda() for u in utts]
    # name -> {tuning knob: value} (include/valle_hip.h: 5 = FeedForward fused / three launches, 7 = slice width,
    # 8 = rows per workgroup, 0 = decode attention variant)
    forms = {'default': {},
             'head + greedy step in one launch': {'env': {'VALLE2_HEAD_FUSED': '1'}},
             'perf mode (bf16 K/V cache)': {'perf': 1},
             'ffn three launches': {5: 1},
             'qkv statistics from row loads': {9: 1},
             'ffn fused 32 x 16 rows': {7: 32, 8: 16},
             'ffn fused 16 x 16 rows': {7: 16, 8: 16}}
    if len(sys.argv) > 1:
        forms = {k: v for k, v in forms.items() if k == 'default' or any(a in k for a in sys.argv[1:])}
    res = {k: [] for k in forms}
    outs = {}
    from valle2_amd import _lib
    lib = _lib.lib()
    for r in range(rounds + 1):
        for name, knobs in forms.items():
            for k in (0, 5, 7, 8, 9):
                lib.vh_set_tuning(k, knobs.get(k, 0))
            os.environ.update(knobs.get('env', {}))
            out = m.generate_batch(texts, firsts, perf_mode=bool(knobs.get('perf')))
            torch.cuda.synchronize()
            if r:
                res[name].append(m.last_generate_stats['decode_ms'] / 511 * 1e3)
            outs[name] = out
            if 'head' in name:                       # the same A/B for the beams of ONE utterance (generate(): shared prompt)
                sh = m.generate_batch([texts[0]] * 32, [firsts[0]] * 32, shared_prompt=True)
                torch.cuda.synchronize()
                if r:
                    res.setdefault(name + ' | shared prompt', []).append(m.last_generate_stats['decode_ms'] / 511 * 1e3)
                outs[name + ' | shared prompt'] = sh
            for k in knobs.get('env', {}):
                os.environ.pop(k)
            if name == 'default':
                sh = m.generate_batch([texts[0]] * 32, [firsts[0]] * 32, shared_prompt=True)
                torch.cuda.synchronize()
                if r:
                    res.setdefault('default | shared prompt', []).append(m.last_generate_stats['decode_ms'] / 511 * 1e3)
                outs['default | shared prompt'] = sh
    for k in (0, 5, 7, 8, 9):
        lib.vh_set_tuning(k, 0)
    for name, v in res.items():
        print(f'{name:46s} decode step {statistics.median(v):7.2f} us (min {min(v):7.2f}, max {max(v):7.2f}, n={len(v)})')
    print('same tokens:', all(bool(torch.equal(outs['default | shared prompt' if 'shared' in k else 'default'], o))
                              for k, o in outs.items()))


if __name__ == '__main__':
    main()
