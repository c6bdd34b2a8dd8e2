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
from valle2_amd import ConfigValle, engine, get_model_class, synth  # noqa: E402


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
    base = dict(TWO_SLAB_RESIDUAL=False, ACC64_RESIDUAL=False, PERSISTENT_CHAIN=False, PIPELINED_ATTENTION=False)
    forms = {'default': dict(base),
             'two-slab': dict(base, TWO_SLAB_RESIDUAL=True),
             'pipelined attention': dict(base, PIPELINED_ATTENTION=True),
             'pipelined kernels in stream order': dict(base, PIPELINED_ATTENTION=True, _pipe_mode=1)}
    if len(sys.argv) > 1:
        forms = {k: v for k, v in forms.items() if k == 'default' or any(a in k for a in sys.argv[1:])}
    res = {k: [] for k in forms}
    outs = {}
    for r in range(rounds + 1):
        for name, flags in forms.items():
            from valle2_amd import _lib
            _lib.lib().vh_set_tuning(0, flags.get('_variant', 0))
            _lib.lib().vh_set_tuning(7, flags.get('_pipe_mode', 0))
            for k, v in flags.items():
                if not k.startswith('_'):
                    setattr(engine, k, v)
            out = m.generate_batch(texts, firsts)
            torch.cuda.synchronize()
            if r:
                res[name].append(m.last_generate_stats['decode_ms'] / 511 * 1e3)
            outs[name] = out
    for name, v in res.items():
        print(f'{name:46s} decode step {statistics.median(v):7.2f} us (min {min(v):7.2f}, max {max(v):7.2f}, n={len(v)})')
    print('same tokens:', all(bool(torch.equal(outs['default'], o)) for o in outs.values()))


if __name__ == '__main__':
    main()
