"""Developer tool: A/B of a tuning knob on the WHOLE configs[3] training step in one process (alternating blocks of steps,
same batches): python tools/ab_train_knob.py <knob> <value A> <value B> [model=ValleAR]"""
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, _lib, get_model_class, synth  # noqa: E402


def main(knob, va, vb, name='ValleAR'):
    knob, va, vb = int(knob), int(va), int(vb)
    dev = torch.device('cuda:0')
    norm = 'LayerNorm' if name == 'ValleAR' else 'AdaptiveLayerNorm'
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm=norm, batch_size=16)
    torch.manual_seed(0)
    model = get_model_class(name)(cfg).to(dev).train()
    opt = model.configure_optimizers()['optimizer']
    batches = []
    for i in range(5):
        b = synth.synth_ar_batch(cfg, 16, seed=101 + i) if name == 'ValleAR' else \
            synth.synth_nar_batch(cfg, 16, n_tokens=80, n_frames=560, seed=101 + i)
        batches.append({k: (v if k.endswith('_lens') else v.to(dev)) for k, v in b.items()})

    def block(v):
        _lib.lib().vh_set_tuning(knob, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            loss = model.training_step(b, **({'stage': 3} if name == 'ValleNAR' else {}))
            loss.backward()
            opt.step(max_norm=1.0, zero_grad=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / len(batches) * 1e3

    block(va); block(vb)
    res = {va: [], vb: []}
    for _ in range(6):
        for v in (va, vb):
            res[v].append(block(v))
    _lib.lib().vh_set_tuning(knob, 0)
    for v in (va, vb):
        r = sorted(res[v])
        print(f'{name} knob {knob} = {v}: median {r[len(r) // 2]:.3f} ms per step (min {r[0]:.3f}, max {r[-1]:.3f})', flush=True)


if __name__ == '__main__':
    main(*sys.argv[1:])
