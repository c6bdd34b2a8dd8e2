"""Soak of the training step (developer tool): N steps of configs[3] AR then NAR with a DIFFERENT batch shape every step
(ragged lengths drawn per seed -> a different row count, tile count and tail split each step), every loss finite, the
loss of one repeated batch going down, device error flags clean, allocator footprint flat.
usage: python tools/soak_train.py [steps=150] [dropout=0.0]   (dropout = config.dropout; 0.1 = the reference default)"""
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, _lib, get_model_class, synth  # noqa: E402


def main(steps=150, dropout=0.0):
    dev = torch.device('cuda:0')
    for name, norm in (('ValleAR', 'LayerNorm'), ('ValleNAR', 'AdaptiveLayerNorm')):
        cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=dropout, norm=norm, batch_size=16)
        torch.manual_seed(0)
        model = get_model_class(name)(cfg).to(dev).train()
        opt = model.configure_optimizers()['optimizer']
        times, losses, probe, shapes = [], [], [], set()
        peak0 = None
        for i in range(steps):
            rep = i % 10 == 0                                       # every 10th step: the same batch again
            seed = 7 if rep else 1000 + i
            if name == 'ValleAR':
                batch = synth.synth_ar_batch(cfg, 16, seed=seed)
            else:
                frames = 560 if rep else 300 + (i * 37) % 500
                batch = synth.synth_nar_batch(cfg, 16, n_tokens=80, n_frames=frames, seed=seed)
            batch = {k: (v if k.endswith('_lens') else v.to(dev)) for k, v in batch.items()}
            shapes.add((batch['codes'].shape[1], batch['tokens'].shape[1]))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = model.training_step(batch, **({'stage': 1 + i % 7} if name == 'ValleNAR' and not rep else
                                                  ({'stage': 3} if name == 'ValleNAR' else {})))
            loss.backward()
            opt.step(max_norm=1.0, zero_grad=True)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            lv = float(loss.detach())
            assert lv == lv and abs(lv) < 1e4, f'{name} step {i}: loss {lv}'
            losses.append(lv)
            if rep:
                probe.append(lv)
            if i == 20:
                peak0 = torch.cuda.memory_reserved(dev)
        _lib.raise_device_errors(dev)
        opt.check_errors()
        grow = (torch.cuda.memory_reserved(dev) - peak0) / 2 ** 20
        assert probe[-1] < probe[0], f'{name}: the repeated batch did not improve ({probe[0]:.3f} -> {probe[-1]:.3f})'
        print(f'{name} (dropout {dropout}): {steps} steps over {len(shapes)} batch shapes, all losses finite; repeated batch {probe[0]:.3f} -> '
              f'{probe[-1]:.3f}; step min {min(times):.1f} median {sorted(times)[len(times) // 2]:.1f} max {max(times[5:]):.1f} ms; '
              f'reserved memory since step 20: {grow:+.0f} MiB ({torch.cuda.memory_reserved(dev) / 2 ** 30:.1f} GiB)', flush=True)
        del model, opt
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main(*([int(sys.argv[1])] if len(sys.argv) > 1 else []), *([float(sys.argv[2])] if len(sys.argv) > 2 else []))
