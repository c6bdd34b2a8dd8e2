"""Host-side profile of the NAR training step at configs[3] with HOST-resident batches and a different stage every step
(what tools/bench_train.py runs): per-phase wall times with synchronisation, then cProfile of the forward."""
import cProfile
import os
import pstats
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, get_model_class, synth  # noqa: E402

cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='AdaptiveLayerNorm', batch_size=16)
torch.manual_seed(0)
model = get_model_class('ValleNAR')(cfg).cuda().train()
opt = model.configure_optimizers()['optimizer']
on_dev = len(sys.argv) > 1 and sys.argv[1] == 'dev'
batches = [synth.synth_nar_batch(cfg, 16, n_tokens=80, n_frames=560, seed=100 + i) for i in range(24)]
if on_dev:
    batches = [{k: (v if k.endswith('_lens') else v.cuda()) for k, v in b.items()} for b in batches]


def step(i, prof=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    loss = model.training_step(batches[i], stage=1 + i % 7)
    if prof:
        prof.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    opt.step(max_norm=1.0, zero_grad=True)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    return [(t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3]


for i in range(16):
    r = step(i)
    print(f'step {i:2d} stage {1 + i % 7}: fwd enqueue {r[0]:6.1f} ms, fwd {r[1]:6.1f}, bwd {r[2]:6.1f}, opt {r[3]:5.1f}', flush=True)
pr = cProfile.Profile()
for i in range(16, 24):
    step(i, pr)
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
