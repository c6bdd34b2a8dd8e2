"""Host-side profile of the AR training step at configs[3] (developer tool): cProfile over a few steps, top entries by
own time, plus enqueue time (no sync) against synchronised step time."""
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
from valle2_amd import ConfigValle, dp, get_model_class, synth  # noqa: E402

cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm', batch_size=16)
torch.manual_seed(0)
model = get_model_class('ValleAR')(cfg).cuda().train()
opt = model.configure_optimizers()['optimizer']
batch = {k: (v if k.endswith('_lens') else v.cuda())          # lengths stay on the host, as the collate functions leave them
         for k, v in synth.synth_ar_batch(cfg, 16, tok_range=(40, 120), code_range=(225, 900), seed=7).items()}


def step():
    loss = model.training_step(batch)
    loss.backward()
    opt.step(max_norm=1.0, zero_grad=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'5 steps: host enqueue {t_enq / 5 * 1e3:.1f} ms per step, with the final sync {t_all / 5 * 1e3:.1f} ms per step')
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
