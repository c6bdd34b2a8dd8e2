"""`train(hparams_fp, model_name)` with the reference's signature (valle/train_model.py:13-35), as a
plain data-parallel loop: one process per GPU (torch.distributed, RCCL over xGMI), per-rank
micro-batches, one sum all-reduce of the flat fp32 gradient per optimizer step, launched bucket by
bucket while backward is still running (`dp.GradReducer`; what Lightning's implicit DDP does for the
reference), then the 1/world mean, global-norm clipping (`gradient_clip_val`) and AdamW in one flat
HIP pass (`optim.FlatAdamW`), CosineAnnealingWarmRestarts from `configure_optimizers`, gradient
accumulation (`grad_accum`).

The reference's data pipeline (HF dataset + g2p + on-the-fly EnCodec, valle/data.py) is out of
scope: pass any iterable of collated batches as `batches`, or none to train on seeded synthetic
batches in the collate wire format.
"""
from __future__ import annotations

import argparse
import os
import random
import time
from pathlib import Path

import torch

from . import dp, dropout, synth
from .config import ConfigValle


def synthetic_batches(model_name, config, rank, world, steps):
    for i in range(steps):
        seed = config.seed + 1000 * i + rank
        if model_name == 'ValleAR':
            yield synth.synth_ar_batch(config, config.batch_size, seed=seed)
        else:
            yield synth.synth_nar_batch(config, config.batch_size, n_tokens=80, n_frames=450, seed=seed)


def train(hparams_fp: Path, model_name: str, batches=None, device=None, log=print):
    from . import get_model_class
    config = ConfigValle.from_json(hparams_fp) if not isinstance(hparams_fp, ConfigValle) else hparams_fp
    rank, local, _ = dp.env_world()
    local = int(os.environ.get('VALLE2_FORCE_DEVICE', local))     # rehearsal: several ranks on one GPU
    device = device or torch.device('cuda', local)
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    rank, world = dp.init_distributed(device=device if device.type == 'cuda' else None)
    torch.manual_seed(config.seed)                       # identical initial weights on every rank
    random.seed(config.seed)                             # and the same NAR stage draw (valle_nar.py:76) on every rank:
    #                                                      the reference calls seed_everything (train_model.py:15)
    dropout.set_rank(rank)                               # ...but every rank its own dropout fields (the rank is part of a
    #                                                      field's site id): same seeds, different masks per shard
    model = get_model_class(model_name)(config).to(device).train()
    opt = model.configure_optimizers()                   # FlatAdamW + CosineAnnealingWarmRestarts
    optimizer, scheduler = opt['optimizer'], opt['lr_scheduler']
    reducer = dp.GradReducer(optimizer.flat_grad, optimizer.slots)
    accum = max(1, config.grad_accum)
    if batches is None:
        batches = synthetic_batches(model_name, config, rank, world, config.max_steps * accum)
    step, t0, losses = 0, time.perf_counter(), []
    optimizer.zero_grad()
    one_shot = iter(batches) is batches                  # a generator: a single epoch, then stop
    pending = 0                                          # micro-batches accumulated since the last optimizer step

    def optimizer_step():
        nonlocal step, pending
        reducer.finish()
        optimizer.sync_touched()                         # same update decision per parameter on every rank
        # mean over ranks + global-norm clip + AdamW + zeroing of the gradients: one flat pass
        optimizer.step(grad_scale=1.0 / world, max_norm=config.gradient_clip_val, zero_grad=True)
        step += 1
        if rank == 0 and step % max(1, config.log_every_n_steps) == 0:
            log(f'step {step}: train/loss {float(sum(losses[-pending:])) / pending:.4f} '
                f'({(time.perf_counter() - t0) / step * 1e3:.0f} ms/step, world {world})')
        pending = 0

    while step < config.max_steps:
        steps_before = step
        for batch in batches:
            pending += 1
            last = pending == accum                      # counted within the epoch: Lightning's accumulation window
            reducer.enabled = last                       # accumulate locally, exchange once per step
            loss = model.training_step(batch)
            (loss / accum).backward()                    # buckets are all-reduced as they complete
            losses.append(loss.detach())                 # stays on the device: no host synchronisation per step
            if last:
                optimizer_step()
                if step >= config.max_steps:
                    break
        if pending and step < config.max_steps:
            # an epoch that ends inside an accumulation window: Lightning steps the optimizer on the incomplete window
            # (the leftover micro-batches were accumulated with the exchange off: gather and reduce everything now)
            optimizer.gather_grads()
            reducer.enabled = True
            optimizer_step()
        # Lightning steps a scheduler returned without an 'interval' once per EPOCH (the reference's
        # configure_optimizers returns {'optimizer', 'lr_scheduler'} only, valle_ar.py:182-194): T_0 =
        # lr_warmup counts epochs there, so it does here.
        scheduler.step()
        if one_shot or step == steps_before:             # a generator is one epoch; an epoch without a step is empty
            break
    optimizer.check_errors()          # a device-side range error of the last step (earlier ones raise at the next step)
    return model, [float(x) for x in losses]


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', type=Path, required=True)
    parser.add_argument('-m', '--model', type=str, choices=['ValleAR', 'ValleNAR'], required=True)
    args = parser.parse_args()
    train(args.config, args.model)            # the reference reads args.hparams here and dies (D1)
