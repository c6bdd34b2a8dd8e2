"""Training-step timing at BASELINE.json configs[3]: 12L/512d AR and NAR forward+backward, per-GPU
batch 16, LibriTTS-shaped synthetic batches (tokens 40..120, codes 225..900), fp32.  Developer tool
(the graded bench is bench.py); under torch.distributed.run it averages gradients over RCCL.
usage: python tools/bench_train.py [steps=5] [dropout=0.1] [attn_bwd=1] [only=ar|nar]   (attn_bwd: VH_TUNE_ATTN_BWD; dropout = config.dropout: dropout1 / dropout2 / FeedForward
dropout; the PositionalEncoding dropout 0.1 is live in train mode either way, D9)"""
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import ConfigValle, dp, get_model_class, synth  # noqa: E402


def main(steps=5, dropout=0.0, attn_bwd=0, only=''):
    os.chdir(tempfile.mkdtemp())
    rank, local, world = dp.env_world()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dp.init_distributed('nccl', dev)
    from valle2_amd import _lib
    _lib.lib().vh_set_tuning(13, attn_bwd)
    for name, norm in (('ValleAR', 'LayerNorm'), ('ValleNAR', 'AdaptiveLayerNorm')):
        if only and only not in name.lower():
            continue
        cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=dropout, norm=norm,
                          batch_size=16)
        torch.manual_seed(0)
        model = get_model_class(name)(cfg).to(dev).train()
        opt = model.configure_optimizers()['optimizer']          # optim.FlatAdamW
        reducer = dp.GradReducer(opt.flat_grad, opt.slots)
        times, phases = [], []
        for i in range(steps + 3):          # 3 warm-up steps: library GEMM selection runs on the first ones
            if name == 'ValleAR':
                batch = synth.synth_ar_batch(cfg, 16, seed=100 + i + 1000 * rank)
            else:
                batch = synth.synth_nar_batch(cfg, 16, n_tokens=80, n_frames=560, seed=100 + i + 1000 * rank)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = model.training_step(batch, **({'stage': 1 + i % 7} if name == 'ValleNAR' else {}))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            reducer.finish()                                     # buckets were launched during backward
            opt.sync_touched()
            opt.step(grad_scale=1.0 / world, max_norm=cfg.gradient_clip_val, zero_grad=True)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            if i >= 3:
                times.append(t3 - t0)
                phases.append((t1 - t0, t2 - t1, t3 - t2))
        if rank == 0:
            f, b, o = (sum(p[k] for p in phases) / len(phases) * 1e3 for k in range(3))
            rows = batch['codes'].shape[0] * (batch['codes'].shape[1] + batch['tokens'].shape[1])
            print(f'{name} (dropout {dropout}): {sum(times) / len(times) * 1e3:.1f} ms/step (fwd {f:.1f}, bwd {b:.1f}, '
                  f'allreduce tail + clip + AdamW {o:.1f}) world={world} last batch rows={rows} loss={float(loss.detach()):.3f} '
                  f'peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)


if __name__ == '__main__':
    kv = dict(a.split('=') for a in sys.argv[1:])
    main(int(kv.get('steps', 5)), float(kv.get('dropout', 0.0)), int(kv.get('attn_bwd', 0)), kv.get('only', ''))
