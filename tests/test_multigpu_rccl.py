"""N > 1 on real GPUs over RCCL (skipped on a one-GPU box; the driver's 8-GPU node and any multi-GPU box
run it): `bench.py --gpus 2` starts its own two ranks, and a 2-rank `train()` keeps parameters identical
across ranks while exchanging gradients over the `nccl` (= RCCL) backend with `device_id`."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL wants a device per rank)')]


def _clean_env():
    return {k: v for k, v in os.environ.items()
            if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'VALLE2_DIST_BACKEND', 'VALLE2_FORCE_DEVICE')}


def test_bench_small_two_gpus_spawns_its_own_ranks():
    out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--small', '--gpus', '2', '--steps', '2',
                          '--warmup', '1'], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0)'
    res = lines[0]
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['value'] > 0
    assert res['rccl_ranks_seen'] == 2 and res['distributed']['backend'] == 'nccl'
    assert 'error' not in res.get('train', {}), res.get('train')
    assert res['train']['ar_allreduce_bytes'] > 0


_TRAIN = r'''
import json, os, sys, tempfile, torch
sys.path.insert(0, {repo!r})
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle
from valle2_amd.train_model import train
import torch.distributed as dist
cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=512, num_layers=2, dropout=0.0, norm='AdaptiveLayerNorm',
                  lr=1e-3, max_steps=3, grad_accum=2, batch_size=2, log_every_n_steps=100, seed=5)
model, losses = train(cfg, 'ValleNAR', log=lambda *a: None)
flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
other = flat.clone()
dist.all_reduce(other, op=dist.ReduceOp.MAX)
same = bool(torch.equal(other, flat))
mn = flat.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
if dist.get_rank() == 0:
    print(json.dumps({{'backend': dist.get_backend(), 'world': dist.get_world_size(), 'same': same and bool(torch.equal(mn, flat)),
                      'losses': losses, 'device': str(flat.device)}}), flush=True)
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_train_on_rccl_keeps_parameters_identical(tmp_path):
    script = tmp_path / 'train2.py'
    script.write_text(_TRAIN.format(repo=str(REPO)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    res = json.loads([line for line in outs[0][0].splitlines() if line.startswith('{')][0])
    assert res['backend'] == 'nccl' and res['world'] == 2 and res['same'], res
    assert len(res['losses']) == 6 and all(l == l for l in res['losses'])
