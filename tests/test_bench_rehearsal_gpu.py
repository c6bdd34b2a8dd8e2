"""Rehearsal of `bench.py --gpus N` on ONE GPU: N ranks spawned by bench.py itself, rendezvous on 127.0.0.1, the
timed region's barrier / max-over-ranks over the host-side group, and the training leg's gradient exchange — every
piece of the multi-GPU path except RCCL itself, which wants one device per rank (transport: gloo,
VALLE2_DIST_BACKEND; every rank on device 0, VALLE2_FORCE_DEVICE).

World size 4, not 8: a GPU box of this pool allows at most 6 processes on its card at once (this pytest process is one
of them), and a node's 8 ranks may only be started by the driver.  The spawning and rendezvous of 8 ranks is covered
without a GPU by tests/test_host_cpu.py::test_bench_gpus_flag_spawns_one_rank_per_gpu, the reducer at world 8 by
tests/test_dp_gloo.py."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_bench_small_four_ranks_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(VALLE2_DIST_BACKEND='gloo', VALLE2_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--small', '--gpus', '4', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline', '--no-roofline'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1, out.stdout                       # rank 0 prints the ONE line
    r = lines[0]
    assert r['n_gpus'] == 4 and r['steps'] == 2 and r['warmup'] == 1 and r['scaling'] == 'weak'
    # the default group's own account of itself (gloo in this rehearsal: RCCL wants a device per rank)
    assert r['distributed'] == {'backend': 'gloo', 'group_world_size': 4, 'allreduce_count': 4, 'env_world_size': 4}
    assert r['rccl_ranks_seen'] is None
    assert r['value'] > 0 and r['config']['sharding'] == 'utterance-batch x4'
    train = r['train']
    assert 'error' not in train, train
    assert train['ar_ms_per_step'] > 0 and train['nar_ms_per_step'] > 0
    assert train['ar_allreduce_bytes'] > 0 and train['nar_allreduce_bytes'] > 0     # the exchange ran (world > 1)
    assert 0 < train['ar_frac'] < 1 and 0 < train['nar_frac'] < 1
    assert 'beams' not in r and 'perf_mode' not in r           # secondary objects belong to the N = 1 line
