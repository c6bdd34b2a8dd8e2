"""RCCL on the one GPU a lease of this project ever had: a process group of ONE rank on the `nccl` (= RCCL) backend,
created as the multi-GPU path creates it (`device_id`), carrying the training path's real exchange — `GradReducer`'s
asynchronous in-place all-reduces of flat-gradient buckets, launched from backward hooks on RCCL's stream while the
hand-written kernels (launched through ctypes on torch's current stream) keep producing gradients — followed by the
fused AdamW step.  With one rank the sum is the identity: the exchanged gradient of the first step must be the
unexchanged one (to the rounding noise of the atomics in the column-sum / scatter kernels) and the parameters after
three steps must stay within a few Adam steps of the run without exchange.  What the test adds is that RCCL loads,
initialises (never seen before round 3 in this project) and runs its collectives beside these streams on this image; a
one-rank sum cannot expose a missing stream dependency — that needs the driver's multi-GPU node."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import json, os, sys, tempfile, torch
sys.path.insert(0, {repo!r})
os.chdir(tempfile.mkdtemp())
import torch.distributed as dist
from valle2_amd import ConfigValle, dp, get_model_class, synth
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)                 # as dp.init_distributed does for world > 1
cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=512, num_layers=2, dropout=0.0, norm='LayerNorm', seed=5)
batches = [synth.synth_ar_batch(cfg, 3, tok_range=(4, 8), code_range=(10, 20), seed=s) for s in range(3)]
batches = [{{k: (v if k.endswith('_lens') else v.to(dev)) for k, v in b.items()}} for b in batches]
finals, grads, launched_x, handles = [], [], 0, 0
for exchange in (True, False):
    torch.manual_seed(0)
    model = get_model_class('ValleAR')(cfg).to(dev).train()
    opt = model.configure_optimizers()['optimizer']
    red = dp.GradReducer(opt.flat_grad, opt.slots, bucket_bytes=1 << 16, algorithm={algo!r})
    if exchange:
        type(red).active = property(lambda self: self.enabled)   # one rank: force the exchange the world > 1 path makes
    launched = 0
    for b in batches:
        torch.manual_seed(1)                                     # same position dropout in both runs
        model.training_step(b).backward()
        launched += sum(w is not None for w in red._work)
        handles = max(handles, max((len(w) for w in red._work if w is not None), default=0))
        red.finish()
        if len(grads) < 2 and b is batches[0]:
            opt.gather_grads()
            grads.append(opt.flat_grad.clone())
        opt.step(grad_scale=1.0, max_norm=1.0, zero_grad=True)
    opt.check_errors()
    finals.append(opt.flat_param.clone())
    red.remove()
    if exchange:
        del type(red).active
        type(red).active = property(lambda self: self.enabled and dist.is_available() and dist.is_initialized()
                                    and dist.get_world_size() > 1)
    buckets = len(red.buckets)
    if exchange:
        launched_x = launched
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(t)
dist.barrier()
gerr = float((grads[0] - grads[1]).abs().max() / grads[1].abs().max())
perr = float((finals[0] - finals[1]).abs().max())
print(json.dumps({{'backend': dist.get_backend(), 'grad_rel_err': gerr, 'param_abs_err': perr, 'buckets': buckets,
                  'launched_in_backward': launched_x, 'handles_per_bucket': handles, 'algorithm': {algo!r}, 'allreduce_ok': bool(torch.equal(t.cpu(), torch.arange(8.)))}}), flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize('algo', ['ring', 'rs_ag'])
def test_rccl_single_rank_carries_the_gradient_exchange(tmp_path, algo):
    """ring: one in-place all_reduce per bucket.  rs_ag (VALLE2_ALLREDUCE=rs_ag, the all-links form): reduce_scatter_tensor whose
    output IS a slice of its input, then all_gather_into_tensor back into the same bucket — this proves RCCL (not only gloo)
    accepts that aliasing and that both work handles of a bucket are kept and waited for."""
    script = tmp_path / 'rccl1.py'
    script.write_text(_SCRIPT.format(repo=str(REPO), algo=algo))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('VALLE2_DIST_BACKEND', 'VALLE2_FORCE_DEVICE')}
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([line for line in out.stdout.splitlines() if line.startswith('{')][0])
    assert res['backend'] == 'nccl' and res['allreduce_ok'], res
    assert res['buckets'] >= 3 and res['launched_in_backward'] >= 1, res      # buckets left while backward was running
    assert res['algorithm'] == algo and res['handles_per_bucket'] == (2 if algo == 'rs_ag' else 1), res
    assert res['grad_rel_err'] < 1e-5, res                    # the exchanged gradient IS the gradient
    assert res['param_abs_err'] < 1e-3, res                   # lr = 1e-4: a few Adam steps (sign flips on noise-level gradients)
