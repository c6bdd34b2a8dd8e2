"""Two data-parallel ranks of `train()` on ONE GPU (gloo transport; RCCL needs a device per rank): the
flat-bucket gradient all-reduce launched from backward hooks, the 1/world mean inside the fused AdamW
kernel and the training loop, end to end on the real kernels."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), VALLE2_DIST_BACKEND='gloo', VALLE2_FORCE_DEVICE='0')
    import tempfile
    os.chdir(tempfile.mkdtemp())
    import torch.distributed as dist
    from valle2_amd import synth
    from valle2_amd.config import ConfigValle
    from valle2_amd.train_model import train
    cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=256, num_layers=2, dropout=0.0, norm='LayerNorm',
                      lr=1e-3, max_steps=3, grad_accum=2, batch_size=2, log_every_n_steps=100, seed=5)
    batches = [synth.synth_ar_batch(cfg, 2, tok_range=(4, 8), code_range=(10, 20), seed=10 * i + rank) for i in range(6)]
    model, losses = train(cfg, 'ValleAR', batches=batches, log=lambda *_: None)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().cpu()
    q.put((rank, losses, flat.sum().item(), flat.abs().sum().item(), flat[::997].tolist()))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _bad_id_worker(rank, world, port, q, dropout):
    """Five optimizer steps by hand; at step 2 rank 1's batch carries a text id beyond the table, already resident on
    the device (so only the kernels can see it).  Every rank must raise at THAT step, in `sync_touched`, skip it, and
    go on in lockstep; with dropout > 0 the ranks also draw different fields (dropout.set_rank)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), VALLE2_DIST_BACKEND='gloo', VALLE2_FORCE_DEVICE='0')
    import tempfile
    os.chdir(tempfile.mkdtemp())
    import torch.distributed as dist
    from valle2_amd import dp, dropout as drp, get_model_class, synth
    from valle2_amd.config import ConfigValle
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dp.init_distributed(device=dev)
    cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=256, num_layers=2, dropout=dropout, norm='LayerNorm',
                      lr=1e-3, batch_size=2, seed=5)
    torch.manual_seed(cfg.seed)
    drp.set_rank(rank)
    model = get_model_class('ValleAR')(cfg).to(dev).train()
    opt = model.configure_optimizers()['optimizer']
    red = dp.GradReducer(opt.flat_grad, opt.slots, bucket_bytes=1 << 16)
    raised, sites = [], []
    drp.RECORD = []
    for step in range(5):
        batch = synth.synth_ar_batch(cfg, 2, tok_range=(4, 8), code_range=(10, 20), seed=10 * step + rank)
        batch = {k: (v if k.endswith('_lens') else v.to(dev)) for k, v in batch.items()}
        if step == 2 and rank == 1:
            batch['tokens'][0, 1] = cfg.vocab_size + 7
        model.training_step(batch).backward()
        red.finish()
        try:
            opt.sync_touched()
            opt.step(grad_scale=1.0 / world, max_norm=1.0, zero_grad=True)
        except IndexError:
            raised.append(step)
    sites = sorted({r['site'] >> 48 for r in drp.RECORD})
    drp.RECORD = None
    opt.check_errors()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().cpu()
    q.put((rank, raised, opt.steps, flat[::499].tolist(), bool(torch.isfinite(flat).all()), sites))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, target=_worker, *extra):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_ranks_stay_in_lockstep_and_differ_from_one_rank():
    two = _run(2)
    (_, l0, s0, a0, v0), (_, l1, s1, a1, v1) = two
    assert l0 != l1                                   # different data per rank ...
    assert v0 == v1 and s0 == s1 and a0 == a1         # ... identical parameters after every exchange
    one = _run(1)
    assert one[0][4] != v0                            # and not what a single rank alone arrives at
    assert all(torch.isfinite(torch.tensor(l0))) and len(l0) == 6


@pytest.mark.parametrize('dropout', [0.0, 0.1])
def test_a_bad_id_on_one_rank_is_raised_by_every_rank_in_the_same_step(dropout):
    (r0, raised0, steps0, v0, fin0, s0), (r1, raised1, steps1, v1, fin1, s1) = _run(2, _bad_id_worker, dropout)
    assert raised0 == raised1 == [2]                  # both ranks, that step, nothing re-raised later
    assert steps0 == steps1 == 4 and fin0 and fin1    # the flagged step was skipped everywhere
    assert v0 == v1                                   # replicas identical after the exchange, dropout or not
    assert (s0, s1) == ([0], [1])                     # ...although the ranks drew different fields (rank in the site id)
