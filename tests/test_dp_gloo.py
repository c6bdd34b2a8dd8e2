"""`gloo` tests (CPU, world sizes 2, 4 and 8) of the N>1 path: utterance sharding, the benchmark's
max-over-ranks reduction and the bucketed gradient mean / overlapped reducer used by the training path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from valle2_amd import dp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, w = dp.init_distributed('gloo')
    assert (r, w) == (rank, world)
    shard = list(dp.shard_range(7, rank, world))
    slow = dp.max_over_ranks(1.0 + rank)
    assert dp.host_group() is None                        # the default group is gloo already
    side = dist.new_group(backend='gloo')                 # what bench.py times through when the default group is RCCL
    assert dp.max_over_ranks(2.0 * rank, group=side) == 2.0 * (world - 1)
    dist.barrier(group=side)
    g = torch.Generator().manual_seed(rank)
    grads = [torch.randn(5, 3, generator=g), torch.randn(11, generator=g), torch.randn(2, 2, 2, generator=g)]
    mine = [t.clone() for t in grads]
    dp.allreduce_mean_(grads, bucket_bytes=64)          # tiny buckets → several collectives
    out.put((rank, shard, slow, [t.tolist() for t in mine], [t.tolist() for t in grads]))
    dist.barrier()
    dist.destroy_process_group()


def _reducer_worker(rank, world, port, out, algo='ring'):
    """GradReducer on a toy model: flat gradient views, hooks firing during backward, two buckets,
    gradient accumulation with the exchange on the last micro-batch only."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), VALLE2_ALLREDUCE=algo)
    dp.init_distributed('gloo')
    from valle2_amd.optim import flat_layout
    torch.manual_seed(0)                                   # same weights on both ranks
    model = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.Tanh(), torch.nn.Linear(10, 5),
                                torch.nn.Tanh(), torch.nn.Linear(5, 3))
    params = list(model.parameters())
    slots, total = flat_layout(params)
    flat = torch.zeros(total)
    for p, off, n in slots:
        p.grad = flat[off:off + n].view_as(p)
    red = dp.GradReducer(flat, slots, bucket_bytes=160)    # 40 floats per bucket → several buckets
    assert red.algorithm == algo and all((e - s) % 64 == 0 for s, e, _ in red.buckets)
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(4, 6, generator=g) for _ in range(2)]
    for i, x in enumerate(xs):                             # 2 micro-batches, exchange on the last
        red.enabled = i == 1
        model(x).square().sum().backward()
    launched_in_backward = sum(w is not None for w in red._work)
    red.finish()
    # reference: plain autograd on fresh copies, summed over micro-batches
    ref_model = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.Tanh(), torch.nn.Linear(10, 5),
                                    torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref_model.load_state_dict(model.state_dict())
    for x in xs:
        ref_model(x).square().sum().backward()
    local = [p.grad.clone() for p in ref_model.parameters()]
    out.put((rank, len(red.buckets), launched_in_backward, [t.tolist() for t in local],
             [p.grad.tolist() for p in params], all(p.grad.data_ptr() == red.view_of[id(p)].data_ptr() for p in params)))
    dist.barrier()
    dist.destroy_process_group()


def _branch_worker(rank, world, port, out):
    """Each rank back-propagates through a DIFFERENT head (as NAR ranks that drew different stages would):
    different parameters receive gradients on each rank, with one small bucket per parameter.  Buckets must
    still leave in index order on every rank, so the collectives pair up slice by slice."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dp.init_distributed('gloo')
    from valle2_amd.optim import flat_layout
    torch.manual_seed(0)
    trunk = torch.nn.Linear(6, 8)
    heads = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(world + 1)])   # equal sizes: a swap would go unnoticed
    params = list(trunk.parameters()) + list(heads.parameters())
    slots, total = flat_layout(params)
    flat = torch.zeros(total)
    for p, off, n in slots:
        p.grad = flat[off:off + n].view_as(p)
    red = dp.GradReducer(flat, slots, bucket_bytes=4)       # every parameter its own bucket
    order = []
    launch = red._launch
    red._launch = lambda b: (order.append(b), launch(b))[1]
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(7))
    heads[rank](trunk(x)).square().sum().backward()         # rank r trains head r; the last head nobody
    red.finish()
    ref = {}
    torch.manual_seed(0)
    trunk2 = torch.nn.Linear(6, 8)
    heads2 = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(world + 1)])
    for r in range(world):
        for p in list(trunk2.parameters()) + list(heads2.parameters()):
            p.grad = None
        heads2[r](trunk2(x)).square().sum().backward()
        for i, p in enumerate(list(trunk2.parameters()) + list(heads2.parameters())):
            ref[i] = ref.get(i, 0) + (p.grad if p.grad is not None else torch.zeros_like(p))
    ok = all(torch.allclose(p.grad, ref[i], atol=1e-6) for i, p in enumerate(params))
    out.put((rank, order, len(red.buckets), ok))
    dist.barrier()
    dist.destroy_process_group()


def _ab_worker(rank, world, port, out):
    """The two forms of the exchange on the SAME flat gradient (several buckets, a knob-sized bucket): ring all_reduce and
    reduce-scatter + all-gather must leave the same numbers in every element on every rank, and an AdamW-shaped update
    from them the same parameters."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), VALLE2_BUCKET_MB='0.002')     # ~2 KiB buckets
    dp.init_distributed('gloo')
    from valle2_amd.optim import flat_layout
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(*shp)) for shp in ((33, 17), (100,), (7, 9, 3), (64, 64), (5,), (1025, 3))]
    slots, total = flat_layout(params)
    g = torch.Generator().manual_seed(500 + rank)
    mine = torch.randn(total, generator=g)
    res = {}
    for algo in ('ring', 'rs_ag'):
        os.environ['VALLE2_ALLREDUCE'] = algo
        flat = mine.clone()
        red = dp.GradReducer(flat, slots)
        assert red.algorithm == algo and len(red.buckets) >= 3 and red.bucket_bytes == int(0.002 * (1 << 20))
        assert red.launches_per_step == len(red.buckets) * (2 if algo == 'rs_ag' else 1)
        assert red.bytes_per_step(world) == int(2 * (world - 1) / world * total * 4)
        red.finish()                                        # nothing arrived through hooks: every bucket leaves here, in order
        red.remove()
        res[algo] = flat
    flatp = torch.cat([p.detach().reshape(-1) for p in params])
    upd = {a: flatp[:64] - 1e-3 * (f[:64] / world) / ((f[:64] / world).abs().sqrt() + 1e-8) for a, f in res.items()}
    out.put((rank, res['ring'].tolist(), res['rs_ag'].tolist(), mine.tolist(), torch.equal(upd['ring'], upd['rs_ag'])))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, *extra):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('world', [2, 4, 8])
def test_grad_reducer_launches_buckets_in_index_order_when_ranks_train_different_parameters(world):
    """One bucket per parameter, every rank training a different head (world sizes up to the 8 of a node): the
    collectives must leave in bucket order on every rank and every reduced slice must be the sum over ranks."""
    for rank, order, n_buckets, ok in _spawn(_branch_worker, world):
        assert order == list(range(n_buckets)) and n_buckets == 2 + 2 * (world + 1), (rank, order)
        assert ok, f'rank {rank}: reduced gradients are not the sum over ranks'


@pytest.mark.parametrize('world', [2, 4, 8])
def test_ring_and_reduce_scatter_all_gather_agree(world):
    """VALLE2_ALLREDUCE = ring | rs_ag (and VALLE2_BUCKET_MB) on one flat gradient: the same sums, bit for bit, in every
    element on every rank — gloo adds the ranks' contributions in rank order in both collectives — and therefore the
    same parameters after the update."""
    res = _spawn(_ab_worker, world)
    total = sum(torch.tensor(r[3], dtype=torch.float64) for r in res)
    for rank, ring, rs_ag, _, same_update in res:
        ring, rs_ag = torch.tensor(ring), torch.tensor(rs_ag)
        assert torch.allclose(ring.double(), total, atol=1e-5) and torch.allclose(rs_ag.double(), total, atol=1e-5)
        assert torch.equal(ring, torch.tensor(res[0][1])) and torch.equal(rs_ag, torch.tensor(res[0][2])), rank   # replicas agree
        assert torch.equal(ring, rs_ag) and same_update, f'rank {rank}: the two forms differ'


@pytest.mark.parametrize('algo', ['ring', 'rs_ag'])
@pytest.mark.parametrize('world', [2, 4, 8])
def test_grad_reducer_overlapped_buckets(world, algo):
    res = _spawn(_reducer_worker, world, algo)
    assert len({r[1] for r in res}) == 1 and res[0][1] >= 2 and all(r[5] for r in res)
    assert all(r[2] >= 1 for r in res)                   # at least one bucket left during backward on every rank
    n_params = len(res[0][3])
    for i in range(n_params):
        total = sum(torch.tensor(r[3][i]) for r in res)  # SUM over ranks (the mean is the optimizer's grad_scale)
        for r in res:
            assert torch.allclose(torch.tensor(r[4][i]), total, atol=1e-5), (r[0], i)


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, m0, g0, a0), (r1, s1, m1, g1, a1) = res
    assert sorted(s0 + s1) == list(range(7)) and not set(s0) & set(s1)      # disjoint cover
    assert m0 == m1 == 2.0                                                    # slowest rank wins
    for x0, x1, y0, y1 in zip(g0, g1, a0, a1):
        mean = (torch.tensor(x0) + torch.tensor(x1)) / 2
        assert torch.allclose(torch.tensor(y0), mean) and torch.allclose(torch.tensor(y1), mean)


@pytest.mark.parametrize('n,world', [(32, 1), (32, 8), (7, 4), (3, 8), (0, 2)])
def test_shard_range_properties(n, world):
    shards = [dp.shard_range(n, r, world) for r in range(world)]
    flat = [i for s in shards for i in s]
    assert flat == list(range(n))
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_single_process_is_a_noop():
    assert dp.max_over_ranks(3.5) == 3.5
    t = [torch.ones(3)]
    dp.allreduce_mean_(t)
    assert torch.equal(t[0], torch.ones(3))
