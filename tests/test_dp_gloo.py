"""world_size-2 `gloo` tests (CPU) of the N>1 path: utterance sharding, the benchmark's
max-over-ranks reduction and the bucketed gradient mean used by the training path."""
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
    g = torch.Generator().manual_seed(rank)
    grads = [torch.randn(5, 3, generator=g), torch.randn(11, generator=g), torch.randn(2, 2, 2, generator=g)]
    mine = [t.clone() for t in grads]
    dp.allreduce_mean_(grads, bucket_bytes=64)          # tiny buckets → several collectives
    out.put((rank, shard, slow, [t.tolist() for t in mine], [t.tolist() for t in grads]))
    dist.barrier()
    dist.destroy_process_group()


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
