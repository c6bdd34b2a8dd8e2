"""Data parallelism of the path: one process per GPU over torch.distributed
("nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Inference shards the utterance batch across ranks and needs NO collective on the data path — only
the barrier / max-time reduction of the benchmark.  Training adds one exchange per optimizer step:
a sum-all-reduce of fp32 gradients (the implicit Lightning-DDP all-reduce of the reference,
valle/train_model.py:28-35), done here in a few large flat buckets because xGMI is point-to-point
(7 links per GPU): bucket size is chosen per link, not per switch.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init_distributed(backend: str | None = None, device: torch.device | None = None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (no-op for world size 1)."""
    rank, _, world = env_world()
    if world == 1 or dist.is_initialized():
        return rank, world
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    kwargs = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
    dist.init_process_group(backend, **kwargs)
    return rank, world


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of `n_items` independent utterances for `rank` (first
    n_items % world ranks get one more).  Shards are disjoint and cover range(n_items)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def max_over_ranks(value: float, device='cpu') -> float:
    """The slowest rank's time (what the benchmark reports)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_mean_(tensors, bucket_bytes: int = 64 << 20):
    """In-place mean over ranks of a list of same-dtype tensors (gradients), flattened into
    buckets of about `bucket_bytes` so each collective is large (DDP semantics: sum ÷ world)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        bucket, size = [], 0

    for t in tensors:
        bucket.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
