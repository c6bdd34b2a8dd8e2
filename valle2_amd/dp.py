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
    # VALLE2_DIST_BACKEND=gloo: rehearsal of several ranks on one GPU (RCCL wants one device per rank)
    backend = os.environ.get('VALLE2_DIST_BACKEND') or backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    kwargs = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
    dist.init_process_group(backend, **kwargs)
    return rank, world


def ranks_seen(device=None) -> dict:
    """What the DEFAULT process group itself reports, for the benchmark's JSON line: its backend name, its world size,
    and the result of a sum all-reduce of one 1 per rank on `device` through it — on the `nccl` backend that number is how
    many ranks RCCL really connected (it comes from the collective, not from the environment).  One rank: no group."""
    if not (dist.is_available() and dist.is_initialized()):
        return {'backend': None, 'group_world_size': 1, 'allreduce_count': 1}
    backend = dist.get_backend()
    one = torch.ones(1, dtype=torch.float32, device=device if backend == 'nccl' else 'cpu')
    dist.all_reduce(one)
    return {'backend': backend, 'group_world_size': dist.get_world_size(), 'allreduce_count': int(one.item())}


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of `n_items` independent utterances for `rank` (first
    n_items % world ranks get one more).  Shards are disjoint and cover range(n_items)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def host_group():
    """A second process group over `gloo` (CPU, TCP on the rendezvous address) for the benchmark's barrier and
    max-time reduction: the inference path has no collective, so its timing should not depend on RCCL coming up.
    None when there is one rank or the default group is gloo already."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    if dist.get_backend() == 'gloo':
        return None
    return dist.new_group(backend='gloo')


def max_over_ranks(value: float, device='cpu', group=None) -> float:
    """The slowest rank's time (what the benchmark reports).  `group`: a host_group() (the tensor then lives on
    the CPU); default: the default group, tensor on `device`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device='cpu' if group is not None else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
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


def exchange_knobs():
    """(algorithm, bucket bytes) of the gradient exchange from the environment:
      VALLE2_ALLREDUCE = ring  (default) one in-place all_reduce per bucket — RCCL picks its own algorithm, a ring at these
                               sizes: 2 (N-1)/N of the bucket over ONE link pair per GPU;
                       = rs_ag reduce_scatter_tensor + all_gather_into_tensor on the same slice, both in place (rank r owns
                               chunk r of the bucket): the all-links form SURVEY.md section 5 asks to compare — every GPU
                               sends 1/N of the bucket to each of its N-1 peers over that peer's own xGMI link, twice;
      VALLE2_BUCKET_MB = 64    (default) bucket size in MiB."""
    algo = os.environ.get('VALLE2_ALLREDUCE', 'ring').strip().lower()
    if algo not in ('ring', 'rs_ag'):
        raise ValueError(f"VALLE2_ALLREDUCE={algo!r}: expected 'ring' or 'rs_ag'")
    mb = float(os.environ.get('VALLE2_BUCKET_MB', '64'))
    if not mb > 0:
        raise ValueError(f'VALLE2_BUCKET_MB={mb}: must be positive')
    return algo, int(mb * (1 << 20))


class GradReducer:
    """Sum-all-reduce of a flat gradient buffer, overlapped with backward.

    `slots` = [(param, offset, numel)] laid out in the order backward produces gradients
    (`optim.flat_layout`).  The buffer is cut into buckets of about `bucket_bytes` on parameter
    boundaries; a post-accumulate hook on every parameter counts arrivals and starts a bucket's
    all-reduce asynchronously on the communicator's stream while backward keeps computing the earlier
    layers.  Buckets are launched STRICTLY IN INDEX ORDER (bucket b only after b-1, as DDP does): which
    parameters receive a gradient can differ between ranks (NAR trains one randomly drawn stage per
    step), and collectives issued in different orders on different ranks would pair mismatched slices
    or hang.  `finish()` starts whatever is left (parameters that received no gradient), in order,
    waits, and returns; the 1/world mean is left to the optimizer kernel (`grad_scale`).
    Few large messages: xGMI is point-to-point, 7 links per GPU, so a bucket is sized per link
    (64 MB default ≈ 3 buckets for the 156 MB AR model), not for a switch.
    `algorithm` / `bucket_bytes` default to the environment's knobs (`exchange_knobs`): 'ring' = one all_reduce per
    bucket, 'rs_ag' = reduce-scatter + all-gather of the same slice.  Both leave the SUM over ranks in every element of
    the bucket; a bucket whose length the world size does not divide (slots are padded to 64 floats, so only world
    sizes that do not divide 64) falls back to all_reduce.  `bytes_per_step` / `launches_per_step`: what one exchange
    of the whole buffer moves per rank (for the bench line).
    """

    def __init__(self, flat_grad: torch.Tensor, slots, bucket_bytes: int | None = None, algorithm: str | None = None):
        env_algo, env_bytes = exchange_knobs()
        self.algorithm = algorithm or env_algo
        if self.algorithm not in ('ring', 'rs_ag'):
            raise ValueError(f'GradReducer: unknown algorithm {self.algorithm!r}')
        bucket_bytes = env_bytes if bucket_bytes is None else bucket_bytes
        self.bucket_bytes = bucket_bytes
        self.flat = flat_grad
        self.enabled = True
        self.buckets = []          # (start, end, n_params)
        self.bucket_of = {}
        self.view_of = {id(p): flat_grad[off:off + n].view_as(p) for p, off, n in slots}
        start, count = 0, 0
        end = 0
        for i, (p, off, n) in enumerate(slots):
            self.bucket_of[id(p)] = len(self.buckets)
            count += 1
            # a bucket ends where the next slot begins (slots are padded — optim.ALIGN = 64 floats — so bucket lengths
            # are multiples of 64 and every power-of-two world size divides them: what rs_ag's equal chunks need)
            end = slots[i + 1][1] if i + 1 < len(slots) else flat_grad.numel()
            if (end - start) * flat_grad.element_size() >= bucket_bytes:
                self.buckets.append((start, end, count))
                start, count = end, 0
        if count:
            self.buckets.append((start, end, count))
        if self.buckets:                                   # the padded tail belongs to the last bucket
            s0, _, c0 = self.buckets[-1]
            self.buckets[-1] = (s0, flat_grad.numel(), c0)
        self._arrived = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._next = 0                 # first bucket not launched yet
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p, _, _ in slots]

    @property
    def active(self):
        return self.enabled and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        buf = self.flat[s:e]
        world = dist.get_world_size()
        if self.algorithm == 'rs_ag' and (e - s) % world == 0:
            # in place on the bucket: rank r reduces chunk r (its slice of the send buffer IS its receive buffer), then
            # every rank gathers the N reduced chunks back into the same bucket.  The two collectives of a bucket — and
            # the buckets among themselves — run in issue order on the communicator's stream (RCCL); gloo's worker
            # threads give no such order, so there the scatter is waited for before the gather is issued.
            chunk = (e - s) // world
            mine = buf[dist.get_rank() * chunk:(dist.get_rank() + 1) * chunk]
            rs = dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, async_op=True)
            if dist.get_backend() != 'nccl':
                rs.wait()
            # BOTH handles are kept and waited for in finish(): an error or a timeout of the reduce-scatter must surface
            # there too, not only the gather's (round-4 advisor finding)
            self._work[b] = (rs, dist.all_gather_into_tensor(buf, mine, async_op=True))
        else:
            self._work[b] = (dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True),)

    @property
    def launches_per_step(self):
        return len(self.buckets) * (2 if self.algorithm == 'rs_ag' else 1)

    def bytes_per_step(self, world: int) -> int:
        """Bytes one rank SENDS for one exchange of the whole buffer: 2 (N-1)/N of it either way — a ring pushes them
        through one link, rs_ag spreads them over the N-1 links."""
        return int(2 * (world - 1) / max(world, 1) * self.flat.numel() * self.flat.element_size())

    def _on_grad(self, p):
        if not self.active:
            return
        view = self.view_of[id(p)]
        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():   # autograd replaced the view
            view.copy_(p.grad)
            p.grad = view
        self._arrived[self.bucket_of[id(p)]] += 1
        while self._next < len(self.buckets) and self._arrived[self._next] == self.buckets[self._next][2]:
            self._launch(self._next)
            self._next += 1

    def finish(self):
        """Call after backward of the last micro-batch: every bucket reduced when this returns (on
        the current stream for NCCL/RCCL)."""
        if self.active:
            for b in range(self._next, len(self.buckets)):
                self._launch(b)
            for works in self._work:
                for w in works:
                    w.wait()
        self._arrived = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._next = 0

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
