"""Developer tool: the five-product attention backward at a forced number of key chunks (VH_TUNE_ATTN_BWD_CHUNKS), to
calibrate the chunk-count rule of csrc/attention.hip.   python tools/sweep_attn_bwd_chunks.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import _lib, kernels as K  # noqa: E402

DEV = torch.device('cuda:0')


def run(name, B, h, T, mode, xl, kvl, ncs, iters=10):
    d = 64 * h
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B * T, d, generator=g).to(DEV)
    k = torch.randn(B, h, T, 64, generator=g).to(DEV)
    v = torch.randn(B, h, T, 64, generator=g).to(DEV)
    dout = torch.randn(B * T, d, generator=g).to(DEV)
    spec = dict(mode=mode, x_len=xl, kv_len=torch.tensor(kvl, dtype=torch.int32, device=DEV))
    out = torch.empty(B * T, d, device=DEV)
    lse2 = torch.empty(B, h, T, device=DEV)
    K.attn_rows(q, k, v, out, B, h, T, T, lse2=lse2, **spec)
    dqkv = torch.empty(B * T, 3 * d, device=DEV)
    auto = _lib.lib().vh_attn_rows_bwd_chunks(B, h, T, mode)
    for nc in [ncs[0]] + list(ncs):              # (the first pass warms the allocator up and is printed twice)
        _lib.lib().vh_set_tuning(14, nc)
        fn = lambda: K.attn_rows_bwd(q, k, v, out, dout, lse2, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, T, **spec)  # noqa: E731
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        print(f'{name} B={B} h={h} T={T} chunks={nc or f"auto={auto}"}: {(time.perf_counter() - t0) / iters * 1e6:8.1f} us', flush=True)
    _lib.lib().vh_set_tuning(14, 0)


if __name__ == '__main__':
    g = torch.Generator().manual_seed(0)
    kvl = (120 + torch.randint(225, 901, (16,), generator=g)).tolist()
    run('AR ', 16, 8, 1020, K.MASK_PREFIX, 120, kvl, [0, 4, 5, 6, 8])
    run('NAR', 16, 8, 640, K.MASK_FULL, 0, [640] * 16, [0, 3, 4, 5, 6, 7, 10])
    run('big', 2, 16, 2875, K.MASK_FULL, 0, [2875] * 2, [0, 12, 15, 16, 18, 23, 24, 30], iters=4)
    run('b8 ', 8, 8, 1020, K.MASK_PREFIX, 120, kvl[:8], [0, 4, 6, 8])
    run('n32', 32, 8, 640, K.MASK_FULL, 0, [640] * 32, [0, 3, 4, 5])
