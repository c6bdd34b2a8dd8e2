"""Developer tool: VH_TUNE_TILE_PERSIST A/B on the tile GEMM (vh_linear_ex) — the 512 resident workgroups walking the
tiles and requesting the next tile's first slab in the middle of the epilogue (knob 15 = 1) against one tile per
workgroup (0) — alternating, same buffers, with a bit-identity check; shapes with more than 512 whole tiles and no
K-sliced tail (others do not take the persistent path).   python tools/ab_tile_persist.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import _lib, kernels as K  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


L = _lib.lib()
for N, Kd in [(2048, 512), (1536, 512), (512, 512), (512, 2048)]:
    for M in [16384, 16320, 32768, 65536]:
        tiles = -(-M // 128) * (N // 128)
        a = torch.randn(M, Kd, device=dev)
        w = torch.randn(N, Kd, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        outs = [torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)]
        line = f'N={N:5d} K={Kd:5d} M={M:6d} tiles={tiles:5d}'
        for label, kw in (('bias+res', dict(bias=b, residual=res)), ('plain', {}), ('gelu', dict(bias=b, act=K.ACT_GELU))):
            best = {0: 1e9, 1: 1e9}
            for _ in range(3):
                for knob in (0, 1):
                    L.vh_set_tuning(15, knob)
                    best[knob] = min(best[knob], timeit(lambda: K.linear_ex(a, w, out=outs[knob], **kw)))
            L.vh_set_tuning(15, 0)
            same = torch.equal(outs[0], outs[1])
            fl = 2.0 * M * N * Kd
            line += (f' | {label}: {fl / best[0] / 1e6:6.1f} -> {fl / best[1] / 1e6:6.1f} TF ({(best[0] / best[1] - 1) * 100:+.1f} %)'
                     f'{"" if same else " DIFFERENT RESULTS"}')
        print(line, flush=True)
