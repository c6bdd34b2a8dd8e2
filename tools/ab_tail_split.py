"""A/B of vh_linear_ex's tail split (knob VH_TUNE_TAIL_SPLIT: 0 = default, 1 = never): the N = 512 products of a
configs[3] training step (M = 16 x 640 positions -> 320 tiles on 256 CUs) and neighbours, time per call + TFLOP/s.
    python tools/ab_tail_split.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from valle2_amd import _lib, kernels as K

dev = torch.device('cuda:0')
TAIL = 10


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


for M, N, Kd in [(8192, 512, 512), (10240, 512, 512), (10240, 512, 1536), (10240, 512, 2048), (10240, 2048, 512),
                 (10240, 1536, 512), (12000, 512, 2048), (16256, 512, 2048), (20480, 512, 512), (9000, 512, 512),
                 (34816, 512, 2048)]:
    a = torch.randn(M, Kd, device=dev)
    w = torch.randn(N, Kd, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    ref = torch.addmm(b, a, w.t()) + res
    row = []
    for knob in (1, 0):
        _lib.lib().vh_set_tuning(TAIL, knob)
        us = timeit(lambda: K.linear_ex(a, w, bias=b, residual=res, out=out))
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        row.append((us, err))
    _lib.lib().vh_set_tuning(TAIL, 0)
    tiles = (M + 127) // 128 * (N // 128)
    fl = 2.0 * M * N * Kd
    print(f'M={M:6d} N={N:5d} K={Kd:5d} tiles={tiles:5d} ws={_lib.lib().vh_linear_ex_ws_bytes(M, N, Kd) >> 20:3d}MiB  '
          f'whole {row[0][0]:7.1f} us {fl / row[0][0] / 1e6:6.1f} TF err {row[0][1]:.1e} | '
          f'split {row[1][0]:7.1f} us {fl / row[1][0] / 1e6:6.1f} TF err {row[1][1]:.1e}', flush=True)
