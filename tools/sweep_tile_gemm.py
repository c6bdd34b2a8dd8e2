"""Developer tool: the tile GEMM (vh_linear_ex) over M for the four (N, K) of a 12L/512d layer, with bias + residual and
plain — TFLOP/s per shape (profiles/r3_tile_gemm_sweep_*.log).   python tools/sweep_tile_gemm.py"""
import sys, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from valle2_amd import _lib, kernels as K
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, Kd in [(512, 512), (2048, 512), (512, 2048), (1536, 512)]:
    for M in [8192, 16384, 32768, 65536]:
        a = torch.randn(M, Kd, device=dev); w = torch.randn(N, Kd, device=dev) * 0.05
        b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); out = torch.empty(M, N, device=dev)
        us = timeit(lambda: K.linear_ex(a, w, bias=b, residual=res, out=out))
        us2 = timeit(lambda: K.linear_ex(a, w, out=out))
        fl = 2.0 * M * N * Kd
        print(f'N={N:5d} K={Kd:5d} M={M:6d} tiles={M//128*N//128:5d}  bias+res {us:7.1f} us {fl/us/1e6:6.1f} TF | plain {us2:7.1f} us {fl/us2/1e6:6.1f} TF', flush=True)
