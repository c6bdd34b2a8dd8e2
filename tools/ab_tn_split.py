"""Developer tool: vh_gemm_tn with its contraction split aimed at 512 / 384 / 256 workgroups (knob VH_TUNE_TN_WGS) on the
weight-gradient shapes of configs[3] (profiles/r3_ab_tn_split.log).   python tools/ab_tn_split.py"""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from valle2_amd import _lib, kernels as K
dev='cuda'
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (16320, 10240):
    for n, k in ((512, 512), (1536, 512), (2048, 512), (512, 2048)):
        dy = torch.randn(M, n, device=dev); x = torch.randn(M, k, device=dev); dw = torch.empty(n, k, device=dev)
        row = []
        for wgs in (512, 384, 256):
            _lib.lib().vh_set_tuning(11, wgs)
            us = timeit(lambda: K.gemm_tn(dy, x, out=dw))
            row.append(f'{wgs}: {us:6.1f} us {2.0*M*n*k/us/1e6:6.1f} TF')
        _lib.lib().vh_set_tuning(11, 0)
        print(f'M={M} dW {n}x{k}: ' + ' | '.join(row), flush=True)
