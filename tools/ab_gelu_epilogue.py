"""A/B of the FeedForward's training epilogues at configs[3] shapes: forward linear_1 with GELU + the pre-activation kept
(VH_ACT_GELU_ERF + pre_out) against GELU + its derivative kept (VH_ACT_GELU_ERF_D); backward dX of linear_2 through
gelu'(pre) (VH_ACT_GELU_BWD) against a multiply by the kept derivative (VH_ACT_MUL).   python tools/ab_gelu_epilogue.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from valle2_amd import kernels as K
dev = 'cuda'


def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (16320, 10240):
    d, dff = 512, 2048
    x = torch.randn(M, d, device=dev); w1 = torch.randn(dff, d, device=dev) * 0.05; b1 = torch.randn(dff, device=dev)
    pre = torch.empty(M, dff, device=dev); hid = torch.empty(M, dff, device=dev)
    dy = torch.randn(M, d, device=dev); w2t = torch.randn(dff, d, device=dev) * 0.05
    dpre = torch.empty(M, dff, device=dev); cs = torch.zeros(dff, device=dev)
    fl = 2.0 * M * d * dff
    for rep in range(2):
        a = timeit(lambda: K.linear_ex(x, w1, bias=b1, out=hid, pre_out=pre, act=K.ACT_GELU))
        b = timeit(lambda: K.linear_ex(x, w1, bias=b1, out=hid, pre_out=pre, act=K.ACT_GELU_D))
        c = timeit(lambda: K.linear_ex(dy, w2t, residual=pre, out=dpre, act=K.ACT_GELU_BWD, colsum=cs))
        e = timeit(lambda: K.linear_ex(dy, w2t, residual=pre, out=dpre, act=K.ACT_MUL, colsum=cs))
        print(f'M={M}: forward GELU+pre {a:6.1f} us ({fl/a/1e6:5.1f} TF) | GELU+derivative {b:6.1f} us ({fl/b/1e6:5.1f} TF) || '
              f"backward gelu'(pre) {c:6.1f} us ({fl/c/1e6:5.1f} TF) | x derivative {e:6.1f} us ({fl/e/1e6:5.1f} TF)", flush=True)
