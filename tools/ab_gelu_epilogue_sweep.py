"""Developer tool: the FeedForward training epilogues over M — backward gelu'(pre) against a multiply by the kept derivative,
forward GELU + pre against GELU + derivative (profiles/r3_ab_gelu_epilogue.log).   python tools/ab_gelu_epilogue_sweep.py"""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from valle2_amd import kernels as K
dev='cuda'
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
d, dff = 512, 2048
for M in (8192, 9000, 10240, 10300, 11000, 12288, 13000, 14000, 15000, 16320, 18000, 20480):
    pre = torch.randn(M, dff, device=dev); dy = torch.randn(M, d, device=dev); w2t = torch.randn(dff, d, device=dev) * 0.05
    dpre = torch.empty(M, dff, device=dev); cs = torch.zeros(dff, device=dev)
    x = torch.randn(M, d, device=dev); w1 = torch.randn(dff, d, device=dev) * 0.05; b1 = torch.randn(dff, device=dev); hid = torch.empty(M, dff, device=dev)
    r = []
    for rep in range(2):
        e = timeit(lambda: K.linear_ex(dy, w2t, residual=pre, out=dpre, act=K.ACT_MUL, colsum=cs))
        c = timeit(lambda: K.linear_ex(dy, w2t, residual=pre, out=dpre, act=K.ACT_GELU_BWD, colsum=cs))
        a = timeit(lambda: K.linear_ex(x, w1, bias=b1, out=hid, pre_out=dpre, act=K.ACT_GELU))
        b = timeit(lambda: K.linear_ex(x, w1, bias=b1, out=hid, pre_out=dpre, act=K.ACT_GELU_D))
        r.append((c, e, a, b))
    c, e, a, b = [min(x[i] for x in r) for i in range(4)]
    print(f'M={M:6d} tiles={(M+127)//128*16:5d}: bwd gelu\'(pre) {c:6.1f} -> mul {e:6.1f} ({e-c:+5.1f}) | fwd pre {a:6.1f} -> deriv {b:6.1f} ({b-a:+5.1f}) | net {e-c+b-a:+5.1f} us', flush=True)
