"""The backward-pass weight products at configs[3] training shapes on the hand-written kernels — dX = dY W as
vh_transpose + vh_linear_ex (NT LDS-DMA tile kernel), dW = dY^T X as vh_gemm_tn — with torch.matmul (rocBLAS /
hipBLASLt) timed beside them as a yardstick only (developer tool; the product never calls a library GEMM)."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import kernels as K  # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    M, d, dff, B, T, h = 16144, 512, 2048, 16, 1009, 8
    tp = (T + 3) // 4 * 4
    cases = {}
    for name, n, k in (('dX qkv', 3 * d, d), ('dX ffn1', dff, d), ('dX ffn2', d, dff), ('dX out', d, d)):
        dy, w, x = torch.randn(M, n, device=DEV), torch.randn(n, k, device=DEV), torch.randn(M, k, device=DEV)
        dx, dw = torch.empty(M, k, device=DEV), torch.empty(n, k, device=DEV)
        flop = 2 * M * n * k
        cases[name] = (flop, lambda dy=dy, w=w, dx=dx: K.linear_ex(dy, K.transpose(w), out=dx), lambda dy=dy, w=w: torch.matmul(dy, w))
        cases[name.replace('dX', 'dW')] = (flop, lambda dy=dy, x=x, dw=dw: K.gemm_tn(dy, x, out=dw),
                                           lambda dy=dy, x=x: torch.matmul(dy.t(), x))
    for name, (flop, mine, lib) in cases.items():
        a = statistics.median(timeit(mine) for _ in range(3))
        b = statistics.median(timeit(lib) for _ in range(3))
        print(f'{name:14s} mine {a:8.1f} us ({flop / a / 1e6:6.1f} TF)   library {b:8.1f} us ({flop / b / 1e6:6.1f} TF)', flush=True)
    if 'attn' not in sys.argv:
        return
    cases = {}
    q = torch.randn(B * T, d, device=DEV)
    qh = q.view(B, T, h, 64).permute(0, 2, 1, 3)
    k_ = torch.randn(B, h, T, 64, device=DEV)
    P = torch.randn(B, h, T, tp, device=DEV)[..., :T]
    o64 = torch.empty(B, h, T, 64, device=DEV)
    Pc = P.contiguous()
    f1, f2 = 2 * B * h * T * T * 64, 2 * B * h * T * T * 64
    cases['S = Q K^T'] = (f1, lambda: K.gemm(qh, k_, P), lambda: torch.matmul(qh, k_.transpose(-1, -2)))
    cases['dQ = dS K'] = (f2, lambda: K.gemm(P, k_, o64, b_kmajor=True), lambda: torch.matmul(Pc, k_))
    cases['dV = P^T dO'] = (f2, lambda: K.gemm(P, k_, o64, a_kmajor=True, b_kmajor=True),
                            lambda: torch.matmul(Pc.transpose(-1, -2), k_))
    for name, (flop, mine, lib) in cases.items():
        a = statistics.median(timeit(mine) for _ in range(3))
        b = statistics.median(timeit(lib) for _ in range(3))
        print(f'{name:14s} mine {a:8.1f} us ({flop / a / 1e6:6.1f} TF)   rocBLAS {b:8.1f} us ({flop / b / 1e6:6.1f} TF)', flush=True)


if __name__ == '__main__':
    main()
