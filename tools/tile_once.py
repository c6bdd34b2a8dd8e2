"""Run the four large-M GEMM shapes of a 12L/512d layer a few times (for rocprofv3 --pmc passes)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import kernels as K  # noqa: E402

M = 32768
for name, N, Kk in (('qkv', 1536, 512), ('out', 512, 512), ('ffn1', 2048, 512), ('ffn2', 512, 2048)):
    a = torch.randn(M, Kk, device='cuda')
    w = 0.02 * torch.randn(N, Kk, device='cuda')
    bias = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda')
    o = torch.empty(M, N, device='cuda')
    for _ in range(6):
        K.linear(a, w, bias, res, out=o)
        torch.addmm(bias, a, w.t(), out=o)
torch.cuda.synchronize()
