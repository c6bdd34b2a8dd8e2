"""Three launches of the perf-mode attention at the NAR-stage shape (64 x 1024, full mask) — or `prompt` for the prompt-pass shape
— for counter passes (tools/pmc_attn16.sh)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import kernels as K  # noqa: E402
from valle2_amd._lib import h16_dtype, lib  # noqa: E402

prompt = len(sys.argv) > 1 and sys.argv[1] == 'prompt'
B, h, T, mode, xl = (32, 8, 1024, K.MASK_PREFIX, 256) if prompt else (64, 8, 1024, K.MASK_FULL, 0)
g = torch.Generator().manual_seed(0)
d = h * 64
q = (torch.randn(B * T, d, generator=g) * K.Q16_PRESCALE).to(h16_dtype()).cuda()
kc = torch.randn(B, h, T, 64, generator=g).to(h16_dtype()).cuda()
vc = torch.randn(B, h, T, 64, generator=g).to(h16_dtype()).cuda()
out = torch.zeros(B * T, d, device='cuda', dtype=h16_dtype())
for _ in range(3):
    rc = lib().vh_attn_rows_bf16(q.data_ptr(), d, kc.data_ptr(), vc.data_ptr(), out.data_ptr(), d, B, h, T, T, T, mode, xl, None, None,
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
