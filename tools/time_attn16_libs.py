"""Time vh_attn_rows_bf16 of several builds of the library in ONE process, alternating (probe builds with parts of the kernel
switched off answer "what bounds it"): tools/time_attn16_libs.py LIB.so [LIB.so ...] [--reps 20] [--rounds 3]."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

P, I = C.c_void_p, C.c_int


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    args = ap.parse_args()
    torch.cuda.init()
    libs = []
    for p in args.libs:
        lib = C.CDLL(str(Path(p).resolve()), mode=C.RTLD_LOCAL)
        lib.vh_attn_rows_bf16.restype = I
        lib.vh_attn_rows_bf16.argtypes = [P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, P]
        libs.append((Path(p).name, lib))
    H16 = torch.bfloat16 if libs[0][1].vh_h16_format() else torch.float16
    g = torch.Generator().manual_seed(0)
    stream = torch.cuda.current_stream().cuda_stream
    for name, B, h, T, mode, xl in (('prompt pass 32x1024 prefix', 32, 8, 1024, 1, 256), ('NAR stage 64x1024 full', 64, 8, 1024, 0, 0)):
        d = h * 64
        q = (torch.randn(B * T, d, generator=g) * 0.18033688).to(H16).cuda()      # pre-scaled by 1/sqrt(64) log2(e) (ABI 127)
        kc = torch.randn(B, h, T, 64, generator=g).to(H16).cuda()
        vc = torch.randn(B, h, T, 64, generator=g).to(H16).cuda()
        out = torch.zeros(B * T, d, device='cuda', dtype=H16)
        ts = {n: [] for n, _ in libs}
        for _ in range(args.rounds):
            for n, lib in libs:
                def fn():
                    assert lib.vh_attn_rows_bf16(q.data_ptr(), d, kc.data_ptr(), vc.data_ptr(), out.data_ptr(), d, B, h, T, T, T, mode, xl,
                                                 None, None, stream) == 0
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn()
                e1.record()
                e1.synchronize()
                ts[n].append(e0.elapsed_time(e1) / args.reps * 1e3)
        for n, _ in libs:
            print(f'{name:28s} {n:16s} median {sorted(ts[n])[len(ts[n]) // 2]:8.1f} us  min {min(ts[n]):8.1f} us', flush=True)


if __name__ == '__main__':
    main()
