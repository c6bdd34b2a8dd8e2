"""A/B of two builds of the library on the fp32 many-row attention (vh_attn_rows) at the NAR-stage (64 x 8 x 1024^2, full mask), prompt-pass
(32 x 8 x 1024^2, prefix mask) and training shapes, alternating in one process: tools/ab_attn32_lib.py OLD.so NEW.so"""
import argparse
import ctypes as C
from pathlib import Path

import torch

P, I = C.c_void_p, C.c_int


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('old')
    ap.add_argument('new')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    torch.cuda.init()
    libs = {}
    for k, p in (('old', args.old), ('new', args.new)):
        lib = C.CDLL(str(Path(p).resolve()), mode=C.RTLD_LOCAL)
        lib.vh_attn_rows.restype = I
        lib.vh_attn_rows.argtypes = [P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, P, P, P]
        libs[k] = lib
    g = torch.Generator().manual_seed(0)
    stream = torch.cuda.current_stream().cuda_stream
    for name, B, h, T, mode, xl in (('NAR stage 64x1024 full', 64, 8, 1024, 0, 0), ('prompt pass 32x1024 prefix', 32, 8, 1024, 1, 256),
                                    ('NAR training 16x640 full', 16, 8, 640, 0, 0)):
        d = h * 64
        q = torch.randn(B * T, d, generator=g).cuda()
        kc = torch.randn(B, h, T, 64, generator=g).cuda()
        vc = torch.randn(B, h, T, 64, generator=g).cuda()
        out = torch.zeros(B * T, d, device='cuda')

        def call(lib):
            assert lib.vh_attn_rows(q.data_ptr(), d, kc.data_ptr(), vc.data_ptr(), out.data_ptr(), d, B, h, T, T, T, mode, xl, None, None,
                                    None, None, stream) == 0
        ts = {'old': [], 'new': []}
        for _ in range(args.rounds):
            for k in ('old', 'new'):
                for _ in range(2):
                    call(libs[k])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    call(libs[k])
                e1.record()
                e1.synchronize()
                ts[k].append(e0.elapsed_time(e1) / args.reps * 1e3)
        call(libs['old'])
        ref = out.clone()
        call(libs['new'])
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        print(f'{name:30s} old {med["old"]:8.1f} us | new {med["new"]:8.1f} us | new / old {med["new"] / med["old"]:.3f} | max |new - old| '
              f'{float((out - ref).abs().max()):.1e}', flush=True)


if __name__ == '__main__':
    main()
