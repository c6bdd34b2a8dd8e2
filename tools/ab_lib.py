"""A/B of two builds of the library on the perf-mode GEMM shapes, alternating in ONE process on ONE device (device-to-device and
run-to-run differences of 5-10 % drown a 3 % change otherwise): tools/ab_lib.py OLD.so NEW.so [--reps 20].  Both are loaded with
ctypes next to each other (the symbols are looked up per handle) and called with raw pointers; torch provides memory and events."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('old')
    ap.add_argument('new')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    torch.cuda.init()
    libs = {}
    for name, path in (('old', args.old), ('new', args.new)):
        lib = C.CDLL(str(Path(path).resolve()), mode=C.RTLD_LOCAL)
        lib.vh_linear_bf16.restype = C.c_int
        lib.vh_linear_bf16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.vh_set_tuning.argtypes = [C.c_int, C.c_int]
        lib.vh_set_tuning(15, 4)
        fmt = lib.vh_h16_format()
        libs[name] = lib
    H16 = torch.bfloat16 if fmt else torch.float16
    g = torch.Generator().manual_seed(0)
    stream = torch.cuda.current_stream().cuda_stream

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / args.reps * 1e3

    for M in (32768, 65536):
        for N, Kd, act, res, o16, name in ((1536, 512, 0, False, True, 'qkv-like'), (512, 512, 0, True, False, 'out-proj'),
                                           (2048, 512, 1, False, True, 'linear_1+gelu'), (512, 2048, 0, True, False, 'linear_2')):
            a = torch.randn(M, Kd, generator=g).to(H16).cuda()
            w = (0.05 * torch.randn(N, Kd, generator=g)).to(H16).cuda()
            bias = torch.randn(N, generator=g).cuda()
            r = torch.randn(M, N, generator=g).cuda() if res else None
            out = torch.empty(M, N, device='cuda', dtype=H16 if o16 else torch.float32)

            def call(lib):
                rc = lib.vh_linear_bf16(a.data_ptr(), Kd, w.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, N,
                                        out.data_ptr(), N, int(o16), M, N, Kd, act, stream)
                assert rc == 0
            ts = {'old': [], 'new': []}
            for _ in range(args.rounds):
                for k in ('old', 'new'):
                    ts[k].append(timeit(lambda: call(libs[k])))
            call(libs['old'])
            o_old = out.clone()
            call(libs['new'])
            same = float((out.float() - o_old.float()).abs().max())
            med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
            print(f'M={M:6d} N={N:5d} K={Kd:5d} {name:14s} old {med["old"]:7.1f} us (min {min(ts["old"]):7.1f}) | new {med["new"]:7.1f} us '
                  f'(min {min(ts["new"]):7.1f}) | new / old {med["new"] / med["old"]:.3f} | max |new - old| {same:.1e}', flush=True)


if __name__ == '__main__':
    main()
