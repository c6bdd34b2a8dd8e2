"""A/B of two builds of the library on the attention kernels, alternating in ONE process on ONE device (the boxes of the pool
differ by 5-10 %, which drowns a 3 % change): tools/ab_attn_lib.py OLD.so NEW.so [--reps 20] [--rounds 5].

  fwd16  vh_attn_rows_bf16      perf-mode many-row attention: prompt pass of configs[1] (32 x 1024, prefix mask over 256 text
                                positions), NAR stage of configs[2] (64 x 1024, full mask)
  bwd    vh_attn_rows_bwd_ws    fp32 attention backward at the configs[3] training shapes (AR: 16 rows, T = 1020, prefix mask,
                                ragged key lengths; NAR: 16 x 640, full mask)

Both libraries are loaded with ctypes next to each other and called with raw pointers; torch provides memory and events.  The
outputs of the two builds are compared bit for bit."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
MASK_FULL, MASK_PREFIX = 0, 1
P, I, Z = C.c_void_p, C.c_int, C.c_size_t


def load(path):
    lib = C.CDLL(str(Path(path).resolve()), mode=C.RTLD_LOCAL)
    lib.vh_attn_rows_bf16.restype = I
    lib.vh_attn_rows_bf16.argtypes = [P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, P]
    lib.vh_attn_rows_lse.restype = I
    lib.vh_attn_rows_lse.argtypes = [P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, P, P, P, P]
    lib.vh_attn_rows_bwd_ws_bytes.restype = Z
    lib.vh_attn_rows_bwd_ws_bytes.argtypes = [I, I, I]
    lib.vh_attn_rows_bwd_ws.restype = I
    lib.vh_attn_rows_bwd_ws.argtypes = [P, I, P, P, P, I, P, I, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, P, Z, P]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('old')
    ap.add_argument('new')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    torch.cuda.init()
    libs = {'old': load(args.old), 'new': load(args.new)}
    H16 = torch.bfloat16 if libs['new'].vh_h16_format() else torch.float16
    g = torch.Generator().manual_seed(0)
    stream = torch.cuda.current_stream().cuda_stream
    dev = 'cuda'

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

    def ab(name, call, outs):
        ts = {'old': [], 'new': []}
        for _ in range(args.rounds):
            for k in ('old', 'new'):
                ts[k].append(timeit(lambda: call(libs[k])))
        call(libs['old'])
        ref = [o.clone() for o in outs]
        call(libs['new'])
        same = max(float((o.float() - r.float()).abs().max()) for o, r in zip(outs, ref))
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        print(f'{name:34s} old {med["old"]:8.1f} us (min {min(ts["old"]):8.1f}) | new {med["new"]:8.1f} us (min {min(ts["new"]):8.1f}) '
              f'| new / old {med["new"] / med["old"]:.3f} | max |new - old| {same:.1e}', flush=True)

    # ---- perf-mode forward
    for name, B, h, T, mode, xl in (('fwd16 prompt pass 32x1024 prefix', 32, 8, 1024, MASK_PREFIX, 256),
                                   ('fwd16 NAR stage 64x1024 full', 64, 8, 1024, MASK_FULL, 0)):
        d = h * 64
        q = (torch.randn(B * T, d, generator=g) * 0.18033688).to(H16).to(dev)     # pre-scaled by 1/sqrt(64) log2(e) (ABI 127)
        kc = torch.randn(B, h, T, 64, generator=g).to(H16).to(dev)
        vc = torch.randn(B, h, T, 64, generator=g).to(H16).to(dev)
        out = torch.zeros(B * T, d, device=dev, dtype=H16)

        def call(lib):
            rc = lib.vh_attn_rows_bf16(q.data_ptr(), d, kc.data_ptr(), vc.data_ptr(), out.data_ptr(), d, B, h, T, T, T, mode, xl,
                                       None, None, stream)
            assert rc == 0
        ab(name, call, [out])
    # ---- fp32 backward at the training shapes
    kv_ar = (120 + torch.randint(225, 901, (16,), generator=g)).to(torch.int32)
    for name, B, h, T, mode, xl, kvl in (('bwd AR step 16x1020 prefix ragged', 16, 8, 1020, MASK_PREFIX, 120, kv_ar),
                                        ('bwd NAR step 16x640 full', 16, 8, 640, MASK_FULL, 0, None)):
        d = h * 64
        q = torch.randn(B * T, d, generator=g).to(dev)
        kc = torch.randn(B, h, T, 64, generator=g).to(dev)
        vc = torch.randn(B, h, T, 64, generator=g).to(dev)
        dout = torch.randn(B * T, d, generator=g).to(dev)
        out = torch.zeros(B * T, d, device=dev)
        lse = torch.zeros(B, h, T, device=dev)
        kvd = kvl.to(dev) if kvl is not None else None
        kvp = kvd.data_ptr() if kvd is not None else None
        lib0 = libs['old']
        rc = lib0.vh_attn_rows_lse(q.data_ptr(), d, kc.data_ptr(), vc.data_ptr(), out.data_ptr(), d, B, h, T, T, T,
                                   mode, xl, None, kvp, None, None, lse.data_ptr(), stream)
        assert rc == 0, rc
        dq, dk, dv = (torch.zeros(B * T, d, device=dev) for _ in range(3))
        nbytes = max(lb.vh_attn_rows_bwd_ws_bytes(B, h, T) for lb in libs.values())
        ws = torch.empty(nbytes // 4 + 4, device=dev)

        def callb(lib):
            rc = lib.vh_attn_rows_bwd_ws(q.data_ptr(), d, kc.data_ptr(), vc.data_ptr(), out.data_ptr(), d, dout.data_ptr(), d,
                                         lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), d, B, h, T, T, mode, xl, None,
                                         kvp, None, None, ws.data_ptr(), nbytes, stream)
            assert rc == 0, rc
        ab(name, callb, [dq, dk, dv])


if __name__ == '__main__':
    main()
