"""Phase timeline of the perf-mode attention kernel: a probe build of the library (the kernel with s_memtime stamps per tile and
wave, summed into a device array; built from csrc/bf16.hip by a patch) is run once per shape and the sums are
printed as cycles per tile and wave.  tools/probe_attn16.py PROBE.so"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

P, I = C.c_void_p, C.c_int
PHASES = ['K reads + score MFMAs', 'mask + row max', 'exp + sum + pack', 'V reads + PV MFMAs', 'wait for the next tile (vmcnt)', 'barrier']


def main():
    os.environ['VH_ATTN16_X1'] = '1'
    torch.cuda.init()
    lib = C.CDLL(str(Path(sys.argv[1]).resolve()), mode=C.RTLD_LOCAL)
    lib.vh_attn_rows_bf16.restype = I
    lib.vh_attn_rows_bf16.argtypes = [P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, P]
    lib.vh_probe_a16.argtypes = [C.POINTER(C.c_ulonglong), I]
    H16 = torch.bfloat16 if lib.vh_h16_format() else torch.float16
    g = torch.Generator().manual_seed(0)
    stream = torch.cuda.current_stream().cuda_stream
    for name, B, h, T, mode, xl in (('prompt pass 32x1024 prefix', 32, 8, 1024, 1, 256), ('NAR stage 64x1024 full', 64, 8, 1024, 0, 0)):
        d = h * 64
        q = torch.randn(B * T, d, generator=g).to(H16).cuda()
        kc = torch.randn(B, h, T, 64, generator=g).to(H16).cuda()
        vc = torch.randn(B, h, T, 64, generator=g).to(H16).cuda()
        out = torch.zeros(B * T, d, device='cuda', dtype=H16)

        def fn():
            assert lib.vh_attn_rows_bf16(q.data_ptr(), d, kc.data_ptr(), vc.data_ptr(), out.data_ptr(), d, B, h, T, T, T, mode, xl,
                                         None, None, stream) == 0
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.vh_probe_a16(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        buf = (C.c_ulonglong * 8)()
        lib.vh_probe_a16(buf, 0)
        tiles = buf[7]
        print(f'{name}: {e0.elapsed_time(e1) * 1e3:.1f} us with stamps; {tiles} wave-tiles visited (incl. skipped ones); per tile and wave:')
        for i, ph in enumerate(PHASES):
            print(f'    {ph:34s} {buf[i] / tiles:9.1f} cycles')
        print(f'    {"loop total":34s} {buf[6] / tiles:9.1f} cycles (s_memtime ticks)')


if __name__ == '__main__':
    main()
