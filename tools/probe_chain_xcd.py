"""How long does the decode GEMM chain of one layer take when its persistent launch (vh_decode_chain) is confined to
ONE XCD?  (developer probe for the XCD-specialisation idea in DESIGN.md section 8.)  Runs the chain launch on CU-masked
streams with 32 workgroups (VH_TUNE_CHAIN_GRID), reports which XCDs the workgroups landed on (HW_REG_XCC_ID), the time
per launch (HIP events on that stream) and the per-stage stamps; for comparison the same launch on all CUs.
usage: probe_chain_xcd.py [rows=32]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import _lib, kernels as K  # noqa: E402

hip = None
for name in ('libamdhip64.so', 'libamdhip64.so.7', 'libamdhip64.so.6'):
    try:
        hip = C.CDLL(name)
        break
    except OSError:
        continue


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s.value


def main(rows=32):
    dev = 'cuda'
    B, d, dff, h, L = rows, 512, 2048, 8, 12
    lib = _lib.lib()
    torch.manual_seed(0)
    attn = torch.randn(B, d, device=dev)
    x = torch.randn(B, d, device=dev)
    q = torch.empty(B, d, device=dev)
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    wq = [0.02 * torch.randn(3 * d, d, device=dev) for _ in range(L)]
    wo = [0.02 * torch.randn(d, d, device=dev) for _ in range(L)]
    w1 = [0.02 * torch.randn(dff, d, device=dev) for _ in range(L)]
    w2 = [0.02 * torch.randn(d, dff, device=dev) for _ in range(L)]
    bo, b1, b2 = torch.zeros(d, device=dev), torch.zeros(dff, device=dev), torch.zeros(d, device=dev)
    fq = [K.ln_fold(wq[l], g, b) for l in range(L)]
    f1 = [K.ln_fold(w1[l], g, b, b1) for l in range(L)]
    kc = torch.zeros(B, h, 64, 64, device=dev)
    vc = torch.zeros_like(kc)
    cl = torch.zeros(B, device=dev, dtype=torch.int32)
    nbytes = lib.vh_decode_chain_ws_bytes(B, d, dff)
    logits = torch.empty(B, 1028, device=dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    e0, e1 = C.c_void_p(), C.c_void_p()
    hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))

    def run(name, bits, grid):
        lib.vh_set_tuning(8, grid)
        s = masked_stream(bits) if bits is not None else torch.cuda.current_stream().cuda_stream
        ws = torch.zeros(nbytes // 4, device=dev)
        sync = torch.zeros(64 + 32 * 256, device=dev, dtype=torch.int32)
        sync[2] = 1
        torch.cuda.synchronize()
        reps = 20
        for warm in range(2):
            hip.hipEventRecord(e0, C.c_void_p(s))
            for step in range(reps):
                for l in range(L):
                    nl = (l + 1) % L
                    rc = lib.vh_decode_chain(ptr(attn), ptr(x), ptr(q), ptr(wo[l]), ptr(bo), ptr(f1[l][0]), ptr(f1[l][1]),
                                             ptr(f1[l][2]), ptr(w2[l]), ptr(b2), ptr(fq[nl][0]), ptr(fq[nl][1]),
                                             ptr(fq[nl][2]), ptr(kc), ptr(vc), ptr(cl), None, ptr(logits), 1028, 1025, B, d,
                                             dff, h, 64, l, C.c_float(1e-5), ptr(ws), C.c_size_t(nbytes), ptr(sync),
                                             C.c_void_p(s))
                    assert rc == 0, (rc, lib.vh_last_error())
                # the tags carry cache_len[0]: advance it once per "step" (on the same stream)
                K.check(lib.vh_greedy_step and 0, 'noop') if False else None
                hip.hipMemsetD32Async(C.c_void_p(cl.data_ptr()), C.c_int(warm * reps + step + 1), C.c_size_t(1), C.c_void_p(s))
            hip.hipEventRecord(e1, C.c_void_p(s))
            hip.hipStreamSynchronize(C.c_void_p(s))
        ms = C.c_float(0)
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        err = int(sync[1].item()) & 0xffffffff
        n = grid if grid else 256
        st = sync[64:].view(torch.int64).view(256, 16)[:n].cpu()
        xcc = sorted(set(int(v) for v in st[:, 6]))
        rr = sum(int(st[i, 6]) == i % 8 for i in range(n))
        stages = (st[:, 1:6] - st[:, 0:5]).double() / 100.0
        print(f'{name:44s} {ms.value / (reps * L) * 1e3:7.2f} us per launch; error word {err:#x}; XCDs {xcc} (workgroup i on XCD i % 8: {rr}/{n}); '
              f'stages (median us) ' + ' '.join(f'{float(stages[:, i].median()):.1f}' for i in range(5)), flush=True)

    full = (1 << 256) - 1
    run('all CUs, 256 workgroups', None, 0)
    run('all CUs, 32 workgroups', None, 32)
    run('mask bits 0-31, 32 workgroups', (1 << 32) - 1, 32)
    run('mask every 8th bit, 32 workgroups', sum(1 << i for i in range(0, 256, 8)), 32)
    run('mask bits 0-63, 64 workgroups', (1 << 64) - 1, 64)
    lib.vh_set_tuning(8, 0)


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:]])
