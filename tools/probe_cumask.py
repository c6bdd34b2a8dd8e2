"""Diagnostic: does the decode GEMM chain keep its speed while decode attention saturates HBM on OTHER CUs?
Two HIP streams with CU masks (hipExtStreamCreateWithCUMask): attention launches back to back on stream A,
the GEMM chain of one layer (out-proj, linear_1, linear_2 split-K + reduce, next QKV) repeatedly on stream B.
Reports the chain time alone, beside attention on disjoint CUs, and beside attention without masks."""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import _lib, kernels as K  # noqa: E402

hip = C.CDLL(None)          # the HIP runtime torch already loaded
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


def on(stream_handle):
    class Ctx:
        def __enter__(self):
            self.old = _lib.stream
            _lib.stream = lambda: stream_handle
            K.stream = _lib.stream
        def __exit__(self, *a):
            _lib.stream = self.old
            K.stream = self.old
    return Ctx()


def main():
    dev = 'cuda'
    B, d, dff, h, S = 32, 512, 2048, 8, 1280
    torch.manual_seed(0)
    q = torch.randn(B, d, device=dev); ao = torch.empty(B, d, device=dev)
    caches = [(torch.randn(B, h, 1536, 64, device=dev), torch.randn(B, h, 1536, 64, device=dev)) for _ in range(12)]
    cl = torch.full((B,), S - 1, device=dev, dtype=torch.int32)
    x = torch.randn(B, d, device=dev); hid = torch.empty(B, dff, device=dev); qo = torch.empty(B, d, device=dev)
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    L = 12
    wq = [0.02 * torch.randn(3 * d, d, device=dev) for _ in range(L)]
    wo = [0.02 * torch.randn(d, d, device=dev) for _ in range(L)]
    w1 = [0.02 * torch.randn(dff, d, device=dev) for _ in range(L)]
    w2 = [0.02 * torch.randn(d, dff, device=dev) for _ in range(L)]
    bo, b1 = torch.zeros(d, device=dev), torch.zeros(dff, device=dev)
    fq = [K.ln_fold(wq[l], g, b) for l in range(L)]
    f1 = [K.ln_fold(w1[l], g, b, b1) for l in range(L)]
    kc = torch.zeros(B, h, 64, 64, device=dev); vc = torch.zeros_like(kc)
    cl0 = torch.zeros(B, device=dev, dtype=torch.int32)
    ws = torch.empty(_lib.lib().vh_linear_ws_bytes(B, d, dff) // 4, device=dev)
    torch.cuda.synchronize()

    def chain(l):
        K.linear(ao, wo[l], bo, x, out=x)
        K.linear_folded(x, f1[l], out=hid, act=1)
        K.linear_ws(hid, w2[l], bo, x, out=x, workspace=ws)
        K.linear_qkv_folded(x, fq[l], qo, kc, vc, B, 1, h, cache_len=cl0)

    def attn(l):
        K.attn_decode(q, caches[l][0], caches[l][1], ao, cl, 1, 1, None)

    full = (1 << 256) - 1
    configs = {
        'chain alone (all CUs)': (full, None),
        'chain on CUs 224-255 alone': (((1 << 32) - 1) << 224, None),
        'chain on CUs 192-255 alone': (((1 << 64) - 1) << 192, None),
        'chain (all CUs) beside attention (all CUs)': (full, full),
        'chain on 224-255 beside attention on 0-223': (((1 << 32) - 1) << 224, (1 << 224) - 1),
        'chain on 192-255 beside attention on 0-191': (((1 << 64) - 1) << 192, (1 << 192) - 1),
        'chain on every 8th CU beside attention on the rest': (int('01' * 0, 2) if False else sum(1 << i for i in range(0, 256, 8)), full ^ sum(1 << i for i in range(0, 256, 8))),
    }
    e = [hip.hipEventCreate, hip.hipEventRecord, hip.hipEventSynchronize, hip.hipEventElapsedTime]
    for name, (mb, ma) in configs.items():
        sb = masked_stream(mb)
        sa = masked_stream(ma) if ma is not None else None
        reps = 40
        e0, e1 = C.c_void_p(), C.c_void_p()
        hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
        a0, a1 = C.c_void_p(), C.c_void_p()
        hip.hipEventCreate(C.byref(a0)); hip.hipEventCreate(C.byref(a1))
        for warm in range(2):
            if sa is not None:
                with on(sa):
                    hip.hipEventRecord(a0, C.c_void_p(sa))
                    for i in range(reps * 4):
                        attn(i % 12)
                    hip.hipEventRecord(a1, C.c_void_p(sa))
            with on(sb):
                hip.hipEventRecord(e0, C.c_void_p(sb))
                for i in range(reps):
                    chain(i % 12)
                hip.hipEventRecord(e1, C.c_void_p(sb))
            torch.cuda.synchronize()
        ms = C.c_float(0)
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        line = f'{name}: chain {ms.value / reps * 1e3:.1f} us per layer'
        if sa is not None:
            hip.hipEventElapsedTime(C.byref(ms), a0, a1)
            line += f'; attention {ms.value / (reps * 4) * 1e3:.1f} us per launch (stream total {ms.value:.2f} ms)'
        print(line, flush=True)


if __name__ == '__main__':
    main()
