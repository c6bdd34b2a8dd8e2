"""First contact of the persistent 256^2 bf16 GEMM (csrc/gemm16p.hip, VH_TUNE_BF16_GEMM = 4) with the hardware: a few
correctness checks against fp64 on the device, then timings of the eight perf-mode shapes beside the 128^2 forms and
torch.matmul (hipBLASLt).  Run it under `timeout`: a barrier mismatch in a persistent kernel is a hang.

    timeout -k 10 300 python tools/try_p256.py [--reps 20] [--no-time]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
DEV = 'cuda'
from valle2_amd._lib import h16_dtype  # noqa: E402
H16 = h16_dtype()          # the library's 16-bit operand format (fp16 by default)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--no-time', action='store_true')
    args = ap.parse_args()
    from valle2_amd import _lib, kernels as K
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    bad = 0
    print('--- correctness (form 4) against fp64 on the same bf16 operands', flush=True)
    for M, N, Kd, o16, act, res in ((256, 256, 256, False, 0, False), (300, 512, 384, False, 0, True), (1, 256, 256, False, 0, False),
                                    (1000, 512, 512, True, 1, False), (4096, 512, 2048, False, 0, True),
                                    (16384 + 77, 2048, 512, True, 1, False), (20000, 1536, 512, False, 1, True),
                                    (65536, 512, 512, True, 0, False)):
        a = torch.randn(M, Kd, generator=g).to(H16).to(DEV)
        w = (0.05 * torch.randn(N, Kd, generator=g)).to(H16).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        r = torch.randn(M, N, generator=g).to(DEV) if res else None
        ref = a.double() @ w.double().T + bias.double()
        if act:
            ref = torch.nn.functional.gelu(ref)
        if res:
            ref = ref + r.double()
        L.vh_set_tuning(15, 4)
        out = K.linear_bf16(a, w, bias, residual=r, act=K.ACT_GELU if act else K.ACT_NONE, out_bf16=o16)
        torch.cuda.synchronize()
        L.vh_set_tuning(15, 0)
        err = (out.double() - ref).abs()
        tol = (1e-4 if act else 1e-5) + ref.abs() * (2 ** -7 if act else 2 ** -8) if o16 else 2e-5 * Kd ** 0.5 + 1e-5 * ref.abs()
        nbad = int((err > tol).sum())
        bad += nbad
        print(f'M={M:6d} N={N:5d} K={Kd:5d} out16={int(o16)} act={act} res={int(res)}: max err {float(err.max()):.3e}, '
              f'{nbad} elements out of tolerance', flush=True)
        if nbad:
            idx = (err > tol).nonzero()[:5].tolist()
            print('   first bad (row, col):', idx, flush=True)
    # integer operands: exact
    M, N, Kd = 520, 512, 256
    m, n, k = torch.arange(M)[:, None], torch.arange(N)[:, None], torch.arange(Kd)[None, :]
    a = ((3 * m + 5 * k) % 7 - 3).float()
    w = ((2 * n + k) % 5 - 2).float()
    ref = a @ w.T
    L.vh_set_tuning(15, 4)
    out = K.linear_bf16(a.to(H16).to(DEV), w.to(H16).to(DEV))
    L.vh_set_tuning(15, 0)
    ok = torch.equal(out.cpu(), ref)
    print('integer operands exact:', ok, flush=True)
    bad += 0 if ok else 1
    # QKV scatter
    for B, T, h, with_len in ((3, 150, 8, True), (40, 7, 8, True), (33, 1024, 8, False)):
        d = 64 * h
        S_max = T + 20
        a = torch.randn(B * T, d, generator=g).to(H16).to(DEV)
        w = (0.1 * torch.randn(3 * d, d, generator=g)).to(H16).to(DEV)
        ref = (a.double() @ w.double().T).float()
        cl = torch.tensor([(3 * i) % 17 for i in range(B)], dtype=torch.int32).to(DEV) if with_len else None
        kc = torch.zeros(B, h, S_max, 64, device=DEV, dtype=H16)
        vc = torch.zeros_like(kc)
        q = torch.empty(B * T, d, device=DEV, dtype=H16)
        L.vh_set_tuning(15, 4)
        K.linear_qkv_bf16(a, w, q, kc, vc, B, T, h, cache_len=cl)
        torch.cuda.synchronize()
        L.vh_set_tuning(15, 0)
        e_q = float((q.float() - ref[:, :d]).abs().max())
        kref = ref[:, d:2 * d].view(B, T, h, 64).permute(0, 2, 1, 3)
        vref = ref[:, 2 * d:].view(B, T, h, 64).permute(0, 2, 1, 3)
        e_k = e_v = 0.0
        stray = 0.0
        for b in range(B):
            p0 = 0 if cl is None else int(cl[b])
            e_k = max(e_k, float((kc[b, :, p0:p0 + T].float() - kref[b]).abs().max()))
            e_v = max(e_v, float((vc[b, :, p0:p0 + T].float() - vref[b]).abs().max()))
            stray += float(kc[b, :, :p0].abs().sum()) + float(kc[b, :, p0 + T:].abs().sum())
        scale = float(ref.abs().max())
        okq = max(e_q, e_k, e_v) < 2 ** -7 * scale and stray == 0
        bad += 0 if okq else 1
        print(f'qkv B={B} T={T} h={h}: |dq| {e_q:.2e} |dK| {e_k:.2e} |dV| {e_v:.2e} (scale {scale:.1f}) stray {stray}: {"ok" if okq else "BAD"}',
              flush=True)
    print('CORRECTNESS:', 'ok' if bad == 0 else f'{bad} problems', flush=True)
    if args.no_time:
        return
    print('--- timings: default 128^2 forms (knob 1/3) | 256^2 persistent (knob 4) | torch.matmul bf16', flush=True)
    for M in (32768, 65536):
        for N, Kd, act, res, o16, name in ((1536, 512, 0, False, True, 'qkv-like'), (512, 512, 0, True, False, 'out-proj'),
                                           (2048, 512, 1, False, True, 'linear_1+gelu'), (512, 2048, 0, True, False, 'linear_2')):
            a = torch.randn(M, Kd, generator=g).to(H16).to(DEV)
            w = (0.05 * torch.randn(N, Kd, generator=g)).to(H16).to(DEV)
            bias = torch.randn(N, generator=g).to(DEV)
            r = torch.randn(M, N, generator=g).to(DEV) if res else None
            out = torch.empty(M, N, device=DEV, dtype=H16 if o16 else torch.float32)
            tv = {}
            for rnd in range(3):
                for form in (1, 3, 4):
                    L.vh_set_tuning(15, form)
                    t = timeit(lambda: K.linear_bf16(a, w, bias, residual=r, out=out, act=K.ACT_GELU if act else K.ACT_NONE,
                                                     out_bf16=o16), args.reps)
                    tv[form] = min(tv.get(form, 1e30), t)
                wt = w.T
                o2 = torch.empty(M, N, device=DEV, dtype=H16)
                t = timeit(lambda: torch.matmul(a, wt, out=o2), args.reps)
                tv['blas'] = min(tv.get('blas', 1e30), t)
            L.vh_set_tuning(15, 0)
            fl = 2.0 * M * N * Kd
            print(f'M={M:6d} N={N:5d} K={Kd:5d} {name:14s} two-slab {tv[1]:7.1f} us | one-slab {tv[3]:7.1f} us | p256 {tv[4]:7.1f} us = '
                  f'{fl / tv[4] * 1e-6:7.1f} TF | torch.matmul (bf16 out, no epilogue) {tv["blas"]:7.1f} us = {fl / tv["blas"] * 1e-6:7.1f} TF',
                  flush=True)


if __name__ == '__main__':
    main()
