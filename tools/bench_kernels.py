"""Kernel-level A/B timings in ONE process (interleaved rounds, median), for tuning only.
usage: python tools/bench_kernels.py [attn] [gemm] [rows]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import _lib, kernels as K  # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def flush_cache(buf):
    buf.add_(1.0)    # touch 1 GiB so L2 / Infinity Cache hold none of the operands


def bench_attn():
    B, h, S_max = 32, 8, 1536
    q = torch.randn(B, 512, device=DEV)
    out = torch.empty(B, 512, device=DEV)
    # 12 layers of cache so successive launches stream different memory, as in the decode step
    caches = [(torch.randn(B, h, S_max, 64, device=DEV), torch.randn(B, h, S_max, 64, device=DEV))
              for _ in range(12)]
    lib = _lib.lib()
    for S in (1024, 1280, 1536):
        cl = torch.full((B,), S - 1, device=DEV, dtype=torch.int32)
        bytes_ = 2 * B * S * 512 * 4
        res = {}
        for rnd in range(3):
            for variant, waves, ns in ((0, 0, 1), (1, 16, 1), (1, 8, 1), (1, 8, 2), (1, 4, 2), (1, 4, 4)):
                lib.vh_set_tuning(0, variant)
                lib.vh_set_tuning(1, waves)
                ws = K.attn_decode_ws(B, h, ns, DEV)
                i = [0]

                def fn():
                    kc, vc = caches[i[0] % 12]
                    i[0] += 1
                    K.attn_decode(q, kc, vc, out, cl, 1, ns, ws)
                res.setdefault((variant, waves, ns), []).append(timeit(fn, iters=48))
        for k, v in sorted(res.items(), key=lambda kv: statistics.median(kv[1])):
            us = statistics.median(v)
            print(f'attn S={S} variant={k[0]} waves={k[1]} n_split={k[2]}: {us:7.2f} us  '
                  f'{bytes_ / us / 1e3:7.1f} GB/s', flush=True)
    lib.vh_set_tuning(0, 0)
    lib.vh_set_tuning(1, 0)


def bench_gemm():
    B, d, dff = 32, 512, 2048
    x = torch.randn(B, d, device=DEV)
    hid = torch.randn(B, dff, device=DEV)
    g, b = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    L = 12
    wqkv = [0.02 * torch.randn(3 * d, d, device=DEV) for _ in range(L)]
    wo = [0.02 * torch.randn(d, d, device=DEV) for _ in range(L)]
    w1 = [0.02 * torch.randn(dff, d, device=DEV) for _ in range(L)]
    w2 = [0.02 * torch.randn(d, dff, device=DEV) for _ in range(L)]
    bo, b1 = torch.zeros(d, device=DEV), torch.zeros(dff, device=DEV)
    kc = torch.zeros(B, 8, 64, 64, device=DEV)
    vc = torch.zeros_like(kc)
    q = torch.empty(B, d, device=DEV)
    o1 = torch.empty(B, d, device=DEV)
    o2 = torch.empty(B, dff, device=DEV)
    cl = torch.zeros(B, device=DEV, dtype=torch.int32)
    big = torch.zeros(256 * 1024 * 1024 // 4, device=DEV)
    ln = (g, b, None, None, 1e-5)
    ws2 = torch.zeros(_lib.lib().vh_linear_ws_bytes(B, d, dff) // 4, device=DEV)
    i = [0]

    def nxt():
        i[0] += 1
        return i[0] % L
    lib = _lib.lib()
    fq = [K.ln_fold(wqkv[l], g, b) for l in range(L)]
    f1 = [K.ln_fold(w1[l], g, b, b1) for l in range(L)]
    b2 = torch.zeros(d, device=DEV)
    ffn_ws = K.ffn_decode_ws(B, d, dff, DEV)
    wh = 0.02 * torch.randn(1025, d, device=DEV)
    logits = torch.empty(B, 1028, device=DEV)

    def tuned(knob, val, fn):
        def run():
            lib.vh_set_tuning(knob, val)
            fn()
            lib.vh_set_tuning(knob, 0)
        return run
    def tuned2(sw, rows, fn):
        def run():
            lib.vh_set_tuning(7, sw)
            lib.vh_set_tuning(8, rows)
            fn()
            lib.vh_set_tuning(7, 0)
            lib.vh_set_tuning(8, 0)
        return run

    def ffn():
        l = nxt()
        K.ffn_decode(x, f1[l], w2[l], b2, out=o1, workspace=ffn_ws)

    def ffn_three():
        l = nxt()
        K.linear_folded(x, f1[l], out=o2, act=1)
        K.linear_ws(o2, w2[l], b2, x, out=o1, workspace=ws2)
    cases = {
        'ffn fused (default plan)': ffn,
        'ffn fused slice 32 x 8 rows': tuned2(32, 8, ffn),
        'ffn fused slice 32 x 16 rows': tuned2(32, 16, ffn),

        'ffn fused slice 16 x 16 rows': tuned2(16, 16, ffn),
        'ffn fused slice 16 x 8 rows': tuned2(16, 8, ffn),
        'ffn three launches (ffn1 fold + ffn2 split-K + reduce)': ffn_three,
        'qkv fold': lambda: K.linear_qkv_folded(x, fq[nxt()], q, kc, vc, B, 1, 8, cache_len=cl),
        'qkv fold, statistics from row loads': tuned(9, 1, lambda: K.linear_qkv_folded(x, fq[nxt()], q, kc, vc, B, 1, 8, cache_len=cl)),
        'ffn1 fold, statistics from row loads': tuned(9, 1, lambda: K.linear_folded(x, f1[nxt()], out=o2, act=1)),
        'qkv fold, no row groups': tuned(2, 2, lambda: K.linear_qkv_folded(x, fq[nxt()], q, kc, vc, B, 1, 8, cache_len=cl)),
        'ffn1 fold, no row groups': tuned(2, 2, lambda: K.linear_folded(x, f1[nxt()], out=o2, act=1)),
        'out-proj, no row groups': tuned(2, 2, lambda: K.linear(x, wo[nxt()], bo, o1, out=o1)),
        'head f32 rows, no row groups': tuned(2, 2, lambda: K.linear(x, wh, out=logits[:, :1025])),
        'out-proj, 16-row groups': tuned(2, 3, lambda: K.linear(x, wo[nxt()], bo, o1, out=o1)),
        'head f32 rows, 16-row groups': tuned(2, 3, lambda: K.linear(x, wh, out=logits[:, :1025])),
        'ffn1 fold': lambda: K.linear_folded(x, f1[nxt()], out=o2, act=1),
        'head f32 rows (N=1025)': lambda: K.linear(x, wh, out=logits[:, :1025]),
        'qkv+ln   (N=1536,K=512)': lambda: K.linear_qkv(x, wqkv[nxt()], q, kc, vc, B, 1, 8, cache_len=cl, ln=ln),
        'out-proj (N=512,K=512)': lambda: K.linear(x, wo[nxt()], bo, o1, out=o1),
        'ffn1+ln  (N=2048,K=512)': lambda: K.linear(x, w1[nxt()], b1, out=o2, act=1, ln=ln),
        'ffn2     (N=512,K=2048)': lambda: K.linear(hid, w2[nxt()], bo, o1, out=o1),
        'ffn2-splitk two launches': lambda: K.linear_ws(hid, w2[nxt()], bo, o1, out=o1, workspace=ws2),
        'ffn2 no split, no row groups': tuned(2, 2, lambda: K.linear(hid, w2[nxt()], bo, o1, out=o1)),
        'ffn2-splitk reduce=64thr, two launches': tuned(3, 64, lambda: K.linear_ws(hid, w2[nxt()], bo, o1, out=o1, workspace=ws2)),
    }
    if len(sys.argv) > 2:
        cases = {k: v for k, v in cases.items() if any(a in k for a in sys.argv[2:])}
    # graph-captured so the Python/ctypes launch cost is out of the picture
    for name, fn in cases.items():
        fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(L):
                fn()
        res = {}
        for mode in ('cold', 'warm'):
            ts = []
            for rnd in range(7):
                if mode == 'cold':
                    flush_cache(big)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / L * 1e3)
            res[mode] = statistics.median(ts)
        print(f'gemm {name}: cold {res["cold"]:6.2f} us  warm {res["warm"]:6.2f} us (per launch, 12 weight sets)',
              flush=True)


def bench_rows():
    B, T, d, dff, h = 32, 1024, 512, 2048, 8
    M = B * T
    x = torch.randn(M, d, device=DEV)
    for name, N, Kk in (('qkv', 1536, 512), ('out', 512, 512), ('ffn1', 2048, 512), ('ffn2', 512, 2048)):
        a = torch.randn(M, Kk, device=DEV)
        w = 0.02 * torch.randn(N, Kk, device=DEV)
        o = torch.empty(M, N, device=DEV)
        us = timeit(lambda: K.linear(a, w, out=o), iters=10, warm=2)
        print(f'tile gemm {name} M={M} N={N} K={Kk}: {us:8.1f} us  {2 * M * N * Kk / us / 1e6:6.1f} TFLOP/s', flush=True)
    q = torch.randn(M, d, device=DEV)
    kc = torch.randn(B, h, T, 64, device=DEV)
    vc = torch.randn(B, h, T, 64, device=DEV)
    o = torch.empty(M, d, device=DEV)
    for mode, xl in ((K.MASK_PREFIX, 256), (K.MASK_FULL, 0)):
        us = timeit(lambda: K.attn_rows(q, kc, vc, o, B, h, T, T, mode=mode, x_len=xl), iters=10, warm=2)
        pairs = (xl * xl + (T - xl) * xl + (T - xl) * (T - xl + 1) // 2) if mode == K.MASK_PREFIX else T * T
        print(f'attn_rows mode={mode}: {us:8.1f} us  {4 * 64 * pairs * B * h / us / 1e6:6.1f} TFLOP/s (visible pairs)',
              flush=True)


def bench_tile():
    """The tile GEMM (bias + residual epilogue, and the QKV scatter) vs the library GEMM on the
    NAR-stage / prefill shapes."""
    B, T, h = 32, 1024, 8
    for M in (32 * 1024, 8 * 1024, 32 * 1500):
        for name, N, Kk in (('qkv', 1536, 512), ('out', 512, 512), ('ffn1', 2048, 512), ('ffn2', 512, 2048)):
            a = torch.randn(M, Kk, device=DEV)
            w = 0.02 * torch.randn(N, Kk, device=DEV)
            bias = torch.randn(N, device=DEV)
            res = torch.randn(M, N, device=DEV)
            o = torch.empty(M, N, device=DEV)
            ref = torch.addmm(bias, a, w.t()) + res
            line = f'tile M={M} {name} N={N} K={Kk}:'
            for knob, label in ((1, 'staged'), (2, 'dma')):
                _lib.lib().vh_set_tuning(4, knob)
                us = statistics.median(timeit(lambda: K.linear(a, w, bias, res, out=o), iters=10, warm=2) for _ in range(3))
                err = float((o - ref).abs().max())
                line += f'  {label} {us:7.1f} us {2 * M * N * Kk / us / 1e6:6.1f} TF (err {err:.1e})'
            _lib.lib().vh_set_tuning(4, 0)
            us = timeit(lambda: torch.addmm(bias, a, w.t(), out=o), iters=10, warm=2)
            line += f'  library: {us:7.1f} us {2 * M * N * Kk / us / 1e6:6.1f} TF'
            print(line, flush=True)
    M = B * T
    a = torch.randn(M, 512, device=DEV)
    w = 0.02 * torch.randn(2048, 512, device=DEV)
    bias = torch.randn(2048, device=DEV)
    o = torch.empty(M, 2048, device=DEV)
    for act in (0, 1):
        us = statistics.median(timeit(lambda: K.linear(a, w, bias, None, out=o, act=act), iters=10, warm=2) for _ in range(3))
        print(f'tile ffn1 bias{" + GELU" if act else ""}: {us:7.1f} us {2 * M * 2048 * 512 / us / 1e6:6.1f} TF', flush=True)
    x = torch.randn(M, 512, device=DEV)
    w = 0.02 * torch.randn(1536, 512, device=DEV)
    q = torch.zeros(M, 512, device=DEV)
    kc = torch.zeros(B, h, T, 64, device=DEV)
    vc = torch.zeros_like(kc)
    us = timeit(lambda: K.linear_qkv(x, w, q, kc, vc, B, T, h), iters=10, warm=2)
    print(f'tile qkv-scatter: {us:7.1f} us {2 * M * 1536 * 512 / us / 1e6:6.1f} TF', flush=True)


def bench_gemm_big():
    """Decode GEMMs of configs[4] (24L / 1024d / dff 4096, 8 rows): weight-streaming regime (50 MB per layer)."""
    B, d, dff, L = 8, 1024, 4096, 24
    x = torch.randn(B, d, device=DEV)
    hid = torch.randn(B, dff, device=DEV)
    g, b = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    wqkv = [0.02 * torch.randn(3 * d, d, device=DEV) for _ in range(L)]
    wo = [0.02 * torch.randn(d, d, device=DEV) for _ in range(L)]
    w1 = [0.02 * torch.randn(dff, d, device=DEV) for _ in range(L)]
    w2 = [0.02 * torch.randn(d, dff, device=DEV) for _ in range(L)]
    bo, b1 = torch.zeros(d, device=DEV), torch.zeros(dff, device=DEV)
    kc = torch.zeros(B, 16, 64, 64, device=DEV)
    vc = torch.zeros_like(kc)
    q = torch.empty(B, d, device=DEV)
    o1 = torch.empty(B, d, device=DEV)
    o2 = torch.empty(B, dff, device=DEV)
    cl = torch.zeros(B, device=DEV, dtype=torch.int32)
    fq = [K.ln_fold(wqkv[l], g, b) for l in range(L)]
    f1 = [K.ln_fold(w1[l], g, b, b1) for l in range(L)]
    ws2 = torch.empty(max(1, _lib.lib().vh_linear_ws_bytes(B, d, dff) // 4), device=DEV)
    wso = torch.empty(max(1, _lib.lib().vh_linear_ws_bytes(B, d, d) // 4), device=DEV)
    i = [0]

    def nxt():
        i[0] += 1
        return i[0] % L
    cases = {
        'qkv fold   (3072 x 1024)': (lambda: K.linear_qkv_folded(x, fq[nxt()], q, kc, vc, B, 1, 16, cache_len=cl), 3 * d * d * 4),
        'out-proj   (1024 x 1024)': (lambda: K.linear_ws(x, wo[nxt()], bo, o1, out=o1, workspace=wso), d * d * 4),
        'ffn1 fold  (4096 x 1024)': (lambda: K.linear_folded(x, f1[nxt()], out=o2, act=1), dff * d * 4),
        'ffn2       (1024 x 4096)': (lambda: K.linear_ws(hid, w2[nxt()], bo, o1, out=o1, workspace=ws2), dff * d * 4),
    }
    for name, (fn, nbytes) in cases.items():
        us = statistics.median(timeit(fn, iters=96, warm=24) for _ in range(5))
        print(f'{name}: {us:7.2f} us  {nbytes / us / 1e3:7.1f} GB/s of weights', flush=True)


def bench_tilesweep():
    """Staged vs LDS-DMA tile kernel over tile counts around multiples of the 512 resident workgroups."""
    for N, Kk in ((2048, 512), (512, 2048)):
        for M in [128 * t for t in ((40, 44, 48, 52, 56, 60, 80, 96, 112) if N == 2048 else (136, 160, 176, 192, 208, 224, 240, 320, 384))]:
            a = torch.randn(M, Kk, device=DEV)
            w = 0.02 * torch.randn(N, Kk, device=DEV)
            bias = torch.randn(N, device=DEV)
            o = torch.empty(M, N, device=DEV)
            line = f'M={M} N={N} K={Kk} tiles={(M + 127) // 128 * ((N + 127) // 128)}:'
            for knob, label in ((1, 'staged'), (2, 'dma')):
                _lib.lib().vh_set_tuning(4, knob)
                us = statistics.median(timeit(lambda: K.linear(a, w, bias, None, out=o), iters=10, warm=2) for _ in range(3))
                line += f'  {label} {us:7.1f} us {2 * M * N * Kk / us / 1e6:6.1f} TF'
            _lib.lib().vh_set_tuning(4, 0)
            print(line, flush=True)


if __name__ == '__main__':
    what = sys.argv[1:] or ['attn', 'gemm', 'rows']
    if 'tilesweep' in what:
        bench_tilesweep()
    if 'gemmbig' in what:
        bench_gemm_big()
    if 'tile' in what:
        bench_tile()
    if 'gemm' in what:
        bench_gemm()
    if 'attn' in what:
        bench_attn()
    if 'rows' in what:
        bench_rows()
