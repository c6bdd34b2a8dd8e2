"""Perf mode (bf16 MFMA) kernels one by one at the configs[1] prompt-pass and configs[2] NAR-stage shapes, and the whole
stage forward in both modes.  SECONDARY numbers by construction (the headline stays fp32).

    python tools/bench_bf16.py [--reps 20]
"""
import argparse
import sys
import time
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
    return e0.elapsed_time(e1) / reps * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--skip-model', action='store_true')
    args = ap.parse_args()
    from valle2_amd import kernels as K
    g = torch.Generator().manual_seed(0)
    print('--- bf16 tile GEMM (random operands): default | the 128^2 forms | the persistent 256^2 form | the fp32 tile GEMM | hipBLASLt')
    print('    (torch.matmul, bf16 result, NO epilogue = the yardstick of the main loop; "same contract" = torch.matmul + the')
    print('     elementwise passes our epilogue fuses: bias (+ GELU) (+ fp32 residual))')
    from valle2_amd import _lib
    wins = 0
    for M in (32768, 65536):
        for N, Kd, act, res, o16, name in ((1536, 512, 0, False, True, 'qkv-like'), (512, 512, 0, True, False, 'out-proj'),
                                           (2048, 512, 1, False, True, 'linear_1+gelu'), (512, 2048, 0, True, False, 'linear_2')):
            a32 = torch.randn(M, Kd, generator=g).to(DEV)
            w32 = (0.05 * torch.randn(N, Kd, generator=g)).to(DEV)
            a, w = a32.to(H16), w32.to(H16)
            bias = torch.randn(N, generator=g).to(DEV)
            r = torch.randn(M, N, generator=g).to(DEV) if res else None
            out = torch.empty(M, N, device=DEV, dtype=H16 if o16 else torch.float32)
            out32 = torch.empty(M, N, device=DEV)
            o2 = torch.empty(M, N, device=DEV, dtype=H16)
            wt = w.T

            def blas_same():
                y = torch.matmul(a, wt, out=o2)
                if o16:
                    y = y + bias.to(H16)
                    return torch.nn.functional.gelu(y) if act else y
                y = y.float() + bias
                return y + r if res else y
            tv = {}
            for rnd in range(3):                                 # alternating arms, best of three: the clock moves between arms
                for form in (0, 1, 3, 4):                        # VH_TUNE_BF16_GEMM
                    _lib.lib().vh_set_tuning(15, form)
                    t = timeit(lambda: K.linear_bf16(a, w, bias, residual=r, out=out, act=K.ACT_GELU if act else K.ACT_NONE,
                                                     out_bf16=o16), args.reps)
                    tv[form] = min(tv.get(form, 1e30), t)
                tv['blas'] = min(tv.get('blas', 1e30), timeit(lambda: torch.matmul(a, wt, out=o2), args.reps))
                tv['same'] = min(tv.get('same', 1e30), timeit(blas_same, args.reps))
            _lib.lib().vh_set_tuning(15, 0)
            t32 = timeit(lambda: K.linear(a32, w32, bias, residual=r, out=out32, act=K.ACT_GELU if act else K.ACT_NONE), args.reps)
            fl = 2.0 * M * N * Kd
            byt = M * Kd * 2 + N * Kd * 2 + M * N * (2 if o16 else 4) + (M * N * 4 if res else 0)
            wins += tv[0] <= tv['same']
            print(f'M={M:6d} N={N:5d} K={Kd:5d} {name:14s} default {tv[0]:7.1f} us = {fl / tv[0] * 1e-6:7.1f} TF, {byt / tv[0] * 1e-6:5.2f} TB/s'
                  f' | two-slab {tv[1]:7.1f}, one-slab {tv[3]:7.1f}, persistent 256^2 {tv[4]:7.1f} us | fp32 {t32:7.1f} us (x{t32 / tv[0]:.2f})'
                  f' | hipBLASLt bare {tv["blas"]:7.1f} us = {fl / tv["blas"] * 1e-6:7.1f} TF, same contract {tv["same"]:7.1f} us')
    print(f'default form at or under hipBLASLt + its elementwise passes on {wins} of 8 shapes')
    print('--- attention, B x h x T x T')
    for B, h, T, mode in ((64, 8, 1024, 'full'), (32, 8, 1024, 'prefix'), (8, 16, 2875, 'full')):
        d = 64 * h
        q32 = torch.randn(B * T, d, generator=g).to(DEV)
        k32 = torch.randn(B, h, T, 64, generator=g).to(DEV)
        v32 = torch.randn(B, h, T, 64, generator=g).to(DEV)
        o32 = torch.empty(B * T, d, device=DEV)
        q, k, v = (q32 * K.Q16_PRESCALE).to(H16), k32.to(H16), v32.to(H16)      # q as the QKV product hands it over: pre-scaled
        o = torch.empty(B * T, d, device=DEV, dtype=H16)
        kw = dict(mode=K.MASK_FULL) if mode == 'full' else dict(mode=K.MASK_PREFIX, x_len=256)
        t16 = timeit(lambda: K.attn_rows_bf16(q, k, v, o, B, h, T, T, **kw), args.reps)
        t32 = timeit(lambda: K.attn_rows(q32, k32, v32, o32, B, h, T, T, **kw), args.reps)
        pairs = T * T if mode == 'full' else 256 * 256 + (T - 256) * 256 + (T - 256) * (T - 255) // 2
        fl = 4.0 * 64 * pairs * B * h
        print(f'B={B} h={h} T={T} {mode:6s} bf16 {t16:8.1f} us = {fl / t16 * 1e-6:6.1f} TF | fp32 {t32:8.1f} us = {fl / t32 * 1e-6:6.1f} TF | x{t32 / t16:.2f}')
    print('--- LayerNorm to bf16 (rows x 512)')
    for rows in (32768, 65536):
        x = torch.randn(rows, 512, generator=g).to(DEV)
        gm, bt = torch.ones(512, device=DEV), torch.zeros(512, device=DEV)
        o = torch.empty(rows, 512, device=DEV, dtype=H16)
        t = timeit(lambda: K.layernorm_bf16(x, gm, bt, out=o), args.reps)
        print(f'rows={rows} {t:7.1f} us = {rows * 512 * 6 / t * 1e-6:5.2f} TB/s')
    if args.skip_model:
        return
    print('--- whole stack forward: configs[2] NAR stage (64 x 1024, 12L/512d) and configs[1] prompt pass (32 x 1024)')
    from valle2_amd import ConfigValle, get_model_class, synth
    from valle2_amd import engine
    kw = dict(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0)
    for name, B, norm, mode, xl in (('nar', 64, 'AdaptiveLayerNorm', K.MASK_FULL, 0), ('prefill', 32, 'LayerNorm', K.MASK_PREFIX, 256)):
        cfg = ConfigValle(**kw, norm=norm)
        mname = 'ValleNAR' if name == 'nar' else 'ValleAR'
        m = get_model_class(mname)(cfg)
        m.load_state_dict(synth.make_state_dict(cfg, mname, seed=0))
        m = m.to(DEV).eval()
        T = 1024
        x0 = torch.randn(B, T, 512, generator=g).to(DEV)
        x = torch.empty_like(x0)
        emb = m.stage_embs[2].weight if name == 'nar' else None
        c32 = engine.KVCache(12, B, 8, T, DEV)
        c16 = engine.KVCache(12, B, 8, T, DEV, dtype=H16)
        s32, s16 = engine.ForwardScratch(B * T, 512, 2048, DEV), engine.ForwardScratch16(B * T, 512, 2048, DEV)

        def f32():
            engine.transformer_forward(m.transformer, x, c32, mode=mode, x_len=xl, embedding=emb, scratch=s32, x_in=x0)

        def f16():
            engine.transformer_forward_bf16(m.transformer, x, c16, mode=mode, x_len=xl, embedding=emb, scratch=s16, x_in=x0)
        from valle2_amd import _lib
        with torch.no_grad():
            forms = {}
            for rnd in range(2):
                for form in (1, 3, 4):
                    _lib.lib().vh_set_tuning(15, form)
                    forms[form] = min(forms.get(form, 1e30), timeit(f16, 10))
            _lib.lib().vh_set_tuning(15, 0)
            print(f'{name:8s} B={B}: bf16 stack with the two-slab GEMM {forms[1] / 1e3:7.2f} ms, with the one-slab GEMM {forms[3] / 1e3:7.2f} ms, '
                  f'with the persistent 256^2 GEMM {forms[4] / 1e3:7.2f} ms')
            t32, t16 = timeit(f32, 5), timeit(f16, 10)
            f32()
            y32 = x.clone()
            f16()
            err = float((x - y32).abs().max())
        pairs = T * T if mode == K.MASK_FULL else 256 * 256 + 768 * 256 + 768 * 769 // 2
        fl = B * T * 75497472.0 + 4.0 * 512 * pairs * 12 * B
        print(f'{name:8s} B={B}: fp32 {t32 / 1e3:7.2f} ms = {fl / t32 * 1e-6:6.1f} TF | bf16 {t16 / 1e3:7.2f} ms = {fl / t16 * 1e-6:6.1f} TF '
              f'| x{t32 / t16:.2f} | max |hidden diff| {err:.2e}')


if __name__ == '__main__':
    main()
