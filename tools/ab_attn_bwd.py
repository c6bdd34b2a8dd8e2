"""Developer tool: the two forms of the flash-attention backward (VH_TUNE_ATTN_BWD: 0 = five products in one kernel + slab
reduce, 1 = two kernels, seven products) back to back at the configs[3] training shapes, on synthetic q/k/v; prints µs per
call, TFLOP/s over the visible pairs priced at FIVE products for both (the algorithmic work), and the largest difference
between the two forms' gradients.   python tools/ab_attn_bwd.py [iters=20]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import _lib, kernels as K  # noqa: E402

DEV = torch.device('cuda:0')


def run(name, B, h, T, mode, xl, kvl, iters):
    d = 64 * h
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B * T, d, generator=g).to(DEV)
    k = torch.randn(B, h, T, 64, generator=g).to(DEV)
    v = torch.randn(B, h, T, 64, generator=g).to(DEV)
    dout = torch.randn(B * T, d, generator=g).to(DEV)
    kv = torch.tensor(kvl, dtype=torch.int32, device=DEV)
    spec = dict(mode=mode, x_len=xl, kv_len=kv)
    out = torch.empty(B * T, d, device=DEV)
    lse2 = torch.empty(B, h, T, device=DEV)
    K.attn_rows(q, k, v, out, B, h, T, T, lse2=lse2, **spec)
    pairs = 0
    for L in kvl:
        if mode == K.MASK_PREFIX:
            pairs += sum(min(L, max(xl, i + 1) if i >= xl else xl) for i in range(T))
        else:
            pairs += T * L
    res = {}
    for form in (1, 0):
        _lib.lib().vh_set_tuning(13, form)
        dqkv = torch.empty(B * T, 3 * d, device=DEV)
        fn = lambda: K.attn_rows_bwd(q, k, v, out, dout, lse2, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, T, **spec)  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
        res[form] = (us, dqkv.clone())
        print(f'{name} B={B} h={h} T={T} form={form}: {us:8.1f} us   {5 * 2 * 64 * pairs * h / us / 1e6:6.1f} TFLOP/s '
              f'(5 products over the visible pairs)', flush=True)
    _lib.lib().vh_set_tuning(13, 0)
    diff = (res[0][1] - res[1][1]).abs().max().item()
    print(f'{name}: max |five-product - two-kernel| = {diff:.3e} (|grad| max {res[1][1].abs().max().item():.3e})', flush=True)


if __name__ == '__main__':
    iters = int(dict(a.split('=') for a in sys.argv[1:]).get('iters', 20))
    g = torch.Generator().manual_seed(0)
    # AR step of configs[3]: 16 rows, text 40..120 padded to its max, codes 225..900 padded to their max; prefix mask
    # over the text (x_len = padded text length), keys up to each row's own length
    T = 120 + 900
    kvl = (120 + torch.randint(225, 901, (16,), generator=g)).tolist()
    run('AR ', 16, 8, T, K.MASK_PREFIX, 120, kvl, iters)
    # NAR step: 80 text + 560 frames, full mask
    run('NAR', 16, 8, 640, K.MASK_FULL, 0, [640] * 16, iters)
    # configs[4] NAR length for one utterance, 16 heads
    run('big', 2, 16, 2875, K.MASK_FULL, 0, [2875] * 2, max(iters // 4, 2))
