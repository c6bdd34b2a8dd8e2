"""Perf mode of the MFMA-bound legs (SECONDARY): the bf16-MFMA kernels of csrc/bf16.hip.

Kernel level: each kernel against fp32 / fp64 torch math ON THE SAME bf16-ROUNDED OPERANDS (the kernel's products are
exact in fp32 accumulators: what may differ is summation order, and — in attention — the bf16 rounding of P), plus
integer-valued operands through the GEMM coming out EXACTLY (MFMA operand / accumulator layout check, asymmetric data).
Model level: teacher-forced logits of the perf-mode forward within SURVEY 8(c)'s atol 5e-2 of the REAL reference's goldens
(`ar_prefill_full`, `nar_full`, `nar_big`); greedy tokens are not asserted in this mode."""
import pytest
import torch
import torch.nn.functional as F

from tests.golden import cases as C
from tests.oracle_runners import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'
# The library's 16-bit operand format ("h16", include/valle_hip.h vh_h16_format): IEEE fp16 by default, bf16 in a -DVH_PERF_BF16
# build.  One rounding to it is half an ulp: 2^-11 relative (fp16) / 2^-8 (bf16) — EPS is that with a factor 2 of slack, as round 5
# had it; MODEL_TOL what the teacher-forced logits of a whole stack may differ from the REAL reference's (SURVEY 8c allows 5e-2; fp16
# measures 2e-3 ... 6e-3, profiles/r6_probe_precision.log).
from valle2_amd._lib import h16_dtype
H16 = h16_dtype()
FP16 = H16 == torch.float16
EPS = 2 ** -10 if FP16 else 2 ** -8
MODEL_TOL = 1.5e-2 if FP16 else 5e-2


def h16(t):
    return t.to(H16)


@pytest.fixture(scope='module')
def K():
    from valle2_amd import kernels
    return kernels


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(params=[1, 3, 4], ids=['two_slabs_k64', 'one_slab_4wg', 'persistent_256sq'])
def gemm_form(request):
    """The tile machines of the perf-mode GEMM (VH_TUNE_BF16_GEMM): two slabs of 64 k / one slab of 64 k with four workgroups per
    CU / the persistent 256 x 256 form of csrc/gemm16p.hip (shapes it does not take — N % 256, K % 128 — fall back to the
    two-slab form).  (Round 5's ring of three slabs was slower on 7 of 8 shapes and is gone.)"""
    from valle2_amd import _lib
    _lib.lib().vh_set_tuning(15, request.param)
    yield request.param
    _lib.lib().vh_set_tuning(15, 0)


def test_to_bf16_rounds_to_nearest_even(K):
    x = torch.randn(37, 264, generator=g(1)) * 3
    t = 2 ** -11 if FP16 else 2 ** -8
    x[0, :4] = torch.tensor([1.0 + t, 1.0 + 3 * t, -0.0, 65280.0])       # exact ties, signed zero
    out = K.to_bf16(x.to(DEV))
    assert torch.equal(out.cpu().view(torch.int16), x.to(H16).view(torch.int16))


@pytest.mark.parametrize('rows,d,ada', [(5, 128, False), (1000, 512, True), (33, 1024, True), (7, 2048, False)])
def test_layernorm_bf16(K, rows, d, ada):
    x = torch.randn(rows, d, generator=g(2)) * 2 + 0.5
    gm, bt = 1 + 0.2 * torch.randn(d, generator=g(3)), 0.2 * torch.randn(d, generator=g(4))
    sc, sh = (1 + 0.1 * torch.randn(d, generator=g(5)), 0.1 * torch.randn(d, generator=g(6))) if ada else (None, None)
    ref = F.layer_norm(x.double(), (d,), gm.double(), bt.double(), 1e-5)
    if ada:
        ref = sc.double() * ref + sh.double()
    out = K.layernorm_bf16(x.to(DEV), gm.to(DEV), bt.to(DEV), ada_scale=None if sc is None else sc.to(DEV),
                           ada_shift=None if sh is None else sh.to(DEV))
    assert out.dtype == H16
    # one bf16 rounding of an fp32-accurate value: half an ulp of bf16 = 2^-9 relative
    torch.testing.assert_close(out.cpu().double(), ref, atol=1e-5, rtol=EPS)


@pytest.mark.parametrize('M,N,K_,out16,act,res', [
    (128, 128, 64, False, 0, False), (300, 256, 128, False, 0, True), (1000, 512, 512, True, 1, False),
    (77, 1536, 512, False, 1, True), (4096, 512, 2048, False, 0, True), (129, 2048, 512, True, 1, False), (1, 128, 64, False, 0, False),
    (200, 128, 192, False, 0, True), (333, 256, 320, True, 0, False)])        # K / 64 = 3, 5: odd numbers of K steps
def test_linear_bf16(K, gemm_form, M, N, K_, out16, act, res):
    a = torch.randn(M, K_, generator=g(10)).to(H16)
    w = (0.05 * torch.randn(N, K_, generator=g(11))).to(H16)
    bias = torch.randn(N, generator=g(12))
    r = torch.randn(M, N, generator=g(13)) if res else None
    ref = a.double() @ w.double().T + bias.double()
    if act:
        ref = F.gelu(ref)
    if res:
        ref = ref + r.double()
    out = K.linear_bf16(a.to(DEV), w.to(DEV), bias.to(DEV), residual=None if r is None else r.to(DEV),
                        act=K.ACT_GELU if act else K.ACT_NONE, out_bf16=out16)
    assert out.dtype == (H16 if out16 else torch.float32)
    if out16:
        # one bf16 rounding (2^-9 relative) of an fp32 sum; with GELU also gelu16_2's 4e-5 absolute / 3e-4 relative
        torch.testing.assert_close(out.cpu().double(), ref, atol=1e-4 if act else 1e-5, rtol=2 * EPS if act else EPS)
    else:
        torch.testing.assert_close(out.cpu().double(), ref, atol=2e-5 * K_ ** 0.5, rtol=1e-5)


@pytest.mark.parametrize('M,N,K_,out16,act,res', [
    (256, 256, 256, False, 0, False), (300, 512, 384, False, 0, True), (1, 256, 256, True, 0, False),
    (16384 + 77, 2048, 512, True, 1, False),      # 520 tiles: up to three per workgroup, the last row of tiles ragged
    (20000, 1536, 512, False, 1, True),           # 474 tiles, fp32 + GELU + residual, ragged
    (65536, 512, 512, True, 0, False), (8192, 1024, 1024, False, 0, True), (5000, 4096, 1024, True, 1, False)])
def test_linear_bf16_persistent_256(K, M, N, K_, out16, act, res):
    """csrc/gemm16p.hip on its own shapes (N % 256 == 0, K % 128 == 0): one tile, ragged rows, many tiles per persistent workgroup
    (the request stream and the counted waits run across tile boundaries and epilogues).  Reference in fp64 on the device."""
    from valle2_amd import _lib
    a = torch.randn(M, K_, generator=g(10)).to(H16).to(DEV)
    w = (0.05 * torch.randn(N, K_, generator=g(11))).to(H16).to(DEV)
    bias = torch.randn(N, generator=g(12)).to(DEV)
    r = torch.randn(M, N, generator=g(13)).to(DEV) if res else None
    ref = a.double() @ w.double().T + bias.double()
    if act:
        ref = F.gelu(ref)
    if res:
        ref = ref + r.double()
    _lib.lib().vh_set_tuning(15, 4)
    try:
        out = K.linear_bf16(a, w, bias, residual=r, act=K.ACT_GELU if act else K.ACT_NONE, out_bf16=out16)
        out2 = K.linear_bf16(a, w, bias, residual=r, act=K.ACT_GELU if act else K.ACT_NONE, out_bf16=out16)
    finally:
        _lib.lib().vh_set_tuning(15, 0)
    assert torch.equal(out, out2)                 # the same bits on every launch (no race between request stream and reads)
    if out16:
        torch.testing.assert_close(out.double(), ref, atol=1e-4 if act else 1e-5, rtol=2 * EPS if act else EPS)
    else:
        torch.testing.assert_close(out.double(), ref, atol=2e-5 * K_ ** 0.5, rtol=1e-5)


def test_linear_bf16_persistent_256_random_shapes_and_repeatable_bits(K):
    """Property test of csrc/gemm16p.hip: 24 seeded draws of (M, N, K, output kind) — M from 1 to 40 000 (ragged last tile rows,
    one to several tiles per persistent workgroup), N / 256 in 1..16, K / 128 in 2..32 (4 to 64 K-tiles: every length of the
    request stream's head and tail), fp32 + residual / 16-bit / 16-bit + GELU — against fp64 on the device, each launched three
    times with bit-identical results (a read that overtakes its LDS-DMA, or a request that overtakes the last read of its
    buffer, shows as a run-to-run difference long before it shows as a wrong sum)."""
    import random
    from valle2_amd import _lib
    rng = random.Random(2026)
    _lib.lib().vh_set_tuning(15, 4)
    try:
        for draw in range(24):
            M = rng.choice([1, 255, 256, 257, 511, 4096, 4097, rng.randrange(1, 40000), rng.randrange(1, 40000)])
            N = 256 * rng.randrange(1, 17)
            K_ = 128 * rng.randrange(2, 33)
            if M * N * K_ > 3e11:                     # keep a draw under ~0.3 s of fp64 on the device
                K_ = max(256, 128 * int(3e11 / (M * N) // 128))
            kind = rng.randrange(3)
            a = torch.randn(M, K_, generator=g(1000 + draw)).to(H16).to(DEV)
            w = (torch.randn(N, K_, generator=g(2000 + draw)) * K_ ** -0.5).to(H16).to(DEV)
            bias = torch.randn(N, generator=g(3000 + draw)).to(DEV)
            r = torch.randn(M, N, generator=g(4000 + draw)).to(DEV) if kind == 0 else None
            ref = a.double() @ w.double().T + bias.double()
            if kind == 2:
                ref = F.gelu(ref)
            if r is not None:
                ref = ref + r.double()
            outs = [K.linear_bf16(a, w, bias, residual=r, act=K.ACT_GELU if kind == 2 else K.ACT_NONE, out_bf16=kind != 0)
                    for _ in range(3)]
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (draw, M, N, K_, kind)
            if kind == 0:
                torch.testing.assert_close(outs[0].double(), ref, atol=2e-5 * K_ ** 0.5, rtol=1e-5, msg=f'draw {draw}: {M}x{N}x{K_}')
            else:
                torch.testing.assert_close(outs[0].double(), ref, atol=1e-4 if kind == 2 else 1e-5, rtol=2 * EPS if kind == 2 else EPS,
                                           msg=f'draw {draw}: {M}x{N}x{K_} kind {kind}')
    finally:
        _lib.lib().vh_set_tuning(15, 0)


def test_linear_bf16_integer_operands_are_exact(K, gemm_form):
    """Small integers are exact in bf16 and their products / sums exact in fp32: any operand-layout or accumulator-map
    error shows as a wrong integer.  Asymmetric data (a[m][k] depends on m and k differently than w[n][k] on n and k)."""
    M, N, K_ = (520, 512, 256) if gemm_form == 4 else (260, 256, 192)
    m, n, k = torch.arange(M)[:, None], torch.arange(N)[:, None], torch.arange(K_)[None, :]
    a = ((3 * m + 5 * k) % 7 - 3).float()
    w = ((2 * n + k) % 5 - 2).float()
    ref = a @ w.T
    out = K.linear_bf16(a.to(H16).to(DEV), w.to(H16).to(DEV))
    assert torch.equal(out.cpu(), ref)
    out16 = K.linear_bf16(a.to(H16).to(DEV), w.to(H16).to(DEV), out_bf16=True)
    assert torch.equal(out16.cpu().float(), ref.to(H16).float())


def test_linear_bf16_refuses_shapes_outside_the_tile_kernel(K):
    from valle2_amd._lib import VhError
    a = torch.zeros(8, 96, device=DEV, dtype=H16)
    with pytest.raises(VhError, match='N % 128'):
        K.linear_bf16(a, torch.zeros(128, 96, device=DEV, dtype=H16))
    with pytest.raises(VhError, match='float16'):
        K.linear_bf16(a.float(), torch.zeros(128, 96, device=DEV, dtype=H16))


@pytest.mark.parametrize('B,T,h,with_len', [(2, 5, 2, False), (3, 150, 4, True), (1, 1000, 8, False), (40, 7, 2, True),
                                           (40, 7, 8, True), (33, 1024, 8, False), (9, 700, 16, True)])
def test_linear_qkv_bf16_scatter(K, gemm_form, B, T, h, with_len):
    d = 64 * h
    S_max = T + 20
    a = torch.randn(B * T, d, generator=g(20)).to(H16)
    w = (0.1 * torch.randn(3 * d, d, generator=g(21))).to(H16)
    ref = (a.double() @ w.double().T).float()
    cl = torch.tensor([(3 * i) % 17 for i in range(B)], dtype=torch.int32) if with_len else None
    kc = torch.zeros(B, h, S_max, 64, device=DEV, dtype=H16)
    vc = torch.zeros_like(kc)
    q = torch.empty(B * T, d, device=DEV, dtype=H16)
    K.linear_qkv_bf16(a.to(DEV), w.to(DEV), q, kc, vc, B, T, h, cache_len=None if cl is None else cl.to(DEV))
    torch.testing.assert_close(q.cpu().float(), ref[:, :d] * K.Q16_PRESCALE, atol=1e-5, rtol=EPS)     # q leaves pre-scaled
    kref = ref[:, d:2 * d].view(B, T, h, 64).permute(0, 2, 1, 3)
    vref = ref[:, 2 * d:].view(B, T, h, 64).permute(0, 2, 1, 3)
    for b in range(B):
        p0 = 0 if cl is None else int(cl[b])
        torch.testing.assert_close(kc[b, :, p0:p0 + T].cpu().float(), kref[b], atol=1e-5, rtol=EPS)
        torch.testing.assert_close(vc[b, :, p0:p0 + T].cpu().float(), vref[b], atol=1e-5, rtol=EPS)
        assert float(kc[b, :, :p0].abs().sum()) == 0 and float(kc[b, :, p0 + T:].abs().sum()) == 0


@pytest.mark.parametrize('B,h,T,mode', [(2, 2, 5, 'prefix'), (3, 2, 150, 'prefix'), (2, 8, 300, 'full'), (2, 2, 129, 'full'),
                                        (1, 1, 64, 'prefix'), (2, 4, 1000, 'prefix'), (1, 2, 2875, 'full')])
def test_attn_rows_bf16(K, B, h, T, mode):
    """Against double-precision attention over the same bf16 q / K / V.  The kernel rounds P to bf16 before P V (relative
    2^-9 per weight): the tolerance is that rounding on O(1) values, not the fp32 path's 3e-5."""
    from oracle.valle_oracle import build_attn_mask
    d = 64 * h
    q = (torch.randn(B, T, d, generator=g(40)) * K.Q16_PRESCALE).to(H16)      # the kernel's q' = q / 8 * log2(e), rounded once
    k = torch.randn(B, h, T, 64, generator=g(41)).to(H16)
    v = torch.randn(B, h, T, 64, generator=g(42)).to(H16)
    xl = T // 3
    kvl = torch.tensor([T - (5 * i) % (T // 2 + 1) for i in range(B)], dtype=torch.int32)
    keypad = torch.arange(T)[None, :] >= kvl[:, None]
    masked = (build_attn_mask(xl, T - xl)[None] | keypad[:, None, :]) if mode == 'prefix' else keypad[:, None, :].expand(B, T, T)
    qh = q.view(B, T, h, 64).permute(0, 2, 1, 3).double() / K.Q16_PRESCALE
    ref = F.scaled_dot_product_attention(qh, k.double(), v.double(), attn_mask=~masked[:, None])
    ref = ref.permute(0, 2, 1, 3).reshape(B * T, d)
    S_max = T + 9
    kc = torch.full((B, h, S_max, 64), float('nan'), dtype=H16)
    vc = torch.full((B, h, S_max, 64), float('nan'), dtype=H16)       # garbage beyond T must never be read as a key
    kc[:, :, :T], vc[:, :, :T] = k, v
    out = torch.full((B * T, d), float('nan'), device=DEV, dtype=H16)
    kw = dict(mode=K.MASK_PREFIX, x_len=xl, kv_len=kvl.to(DEV)) if mode == 'prefix' else dict(mode=K.MASK_FULL, kv_len=kvl.to(DEV))
    K.attn_rows_bf16(q.view(B * T, d).to(DEV), kc.to(DEV), vc.to(DEV), out, B, h, T, T, **kw)
    assert bool(torch.isfinite(out.float()).all())
    torch.testing.assert_close(out.cpu().double(), ref, atol=1e-3 if FP16 else 6e-3, rtol=2 * EPS)


def test_attn_rows_bf16_per_row_text_lengths(K):
    """A ragged batch (ValleAR.generate_batch with perf_mode: per-row text length x_len_dev AND per-row key length kv_len):
    row b's prefix-LM mask is build_attn_mask(xl[b], .) over its own kv_len[b] keys."""
    from oracle.valle_oracle import build_attn_mask
    B, h, T = 4, 2, 200
    d = 64 * h
    q = (torch.randn(B, T, d, generator=g(47)) * K.Q16_PRESCALE).to(H16)
    k = torch.randn(B, h, T, 64, generator=g(48)).to(H16)
    v = torch.randn(B, h, T, 64, generator=g(49)).to(H16)
    xl = torch.tensor([10, 64, 1, 130], dtype=torch.int32)
    kvl = torch.tensor([200, 150, 77, 131], dtype=torch.int32)
    out = torch.empty(B * T, d, device=DEV, dtype=H16)
    K.attn_rows_bf16(q.view(B * T, d).to(DEV), k.to(DEV), v.to(DEV), out, B, h, T, T, mode=K.MASK_PREFIX, x_len_dev=xl.to(DEV),
                     kv_len=kvl.to(DEV))
    out = out.cpu().double().view(B, T, d)
    for b in range(B):
        n = int(kvl[b])
        masked = build_attn_mask(int(xl[b]), T - int(xl[b])) | (torch.arange(T)[None, :] >= n)
        qh = q[b].view(T, h, 64).permute(1, 0, 2).double() / K.Q16_PRESCALE
        ref = F.scaled_dot_product_attention(qh[None], k[b:b + 1].double(), v[b:b + 1].double(), attn_mask=~masked[None, None])
        ref = ref[0].permute(1, 0, 2).reshape(T, d)
        torch.testing.assert_close(out[b, :n], ref[:n], atol=1e-3 if FP16 else 6e-3, rtol=2 * EPS)          # rows beyond the row's length: don't care


def test_attn_rows_bf16_peaked_softmax(K):
    """|q.k| in the hundreds at a late tile: the online-softmax rescale branch."""
    B, h, T = 1, 2, 200
    k = torch.randn(B, h, T, 64, generator=g(44))
    v = torch.randn(B, h, T, 64, generator=g(45)).to(H16)
    q = (torch.randn(B, T, h, 64, generator=g(46)) * K.Q16_PRESCALE).to(H16)
    k[:, :, 170] *= 40.0
    k[:, :, 3] *= 15.0
    k = k.to(H16)
    out = torch.empty(B * T, 64 * h, device=DEV, dtype=H16)
    K.attn_rows_bf16(q.reshape(B * T, -1).to(DEV), k.to(DEV), v.to(DEV), out, B, h, T, T, mode=K.MASK_FULL)
    ref = F.scaled_dot_product_attention(q.permute(0, 2, 1, 3).double() / K.Q16_PRESCALE, k.double(), v.double())
    torch.testing.assert_close(out.cpu().double(), ref.permute(0, 2, 1, 3).reshape(B * T, -1), atol=1e-3 if FP16 else 6e-3, rtol=2 * EPS)


@pytest.mark.parametrize('step', [5.0, 9.0, -6.0])
def test_attn_rows_bf16_drifting_maximum(K, step):
    """The row's reference point moves lazily (round 6: only when a tile's maximum exceeds it by more than 2^8): scores whose
    maximum drifts by `step` base-2 exponents per 64-key tile — below the threshold (the reference point lags: weights up to 2^8),
    above it (it moves every tile), and downwards (the first tile holds the maximum: it never moves again)."""
    B, h, T = 2, 2, 640
    d = 64 * h
    q = 0.05 * torch.randn(B, T, h, 64, generator=g(60))
    k = 0.05 * torch.randn(B, h, T, 64, generator=g(61))
    v = torch.randn(B, h, T, 64, generator=g(62)).to(H16)
    q[..., 0] = 1.0                                       # exponent of key j (pre-scaled q): ~ step * j / 64
    k[..., 0] = step * torch.arange(T) / 64.0
    q, k = q.to(H16), k.to(H16)
    out = torch.empty(B * T, d, device=DEV, dtype=H16)
    K.attn_rows_bf16(q.reshape(B * T, d).to(DEV), k.to(DEV), v.to(DEV), out, B, h, T, T, mode=K.MASK_FULL)
    ref = F.scaled_dot_product_attention(q.permute(0, 2, 1, 3).double() / K.Q16_PRESCALE, k.double(), v.double())
    assert bool(torch.isfinite(out.float()).all())
    torch.testing.assert_close(out.cpu().double(), ref.permute(0, 2, 1, 3).reshape(B * T, d), atol=2e-3 if FP16 else 1e-2, rtol=2 * EPS)


# ---- model level: the perf-mode forward against the REAL reference's goldens (atol 5e-2) --------------------
def build(name, kw, sd):
    from valle2_amd import get_model_class
    m = get_model_class(name)(C.cfg_of(kw))
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def test_perf_mode_prefill_logits_within_tolerance_of_the_reference():
    gold = load_golden('ar_prefill_full')
    kw, sd, text, codes, pos = C.ar_prefill_full_inputs()
    m = build('ValleAR', kw, sd)
    b = text.shape[0]
    batch = {'tokens': text, 'tokens_lens': torch.full((b,), text.shape[1]), 'codes': codes,
             'codes_lens': torch.full((b,), codes.shape[1])}
    with torch.no_grad():
        logits = m.forward_logits(batch, perf_mode=True)
        exact = m.forward_logits(batch)
    err = float((logits[:, pos.to(DEV)].cpu() - gold['logits']).abs().max())
    print(f'perf-mode prefill (configs[1], 12L/512d): max |logit error| vs the reference = {err:.2e} '
          f'(parity path {float((exact[:, pos.to(DEV)].cpu() - gold["logits"]).abs().max()):.2e})')
    assert err < MODEL_TOL, err


def test_both_gemm_forms_give_the_same_stack_output():
    """The 128^2 tile machines accumulate k in the same order (32x32x16 MFMAs over ascending k) and add the bias last: the whole
    stack's output is bit-identical between them.  The persistent 256^2 form (the default where the shape allows it) has the same
    k order but starts its accumulators FROM the bias (one value per lane, no add in the epilogue): fp32 roundings differ in the
    last bit, a bf16 rounding downstream flips here and there — its logits agree to a bf16 ulp of the activations, no more."""
    from valle2_amd import _lib
    kw, sd, batch = C.nar_inputs()
    m = build('ValleNAR', kw, sd)
    outs = {}
    for form in (0, 1, 3, 4):
        _lib.lib().vh_set_tuning(15, form)
        try:
            with torch.no_grad():
                outs[form] = m.stage_logits(batch, 3, perf_mode=True)[0].clone()
        finally:
            _lib.lib().vh_set_tuning(15, 0)
    assert torch.equal(outs[1], outs[3])
    for form in (0, 4):
        err = float((outs[form] - outs[1]).abs().max())
        assert err < 2e-2, (form, err)


def test_perf_mode_nar_stage_logits_within_tolerance_of_the_reference():
    gold = load_golden('nar_full')
    kw, sd, batch = C.nar_full_inputs()
    m = build('ValleNAR', kw, sd)
    with torch.no_grad():
        logits, p = m.stage_logits(batch, C.NAR_FULL_STAGE, perf_mode=True)
    assert p == int(gold['prefix'])
    got = logits[list(C.NAR_FULL_ROWS)][:, ::C.NAR_FULL_STRIDE].cpu()
    err = float((got - gold['logits']).abs().max())
    print(f'perf-mode NAR stage (configs[2], 64 x 1024): max |logit error| vs the reference = {err:.2e}')
    assert err < MODEL_TOL, err


def test_perf_mode_nar_big_logits_within_tolerance_of_the_reference():
    gold = load_golden('nar_big')
    kw, sd, batch = C.nar_big_inputs()
    m = build('ValleNAR', kw, sd)
    for stage in (2, 7):
        with torch.no_grad():
            logits, p = m.stage_logits(batch, stage, perf_mode=True)
        got = logits[:, ::C.NAR_BIG_STRIDE].cpu()
        err = float((got - gold[f'logits_{stage}']).abs().max())
        print(f'perf-mode NAR stage {stage} (configs[4], 24L/1024d, 2875 positions): max |logit error| = {err:.2e}')
        assert err < MODEL_TOL, err


def test_perf_mode_generate_batch_ragged_rows_decode_like_the_parity_path():
    """generate_batch(perf_mode=True) on RAGGED rows (per-row text / prompt lengths through the bf16 prompt pass, then decode
    over the bf16 cache it wrote): on the well-separated goldens' model the greedy tokens equal the fp32 run's for (almost)
    every row and step; the stats say which prompt pass ran."""
    from valle2_amd import synth
    kw = dict(C.MID, norm='LayerNorm', num_beams=1, top_k=1, max_audio_len=24)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=5, rich=True), cfg)
    m = build('ValleAR', kw, sd)
    utts = [synth.synth_utterance(cfg, 5 + 3 * i, 4 + i, 20 + 7 * i, seed=40 + i) for i in range(32)]     # 32 rows x 8 heads: one (row, head) per CU
    texts = [torch.cat([u[0], u[2]]).to(DEV) for u in utts]
    firsts = [u[1][:, 0].to(DEV) for u in utts]
    a = m.generate_batch(texts, firsts)
    assert not m.last_generate_stats['prefill_bf16']
    b = m.generate_batch(texts, firsts, perf_mode=True)
    st = m.last_generate_stats
    assert st['prefill_bf16'] and st['kv_bf16']
    agree = float((a == b).float().mean())
    print(f'ragged perf-mode generate: {agree:.4f} of the tokens equal the fp32 run')
    assert agree > 0.9
