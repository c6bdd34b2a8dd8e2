"""GPU parity of each C-ABI primitive against plain torch fp32 on the CPU (same seeded inputs).

Tolerances (fp32 everywhere; the only differences are summation order and exp/erf/rsqrt ulps):
GEMM/attention outputs atol 2e-5 + rtol 2e-5 on O(1) values unless stated; integer-valued
operands must come out exactly (that is the MFMA operand/accumulator layout check: asymmetric
integer data, cdna_hip_programming.md §3)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(scope='module')
def K():
    from valle2_amd import kernels
    return kernels


@pytest.fixture(params=[1, 2], ids=['staged', 'lds-dma'])
def tile_staging(request):
    """Run a large-M GEMM test under both operand-staging kernels (VH_TUNE_TILE_DMA: 1 = through registers,
    2 = LDS-DMA whenever K % 32 == 0); the default (0) picks between them by shape."""
    from valle2_amd import _lib
    _lib.lib().vh_set_tuning(4, request.param)
    yield request.param
    _lib.lib().vh_set_tuning(4, 0)


def close(a, b, atol=2e-5, rtol=2e-5):
    torch.testing.assert_close(a.cpu(), b, atol=atol, rtol=rtol)


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K_', [(5, 16, 16), (32, 1536, 512), (32, 512, 2048), (4, 1025, 128),
                                    (64, 48, 64), (1, 256, 128), (17, 20, 1040)])
def test_linear_skinny_integer_exact(K, M, N, K_):
    a = torch.randint(-3, 4, (M, K_), generator=g(1)).float()
    w = torch.randint(-3, 4, (N, K_), generator=g(2)).float()
    w[:, 0] += torch.arange(N).float() % 5          # asymmetric
    out = K.linear(a.to(DEV), w.to(DEV))
    assert torch.equal(out.cpu(), a @ w.T)


@pytest.mark.parametrize('M,N,K_', [(65, 16, 16), (128, 128, 32), (300, 1025, 128), (257, 1536, 512),
                                    (1000, 100, 2048), (65, 16, 32), (100, 300, 64), (129, 4, 96)])
def test_linear_tile_integer_exact(K, M, N, K_, tile_staging):
    a = torch.randint(-3, 4, (M, K_), generator=g(3)).float()
    w = torch.randint(-3, 4, (N, K_), generator=g(4)).float()
    w[:, 1] += torch.arange(N).float() % 7
    out = K.linear(a.to(DEV), w.to(DEV))
    assert torch.equal(out.cpu(), a @ w.T)


@pytest.mark.parametrize('M,N,K_', [(65, 16, 16), (256, 256, 32), (300, 1025, 128), (513, 1536, 512),
                                    (1000, 100, 2048)])
def test_linear_tile_epilogues(K, M, N, K_, tile_staging):
    """bias / GELU / in-place residual through the LDS-transposed float4 epilogue, ragged M and N."""
    a = torch.randn(M, K_, generator=g(5))
    w = 0.05 * torch.randn(N, K_, generator=g(6))
    bias, res = torch.randn(N, generator=g(7)), torch.randn(M, N, generator=g(8))
    for act in (0, 1):
        ref = F.linear(a, w, bias)
        ref = (F.gelu(ref) if act else ref) + res
        resd = torch.zeros(M, (N + 3) // 4 * 4, device=DEV)[:, :N]    # row stride a multiple of 4
        resd.copy_(res)
        out = K.linear(a.to(DEV), w.to(DEV), bias.to(DEV), resd, out=resd, act=act)   # in place
        close(out, ref, atol=5e-5)
    close(K.linear(a.to(DEV), w.to(DEV), bias.to(DEV)), F.linear(a, w, bias), atol=5e-5)


def test_linear_qkv_scatter_large_m(K, tile_staging):
    B, T, h = 3, 200, 4
    d = 64 * h
    S_max = 260
    a = torch.randn(B * T, d, generator=g(60)).to(DEV)
    w = (0.1 * torch.randn(3 * d, d, generator=g(61))).to(DEV)
    cl = torch.tensor([2, 0, 60], dtype=torch.int32, device=DEV)
    kc = torch.zeros(B, h, S_max, 64, device=DEV)
    vc = torch.zeros_like(kc)
    qo = torch.zeros(B * T, d, device=DEV)
    K.linear_qkv(a, w, qo, kc, vc, B, T, h, cache_len=cl)
    ref = F.linear(a.cpu(), w.cpu())
    close(qo, ref[:, :d], atol=5e-5)
    kref = ref[:, d:2 * d].view(B, T, h, 64).permute(0, 2, 1, 3)
    vref = ref[:, 2 * d:].view(B, T, h, 64).permute(0, 2, 1, 3)
    for b in range(B):
        p0 = int(cl[b])
        close(kc[b, :, p0:p0 + T], kref[b], atol=5e-5)
        close(vc[b, :, p0:p0 + T], vref[b], atol=5e-5)
        assert float(kc[b, :, :p0].abs().sum()) == 0 and float(kc[b, :, p0 + T:].abs().sum()) == 0


@pytest.mark.parametrize('M,N,K_', [(384, 256, 64), (300, 260, 96), (1000, 512, 512)])
def test_linear_tile_strided_operands(K, M, N, K_, tile_staging):
    """A, the residual and the output as column slices of wider buffers (row strides != K, N), interior and edge
    tiles: the tile kernels address rows as base + row * stride with 32-bit per-lane offsets."""
    abig = torch.randn(M, K_ + 24, generator=g(310)).to(DEV)
    a = abig[:, 8:8 + K_]                                   # 32-byte offset, stride K + 24
    w = (0.05 * torch.randn(N, K_, generator=g(311))).to(DEV)
    bias = torch.randn(N, generator=g(312)).to(DEV)
    rbig = torch.randn(M, N + 12, generator=g(313)).to(DEV)
    res = rbig[:, 4:4 + N]
    obig = torch.full((M, N + 20), 7.0, device=DEV)
    out = obig[:, 16:16 + N]
    K.linear(a, w, bias, res, out=out, act=1)
    ref = F.gelu(F.linear(a.cpu(), w.cpu(), bias.cpu())) + res.cpu()
    close(out, ref, atol=5e-5)
    assert float((obig[:, :16] - 7.0).abs().max()) == 0 and float((obig[:, 16 + N:] - 7.0).abs().max()) == 0


@pytest.mark.parametrize('M,N,K_', [(257, 1536, 512), (1000, 100, 2048), (6144, 2048, 64), (130, 132, 96)])
def test_linear_tile_staging_kernels_bit_identical(K, M, N, K_):
    """The register-staged and the LDS-DMA tile kernels accumulate K in the same order: same bits."""
    from valle2_amd import _lib
    a = torch.randn(M, K_, generator=g(300)).to(DEV)
    w = (0.05 * torch.randn(N, K_, generator=g(301))).to(DEV)
    bias = torch.randn(N, generator=g(302)).to(DEV)
    res = torch.randn(M, N, generator=g(303)).to(DEV)
    outs = []
    for mode in (1, 2, 0):
        _lib.lib().vh_set_tuning(4, mode)
        outs.append(K.linear(a, w, bias, res, act=1).clone())
    _lib.lib().vh_set_tuning(4, 0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    close(outs[0], F.gelu(F.linear(a.cpu(), w.cpu(), bias.cpu())) + res.cpu(), atol=5e-5)


@pytest.mark.parametrize('M,N,K_', [(32, 512, 2048), (16, 512, 2048), (4, 1024, 4096), (33, 100, 1280),
                                    (32, 512, 512), (8, 48, 3072)])
def test_linear_splitk_workspace_path(K, M, N, K_):
    # wide-K split-K + fixed-order reduce: integer data exact; random data with the full epilogue
    a = torch.randint(-3, 4, (M, K_), generator=g(80)).float()
    w = torch.randint(-3, 4, (N, K_), generator=g(81)).float()
    w[:, 2] += torch.arange(N).float() % 5
    assert torch.equal(K.linear_ws(a.to(DEV), w.to(DEV)).cpu(), a @ w.T)
    a = torch.randn(M, K_, generator=g(82))
    w = 0.05 * torch.randn(N, K_, generator=g(83))
    bias, res = torch.randn(N, generator=g(84)), torch.randn(M, N, generator=g(85))
    resd = res.to(DEV)
    out = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV), resd, out=resd, act=1)
    close(out, F.gelu(F.linear(a, w, bias)) + res, atol=5e-5)
    # bitwise reproducible (fixed-order reduction, no atomics)
    o1 = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV))
    o2 = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV))
    assert torch.equal(o1, o2)


@pytest.mark.parametrize('M,N,K_', [(884, 1024, 4096), (300, 512, 2048), (129, 1025, 1024), (1137, 1024, 1024),
                                    (700, 128, 1536)])
def test_linear_splitk_tile_path(K, M, N, K_, tile_staging):
    """65 <= M with few 128x128 tiles and a long K: K slices in the tile kernel + fixed-order reduce."""
    from valle2_amd import _lib
    assert _lib.lib().vh_linear_ws_bytes(M, N, K_) > 0
    a = torch.randint(-3, 4, (M, K_), generator=g(180)).float()
    w = torch.randint(-3, 4, (N, K_), generator=g(181)).float()
    w[:, 2] += torch.arange(N).float() % 5
    assert torch.equal(K.linear_ws(a.to(DEV), w.to(DEV)).cpu(), a @ w.T)
    a = torch.randn(M, K_, generator=g(182))
    w = torch.randn(N, K_, generator=g(183)) / K_ ** 0.5
    bias, res = torch.randn(N, generator=g(184)), torch.randn(M, N, generator=g(185))
    resd = torch.zeros(M, (N + 3) // 4 * 4, device=DEV)[:, :N]
    resd.copy_(res)
    out = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV), resd, out=resd, act=1)
    close(out, F.gelu(F.linear(a, w, bias)) + res, atol=5e-5)
    o1 = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV))
    o2 = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV))
    assert torch.equal(o1, o2)                               # fixed-order reduction
    close(o1, K.linear(a.to(DEV), w.to(DEV), bias.to(DEV)).cpu(), atol=5e-5)


@pytest.mark.parametrize('M', [7, 32, 200])
@pytest.mark.parametrize('act', [0, 1])
def test_linear_epilogues(K, M, act):
    N, K_ = 520, 128
    a = torch.randn(M, K_, generator=g(5))
    w = 0.1 * torch.randn(N, K_, generator=g(6))
    bias = torch.randn(N, generator=g(7))
    res = torch.randn(M, N, generator=g(8))
    ref = F.linear(a, w, bias)
    if act:
        ref = F.gelu(ref)
    ref = ref + res
    resd = res.to(DEV)
    out = K.linear(a.to(DEV), w.to(DEV), bias.to(DEV), resd, out=resd, act=act)  # in place
    close(out, ref)


@pytest.mark.parametrize('M,ada', [(3, False), (32, False), (32, True), (64, True)])
def test_linear_fused_layernorm(K, M, ada):
    N, K_ = 96, 512
    a = 2.0 * torch.randn(M, K_, generator=g(9)) + 0.5
    w = 0.1 * torch.randn(N, K_, generator=g(10))
    gm, bt = 1 + 0.1 * torch.randn(K_, generator=g(11)), 0.1 * torch.randn(K_, generator=g(12))
    sc, sh = 1 + 0.1 * torch.randn(K_, generator=g(13)), 0.1 * torch.randn(K_, generator=g(14))
    xn = F.layer_norm(a, (K_,), gm, bt, 1e-5)
    if ada:
        xn = sc * xn + sh
    ref = F.linear(xn, w)
    ln = (gm.to(DEV), bt.to(DEV), sc.to(DEV) if ada else None, sh.to(DEV) if ada else None, 1e-5)
    close(K.linear(a.to(DEV), w.to(DEV), ln=ln), ref, atol=5e-5)


@pytest.mark.parametrize('M,N,K_,act', [(3, 96, 128, 0), (32, 2048, 512, 1), (32, 1536, 512, 0), (64, 64, 1024, 1),
                                        (17, 512, 256, 0)])
def test_linear_folded_layernorm(K, M, N, K_, act):
    """LN(x) W^T + b rebuilt as rstd (x Wf^T - mean c1) + c2 (vh_ln_fold + vh_linear_folded)."""
    a = 2.0 * torch.randn(M, K_, generator=g(9)) + 0.5
    a[0] += 3.0                                            # a row whose mean exceeds its deviation
    w = 0.1 * torch.randn(N, K_, generator=g(10))
    gm, bt = 1 + 0.1 * torch.randn(K_, generator=g(11)), 0.1 * torch.randn(K_, generator=g(12))
    bias, res = torch.randn(N, generator=g(13)), torch.randn(M, N, generator=g(14))
    ref = F.linear(F.layer_norm(a.double(), (K_,), gm.double(), bt.double(), 1e-5), w.double(), bias.double())
    ref = ((F.gelu(ref) if act else ref) + res.double()).float()
    folded = K.ln_fold(w.to(DEV), gm.to(DEV), bt.to(DEV), bias.to(DEV))
    torch.testing.assert_close(folded[0].cpu(), w * gm, atol=0, rtol=0)
    torch.testing.assert_close(folded[1].cpu(), (w * gm).double().sum(1).float(), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(folded[2].cpu(), (w.double() @ bt.double() + bias.double()).float(), atol=1e-6, rtol=1e-6)
    resd = res.to(DEV)
    out = K.linear_folded(a.to(DEV), folded, residual=resd, out=resd, act=act)
    close(out, ref, atol=5e-5)
    # and it agrees with the operand-load LayerNorm kernel to the same tolerance
    ln = (gm.to(DEV), bt.to(DEV), None, None, 1e-5)
    close(K.linear(a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), act=act, ln=ln), ref, atol=5e-5)


def test_linear_folded_rejects_unsupported_shapes(K):
    from valle2_amd._lib import VhError
    w = torch.randn(96, 384, device=DEV)
    folded = K.ln_fold(w, torch.ones(384, device=DEV), torch.zeros(384, device=DEV))
    with pytest.raises(VhError, match='folded LayerNorm'):
        K.linear_folded(torch.randn(4, 384, device=DEV), folded)
    w = torch.randn(96, 512, device=DEV)
    folded = K.ln_fold(w, torch.ones(512, device=DEV), torch.zeros(512, device=DEV))
    with pytest.raises(VhError, match='folded LayerNorm'):
        K.linear_folded(torch.randn(65, 512, device=DEV), folded)


def test_linear_qkv_folded_scatter(K):
    B, h, S_max = 5, 4, 20
    d = 64 * h
    x = torch.randn(B, d, generator=g(62)) + 0.3
    w = 0.1 * torch.randn(3 * d, d, generator=g(61))
    gm, bt = 1 + 0.2 * torch.randn(d, generator=g(63)), 0.2 * torch.randn(d, generator=g(64))
    ref1 = F.linear(F.layer_norm(x, (d,), gm, bt, 1e-5), w)
    kc = torch.zeros(B, h, S_max, 64, device=DEV)
    vc = torch.zeros_like(kc)
    qo = torch.empty(B, d, device=DEV)
    cl = torch.tensor([3, 11, 0, 19, 7], dtype=torch.int32)
    folded = K.ln_fold(w.to(DEV), gm.to(DEV), bt.to(DEV))
    K.linear_qkv_folded(x.to(DEV), folded, qo, kc, vc, B, 1, h, cache_len=cl.to(DEV))
    close(qo, ref1[:, :d], atol=5e-5)
    for b in range(B):
        close(kc[b, :, int(cl[b])], ref1[b, d:2 * d].view(h, 64), atol=5e-5)
        close(vc[b, :, int(cl[b])], ref1[b, 2 * d:].view(h, 64), atol=5e-5)
    assert int((kc != 0).any(-1).sum()) == B * h       # exactly one row per (b, head) written


def test_linear_random_shape_sweep(K):
    """Every dispatch path of vh_linear (compact / guarded skinny kernels, row groups, tile kernel and
    its ragged edges) on 60 seeded random shapes with the full epilogue, against torch fp32 on the CPU."""
    gen = torch.Generator().manual_seed(2024)
    ks = [16, 32, 48, 64, 128, 144, 256, 384, 512, 1024, 1040, 2048]
    for case in range(60):
        M = int(torch.randint(1, 330, (1,), generator=gen))
        N = int(torch.randint(1, 1100, (1,), generator=gen))
        K_ = ks[int(torch.randint(0, len(ks), (1,), generator=gen))]
        act = int(torch.randint(0, 2, (1,), generator=gen))
        a = torch.randn(M, K_, generator=gen)
        w = torch.randn(N, K_, generator=gen) / K_ ** 0.5
        bias = torch.randn(N, generator=gen) if case % 3 else None
        res = torch.randn(M, N, generator=gen) if case % 2 else None
        ref = F.linear(a, w, bias)
        ref = F.gelu(ref) if act else ref
        if res is not None:
            ref = ref + res
        resd = None
        if res is not None:
            resd = torch.zeros(M, (N + 3) // 4 * 4, device=DEV)[:, :N]
            resd.copy_(res)
        out = K.linear(a.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV), resd, act=act)
        torch.testing.assert_close(out.cpu(), ref, atol=5e-5, rtol=5e-5, msg=lambda m: f'M={M} N={N} K={K_} act={act}: {m}')


@pytest.mark.parametrize('M', [17, 32, 48, 64])
def test_decode_gemm_row_groups_are_bit_identical(K, M):
    """16 < M <= 64: one workgroup per (16 rows, 16 columns) (default) against one per 16 columns."""
    from valle2_amd import _lib
    lib = _lib.lib()
    d, dff, h = 256, 512, 4
    gen = torch.Generator().manual_seed(M)
    x = (torch.randn(M, d, generator=gen) + 0.3).to(DEV)
    wq, w1, wo = (0.05 * torch.randn(n, d, generator=gen) for n in (3 * d, dff, d))
    gm, bt = 1 + 0.1 * torch.randn(d, generator=gen), 0.1 * torch.randn(d, generator=gen)
    b1, bo = torch.randn(dff, generator=gen), torch.randn(d, generator=gen)
    wh = (0.05 * torch.randn(1025, d, generator=gen)).to(DEV)
    res0 = torch.randn(M, d, generator=gen)
    fq = K.ln_fold(wq.to(DEV), gm.to(DEV), bt.to(DEV))
    f1 = K.ln_fold(w1.to(DEV), gm.to(DEV), bt.to(DEV), b1.to(DEV))
    cl = torch.randint(0, 20, (M,), generator=gen, dtype=torch.int32).to(DEV)
    outs = []
    try:
        for knob in (1, 2, 3):
            lib.vh_set_tuning(2, knob)
            kc = torch.zeros(M, h, 24, 64, device=DEV)
            vc = torch.zeros_like(kc)
            q = torch.zeros(M, d, device=DEV)
            K.linear_qkv_folded(x, fq, q, kc, vc, M, 1, h, cache_len=cl)
            res = res0.to(DEV)
            outs.append((q, kc, vc, K.linear_folded(x, f1, act=1), K.linear(x, wo.to(DEV), bo.to(DEV), res, out=res),
                         K.linear(x, wh), K.linear(x, wq.to(DEV), ln=(gm.to(DEV), bt.to(DEV), None, None, 1e-5))))
    finally:
        lib.vh_set_tuning(2, 0)
    for u, v, t in zip(*outs):
        assert torch.equal(u, v) and torch.equal(u, t)
    ref = F.linear(F.layer_norm(x.cpu(), (d,), gm, bt, 1e-5), w1, b1)
    close(outs[0][3], F.gelu(ref), atol=5e-5)


# ---- residual stream in fp64 accumulator form (vh_linear_acc64 and its consumers) --------------------
@pytest.mark.parametrize('M,N,K_', [(32, 512, 2048), (5, 64, 512), (64, 128, 128), (17, 512, 4096), (32, 96, 1280)])
def test_linear_acc64_exact_atomics(K, M, N, K_):
    a = torch.randn(M, K_, generator=g(90))
    w = 0.05 * torch.randn(N, K_, generator=g(91))
    bias, res = torch.randn(N, generator=g(92)), torch.randn(M, N, generator=g(93))
    ref = a.double() @ w.double().T + bias.double() + res.double()
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    runs = []
    for _ in range(3):
        acc = torch.zeros(M, N, device=DEV, dtype=torch.float64)
        K.linear_acc64(ad, wd, acc, bd, rd)
        runs.append(acc)
    torch.testing.assert_close(runs[0].cpu(), ref, atol=3e-5, rtol=1e-6)
    # exact sums of grid-rounded addends: independent of the arrival order of the K slices
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    assert torch.equal(runs[0] * 2 ** 32, torch.round(runs[0] * 2 ** 32))      # multiples of 2^-32
    # accumulates onto what is there (no bias / residual the second time)
    K.linear_acc64(ad, wd, runs[0])
    torch.testing.assert_close(runs[0].cpu(), ref + a.double() @ w.double().T, atol=6e-5, rtol=1e-6)
    # integer data: exact
    ai = torch.randint(-3, 4, (M, K_), generator=g(94)).float()
    wi = torch.randint(-3, 4, (N, K_), generator=g(95)).float()
    acc = torch.zeros(M, N, device=DEV, dtype=torch.float64)
    K.linear_acc64(ai.to(DEV), wi.to(DEV), acc)
    assert torch.equal(acc.cpu(), (ai.double() @ wi.double().T))


@pytest.mark.parametrize('M,N,K_', [(32, 512, 512), (3, 64, 128), (64, 128, 256)])
def test_linear_x64_consumes_and_clears_the_residual(K, M, N, K_):
    a = torch.randn(M, K_, generator=g(96))
    w = 0.05 * torch.randn(N, K_, generator=g(97))
    bias = torch.randn(N, generator=g(98))
    res64 = torch.randn(M, N, generator=g(99), dtype=torch.float64)
    ref = (a.double() @ w.double().T + bias.double() + res64).float()
    rd = res64.to(DEV)
    out = K.linear_x64(a.to(DEV), w.to(DEV), bias.to(DEV), residual64=rd)
    close(out, ref, atol=3e-5)
    assert int(rd.count_nonzero()) == 0


@pytest.mark.parametrize('M,N,K_', [(32, 1025, 512), (4, 100, 128), (33, 48, 1024)])
def test_linear_x64_fp64_rows(K, M, N, K_):
    a64 = torch.randn(M, K_, generator=g(100), dtype=torch.float64)
    w = 0.05 * torch.randn(N, K_, generator=g(101))
    out = K.linear_x64(a64.to(DEV), w.to(DEV))
    close(out, (a64.float().double() @ w.double().T).float(), atol=3e-5)
    ai = torch.randint(-3, 4, (M, K_), generator=g(102)).double()
    wi = torch.randint(-3, 4, (N, K_), generator=g(103)).float()
    assert torch.equal(K.linear_x64(ai.to(DEV), wi.to(DEV)).cpu(), (ai @ wi.double().T).float())


def test_linear_x64_argument_checks(K):
    from valle2_amd._lib import VhError
    w = torch.randn(64, 128, device=DEV)
    a = torch.randn(4, 128, device=DEV)
    with pytest.raises(VhError, match='exactly one'):
        K.linear_x64(a, w)                                   # neither operand in fp64 form
    with pytest.raises(VhError, match='exactly one'):
        K.linear_x64(a.double(), w, residual64=torch.zeros(4, 64, device=DEV, dtype=torch.float64))
    with pytest.raises(VhError, match='decode path'):
        K.linear_x64(torch.randn(65, 128, device=DEV).double(), w)
    with pytest.raises(VhError, match='decode path'):
        K.linear_acc64(torch.randn(4, 128, device=DEV), torch.randn(60, 128, device=DEV),
                       torch.zeros(4, 60, device=DEV, dtype=torch.float64))


def test_qkv_folded_from_fp64_rows_matches_fp32_rows(K):
    B, h, S_max = 6, 2, 9
    d = 64 * h
    x = torch.randn(B, d, generator=g(104)) + 0.2
    w = 0.1 * torch.randn(3 * d, d, generator=g(105))
    gm, bt = 1 + 0.2 * torch.randn(d, generator=g(106)), 0.2 * torch.randn(d, generator=g(107))
    folded = K.ln_fold(w.to(DEV), gm.to(DEV), bt.to(DEV))
    cl = torch.tensor([3, 0, 8, 1, 5, 2], dtype=torch.int32, device=DEV)
    outs = []
    for xx in (x.to(DEV), x.double().to(DEV)):
        kc = torch.zeros(B, h, S_max, 64, device=DEV)
        vc = torch.zeros_like(kc)
        qo = torch.empty(B, d, device=DEV)
        K.linear_qkv_folded(xx, folded, qo, kc, vc, B, 1, h, cache_len=cl)
        outs.append((qo, kc, vc))
    for u, v in zip(*outs):
        assert torch.equal(u, v)       # fp32 → fp64 → fp32 is exact: identical arithmetic


@pytest.mark.parametrize('rows,d', [(1, 128), (37, 512), (5, 1024), (3, 2048), (9, 132)])
def test_layernorm(K, rows, d):
    x = 3 * torch.randn(rows, d, generator=g(15)) + 1
    gm, bt = torch.randn(d, generator=g(16)), torch.randn(d, generator=g(17))
    sc, sh = torch.randn(d, generator=g(18)), torch.randn(d, generator=g(19))
    close(K.layernorm(x.to(DEV), gm.to(DEV), bt.to(DEV)), F.layer_norm(x, (d,), gm, bt, 1e-5))
    close(K.layernorm(x.to(DEV), gm.to(DEV), bt.to(DEV), ada_scale=sc.to(DEV), ada_shift=sh.to(DEV)),
          sc * F.layer_norm(x, (d,), gm, bt, 1e-5) + sh)


def test_embed_sum_pe_bit_exact(K):
    from valle2_amd.synth import sinusoid_table
    d, B, T, Q = 128, 3, 11, 8
    tabs = [torch.randn(50, d, generator=g(20 + j)) for j in range(Q)]
    ids = torch.randint(0, 50, (B, T, Q), generator=g(30))
    pe = sinusoid_table(d, 64)
    out = torch.zeros(B, T + 4, d, device=DEV)
    K.embed_sum_pe(ids.to(DEV), [t.to(DEV) for t in tabs], pe.to(DEV), 2, out, out_t0=4)
    ref = F.embedding(ids[..., 0], tabs[0])
    for j in range(1, Q):
        ref = ref + F.embedding(ids[..., j], tabs[j])
    ref = ref + pe[2:2 + T, 0]
    assert torch.equal(out[:, 4:].cpu(), ref)          # pure adds in the same order: exact
    assert out[:, :4].abs().sum().item() == 0
    # single table, strided ids view (first codebook of a (T,Q) tensor), ragged lens
    out2 = torch.zeros(B, T, d, device=DEV)
    lens = torch.tensor([11, 4, 0], dtype=torch.int32)
    K.embed_sum_pe(ids.to(DEV)[..., 0], [tabs[0].to(DEV)], pe.to(DEV), 0, out2, lens=lens.to(DEV))
    ref2 = F.embedding(ids[..., 0], tabs[0]) + pe[:T, 0]
    for b in range(B):
        assert torch.equal(out2[b, :lens[b]].cpu(), ref2[b, :lens[b]])
        assert out2[b, lens[b]:].abs().sum().item() == 0


def _sdpa_ref(q, k, v, mask_bool_visible):
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask_bool_visible)


def _to_cache(k, S_max):
    B, h, S, hd = k.shape
    c = torch.zeros(B, h, S_max, hd)
    c[:, :, :S] = k
    return c


@pytest.mark.parametrize('B,h,T,mode', [(2, 2, 5, 'prefix'), (3, 2, 150, 'prefix'), (2, 8, 300, 'full'),
                                        (2, 2, 129, 'full'), (1, 1, 32, 'prefix'), (2, 2, 77, 'explicit')])
def test_attn_rows(K, B, h, T, mode):
    d = 64 * h
    q = torch.randn(B, T, d, generator=g(40))
    k = torch.randn(B, h, T, 64, generator=g(41))
    v = torch.randn(B, h, T, 64, generator=g(42))
    xl = T // 3
    kvl = torch.tensor([T - (5 * i) % (T // 2 + 1) for i in range(B)], dtype=torch.int32)
    from oracle.valle_oracle import build_attn_mask
    keypad = torch.arange(T)[None, :] >= kvl[:, None]                     # True = masked
    if mode == 'prefix':
        masked = build_attn_mask(xl, T - xl)[None] | keypad[:, None, :]
    elif mode == 'full':
        masked = keypad[:, None, :].expand(B, T, T)
    else:
        rnd = torch.rand(T, T, generator=g(43)) < 0.3
        rnd[:, 0] = False                                                 # no fully-masked row
        keypad[:, 0] = False
        masked = rnd[None] | keypad[:, None, :]
    qh = q.view(B, T, h, 64).permute(0, 2, 1, 3)
    ref = _sdpa_ref(qh, k, v, ~masked[:, None]).permute(0, 2, 1, 3).reshape(B, T, d)
    S_max = T + 9
    out = torch.empty(B * T, d, device=DEV)
    kw = {}
    if mode == 'prefix':
        kw = dict(mode=K.MASK_PREFIX, x_len=xl, kv_len=kvl.to(DEV))
    elif mode == 'full':
        kw = dict(mode=K.MASK_FULL, kv_len=kvl.to(DEV))
    else:
        kw = dict(mode=K.MASK_EXPLICIT, mask=rnd.to(torch.uint8).to(DEV),
                  pad=keypad.to(torch.uint8).to(DEV))
    K.attn_rows(q.view(B * T, d).to(DEV), _to_cache(k, S_max).to(DEV), _to_cache(v, S_max).to(DEV),
                out, B, h, T, T, **kw)
    close(out.view(B, T, d), ref, atol=3e-5)


def test_attn_rows_peaked_softmax(K):
    # sharply peaked scores (|q.k| up to 100s): exercises the online-softmax rescale branch where
    # the running max jumps by a lot at a late tile (cdna_hip_programming.md rule 26)
    B, h, T = 1, 2, 200
    k = torch.randn(B, h, T, 64, generator=g(44))
    v = torch.randn(B, h, T, 64, generator=g(45))
    q = torch.randn(B, T, h, 64, generator=g(46))
    k[:, :, 170] *= 40.0            # a spike key in the 6th tile
    k[:, :, 3] *= 15.0              # and a smaller one in the first
    out = torch.empty(B * T, 64 * h, device=DEV)
    K.attn_rows(q.reshape(B * T, -1).to(DEV), k.to(DEV), v.to(DEV), out, B, h, T, T,
                mode=K.MASK_FULL)
    ref = _sdpa_ref(q.permute(0, 2, 1, 3), k, v, None).permute(0, 2, 1, 3).reshape(B * T, -1)
    close(out, ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('B,h,S,n_split', [(4, 2, 37, 1), (32, 8, 300, 1), (2, 2, 1000, 4),
                                           (3, 16, 65, 2), (1, 1, 1, 1), (2, 2, 31, 3)])
def test_attn_decode(K, B, h, S, n_split):
    d = 64 * h
    S_max = S + 40
    q = torch.randn(B, d, generator=g(50))
    k = torch.randn(B, h, S_max, 64, generator=g(51))
    v = torch.randn(B, h, S_max, 64, generator=g(52))
    lens = torch.tensor([max(1, S - 3 * i) for i in range(B)], dtype=torch.int32)
    ref = torch.empty(B, d)
    for b in range(B):
        L = int(lens[b])
        r = _sdpa_ref(q[b].view(1, h, 1, 64), k[b:b + 1, :, :L], v[b:b + 1, :, :L], None)
        ref[b] = r.reshape(d)
    out = torch.empty(B, d, device=DEV)
    ws = K.attn_decode_ws(B, h, n_split, DEV)
    K.attn_decode(q.to(DEV), k.to(DEV), v.to(DEV), out, (lens - 1).to(DEV), 1, n_split, ws)
    close(out, ref, atol=3e-5)


@pytest.mark.parametrize('variant', [7, 4, 8, 9, 1])
@pytest.mark.parametrize('S_max', [40, 300, 1100])
def test_attn_decode_ring_kernels_short_and_ragged_rows(K, variant, S_max):
    """The ring kernels issue their first burst before the row's length is known and never predicate a load:
    rows shorter than a burst, lengths at chunk edges, and NaN / Inf garbage beyond every row's length (what a
    torch.empty cache may hold) must not leak into the result.  One (b, head) per CU (B x h = 256, n_split = 1),
    as in the decode step."""
    from valle2_amd import _lib
    B, h = 32, 8
    d = 64 * h
    gen = g(70 + S_max)
    q = torch.randn(B, d, generator=gen)
    k = torch.randn(B, h, S_max, 64, generator=gen)
    v = torch.randn(B, h, S_max, 64, generator=gen)
    edges = [1, 2, 5, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1024, 1087]
    lens = torch.tensor([min(S_max, edges[i % len(edges)] + (i // len(edges))) for i in range(B)], dtype=torch.int32)
    ref = torch.empty(B, d)
    for b in range(B):
        L = int(lens[b])
        ref[b] = _sdpa_ref(q[b].view(1, h, 1, 64), k[b:b + 1, :, :L], v[b:b + 1, :, :L], None).reshape(d)
        k[b, :, L:] = float('nan')                      # poison everything beyond the row's length
        v[b, :, L:] = float('inf')
    out = torch.full((B, d), float('nan'), device=DEV)
    lib = _lib.lib()
    lib.vh_set_tuning(0, variant)
    try:
        K.attn_decode(q.to(DEV), k.to(DEV), v.to(DEV), out, (lens - 1).to(DEV), 1, 1, None)
    finally:
        lib.vh_set_tuning(0, 0)
    assert bool(torch.isfinite(out).all()), 'garbage beyond a row\'s length leaked into the attention output'
    close(out, ref, atol=3e-5)


@pytest.mark.parametrize('S_max', [64, 320, 1120])
def test_pipelined_decode_kernels_in_stream_order(K, S_max):
    """vh_linear_qkv_folded_pipe -> vh_attn_decode_pipe -> vh_linear_ll_in on ONE stream (every wait finds its pairs
    already there) against vh_linear_qkv_folded -> vh_attn_decode (8-wave ring) -> vh_linear: q and the K / V rows bit for
    bit, the attention output and the out-projection to rounding (the same chunk order per wave, but the compiler
    contracts the two kernels' multiply-adds differently: atol 2e-6 / 1e-5); ragged lengths at
    chunk edges with NaN / Inf beyond every row's length; a stale tag (wrong layer) times out into the error word."""
    from valle2_amd import _lib
    lib = _lib.lib()
    B, h, d = 32, 8, 512
    gen = g(90 + S_max)
    x = torch.randn(B, d, generator=gen).to(DEV)
    wqkv = (0.05 * torch.randn(3 * d, d, generator=gen)).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=gen)).to(DEV), (0.1 * torch.randn(d, generator=gen)).to(DEV)
    wo, bo = (0.05 * torch.randn(d, d, generator=gen)).to(DEV), (0.1 * torch.randn(d, generator=gen)).to(DEV)
    folded = K.ln_fold(wqkv, gamma, beta)
    edges = [1, 2, 5, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1024, 1087]
    lens = torch.tensor([min(S_max, edges[i % len(edges)] + (i // len(edges))) for i in range(B)], dtype=torch.int32)
    lens[0] = min(S_max, 300)                                  # cache_len[0] carries the step of the tags
    kc = torch.randn(B, h, S_max, 64, generator=gen)
    vc = torch.randn(B, h, S_max, 64, generator=gen)
    for b in range(B):
        kc[b, :, int(lens[b]) - 1:] = float('nan')             # the newest row (len - 1) comes from the QKV launch
        vc[b, :, int(lens[b]) - 1:] = float('inf')
    cl = (lens - 1).to(DEV)
    s = K.stream()
    # reference chain: plain kernels
    kc0, vc0 = kc.to(DEV), vc.to(DEV)
    q0 = torch.empty(B, d, device=DEV)
    K.linear_qkv_folded(x, folded, q0, kc0, vc0, B, 1, h, cache_len=cl)
    a0 = torch.empty(B, d, device=DEV)
    lib.vh_set_tuning(0, 7)
    try:
        K.attn_decode(q0, kc0, vc0, a0, cl, 1, 1, None)
    finally:
        lib.vh_set_tuning(0, 0)
    y0 = K.linear(a0, wo, bias=bo, residual=x)
    # pipelined kernels, in stream order
    kc1, vc1 = kc.to(DEV), vc.to(DEV)
    qkv_ll = torch.zeros(3, B, d, 2, device=DEV)
    attn_ll = torch.zeros(B, d, 2, device=DEV)
    err = torch.zeros(16 + 512, device=DEV, dtype=torch.int32)
    y1 = torch.full((B, d), float('nan'), device=DEV)
    layer = 3
    K.check(lib.vh_linear_qkv_folded_pipe(x.data_ptr(), d, folded[0].data_ptr(), folded[1].data_ptr(), folded[2].data_ptr(),
                                          kc1.data_ptr(), vc1.data_ptr(), cl.data_ptr(), B, d, h, S_max, 1e-5,
                                          qkv_ll.data_ptr(), layer, s), 'vh_linear_qkv_folded_pipe')
    K.check(lib.vh_attn_decode_pipe(qkv_ll.data_ptr(), kc1.data_ptr(), vc1.data_ptr(), attn_ll.data_ptr(), cl.data_ptr(),
                                    B, h, S_max, layer, err.data_ptr(), s), 'vh_attn_decode_pipe')
    K.check(lib.vh_linear_ll_in(attn_ll.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), d, y1.data_ptr(), d,
                                B, d, d, cl.data_ptr(), layer, err.data_ptr(), s), 'vh_linear_ll_in')
    torch.cuda.synchronize()
    assert int(err[0]) == 0
    bits = lambda t: t.view(torch.int32)                       # NaN-proof equality
    assert torch.equal(bits(kc1), bits(kc0)) and torch.equal(bits(vc1), bits(vc0)), 'K / V rows of the new position'
    assert torch.equal(qkv_ll[0, :, :, 0], q0), 'published q'
    tag = ((int(cl[0]) + 1) * 64 + layer) * 8 + 5
    assert bool((qkv_ll[..., 1].view(torch.int32) == tag).all())
    assert bool(torch.isfinite(attn_ll[..., 0]).all()), 'garbage beyond a row\'s length leaked into the attention output'
    close(attn_ll[..., 0], a0.cpu(), atol=2e-6, rtol=0)
    assert bool((attn_ll[..., 1].view(torch.int32) == tag + 1).all())
    close(y1, y0.cpu(), atol=1e-5, rtol=0)
    # argument errors come back as codes, nothing is launched: a cache block that is not whole 32-key chunks, a layer
    # the tag has no room for, an out-projection of another width
    assert lib.vh_attn_decode_pipe(qkv_ll.data_ptr(), kc1.data_ptr(), vc1.data_ptr(), attn_ll.data_ptr(), cl.data_ptr(),
                                   B, h, S_max - 1, layer, err.data_ptr(), s) != 0
    assert lib.vh_linear_qkv_folded_pipe(x.data_ptr(), d, folded[0].data_ptr(), folded[1].data_ptr(), folded[2].data_ptr(),
                                         kc1.data_ptr(), vc1.data_ptr(), cl.data_ptr(), B, d, h, S_max, 1e-5,
                                         qkv_ll.data_ptr(), 64, s) != 0
    assert lib.vh_linear_ll_in(attn_ll.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), d, y1.data_ptr(), d,
                               B, d, 256, cl.data_ptr(), layer, err.data_ptr(), s) != 0
    # a consumer whose pairs never come (tags of another layer) gives up: error word set, no hang
    K.check(lib.vh_linear_ll_in(attn_ll.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), d, y1.data_ptr(), d,
                                B, d, d, cl.data_ptr(), layer + 1, err.data_ptr(), s), 'vh_linear_ll_in')
    torch.cuda.synchronize()
    assert (int(err[0]) & 0xffffffff) >> 28 == 0x8


def test_linear_qkv_scatter(K):
    B, T, h = 3, 5, 2
    d = 64 * h
    S_max = 12
    a = torch.randn(B * T, d, generator=g(60))
    w = 0.1 * torch.randn(3 * d, d, generator=g(61))
    ref = F.linear(a, w)
    for cache_len in (None, torch.tensor([2, 0, 7], dtype=torch.int32)):
        kc = torch.zeros(B, h, S_max, 64, device=DEV)
        vc = torch.zeros_like(kc)
        qo = torch.empty(B * T, d, device=DEV)
        K.linear_qkv(a.to(DEV), w.to(DEV), qo, kc, vc, B, T, h,
                     cache_len=None if cache_len is None else cache_len.to(DEV))
        close(qo, ref[:, :d])
        kref = ref[:, d:2 * d].view(B, T, h, 64).permute(0, 2, 1, 3)
        vref = ref[:, 2 * d:].view(B, T, h, 64).permute(0, 2, 1, 3)
        for b in range(B):
            p0 = 0 if cache_len is None else int(cache_len[b])
            close(kc[b, :, p0:p0 + T], kref[b])
            close(vc[b, :, p0:p0 + T], vref[b])
            assert kc[b, :, :p0].abs().sum().item() == 0 and kc[b, :, p0 + T:].abs().sum().item() == 0
    # decode shape (T=1, fused LN) through the skinny kernel
    x = torch.randn(B, d, generator=g(62))
    gm, bt = torch.randn(d, generator=g(63)), torch.randn(d, generator=g(64))
    ref1 = F.linear(F.layer_norm(x, (d,), gm, bt, 1e-5), w)
    kc = torch.zeros(B, h, S_max, 64, device=DEV)
    vc = torch.zeros_like(kc)
    qo = torch.empty(B, d, device=DEV)
    cl = torch.tensor([3, 11, 0], dtype=torch.int32)
    K.linear_qkv(x.to(DEV), w.to(DEV), qo, kc, vc, B, 1, h, cache_len=cl.to(DEV),
                 ln=(gm.to(DEV), bt.to(DEV), None, None, 1e-5))
    close(qo, ref1[:, :d], atol=5e-5)
    for b in range(B):
        close(kc[b, :, int(cl[b])], ref1[b, d:2 * d].view(h, 64), atol=5e-5)
        close(vc[b, :, int(cl[b])], ref1[b, 2 * d:].view(h, 64), atol=5e-5)


def test_greedy_step(K):
    B, V, d, eos = 5, 1025, 128, 1024
    logits = torch.randn(B, 1028, generator=g(70))
    logits[1, 7] = logits[1, 900] = 50.0          # tie → lowest index
    logits[2, eos] = 60.0                          # emits EOS
    logits[:, V:] = 1e9                            # padding columns must be ignored
    emb = torch.randn(V + 1, d, generator=g(71))
    from valle2_amd.synth import sinusoid_table
    pe = sinusoid_table(d, 64)
    codes = torch.zeros(B, 20, dtype=torch.int64)
    codes[:, :4] = torch.randint(0, 1024, (B, 4), generator=g(72))
    codes[3, 3] = eos                              # already finished row stays EOS
    apos = torch.full((B,), 4, dtype=torch.int32)
    clen = torch.tensor([10, 11, 12, 13, 14], dtype=torch.int32)
    cnt = torch.zeros(20, dtype=torch.int32)
    dv = {k: v.to(DEV) for k, v in dict(logits=logits, emb=emb, pe=pe, codes=codes, apos=apos,
                                        clen=clen, cnt=cnt).items()}
    x = torch.empty(B, d, device=DEV)
    K.greedy_step(dv['logits'], V, eos, dv['codes'], dv['cnt'], dv['emb'], dv['pe'], dv['apos'],
                  dv['clen'], x)
    exp = torch.argmax(logits[:, :V], dim=-1)
    exp[1] = 7
    exp[3] = eos
    assert dv['codes'][:, 4].cpu().tolist() == exp.tolist()
    assert dv['apos'].cpu().tolist() == [5] * B and dv['clen'].cpu().tolist() == [11, 12, 13, 14, 15]
    assert dv['cnt'].cpu()[4].item() == 2 and dv['cnt'].cpu().sum().item() == 2
    assert torch.equal(x.cpu(), emb[exp] + pe[4, 0])
    # the same step writing the fp64 accumulator form of the embedding
    dv2 = {k: v.to(DEV) for k, v in dict(codes=codes, apos=apos, clen=clen, cnt=cnt).items()}
    x64 = torch.empty(B, d, device=DEV, dtype=torch.float64)
    K.greedy_step(dv['logits'], V, eos, dv2['codes'], dv2['cnt'], dv['emb'], dv['pe'], dv2['apos'],
                  dv2['clen'], x64)
    assert torch.equal(x64.cpu(), (emb[exp] + pe[4, 0]).double())
    assert torch.equal(dv2['codes'], dv['codes'])


@pytest.mark.parametrize('M', [1, 8, 20, 32, 64])
def test_two_slab_residual_stream_primitives(K, M):
    """vh_linear_to_x2 / vh_linear_x2 / vh_linear_qkv_folded(a_form=2): linear_2 as two K slices kept apart, the
    consumers adding the slabs on load.  Against torch fp32 on the CPU; slab sums reproducible bit for bit."""
    g = torch.Generator().manual_seed(M)
    d, dff, h = 512, 2048, 8   # noqa: E741
    hid = torch.randn(M, dff, generator=g)
    w2, b2 = 0.05 * torch.randn(d, dff, generator=g), 0.1 * torch.randn(d, generator=g)
    xmid = torch.randn(M, d, generator=g)
    slabs = torch.empty(2, M, d, device=DEV)
    K.linear_to_x2(hid.to(DEV), w2.to(DEV), slabs, bias=b2.to(DEV), residual=xmid.to(DEV))
    x_ref = F.linear(hid, w2, b2) + xmid
    close(slabs.sum(0), x_ref, atol=5e-5)
    close(slabs[1], F.linear(hid[:, dff // 2:], w2[:, dff // 2:]), atol=5e-5)      # slice 1: raw partial
    again = torch.empty_like(slabs)
    K.linear_to_x2(hid.to(DEV), w2.to(DEV), again, bias=b2.to(DEV), residual=xmid.to(DEV))
    assert torch.equal(again, slabs)
    x = slabs.sum(0)                                                  # what every consumer should see
    # out-projection with the residual in two-slab form
    attn, wo, bo = torch.randn(M, d, generator=g), 0.05 * torch.randn(d, d, generator=g), 0.1 * torch.randn(d, generator=g)
    o = K.linear_x2(attn.to(DEV), wo.to(DEV), bias=bo.to(DEV), residual=slabs)
    close(o, F.linear(attn, wo, bo) + x.cpu(), atol=5e-5)
    assert torch.equal(o, K.linear(attn.to(DEV), wo.to(DEV), bo.to(DEV), residual=x.contiguous()))
    # head on two-slab rows (ragged N = 1025)
    wh = 0.05 * torch.randn(1025, d, generator=g)
    lg = K.linear_x2(slabs, wh.to(DEV))
    close(lg, F.linear(x.cpu(), wh), atol=5e-5)
    assert torch.equal(lg, K.linear(x.contiguous(), wh.to(DEV)))
    # folded LayerNorm + QKV on two-slab rows == the same kernel on the summed rows
    wq = 0.05 * torch.randn(3 * d, d, generator=g)
    gm, bt = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    folded = K.ln_fold(wq.to(DEV), gm.to(DEV), bt.to(DEV))
    outs = []
    for a in (slabs, x.contiguous()):
        q = torch.empty(M, d, device=DEV)
        kc, vc = torch.zeros(M, h, 4, 64, device=DEV), torch.zeros(M, h, 4, 64, device=DEV)
        cl = torch.full((M,), 2, device=DEV, dtype=torch.int32)
        K.linear_qkv_folded(a, folded, q, kc, vc, M, 1, h, cache_len=cl)
        outs.append((q, kc, vc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = F.linear(F.layer_norm(x.cpu(), (d,), gm, bt, 1e-5), wq)
    close(outs[0][0], ref[:, :d], atol=1e-4)
    close(outs[0][1][:, :, 2].reshape(M, d), ref[:, d:2 * d], atol=1e-4)
