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
from valle2_amd._lib import h16_dtype
H16 = h16_dtype()      # the library's 16-bit operand format (fp16 by default, bf16 with -DVH_PERF_BF16)

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


@pytest.mark.parametrize('M,N,K_', [(9000, 512, 2048), (8800, 512, 512), (4300, 1024, 1024)])
def test_linear_ws_tail_split(K, M, N, K_):
    """vh_linear_ws with more than 256 tiles and a short tail (284 / 276 / 272 tiles): the tile kernel's tail split behind
    the inference entry point (a prompt pass of 8 x 1100 positions); integer data exact, in-place residual, replays equal."""
    from valle2_amd import _lib
    assert _lib.lib().vh_linear_ws_bytes(M, N, K_) > 0
    a = torch.randint(-3, 4, (M, K_), generator=g(190)).float()
    w = torch.randint(-3, 4, (N, K_), generator=g(191)).float()
    assert torch.equal(K.linear_ws(a.to(DEV), w.to(DEV)).cpu(), a @ w.T)
    a = torch.randn(M, K_, generator=g(192))
    w = torch.randn(N, K_, generator=g(193)) / K_ ** 0.5
    bias, res = torch.randn(N, generator=g(194)), torch.randn(M, N, generator=g(195))
    resd = res.to(DEV)
    out = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV), resd, out=resd, act=1)      # residual aliases out
    close(out, F.gelu(F.linear(a, w, bias)) + res, atol=5e-5)
    o1 = K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV))
    assert torch.equal(o1, K.linear_ws(a.to(DEV), w.to(DEV), bias.to(DEV)))
    ref = K.linear(a.to(DEV), w.to(DEV), bias.to(DEV))                               # the unsplit launch
    whole = (M + 127) // 128 * (N // 128) // 256 * 256 // (N // 128) * 128
    assert torch.equal(o1[:whole], ref[:whole])
    close(o1[whole:], ref[whole:].cpu(), atol=5e-5)


@pytest.mark.parametrize('seed', range(24))
def test_linear_dispatch_sweep(K, seed):
    """Seeded random shapes across every dispatch boundary of the GEMM entry points (skinny <= 64 rows / tile kernel,
    split-K of few tiles, the tail split beyond a multiple of 256 tiles, ragged N, K a multiple of 16 but not of 32 ->
    register-staged tile kernel): vh_linear, vh_linear_ws and vh_linear_ex with random epilogues against torch; integer
    operands first (sums exact in fp32, so any dropped or doubled K slice shows), then real data."""
    import random
    from valle2_amd import _lib
    rnd = random.Random(1000 + seed)
    M = rnd.choice([1, 5, 16, 33, 64, 65, 127, 300, 1000, 2049, 4300, 8800, 9001, 10240, 12000])
    N = rnd.choice([64, 128, 512, 1024, 1025, 1536, 2048]) if M < 9000 else rnd.choice([512, 1024])
    K_ = rnd.choice([128, 256, 512, 1024, 2048, 48, 1008])
    use_bias, use_res, act = rnd.random() < 0.7, rnd.random() < 0.6, rnd.choice([0, 1])
    a = torch.randint(-2, 3, (M, K_), generator=g(seed)).float()
    w = torch.randint(-2, 3, (N, K_), generator=g(seed + 50)).float()
    ad, wd = a.to(DEV), w.to(DEV)
    ref = a @ w.T
    assert torch.equal(K.linear(ad, wd).cpu(), ref)
    assert torch.equal(K.linear_ws(ad, wd).cpu(), ref)
    assert torch.equal(K.linear_ex(ad, wd).cpu(), ref)
    a = torch.randn(M, K_, generator=g(seed + 100))
    w = torch.randn(N, K_, generator=g(seed + 150)) / K_ ** 0.5
    bias = torch.randn(N, generator=g(seed + 200)) if use_bias else None
    res = torch.randn(M, N, generator=g(seed + 250)) if use_res else None
    ref = F.linear(a, w, bias)
    ref = (F.gelu(ref) if act else ref) + (res if use_res else 0)
    dv = lambda t: None if t is None else t.to(DEV)
    resd = None
    if use_res:                                             # rows 16-byte aligned (the boundary's contract): pad N = 1025
        resd = torch.zeros(M, (N + 3) // 4 * 4, device=DEV)[:, :N]
        resd.copy_(res)
    for fn in (K.linear, K.linear_ws, K.linear_ex):
        out = fn(a.to(DEV), w.to(DEV), bias=dv(bias), residual=resd, act=act)
        close(out, ref, atol=6e-5)


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
    """16 < M <= 64: one workgroup per (16 rows, 16 columns) (default) against one per 16 columns: the same sums in the
    same order, bit for bit — with the row statistics of the folded LayerNorm taken the same way in both (from row
    loads: knob 9 = 1).  The default takes them from the operand fragments where a workgroup holds one row tile (a
    one-pass formula about the row's first element): equal to rounding, checked at the end."""
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
        for knob, stats in ((1, 1), (2, 1), (3, 1), (1, 0)):
            lib.vh_set_tuning(2, knob)
            lib.vh_set_tuning(9, stats)
            kc = torch.zeros(M, h, 24, 64, device=DEV)
            vc = torch.zeros_like(kc)
            q = torch.zeros(M, d, device=DEV)
            K.linear_qkv_folded(x, fq, q, kc, vc, M, 1, h, cache_len=cl)
            res = res0.to(DEV)
            outs.append((q, kc, vc, K.linear_folded(x, f1, act=1), K.linear(x, wo.to(DEV), bo.to(DEV), res, out=res),
                         K.linear(x, wh), K.linear(x, wq.to(DEV), ln=(gm.to(DEV), bt.to(DEV), None, None, 1e-5))))
    finally:
        lib.vh_set_tuning(2, 0)
        lib.vh_set_tuning(9, 0)
    for u, v, t, frag in zip(*outs):
        assert torch.equal(u, v) and torch.equal(u, t)
        torch.testing.assert_close(frag, u, atol=2e-5, rtol=1e-5)          # statistics from the fragments: to rounding
    ref = F.linear(F.layer_norm(x.cpu(), (d,), gm, bt, 1e-5), w1, b1)
    close(outs[0][3], F.gelu(ref), atol=5e-5)


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


@pytest.mark.parametrize('B,h,n_split', [(4, 8, 8), (8, 16, 2), (3, 2, 16), (1, 8, 5)])
def test_attn_decode_split_combine_in_launch_equals_the_second_launch(K, B, h, n_split):
    """Key splits combined by the last workgroup to arrive (VH_TUNE_DECODE_COMBINE = 1; the default at two splits since round 6)
    against the separate combine launch (2; the default at more splits): the records are added in split order either way, so
    the results are the same BITS; the ticket words re-arm themselves — 25 launches on one workspace — and end at zero."""
    from valle2_amd import _lib
    d, S = 64 * h, 700
    gen = g(90 + n_split)
    q = torch.randn(B, d, generator=gen).to(DEV)
    k = torch.randn(B, h, S, 64, generator=gen).to(DEV)
    v = torch.randn(B, h, S, 64, generator=gen).to(DEV)
    lens = torch.tensor([max(1, S - 37 * i) for i in range(B)], dtype=torch.int32).to(DEV)
    ws = K.attn_decode_ws(B, h, n_split, DEV)
    lib = _lib.lib()
    outs = {}
    try:
        for knob in (2, 1, 0):
            lib.vh_set_tuning(12, knob)
            out = torch.empty(B, d, device=DEV)
            for _ in range(25 if knob == 1 else 1):
                out.fill_(float('nan'))
                K.attn_decode(q, k, v, out, lens - 1, 1, n_split, ws)
            outs[knob] = out.clone()
    finally:
        lib.vh_set_tuning(12, 0)
    assert torch.equal(outs[2], outs[1]) and torch.equal(outs[2], outs[0])
    tickets = ws.view(torch.int32)[B * h * n_split * 72:][: B * h]
    assert int(tickets.abs().sum()) == 0
    ref = torch.empty(B, d)
    for b in range(B):
        L = int(lens[b])
        ref[b] = _sdpa_ref(q[b].cpu().view(1, h, 1, 64), k[b:b + 1, :, :L].cpu(), v[b:b + 1, :, :L].cpu(), None).reshape(d)
    close(outs[0], ref, atol=3e-5)


@pytest.mark.parametrize('variant', [0, 1])        # the ring kernel (default at 256 (row, head) pairs) and the burst kernel
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


def _decode_ref64(q, k, v, lens, h):
    """Double-precision one-query attention per (row, head) over the row's first lens[b] keys."""
    B, d = q.shape
    S = k.shape[2]
    qd = q.double().view(B, h, 1, 64)
    s = (qd @ k.double().transpose(-1, -2)) / 8.0                          # (B, h, 1, S)
    s = s.masked_fill(torch.arange(S)[None, None, None, :] >= lens.long()[:, None, None, None], float('-inf'))
    return (torch.softmax(s, dim=-1) @ v.double()).reshape(B, d).float()


@pytest.mark.parametrize('S', [2048, 2907, 4999])
@pytest.mark.parametrize('form', ['ring', 'burst', 'split2', 'split4', 'split16', 'kv16'])
def test_attn_decode_long_context(K, form, S):
    """Round 5: the decode-attention kernels over the 1.5 k - 5 k contexts configs[4] decodes at
    (PositionalEncoding max_len 5000, modules.py:56): ring and burst kernels at 256 (row, head) pairs, key splits 2 / 4 /
    16 + combine at fewer pairs, and the bf16-cache variant — against a double-precision reference, ragged lengths that
    end inside / at the edge of 32-key chunks and of the split ranges, NaN / Inf beyond every row's length."""
    from valle2_amd import _lib
    n_split = int(form[5:]) if form.startswith('split') else 1
    B, h = (16, 16) if n_split == 1 else {2: (8, 16), 4: (8, 8), 16: (2, 8)}[n_split]
    d = 64 * h
    S_max = (S + 31) // 32 * 32
    gen = g(300 + S + n_split)
    q = torch.randn(B, d, generator=gen)
    k = torch.randn(B, h, S_max, 64, generator=gen)
    v = torch.randn(B, h, S_max, 64, generator=gen)
    if form == 'kv16':
        k, v = k.to(H16), v.to(H16)
    tails = [0, 1, 31, 32, 33, 255, 256, 1000, S // 2, S - 1537, 63, 64, 65, 511, 512, 513]
    lens = torch.tensor([max(1, S - tails[i % len(tails)]) for i in range(B)], dtype=torch.int32)
    ref = _decode_ref64(q, k.float(), v.float(), lens, h)
    for b in range(B):
        k[b, :, int(lens[b]):] = float('nan')
        v[b, :, int(lens[b]):] = float('inf')
    out = torch.full((B, d), float('nan'), device=DEV)
    lib = _lib.lib()
    if form == 'kv16':
        K.attn_decode_kv16(q.to(DEV), k.to(DEV), v.to(DEV), out, (lens - 1).to(DEV), 1)
    else:
        lib.vh_set_tuning(0, 1 if form == 'burst' else 0)
        try:
            ws = K.attn_decode_ws(B, h, n_split, DEV) if n_split > 1 else None
            K.attn_decode(q.to(DEV), k.to(DEV), v.to(DEV), out, (lens - 1).to(DEV), 1, n_split, ws)
        finally:
            lib.vh_set_tuning(0, 0)
    assert bool(torch.isfinite(out).all()), 'garbage beyond a row\'s length leaked into the attention output'
    close(out, ref, atol=2e-5)


@pytest.mark.parametrize('B,h,prefix_len,n_split', [(32, 8, 1024, 1), (4, 8, 1024, 8), (8, 16, 626, 2), (40, 2, 31, 1),
                                                    (64, 4, 1, 1), (3, 2, 2651, 3), (33, 8, 100, 1), (8, 16, 2907, 2)])
def test_attn_decode_shared_prompt(K, B, h, prefix_len, n_split):
    """vh_attn_decode_shared: B beams over ONE shared prompt (read once for all beams: a beam is a lane of the score tile) +
    each beam's own rows, against double-precision attention over the concatenated keys; ragged suffix lengths (one beam
    with a single own row), NaN / Inf in the prefix cache beyond prefix_len and in the suffix caches beyond every beam's
    length, beams beyond 32 (second pass of the prefix kernel), key splits of the suffix with the combine launch."""
    d = 64 * h
    S_suf = 96
    prefix_S = (prefix_len + 31) // 32 * 32 + 32
    gen = g(500 + prefix_len + B)
    q = torch.randn(B, d, generator=gen)
    kp = torch.randn(1, h, prefix_S, 64, generator=gen)
    vp = torch.randn(1, h, prefix_S, 64, generator=gen)
    ks = torch.randn(B, h, S_suf, 64, generator=gen)
    vs = torch.randn(B, h, S_suf, 64, generator=gen)
    slen = torch.tensor([(7 * i) % 90 for i in range(B)], dtype=torch.int32)          # rows in the suffix BEFORE the new one
    ref = torch.empty(B, d)
    for b in range(B):
        n = int(slen[b]) + 1
        kk = torch.cat([kp[0, :, :prefix_len], ks[b, :, :n]], dim=1).double()
        vv = torch.cat([vp[0, :, :prefix_len], vs[b, :, :n]], dim=1).double()
        s = (q[b].double().view(h, 1, 64) @ kk.transpose(-1, -2)) / 8.0
        ref[b] = (torch.softmax(s, dim=-1) @ vv).reshape(d).float()
        ks[b, :, n:] = float('nan')
        vs[b, :, n:] = float('inf')
    kp[:, :, prefix_len:] = float('nan')
    vp[:, :, prefix_len:] = float('inf')
    out = torch.full((B, d), float('nan'), device=DEV)
    K.attn_decode_shared(q.to(DEV), kp.to(DEV), vp.to(DEV), prefix_len, ks.to(DEV), vs.to(DEV), out, slen.to(DEV), 1,
                         n_split=n_split)
    assert bool(torch.isfinite(out).all()), 'garbage beyond a length leaked into the attention output'
    close(out, ref, atol=2e-5)
    # deterministic: records are merged in chunk order
    out2 = torch.empty_like(out)
    K.attn_decode_shared(q.to(DEV), kp.to(DEV), vp.to(DEV), prefix_len, ks.to(DEV), vs.to(DEV), out2, slen.to(DEV), 1,
                         n_split=n_split)
    assert torch.equal(out, out2)


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


@pytest.mark.parametrize('B,d,V', [(1, 512, 1025), (5, 128, 1025), (16, 512, 1025), (17, 256, 1025), (24, 512, 1025), (32, 512, 1025),
                                   (32, 1024, 1025), (40, 512, 1025), (64, 512, 1025), (7, 128, 37), (32, 512, 2048)])
def test_head_greedy_one_launch_equals_head_then_greedy_step(K, B, d, V):
    """vh_head_greedy (the head product with the greedy step in the same launch, valle_ar.py:158-171) against vh_linear +
    vh_greedy_step on the same operands: logits, tokens, counters, positions and the next input rows bit for bit — over three
    consecutive steps in place (x_next = x, as the decoder runs it; the arrival counters must come back to zero), with tied
    logits (duplicate head rows: the lowest column wins), a row that emits EOS and a row that had already finished."""
    eos = V - 1
    gen = g(900 + B + d + V)
    w = 0.05 * torch.randn(V, d, generator=gen)
    if V > 900:
        w[900] = w[7]                               # columns 7 and 900 tie in every row
    emb = torch.randn(V + 1, d, generator=gen)
    from valle2_amd.synth import sinusoid_table
    pe = sinusoid_table(d, 64)[:, 0].contiguous()
    x0 = torch.randn(B, d, generator=gen)
    if B > 2:
        x0[2] = 40.0 * w[eos] / w[eos].norm()       # row 2 emits EOS at the first step
    codes = torch.zeros(B, 20, dtype=torch.int64)
    codes[:, :4] = torch.randint(0, V - 1, (B, 4), generator=gen)
    if B > 3:
        codes[3, 3] = eos                           # an already finished row keeps EOS
    ldl = (V + 3) // 4 * 4

    def state():
        return dict(x=x0.clone().to(DEV), logits=torch.zeros(B, ldl, device=DEV), codes=codes.clone().to(DEV),
                    cnt=torch.zeros(21, dtype=torch.int32, device=DEV), apos=torch.full((B,), 4, dtype=torch.int32, device=DEV),
                    clen=(10 + torch.arange(B, dtype=torch.int32)).to(DEV))
    wd, embd, ped = w.to(DEV), emb.to(DEV), pe.to(DEV)
    a, b = state(), state()
    ws = K.head_greedy_ws(B, V, DEV)
    for step in range(3):
        K.linear(a['x'], wd, None, out=a['logits'][:, :V])
        K.greedy_step(a['logits'], V, eos, a['codes'], a['cnt'], embd, ped, a['apos'], a['clen'], a['x'])
        K.head_greedy(b['x'], wd, b['logits'], V, eos, b['codes'], b['cnt'], embd, ped, b['apos'], b['clen'], b['x'], ws)
        for k in a:
            assert torch.equal(a[k][:, :V] if k == 'logits' else a[k], b[k][:, :V] if k == 'logits' else b[k]), (step, k)
        assert int(ws[:64].abs().sum()) == 0, 'arrival counters not reset'
    if V > 900:
        assert not (b['codes'][:, 4:7] == 900).any()
    if B > 3:
        assert b['codes'][2, 4:7].tolist() == [eos] * 3 and b['codes'][3, 4:7].tolist() == [eos] * 3


def _ffn_ref(x, gm, bt, w1, b1, w2, b2):
    return x + F.linear(F.gelu(F.linear(F.layer_norm(x, (x.shape[1],), gm, bt, 1e-5), w1, b1)), w2, b2)


@pytest.mark.parametrize('M,d,dff', [(32, 512, 2048), (1, 512, 2048), (5, 512, 2048), (20, 512, 2048), (64, 512, 2048),
                                     (4, 128, 512), (16, 128, 512), (8, 1024, 4096), (33, 256, 1024), (9, 512, 80),
                                     (32, 512, 2064)])
def test_ffn_decode_matches_torch(K, M, d, dff):
    """vh_ffn_decode (FeedForward + residual of a decode step as one launch split over dim_feedforward + the slab
    reduce, modules.py:215-221,278-279) against torch fp32 on the CPU: atol 1e-4 on O(1) rows; in place (out = x);
    repeated launches bit-identical (slabs summed in slice order, no atomics); every slice width / row grouping."""
    from valle2_amd import _lib
    lib = _lib.lib()
    gen = g(300 + M + d + dff)
    x = torch.randn(M, d, generator=gen) + 0.2
    gm, bt = 1 + 0.1 * torch.randn(d, generator=gen), 0.1 * torch.randn(d, generator=gen)
    w1, b1 = 0.05 * torch.randn(dff, d, generator=gen), 0.1 * torch.randn(dff, generator=gen)
    w2, b2 = 0.05 * torch.randn(d, dff, generator=gen), 0.1 * torch.randn(d, generator=gen)
    ref = _ffn_ref(x, gm, bt, w1, b1, w2, b2)
    folded = K.ln_fold(w1.to(DEV), gm.to(DEV), bt.to(DEV), b1.to(DEV))
    w2d, b2d = w2.to(DEV), b2.to(DEV)
    out = K.ffn_decode(x.to(DEV), folded, w2d, b2d)
    close(out, ref, atol=1e-4)
    assert torch.equal(out, K.ffn_decode(x.to(DEV), folded, w2d, b2d)), 'not reproducible'
    xi = x.to(DEV)
    K.ffn_decode(xi, folded, w2d, b2d, out=xi)                       # in place on the residual stream
    assert torch.equal(xi, out)
    close(K.ffn_decode(x.to(DEV), folded, w2d, None), ref - b2, atol=1e-4)
    # the unfused composition of the same step (folded linear_1 + GELU, split-K linear_2 + residual)
    hid = K.linear_folded(x.to(DEV), folded, act=1)
    two = K.linear_ws(hid, w2d, b2d, residual=x.to(DEV))
    close(out, two.cpu(), atol=5e-5)
    try:
        for sw in (16, 32):
            for rows in (8, 16):
                if dff % sw:
                    continue
                lib.vh_set_tuning(7, sw)
                lib.vh_set_tuning(8, rows)
                o = K.ffn_decode(x.to(DEV), folded, w2d, b2d)
                close(o, ref, atol=1e-4)
    finally:
        lib.vh_set_tuning(7, 0)
        lib.vh_set_tuning(8, 0)


def test_ffn_decode_integer_exact_and_strided(K):
    """Integer-valued operands come out exactly whatever the slice / wave / part summation order (MFMA layout check);
    x and out as column blocks of wider buffers (row strides > d)."""
    M, d, dff = 19, 256, 512
    gen = g(411)
    # LayerNorm folded with gamma = 1, beta = 0 on rows whose mean is 0 and variance is 1 exactly: x = +-1 balanced
    x = torch.ones(M, d)
    x[:, ::2] = -1
    x = x[:, torch.randperm(d, generator=gen)]
    w1 = torch.randint(-2, 3, (dff, d), generator=gen).float()
    b1 = torch.randint(-2, 3, (dff,), generator=gen).float()
    w2 = torch.randint(-2, 3, (d, dff), generator=gen).float()
    b2 = torch.randint(-2, 3, (d,), generator=gen).float()
    folded = K.ln_fold(w1.to(DEV), torch.ones(d, device=DEV), torch.zeros(d, device=DEV), b1.to(DEV))
    big_in = torch.full((M, d + 64), 7.0, device=DEV)
    big_in[:, 32:32 + d] = x.to(DEV)
    big_out = torch.full((M, d + 128), -3.0, device=DEV)
    K.ffn_decode(big_in[:, 32:32 + d], folded, w2.to(DEV), b2.to(DEV), out=big_out[:, 64:64 + d])
    rs = 1.0 / (1.0 + 1e-5) ** 0.5                                   # rstd of a unit-variance row
    hid = F.gelu((x @ w1.T) * rs + b1)
    ref = x + hid @ w2.T + b2
    close(big_out[:, 64:64 + d], ref, atol=2e-3, rtol=1e-5)           # hid is O(100): fp32 rounding of the GELU products
    assert bool((big_out[:, :64] == -3.0).all()) and bool((big_out[:, 64 + d:] == -3.0).all())
    # exactness of the second product: a hidden tile of small integers survives phase 2 bit for bit
    w1z = torch.zeros(dff, d)
    b1i = torch.randint(0, 4, (dff,), generator=gen).float() * 8.0   # gelu(8k) == 8k in fp32 for k >= 1, gelu(0) = 0
    fz = K.ln_fold(w1z.to(DEV), torch.ones(d, device=DEV), torch.zeros(d, device=DEV), b1i.to(DEV))
    out = K.ffn_decode(x.to(DEV), fz, w2.to(DEV), b2.to(DEV))
    assert torch.equal(out.cpu(), x + (F.gelu(b1i)[None, :] @ w2.T) + b2)


def test_ffn_decode_argument_checks(K):
    from valle2_amd._lib import VhError
    d, dff = 512, 2048
    w1 = torch.randn(dff, d, device=DEV)
    folded = K.ln_fold(w1, torch.ones(d, device=DEV), torch.zeros(d, device=DEV))
    w2 = torch.randn(d, dff, device=DEV)
    with pytest.raises(VhError, match='unsupported shape|M=65'):
        K.ffn_decode(torch.randn(65, d, device=DEV), folded, w2)
    with pytest.raises(VhError, match='w2'):
        K.ffn_decode(torch.randn(4, d, device=DEV), folded, w2[:, :1024].contiguous())
    with pytest.raises(VhError, match='workspace'):
        K.ffn_decode(torch.randn(4, d, device=DEV), folded, w2, workspace=torch.empty(16, device=DEV))
    w1s = torch.randn(64, 192, device=DEV)                           # d_model outside the kernel's set
    fs = K.ln_fold(w1s, torch.ones(192, device=DEV), torch.zeros(192, device=DEV))
    with pytest.raises(VhError, match='d_model'):
        K.ffn_decode(torch.randn(4, 192, device=DEV), fs, torch.randn(192, 64, device=DEV),
                     workspace=torch.empty(1 << 16, device=DEV))




# ---- perf mode: bf16 K/V cache (opt-in) ---------------------------------------------------------
@pytest.mark.parametrize('S_max', [64, 320, 1120])
def test_attn_decode_kv16_matches_fp32_math_on_the_rounded_cache(K, S_max):
    """vh_attn_decode_kv16 against fp32 attention over the SAME bf16-rounded K/V (the kernel's arithmetic is fp32: what it
    may differ by is summation order), ragged lengths at chunk edges, NaN / Inf beyond every row's length."""
    B, h = 32, 8
    d = 64 * h
    gen = g(170 + S_max)
    q = torch.randn(B, d, generator=gen)
    k = torch.randn(B, h, S_max, 64, generator=gen).to(H16)
    v = torch.randn(B, h, S_max, 64, generator=gen).to(H16)
    edges = [1, 2, 5, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1024, 1087]
    lens = torch.tensor([min(S_max, edges[i % len(edges)] + (i // len(edges))) for i in range(B)], dtype=torch.int32)
    ref = torch.empty(B, d)
    for b in range(B):
        L = int(lens[b])
        ref[b] = _sdpa_ref(q[b].view(1, h, 1, 64), k[b:b + 1, :, :L].float(), v[b:b + 1, :, :L].float(), None).reshape(d)
        k[b, :, L:] = float('nan')
        v[b, :, L:] = float('inf')
    out = torch.full((B, d), float('nan'), device=DEV)
    K.attn_decode_kv16(q.to(DEV), k.to(DEV), v.to(DEV), out, (lens - 1).to(DEV), 1)
    assert bool(torch.isfinite(out).all()), 'garbage beyond a row\'s length leaked into the attention output'
    close(out, ref, atol=3e-5)


def test_kv_narrowing_and_bf16_append_round_to_nearest_even(K):
    """vh_kv_to_bf16 == torch's .to(H16) bit for bit on the valid rows; vh_linear_qkv_folded_kv16 appends exactly the
    bf16 rounding of what vh_linear_qkv_folded appends in fp32, and leaves q untouched."""
    from valle2_amd import engine
    L, B, h, S0, S1 = 2, 5, 2, 37, 64
    cache = engine.KVCache(L, B, h, S0, DEV)
    cache.buf.copy_(torch.randn(cache.buf.shape, generator=g(191)) * 3)
    narrow = cache.narrowed(S1)
    assert narrow.buf.dtype == H16 and tuple(narrow.buf.shape) == (L, 2, B, h, S1, 64)
    assert torch.equal(narrow.buf[..., :S0, :].view(torch.int16), cache.buf.to(H16).view(torch.int16))
    d = 64 * h
    gen = g(192)
    x = (torch.randn(B, d, generator=gen) + 0.2).to(DEV)
    w = (0.1 * torch.randn(3 * d, d, generator=gen)).to(DEV)
    folded = K.ln_fold(w, (1 + 0.2 * torch.randn(d, generator=gen)).to(DEV), (0.2 * torch.randn(d, generator=gen)).to(DEV))
    cl = torch.tensor([3, 0, 36, 10, 63], dtype=torch.int32, device=DEV)
    k32, v32 = torch.zeros(B, h, S1, 64, device=DEV), torch.zeros(B, h, S1, 64, device=DEV)
    q32 = torch.empty(B, d, device=DEV)
    K.linear_qkv_folded(x, folded, q32, k32, v32, B, 1, h, cache_len=cl)
    k16 = torch.zeros(B, h, S1, 64, device=DEV, dtype=H16)
    v16 = torch.zeros_like(k16)
    q16 = torch.empty(B, d, device=DEV)
    K.linear_qkv_folded_kv16(x, folded, q16, k16, v16, h, cl)
    assert torch.equal(q16, q32)
    assert torch.equal(k16.view(torch.int16), k32.to(H16).view(torch.int16))
    assert torch.equal(v16.view(torch.int16), v32.to(H16).view(torch.int16))
