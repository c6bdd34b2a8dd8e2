"""vh_gemm_batched (the backward-pass GEMM): all four operand layouts, ragged sizes, split-K with
atomics, batched strided views read in place.  Integer operands must come out exactly (fp32 MFMA is
an exact FMA chain and integer partial sums are order-independent)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def ints(shape, seed):
    return torch.randint(-3, 4, shape, generator=torch.Generator().manual_seed(seed)).float()


def padded(t):
    """device copy whose row stride is a multiple of 4 floats, as the ABI requires"""
    r, c = t.shape
    buf = torch.zeros(r, (c + 3) // 4 * 4)
    buf[:, :c] = t
    return buf.to(DEV)[:, :c]


@pytest.mark.parametrize('M,N,K', [(5, 7, 16), (128, 128, 32), (300, 1025, 131), (257, 64, 1000),
                                   (1, 1024, 512), (200, 130, 5000)])
@pytest.mark.parametrize('ak,bk', [(False, False), (False, True), (True, False), (True, True)])
def test_layouts_integer_exact(M, N, K, ak, bk):
    from valle2_amd import kernels as Kx
    a, b = ints((M, K), 1), ints((N, K), 2)
    b[:, 0] += torch.arange(N).float() % 5
    ref = a @ b.T
    a_st = padded(a.T.contiguous()) if ak else padded(a)
    b_st = padded(b.T.contiguous()) if bk else padded(b)
    out = torch.full((M, (N + 3) // 4 * 4), 7.0, device=DEV)[:, :N]
    Kx.gemm(a_st, b_st, out, a_kmajor=ak, b_kmajor=bk)          # (200,130,5000) takes the split-K path
    assert torch.equal(out.cpu(), ref)


def test_batched_views_in_place():
    from valle2_amd import kernels as Kx
    B, T, h = 2, 37, 3
    d = 64 * h
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * T, d, generator=g)
    k = torch.randn(B, h, T, 64, generator=g)
    qh = q.view(B, T, h, 64).permute(0, 2, 1, 3)
    tp = (T + 3) // 4 * 4
    S = torch.empty(B, h, T, tp, device=DEV)[..., :T]
    Kx.gemm(q.to(DEV).view(B, T, h, 64).permute(0, 2, 1, 3), k.to(DEV), S)       # S = Q K^T per (b,h)
    torch.testing.assert_close(S.cpu(), qh @ k.transpose(-1, -2), atol=2e-5, rtol=2e-5)
    # dV = P^T dO written straight into a (B*T, 3d) gradient buffer
    P = torch.randn(B, h, T, T, generator=g)
    do = torch.randn(B * T, d, generator=g)
    dqkv = torch.zeros(B * T, 3 * d, device=DEV)
    dv = dqkv.view(B, T, 3, h, 64)[:, :, 2].permute(0, 2, 1, 3)
    Pd = torch.zeros(B, h, T, tp, device=DEV)
    Pd[..., :T] = P.to(DEV)
    Kx.gemm(Pd[..., :T], do.to(DEV).view(B, T, h, 64).permute(0, 2, 1, 3), dv, a_kmajor=True, b_kmajor=True)
    ref = (P.transpose(-1, -2) @ do.view(B, T, h, 64).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B * T, d)
    torch.testing.assert_close(dqkv[:, 2 * d:].cpu(), ref, atol=3e-5, rtol=3e-5)
    assert float(dqkv[:, :2 * d].abs().sum()) == 0.0
