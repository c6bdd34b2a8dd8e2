"""The training backward's matrix products are hand-written kernels only (no library GEMM, no engine
switch): `vh_gemm_tn` (dW = dY^T X, token-major operands read in place, deterministic split-K),
`vh_linear_ex` (dX = dY W on the NT LDS-DMA tile kernel with W^T from `vh_transpose`; GELU forward with the
pre-activation kept; GELU backward fused into the product) — checked against torch fp32 on the CPU — and
the second, materialised derivation of the attention backward (vh_gemm_batched) against the same gradient
tests as the flash kernels."""
import pytest
import torch
import torch.nn.functional as F

from tests import test_train_gpu as T

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture
def materialized():
    from valle2_amd import autograd as A
    old = A.ATTENTION_BACKWARD
    A.ATTENTION_BACKWARD = 'materialized'
    yield
    A.ATTENTION_BACKWARD = old


def test_ar_gradients_materialized_attention(materialized):
    T.test_ar_training_step_gradients_vs_oracle_and_reference()


@pytest.mark.parametrize('stage', [2, 7])
def test_nar_gradients_materialized_attention(materialized, stage):
    T.test_nar_training_step_gradients_vs_oracle(stage)


@pytest.mark.parametrize('mode', ['prefix', 'full'])
def test_attention_backward_materialized(materialized, mode):
    T.test_qkv_attention_backward(mode)


def test_no_library_gemm_in_the_backward():
    """The autograd module has no engine switch and never calls a torch matrix product."""
    import ast
    import inspect
    from valle2_amd import autograd as A
    assert not hasattr(A, 'BACKWARD_GEMM')
    tree = ast.parse(inspect.getsource(A))
    for node in ast.walk(tree):
        assert not (isinstance(node, ast.BinOp) and isinstance(node.op, ast.MatMult)), 'a @ product in autograd.py'
        if isinstance(node, ast.Attribute):
            assert node.attr not in ('matmul', 'mm', 'bmm', 'addmm', 'baddbmm', 'einsum'), node.attr


@pytest.mark.parametrize('M,NI,NJ', [(1, 128, 128), (31, 64, 200), (32, 128, 128), (257, 1025, 512), (1000, 512, 2048),
                                     (4096 + 17, 1536, 512), (16000, 512, 512), (300, 130, 36)])
def test_gemm_tn_matches_torch_and_is_deterministic(M, NI, NJ):
    from valle2_amd import kernels as K
    g = torch.Generator().manual_seed(M + NI)
    lda, ldb = (NI + 3) // 4 * 4 + 8, (NJ + 3) // 4 * 4
    a = torch.randn(M, lda, generator=g)
    b = torch.randn(M, ldb, generator=g)
    ad, bd = a.to(DEV)[:, :NI], b.to(DEV)[:, :NJ]
    out = K.gemm_tn(ad, bd)
    ref = (a[:, :NI].double().T @ b[:, :NJ].double()).float()
    tol = 2e-6 * (M ** 0.5) * 4 + 1e-5
    torch.testing.assert_close(out.cpu(), ref, atol=tol * 3, rtol=1e-5)
    out2 = K.gemm_tn(ad, bd)
    assert torch.equal(out, out2), 'slab sums must be bitwise reproducible'
    # integer-valued operands: every product and partial sum is exact in fp32 → exact result (layout check)
    ai = torch.randint(-3, 4, (M, lda), generator=g).float()
    bi = torch.randint(-3, 4, (M, ldb), generator=g).float()
    outi = K.gemm_tn(ai.to(DEV)[:, :NI], bi.to(DEV)[:, :NJ])
    assert torch.equal(outi.cpu(), ai[:, :NI].T @ bi[:, :NJ])


def test_transpose_zero_pads():
    from valle2_amd import kernels as K
    w = torch.randn(1025, 512)
    t = K.transpose(w.to(DEV))
    assert t.shape == (512, 1056)
    assert torch.equal(t[:, :1025].cpu(), w.T) and float(t[:, 1025:].abs().max()) == 0.0
    t2 = K.transpose(torch.randn(70, 33).to(DEV), ldo=96)
    assert t2.shape == (33, 96) and float(t2[:, 70:].abs().max()) == 0.0


@pytest.mark.parametrize('M', [5, 64, 129, 1000])
def test_linear_ex_training_epilogues(M):
    from valle2_amd import kernels as K
    g = torch.Generator().manual_seed(M)
    d, dff = 128, 512
    x = torch.randn(M, d, generator=g)
    w1, b1 = 0.1 * torch.randn(dff, d, generator=g), 0.1 * torch.randn(dff, generator=g)
    pre = torch.empty(M, dff, device=DEV)
    hid = K.linear_ex(x.to(DEV), w1.to(DEV), bias=b1.to(DEV), pre_out=pre, act=K.ACT_GELU,
                      out=torch.empty(M, dff, device=DEV))
    ref_pre = F.linear(x, w1, b1)
    torch.testing.assert_close(pre.cpu(), ref_pre, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(hid.cpu(), F.gelu(ref_pre), atol=2e-5, rtol=1e-5)
    # backward through the activation fused into dX = dY . W2:  (dy @ w2) * gelu'(pre)
    w2 = 0.1 * torch.randn(d, dff, generator=g)
    dy = torch.randn(M, d, generator=g)
    wt = K.transpose(w2.to(DEV))                                     # (dff, 128)
    dpre = K.linear_ex(dy.to(DEV), wt, residual=pre, act=K.ACT_GELU_BWD, out=torch.empty(M, dff, device=DEV))
    p = ref_pre.clone().requires_grad_()
    F.gelu(p).backward(dy @ w2)
    torch.testing.assert_close(dpre.cpu(), p.grad, atol=3e-5, rtol=1e-5)
    # the pair the training step uses: the forward keeps gelu'(pre) instead of pre, the backward multiplies by it
    dact = torch.empty(M, dff, device=DEV)
    hid2 = K.linear_ex(x.to(DEV), w1.to(DEV), bias=b1.to(DEV), pre_out=dact, act=K.ACT_GELU_D,
                       out=torch.empty(M, dff, device=DEV))
    torch.testing.assert_close(hid2.cpu(), F.gelu(ref_pre), atol=2e-5, rtol=1e-5)
    pp = ref_pre.clone().requires_grad_()
    F.gelu(pp).backward(torch.ones_like(pp))
    torch.testing.assert_close(dact.cpu(), pp.grad, atol=2e-5, rtol=1e-5)
    dpre2 = K.linear_ex(dy.to(DEV), wt, residual=dact, act=K.ACT_MUL, out=torch.empty(M, dff, device=DEV))
    torch.testing.assert_close(dpre2.cpu(), p.grad, atol=3e-5, rtol=1e-5)
    # ragged head: K = 1025 zero-padded to 1056 on both operands
    wp = 0.05 * torch.randn(1025, d, generator=g)
    dl = torch.zeros(M, 1056)
    dl[:, :1025] = torch.randn(M, 1025, generator=g)
    dx = K.linear_ex(dl.to(DEV)[:, :1025], K.transpose(wp.to(DEV)), K=1056, out=torch.empty(M, d, device=DEV))
    torch.testing.assert_close(dx.cpu(), dl[:, :1025] @ wp, atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize('M,N,Kd', [(10240, 512, 512), (10240 - 37, 512, 1536), (2300, 2048, 512), (8192 + 5, 512, 256)])
def test_linear_ex_tail_split(M, N, Kd):
    """vh_linear_ex with a workspace: the tiles beyond the last multiple of 256 run as K slices + the fix-up launch
    (320 / 320 / 288 / 260 tiles here; the last shape splits 2 ways only: K / split >= 128).  Every epilogue of the
    fix-up against torch, and against the unsplit kernel (knob 10 = 1) on the same inputs."""
    from valle2_amd import _lib, kernels as K
    assert _lib.lib().vh_linear_ex_ws_bytes(M, N, Kd) > 0
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, Kd, generator=g).to(DEV)
    w = (0.05 * torch.randn(N, Kd, generator=g)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    ref_pre = torch.addmm(b, a, w.t())
    outs = {}
    for knob in (0, 1):
        _lib.lib().vh_set_tuning(10, knob)
        try:
            pre, cs = torch.empty(M, N, device=DEV), torch.zeros(N, device=DEV)
            plain = K.linear_ex(a, w, bias=b, residual=res, out=torch.empty(M, N, device=DEV), colsum=cs)
            act = K.linear_ex(a, w, bias=b, pre_out=pre, act=K.ACT_GELU, out=torch.empty(M, N, device=DEV))
            bwd = K.linear_ex(a, w, residual=ref_pre, act=K.ACT_GELU_BWD, out=torch.empty(M, N, device=DEV))
            dact = torch.empty(M, N, device=DEV)
            act_d = K.linear_ex(a, w, bias=b, pre_out=dact, act=K.ACT_GELU_D, out=torch.empty(M, N, device=DEV))
            mul = K.linear_ex(a, w, residual=res, act=K.ACT_MUL, out=torch.empty(M, N, device=DEV))
        finally:
            _lib.lib().vh_set_tuning(10, 0)
        outs[knob] = (plain, cs, pre, act, bwd, act_d, dact, mul)
    plain, cs, pre, act, bwd, act_d, dact, mul = outs[0]
    tol = dict(atol=5e-5, rtol=2e-5)
    torch.testing.assert_close(plain, ref_pre + res, **tol)
    torch.testing.assert_close(cs, (ref_pre + res).sum(0), atol=2e-2, rtol=1e-4)
    torch.testing.assert_close(pre, ref_pre, **tol)
    torch.testing.assert_close(act, F.gelu(ref_pre), **tol)
    p = ref_pre.clone().requires_grad_()
    F.gelu(p).backward(a @ w.t())
    torch.testing.assert_close(bwd, p.grad, **tol)
    torch.testing.assert_close(act_d, F.gelu(ref_pre), **tol)
    pg = ref_pre.clone().requires_grad_()
    F.gelu(pg).backward(torch.ones_like(pg))
    torch.testing.assert_close(dact, pg.grad, **tol)
    torch.testing.assert_close(mul, (a @ w.t()) * res, atol=2e-4, rtol=2e-5)
    n_whole = ((M + 127) // 128 * (N // 128)) // 256 * 256          # whole tiles: the same kernel path, bit for bit
    rows_whole = n_whole // (N // 128) * 128
    for x, y in zip(outs[0], outs[1]):
        if x.dim() == 2:
            assert torch.equal(x[:rows_whole], y[:rows_whole])
            torch.testing.assert_close(x[rows_whole:], y[rows_whole:], **tol)


def test_ffn_function_matches_the_unfused_composition():
    """FfnFn (fused GELU forward / backward) against LinearFn + GeluFn + LinearFn on the same inputs."""
    from valle2_amd import autograd as A
    g = torch.Generator().manual_seed(3)
    M, d, dff = 300, 128, 512
    mk = lambda *s: (0.2 * torch.randn(*s, generator=g)).to(DEV)
    base = [mk(M, d), mk(dff, d), mk(dff), mk(d, dff), mk(d), mk(M, d)]
    dy = mk(M, d)
    outs = []
    for fused in (True, False):
        xn, w1, b1, w2, b2, res = [t.clone().requires_grad_() for t in base]
        if fused:
            y = A.FfnFn.apply(xn, w1, b1, w2, b2, res)
        else:
            y = A.linear(A.GeluFn.apply(A.linear(xn, w1, b1)), w2, b2, residual=res)
        y.backward(dy)
        outs.append([y.detach()] + [t.grad for t in (xn, w1, b1, w2, b2, res)])
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, atol=2e-5, rtol=1e-5)
