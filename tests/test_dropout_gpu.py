"""GPU parity of train-mode dropout (valle/models/modules.py:56-58,219,277-278; default p = 0.1, valle/config.py:26).

The HIP path never stores a mask: every dropout is a counter-based field (include/valle_hip.h `vh_dropout_spec`)
regenerated inside the GEMM epilogues, the LayerNorm backward and the embedding kernels.  Parity is proven by EXPORTING
the fields a training step used (`vh_dropout_mask`; checked bit for bit against oracle/philox.py) and running the CPU
oracle — whose dropout placement is pinned to the real reference by tests/golden/*_dropout.npz — with those very fields.

Tolerances as the dropout-free training tests: loss rtol 1e-5, every parameter's gradient within 1e-3 of its norm."""
import numpy as np
import pytest
import torch

from tests.golden import cases as C

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def g(seed):
    return torch.Generator().manual_seed(seed)


def _spec(rec):
    from valle2_amd import dropout
    return dropout.spec(rec['seed'], rec['site'], rec['p'])


def _field(rec, check=True):
    """(rows, cols) bool keep field of a recorded site, from the device — and equal to the numpy restatement."""
    from oracle import philox as P
    from valle2_amd import dropout
    keep = dropout.mask(_spec(rec), rec['rows'], rec['cols'], DEV).cpu()
    if check:
        want = P.keep_field(rec['seed'], rec['site'], rec['p'], min(rec['rows'], 257), rec['cols'])
        assert np.array_equal(keep[: want.shape[0]].numpy(), want), rec['name']
    return keep != 0


def _oracle_masks(recs, b, tx, check=True):
    """Recorded fields -> {oracle site name: keep tensor in the layout of the tensor it multiplies}."""
    out = {}
    for r in recs:
        f = _field(r, check)
        if r['name'] == 'tokens_position_emb.dropout':         # a field over the joint (B, tx + t, d) buffer
            f = f.view(b, -1, r['cols'])[:, :tx]
        elif r['name'] == 'audio_position_emb.dropout':
            f = f.view(b, -1, r['cols'])[:, tx:]
        out[r['name']] = f
    return out


def _train_step(model, batch, seed, **kw):
    from valle2_amd import dropout
    dropout.RECORD = []
    try:
        torch.manual_seed(seed)
        loss = model.training_step({k: v.clone() for k, v in batch.items()}, **kw)
        loss.backward()
        return loss, dropout.RECORD
    finally:
        dropout.RECORD = None


def _grad_check(model, ref_params, names, tol=1e-3):
    for n in names:
        got = dict(model.named_parameters())[n].grad
        assert got is not None, f'no gradient reached {n}'
        ref = ref_params[n].grad
        err = (got.cpu() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        assert err < tol, f'{n}: relative gradient error {err:.2e}'


# ---- kernels ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rows,cols,p', [(1, 4, 0.5), (37, 128, 0.1), (300, 2048, 0.1), (129, 512, 0.9)])
def test_field_matches_numpy_philox_and_free_standing_dropout(rows, cols, p):
    from oracle import philox as P
    from valle2_amd import dropout
    sp = dropout.spec(0x1234567890ABCDEF, (3 << 48) | (5 << 8) | 4, p)
    keep = dropout.mask(sp, rows, cols, DEV).cpu().numpy()
    assert np.array_equal(keep, P.keep_field(sp.seed, sp.site, p, rows, cols))
    x = torch.randn(rows, cols, generator=g(1))
    y = dropout.apply_raw(x.to(DEV), sp).cpu()
    want = x * torch.from_numpy(keep).float() * np.float32(1.0 / (1.0 - np.float64(np.float32(p))))
    torch.testing.assert_close(y, want, rtol=1e-6, atol=0)
    assert torch.equal(y != 0, (torch.from_numpy(keep) != 0) & (x != 0))


def test_dropout_module_forward_backward_and_eval_identity():
    from valle2_amd import dropout
    m = torch.nn.Dropout(0.3)
    x = torch.randn(6, 5, 64, generator=g(2)).to(DEV).requires_grad_()
    dropout.RECORD = None
    y = dropout.apply(m.train(), x)
    kept = (y != 0)
    assert 0.55 < kept.float().mean().item() < 0.85
    torch.testing.assert_close(y[kept], (x / 0.7)[kept].detach(), rtol=1e-6, atol=0)
    y.backward(torch.ones_like(y))
    torch.testing.assert_close(x.grad, kept.float() / 0.7, rtol=1e-6, atol=0)     # the same field backward
    assert dropout.apply(m.eval(), x) is x
    assert float(dropout.apply(torch.nn.Dropout(1.0).train(), x).detach().abs().sum()) == 0.0
    y2 = dropout.apply(m.train(), x)                                              # a new call draws a new field
    assert not torch.equal(y2 != 0, kept)


@pytest.mark.parametrize('M,N,K_,act', [(300, 512, 512, 'none'), (640, 2048, 512, 'gelu_d'), (10240, 512, 2048, 'none'),
                                        (96, 128, 64, 'gelu'), (16 * 641, 512, 512, 'none')])
def test_linear_epilogue_dropout(M, N, K_, act):
    """out = dropout(act(a W^T + b)) + residual in the tile kernel's epilogue (interior tiles, edge tiles and the tail
    split's fix-up launch: 10240 x 512 is 320 tiles = 256 whole + 64 split) against torch with the exported field."""
    from valle2_amd import dropout, kernels as K
    a = torch.randn(M, K_, generator=g(3)).to(DEV)
    w = (0.05 * torch.randn(N, K_, generator=g(4))).to(DEV)
    b = torch.randn(N, generator=g(5)).to(DEV)
    res = torch.randn(M, N, generator=g(6)).to(DEV)
    sp = dropout.spec(99, dropout.site(dropout.ATTN_RES, 3), 0.1)
    f = dropout.mask(sp, M, N, DEV).float() / 0.9
    pre = a.double() @ w.double().T + b.double()
    out = torch.empty(M, N, device=DEV)
    if act == 'gelu_d':
        aux = torch.empty(M, N, device=DEV)
        K.linear_ex(a, w, bias=b, out=out, pre_out=aux, act=K.ACT_GELU_D, drop=sp)
        pd = pre.clone().requires_grad_()
        gl = torch.nn.functional.gelu(pd)
        gl.sum().backward()
        torch.testing.assert_close(out, (gl.detach() * f).float(), atol=2e-4, rtol=1e-4)
        torch.testing.assert_close(aux, (pd.grad * f).float(), atol=2e-4, rtol=1e-4)
        return
    acts = {'none': (K.ACT_NONE, lambda t: t), 'gelu': (K.ACT_GELU, torch.nn.functional.gelu)}
    code, fn = acts[act]
    K.linear_ex(a, w, bias=b, residual=res, out=out, act=code, drop=sp)
    torch.testing.assert_close(out, (fn(pre) * f + res).float(), atol=3e-4, rtol=1e-4)
    plain = K.linear_ex(a, w, bias=b, residual=res, act=code)                     # and the spec-free call is untouched
    torch.testing.assert_close(plain, (fn(pre) + res).float(), atol=3e-4, rtol=1e-4)


@pytest.mark.parametrize('rows,d,ada', [(300, 512, False), (65, 128, True), (9, 1024, True)])
def test_layernorm_backward_emits_the_dropped_branch_gradient(rows, d, ada):
    from valle2_amd import autograd as A, dropout
    x = torch.randn(rows, d, generator=g(7)).to(DEV)
    gm, bt = (1 + 0.1 * torch.randn(d, generator=g(8))).to(DEV), (0.1 * torch.randn(d, generator=g(9))).to(DEV)
    s = (1 + 0.2 * torch.randn(d, generator=g(10))).to(DEV) if ada else None
    dy, dres = torch.randn(rows, d, generator=g(11)).to(DEV), torch.randn(rows, d, generator=g(12)).to(DEV)
    sp = dropout.spec(5, dropout.site(dropout.FFN_RES, 1), 0.1)
    gmp, btp = torch.nn.Parameter(gm), torch.nn.Parameter(bt)
    dst = torch.zeros(2, d, device=DEV)
    col0, col1 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dx0, _, _, none = A._ln_bwd(x, gmp, btp, s, dy, dres, col0, 1e-5, dst)
    dx1, _, _, dxd = A._ln_bwd(x, gmp, btp, s, dy, dres, col1, 1e-5, torch.zeros(2, d, device=DEV), drop=sp)
    assert none is None and torch.equal(dx0, dx1)
    f = dropout.mask(sp, rows, d, DEV).float() / 0.9
    torch.testing.assert_close(dxd, dx1 * f, rtol=1e-6, atol=0)
    torch.testing.assert_close(col1, dxd.sum(0), rtol=1e-4, atol=1e-4)            # the column sums are of the dropped copy
    torch.testing.assert_close(col0, dx0.sum(0), rtol=1e-4, atol=1e-4)


def test_embedding_dropout_forward_backward():
    """Position dropout inside the gather (modules.py:80) and regenerated by the scatter: two parts of one buffer with
    different p and sites, against torch with the exported fields."""
    import torch.nn.functional as F
    from valle2_amd import autograd as A, dropout
    from valle2_amd.synth import sinusoid_table
    d, B, tx, t = 128, 3, 11, 26
    tok_tab = torch.randn(50, d, generator=g(80)).requires_grad_()
    tab = torch.randn(30, d, generator=g(81)).requires_grad_()
    tokens, codes = torch.randint(0, 50, (B, tx), generator=g(90)), torch.randint(0, 30, (B, t), generator=g(91))
    pe = sinusoid_table(d, 64)
    s1, s2 = dropout.spec(7, dropout.site(dropout.PE_TEXT), 0.1), dropout.spec(7, dropout.site(dropout.PE_AUDIO), 0.25)
    f1 = dropout.mask(s1, B * (tx + t), d, DEV).cpu().view(B, tx + t, d).float() / 0.9
    f2 = dropout.mask(s2, B * (tx + t), d, DEV).cpu().view(B, tx + t, d).float() / 0.75
    ref = torch.cat([(F.embedding(tokens, tok_tab) + pe[:tx, 0]) * f1[:, :tx],
                     (F.embedding(codes, tab) + pe[:t, 0]) * f2[:, tx:]], dim=1)
    dy = torch.randn(B, tx + t, d, generator=g(92))
    ref.backward(dy)
    dt = [x.detach().to(DEV).requires_grad_() for x in (tok_tab, tab)]
    out = A.EmbedConcatFn.apply([(tokens.to(DEV), pe.to(DEV), 0, [0], s1), (codes.to(DEV), pe.to(DEV), 0, [1], s2)], *dt)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    out.backward(dy.to(DEV))
    torch.testing.assert_close(dt[0].grad.cpu(), tok_tab.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dt[1].grad.cpu(), tab.grad, rtol=1e-4, atol=2e-5)


# ---- models: loss and every gradient against the oracle fed the SAME fields ---------------------------------------------
def _build_train(cls_name, kw, sd):
    from tests.test_models_gpu import build
    return build(cls_name, kw, sd).train()


def test_ar_training_step_with_dropout_vs_oracle_with_the_same_fields():
    from oracle import valle_oracle as O
    _, sd, batch = C.ar_train_inputs()
    kw = C.AR_TINY_DROPOUT
    cfg = C.cfg_of(kw)
    model = _build_train('ValleAR', kw, sd)
    loss, recs = _train_step(model, batch, seed=11)
    assert len(recs) == 2 + 3 * cfg.num_layers
    b, tx = batch['tokens'].shape[0], int(max(batch['tokens_lens']))
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref = O.ar_training_loss(params, cfg, batch, O.Dropout(cfg.dropout, 0.1, _oracle_masks(recs, b, tx)))
    ref.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    _grad_check(model, params, sorted(k for k in params if not k.endswith('.pe')))
    # and the dropout-free oracle is somewhere else entirely
    assert abs(float(O.ar_training_loss(sd, cfg, batch)) - float(ref)) > 1e-3


@pytest.mark.parametrize('stage', [1, 5])
def test_nar_training_step_with_dropout_vs_oracle_with_the_same_fields(stage):
    from oracle import valle_oracle as O
    kw, sd, batch = C.nar_inputs()
    kw = dict(kw, dropout=0.1)
    cfg = C.cfg_of(kw)
    model = _build_train('ValleNAR', kw, sd)
    loss, recs = _train_step(model, batch, seed=12, stage=stage)
    assert len(recs) == 2 + 3 * cfg.num_layers
    b, tx = batch['tokens'].shape[0], int(batch['tokens_lens'].max())
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref = O.nar_training_loss(params, cfg, batch, stage, O.Dropout(cfg.dropout, 0.1, _oracle_masks(recs, b, tx)))
    ref.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    used = sorted(k for k, v in params.items() if v.grad is not None and v.grad.abs().sum() > 0)
    _grad_check(model, params, used)


def test_ar_training_step_with_dropout_at_configs3_size():
    """configs[3] (12L/512d, 16 ragged utterances) in train mode at the reference's default p = 0.1: loss and every
    parameter's gradient against the oracle run with the exported fields (tail-split GEMM shapes, 16 k rows)."""
    from oracle import valle_oracle as O
    kw, sd, batch = C.ar_train_full_inputs()
    kw = dict(kw, dropout=0.1)
    cfg = C.cfg_of(kw)
    model = _build_train('ValleAR', kw, sd)
    loss, recs = _train_step(model, batch, seed=13)
    b, tx = batch['tokens'].shape[0], int(max(batch['tokens_lens']))
    masks = _oracle_masks(recs, b, tx, check=False)
    _field(recs[3])                                              # (one site checked against numpy in full-size runs)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref = O.ar_training_loss(params, cfg, batch, O.Dropout(cfg.dropout, 0.1, masks))
    ref.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    _grad_check(model, params, sorted(k for k in params if not k.endswith('.pe')))


# ---- the fields of a training step: rates, independence, repeatability --------------------------------------------------
def test_fields_of_a_step_keep_rate_and_independence():
    _, sd, batch = C.ar_train_inputs()
    model = _build_train('ValleAR', C.AR_TINY_DROPOUT, sd)
    _, recs_a = _train_step(model, batch, seed=21)
    model.zero_grad()
    _, recs_b = _train_step(model, batch, seed=22)               # another step: another seed draw
    p = 0.1
    fields = {}
    for tag, recs in (('a', recs_a), ('b', recs_b)):
        for r in recs:
            f = _field(r, check=False).double()
            n = f.numel()
            assert abs(f.mean().item() - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5, r['name']
            fields[(tag, r['name'])] = f
    both = (1 - p) ** 2
    keys = sorted(fields)
    for i, ka in enumerate(keys):
        for kb in keys[i + 1:]:
            fa, fb = fields[ka], fields[kb]
            if fa.shape != fb.shape:
                continue
            n = fa.numel()
            assert abs((fa * fb).mean().item() - both) < 4.5 * (both * (1 - both) / n) ** 0.5, (ka, kb)
    assert len({(r['seed'], r['site']) for r in recs_a + recs_b}) == len(recs_a) + len(recs_b)


def test_same_seed_same_step_and_p0_is_the_dropout_free_path():
    _, sd, batch = C.ar_train_inputs()
    model = _build_train('ValleAR', C.AR_TINY_DROPOUT, sd)
    logits = []
    hook = model.proj.register_forward_hook(lambda m, i, o: None)
    hook.remove()
    from valle2_amd import autograd as A
    outs = []
    for seed in (31, 31, 32):
        model.zero_grad()
        torch.manual_seed(seed)
        lg = model._logits_with_graph(batch)
        lg.sum().backward()
        outs.append((lg.detach().clone(), model.transformer.layers[0].ffn.linear_1.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])                   # same seed: the same bits forward ...
    assert torch.equal(outs[0][1], outs[1][1])                   # ... and in a weight gradient (fixed-order GEMMs)
    assert not torch.equal(outs[0][0], outs[2][0])
    # p = 0 in train mode runs the very kernels of the dropout-free path (eval mode with gradients: what the goldens pin)
    m0 = _build_train('ValleAR', dict(C.AR_TINY, dropout=0.0), sd)
    for mod in (m0.tokens_position_emb.dropout, m0.audio_position_emb.dropout):
        mod.p = 0.0
    a = m0._logits_with_graph(batch).detach().clone()
    b = m0.eval()._logits_with_graph(batch).detach().clone()
    assert torch.equal(a, b)


def test_ranks_draw_different_fields():
    from valle2_amd import dropout
    try:
        dropout.set_rank(0)
        s0 = dropout.spec(1, dropout.site(dropout.ATTN_RES, 2), 0.1)
        dropout.set_rank(1)
        s1 = dropout.spec(1, dropout.site(dropout.ATTN_RES, 2), 0.1)
    finally:
        dropout.set_rank(0)
    a, b = dropout.mask(s0, 256, 512, DEV).double(), dropout.mask(s1, 256, 512, DEV).double()
    assert not torch.equal(a, b)
    both = 0.81
    assert abs((a * b).mean().item() - both) < 4 * (both * (1 - both) / a.numel()) ** 0.5


def test_module_level_forwards_in_train_mode_use_the_hip_fields():
    """EncoderLayer / FeedForward / PositionalEncoding called as modules in train mode (the reference's module API):
    dropout runs through vh_dropout — outputs differ from eval mode, are finite, and eval mode is deterministic."""
    from valle2_amd.modules import EncoderLayer, PositionalEncoding
    cfg = C.cfg_of(dict(C.TINY, norm='LayerNorm', dropout=0.1))
    layer = EncoderLayer(cfg).to(DEV)
    x = torch.randn(2, 40, cfg.d_model, generator=g(41)).to(DEV)
    ye, _ = layer.eval()(x)
    yt, _ = layer.train()(x)
    assert torch.isfinite(yt).all() and not torch.equal(ye, yt)
    assert torch.equal(ye, layer.eval()(x)[0])
    pe = PositionalEncoding(cfg.d_model).to(DEV).train()
    z = pe(x)
    frac = (z == 0).float().mean().item()
    assert 0.07 < frac < 0.13
