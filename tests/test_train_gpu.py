"""GPU parity of the training path: backward row kernels vs torch autograd on the CPU, and whole
models' losses / gradients vs the CPU oracle differentiated by torch autograd (and vs the per-
parameter gradient norms recorded from the real reference).

Tolerances: row-kernel gradients atol 2e-5 + rtol 1e-4 (fp32; column sums use atomics, so the
order of additions varies); model gradients rtol 1e-3 of the parameter's gradient norm
(grads rtol 1e-3); loss rtol 1e-5."""
import pytest
import torch
import torch.nn.functional as F

from tests.golden import cases as C
from tests.oracle_runners import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(a, b, atol=2e-5, rtol=1e-4):
    torch.testing.assert_close(a.detach().cpu(), b.detach(), atol=atol, rtol=rtol)


@pytest.mark.parametrize('rows,d,ada', [(7, 128, False), (300, 512, False), (65, 512, True), (9, 1024, True)])
def test_layernorm_backward(rows, d, ada):
    from valle2_amd import autograd as A
    x = (2 * torch.randn(rows, d, generator=g(1)) + 0.3).requires_grad_()
    gm, bt = (1 + 0.1 * torch.randn(d, generator=g(2))).requires_grad_(), (0.1 * torch.randn(d, generator=g(3))).requires_grad_()
    s, t = (1 + 0.2 * torch.randn(d, generator=g(4))).requires_grad_(), (0.2 * torch.randn(d, generator=g(5))).requires_grad_()
    dy = torch.randn(rows, d, generator=g(6))
    y = F.layer_norm(x, (d,), gm, bt, 1e-5)
    if ada:
        y = s * y + t
    y.backward(dy)
    dev = [v.detach().to(DEV).requires_grad_() for v in (x, gm, bt, s, t)]
    yd = A.layer_norm(dev[0], dev[1], dev[2], dev[3] if ada else None, dev[4] if ada else None)
    close(yd, y)
    yd.backward(dy.to(DEV))
    for a, b in zip(dev[:3] + (dev[3:] if ada else []), [x, gm, bt] + ([s, t] if ada else [])):
        close(a.grad, b.grad, atol=1e-4 if b.dim() == 1 else 2e-5)


def test_gelu_forward_backward():
    from valle2_amd import autograd as A
    x = (3 * torch.randn(37, 64, generator=g(7))).requires_grad_()
    dy = torch.randn(37, 64, generator=g(8))
    y = F.gelu(x)
    y.backward(dy)
    xd = x.detach().to(DEV).requires_grad_()
    yd = A.GeluFn.apply(xd)
    close(yd, y)
    yd.backward(dy.to(DEV))
    close(xd.grad, x.grad)


def test_gelu_dense_grid_matches_exact_erf():
    """The device erf is a two-branch polynomial / exp2 form (vh_common.h vh_erf), not libm's: check GELU against
    the float64 erf form on a dense grid, including the branch point |x|/sqrt(2) = 1 and the saturated tails.
    Tolerance: 1.2e-7 on erf -> 0.6e-7 * |x| on GELU, plus two fp32 roundings of the result."""
    from valle2_amd import autograd as A
    x = torch.linspace(-9.0, 9.0, 1 << 20)
    x = torch.cat([x, torch.tensor([0.0, -0.0, 2 ** 0.5, -2 ** 0.5, 1e-20, -1e-20, 30.0, -30.0])])
    x = torch.cat([x, torch.zeros((-len(x)) % 64)]).view(-1, 64)
    y = A.GeluFn.apply(x.to(DEV)).cpu().double()
    xd = x.double()
    ref = 0.5 * xd * (1.0 + torch.erf(xd * 0.5 ** 0.5))
    err = (y - ref).abs()
    bound = 1.0e-7 * xd.abs() + 2.0 ** -23 * ref.abs() + 1e-30
    assert bool((err <= bound).all()), float((err / bound).max())
    assert float(A.GeluFn.apply(torch.full((1, 64), -30.0, device=DEV)).abs().max()) == 0.0


def test_cross_entropy_forward_backward():
    from valle2_amd import autograd as A
    logits = (2 * torch.randn(45, 1025, generator=g(9))).requires_grad_()
    target = torch.randint(0, 1025, (45,), generator=g(10))
    loss = F.cross_entropy(logits, target)
    loss.backward()
    ld = logits.detach().to(DEV).requires_grad_()
    lossd = A.CrossEntropyFn.apply(ld, target.to(DEV))
    torch.testing.assert_close(lossd.cpu(), loss.detach(), rtol=1e-6, atol=1e-6)
    (3.0 * lossd).backward()
    close(ld.grad, 3.0 * logits.grad, atol=1e-7, rtol=1e-4)


def test_embedding_backward_and_colsum():
    from valle2_amd import autograd as A
    from valle2_amd.synth import sinusoid_table
    d = 128
    tabs = [torch.randn(30, d, generator=g(20 + j)).requires_grad_() for j in range(3)]
    ids = torch.randint(0, 30, (4, 9, 3), generator=g(30))
    pe = sinusoid_table(d, 32)
    dy = torch.randn(4, 9, d, generator=g(31))
    ref = sum(F.embedding(ids[..., j], tabs[j]) for j in range(3)) + pe[:9, 0]
    ref.backward(dy)
    dt = [t.detach().to(DEV).requires_grad_() for t in tabs]
    out = A.EmbedSumPeFn.apply(ids.to(DEV), pe.to(DEV), 0, None, *dt)
    close(out, ref)
    out.backward(dy.to(DEV))
    for a, b in zip(dt, tabs):
        close(a.grad, b.grad, atol=1e-5)
    # a padded batch: every row ends in a run of ONE id (the kernel adds such runs up before its atomic), 512 columns,
    # a length that is not a multiple of the 32 positions a workgroup walks
    d, T = 512, 107
    tab = torch.randn(40, d, generator=g(33)).requires_grad_()
    ids = torch.randint(0, 40, (3, T, 1), generator=g(34))
    for b, n_real in enumerate((107, 60, 1)):
        ids[b, n_real:] = 39
    pe = sinusoid_table(d, 128)
    dy = torch.randn(3, T, d, generator=g(35))
    ref = F.embedding(ids[..., 0], tab) + pe[:T, 0]
    ref.backward(dy)
    td = tab.detach().to(DEV).requires_grad_()
    out = A.EmbedSumPeFn.apply(ids.to(DEV), pe.to(DEV), 0, None, td)
    close(out, ref)
    out.backward(dy.to(DEV))
    close(td.grad, tab.grad, atol=2e-5)


def test_embed_concat_matches_parts_and_cat():
    """EmbedConcatFn (text | prefix frames over all codebooks | target frames over the first codebooks, one buffer, one
    gradient per table) against F.embedding + torch.cat, forward and every table's gradient — codebook tables are read
    by two parts."""
    from valle2_amd import autograd as A
    from valle2_amd.synth import sinusoid_table
    d, B, tx, p, t, q, stage = 128, 3, 11, 7, 26, 4, 2
    tok_tab = torch.randn(50, d, generator=g(80)).requires_grad_()
    tabs = [torch.randn(30, d, generator=g(81 + j)).requires_grad_() for j in range(q)]
    tokens = torch.randint(0, 50, (B, tx), generator=g(90))
    codes = torch.randint(0, 30, (B, t, q), generator=g(91))
    pe_t, pe_a = sinusoid_table(d, 64), sinusoid_table(d, 64)
    dy = torch.randn(B, tx + t, d, generator=g(92))
    ref = torch.cat([F.embedding(tokens, tok_tab) + pe_t[:tx, 0],
                     sum(F.embedding(codes[:, :p, j], tabs[j]) for j in range(q)) + pe_a[:p, 0],
                     sum(F.embedding(codes[:, p:, j], tabs[j]) for j in range(stage)) + pe_a[p:t, 0]], dim=1)
    ref.backward(dy)
    dev_tabs = [x.detach().to(DEV).requires_grad_() for x in [tok_tab] + tabs]
    cd, td = codes.to(DEV), tokens.to(DEV)
    spec = [(td, pe_t.to(DEV), 0, [0], None), (cd[:, :p], pe_a.to(DEV), 0, list(range(1, 1 + q)), None),
            (cd[:, p:], pe_a.to(DEV), p, list(range(1, 1 + stage)), None)]
    out = A.EmbedConcatFn.apply(spec, *dev_tabs)
    close(out, ref)
    out.backward(dy.to(DEV))
    for a, b in zip(dev_tabs, [tok_tab] + tabs):
        close(a.grad, b.grad, atol=2e-5)


@pytest.mark.parametrize('n,N,K_', [(5, 256, 128), (24, 1024, 512), (3, 200, 1024)])
def test_adaproj_forward_backward(n, N, K_):
    """AdaProjFn (every AdaptiveLayerNorm project_layer of a stack in one launch each way) against nn.Linear: outputs,
    weight / bias gradients (outer products) and the gradient of the shared stage embedding (every (layer, row block)
    adds onto the same K addresses: reduced over the workgroup's waves before the atomics)."""
    from valle2_amd import autograd as A
    emb = torch.randn(1, K_, generator=g(70)).requires_grad_()
    ws = [(0.1 * torch.randn(N, K_, generator=g(71 + i))).requires_grad_() for i in range(n)]
    bs = [torch.randn(N, generator=g(171 + i)).requires_grad_() for i in range(n)]
    dout = torch.randn(n, N, generator=g(72))
    ref = torch.cat([F.linear(emb, w, b) for w, b in zip(ws, bs)])
    ref.backward(dout)
    ed = emb.detach().to(DEV).requires_grad_()
    pd = [t.detach().to(DEV).requires_grad_() for pair in zip(ws, bs) for t in pair]
    out = A.AdaProjFn.apply(ed, *pd)
    close(out, ref, atol=2e-5)
    out.backward(dout.to(DEV))
    close(ed.grad, emb.grad, atol=1e-4, rtol=1e-4)
    for i in range(n):
        close(pd[2 * i].grad, ws[i].grad, atol=1e-6)
        close(pd[2 * i + 1].grad, bs[i].grad, atol=1e-6)


@pytest.mark.parametrize('mode', ['prefix', 'full'])
def test_qkv_attention_backward(mode):
    from oracle.valle_oracle import build_attn_mask
    from valle2_amd import autograd as A, kernels as K
    B, T, h = 2, 70, 2
    d = 64 * h
    x = torch.randn(B * T, d, generator=g(40)).requires_grad_()
    w = (0.1 * torch.randn(3 * d, d, generator=g(41))).requires_grad_()
    dy = torch.randn(B * T, d, generator=g(42))
    kvl = torch.tensor([T, T - 9], dtype=torch.int32)
    xl = 20
    keypad = torch.arange(T)[None, :] >= kvl[:, None]
    masked = (build_attn_mask(xl, T - xl)[None] | keypad[:, None, :]) if mode == 'prefix' else \
        keypad[:, None, :].expand(B, T, T)
    qkv = F.linear(x, w).view(B, T, 3, h, 64)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=~masked[:, None]).permute(0, 2, 1, 3).reshape(B * T, d)
    ref.backward(dy)
    xd, wd = x.detach().to(DEV).requires_grad_(), w.detach().to(DEV).requires_grad_()
    spec = dict(mode=K.MASK_PREFIX, x_len=xl, kv_len=kvl.to(DEV)) if mode == 'prefix' else \
        dict(mode=K.MASK_FULL, kv_len=kvl.to(DEV))
    out = A.QkvAttentionFn.apply(xd, wd, B, T, h, spec)
    close(out, ref, atol=3e-5)
    out.backward(dy.to(DEV))
    close(xd.grad, x.grad, atol=5e-5)
    close(wd.grad, w.grad, atol=2e-4)


def _grad_check(model, ref_params, names):
    worst = 0.0
    for n in names:
        got = dict(model.named_parameters())[n].grad
        assert got is not None, f'no gradient reached {n}'
        ref = ref_params[n].grad
        err = (got.cpu() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        worst = max(worst, err)
        assert err < 1e-3, f'{n}: relative gradient error {err:.2e}'
    return worst


def test_ar_training_step_gradients_vs_oracle_and_reference():
    from oracle import valle_oracle as O
    from tests.test_models_gpu import build
    gold = load_golden('ar_train')
    kw, sd, batch = C.ar_train_inputs()
    cfg = C.cfg_of(kw)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref_loss = O.ar_training_loss(params, cfg, batch)
    ref_loss.backward()
    model = build('ValleAR', kw, sd)            # eval mode: dropout off, as the fixture
    loss = model.training_step({k: v.clone() for k, v in batch.items()})
    assert loss.requires_grad
    torch.testing.assert_close(loss.detach().cpu(), gold['loss'], rtol=1e-5, atol=1e-6)
    loss.backward()
    names = sorted(k for k in params if not k.endswith('.pe'))
    _grad_check(model, params, names)
    norms = torch.stack([dict(model.named_parameters())[n].grad.norm().cpu() for n in names])
    torch.testing.assert_close(norms, gold['grad_norms'], rtol=1e-3, atol=1e-7)   # the real reference


@pytest.mark.parametrize('stage', [1, 5])
def test_nar_training_step_gradients_vs_oracle(stage):
    from oracle import valle_oracle as O
    from tests.test_models_gpu import build
    kw, sd, batch = C.nar_inputs()
    cfg = C.cfg_of(kw)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref_loss = O.nar_training_loss(params, cfg, batch, stage)
    ref_loss.backward()
    model = build('ValleNAR', kw, sd)
    loss = model.training_step(batch, stage=stage)
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), rtol=1e-5, atol=1e-6)
    loss.backward()
    used = sorted(k for k, v in params.items() if v.grad is not None and v.grad.abs().sum() > 0)
    assert f'stage_embs.{stage - 1}.word_embeddings.weight' in used
    _grad_check(model, params, used)
    for n, p in model.named_parameters():      # parameters the stage does not touch get no/zero grad
        if n not in used:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n


def test_train_loop_reduces_the_loss(tmp_path):
    from valle2_amd import synth
    from valle2_amd.config import ConfigValle
    from valle2_amd.train_model import train
    cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=256, num_layers=2, dropout=0.0, norm='LayerNorm',
                      lr=3e-3, max_steps=12, grad_accum=2, batch_size=3, log_every_n_steps=4, seed=5)
    fixed = synth.synth_ar_batch(cfg, 3, tok_range=(4, 8), code_range=(10, 20), seed=1)
    logs = []
    model, losses = train(cfg, 'ValleAR', batches=[fixed] * 24, log=logs.append)
    assert len(losses) == 24 and len(logs) == 3
    assert sum(losses[-4:]) / 4 < 0.8 * sum(losses[:4]) / 4, losses


# ---- optimizer step (optim.FlatAdamW = AdamW + global-norm clip + 1/world scale in one flat pass) ---------
@pytest.mark.parametrize('max_norm,grad_scale', [(1.0, 1.0), (0.0, 0.5), (1e9, 0.25)])
def test_flat_adamw_matches_torch_adamw_and_clip(max_norm, grad_scale):
    from valle2_amd.optim import FlatAdamW
    shapes = [(7, 5), (13,), (64, 64), (1025, 128), (3,), (2, 3, 4)]
    g0 = torch.Generator().manual_seed(7)
    init = [torch.randn(*s, generator=g0) for s in shapes]
    mine = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    kw = dict(lr=3e-3, betas=(0.9, 0.95), weight_decay=0.05)
    opt = FlatAdamW(mine, **kw)
    ropt = torch.optim.AdamW(ref, eps=1e-8, **kw)
    for p in mine:                                           # parameters live in the flat buffer, 16-B aligned
        assert p.data_ptr() % 16 == 0 and p.grad is None     # gradients start released (set_to_none semantics)
    assert all(opt.grad_view(sl).data_ptr() % 16 == 0 for sl in opt.slots)
    for step in range(6):
        lr = 3e-3 * (1.0 - 0.1 * step)                       # a scheduler changing the group's lr
        opt.param_groups[0]['lr'] = ropt.param_groups[0]['lr'] = lr
        grads = [(10.0 if step == 2 else 1.0) * torch.randn(*s, generator=g0) for s in shapes]
        for p, r, g in zip(mine, ref, grads):
            p.grad = g.to(DEV)                               # assigned (or produced by autograd): counts as present
            r.grad = (g * grad_scale).to(DEV)
        norm = opt.step(grad_scale=grad_scale, max_norm=max_norm, zero_grad=True)
        ref_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm if max_norm > 0 else float('inf'))
        ropt.step()
        torch.testing.assert_close(norm.cpu().reshape(()), ref_norm.cpu().reshape(()), rtol=1e-5, atol=0)
        for p, r in zip(mine, ref):
            torch.testing.assert_close(p.detach().cpu(), r.detach().cpu(), rtol=2e-5, atol=2e-7)
            assert p.grad is None                            # zero_grad=True: flat gradient cleared, grads released
        assert float(opt.flat_grad.abs().sum()) == 0.0
    # bitwise reproducible: the same sequence again gives the same bits
    again = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    opt2 = FlatAdamW(again, **kw)
    g1 = torch.Generator().manual_seed(7)
    _ = [torch.randn(*s, generator=g1) for s in shapes]
    for step in range(6):
        opt2.param_groups[0]['lr'] = 3e-3 * (1.0 - 0.1 * step)
        for p, s in zip(again, shapes):
            p.grad = ((10.0 if step == 2 else 1.0) * torch.randn(*s, generator=g1)).to(DEV)
        opt2.step(grad_scale=grad_scale, max_norm=max_norm, zero_grad=True)
    assert torch.equal(opt.flat_param, opt2.flat_param)


def test_flat_adamw_drives_the_model_and_invalidates_folded_weights():
    """configure_optimizers → FlatAdamW: module parameters become views of the flat buffer, a step
    changes what the kernels read, and the decode path's folded LayerNorm weights are rebuilt."""
    from tests.test_models_gpu import build
    from valle2_amd import engine
    kw, sd, batch = C.ar_train_inputs()
    model = build('ValleAR', kw, sd).train()
    before = engine.folded_layer_norms(model.transformer)
    conf = model.configure_optimizers()
    opt, sched = conf['optimizer'], conf['lr_scheduler']
    assert all(p.data_ptr() >= opt.flat_param.data_ptr() for p in model.parameters())
    loss0 = model.training_step({k: v.clone() for k, v in batch.items()})
    loss0.backward()
    # every first gradient of the step was written straight into its slice and adopted by autograd: no copies
    assert all(s[0].grad.data_ptr() == opt.grad_view(s).data_ptr() for s in opt.slots)
    opt.step(max_norm=model.config.gradient_clip_val, zero_grad=True)
    sched.step()
    after = engine.folded_layer_norms(model.transformer)
    assert after is not before
    w = model.transformer.layers[0].self_attn.qkv.weight.detach()
    g = model.transformer.layers[0].norm1.weight.detach()
    torch.testing.assert_close(after[0][0][0], w * g, rtol=0, atol=0)
    loss1 = model.training_step({k: v.clone() for k, v in batch.items()})
    assert float(loss1.detach()) < float(loss0.detach())
    # a checkpoint loads INTO the flat buffer (parameters stay views of it) and the optimizer state round-trips
    ptrs = [p.data_ptr() for p in model.parameters()]
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    assert [p.data_ptr() for p in model.parameters()] == ptrs
    w0 = model.transformer.layers[0].self_attn.qkv.weight
    assert torch.equal(w0.detach().cpu(), sd['transformer.layers.0.self_attn.qkv.weight'])
    off = next(o for p_, o, n in opt.slots if p_ is w0)
    assert torch.equal(opt.flat_param[off:off + w0.numel()].view_as(w0), w0.detach())
    state = opt.state_dict()
    opt2 = type(opt)(model.parameters(), lr=1.0)
    opt2.load_state_dict(state)
    assert opt2.steps == opt.steps and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert opt2.param_groups[0]['lr'] == opt.param_groups[0]['lr']


def test_flat_adamw_gradient_accumulation_adds_onto_the_adopted_slices():
    """Two backward passes without zero_grad in between (accumulate_grad_batches = 2): the first writes every gradient
    straight into its slice of the flat buffer (autograd adopts the slice), the second is added onto it in place — the
    flat gradient is the sum of the two passes' gradients computed separately, and every p.grad is still its slice."""
    from tests.test_models_gpu import build
    kw, sd, batch = C.ar_train_inputs()
    model = build('ValleAR', kw, sd)                          # eval mode: dropout off, the passes are repeatable
    opt = model.configure_optimizers()['optimizer']
    b1 = {k: v.clone() for k, v in batch.items()}
    b2 = {k: v.clone() for k, v in batch.items()}
    b2['codes'] = (b2['codes'] + 7) % 1000                     # a different micro-batch of the same shape
    singles = []
    for b in (b1, b2):
        opt.zero_grad()
        assert all(p.grad is None for p in model.parameters())
        model.training_step({k: v.clone() for k, v in b.items()}).backward()
        singles.append(opt.flat_grad.clone())
    # the same gradients as a model without the flat optimizer produces (plain autograd accumulation)
    plain = build('ValleAR', kw, sd)
    plain.training_step({k: v.clone() for k, v in b1.items()}).backward()
    got = {n: opt.grad_view(sl) for sl in opt.slots for n, p in model.named_parameters() if p is sl[0]}
    opt.zero_grad()
    model.training_step({k: v.clone() for k, v in b1.items()}).backward()
    for n, p in plain.named_parameters():
        torch.testing.assert_close(got[n], p.grad, rtol=1e-5, atol=1e-7)
    opt.zero_grad()
    model.training_step(b1).backward()
    model.training_step(b2).backward()
    assert all(s[0].grad.data_ptr() == opt.grad_view(s).data_ptr() for s in opt.slots)
    torch.testing.assert_close(opt.flat_grad, singles[0] + singles[1], rtol=1e-5, atol=1e-6)
    assert float(singles[0].abs().sum()) > 0 and not torch.equal(singles[0], singles[1])


# ---- attention backward kernels (vh_attn_rows_bwd) directly against torch autograd -----------------------
@pytest.mark.parametrize('form', [0, 1])         # VH_TUNE_ATTN_BWD: 0 = five products in one kernel + slab reduce, 1 = two kernels
@pytest.mark.parametrize('B,h,T,mode', [(2, 2, 70, 'prefix'), (3, 2, 300, 'prefix'), (2, 4, 257, 'full'),
                                         (1, 2, 129, 'explicit'), (2, 1, 33, 'full'), (2, 2, 1021, 'prefix'),
                                         (2, 1, 640, 'full'), (1, 2, 515, 'explicit'), (1, 1, 256, 'prefix')])
def test_attn_rows_bwd_vs_torch_autograd(B, h, T, mode, form):
    from valle2_amd import _lib
    _lib.lib().vh_set_tuning(13, form)
    try:
        _attn_rows_bwd_vs_torch_autograd(B, h, T, mode)
    finally:
        _lib.lib().vh_set_tuning(13, 0)


def test_attn_rows_bwd_row_without_valid_keys_leaves_zero_gradients():
    """Round-4 advisor finding: a batch row with kv_len = 0 in a single-chunk launch (T <= 256: no slab, no reduce launch)
    took the fused kernel's all-padding exit, which zeroed dK / dV but never wrote dQ — the row's dq slice kept whatever
    the caller's torch.empty held.  It must be zero (no key is attended, nothing depends on q), and the other rows exact."""
    from valle2_amd import kernels as K
    B, h, T = 3, 2, 40
    d = 64 * h
    gen = torch.Generator().manual_seed(77)
    q = torch.randn(B * T, d, generator=gen).to(DEV)
    k = torch.randn(B, h, T, 64, generator=gen).to(DEV)
    v = torch.randn(B, h, T, 64, generator=gen).to(DEV)
    dout = torch.randn(B * T, d, generator=gen).to(DEV)
    kvl = torch.tensor([T, 0, 17], dtype=torch.int32, device=DEV)
    out = torch.empty(B * T, d, device=DEV)
    lse2 = torch.empty(B, h, T, device=DEV)
    K.attn_rows(q, k, v, out, B, h, T, T, lse2=lse2, mode=K.MASK_FULL, kv_len=kvl)
    dqkv = torch.full((B * T, 3 * d), float('nan'), device=DEV)
    K.attn_rows_bwd(q, k, v, out, dout, lse2, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, T, mode=K.MASK_FULL, kv_len=kvl)
    row1 = dqkv.view(B, T, 3 * d)[1]
    assert float(row1.abs().max()) == 0.0, 'the row without valid keys must get zero dq / dk / dv'
    rest = dqkv.view(B, T, 3 * d)[[0, 2]]
    assert bool(torch.isfinite(rest).all())


def _attn_rows_bwd_vs_torch_autograd(B, h, T, mode):
    from oracle.valle_oracle import build_attn_mask
    from valle2_amd import kernels as K
    d = 64 * h
    gen = torch.Generator().manual_seed(T)
    q = torch.randn(B, h, T, 64, generator=gen).requires_grad_()
    k = torch.randn(B, h, T, 64, generator=gen).requires_grad_()
    v = torch.randn(B, h, T, 64, generator=gen).requires_grad_()
    dout = torch.randn(B, T, d, generator=gen)
    kvl = torch.randint(max(T // 2, 1), T + 1, (B,), generator=gen, dtype=torch.int32)
    kvl[0] = T
    xl = T // 4
    keypad = torch.arange(T)[None, :] >= kvl[:, None]
    if mode == 'prefix':
        masked = build_attn_mask(xl, T - xl)[None] | keypad[:, None, :]
        spec = dict(mode=K.MASK_PREFIX, x_len=xl, kv_len=kvl.to(DEV))
    elif mode == 'full':
        masked = keypad[:, None, :].expand(B, T, T)
        spec = dict(mode=K.MASK_FULL, kv_len=kvl.to(DEV))
    else:
        m2 = torch.rand(T, T, generator=gen) < 0.3
        m2[torch.arange(T), torch.arange(T)] = False            # every row keeps at least its own key
        masked = m2[None] | keypad[:, None, :]
        masked[:, torch.arange(T), torch.arange(T)] = False
        pad = keypad.clone()
        spec = dict(mode=K.MASK_EXPLICIT, mask=m2.to(torch.uint8).to(DEV), pad=None)
        masked = m2[None].expand(B, T, T)
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=~masked[:, None])
    ref.permute(0, 2, 1, 3).reshape(B, T, d).backward(dout)
    # device: q as (B*T, d) rows with heads in columns; k, v in the cache layout
    qd = q.detach().permute(0, 2, 1, 3).reshape(B * T, d).contiguous().to(DEV)
    kd, vd = k.detach().contiguous().to(DEV), v.detach().contiguous().to(DEV)
    out = torch.empty(B * T, d, device=DEV)
    lse2 = torch.empty(B, h, T, device=DEV)
    K.attn_rows(qd, kd, vd, out, B, h, T, T, lse2=lse2, **spec)
    close(out, ref.detach().permute(0, 2, 1, 3).reshape(B * T, d), atol=3e-5)
    dqkv = torch.full((B * T, 3 * d), float('nan'), device=DEV)
    K.attn_rows_bwd(qd, kd, vd, out, dout.reshape(B * T, d).to(DEV), lse2, dqkv[:, :d], dqkv[:, d:2 * d],
                    dqkv[:, 2 * d:], B, h, T, **spec)
    got = dqkv.cpu().view(B, T, 3, h, 64).permute(2, 0, 3, 1, 4)
    for name, gg, rr in zip('qkv', got, (q.grad, k.grad, v.grad)):
        torch.testing.assert_close(gg, rr, atol=6e-5, rtol=1e-4, msg=lambda m: f'd{name}: {m}')
    # bitwise reproducible (no atomics)
    dqkv2 = torch.empty_like(dqkv)
    K.attn_rows_bwd(qd, kd, vd, out, dout.reshape(B * T, d).to(DEV), lse2, dqkv2[:, :d], dqkv2[:, d:2 * d],
                    dqkv2[:, 2 * d:], B, h, T, **spec)
    assert torch.equal(dqkv, dqkv2)


@pytest.mark.parametrize('mode', ['prefix', 'full'])
def test_attention_backward_materialized_engine(mode):
    from valle2_amd import autograd as A
    old = A.ATTENTION_BACKWARD
    A.ATTENTION_BACKWARD = 'materialized'
    try:
        test_qkv_attention_backward(mode)
    finally:
        A.ATTENTION_BACKWARD = old


def test_flat_adamw_skips_parameters_without_gradient_like_torch():
    """torch.optim.AdamW skips a parameter whose grad is None (no decay, no moment update) and counts `step` per
    parameter — what happens to the NAR heads / stage embeddings of the 6 stages a step does not train.  FlatAdamW
    must do the same (per-slot step counts in the kernel), through autograd and through assigned gradients."""
    from valle2_amd.optim import FlatAdamW
    g0 = torch.Generator().manual_seed(11)
    shapes = [(64, 32), (7,), (130, 16), (3, 5)]
    init = [torch.randn(*s, generator=g0) for s in shapes]
    mine = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    kw = dict(lr=2e-3, betas=(0.9, 0.98), weight_decay=0.1)
    opt, ropt = FlatAdamW(mine, **kw), torch.optim.AdamW(ref, **kw)
    used = [(0, 1, 2, 3), (0, 2), (0, 1), (0, 2, 3), (0,), (0, 1, 2, 3)]       # which parameters each step trains
    for step, idx in enumerate(used):
        x = torch.randn(4, generator=g0).to(DEV)
        for params, o in ((mine, opt), (ref, ropt)):
            o.zero_grad(set_to_none=True) if o is ropt else o.zero_grad()
            loss = sum((params[i] * (1.0 + x[i])).square().sum() * (0.1 + 0.05 * step) for i in idx)
            loss.backward()
            o.step()
        for i, (a, b) in enumerate(zip(mine, ref)):
            torch.testing.assert_close(a.detach(), b.detach(), atol=2e-6, rtol=2e-5, msg=lambda s, i=i: f'step {step} param {i}: {s}')
    assert opt.slot_steps != [len(used)] * 4 and max(opt.slot_steps) == len(used)
    # an untouched parameter did not move at all in a step that skipped it
    before = [p.detach().clone() for p in mine]
    opt.zero_grad()
    (mine[0] * 2.0).sum().backward()
    opt.step()
    assert not torch.equal(mine[0], before[0]) and all(torch.equal(mine[i], before[i]) for i in (1, 2, 3))


# ---- round-3 advisor findings --------------------------------------------------------------------------------
def _tiny_ar(seed=3):
    from valle2_amd import get_model_class, synth
    cfg = C.cfg_of(dict(C.AR_TINY))
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 'ValleAR', seed=seed, rich=True))
    return cfg, m.to(DEV).train()


def test_poisoned_batch_leaves_host_and_device_state_consistent():
    """A step whose kernels saw a bad id is guarded off on the device.  The host must agree: step counters rolled back,
    gradient bookkeeping reset (the guarded launch still clears the flat gradient), so that a caller who catches the
    IndexError and skips the batch carries on from exactly the state before the bad batch — nothing stale in the flat
    gradient for the next backward's accumulate-type kernels, bias corrections not a step ahead."""
    from valle2_amd import synth
    cfg, a = _tiny_ar()
    good = [synth.synth_ar_batch(cfg, 2, tok_range=(5, 9), code_range=(13, 20), seed=s) for s in (1, 2)]
    good = [{k: (v.to(DEV) if not k.endswith('_lens') else v) for k, v in g.items()} for g in good]
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in good[1].items()}
    bad['target'][0, 2] = -100
    opt = a.configure_optimizers()['optimizer']

    def one_step(batch):
        with torch.enable_grad():
            a.training_step(batch).backward()
        opt.step(max_norm=1.0, zero_grad=True)
        opt.check_errors()                                  # synchronises: this step's flag
    one_step(good[0])
    snap = [t.clone() for t in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq)]
    counters = (opt.steps, list(opt.slot_steps))
    with pytest.raises(IndexError, match='target'):
        one_step(bad)
    assert (opt.steps, opt.slot_steps) == counters, 'host step counters ran ahead of the device'
    for now, was in zip((opt.flat_param, opt.exp_avg, opt.exp_avg_sq), snap):
        assert torch.equal(now, was), 'the guarded step touched parameters or moments'
    assert float(opt.flat_grad.abs().sum()) == 0.0, 'stale gradient of the skipped step left in the flat buffer'
    assert all(opt._clean) and not any(opt._claimed) and not any(opt._touched)
    assert all(p.grad is None for p, _, _ in opt.slots)
    # the next step runs as step 2 on a clean buffer: the gradient it sees is this batch's alone
    torch.manual_seed(77)                                   # (the position dropout draws from torch's generator)
    with torch.enable_grad():
        a.training_step(good[1]).backward()
    opt.gather_grads()
    g_after_skip = opt.flat_grad.clone()
    opt.step(max_norm=1.0, zero_grad=True)
    opt.check_errors()
    assert opt.steps == 2 and not torch.equal(opt.flat_param, snap[0])
    # the same batch on the same parameters from a freshly zeroed buffer gives the same gradient (to the rounding noise of
    # the atomics in the column-sum / scatter kernels): nothing of the poisoned batch was mixed in
    opt.flat_param.copy_(snap[0])
    opt.zero_grad()
    torch.manual_seed(77)
    with torch.enable_grad():
        a.training_step(good[1]).backward()
    opt.gather_grads()
    torch.testing.assert_close(g_after_skip, opt.flat_grad, atol=1e-5, rtol=1e-4)


def test_a_second_flat_adamw_detaches_the_first():
    """configure_optimizers() twice (a resume): the old optimizer's hooks must not keep it — four flat buffers — alive,
    nor keep marking its slots; its registry entries go with it."""
    import gc
    import weakref
    from valle2_amd import optim, synth
    cfg, m = _tiny_ar()
    first = m.configure_optimizers()['optimizer']
    ref = weakref.ref(first)
    n_slots = len(optim.GRAD_SLOTS)
    second = m.configure_optimizers()['optimizer']
    assert first._hooks == [] and len(optim.GRAD_SLOTS) == n_slots        # re-keyed to the new flat buffer
    del first
    gc.collect()
    assert ref() is None, 'the first optimizer is still referenced (hook closures?)'
    batch = synth.synth_ar_batch(cfg, 2, tok_range=(5, 9), code_range=(13, 20), seed=1)
    before = second.flat_param.clone()
    with torch.enable_grad():
        m.training_step(batch).backward()
    second.step()
    second.check_errors()
    assert not torch.equal(second.flat_param, before)
    assert all(p.data_ptr() == second.flat_param[off:off + n].data_ptr() for p, off, n in second.slots)


@pytest.mark.parametrize('n_batches,accum,max_steps,expect_steps', [(3, 2, 4, 4), (1, 4, 3, 3), (5, 2, 3, 3)])
def test_train_accumulation_windows_end_with_the_epoch(n_batches, accum, max_steps, expect_steps):
    """A re-iterable dataset whose length is not a multiple of grad_accum (or shorter than it): every epoch ends with
    an optimizer step on the incomplete window (Lightning's behaviour), leftovers never leak into the next epoch, and
    the loop terminates."""
    from valle2_amd import synth
    from valle2_amd.config import ConfigValle
    from valle2_amd.train_model import train
    cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=256, num_layers=2, dropout=0.0, norm='LayerNorm',
                      lr=1e-3, max_steps=max_steps, grad_accum=accum, batch_size=2, log_every_n_steps=1000, seed=5)
    data = [synth.synth_ar_batch(cfg, 2, tok_range=(4, 8), code_range=(10, 20), seed=s) for s in range(n_batches)]
    seen = []

    class Data:                                   # re-iterable, counts what the loop consumed
        def __iter__(self):
            for i, b in enumerate(data):
                seen.append(i)
                yield b
    model, losses = train(cfg, 'ValleAR', batches=Data())
    windows_per_epoch = -(-n_batches // accum)
    full_epochs, rest = divmod(expect_steps, windows_per_epoch)
    want = full_epochs * n_batches + (min(n_batches, rest * accum) if rest else 0)
    assert len(losses) == len(seen) == want, (len(losses), want)


def test_attn_rows_bwd_forms_agree_on_random_shapes():
    """Property test: the five-product kernel (both wave counts, every chunk plan the rule picks) and the two-kernel form
    compute the same dQ / dK / dV on 40 random (batch, heads, length, mask family, prefix length, key lengths) draws —
    lengths that are no multiple of the 32-key wave block or the 32-query tile, rows whose keys end inside a chunk, chunks
    that are all padding, one-key rows — and a forced chunk count never changes more than the summation order."""
    from valle2_amd import _lib
    from valle2_amd import kernels as K
    L = _lib.lib()
    g = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))     # noqa: E731
    for trial in range(40):
        B, h = ri(1, 5), ri(1, 4)
        T = [ri(1, 40), ri(33, 300), ri(257, 700), ri(700, 1300)][trial % 4]
        mode = [K.MASK_PREFIX, K.MASK_FULL, K.MASK_EXPLICIT][trial % 3]
        d = 64 * h
        q = torch.randn(B * T, d, generator=g).to(DEV)
        k = torch.randn(B, h, T, 64, generator=g).to(DEV)
        v = torch.randn(B, h, T, 64, generator=g).to(DEV)
        dout = torch.randn(B * T, d, generator=g).to(DEV)
        kvl = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
        if trial % 5 == 0:
            kvl[0] = T
        if mode == K.MASK_EXPLICIT:
            m2 = torch.rand(T, T, generator=g) < 0.4
            m2[torch.arange(T), torch.arange(T)] = False
            spec = dict(mode=mode, mask=m2.to(torch.uint8).to(DEV))
        elif mode == K.MASK_PREFIX:
            xl = ri(0, T)
            # a query must see at least one key: rows shorter than the prefix still see keys < their length
            spec = dict(mode=mode, x_len=xl, kv_len=kvl.to(DEV))
        else:
            spec = dict(mode=mode, kv_len=kvl.to(DEV))
        out = torch.empty(B * T, d, device=DEV)
        lse2 = torch.empty(B, h, T, device=DEV)
        K.attn_rows(q, k, v, out, B, h, T, T, lse2=lse2, **spec)
        res = []
        for form, chunks in ((1, 0), (0, 0), (0, ri(1, 12))):
            L.vh_set_tuning(13, form)
            L.vh_set_tuning(14, chunks)
            try:
                dqkv = torch.full((B * T, 3 * d), float('nan'), device=DEV)
                K.attn_rows_bwd(q, k, v, out, dout, lse2, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, T, **spec)
                again = torch.empty_like(dqkv)
                K.attn_rows_bwd(q, k, v, out, dout, lse2, again[:, :d], again[:, d:2 * d], again[:, 2 * d:], B, h, T, **spec)
            finally:
                L.vh_set_tuning(13, 0)
                L.vh_set_tuning(14, 0)
            assert torch.equal(dqkv, again), (trial, form, chunks)
            res.append(dqkv)
        ok = torch.isfinite(res[0])                  # (a fully masked query row is NaN in every form, as SDPA gives)
        for r in res[1:]:
            assert torch.equal(torch.isfinite(r), ok), trial
            torch.testing.assert_close(r[ok], res[0][ok], atol=3e-5, rtol=1e-4, msg=lambda m: f'trial {trial} B={B} h={h} T={T} mode={mode}: {m}')


@pytest.mark.parametrize('p', [0.0, 0.1])
def test_a_gradient_hook_that_edits_a_layer_outputs_gradient_in_place_is_honoured(p):
    """Round-4 advisor finding: a layer's backward re-used the pre-dropped copy of its incoming gradient and the column
    sums the layer above had made of it whenever the data pointer matched — a tensor hook that scales the gradient IN PLACE
    keeps the pointer.  The hand-over now also compares the version counter: an in-place hook, a hook that returns a new
    tensor and no hook on a pre-scaled loss all give the same gradients (same seed: same dropout fields)."""
    from valle2_amd import ConfigValle, autograd as A, dropout, get_model_class, synth
    cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=256, num_layers=3, dropout=p, norm='LayerNorm')
    sd = synth.make_state_dict(cfg, 'ValleAR', seed=4, rich=True)
    batch = synth.synth_ar_batch(cfg, 2, tok_range=(5, 9), code_range=(12, 20), seed=5)

    def run(hook):
        m = get_model_class('ValleAR')(cfg)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        m.tokens_position_emb.dropout.p = m.audio_position_emb.dropout.p = 0.0
        dropout.manual_seed(99)
        A.LAYER_OUTPUT_HOOK = hook
        try:
            with torch.enable_grad():
                m.training_step({k: v.clone() for k, v in batch.items()}).backward()
        finally:
            A.LAYER_OUTPUT_HOOK = None
        return {n: q.grad.clone() for n, q in m.named_parameters()}

    def in_place(i, g):
        if i == 1:
            g.mul_(0.5)                      # edits the tensor autograd hands on; returns None

    def new_tensor(i, g):
        return g * 0.5 if i == 1 else None

    a, b = run(in_place), run(new_tensor)
    base = run(None)
    for n in a:
        torch.testing.assert_close(a[n], b[n], atol=1e-7, rtol=1e-5, msg=lambda s, n=n: f'{n}: {s}')
    changed = [n for n in a if 'layers.0.' in n and not torch.allclose(a[n], base[n], rtol=1e-3, atol=1e-9)]
    assert changed, 'the hook must have changed the gradients below it'
    top = [n for n in a if 'layers.2.' in n]
    # (compared to rounding: the LayerNorm backward's column sums are fp32 atomics, not bit-reproducible run to run)
    assert all(torch.allclose(a[n], base[n], rtol=1e-5, atol=1e-8) for n in top), 'layers above the hook are untouched'
