"""A head width other than 64.  valle/models/modules.py:109-111 allows any divisor of d_model (`head_dim = d_model //
n_heads`); every configuration of the path and of the reference's own tests has 64, which is what the flash / decode
kernels and the native composites are built for.  Other widths run on the general kernels (batched MFMA GEMM + masked row
softmax with the probabilities materialised: kernels.attn_generic, engine._transformer_forward_generic,
autograd.QkvAttentionAnyHeadDimFn, and generation by recomputation) — these tests pin them to the CPU oracle, which
restates the reference for any width: modules (masks, cache protocol), both stacks, AR greedy generate token for
token, AR / NAR training loss and every gradient.  Tolerances as everywhere: activations atol 2e-4 / rtol 1e-4, loss rtol
1e-5, gradients 1e-3 of the parameter's gradient norm."""
import pytest
import torch

from tests.golden import cases as C

pytestmark = pytest.mark.gpu
DEV = 'cuda'
WIDTHS = [(128, 4), (256, 2), (96, 2)]          # head_dim 32, 128, 48


def close(a, b, atol=2e-4, rtol=1e-4):
    torch.testing.assert_close(a.detach().cpu(), b.detach(), atol=atol, rtol=rtol)


@pytest.mark.parametrize('d,h', [(128, 2), (128, 4)])
def test_multi_head_attention_takes_one_mask_per_batch_row(d, h):
    """modules.py:187-188: a 3-D attn_mask (B,T,T) is one mask per batch row ('b t t -> b 1 t t'), merged with the key padding by
    addition.  The HIP path hands it to the explicit-mask mode with a batch stride (vh_attn_rows_bmask; the materialised path
    at other head widths applies it row by row).  Against oracle.multi_head_attention."""
    from oracle import valle_oracle as O
    from valle2_amd.modules import MultiHeadAttention
    b, n = 3, 70
    g = torch.Generator().manual_seed(3)
    m = MultiHeadAttention(d, h)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(b, n, d, generator=g)
    am = (torch.rand(b, n, n, generator=g) < 0.3).float()
    am[:, torch.arange(n), torch.arange(n)] = 0          # every query sees itself: no empty rows
    pm = torch.zeros(b, n)
    pm[1, -9:] = 1
    pm[2, -1:] = 1
    m = m.to(DEV).eval()
    with torch.no_grad():
        out, _ = m(x.to(DEV), attn_mask=am.to(DEV), padding_mask=pm.to(DEV))
        ref, _ = O.multi_head_attention(sd, '', x, h, attn_mask=am, padding_mask=pm)
    torch.testing.assert_close(out.cpu(), ref, atol=3e-5, rtol=1e-4)
    with torch.no_grad():                                  # without padding, and a float mask with values other than 1
        out, _ = m(x.to(DEV), attn_mask=(3 * am).to(DEV))
        ref, _ = O.multi_head_attention(sd, '', x, h, attn_mask=3 * am)
    torch.testing.assert_close(out.cpu(), ref, atol=3e-5, rtol=1e-4)


@pytest.mark.parametrize('d,h', WIDTHS)
def test_multi_head_attention_any_head_dim_matches_the_oracle(d, h):
    from oracle import valle_oracle as O
    from valle2_amd.modules import MultiHeadAttention
    from valle2_amd.utils import build_attn_mask, build_pad_mask
    g = torch.Generator().manual_seed(d + h)
    b, t = 3, 37
    m = MultiHeadAttention(d, h)
    sd = {k: 0.2 * torch.randn(v.shape, generator=g) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    assert m.head_dim == d // h != 64
    x = torch.randn(b, t, d, generator=g)
    causal = torch.triu(torch.ones(t, t), diagonal=1)
    lens = torch.tensor([t, t - 9, 5])
    pad = build_pad_mask(lens, 'cpu')
    for am, pm in ((causal, None), (causal, pad), (build_attn_mask(11, t - 11, 'cpu'), pad), (None, None)):
        ref, (rk, rv) = O.multi_head_attention(sd, '', x, h, attn_mask=am, padding_mask=pm, use_cache=True)
        out, (k, v) = m(x.to(DEV), attn_mask=None if am is None else am.to(DEV),
                        padding_mask=None if pm is None else pm.to(DEV), use_cache=True)
        close(out, ref)
        close(k, rk)
        close(v, rv)
        assert k.shape == (b, h, t, d // h)
    # a cached step on the cache the module returned (valle/models/modules.py:149-157)
    xn = torch.randn(b, 1, d, generator=g)
    ref, (rk, _) = O.multi_head_attention(sd, '', xn, h, kv_cache=(rk, rv), use_cache=True)
    out, (k2, _) = m(xn.to(DEV), kv_cache=(k, v), use_cache=True)
    close(out, ref)
    close(k2, rk)


@pytest.mark.parametrize('norm', ['LayerNorm', 'AdaptiveLayerNorm'])
@pytest.mark.parametrize('d,h', WIDTHS[:2])
def test_transformer_any_head_dim_matches_the_oracle(d, h, norm):
    from oracle import valle_oracle as O
    from valle2_amd import synth
    from valle2_amd.modules import Transformer
    from valle2_amd.utils import build_attn_mask, build_pad_mask
    kw = dict(d_model=d, n_heads=h, dim_feedforward=2 * d, num_layers=2, dropout=0.0, norm=norm)
    cfg = C.cfg_of(kw)
    g = torch.Generator().manual_seed(7)
    tr = Transformer(cfg)
    sd = {k: (0.1 * torch.randn(v.shape, generator=g) + (1.0 if k.endswith('norm.weight') or k.endswith('norm1.weight')
                                                            or k.endswith('norm2.weight') else 0.0))
          for k, v in tr.state_dict().items()}
    tr.load_state_dict(sd)
    tr = tr.to(DEV).eval()
    b, xl, yl = 3, 20, 70                       # 270 rows: the path the native composite would have taken
    x = torch.randn(b, xl + yl, d, generator=g)
    emb = torch.randn(1, d, generator=g) if norm != 'LayerNorm' else None
    am = build_attn_mask(xl, yl, 'cpu')
    pad = build_pad_mask(torch.tensor([xl + yl, xl + 33, xl + 1]), 'cpu')
    psd = {'t.' + k: v for k, v in sd.items()}
    ref, rkv = O.transformer(psd, 't.', x, cfg, padding_mask=pad, attn_mask=am, embedding=emb, use_cache=True)
    y, kv = tr(x.to(DEV), padding_mask=pad.to(DEV), attn_mask=am.to(DEV),
               embedding=None if emb is None else emb.to(DEV), use_cache=True)
    close(y, ref)
    assert len(kv) == 2
    close(kv[1][0], rkv[1][0])
    # cached continuation: one more row through every layer's (k, v)
    xn = torch.randn(b, xl + yl + 1, d, generator=g)
    ref2, _ = O.transformer(psd, 't.', xn, cfg, embedding=emb, kv_cache=rkv, use_cache=True)
    y2, _ = tr(xn.to(DEV), embedding=None if emb is None else emb.to(DEV), kv_cache=kv, use_cache=True)
    close(y2, ref2)
    del synth


@pytest.mark.parametrize('d,h', WIDTHS[:2])
def test_ar_model_any_head_dim_generates_and_trains_like_the_oracle(d, h):
    from oracle import valle_oracle as O
    from tests.test_train_gpu import _grad_check
    from valle2_amd import get_model_class, synth
    kw = dict(d_model=d, n_heads=h, dim_feedforward=2 * d, num_layers=2, dropout=0.0, norm='LayerNorm', num_beams=3,
              top_k=1, max_audio_len=24)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=21, rich=True), cfg)
    model = get_model_class('ValleAR')(cfg)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    # greedy generate, token for token (margins as the other generate tests use them)
    utt = synth.synth_utterance(cfg, 9, 7, 25, seed=77)
    trace = {}
    ref = O.ar_generate(sd, cfg, *utt, trace=trace)
    out = model.generate(*[u.to(DEV) for u in utt]).cpu()
    n = min(len(out), len(ref))
    bad = (out[:n] != ref[:n]).nonzero()
    assert len(out) == len(ref) and (bad.numel() == 0 or trace['margin'][int(bad[0])] < 1e-4), (out, ref)
    # teacher-forced loss and every gradient
    batch = synth.synth_ar_batch(cfg, 3, tok_range=(5, 11), code_range=(20, 45), seed=8)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    rl = O.ar_training_loss(params, cfg, batch)
    rl.backward()
    loss = model.training_step({k: v.clone() for k, v in batch.items()})
    torch.testing.assert_close(loss.detach().cpu(), rl.detach(), rtol=1e-5, atol=1e-6)
    loss.backward()
    _grad_check(model, params, sorted(k for k in params if not k.endswith('.pe')))
    with torch.no_grad():
        close(model.training_step({k: v.clone() for k, v in batch.items()}), rl.detach(), atol=1e-5)   # the inference kernels


def test_nar_model_any_head_dim_matches_the_oracle():
    from oracle import valle_oracle as O
    from tests.test_train_gpu import _grad_check
    from valle2_amd import get_model_class, synth
    kw = dict(d_model=128, n_heads=4, dim_feedforward=256, num_layers=2, dropout=0.0, norm='AdaptiveLayerNorm')
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=5, rich=True)
    model = get_model_class('ValleNAR')(cfg)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    batch = synth.synth_nar_batch(cfg, 3, n_tokens=9, n_frames=48, seed=3)
    for stage in (2, 7):
        ref, _ = O.nar_stage_logits(sd, cfg, batch, stage)
        got, _ = model.stage_logits(batch, stage)
        close(got, ref)
    stage = 4
    used = [k for k in sd if not k.endswith('.pe')]
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    rl = O.nar_training_loss(params, cfg, batch, stage)
    rl.backward()
    loss = model.training_step(batch, stage=stage)
    torch.testing.assert_close(loss.detach().cpu(), rl.detach(), rtol=1e-5, atol=1e-6)
    loss.backward()
    _grad_check(model, params, sorted(k for k in used if params[k].grad is not None and params[k].grad.abs().sum() > 0))


def test_nar_generate_batch_any_head_dim_matches_the_oracle_per_utterance():
    """Ragged greedy NAR decoding (7 stage forwards per call) with head width 32 — the stack takes the general kernels,
    per-row key lengths included — token for token against the oracle on each utterance alone."""
    from oracle import valle_oracle as O
    from tests.test_nar_generate_gpu import _model, _utterances
    from valle2_amd import synth
    kw = dict(d_model=128, n_heads=4, dim_feedforward=256, num_layers=2, dropout=0.0, norm='AdaptiveLayerNorm')
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=43, rich=True)
    us = _utterances(cfg, [(11, 9, 17), (7, 21, 30), (16, 4, 6)], seed=9)
    outs = _model(kw, sd).generate_batch([u[0].to(DEV) for u in us], [u[1].to(DEV) for u in us],
                                         [u[2].to(DEV) for u in us], greedy=True)
    for (text, pc, first), got in zip(us, outs):
        ref = O.nar_generate(sd, cfg, text[:4], pc, text[4:], first, greedy=True)
        assert torch.equal(got.cpu(), ref), f'{(got.cpu() != ref).sum().item()} of {ref.numel()} tokens differ'


def test_head_dim_32_and_128_match_the_real_reference():
    """tests/golden/head_dim.npz was written by the REAL reference (gen_golden.py): MultiHeadAttention at head widths 32
    and 128 (causal, causal + padding, a cached step), and a 2-layer ValleAR with head width 32 — greedy generate with its
    margins, training loss and per-parameter gradient norms."""
    from tests.oracle_runners import load_golden
    from valle2_amd import get_model_class
    from valle2_amd.modules import MultiHeadAttention
    gold = load_golden('head_dim')
    for d, h, b, t in C.HD_MHA_SHAPES:
        sd, x, causal, pad = C.mha_inputs(d, h, b, t)
        m = MultiHeadAttention(d, h)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        o, (k, v) = m(x.to(DEV), attn_mask=causal.to(DEV), use_cache=True)
        o2, _ = m(x.to(DEV), attn_mask=causal.to(DEV), padding_mask=pad.to(DEV))
        o4, (k4, _) = m(C._randn((b, 1, d), 300 + d).to(DEV), kv_cache=(k, v), use_cache=True)
        for got, key in ((o, f'out_{d}'), (k, f'k_{d}'), (o2, f'out_pad_{d}'), (o4, f'out_step_{d}'), (k4, f'k_step_{d}')):
            close(got, gold[key])
    kw, sd, utt, batch = C.head_dim_inputs()
    model = get_model_class('ValleAR')(C.cfg_of(kw))
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    out = model.generate(*[u.to(DEV) for u in utt]).cpu()
    ref = gold['tokens']
    n = min(len(out), len(ref))
    bad = (out[:n] != ref[:n]).nonzero()
    assert len(out) == len(ref) and (bad.numel() == 0 or float(gold['margin'][int(bad[0])]) < 1e-4), (out, ref)
    loss = model.training_step({k: v.clone() for k, v in batch.items()})
    torch.testing.assert_close(loss.detach().cpu(), gold['loss'], rtol=1e-5, atol=1e-6)
    loss.backward()
    names = sorted(n for n, _ in model.named_parameters())
    norms = torch.stack([dict(model.named_parameters())[n].grad.norm().cpu() for n in names])
    torch.testing.assert_close(norms, gold['grad_norms'], rtol=1e-3, atol=1e-7)
