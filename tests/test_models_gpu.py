"""GPU parity of the drop-in modules / models: HIP path vs the golden vectors of the real reference
(tests/golden/*.npz) and vs the CPU oracle on fresh seeded inputs.

Stated tolerances (fp32 end to end; differences are summation order + exp/erf/rsqrt ulps):
  activations / K,V rows  atol 2e-4 + rtol 1e-4   (logits atol 2e-4, rtol 1e-4)
  loss                    rtol 1e-5
  greedy token ids        EXACT at every step whose reference top-1/top-2 margin is > 1e-4; if a
                          step diverges it must be such a near-tie (then later steps are not
                          compared).  In practice no step diverges.
"""
import numpy as np
import pytest
import torch

from tests.golden import cases as C
from tests.oracle_runners import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ATOL, RTOL = 2e-4, 1e-4
MARGIN = 1e-4


def close(a, b, atol=ATOL, rtol=RTOL, msg=None):
    torch.testing.assert_close(a.detach().cpu().float(), b.float(), atol=atol, rtol=rtol, msg=msg)


def tokens_match(got, gold_tokens, margins):
    got, gold_tokens = got.cpu(), gold_tokens.cpu()
    n = min(len(got), len(gold_tokens))
    neq = (got[:n] != gold_tokens[:n]).nonzero()
    if neq.numel() == 0:
        assert len(got) == len(gold_tokens)
        return
    first = int(neq[0])
    assert float(margins[first]) <= MARGIN, (
        f'greedy tokens diverge at step {first} where the reference margin is {float(margins[first]):.3e}')


def build(cls_name, kw, sd):
    from valle2_amd import get_model_class
    m = get_model_class(cls_name)(C.cfg_of(kw))
    m.load_state_dict(sd)
    return m.to(DEV).eval()


# ---- the reference's own tests, as the reference writes them: CPU module, CPU tensors ---------
@pytest.mark.parametrize('d_model,n_heads,batch_size,seq_len', C.MHA_SHAPES)
def test_reference_mha_shape_test_with_cpu_tensors(d_model, n_heads, batch_size, seq_len):
    """Restates /root/reference/tests/test_modules.py:7-30: the module is never moved to a device and
    the inputs are CPU tensors.  The arithmetic must still be the HIP kernels' (device mirror of the
    parameters, results copied back): outputs are CPU tensors of the reference's shapes, and equal
    the same module run on the device."""
    from valle.models.modules import MultiHeadAttention
    attention = MultiHeadAttention(d_model=d_model, n_heads=n_heads)
    head_dim = d_model // n_heads
    assert attention.head_dim == head_dim
    x = torch.randn(batch_size, seq_len, d_model)
    mask = torch.triu(torch.ones(seq_len, seq_len), diagonal=1)
    output, kv = attention(x, attn_mask=mask, use_cache=True)
    k, v = kv
    assert output.shape == (batch_size, seq_len, d_model)
    assert k.shape == (batch_size, n_heads, seq_len, head_dim)
    assert v.shape == (batch_size, n_heads, seq_len, head_dim)
    assert not output.is_cuda and not k.is_cuda and not v.is_cuda
    assert all(not p.is_cuda for p in attention.parameters()), 'the CPU module itself must not move'
    import copy
    dev = copy.deepcopy(attention).to(DEV)
    o2, (k2, v2) = dev(x.to(DEV), attn_mask=mask.to(DEV), use_cache=True)
    assert torch.equal(output, o2.cpu()) and torch.equal(k, k2.cpu()) and torch.equal(v, v2.cpu())
    # a cached step on the CPU-resident cache the first call returned (foreign cache → adopted)
    xn = torch.randn(batch_size, 1, d_model)
    o3, (k3, _) = attention(xn, kv_cache=(k, v), use_cache=True)
    assert o3.shape == (batch_size, 1, d_model) and k3.shape == (batch_size, n_heads, seq_len + 1, head_dim)
    assert torch.equal(k3[:, :, :seq_len], k)


def test_cpu_resident_modules_match_reference_golden():
    """The device round trip carries the reference's numbers: MHA and Transformer goldens with the
    module and every input on the CPU; the mirror follows a parameter update."""
    from valle2_amd.modules import MultiHeadAttention, Transformer
    from valle2_amd.utils import build_attn_mask
    gold = load_golden('mha')
    d, h, b, t = C.MHA_SHAPES[0]
    sd, x, causal, pad = C.mha_inputs(d, h, b, t)
    m = MultiHeadAttention(d, h).eval()
    m.load_state_dict(sd)
    o, (k, v) = m(x, attn_mask=causal, use_cache=True)
    close(o, gold[f'out_{d}']); close(k, gold[f'k_{d}'])
    o2, _ = m(x, attn_mask=causal, padding_mask=pad)
    close(o2, gold[f'out_pad_{d}'])
    with torch.no_grad():
        m.out.bias.add_(1.0)                      # in-place update → version bump → mirror rebuilt
    o3, _ = m(x, attn_mask=causal, use_cache=True)
    close(o3, gold[f'out_{d}'] + 1.0)
    tg = load_golden('transformer')
    kw, sd, x, xl, yl, padm, emb = C.transformer_inputs('AdaptiveLayerNorm')
    tr = Transformer(C.cfg_of(kw)).eval()
    tr.load_state_dict(sd)
    y, kv = tr(x, padding_mask=padm, attn_mask=build_attn_mask(xl, yl, 'cpu'), embedding=emb, use_cache=True)
    assert not y.is_cuda and len(kv) == kw['num_layers']
    close(y, tg['AdaptiveLayerNorm_y'])


def test_cpu_resident_model_generates_on_the_device():
    from oracle import valle_oracle as O
    from valle2_amd import get_model_class, synth
    kw = dict(C.AR_TINY, max_audio_len=12)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=99, rich=True), cfg)
    utt = synth.synth_utterance(cfg, 9, 11, 33, seed=4321)
    trace = {}
    ref = O.ar_generate(sd, cfg, *utt, trace=trace)
    m = get_model_class('ValleAR')(cfg).eval()
    m.load_state_dict(sd)
    out = m.generate(*utt)                        # model and inputs on the CPU
    assert not out.is_cuda and m.last_generate_stats['tokens_appended'] == 12
    tokens_match(out, ref, torch.tensor(trace['margin']))
    with pytest.raises(Exception, match='HIP device'), torch.enable_grad():
        m.training_step(synth.synth_ar_batch(cfg, 2, tok_range=(5, 9), code_range=(13, 20), seed=1))


def test_tagged_masks_take_the_analytic_kernel_path(monkeypatch):
    """build_attn_mask + build_pad_mask inputs must reach the kernel as MASK_PREFIX + per-row lengths
    (no u8 mask tensors), and give the explicit path's result."""
    from valle2_amd import kernels, modules
    from valle2_amd.utils import build_attn_mask, build_pad_mask
    d, h, b, xl, yl = 128, 2, 3, 5, 11
    m = modules.MultiHeadAttention(d, h).to(DEV).eval()
    x = torch.randn(b, xl + yl, d, device=DEV)
    lens = torch.tensor([16, 9, 12])
    am, pm = build_attn_mask(xl, yl, DEV), build_pad_mask(lens, DEV)
    seen = []
    real = kernels.attn_rows
    monkeypatch.setattr(kernels, 'attn_rows', lambda *a, **k: (seen.append(k), real(*a, **k))[1])
    o1, _ = m(x, attn_mask=am, padding_mask=pm)
    assert seen[-1]['mode'] == kernels.MASK_PREFIX and seen[-1]['x_len'] == xl
    assert seen[-1]['kv_len'].dtype == torch.int32 and seen[-1]['kv_len'].tolist() == lens.tolist()
    o2, _ = m(x, attn_mask=am.clone(), padding_mask=pm.clone())      # untagged copies → explicit masks
    assert seen[-1]['mode'] == kernels.MASK_EXPLICIT
    close(o1, o2.cpu(), atol=1e-6, rtol=1e-6)


def test_mha_golden():
    from valle2_amd.modules import MultiHeadAttention
    gold = load_golden('mha')
    for d, h, b, t in C.MHA_SHAPES:
        sd, x, causal, pad = C.mha_inputs(d, h, b, t)
        m = MultiHeadAttention(d, h)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        xd = x.to(DEV)
        o, (k, v) = m(xd, attn_mask=causal.to(DEV), use_cache=True)
        close(o, gold[f'out_{d}']); close(k, gold[f'k_{d}']); close(v, gold[f'v_{d}'])
        o2, _ = m(xd, attn_mask=causal.to(DEV), padding_mask=pad.to(DEV))
        close(o2, gold[f'out_pad_{d}'])
        o3, none = m(xd)
        assert none is None
        close(o3, gold[f'out_nomask_{d}'])
        xn = C._randn((b, 1, d), 300 + d).to(DEV)
        o4, (k4, v4) = m(xn, kv_cache=(k, v), use_cache=True)       # in-place append
        close(o4, gold[f'out_step_{d}']); close(k4, gold[f'k_step_{d}'])
        assert k4.shape == (b, h, t + 1, 64)
        # a foreign (untagged) cache is adopted
        o5, (k5, _) = m(xn, kv_cache=(gold[f'k_{d}'].to(DEV), gold[f'v_{d}'].to(DEV)), use_cache=True)
        close(o5, gold[f'out_step_{d}']); close(k5, gold[f'k_step_{d}'])


@pytest.mark.parametrize('norm', ['LayerNorm', 'AdaptiveLayerNorm'])
def test_transformer_golden(norm):
    from valle2_amd.modules import Transformer
    from valle2_amd.utils import build_attn_mask
    gold = load_golden('transformer')
    kw, sd, x, xl, yl, pad, emb = C.transformer_inputs(norm)
    m = Transformer(C.cfg_of(kw))
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    e = emb.to(DEV) if norm != 'LayerNorm' else None
    mask = build_attn_mask(xl, yl, DEV)
    xd = x.to(DEV)
    # analytic path (tagged prefix mask, untagged pad → explicit) and the explicit path must agree
    y, kv = m(xd, padding_mask=pad.to(DEV), attn_mask=mask, embedding=e, use_cache=True)
    close(y, gold[f'{norm}_y']); close(kv[0][0], gold[f'{norm}_k0'])
    y2, _ = m(xd, padding_mask=pad.to(DEV), attn_mask=mask.clone(), embedding=e)   # tag dropped
    close(y2, gold[f'{norm}_y'])
    yfull, empty = m(xd, embedding=e)
    assert empty == ()
    close(yfull, gold[f'{norm}_yfull'])
    xn = torch.cat([xd, C._randn((x.shape[0], 1, x.shape[2]), 19).to(DEV)], dim=1)
    ystep, kv2 = m(xn, attn_mask=mask, embedding=e, kv_cache=kv, use_cache=True)
    assert ystep.shape[1] == 1
    close(ystep, gold[f'{norm}_ystep']); close(kv2[-1][1], gold[f'{norm}_vlast'])
    assert torch.equal(xd.cpu(), x), 'inputs must not be modified'


def test_transformer_small_batch_python_path_matches_native():
    # rows <= 64 take the per-layer Python path with LN fused in the skinny GEMMs; rows > 64 take
    # the native composite: both must give the same numbers
    from valle2_amd.modules import Transformer
    kw, sd, x, xl, yl, pad, emb = C.transformer_inputs('LayerNorm')
    m = Transformer(C.cfg_of(kw))
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    xd = x.to(DEV)                                   # 3 x 24 = 72 rows → native
    y_native, _ = m(xd)
    y_py = torch.cat([m(xd[i:i + 1])[0] for i in range(3)])          # 24 rows each → python path
    close(y_py, y_native.cpu(), atol=5e-5, rtol=5e-5)


def test_ar_training_forward_golden():
    gold = load_golden('ar_train')
    kw, sd, batch = C.ar_train_inputs()
    m = build('ValleAR', kw, sd)
    loss = m.training_step({k: v.clone() for k, v in batch.items()})
    torch.testing.assert_close(loss.cpu(), gold['loss'], rtol=1e-5, atol=1e-6)
    from oracle import valle_oracle as O
    ref_logits = O.ar_logits(sd, C.cfg_of(kw), batch).permute(0, 2, 1)
    close(m.forward_logits(batch), ref_logits)


@pytest.mark.parametrize('which', ['tiny', 'mid'])
@pytest.mark.parametrize('graph', [True, False])
def test_ar_generate_golden(which, graph):
    gold = load_golden(f'ar_generate_{which}')
    kw, sd, utt = C.ar_generate_inputs(which)
    m = build('ValleAR', kw, sd)
    if graph:
        out = m.generate(*[u.to(DEV) for u in utt])
    else:
        beams = m.config.num_beams
        text = torch.cat(utt[0::2])
        rows = m.generate_batch([text] * beams, [utt[1][:, 0]] * beams, use_graph=False)
        out = rows[0, utt[1].shape[0] + 1:]
    assert out.dtype == torch.int64 and out.dim() == 1
    tokens_match(out, gold['tokens'], gold['margin'])


@pytest.mark.parametrize('fold,fused', [(False, False), (True, False), (True, True)])
def test_ar_generate_golden_every_decode_engine(fold, fused):
    """The decode step has three forms of its GEMM chain: LayerNorm in the operand load + linear_1 + split-K linear_2
    + reduce; the same with the LayerNorm folded into the weights; folded + the FeedForward as one launch split over
    dim_feedforward (vh_ffn_decode, the default).  Each must reproduce the reference's greedy tokens."""
    from valle2_amd import _lib, engine
    old = engine.FOLD_LAYERNORM
    engine.FOLD_LAYERNORM = fold
    _lib.lib().vh_set_tuning(5, 0 if fused else 1)
    try:
        for which in ('tiny', 'mid'):
            gold = load_golden(f'ar_generate_{which}')
            kw, sd, utt = C.ar_generate_inputs(which)
            m = build('ValleAR', kw, sd)
            out = m.generate(*[u.to(DEV) for u in utt])
            tokens_match(out, gold['tokens'], gold['margin'])
            assert m.last_generate_stats['ffn_fused'] == fold      # the workspace exists whenever the weights are folded
    finally:
        engine.FOLD_LAYERNORM = old
        _lib.lib().vh_set_tuning(5, 0)


@pytest.mark.parametrize('graph', [True, False])
def test_head_and_greedy_step_in_one_launch_same_tokens(monkeypatch, graph):
    """VALLE2_HEAD_FUSED=1 (vh_head_greedy: the head's 16-column workgroups publish their candidates, the last arriver does
    the greedy step): the reference's greedy tokens on every ar_generate_* golden (shared prompt), and
    ragged independent rows equal to the two-launch form token for token."""
    from valle2_amd import synth
    monkeypatch.setenv('VALLE2_HEAD_FUSED', '1')
    for which in ('tiny', 'mid'):
        gold = load_golden(f'ar_generate_{which}')
        kw, sd, utt = C.ar_generate_inputs(which)
        m = build('ValleAR', kw, sd)
        if graph:
            out = m.generate(*[u.to(DEV) for u in utt])
        else:
            beams = m.config.num_beams
            rows = m.generate_batch([torch.cat(utt[0::2])] * beams, [utt[1][:, 0]] * beams, use_graph=False)
            out = rows[0, utt[1].shape[0] + 1:]
        assert m.last_generate_stats['head_fused']
        tokens_match(out, gold['tokens'], gold['margin'])
    gold = load_golden('ar_generate_eos')
    kw, sd, utt = C.ar_eos_inputs(gold['eos_row'])
    m = build('ValleAR', kw, sd)
    assert torch.equal(m.generate(*[u.to(DEV) for u in utt]).cpu(), gold['tokens'])
    assert m.last_generate_stats['head_fused'] and m.last_generate_stats['tokens_appended'] == int(gold['steps']) - 1
    kw = dict(d_model=256, n_heads=4, dim_feedforward=1024, num_layers=3, dropout=0.0, norm='LayerNorm', num_beams=1, top_k=1,
              max_audio_len=24)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=17, rich=True, std=0.15), cfg)
    utts = [synth.synth_utterance(cfg, 5 + r % 4, 6 + r % 5, 21, seed=950 + r) for r in range(40)]
    m = build('ValleAR', kw, sd)
    texts = [torch.cat([u[0], u[2]]).to(DEV) for u in utts]
    prompts = [u[1][:, 0].to(DEV) for u in utts]
    fused = m.generate_batch(texts, prompts, use_graph=graph)
    assert m.last_generate_stats['head_fused']
    monkeypatch.setenv('VALLE2_HEAD_FUSED', '0')
    plain = m.generate_batch(texts, prompts, use_graph=graph)
    assert not m.last_generate_stats['head_fused']
    assert torch.equal(fused, plain)


@pytest.mark.parametrize('rows', [5, 40])
def test_fused_feedforward_same_tokens_graph_and_eager(rows):
    """vh_ffn_decode against the three-launch FeedForward inside the decoder: ragged rows, graph replay and eager
    stepping, more rows than one row group (40): the greedy tokens of a 40-step generate must be the same."""
    from valle2_amd import _lib, synth
    kw = dict(C.MID, norm='LayerNorm', num_beams=rows, top_k=1, max_audio_len=40)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=3, rich=True), cfg)
    m = build('ValleAR', kw, sd)
    g = torch.Generator().manual_seed(8)
    tl = [(30, 17, 44, 30, 9)[i % 5] + i // 5 for i in range(rows)]
    pl = [(50, 61, 20, 50, 33)[i % 5] + i // 5 for i in range(rows)]
    texts = [torch.randint(0, 256, (n,), generator=g).to(DEV) for n in tl]
    firsts = [torch.randint(0, 1024, (n,), generator=g).to(DEV) for n in pl]
    outs = {}
    try:
        for unfused in (0, 1):
            _lib.lib().vh_set_tuning(5, unfused)
            for graph in (True, False):
                outs[(unfused, graph)] = m.generate_batch(texts, firsts, use_graph=graph)
    finally:
        _lib.lib().vh_set_tuning(5, 0)
    assert torch.equal(outs[(0, True)], outs[(0, False)]), 'fused FeedForward: graph replay differs from eager steps'
    assert torch.equal(outs[(1, True)], outs[(1, False)])
    # the two forms sum in different orders: equal tokens wherever the decision is not a rounding-level tie
    same = (outs[(0, True)] == outs[(1, True)]).float().mean().item()
    assert same == 1.0, f'fused and three-launch FeedForward agree on {same:.4f} of the tokens'


def test_generate_batch_distinct_rows_vs_oracle():
    """Rows are independent utterances: each row of generate_batch must equal the oracle run on
    that utterance alone (a kernel that mixed rows or read row 0 for everyone would pass the
    identical-beam fixtures)."""
    from oracle import valle_oracle as O
    from valle2_amd import synth
    kw = dict(d_model=256, n_heads=4, dim_feedforward=1024, num_layers=3, dropout=0.0,
              norm='LayerNorm', num_beams=1, top_k=1, max_audio_len=24)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=17, rich=True, std=0.15), cfg)
    utts = [synth.synth_utterance(cfg, 7, 9, 21, seed=900 + r) for r in range(5)]
    m = build('ValleAR', kw, sd)
    rows = m.generate_batch([torch.cat([u[0], u[2]]).to(DEV) for u in utts],
                            [u[1][:, 0].to(DEV) for u in utts])
    outs = set()
    for r, u in enumerate(utts):
        trace = {}
        ref = O.ar_generate(sd, cfg, *u, trace=trace)
        tokens_match(rows[r, 22:], ref, torch.tensor(trace['margin']))
        outs.add(tuple(ref.tolist()))
    assert len(outs) > 1, 'test inputs must lead to different continuations'


def test_ar_generate_eos_golden():
    gold = load_golden('ar_generate_eos')
    kw, sd, utt = C.ar_eos_inputs(gold['eos_row'])
    m = build('ValleAR', kw, sd)
    out = m.generate(*[u.to(DEV) for u in utt])
    assert torch.equal(out.cpu(), gold['tokens'])
    assert m.last_generate_stats['tokens_appended'] == int(gold['steps']) - 1


def test_ar_generate_rejects_what_the_reference_cannot_do():
    kw, sd, utt = C.ar_generate_inputs('tiny')
    with pytest.raises(AssertionError):
        build('ValleAR', kw, sd).generate(utt[0].unsqueeze(0).to(DEV), utt[1].to(DEV))


def test_ar_generate_without_kv_cache_recomputes_every_step_and_gives_the_same_tokens():
    """config.use_kv_cache = False raises inside the reference (D2: valle_ar.py:150-155).  Here the flag does what it
    says — every step runs the whole stack over the whole sequence, nothing but the tokens is carried over — which
    makes it an independent check of the in-place KV cache: same tokens as the cached decoder and as the reference's
    golden run, for identical beams, an EOS-terminated run and a ragged batch."""
    gold = load_golden('ar_generate_tiny')
    kw, sd, utt = C.ar_generate_inputs('tiny')
    dev_utt = [u.to(DEV) for u in utt]
    cached = build('ValleAR', kw, sd).generate(*dev_utt)
    m = build('ValleAR', dict(kw, use_kv_cache=False), sd)
    out = m.generate(*dev_utt)
    assert torch.equal(out, cached)
    tokens_match(out, gold['tokens'], gold['margin'])
    with pytest.raises(ValueError):
        m.generate_batch([torch.cat([utt[0], utt[2]]).to(DEV)], [utt[1][:, 0].to(DEV)], perf_mode=True)
    # EOS-terminated run
    gold = load_golden('ar_generate_eos')
    kw, sd, utt = C.ar_eos_inputs(gold['eos_row'])
    out = build('ValleAR', dict(kw, use_kv_cache=False), sd).generate(*[u.to(DEV) for u in utt])
    assert torch.equal(out.cpu(), gold['tokens'])
    # ragged rows: per-row lengths through the prefix mask at every recomputed step
    from valle2_amd import synth
    kw, sd, _ = C.ar_generate_inputs('tiny')
    cfg = C.cfg_of(kw)
    utts = [synth.synth_utterance(cfg, 5 + 3 * i, 4 + i, 9 + 5 * i, seed=40 + i) for i in range(3)]
    texts = [torch.cat([u[0], u[2]]).to(DEV) for u in utts]
    firsts = [u[1][:, 0].to(DEV) for u in utts]
    a = build('ValleAR', dict(kw, max_audio_len=12), sd).generate_batch(texts, firsts)
    b = build('ValleAR', dict(kw, max_audio_len=12, use_kv_cache=False), sd).generate_batch(texts, firsts)
    assert torch.equal(a, b)


def test_nar_golden():
    gold = load_golden('nar')
    kw, sd, batch = C.nar_inputs()
    m = build('ValleNAR', kw, sd)
    for stage in (1, 4, 7):
        y, p = m._prepare_audio_codes(batch['codes'], stage)
        assert p == int(gold[f'prefix_{stage}'])
        assert torch.equal(y.cpu(), gold[f'prep_{stage}'])          # pure adds: bit-exact
        logits, p2 = m.stage_logits(batch, stage)
        assert p2 == p
        close(logits, gold[f'logits_{stage}'])


def test_nar_generate_matches_oracle_greedy():
    from oracle import valle_oracle as O
    kw, sd, (pt, pc, tt, first) = C.nar_generate_inputs()
    ref = O.nar_generate(sd, C.cfg_of(kw), pt, pc, tt, first, greedy=True)
    m = build('ValleNAR', kw, sd)
    out = m.generate(pt.to(DEV), pc.to(DEV), tt.to(DEV), first.to(DEV), greedy=True)
    assert out.shape == ref.shape == (20, 8)
    assert torch.equal(out[:, 0].cpu(), first)
    # stage n+1 consumes stage n's tokens, so compare stage by stage until a near-tie could differ
    agree = (out.cpu() == ref).float().mean().item()
    assert agree == 1.0, f'NAR greedy tokens agree on {agree:.3f} of entries'


def test_fresh_inputs_vs_oracle_mid_size():
    """Not a fixture: a 4-layer/256-d AR model on new seeds, HIP generate vs oracle generate."""
    from oracle import valle_oracle as O
    from valle2_amd import synth
    kw = dict(d_model=256, n_heads=4, dim_feedforward=1024, num_layers=4, dropout=0.0,
              norm='LayerNorm', num_beams=3, top_k=1, max_audio_len=40)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=99, rich=True), cfg)
    utt = synth.synth_utterance(cfg, 17, 30, 101, seed=4321)
    trace = {}
    ref = O.ar_generate(sd, cfg, *utt, trace=trace)
    m = build('ValleAR', kw, sd)
    out = m.generate(*[u.to(DEV) for u in utt])
    tokens_match(out, ref, torch.tensor(trace['margin']))


def test_module_level_errors_are_loud():
    from valle2_amd import _lib
    from valle2_amd.modules import FeedForward, MultiHeadAttention
    with pytest.raises(_lib.VhError):
        MultiHeadAttention(10, 5).to(DEV)(torch.randn(1, 3, 10, device=DEV))     # head_dim 2: not a multiple of 4
    with pytest.raises(_lib.VhError):
        MultiHeadAttention(10, 5)(torch.randn(1, 3, 10))                         # same, through the mirror
    y, _ = MultiHeadAttention(128, 4)(torch.randn(1, 3, 128))                    # head_dim 32: the general kernels (test_head_dim_gpu.py)
    assert not y.is_cuda and y.shape == (1, 3, 128)
    y = FeedForward(128, 512).to(DEV)(torch.randn(1, 3, 128))                    # CPU input, device module
    assert not y.is_cuda and y.shape == (1, 3, 128)


def test_generate_batch_ragged_rows_vs_oracle():
    """Rows of different text / prompt lengths in one batch (per-row prefix-LM mask, positions,
    cache lengths): every row must equal the oracle run on that utterance alone."""
    from oracle import valle_oracle as O
    from valle2_amd import synth
    kw = dict(d_model=256, n_heads=4, dim_feedforward=1024, num_layers=3, dropout=0.0,
              norm='LayerNorm', num_beams=1, top_k=1, max_audio_len=20)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=23, rich=True, std=0.15), cfg)
    shapes = [(5, 6, 40), (9, 3, 17), (2, 2, 140), (30, 20, 33)]       # (prompt text, target text, frames)
    utts = [synth.synth_utterance(cfg, a, b, f, seed=700 + i) for i, (a, b, f) in enumerate(shapes)]
    m = build('ValleAR', kw, sd)
    rows = m.generate_batch([torch.cat([u[0], u[2]]).to(DEV) for u in utts], [u[1][:, 0].to(DEV) for u in utts])
    starts = m.last_generate_stats['prompt_lens']
    assert starts == [f + 1 for _, _, f in shapes] and rows.shape == (4, 141 + 20)
    for r, u in enumerate(utts):
        trace = {}
        ref = O.ar_generate(sd, cfg, *u, trace=trace)
        tokens_match(rows[r, starts[r]: starts[r] + 20], ref, torch.tensor(trace['margin']))
        assert torch.equal(rows[r, 1:starts[r]].cpu(), u[1][:, 0]) and int(rows[r, 0]) == cfg.bos_token


def test_generate_batch_beyond_64_rows_runs_in_groups():
    """More rows than one decode launch serves (64): consecutive groups, every row equal to itself in a small batch."""
    from valle2_amd import synth
    kw = dict(C.AR_TINY, max_audio_len=10)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=5, rich=True), cfg)
    m = build('ValleAR', kw, sd)
    g = torch.Generator().manual_seed(1)
    texts = [torch.randint(0, 256, (8 + i % 5,), generator=g).to(DEV) for i in range(70)]
    firsts = [torch.randint(0, 1024, (12 + i % 7,), generator=g).to(DEV) for i in range(70)]
    out = m.generate_batch(texts, firsts)
    starts = m.last_generate_stats['prompt_lens']
    assert out.shape[0] == 70 and len(starts) == 70 and m.last_generate_stats['sum_logprobs'].shape == (70,)
    for r in (0, 63, 64, 69):
        alone = m.generate_batch([texts[r]], [firsts[r]])
        assert torch.equal(out[r, starts[r]:starts[r] + 10], alone[0, starts[r]:starts[r] + 10]), r


def test_host_logits_and_module_level_positional_encoding_run_on_the_device():
    """`topk_sampling` on CPU logits hops to the device like every module does (a reference caller holding
    CPU logits keeps working: valle/models/utils.py:46-68) and returns host tensors; `PositionalEncoding.forward` as a module
    call is a library kernel (vh_add_pe), bit-identical to the fp32 add it replaces (valle/models/modules.py:78-80)."""
    from valle2_amd.modules import PositionalEncoding
    from valle2_amd.utils import topk_sampling
    gen = torch.Generator().manual_seed(3)
    logits = 3 * torch.randn(5, 1025, generator=gen)
    tok, lp = topk_sampling(logits, top_k=1)
    assert not tok.is_cuda and not lp.is_cuda and tuple(tok.shape) == (5, 1) and tok.dtype == torch.int64
    assert torch.equal(tok[:, 0], logits.argmax(-1)) and float(lp.abs().max()) == 0.0
    t_dev, lp_dev = topk_sampling(logits.to(DEV), top_k=20, tok_p=0.9, temperature=0.8, seed=11)
    t_cpu, lp_cpu = topk_sampling(logits, top_k=20, tok_p=0.9, temperature=0.8, seed=11)
    assert torch.equal(t_dev.cpu(), t_cpu) and torch.equal(lp_dev.cpu(), lp_cpu)
    pe = PositionalEncoding(128, dropout=0.0).eval()
    x = torch.randn(3, 37, 128, generator=gen)
    want = x + pe.pe[:37, 0]
    assert torch.equal(pe(x), want)                                     # CPU module, CPU input: through the device mirror
    assert torch.equal(pe.to(DEV)(x.to(DEV)).cpu(), want)
    from valle2_amd._lib import VhError
    with pytest.raises(VhError, match='max_len'):
        PositionalEncoding(128, max_len=16).to(DEV)(x.to(DEV))


@pytest.mark.parametrize('which', ['tiny', 'mid', 'eos'])
def test_generate_with_a_shared_prompt_matches_the_reference_and_the_independent_rows(which):
    """generate() replicates one utterance over its beams (valle_ar.py:135-138), so the prompt pass runs
    for ONE row and every decode step reads the prompt's K/V once for all beams (vh_attn_decode_shared).  Tokens: the REAL
    reference's (goldens), and the same as decoding the beams as independent rows; graph and eager; EOS stop included."""
    gold = load_golden({'tiny': 'ar_generate_tiny', 'mid': 'ar_generate_mid', 'eos': 'ar_generate_eos'}[which])
    kw, sd, utt = C.ar_generate_inputs(which) if which != 'eos' else C.ar_eos_inputs(gold['eos_row'])
    m = build('ValleAR', kw, sd)
    beams = kw['num_beams']
    text = torch.cat([utt[0], utt[2]]).to(DEV)
    first = utt[1][:, 0].to(DEV)
    outs = {}
    for shared in (True, False):
        for graph in (True, False):
            rows = m.generate_batch([text] * beams, [first] * beams, shared_prompt=shared, use_graph=graph)
            assert m.last_generate_stats['shared_prompt'] == shared
            outs[shared, graph] = rows.cpu()
    n = utt[1].shape[0] + 1
    for key, rows in outs.items():
        assert bool((rows == rows[:1]).all()), f'{key}: identical beams must decode identical tokens under top_k = 1'
        got = rows[0, n:]
        got = got[got != m.eos_token]
        if which == 'eos':
            assert torch.equal(got, gold['tokens']), key
        else:
            tokens_match(got, gold['tokens'], gold['margin'])
    out = m.generate(*[u.to(DEV) for u in utt])                          # the reference's own entry point: shared by default
    assert m.last_generate_stats['shared_prompt']
    if which == 'eos':
        assert torch.equal(out.cpu(), gold['tokens'])
    else:
        tokens_match(out, gold['tokens'], gold['margin'])
    with pytest.raises(ValueError, match='same text and prompt'):
        m.generate_batch([text, text.flip(0)], [first, first], shared_prompt=True)


def test_shared_prompt_sampling_beams_diverge_and_scores_are_per_beam():
    """Default sampling (top_k = 50): beams share the prompt's K/V but draw their own tokens — rows differ, every row's
    sum of log-probabilities is its own, and a fixed torch seed reproduces the run."""
    kw = dict(C.AR_TINY, top_k=50, num_beams=4, max_audio_len=24)
    _, sd, utt = C.ar_generate_inputs('tiny')
    m = build('ValleAR', kw, sd)
    text = torch.cat([utt[0], utt[2]]).to(DEV)
    first = utt[1][:, 0].to(DEV)
    torch.manual_seed(5)
    a = m.generate_batch([text] * 4, [first] * 4, shared_prompt=True)
    lp_a = m.last_generate_stats['sum_logprobs'].clone()
    torch.manual_seed(5)
    b = m.generate_batch([text] * 4, [first] * 4, shared_prompt=True)
    assert torch.equal(a, b) and torch.equal(lp_a, m.last_generate_stats['sum_logprobs'])
    assert len({tuple(r.tolist()) for r in a.cpu()}) > 1, 'sampled beams must not be copies of each other'
    assert len(set(lp_a.cpu().tolist())) > 1 and bool((lp_a < 0).all())


def test_derived_weights_live_beside_the_module_and_can_be_invalidated():
    """Round-4 advisor finding: the AdaLN table / folded-LayerNorm caches sat in `transformer.__dict__` (weak references:
    `torch.save(model)` failed after the first generate) and nothing told them about writes through `p.data`.  They live
    in a WeakKeyDictionary beside the module now; `engine.invalidate_derived()` / `bump_weights_epoch()` is the hook for
    out-of-band weight writes (an EMA swap by `p.data.copy_`)."""
    import io
    import pickle
    from valle2_amd import engine
    kw, sd, batch = C.nar_inputs()
    m = build('ValleNAR', kw, sd)
    with torch.no_grad():
        a, _ = m.stage_logits(batch, 3)
    assert '_vh_ada' not in m.transformer.__dict__ and '_vh_folded' not in m.transformer.__dict__
    pickle.dumps(m)                                                   # (weak references in the module would raise here)
    buf = io.BytesIO()
    torch.save(m, buf)
    assert m.transformer in engine._DERIVED and 'ada' in engine._DERIVED[m.transformer]
    # an out-of-band write: p.data moves neither the tensor's identity nor its version counter
    proj = m.transformer.layers[0].norm1.project_layer
    v0 = proj.weight._version
    proj.weight.data.mul_(1.5)
    assert proj.weight._version == v0
    with torch.no_grad():
        stale, _ = m.stage_logits(batch, 3)
        engine.invalidate_derived(m.transformer)
        fresh, _ = m.stage_logits(batch, 3)
    assert torch.equal(stale, a), 'the cache cannot see a p.data write (documented): same table, same logits'
    assert not torch.equal(fresh, a), 'after invalidate_derived the table is rebuilt from the new weights'
    ref = build('ValleNAR', kw, {k: (v * 1.5 if k == 'transformer.layers.0.norm1.project_layer.weight' else v) for k, v in sd.items()})
    with torch.no_grad():
        want, _ = ref.stage_logits(batch, 3)
    assert torch.equal(fresh, want)
    del m
    import gc
    gc.collect()
    assert all(k is not None for k in list(engine._DERIVED.keys()))     # entries die with their modules
