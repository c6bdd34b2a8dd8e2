"""Parity at BASELINE.json's FULL sizes through size-independent properties (the CPU oracle would
need minutes per case there):

  configs[1]  12L/512d AR, 32 rows, 1024-token prompt:  graph replay == eager launches (token for
              token); run-to-run determinism; identical rows of a batch decode identically; a row's
              tokens do not depend on which other rows share its batch; prefill logits of a row in a
              batch of 32 == the same row alone.
  configs[2]  12L/512d NAR, 64 x 1024: stage logits are batch-invariant (row i of 64 == row i alone)
              and invariant to right-padding the batch with extra frames' worth of rows.
  configs[4]  24L/1024d, 16 heads (B*h = 128 → split-KV decode + combine kernel): oracle parity on a
              short prompt, then graph == eager over the config's whole 2250 new tokens (context 626 → 2876); the NAR
              stack at the bench's batch of 8: row 0 == the B = 1 golden of the real reference; and the joint pipeline
              AR generate_batch → NAR generate_batch on 8 utterances, its stages re-derived by stage_logits.
"""
import pytest
import torch

from tests.golden import cases as C

pytestmark = pytest.mark.gpu
DEV = 'cuda'
AR12 = dict(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm',
            top_k=1, use_kv_cache=True)


def build(name, kw, seed=0, rich=True, std=0.02):
    from valle2_amd import get_model_class, synth
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, name, seed=seed, rich=rich, std=std)
    if name == 'ValleAR':
        synth.silence_eos(sd, cfg)
    m = get_model_class(name)(cfg)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), cfg, sd


def utterances(cfg, n, text, frames, seed0):
    from valle2_amd import synth
    us = [synth.synth_utterance(cfg, text // 2, text - text // 2, frames, seed=seed0 + i) for i in range(n)]
    return [torch.cat([u[0], u[2]]).to(DEV) for u in us], [u[1][:, 0].to(DEV) for u in us]


def test_config2_full_size_properties():
    # all 512 new tokens of configs[1] (context 1024 -> 1536), as the bench decodes them
    m, cfg, _ = build('ValleAR', dict(AR12, num_beams=32, max_audio_len=512), std=0.05)
    texts, firsts = utterances(cfg, 16, 256, 767, 4000)
    texts, firsts = texts + texts, firsts + firsts               # rows 16..31 repeat rows 0..15
    a = m.generate_batch(texts, firsts, max_new=512)
    b = m.generate_batch(texts, firsts, max_new=512)
    e = m.generate_batch(texts, firsts, max_new=512, use_graph=False)
    assert a.shape == (32, 768 + 512)
    assert int((a[:, 768:] == cfg.eos_token).sum()) == 0
    assert torch.equal(a, b), 'two graph runs differ: the decode step is not deterministic'
    assert torch.equal(a, e), 'graph replay and eager launches disagree'
    assert torch.equal(a[:16], a[16:]), 'identical rows of one batch decoded differently'
    assert len({tuple(r.tolist()) for r in a[:16, 768:].cpu()}) > 1, 'distinct utterances expected to differ'
    # batch invariance: 4 of the rows alone in a batch of 4 (different kernel shapes: MT=1, split-KV)
    sub = m.generate_batch(texts[:4], firsts[:4], max_new=96)
    same = (sub[:, 768:] == a[:4, 768:768 + 96])
    first_diff = (~same).float().argmax(dim=1)
    assert bool(same.all()) or bool((first_diff[~same.all(dim=1)] > 8).all()), \
        'a row diverges from its batched self within the first steps'


def test_config2_prefill_logits_batch_invariant():
    from valle2_amd import synth
    m, cfg, _ = build('ValleAR', dict(AR12, num_beams=32, max_audio_len=8))
    batch = synth.synth_ar_batch(cfg, 32, tok_range=(200, 256), code_range=(700, 767), seed=9)
    full = m.forward_logits(batch)                               # (32, Ty, 1025) teacher-forced
    one = {k: v[5:6] for k, v in batch.items()}
    tx, ty = int(batch['tokens_lens'].max()), int(batch['codes_lens'].max())
    # same padded geometry for the single row, so the positions line up
    alone = m.forward_logits({'tokens': one['tokens'], 'codes': one['codes'], 'target': one['target'],
                              'tokens_lens': torch.tensor([tx]), 'codes_lens': batch['codes_lens'][5:6]})
    n = int(batch['codes_lens'][5])
    torch.testing.assert_close(alone[0, :n], full[5, :n], atol=2e-4, rtol=1e-4)
    assert alone.shape[1] <= ty


def test_config3_nar_stage_batch_invariant():
    from valle2_amd import synth
    m, cfg, _ = build('ValleNAR', dict(AR12, norm='AdaptiveLayerNorm'), seed=2)
    batch = synth.synth_nar_batch(cfg, 64, 256, 768, seed=11)
    logits, p = m.stage_logits(batch, 4)
    assert p == 150 and logits.shape == (64, 768 - 150, 1024)       # prefix = min(768//3, 3*50)
    assert bool(torch.isfinite(logits).all())
    few = {k: v[40:42] for k, v in batch.items()}
    sub, _ = m.stage_logits(few, 4)
    torch.testing.assert_close(sub, logits[40:42], atol=2e-4, rtol=1e-4)
    # a different stage really changes the answer (AdaLN conditioning + fewer codebooks summed)
    other, _ = m.stage_logits(few, 1)
    assert float((other - sub).abs().max()) > 1e-3


def test_config5_shape_24_layers_1024d():
    from oracle import valle_oracle as O
    from valle2_amd import synth
    kw = dict(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0,
              norm='LayerNorm', top_k=1, num_beams=8, max_audio_len=12)
    m, cfg, sd = build('ValleAR', kw, seed=3, std=0.03)
    utt = synth.synth_utterance(cfg, 20, 20, 30, seed=77)
    trace = {}
    ref = O.ar_generate(sd, cfg, *utt, trace=trace)                 # CPU oracle on the short case
    out = m.generate(*[u.to(DEV) for u in utt]).cpu()
    st = m.last_generate_stats                                       # generate(): the 8 beams share the prompt's K/V (round 5)
    assert st['shared_prompt'] and st['n_split'] == 4                # 8 x 16 (beam, head) pairs -> 4 key splits of the beams' rows
    n = min(len(out), len(ref))
    bad = (out[:n] != ref[:n]).nonzero()
    assert len(out) == len(ref) and (bad.numel() == 0 or trace['margin'][int(bad[0])] < 1e-4)
    # the same beams as 8 independent rows: 8 x 16 heads -> split-KV decode attention + combine
    text, first = torch.cat([utt[0], utt[2]]).to(DEV), utt[1][:, 0].to(DEV)
    rows = m.generate_batch([text] * 8, [first] * 8).cpu()
    assert not m.last_generate_stats['shared_prompt'] and m.last_generate_stats['n_split'] == 2
    got = rows[0, 31:]
    got = got[got != cfg.eos_token]
    bad = (got[:n] != ref[:n]).nonzero()
    assert len(got) == len(ref) and (bad.numel() == 0 or trace['margin'][int(bad[0])] < 1e-4)
    # configs[4] as worded: 400 text + BOS + 225 prompt frames + 2250 new tokens → context 626 → 2876 (PE table 5000)
    texts, firsts = utterances(cfg, 8, 400, 225, 8000)
    g = m.generate_batch(texts, firsts, max_new=2250)
    assert m.last_generate_stats['tokens_appended'] == 2250
    e = m.generate_batch(texts, firsts, max_new=2250, use_graph=False)
    assert g.shape == (8, 226 + 2250) and torch.equal(g, e)
    assert int((g[:, 226:] == cfg.eos_token).sum()) == 0            # (EOS silenced: every row decodes all 2250)
    assert len({tuple(r.tolist()) for r in g[:, 226:].cpu()}) == 8


def test_config5_nar_batch_of_8_row0_matches_the_b1_golden():
    """configs[4]'s NAR leg at the shape bench.py runs it (B = 8 x 2875 positions, 24L/1024d): row 0 of the batch is
    the utterance of the real reference's B = 1 golden (nar_big.npz), 7 other utterances beside it — batch invariance
    ties the bench shape to the golden (logits atol 1e-3, rtol 1e-4 as the B = 1 test)."""
    from tests.oracle_runners import load_golden
    from valle2_amd import get_model_class, synth
    gold = load_golden('nar_big')
    kw, sd, one = C.nar_big_inputs()
    cfg = C.cfg_of(kw)
    others = synth.synth_nar_batch(cfg, 7, n_tokens=C.NAR_BIG_TEXT, n_frames=C.NAR_BIG_FRAMES, seed=999)
    batch = {k: torch.cat([one[k], others[k]]) for k in one}
    assert batch['codes'].shape == (8, C.NAR_BIG_FRAMES, 8) and batch['tokens'].shape == (8, C.NAR_BIG_TEXT)
    m = get_model_class('ValleNAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    for stage in (2, 7):
        with torch.no_grad():
            logits, p = m.stage_logits(batch, stage)
        assert p == int(gold[f'prefix_{stage}']) and logits.shape[0] == 8
        torch.testing.assert_close(logits[:1, ::C.NAR_BIG_STRIDE].cpu(), gold[f'logits_{stage}'], atol=1e-3, rtol=1e-4)
        assert float((logits[1] - logits[0]).abs().max()) > 1e-2     # the other rows are other utterances


def test_config5_joint_ar_then_nar_pipeline():
    """configs[4] end to end on 8 utterances: ValleAR.generate_batch (2250 first-codebook tokens per row) feeds
    ValleNAR.generate_batch (greedy) → (2250, 8) codes per row whose column 0 IS the AR output.  Every NAR stage is then
    re-derived by `stage_logits` on the finished codes: with a 150-frame prompt the training-shaped forward's prefix
    (min(T // 3, 3 x 50) = 150 frames, all codebooks) is exactly the acoustic prompt, so stage n's arg-max over the
    target frames must be codebook n of the result (near-ties < 1e-3 excepted: ragged-batch vs dense kernels)."""
    big = dict(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0, top_k=1)
    ar, cfg, _ = build('ValleAR', dict(big, norm='LayerNorm', num_beams=8, max_audio_len=2250), seed=3, std=0.03)
    nar, ncfg, _ = build('ValleNAR', dict(big, norm='AdaptiveLayerNorm'), seed=4, std=0.03)
    from valle2_amd import synth
    B, n_text, n_prompt, n_new = 8, 400, 150, 2250
    us = [synth.synth_utterance(cfg, n_text // 2, n_text // 2, n_prompt, seed=8100 + i) for i in range(B)]
    texts = [torch.cat([u[0], u[2]]).to(DEV) for u in us]
    prompts = [u[1].to(DEV) for u in us]                              # (150, 8) codes
    rows = ar.generate_batch(texts, [p[:, 0] for p in prompts], max_new=n_new)
    assert rows.shape == (B, 1 + n_prompt + n_new) and ar.last_generate_stats['tokens_appended'] == n_new
    firsts = [rows[b, 1 + n_prompt:] for b in range(B)]
    assert all(int((f >= ncfg.num_audio_tokens).sum()) == 0 for f in firsts)
    out = nar.generate_batch(texts, prompts, firsts, greedy=True)
    assert len(out) == B and all(tuple(o.shape) == (n_new, 8) for o in out)
    assert all(torch.equal(o[:, 0], f) for o, f in zip(out, firsts))
    codes = torch.stack([torch.cat([p, o]) for p, o in zip(prompts, out)])          # (B, 150 + 2250, 8)
    batch = {'tokens': torch.stack(texts), 'tokens_lens': torch.full((B,), n_text), 'codes': codes,
             'codes_lens': torch.full((B,), n_prompt + n_new)}
    for stage in (1, 7):
        with torch.no_grad():
            logits, p = nar.stage_logits(batch, stage)
        assert p == n_prompt and tuple(logits.shape) == (B, n_new, ncfg.num_audio_tokens)
        top2 = logits.topk(2, dim=-1)
        got = torch.stack([o[:, stage] for o in out])
        off = top2.indices[..., 0] != got
        margin = (top2.values[..., 0] - top2.values[..., 1])[off]
        assert off.float().mean().item() < 1e-3 and (margin.numel() == 0 or float(margin.max()) < 1e-3), \
            (stage, int(off.sum()), float(margin.max()) if margin.numel() else 0.0)


def test_maximum_length_of_the_positional_table():
    """The longest sequence the path can represent: the sinusoid tables hold 5000 positions per stream
    (valle/models/modules.py:56-66, max_len 5000).  A tiny AR model over 30 text + 5000 audio positions (BOS + 4999
    frames; row 1 shorter, so key padding is live): teacher-forced loss and every gradient against the oracle's autograd —
    157 query tiles, 20 key chunks through the attention backward — and one position more is refused, not read past
    the table."""
    from oracle import valle_oracle as O
    from tests.test_train_gpu import _grad_check
    from valle2_amd import _lib, get_model_class, synth
    kw = dict(C.TINY, norm='LayerNorm')
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleAR', seed=11, rich=True)
    batch = synth.synth_ar_batch(cfg, 2, tok_range=(30, 30), code_range=(4999, 4999), seed=5)
    cut = 3777                                                # row 1: 3777 real frames, then padding (zeros, as collate pads)
    batch['codes'][1, cut + 1:] = 0
    batch['target'][1, cut:] = 0
    batch['target'][1, cut] = cfg.eos_token
    batch['codes_lens'][1] = cut + 1
    assert batch['codes'].shape[1] == 5000
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref = O.ar_training_loss(params, cfg, batch)
    ref.backward()
    model = get_model_class('ValleAR')(cfg)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    assert _lib.lib().vh_attn_rows_bwd_chunks(2, cfg.n_heads, 5030, 1) >= 20
    loss = model.training_step({k: v.clone() for k, v in batch.items()})
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    loss.backward()
    _grad_check(model, params, sorted(k for k in params if not k.endswith('.pe')))
    # one more audio position than the table holds
    longer = synth.synth_ar_batch(cfg, 1, tok_range=(30, 30), code_range=(5000, 5000), seed=6)
    with pytest.raises((_lib.VhError, IndexError)):
        model.training_step(longer)
    utt = synth.synth_utterance(cfg, 4, 4, 4990, seed=3)
    gen = get_model_class('ValleAR')(C.cfg_of(dict(kw, num_beams=2, top_k=1, max_audio_len=64)))
    gen.load_state_dict(sd)
    with pytest.raises(_lib.VhError):
        gen.to(DEV).eval().generate(*[u.to(DEV) for u in utt])
