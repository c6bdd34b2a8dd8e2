"""NAR decoding on the device: the batched, ragged `ValleNAR.generate_batch` and the stage sampler
`vh_categorical_rows` (Categorical(logits / temperature), valle/models/valle_nar.py:142-160).

The reference's ValleNAR.generate raises, so the checker for the token stream is the
oracle's intended algorithm (greedy, per utterance alone); the sampler is checked distributionally
(its RNG stream is not torch's) and for its log-probabilities."""
import pytest
import torch
import torch.nn.functional as F

from tests.golden import cases as C

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _utterances(cfg, shapes, seed):
    g = torch.Generator().manual_seed(seed)
    us = []
    for tx, tc, ty in shapes:
        us.append((torch.randint(0, cfg.vocab_size, (tx,), generator=g),
                   torch.randint(0, cfg.num_audio_tokens, (tc, cfg.num_quantizers), generator=g),
                   torch.randint(0, cfg.num_audio_tokens, (ty,), generator=g)))
    return us


def _model(kw, sd):
    from valle2_amd import get_model_class
    m = get_model_class('ValleNAR')(C.cfg_of(kw))
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def test_generate_batch_ragged_rows_match_the_oracle_per_utterance():
    from oracle import valle_oracle as O
    from valle2_amd import synth
    kw = dict(d_model=256, n_heads=4, dim_feedforward=1024, num_layers=3, dropout=0.0, norm='AdaptiveLayerNorm')
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=41, rich=True)
    us = _utterances(cfg, [(15, 12, 20), (9, 30, 41), (22, 5, 7), (15, 12, 20)], seed=5)
    m = _model(kw, sd)
    outs = m.generate_batch([u[0].to(DEV) for u in us], [u[1].to(DEV) for u in us], [u[2].to(DEV) for u in us],
                            greedy=True)
    assert len(outs) == 4
    for (text, pc, first), got in zip(us, outs):
        ref = O.nar_generate(sd, cfg, text[:4], pc, text[4:], first, greedy=True)     # text split is immaterial
        assert got.shape == ref.shape == (first.shape[0], 8) and got.dtype == torch.int64
        assert torch.equal(got[:, 0].cpu(), first)
        assert torch.equal(got.cpu(), ref), f'{(got.cpu() != ref).sum().item()} of {ref.numel()} tokens differ'
    # a row decodes the same alone as inside the ragged batch (other kernel shapes, no padding)
    alone = m.generate(us[1][0][:3].to(DEV), us[1][1].to(DEV), us[1][0][3:].to(DEV), us[1][2].to(DEV), greedy=True)
    assert torch.equal(alone, outs[1])
    # CPU inputs are accepted (ids are range-checked on the host before they travel)
    cpu = m.generate_batch([us[2][0]], [us[2][1]], [us[2][2]], greedy=True)
    assert torch.equal(cpu[0], outs[2].cpu())


def test_generate_batch_sampling_is_seeded_and_in_range():
    kw, sd, (pt, pc, tt, first) = C.nar_generate_inputs()
    m = _model(dict(kw, temperature=0.9), sd)
    args = ([torch.cat([pt, tt]).to(DEV)] * 3, [pc.to(DEV)] * 3, [first.to(DEV)] * 3)
    torch.manual_seed(11)
    a = m.generate_batch(*args)
    torch.manual_seed(11)
    b = m.generate_batch(*args)
    torch.manual_seed(12)
    c = m.generate_batch(*args)
    assert all(torch.equal(x, y) for x, y in zip(a, b)), 'torch.manual_seed must make sampling repeatable'
    assert any(not torch.equal(x, y) for x, y in zip(a, c))
    assert not torch.equal(a[0], a[1]), 'identical rows draw independently (the RNG is keyed on the row)'
    for x in a:
        assert x.shape == (20, 8) and int(x.min()) >= 0 and int(x.max()) < 1024
        assert torch.equal(x[:, 0].cpu(), first)


@pytest.mark.parametrize('V,temperature', [(1024, 1.0), (1024, 0.6), (1025, 1.7), (37, 1.0), (64, 0.5), (5, 1.0)])
def test_categorical_rows_distribution_and_logprob(V, temperature):
    from valle2_amd import kernels as K
    n = 40000
    g = torch.Generator().manual_seed(V)
    row = 2.0 * torch.randn(V, generator=g)
    probs = F.softmax(row / temperature, dim=-1)
    logits = row.to(DEV)[None].expand(n, -1).contiguous()
    tok = torch.empty(n, device=DEV, dtype=torch.int64)
    lp = torch.empty(n, device=DEV)
    K.categorical_rows(logits, tok, temperature=temperature, seed=99, stream_id=3, logprob=lp)
    tok, lp = tok.cpu(), lp.cpu()
    assert int(tok.min()) >= 0 and int(tok.max()) < V
    freq = torch.bincount(tok, minlength=V).float() / n
    sigma = torch.sqrt(probs * (1 - probs) / n)
    assert bool(((freq - probs).abs() <= 4 * sigma + 1e-4).all()), (freq - probs).abs().max()
    torch.testing.assert_close(lp, torch.log(probs[tok]), atol=2e-5, rtol=1e-5)
    tok2 = torch.empty(n, device=DEV, dtype=torch.int64)
    K.categorical_rows(logits, tok2, temperature=temperature, seed=99, stream_id=3)
    assert torch.equal(tok2.cpu(), tok), 'same (seed, row, stream) must give the same draw'
    K.categorical_rows(logits, tok2, temperature=temperature, seed=99, stream_id=4)
    assert not torch.equal(tok2.cpu(), tok), 'another stage must draw afresh'


def test_categorical_rows_greedy_ties_strides_and_errors():
    from valle2_amd import _lib, kernels as K
    logits = torch.randn(70, 1024, generator=torch.Generator().manual_seed(2))
    logits[3, 17] = logits[3, 900] = 9.0                              # exact tie: the lowest index wins
    logits[4, 1023] = 11.0                                             # maximum in the last lane's range
    out = torch.full((70, 8), -1, device=DEV, dtype=torch.int64)
    K.categorical_rows(logits.to(DEV), out[:, 5], greedy=True)       # strided token column (codebook 5)
    exp = torch.argmax(logits, dim=-1)
    exp[3] = 17
    assert torch.equal(out[:, 5].cpu(), exp) and int((out[:, :5] != -1).sum()) == 0
    padded = torch.zeros(6, 1028, device=DEV)
    padded[:, :1025] = torch.randn(6, 1025, device=DEV)
    padded[:, 1025:] = 100.0                                           # beyond V: must never be read as a score
    tok = torch.empty(6, device=DEV, dtype=torch.int64)
    K.categorical_rows(padded[:, :1025], tok, greedy=True)
    assert torch.equal(tok.cpu(), torch.argmax(padded[:, :1025].cpu(), dim=-1))
    with pytest.raises(_lib.VhError):
        K.categorical_rows(padded[:, :1025], tok, temperature=0.0)


def test_out_of_range_ids_raise_index_error_like_the_reference():
    """nn.Embedding / F.cross_entropy raise IndexError for ids outside their table.  Host-resident ids are
    checked before they travel; device-resident ids by the kernels (never an out-of-bounds access) with the
    error raised at the next synchronisation (end of generate / the optimizer step)."""
    from valle2_amd import _lib, get_model_class, synth
    kw = dict(C.NAR_TINY)
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=13, rich=True)
    m = _model(kw, sd)
    batch = synth.synth_nar_batch(cfg, 2, n_tokens=10, n_frames=36, seed=31)
    bad = {k: v.clone() for k, v in batch.items()}
    bad['codes'][1, 7, 3] = cfg.num_audio_tokens                       # EOS inside NAR codes: no such row
    with pytest.raises(IndexError):
        m.stage_logits(bad, 4)                                         # host tensors: checked on the host
    _lib.raise_device_errors()                                         # nothing pending
    dev_bad = {k: (v.to(DEV) if not k.endswith('_lens') else v) for k, v in bad.items()}
    logits, _ = m.stage_logits(dev_bad, 4)                             # device tensors: kernel clamps + flags
    assert bool(torch.isfinite(logits).all())
    with pytest.raises(IndexError, match='embedding table'):
        _lib.raise_device_errors(DEV)
    _lib.raise_device_errors(DEV)                                      # flag cleared by the raise
    # AR training: a device-resident target of -100 (torch's ignore_index) is refused at the optimizer step
    akw = dict(C.AR_TINY)
    acfg = C.cfg_of(akw)
    ar = get_model_class('ValleAR')(acfg)
    ar.load_state_dict(synth.make_state_dict(acfg, 'ValleAR', seed=3, rich=True))
    ar = ar.to(DEV).train()
    opt = ar.configure_optimizers()['optimizer']
    b = synth.synth_ar_batch(acfg, 2, tok_range=(5, 9), code_range=(13, 20), seed=1)
    b = {k: (v.to(DEV) if not k.endswith('_lens') else v) for k, v in b.items()}
    b['target'][0, 2] = -100
    before = opt.flat_param.clone()
    with torch.enable_grad():
        ar.training_step(b).backward()
    opt.step()                               # the update launch sees the flag and leaves everything untouched ...
    with pytest.raises(IndexError, match='target'):
        opt.check_errors()                   # ... and the host raises when it looks (the next step does so by itself)
    assert torch.equal(opt.flat_param, before), 'a step with a poisoned batch must not touch the parameters'
    opt.zero_grad()
    b['target'][0, 2] = 5
    with torch.enable_grad():
        ar.training_step(b).backward()
    opt.step()
    assert not torch.equal(opt.flat_param, before)
