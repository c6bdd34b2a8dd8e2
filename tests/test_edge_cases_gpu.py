"""Edges of the path's input space against the CPU oracle (the reference's op sequence): the smallest inputs the
reference's signatures admit, the largest the positional table admits, and inputs the reference refuses.

  generate():  no target text, one text token, a one-frame acoustic prompt, one beam, one / two new tokens, an EOS at the first
               step, a context that ends exactly at the last row of the positional table (max_len 5000, modules.py:56);
  NAR:         one- to four-frame utterances (the prefix rule min(T // 3, 150) gives a prefix of 0 or 1 frame), one text token;
  training:    AR / NAR steps on one-row batches, one text token, two codec frames, rows of very different lengths;
  refusals:    ranks the reference asserts on (valle_ar.py:109-112), ids outside their tables (IndexError, as nn.Embedding), a
               sequence beyond the table.
"""
import pytest
import torch

from tests.golden import cases as C

pytestmark = pytest.mark.gpu
DEV = 'cuda'
KW = dict(d_model=128, n_heads=2, dim_feedforward=512, num_layers=2, dropout=0.0)


def _ar(seed=5, **over):
    from valle2_amd import get_model_class, synth
    kw = dict(KW, norm='LayerNorm', num_beams=4, top_k=1, max_audio_len=12)
    kw.update(over)
    cfg = C.cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=seed, rich=True, std=0.15), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    return cfg, sd, m.to(DEV).eval()


def _same_tokens(m, sd, cfg, utt):
    from oracle import valle_oracle as O
    trace = {}
    ref = O.ar_generate(sd, cfg, *utt, trace=trace)
    out = m.generate(*[None if u is None else u.to(DEV) for u in utt]).cpu()
    n = min(len(out), len(ref))
    bad = (out[:n] != ref[:n]).nonzero()
    assert len(out) == len(ref) and (bad.numel() == 0 or trace['margin'][int(bad[0])] < 1e-4), (out.tolist(), ref.tolist())
    return out


@pytest.mark.parametrize('n_prompt_tokens,n_target_tokens,n_frames', [(1, None, 1), (1, 1, 1), (7, None, 30), (3, 5, 1), (40, 0, 2)])
def test_generate_smallest_inputs(n_prompt_tokens, n_target_tokens, n_frames):
    """One text token, no target text (target_tokens=None, valle_ar.py:95-97,120-123) or an EMPTY one, a one-frame prompt: the
    prompt pass is then 3 positions (text, BOS, one frame) and every kernel runs far below its tile sizes."""
    from valle2_amd import synth
    cfg, sd, m = _ar()
    u = synth.synth_utterance(cfg, n_prompt_tokens, max(n_target_tokens or 0, 1), n_frames, seed=100 + n_frames)
    target = None if n_target_tokens is None else u[2][:n_target_tokens]
    out = _same_tokens(m, sd, cfg, (u[0], u[1], target))
    assert out.dtype == torch.int64 and out.dim() == 1 and len(out) <= cfg.max_audio_len


@pytest.mark.parametrize('beams,max_new', [(1, 1), (1, 2), (4, 1), (2, 3), (64, 2)])
def test_generate_one_beam_and_one_new_token(beams, max_new):
    from valle2_amd import synth
    cfg, sd, m = _ar(num_beams=beams, max_audio_len=max_new)
    u = synth.synth_utterance(cfg, 9, 4, 6, seed=200 + beams)
    out = _same_tokens(m, sd, cfg, u)
    assert len(out) == max_new
    assert m.last_generate_stats['steps_run'] == max_new


def test_generate_that_emits_eos_at_the_first_step_returns_nothing():
    """valle_ar.py:168-171,174-180: every beam emits EOS at step 0, the loop breaks before anything is appended and the best beam
    stripped of EOS is an empty 1-D int64 tensor."""
    from valle2_amd import get_model_class, synth
    kw = dict(KW, norm='LayerNorm', num_beams=3, top_k=1, max_audio_len=8)
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleAR', seed=6, rich=True, std=0.15)
    sd['proj.weight'][:] = 0.0
    sd['proj.weight'][cfg.num_audio_tokens] = 0.0
    # logits = proj . y: make EOS the arg-max for any hidden state — every other row the negative of the EOS row would need y;
    # simpler: a zero matrix gives all-equal logits (arg-max = column 0), so put the only nonzero row at EOS with a sign that is
    # decided by the oracle itself: try both and keep the one whose oracle run ends at step 0
    from oracle import valle_oracle as O
    u = synth.synth_utterance(cfg, 5, 5, 4, seed=7)
    chosen = None
    for sign in (1.0, -1.0):
        sd['proj.weight'][cfg.num_audio_tokens] = sign * sd['audio_emb.word_embeddings.weight'][0, :].sign() * 0.5
        trace = {}
        ref = O.ar_generate(sd, cfg, *u, trace=trace)
        if len(ref) == 0 and len(trace['tokens']) == 1:
            chosen = sign
            break
    if chosen is None:
        pytest.skip('no sign makes EOS the first arg-max for this seed')
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    out = m.generate(*[t.to(DEV) for t in u])
    assert out.dtype == torch.int64 and out.dim() == 1 and out.numel() == 0
    assert m.last_generate_stats['tokens_appended'] == 0


def test_generate_up_to_the_last_row_of_the_positional_table():
    """modules.py:56: PositionalEncoding holds max_len = 5000 rows.  A prompt of 4990 frames + BOS and 9 new tokens ends exactly
    on row 4999 of the audio table (tokens equal to the oracle's); one token more is refused before anything is launched."""
    from valle2_amd import _lib, synth
    cfg, sd, m = _ar(num_beams=1, max_audio_len=9)
    u = synth.synth_utterance(cfg, 6, 3, 4990, seed=300)
    _same_tokens(m, sd, cfg, u)
    assert m.last_generate_stats['s0'] == 9 + 4991
    m.config.max_audio_len = 10
    with pytest.raises(_lib.VhError, match='positional table'):
        m.generate(*[t.to(DEV) for t in u])
    m.config.max_audio_len = 9
    long_text = torch.zeros(5001, dtype=torch.int64)
    with pytest.raises(_lib.VhError, match='positional table'):
        m.generate(long_text.to(DEV), u[1].to(DEV))


def test_generate_with_a_prompt_beyond_the_shared_kernels_record_bound_decodes_as_independent_rows():
    """vh_attn_decode_shared merges at most 256 records per (beam, head): ceil(s0 / 32) prompt blocks + the suffix splits (16 at
    4 beams x 2 heads) — a context of more than 7680 keys is legal (text <= 5000, prompt <= 5000 - max_new positions) and
    must not raise at graph capture: generate() falls back to independent rows (round-5 advisor finding), the decoder's
    create refuses the over-long shared form.  s0 = 4990 text ids + BOS + 2690 frames = 7681; tokens against a forced
    independent-rows run."""
    from valle2_amd import engine, synth
    cfg, sd, m = _ar(max_audio_len=3)
    assert engine.shared_prompt_fits(4, 2, 7680) and not engine.shared_prompt_fits(4, 2, 7681)
    u = synth.synth_utterance(cfg, 2495, 2495, 2690, seed=77)
    out = m.generate(*[t.to(DEV) for t in u]).cpu()
    st = m.last_generate_stats
    assert not st['shared_prompt'] and st['steps_run'] == 3
    text = torch.cat([u[0], u[2]]).to(DEV)
    rows = m.generate_batch([text] * 4, [u[1][:, 0].to(DEV)] * 4)            # the same four rows, never shared
    assert torch.equal(out, rows[0, u[1].shape[0] + 1:u[1].shape[0] + 1 + len(out)].cpu())
    # one step below the bound the shared form still runs
    u2 = synth.synth_utterance(cfg, 2495, 2495, 2689, seed=78)
    m.generate(*[t.to(DEV) for t in u2])
    assert m.last_generate_stats['shared_prompt']


def test_generate_refuses_what_the_reference_asserts_on_and_ids_outside_their_tables():
    from valle2_amd import _lib, synth
    cfg, sd, m = _ar()
    u = [t.to(DEV) for t in synth.synth_utterance(cfg, 5, 5, 6, seed=9)]
    with pytest.raises(AssertionError, match='1D'):
        m.generate(u[0].unsqueeze(0), u[1], u[2])
    with pytest.raises(AssertionError, match='2D'):
        m.generate(u[0], u[1][:, 0], u[2])
    with pytest.raises(AssertionError, match='1D'):
        m.generate(u[0], u[1], u[2].unsqueeze(0))
    # an id outside its table: nn.Embedding raises IndexError in the reference; here the gather kernels flag it on the device
    # (no host read of device-resident ids) and the generate raises IndexError when it next synchronises — and the flag is
    # cleared, so the next call is clean
    bad_text = u[0].clone()
    bad_text[2] = cfg.vocab_size
    with pytest.raises(IndexError, match='embedding table'):
        m.generate(bad_text, u[1], u[2])
    bad_codes = u[1].clone()
    bad_codes[3, 0] = cfg.num_audio_tokens + 5
    with pytest.raises(IndexError, match='embedding table'):
        m.generate(u[0], bad_codes, u[2])
    assert m.generate(u[0], u[1], u[2]).dim() == 1
    with pytest.raises(ValueError, match='non-empty'):
        m.generate_batch([], [])


@pytest.mark.parametrize('n_tokens,n_frames', [(1, 1), (1, 2), (2, 3), (5, 4), (3, 7)])
def test_nar_stage_on_utterances_of_a_few_frames(n_tokens, n_frames):
    """valle_nar.py:179: prefix = min(T // 3, 150) — 0 frames for T < 3 (every frame is predicted), 1 for T = 3 .. 5."""
    from oracle import valle_oracle as O
    from valle2_amd import get_model_class, synth
    cfg = C.cfg_of(dict(KW, norm='AdaptiveLayerNorm'))
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=11, rich=True, std=0.15)
    m = get_model_class('ValleNAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    batch = synth.synth_nar_batch(cfg, 3, n_tokens=n_tokens, n_frames=n_frames, seed=400 + n_frames)
    for stage in (1, 4, 7):
        ref, p_ref = O.nar_stage_logits(sd, cfg, batch, stage)
        got, p = m.stage_logits(batch, stage)
        assert p == p_ref == min(n_frames // 3, 150)
        assert tuple(got.shape) == tuple(ref.shape) == (3, n_frames - p, cfg.num_audio_tokens)
        torch.testing.assert_close(got.cpu(), ref, atol=2e-4, rtol=1e-4)


def _grads_match(model, params, names, tol=1e-3):
    for n in names:
        got, ref = dict(model.named_parameters())[n].grad, params[n].grad
        assert got is not None, f'no gradient reached {n}'
        err = (got.cpu() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        assert err < tol, f'{n}: relative gradient error {err:.2e}'


@pytest.mark.parametrize('batch,tok_range,code_range', [(1, (1, 1), (2, 2)), (1, (3, 3), (40, 40)), (2, (1, 9), (2, 30)),
                                                       (5, (1, 4), (5, 70))])
def test_ar_training_step_on_the_smallest_batches(batch, tok_range, code_range):
    """valle_ar.py:43-90 on batches at the lower edge of the collate format (one row; one text token; two codec frames;
    rows of very different lengths, so most positions of a row are padding that still counts in the loss): loss and every
    parameter's gradient against the oracle's autograd."""
    from oracle import valle_oracle as O
    from valle2_amd import get_model_class, synth
    kw = dict(KW, norm='LayerNorm')
    cfg = C.cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleAR', seed=21, rich=True, std=0.15)
    b = synth.synth_ar_batch(cfg, batch, tok_range=tok_range, code_range=code_range, seed=500 + batch)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref_loss = O.ar_training_loss(params, cfg, b)
    ref_loss.backward()
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    loss = m.training_step({k: v.clone() for k, v in b.items()})
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), rtol=1e-5, atol=1e-6)
    loss.backward()
    _grads_match(m, params, sorted(k for k, v in params.items() if v.grad is not None and float(v.grad.abs().sum()) > 0))


@pytest.mark.parametrize('batch,n_tokens,n_frames,stage', [(1, 1, 1, 1), (1, 2, 3, 7), (2, 1, 2, 4), (3, 4, 6, 2)])
def test_nar_training_step_on_the_smallest_batches(batch, n_tokens, n_frames, stage):
    from oracle import valle_oracle as O
    from valle2_amd import get_model_class, synth
    cfg = C.cfg_of(dict(KW, norm='AdaptiveLayerNorm'))
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=22, rich=True, std=0.15)
    b = synth.synth_nar_batch(cfg, batch, n_tokens=n_tokens, n_frames=n_frames, seed=600 + n_frames)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    ref_loss = O.nar_training_loss(params, cfg, b, stage)
    ref_loss.backward()
    m = get_model_class('ValleNAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    loss = m.training_step(b, stage=stage)
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), rtol=1e-5, atol=1e-6)
    loss.backward()
    _grads_match(m, params, sorted(k for k, v in params.items() if v.grad is not None and float(v.grad.abs().sum()) > 0))


def test_nar_generate_on_the_smallest_utterances_and_an_extremely_ragged_batch():
    """valle_nar.py:107-165 (the oracle's greedy form): a one-frame target after a one-frame prompt and one text token; and a
    batch whose rows differ by two orders of magnitude in every length — each row equal to the oracle run on it alone."""
    from oracle import valle_oracle as O
    from valle2_amd import get_model_class, synth
    cfg = C.cfg_of(dict(KW, norm='AdaptiveLayerNorm'))
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=31, rich=True, std=0.15)
    m = get_model_class('ValleNAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(7)
    shapes = [(2, 1, 1), (2, 1, 3), (150, 90, 200), (3, 200, 2), (40, 2, 120)]          # (text, prompt frames, target frames)
    us = [(torch.randint(0, cfg.vocab_size, (t,), generator=g), torch.randint(0, cfg.num_audio_tokens, (p, cfg.num_quantizers), generator=g),
           torch.randint(0, cfg.num_audio_tokens, (y,), generator=g)) for t, p, y in shapes]
    outs = m.generate_batch([u[0].to(DEV) for u in us], [u[1].to(DEV) for u in us], [u[2].to(DEV) for u in us], greedy=True)
    for (text, pc, first), got in zip(us, outs):
        ref = O.nar_generate(sd, cfg, text[:1], pc, text[1:], first, greedy=True)
        assert got.shape == ref.shape == (first.shape[0], cfg.num_quantizers)
        assert torch.equal(got.cpu(), ref), f'{(got.cpu() != ref).sum().item()} of {ref.numel()} codes differ'
    one = m.generate(us[0][0][:1].to(DEV), us[0][1].to(DEV), us[0][0][1:].to(DEV), us[0][2].to(DEV), greedy=True)
    assert torch.equal(one, outs[0])


def test_ar_generate_batch_extremely_ragged_rows():
    """Rows of one decode batch with 1 .. 120 text tokens and 1 .. 300 prompt frames: every row equal to the oracle on that
    utterance alone (per-row key lengths, per-row positions, EOS-filled tails)."""
    from oracle import valle_oracle as O
    from valle2_amd import synth
    cfg, sd, m = _ar(num_beams=1, max_audio_len=10)
    shapes = [(1, 1), (120, 300), (2, 299), (119, 1), (30, 40), (1, 300)]
    us = [synth.synth_utterance(cfg, t, 1, p, seed=700 + i) for i, (t, p) in enumerate(shapes)]
    rows = m.generate_batch([torch.cat([u[0], u[2]]).to(DEV) for u in us], [u[1][:, 0].to(DEV) for u in us])
    for r, u in enumerate(us):
        trace = {}
        ref = O.ar_generate(sd, cfg, *u, trace=trace)
        p0 = m.last_generate_stats['prompt_lens'][r]
        got = rows[r, p0:p0 + len(ref)].cpu()
        bad = (got != ref).nonzero()
        assert bad.numel() == 0 or trace['margin'][int(bad[0])] < 1e-4, (r, got.tolist(), ref.tolist())


@pytest.mark.parametrize('top_k', [1, 20])
def test_a_decoder_survives_generate_and_the_next_call_of_the_shape_reuses_it(top_k):
    """DESIGN 8.2 / VERDICT r5 item 5: the ArDecoder of a shape (graphs, K/V caches, counters, workspaces) is kept on the model;
    the second generate() of that shape builds and captures nothing.  Same tokens as the first call (greedy), the same tokens
    under the same torch seed whether the decoder is fresh or reused (the sampling seed is a device scalar the captured steps
    read, not a frozen kernel argument), a different prompt of the same shape decodes like a fresh model does, another shape
    gets its own slot, and changed weights are noticed."""
    from valle2_amd import get_model_class, synth
    cfg, sd, m = _ar(top_k=top_k, max_audio_len=20)
    u1 = synth.synth_utterance(cfg, 9, 4, 14, seed=301)
    u2 = synth.synth_utterance(cfg, 9, 4, 14, seed=302)            # the same shape, other ids
    u3 = synth.synth_utterance(cfg, 11, 6, 9, seed=303)            # another shape
    dev = lambda u: [t.to(DEV) for t in u]

    def fresh(u, seed):
        m2 = get_model_class('ValleAR')(cfg)
        m2.load_state_dict(sd)
        m2 = m2.to(DEV).eval()
        torch.manual_seed(seed)
        return m2.generate(*dev(u)).cpu()

    torch.manual_seed(11)
    a = m.generate(*dev(u1)).cpu()
    assert not m.last_generate_stats['decoder_reused']
    torch.manual_seed(11)
    b = m.generate(*dev(u1)).cpu()
    st = m.last_generate_stats
    assert st['decoder_reused'] and st['slot_uses'] == 2 and st['host_decoder_ms'] < 0.2, st
    assert torch.equal(a, b) and torch.equal(a, fresh(u1, 11))
    torch.manual_seed(12)
    c = m.generate(*dev(u2)).cpu()
    assert m.last_generate_stats['decoder_reused'] and torch.equal(c, fresh(u2, 12))
    if top_k != 1:                                                  # another seed draws other tokens through the same graph
        torch.manual_seed(13)
        assert not torch.equal(m.generate(*dev(u1)).cpu(), a)
    torch.manual_seed(14)
    d = m.generate(*dev(u3)).cpu()
    assert not m.last_generate_stats['decoder_reused'] and torch.equal(d, fresh(u3, 14))
    # weights moved (a plain in-place update): the old decoder's tables are stale and must not be used
    with torch.no_grad():
        m.proj.weight.mul_(-1.0)
    torch.manual_seed(11)
    e = m.generate(*dev(u1)).cpu()
    assert not m.last_generate_stats['decoder_reused']
    m.release_decoders()
    assert not m.__dict__.get('_decode_slots')


def test_two_host_threads_on_their_own_streams_decode_side_by_side():
    """include/valle_hip.h: "calls are thread-safe for distinct streams".  Two host threads, each with its own torch stream and
    its own model, run generates (prompt pass, graph capture, replays, EOS polls) at the same time; each must return what it
    returns alone.  (Capture is serialised on the engine's shared capture stream; hipStreamCaptureModeThreadLocal keeps the
    other thread's allocations out of it.)"""
    import threading
    from valle2_amd import synth
    jobs = []
    for i in range(2):
        cfg, sd, m = _ar(seed=40 + i, num_beams=1, max_audio_len=40)
        us = [synth.synth_utterance(cfg, 10 + r, 6, 20 + 3 * r, seed=800 + 10 * i + r) for r in range(6)]
        texts, prompts = [torch.cat([u[0], u[2]]).to(DEV) for u in us], [u[1][:, 0].to(DEV) for u in us]
        jobs.append((m, texts, prompts, m.generate_batch(texts, prompts).clone()))
    torch.cuda.synchronize()
    errors, results = [], [[], []]

    def work(i):
        try:
            m, texts, prompts, _ = jobs[i]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(4):
                    results[i].append(m.generate_batch(texts, prompts).clone())
                st.synchronize()
        except Exception as e:                       # noqa: BLE001 — reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for i in range(2):
        assert len(results[i]) == 4 and all(torch.equal(r, jobs[i][3]) for r in results[i]), f'thread {i} decoded something else'
