"""Golden-vector case definitions shared by the generator (real reference), the oracle tests
(CPU restatement) and the GPU parity tests (HIP path).

Every case is a pure function of seeds: inputs and weights come from `valle2_amd.synth`, so the
.npz fixtures hold only the reference's OUTPUTS.  `REFERENCE_RUNNERS[name](ref)` runs the real
reference modules (only in the build container); `ORACLE_RUNNERS[name]()` runs oracle/ on the
same inputs and must reproduce the fixture.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from valle2_amd import synth
from valle2_amd.config import ConfigValle

# ---------------------------------------------------------------------------------------------
# configs
# ---------------------------------------------------------------------------------------------
TINY = dict(d_model=128, n_heads=2, dim_feedforward=512, num_layers=2, dropout=0.0)
MID = dict(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0)


def cfg_of(kw, cls=ConfigValle):
    return cls(**kw)


AR_TINY = dict(TINY, norm='LayerNorm', num_beams=4, top_k=1, max_audio_len=64)
AR_MID = dict(MID, norm='LayerNorm', num_beams=2, top_k=1, max_audio_len=48)
NAR_TINY = dict(TINY, norm='AdaptiveLayerNorm')
# full BASELINE.json sizes (round 2): configs[1] generate, configs[3] training batch, configs[4] NAR stage
AR_FULL = dict(MID, norm='LayerNorm', num_beams=32, top_k=1, max_audio_len=512)
AR_TRAIN_FULL = dict(MID, norm='LayerNorm')
BIG = dict(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0)
NAR_BIG = dict(BIG, norm='AdaptiveLayerNorm')
AR_BIG = dict(BIG, norm='LayerNorm', num_beams=8, top_k=1, max_audio_len=48)   # configs[4]'s AR leg (round 3)
FULL_TEXT, FULL_FRAMES = 256, 767           # configs[1]: 256 text + BOS + 767 codec tokens = 1024
TRAIN_FULL_BATCH = 16                       # configs[3]: per-GPU B=16

TRAIN_LOGIT_STRIDE = 61                     # teacher-forced logits kept at every 61st audio position
NAR_BIG_TEXT, NAR_BIG_FRAMES = 400, 2475    # configs[4]: 400 text + 225 prompt + 2250 target frames
NAR_FULL = dict(MID, norm='AdaptiveLayerNorm')   # configs[2] at full size (round 3): 12L/512d, batch 64, 256 text + 768 frames
NAR_FULL_BATCH, NAR_FULL_TEXT, NAR_FULL_FRAMES = 64, 256, 768
NAR_FULL_ROWS, NAR_FULL_STRIDE, NAR_FULL_STAGE = (0, 1, 31, 63), 29, 3      # what the fixture keeps of the (64, 618, 1024) logits
NAR_BIG_STRIDE = 29
SAMPLING_FILTERS = [(50, 1.0, 1.0), (5, 0.9, 0.7), (0, 0.8, 1.0), (20, 0.5, 1.3)]   # (top_k, tok_p, temperature)
MHA_SHAPES = [(512, 8, 4, 5), (256, 4, 8, 10), (128, 2, 16, 20)]  # reference tests/test_modules.py:9-13


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def mha_inputs(d, h, b, t):
    g = torch.Generator().manual_seed(100 + d + t)
    sd = {'qkv.weight': 0.05 * torch.randn(3 * d, d, generator=g),
          'out.weight': 0.05 * torch.randn(d, d, generator=g),
          'out.bias': 0.05 * torch.randn(d, generator=g)}
    x = torch.randn(b, t, d, generator=g)
    causal = torch.triu(torch.ones(t, t), diagonal=1)          # float mask, as the reference test
    lens = torch.tensor([max(1, t - (i % t)) for i in range(b)])
    pad = (torch.arange(t)[None, :] >= lens[:, None]).to(torch.int64)
    return sd, x, causal, pad


def transformer_inputs(norm):
    kw = dict(TINY, norm=norm)
    cfg = cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'Transformer', seed=7, rich=True, std=0.05)
    b, xl, yl = 3, 8, 16
    x = _randn((b, xl + yl, cfg.d_model), 11)
    lens = torch.tensor([16, 11, 5])
    pad = F.pad(torch.arange(yl)[None, :] >= lens[:, None], (xl, 0), value=False)
    emb = _randn((1, cfg.d_model), 12)
    return kw, sd, x, xl, yl, pad, emb


def ar_train_inputs():
    cfg = cfg_of(AR_TINY)
    sd = synth.make_state_dict(cfg, 'ValleAR', seed=3, rich=True)
    batch = synth.synth_ar_batch(cfg, 3, tok_range=(5, 12), code_range=(13, 30), seed=21)
    return AR_TINY, sd, batch


def ar_generate_inputs(which):
    kw = {'tiny': AR_TINY, 'mid': AR_MID, 'full': AR_FULL, 'big': AR_BIG}[which]
    cfg = cfg_of(kw)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=5, rich=True), cfg)
    if which == 'full':   # BASELINE.json configs[1] at full size: 32 beams, 256 text + BOS + 767 codec tokens
        utt = synth.synth_utterance(cfg, FULL_TEXT // 2, FULL_TEXT // 2, FULL_FRAMES, seed=1234)
    elif which == 'big':    # BASELINE.json configs[4], AR leg: 24L/1024d/h16, 8 beams, 400 text + BOS + 225 codec tokens
        utt = synth.synth_utterance(cfg, 200, 200, 225, seed=1234)
    elif which == 'tiny':   # BASELINE.json configs[0]: 128 text + 256 EnCodec tokens (255 + BOS)
        utt = synth.synth_utterance(cfg, 64, 64, 255, seed=1234)
    else:                 # reduced configs[1]: 12L/512d, short prompt so the fixture stays small
        utt = synth.synth_utterance(cfg, 24, 24, 47, seed=1234)
    return kw, sd, utt


FORCED_BIG_NEW = 2250                                        # configs[4]: 2250 new tokens after a 225-frame prompt
FORCED_BIG_POS = (225, 230, 900, 1536, 2047, 2048, 2474)     # audio positions whose teacher-forced logits are kept


def ar_forced_big_inputs():
    """configs[4]'s AR leg over its WHOLE context range (round 5): the 24L/1024d/h16 model and utterance of
    `ar_generate_inputs('big')` (400 text + BOS + 225 prompt frames) followed by 2250 FORCED tokens — 2875 positions, the
    end of a 30 s utterance.  Audio position p = 225 + t holds the logits decode step t produces (context 626 + t)."""
    kw, sd, utt = ar_generate_inputs('big')
    forced = torch.randint(0, cfg_of(kw).num_audio_tokens, (FORCED_BIG_NEW,), generator=torch.Generator().manual_seed(4242))
    return kw, sd, utt, forced


def ar_prefill_full_inputs():
    """4 DISTINCT utterances at the configs[1] prompt shape (generate() itself replicates one utterance
    over its beams, valle_ar.py:136-138, so distinct rows go through the reference's sub-modules)."""
    kw, sd, _ = ar_generate_inputs('full')
    cfg = cfg_of(kw)
    utts = [synth.synth_utterance(cfg, FULL_TEXT // 2, FULL_TEXT // 2, FULL_FRAMES, seed=4321 + i) for i in range(4)]
    text = torch.stack([torch.cat([u[0], u[2]]) for u in utts])
    codes = torch.stack([F.pad(u[1][:, 0], (1, 0), value=cfg.bos_token) for u in utts])
    pos = torch.tensor([0, 1, 100, 333, 500, 640, 766, 767])       # audio positions whose logits are kept
    return kw, sd, text, codes, pos


def ar_train_full_inputs():
    """configs[3]-shaped AR training batch: 12L/512d, tokens_lens ~U{40..120}, codes_lens ~U{225..900}."""
    cfg = cfg_of(AR_TRAIN_FULL)
    sd = synth.make_state_dict(cfg, 'ValleAR', seed=3, rich=True)
    batch = synth.synth_ar_batch(cfg, TRAIN_FULL_BATCH, seed=2121)
    return AR_TRAIN_FULL, sd, batch


def nar_big_inputs():
    """configs[4] NAR leg: 24L/1024d/h16, one utterance, 400 text + 2475 frames (225 prompt + 2250 target)."""
    cfg = cfg_of(NAR_BIG)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=17, rich=True)
    batch = synth.synth_nar_batch(cfg, 1, n_tokens=NAR_BIG_TEXT, n_frames=NAR_BIG_FRAMES, seed=313)
    return NAR_BIG, sd, batch


def nar_full_inputs(rows=None):
    """configs[2]: the 12L/512d NAR stack over batch 64 x (256 text + 768 frames) = 1024 positions.  `rows`: only these
    utterances of the batch (rows are independent in the NAR forward: the CPU suite re-runs two of them)."""
    cfg = cfg_of(NAR_FULL)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=29, rich=True)
    batch = synth.synth_nar_batch(cfg, NAR_FULL_BATCH, n_tokens=NAR_FULL_TEXT, n_frames=NAR_FULL_FRAMES, seed=1234)
    if rows is not None:
        batch = {k: v[list(rows)] for k, v in batch.items()}
    return NAR_FULL, sd, batch


def ar_eos_inputs(eos_row=None):
    """A run that really reaches EOS.  `eos_row` (stored in the fixture by the generator, which
    derives it from a free run of the reference: 1.05 x the head row of the token emitted at
    step 5) is planted as the EOS row of the head, so EOS overtakes that token no later than
    step 5 and every beam stops (valle/models/valle_ar.py:167-170)."""
    kw = dict(AR_TINY, max_audio_len=40)
    cfg = cfg_of(kw)
    sd = synth.make_state_dict(cfg, 'ValleAR', seed=9, rich=True)
    if eos_row is not None:
        sd['proj.weight'][cfg.num_audio_tokens] = torch.as_tensor(eos_row)
    else:
        synth.silence_eos(sd, cfg)
    utt = synth.synth_utterance(cfg, 6, 7, 9, seed=77)
    return kw, sd, utt


def nar_inputs():
    cfg = cfg_of(NAR_TINY)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=13, rich=True)
    batch = synth.synth_nar_batch(cfg, 2, n_tokens=10, n_frames=36, seed=31)
    return NAR_TINY, sd, batch


def nar_generate_inputs():
    cfg = cfg_of(NAR_TINY)
    sd = synth.make_state_dict(cfg, 'ValleNAR', seed=13, rich=True)
    g = torch.Generator().manual_seed(55)
    pt = torch.randint(0, cfg.vocab_size, (6,), generator=g)
    tt = torch.randint(0, cfg.vocab_size, (9,), generator=g)
    pc = torch.randint(0, cfg.num_audio_tokens, (12, cfg.num_quantizers), generator=g)
    first = torch.randint(0, cfg.num_audio_tokens, (20,), generator=g)
    return NAR_TINY, sd, (pt, pc, tt, first)


def sampling_inputs():
    logits = 3.0 * _randn((6, 1025), 41)
    x = torch.tensor([[5, 9, 1024, 1024, 1024], [5, 7, 3, 2, 1024], [1, 2, 3, 4, 5]])
    lp = torch.tensor([-1.0, -3.5, -3.0])
    return logits, x, lp


# ---------------------------------------------------------------------------------------------
# reference runners (real reference modules)
# ---------------------------------------------------------------------------------------------
def _ref_masks(ref):
    u = ref['utils']
    out = {'attn_5_5': u.build_attn_mask(5, 5, device='cpu'),
           'attn_3_7': u.build_attn_mask(3, 7, device='cpu'),
           'pad_a': u.build_pad_mask(torch.tensor([5, 5, 5, 5]), device='cpu'),
           'pad_b': u.build_pad_mask(torch.tensor([5, 4, 3, 2]), device='cpu')}
    for d, h, b, t in MHA_SHAPES[:2]:
        _, _, causal, pad = mha_inputs(d, h, b, t)
        m = ref['modules'].MultiHeadAttention(d, h)
        out[f'merge_{d}'] = m.merge_masks(b, causal, pad)
    return out


def _ref_mha(ref):
    out = {}
    for d, h, b, t in MHA_SHAPES:
        sd, x, causal, pad = mha_inputs(d, h, b, t)
        m = ref['modules'].MultiHeadAttention(d, h).eval()
        m.load_state_dict(sd)
        o, (k, v) = m(x, attn_mask=causal, use_cache=True)
        o2, _ = m(x, attn_mask=causal, padding_mask=pad)
        o3, _ = m(x)                                     # no mask at all
        # one cached decode step: append one new row to the cache just returned
        xn = _randn((b, 1, d), 300 + d)
        o4, (k4, v4) = m(xn, kv_cache=(k, v), use_cache=True)
        out.update({f'out_{d}': o, f'k_{d}': k, f'v_{d}': v, f'out_pad_{d}': o2,
                    f'out_nomask_{d}': o3, f'out_step_{d}': o4, f'k_step_{d}': k4})
    return out


def _ref_transformer(ref):
    out = {}
    for norm in ('LayerNorm', 'AdaptiveLayerNorm'):
        kw, sd, x, xl, yl, pad, emb = transformer_inputs(norm)
        cfg = cfg_of(kw, ref['config'].ConfigValle)
        m = ref['modules'].Transformer(cfg).eval()
        m.load_state_dict(sd)
        mask = ref['utils'].build_attn_mask(xl, yl, device='cpu')
        e = emb if norm != 'LayerNorm' else None
        y, kv = m(x, padding_mask=pad, attn_mask=mask, embedding=e, use_cache=True)
        yfull, _ = m(x, embedding=e)                      # full attention (NAR shape)
        xn = torch.cat([x, _randn((x.shape[0], 1, x.shape[2]), 19)], dim=1)
        ystep, kv2 = m(xn, attn_mask=mask, embedding=e, kv_cache=kv, use_cache=True)
        out.update({f'{norm}_y': y, f'{norm}_yfull': yfull, f'{norm}_ystep': ystep,
                    f'{norm}_k0': kv[0][0], f'{norm}_vlast': kv2[-1][1]})
    return out


def _ref_ar_train(ref):
    kw, sd, batch = ar_train_inputs()
    with torch.enable_grad():
        cfg = cfg_of(kw, ref['config'].ConfigValle)
        m = ref['ar'].ValleAR(cfg).eval()
        m.load_state_dict(sd)
        loss = m.training_step({k: v.clone() for k, v in batch.items()})
        loss.backward()
        grads = {n: p.grad.norm() for n, p in m.named_parameters()}
    names = sorted(grads)
    return {'loss': loss.detach(), 'grad_norms': torch.stack([grads[n] for n in names]).detach()}


DROPOUT_SEED = 4242        # torch.manual_seed right before the train-mode forward, on both sides
AR_TINY_DROPOUT = dict(AR_TINY, dropout=0.1)        # the reference's default p (valle/config.py:26)


def _ref_ar_train_dropout(ref):
    """ValleAR.training_step in TRAIN mode (dropout1/2 + FeedForward dropout p = 0.1, PositionalEncoding dropout 0.1):
    loss, per-parameter gradient norms and the logits under torch.manual_seed(DROPOUT_SEED).  The oracle must draw the
    same Bernoulli fields from the same generator state — this pins WHERE it applies dropout."""
    _, sd, batch = ar_train_inputs()
    with torch.enable_grad():
        cfg = cfg_of(AR_TINY_DROPOUT, ref['config'].ConfigValle)
        m = ref['ar'].ValleAR(cfg).train()
        m.load_state_dict(sd)
        rows = []
        hook = m.proj.register_forward_hook(lambda mod, i, o: rows.append(o.detach().clone()))
        torch.manual_seed(DROPOUT_SEED)
        loss = m.training_step({k: v.clone() for k, v in batch.items()})
        hook.remove()
        loss.backward()
        grads = {n: p.grad.norm() for n, p in m.named_parameters()}
    names = sorted(grads)
    return {'loss': loss.detach(), 'grad_norms': torch.stack([grads[n] for n in names]).detach(), 'logits': rows[0]}


def _ref_transformer_dropout(ref):
    """Transformer.forward in TRAIN mode, both norms (AdaLN = the NAR stack, whose training_step raises in the reference)."""
    out = {}
    for norm in ('LayerNorm', 'AdaptiveLayerNorm'):
        kw, sd, x, xl, yl, pad, emb = transformer_inputs(norm)
        cfg = cfg_of(dict(kw, dropout=0.1), ref['config'].ConfigValle)
        m = ref['modules'].Transformer(cfg).train()
        m.load_state_dict(sd)
        mask = ref['utils'].build_attn_mask(xl, yl, device='cpu')
        e = emb if norm != 'LayerNorm' else None
        torch.manual_seed(DROPOUT_SEED)
        y, _ = m(x, padding_mask=pad, attn_mask=mask, embedding=e)
        out[f'{norm}_y'] = y
    return out


def _ref_generate(ref, which):
    kw, sd, utt = ar_generate_inputs(which)
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['ar'].ValleAR(cfg).eval()
    m.load_state_dict(sd)
    # record per-step logits of the real loop through a hook on the head
    rows = []
    hook = m.proj.register_forward_hook(lambda mod, i, o: rows.append(o[:, -1].clone()))
    torch.manual_seed(0)
    tokens = m.generate(*utt)
    hook.remove()
    logits = torch.stack(rows)                            # (steps, beams, V)
    top2 = torch.topk(logits[:, 0], 2, dim=-1)[0]
    return {'tokens': tokens, 'margin': top2[:, 0] - top2[:, 1],
            'logits_row0': logits[:, 0][:: max(1, len(rows) // 8)], 'steps': torch.tensor(len(rows))}


def _ref_prefill_full(ref):
    kw, sd, text, codes, pos = ar_prefill_full_inputs()
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['ar'].ValleAR(cfg).eval()
    m.load_state_dict(sd)
    # valle_ar.py:121-158 at step 0, for distinct rows: embed + PE, prefix-LM mask, stack, head
    tok = m.tokens_position_emb(m.tokens_emb(text))
    aud = m.audio_position_emb(m.audio_emb(codes))
    mask = ref['utils'].build_attn_mask(text.shape[1], codes.shape[1], device='cpu')
    y, _ = m.transformer(torch.cat([tok, aud], dim=1), attn_mask=mask, use_cache=True)
    logits = m.proj(y[:, text.shape[1]:][:, pos])
    return {'logits': logits, 'hidden_last': y[:, -1]}


def _ref_ar_forced_big(ref):
    """ONE teacher-forced pass of the real reference's sub-modules over 400 text + 2475 audio positions under its own
    prefix-LM mask (valle_ar.py:61-83 / :141-158 at kv_cache=None): embed + PE, build_attn_mask, Transformer, proj."""
    kw, sd, utt, forced = ar_forced_big_inputs()
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['ar'].ValleAR(cfg).eval()
    m.load_state_dict(sd)
    text = torch.cat([utt[0], utt[2]])[None]
    codes = torch.cat([torch.tensor([cfg.bos_token]), utt[1][:, 0], forced[:-1]])[None]      # (1, 2475)
    assert text.shape[1] == 400 and codes.shape[1] == 2475
    tok = m.tokens_position_emb(m.tokens_emb(text))
    aud = m.audio_position_emb(m.audio_emb(codes))
    mask = ref['utils'].build_attn_mask(text.shape[1], codes.shape[1], device='cpu')
    y, _ = m.transformer(torch.cat([tok, aud], dim=1), attn_mask=mask)
    return {'logits': m.proj(y[0, text.shape[1]:][list(FORCED_BIG_POS)])}


def _ref_ar_train_full(ref):
    kw, sd, batch = ar_train_full_inputs()
    with torch.enable_grad():
        cfg = cfg_of(kw, ref['config'].ConfigValle)
        m = ref['ar'].ValleAR(cfg).eval()
        m.load_state_dict(sd)
        rows = []
        hook = m.proj.register_forward_hook(lambda mod, i, o: rows.append(o.detach()[:, ::TRAIN_LOGIT_STRIDE].clone()))
        loss = m.training_step({k: v.clone() for k, v in batch.items()})
        hook.remove()
        loss.backward()
        grads = {n: p.grad.norm() for n, p in m.named_parameters()}
    names = sorted(grads)
    return {'loss': loss.detach(), 'grad_norms': torch.stack([grads[n] for n in names]).detach(),
            'logits_sub': rows[0]}


def _ref_nar_big(ref):
    kw, sd, batch = nar_big_inputs()
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['nar'].ValleNAR(cfg).eval()
    m.load_state_dict(sd)
    out = {}
    tx = int(batch['tokens_lens'].max())
    tok = m.tokens_position_emb(m.tokens_emb(batch['tokens']))
    for stage in (2, 7):
        y, p = m._prepare_audio_codes(batch['codes'], stage)
        z, _ = m.transformer(torch.cat([tok, m.audio_position_emb(y)], dim=1),
                             embedding=m.stage_embs[stage - 1].weight)
        out[f'logits_{stage}'] = m.proj_layers[stage - 1](z[:, tx + p:][:, ::NAR_BIG_STRIDE])
        out[f'prefix_{stage}'] = torch.tensor(p)
    return out


def _ref_nar_full(ref):
    kw, sd, batch = nar_full_inputs()
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['nar'].ValleNAR(cfg).eval()
    m.load_state_dict(sd)
    tx = int(batch['tokens_lens'].max())
    tok = m.tokens_position_emb(m.tokens_emb(batch['tokens']))
    y, p = m._prepare_audio_codes(batch['codes'], NAR_FULL_STAGE)
    z, _ = m.transformer(torch.cat([tok, m.audio_position_emb(y)], dim=1), embedding=m.stage_embs[NAR_FULL_STAGE - 1].weight)
    logits = m.proj_layers[NAR_FULL_STAGE - 1](z[list(NAR_FULL_ROWS), tx + p:][:, ::NAR_FULL_STRIDE])
    return {'logits': logits, 'prefix': torch.tensor(p)}


def _ref_sampling_filter(ref):
    """The deterministic part of topk_sampling at top_k > 1 / top_p < 1 (valle/models/utils.py:46-68): the
    filtered scores that top_k_top_p_filtering hands to multinomial (their -inf pattern = the support) and
    the log_softmax row the returned log-prob is gathered from."""
    u = ref['utils']
    logits, _, _ = sampling_inputs()
    out = {}
    real = u.top_k_top_p_filtering
    for i, (k, p, temp) in enumerate(SAMPLING_FILTERS):
        seen = []

        def spy(lg, **kw):
            res = real(lg, **kw)
            seen.append(res.clone())
            return res
        u.top_k_top_p_filtering = spy
        try:
            torch.manual_seed(i)
            tok, lp = u.topk_sampling(logits.clone(), top_k=k, tok_p=p, temperature=temp)
        finally:
            u.top_k_top_p_filtering = real
        filt = seen[0]
        out[f'keep_{i}'] = torch.isfinite(filt)
        out[f'logprobs_{i}'] = F.log_softmax(filt, dim=-1)
        out[f'tok_{i}'] = tok
        out[f'lp_{i}'] = lp
    return out


def _ref_generate_eos(ref):
    kw, sd, utt = ar_eos_inputs()
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['ar'].ValleAR(cfg).eval()
    m.load_state_dict(sd)
    torch.manual_seed(0)
    free = m.generate(*utt)                               # EOS silenced: runs all 40 steps
    eos_row = 1.05 * sd['proj.weight'][int(free[5])].clone()
    kw, sd, utt = ar_eos_inputs(eos_row)
    m.load_state_dict(sd)
    rows = []
    hook = m.proj.register_forward_hook(lambda mod, i, o: rows.append(o[:, -1].clone()))
    torch.manual_seed(0)
    stopped = m.generate(*utt)
    hook.remove()
    return {'eos_row': eos_row, 'free_tokens': free, 'tokens': stopped,
            'steps': torch.tensor(len(rows))}


def _ref_nar(ref):
    kw, sd, batch = nar_inputs()
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['nar'].ValleNAR(cfg).eval()
    m.load_state_dict(sd)
    out = {}
    for stage in (1, 4, 7):
        y, p = m._prepare_audio_codes(batch['codes'], stage)   # works in the reference
        out[f'prep_{stage}'] = y
        out[f'prefix_{stage}'] = torch.tensor(p)
        # sub-expression chain of the intended forward, each piece computed by reference modules
        tx = int(batch['tokens_lens'].max())
        tok = m.tokens_position_emb(m.tokens_emb(batch['tokens']))
        yy = m.audio_position_emb(y)
        z, _ = m.transformer(torch.cat([tok, yy], dim=1), embedding=m.stage_embs[stage - 1].weight)
        out[f'logits_{stage}'] = m.proj_layers[stage - 1](z[:, tx + p:])
    return out


def _ref_sampling(ref):
    logits, x, lp = sampling_inputs()
    torch.manual_seed(0)
    tok, cur = ref['utils'].topk_sampling(logits, top_k=1, tok_p=1.0, temperature=1.0)
    return {'greedy_tok': tok, 'greedy_lp': cur,
            'best_beam_1': ref['utils'].get_best_beam(x, lp, 1024, 1.0),
            'best_beam_2': ref['utils'].get_best_beam(x, lp, 1024, 0.0)}


def collate_inputs():
    g = torch.Generator().manual_seed(61)
    return [{'codes': torch.randint(0, 1024, (8, t), generator=g), 'tokens': torch.randint(0, 256, (n,), generator=g)}
            for t, n in ((12, 5), (9, 3), (20, 11))]


def _ref_collate(ref):
    import valle.collate as rcollate            # the reference's module (imported by gen_golden's harness)
    cfg = cfg_of(AR_TINY, ref['config'].ConfigValle)
    return dict(rcollate.ValleARCollate(cfg)(collate_inputs()))


# ---- head widths other than 64 (round 4): d_model // n_heads = 32 and 128 -------------------------------------------
HD_MHA_SHAPES = [(128, 4, 3, 19), (256, 2, 2, 23)]            # (d_model, n_heads, batch, seq): head_dim 32, 128
AR_HD32 = dict(d_model=128, n_heads=4, dim_feedforward=256, num_layers=2, dropout=0.0, norm='LayerNorm', num_beams=3,
               top_k=1, max_audio_len=24)


def head_dim_inputs():
    cfg = cfg_of(AR_HD32)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=31, rich=True), cfg)
    utt = synth.synth_utterance(cfg, 9, 7, 25, seed=77)
    batch = synth.synth_ar_batch(cfg, 3, tok_range=(5, 11), code_range=(20, 45), seed=8)
    return AR_HD32, sd, utt, batch


def _ref_head_dim(ref):
    """The real reference at head widths 32 and 128: MultiHeadAttention (causal, causal + padding, a cached step) and a
    2-layer ValleAR with head width 32 — greedy generate with margins, training loss and per-parameter gradient norms."""
    out = {}
    for d, h, b, t in HD_MHA_SHAPES:
        sd, x, causal, pad = mha_inputs(d, h, b, t)
        m = ref['modules'].MultiHeadAttention(d, h).eval()
        m.load_state_dict(sd)
        assert m.head_dim == d // h != 64
        o, (k, v) = m(x, attn_mask=causal, use_cache=True)
        o2, _ = m(x, attn_mask=causal, padding_mask=pad)
        xn = _randn((b, 1, d), 300 + d)
        o4, (k4, _) = m(xn, kv_cache=(k, v), use_cache=True)
        out.update({f'out_{d}': o, f'k_{d}': k, f'out_pad_{d}': o2, f'out_step_{d}': o4, f'k_step_{d}': k4})
    kw, sd, utt, batch = head_dim_inputs()
    cfg = cfg_of(kw, ref['config'].ConfigValle)
    m = ref['ar'].ValleAR(cfg).eval()
    m.load_state_dict(sd)
    rows = []
    hook = m.proj.register_forward_hook(lambda mod, i, o: rows.append(o[:, -1].clone()))
    torch.manual_seed(0)
    tokens = m.generate(*utt)
    hook.remove()
    top2 = torch.topk(torch.stack(rows)[:, 0], 2, dim=-1)[0]
    out.update({'tokens': tokens, 'margin': top2[:, 0] - top2[:, 1]})
    with torch.enable_grad():
        m = ref['ar'].ValleAR(cfg).eval()
        m.load_state_dict(sd)
        loss = m.training_step({k: v.clone() for k, v in batch.items()})
        loss.backward()
        grads = {n: p.grad.norm() for n, p in m.named_parameters()}
    out.update({'loss': loss.detach(), 'grad_norms': torch.stack([grads[n] for n in sorted(grads)]).detach()})
    return out


REFERENCE_RUNNERS = {
    'head_dim': _ref_head_dim,
    'collate': _ref_collate,
    'masks': _ref_masks,
    'mha': _ref_mha,
    'transformer': _ref_transformer,
    'ar_train': _ref_ar_train,
    'ar_train_dropout': _ref_ar_train_dropout,
    'transformer_dropout': _ref_transformer_dropout,
    'ar_generate_tiny': lambda ref: _ref_generate(ref, 'tiny'),
    'ar_generate_mid': lambda ref: _ref_generate(ref, 'mid'),
    'ar_generate_eos': _ref_generate_eos,
    'nar': _ref_nar,
    'sampling': _ref_sampling,
    'sampling_filter': _ref_sampling_filter,
    'ar_generate_full': lambda ref: _ref_generate(ref, 'full'),
    'ar_generate_big': lambda ref: _ref_generate(ref, 'big'),
    'nar_full': _ref_nar_full,
    'ar_prefill_full': _ref_prefill_full,
    'ar_train_full': _ref_ar_train_full,
    'nar_big': _ref_nar_big,
    'ar_forced_big': _ref_ar_forced_big,
}
