"""Run oracle/ on the golden cases' inputs (same keys as tests/golden/cases.REFERENCE_RUNNERS)."""
from __future__ import annotations

import numpy as np
import torch

from oracle import valle_oracle as O
from tests.golden import cases as C

GOLDEN_DIR = C.__file__.rsplit('/', 1)[0]


def load_golden(name):
    with np.load(f'{GOLDEN_DIR}/{name}.npz') as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def masks():
    out = {'attn_5_5': O.build_attn_mask(5, 5), 'attn_3_7': O.build_attn_mask(3, 7),
           'pad_a': O.build_pad_mask(torch.tensor([5, 5, 5, 5])),
           'pad_b': O.build_pad_mask(torch.tensor([5, 4, 3, 2]))}
    for d, h, b, t in C.MHA_SHAPES[:2]:
        _, _, causal, pad = C.mha_inputs(d, h, b, t)
        out[f'merge_{d}'] = O.merge_masks(b, h, causal, pad)
    return out


def mha():
    out = {}
    for d, h, b, t in C.MHA_SHAPES:
        sd, x, causal, pad = C.mha_inputs(d, h, b, t)
        o, (k, v) = O.multi_head_attention(sd, '', x, h, attn_mask=causal, use_cache=True)
        o2, _ = O.multi_head_attention(sd, '', x, h, attn_mask=causal, padding_mask=pad)
        o3, _ = O.multi_head_attention(sd, '', x, h)
        xn = C._randn((b, 1, d), 300 + d)
        o4, (k4, _) = O.multi_head_attention(sd, '', xn, h, kv_cache=(k, v), use_cache=True)
        out.update({f'out_{d}': o, f'k_{d}': k, f'v_{d}': v, f'out_pad_{d}': o2,
                    f'out_nomask_{d}': o3, f'out_step_{d}': o4, f'k_step_{d}': k4})
    return out


def transformer():
    out = {}
    for norm in ('LayerNorm', 'AdaptiveLayerNorm'):
        kw, sd, x, xl, yl, pad, emb = C.transformer_inputs(norm)
        cfg = C.cfg_of(kw)
        mask = O.build_attn_mask(xl, yl)
        e = emb if norm != 'LayerNorm' else None
        y, kv = O.transformer(sd, '', x, cfg, padding_mask=pad, attn_mask=mask, embedding=e,
                              use_cache=True)
        yfull, _ = O.transformer(sd, '', x, cfg, embedding=e)
        xn = torch.cat([x, C._randn((x.shape[0], 1, x.shape[2]), 19)], dim=1)
        ystep, kv2 = O.transformer(sd, '', xn, cfg, attn_mask=mask, embedding=e, kv_cache=kv,
                                   use_cache=True)
        out.update({f'{norm}_y': y, f'{norm}_yfull': yfull, f'{norm}_ystep': ystep,
                    f'{norm}_k0': kv[0][0], f'{norm}_vlast': kv2[-1][1]})
    return out


def head_dim():
    out = {}
    for d, h, b, t in C.HD_MHA_SHAPES:
        sd, x, causal, pad = C.mha_inputs(d, h, b, t)
        o, (k, v) = O.multi_head_attention(sd, '', x, h, attn_mask=causal, use_cache=True)
        o2, _ = O.multi_head_attention(sd, '', x, h, attn_mask=causal, padding_mask=pad)
        xn = C._randn((b, 1, d), 300 + d)
        o4, (k4, _) = O.multi_head_attention(sd, '', xn, h, kv_cache=(k, v), use_cache=True)
        out.update({f'out_{d}': o, f'k_{d}': k, f'out_pad_{d}': o2, f'out_step_{d}': o4, f'k_step_{d}': k4})
    kw, sd, utt, batch = C.head_dim_inputs()
    cfg = C.cfg_of(kw)
    trace = {}
    out['tokens'] = O.ar_generate(sd, cfg, *utt, trace=trace)
    out['margin'] = torch.tensor(trace['margin'])
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    with torch.enable_grad():
        loss = O.ar_training_loss(params, cfg, batch)
        loss.backward()
    names = sorted(k for k in params if not k.endswith('.pe'))
    out.update({'loss': loss.detach(), 'grad_norms': torch.stack([params[n].grad.norm() for n in names])})
    return out


def ar_train():
    kw, sd, batch = C.ar_train_inputs()
    cfg = C.cfg_of(kw)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    with torch.enable_grad():
        loss = O.ar_training_loss(params, cfg, batch)
        loss.backward()
    names = sorted(k for k in params if not k.endswith('.pe'))
    return {'loss': loss.detach(),
            'grad_norms': torch.stack([params[n].grad.norm() for n in names])}


def ar_train_dropout():
    _, sd, batch = C.ar_train_inputs()
    cfg = C.cfg_of(C.AR_TINY_DROPOUT)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    with torch.enable_grad():
        torch.manual_seed(C.DROPOUT_SEED)
        logits = O.ar_logits(params, cfg, batch, O.Dropout(cfg.dropout))
        loss = torch.nn.functional.cross_entropy(logits, batch['target'])
        loss.backward()
    names = sorted(k for k in params if not k.endswith('.pe'))
    return {'loss': loss.detach(), 'grad_norms': torch.stack([params[n].grad.norm() for n in names]),
            'logits': logits.detach().permute(0, 2, 1).contiguous()}


def transformer_dropout():
    out = {}
    for norm in ('LayerNorm', 'AdaptiveLayerNorm'):
        kw, sd, x, xl, yl, pad, emb = C.transformer_inputs(norm)
        cfg = C.cfg_of(dict(kw, dropout=0.1))
        torch.manual_seed(C.DROPOUT_SEED)
        y, _ = O.transformer(sd, '', x, cfg, padding_mask=pad, attn_mask=O.build_attn_mask(xl, yl),
                             embedding=emb if norm != 'LayerNorm' else None, drop=O.Dropout(cfg.dropout))
        out[f'{norm}_y'] = y
    return out


def _generate(kw, sd, utt):
    cfg = C.cfg_of(kw)
    trace = {}
    torch.manual_seed(0)
    tokens = O.ar_generate(sd, cfg, *utt, trace=trace)
    logits = torch.stack(trace['logits'])
    n = logits.shape[0]
    return {'tokens': tokens, 'margin': torch.tensor(trace['margin']),
            'logits_row0': logits[:, 0][:: max(1, n // 8)], 'steps': torch.tensor(n)}


def ar_generate_tiny():
    return _generate(*C.ar_generate_inputs('tiny'))


def ar_generate_mid():
    return _generate(*C.ar_generate_inputs('mid'))


def ar_generate_eos():
    gold = load_golden('ar_generate_eos')
    kw, sd, utt = C.ar_eos_inputs()
    torch.manual_seed(0)
    free = O.ar_generate(sd, C.cfg_of(kw), *utt)
    kw, sd, utt = C.ar_eos_inputs(gold['eos_row'])
    res = _generate(kw, sd, utt)
    return {'eos_row': gold['eos_row'], 'free_tokens': free, 'tokens': res['tokens'],
            'steps': res['steps']}


def nar():
    kw, sd, batch = C.nar_inputs()
    cfg = C.cfg_of(kw)
    out = {}
    for stage in (1, 4, 7):
        y, p = O.nar_prepare_audio_codes(sd, cfg, batch['codes'], stage)
        out[f'prep_{stage}'] = y
        out[f'prefix_{stage}'] = torch.tensor(p)
        out[f'logits_{stage}'] = O.nar_stage_logits(sd, cfg, batch, stage)[0]
    return out


def sampling():
    logits, x, lp = C.sampling_inputs()
    torch.manual_seed(0)
    tok, cur = O.topk_sampling(logits, top_k=1, tok_p=1.0, temperature=1.0)
    return {'greedy_tok': tok, 'greedy_lp': cur,
            'best_beam_1': O.get_best_beam(x, lp, 1024, 1.0),
            'best_beam_2': O.get_best_beam(x, lp, 1024, 0.0)}


def sampling_filter():
    logits, _, _ = C.sampling_inputs()
    out = {}
    for i, (k, p, temp) in enumerate(C.SAMPLING_FILTERS):
        filt = O._top_k_top_p_filter(logits / temp, top_k=k, top_p=p)
        out[f'keep_{i}'] = torch.isfinite(filt)
        out[f'logprobs_{i}'] = torch.log_softmax(filt, dim=-1)
        torch.manual_seed(i)
        out[f'tok_{i}'], out[f'lp_{i}'] = O.topk_sampling(logits.clone(), top_k=k, tok_p=p, temperature=temp)
    return out


ORACLE_FULL_STEPS = 20      # the oracle re-runs the first steps of the 512-step full-size golden (CPU time)


def ar_generate_full():
    kw, sd, utt = C.ar_generate_inputs('full')
    res = _generate(dict(kw, max_audio_len=ORACLE_FULL_STEPS), sd, utt)
    return {'tokens': res['tokens'], 'margin': res['margin']}


def ar_generate_big():
    """configs[4]'s AR leg (24L/1024d, 8 beams, 626-token prompt): the oracle re-runs the first steps of the 48."""
    kw, sd, utt = C.ar_generate_inputs('big')
    res = _generate(dict(kw, max_audio_len=4), sd, utt)
    return {'tokens': res['tokens'], 'margin': res['margin']}


def ar_prefill_full():
    kw, sd, text, codes, pos = C.ar_prefill_full_inputs()
    cfg = C.cfg_of(kw)
    tok = O.add_position(O.embed(sd['tokens_emb.word_embeddings.weight'], text), sd['tokens_position_emb.pe'])
    aud = O.add_position(O.embed(sd['audio_emb.word_embeddings.weight'], codes), sd['audio_position_emb.pe'])
    mask = O.build_attn_mask(text.shape[1], codes.shape[1])
    y, _ = O.transformer(sd, 'transformer.', torch.cat([tok, aud], dim=1), cfg, attn_mask=mask, use_cache=True)
    return {'logits': torch.nn.functional.linear(y[:, text.shape[1]:][:, pos], sd['proj.weight']),
            'hidden_last': y[:, -1]}


def ar_forced_big():
    """configs[4]'s AR leg, one teacher-forced pass over 400 text + 2475 audio positions (24L/1024d/h16)."""
    kw, sd, utt, forced = C.ar_forced_big_inputs()
    cfg = C.cfg_of(kw)
    text = torch.cat([utt[0], utt[2]])[None]
    codes = torch.cat([torch.tensor([cfg.bos_token]), utt[1][:, 0], forced[:-1]])[None]
    tok = O.add_position(O.embed(sd['tokens_emb.word_embeddings.weight'], text), sd['tokens_position_emb.pe'])
    aud = O.add_position(O.embed(sd['audio_emb.word_embeddings.weight'], codes), sd['audio_position_emb.pe'])
    mask = O.build_attn_mask(text.shape[1], codes.shape[1])
    y, _ = O.transformer(sd, 'transformer.', torch.cat([tok, aud], dim=1), cfg, attn_mask=mask)
    return {'logits': torch.nn.functional.linear(y[0, text.shape[1]:][list(C.FORCED_BIG_POS)], sd['proj.weight'])}


def ar_train_full():
    kw, sd, batch = C.ar_train_full_inputs()
    cfg = C.cfg_of(kw)
    params = {k: v.clone().requires_grad_(not k.endswith('.pe')) for k, v in sd.items()}
    with torch.enable_grad():
        logits = O.ar_logits(params, cfg, batch)                       # (B, V, Ty)
        loss = torch.nn.functional.cross_entropy(logits, batch['target'])
        loss.backward()
    names = sorted(k for k in params if not k.endswith('.pe'))
    return {'loss': loss.detach(), 'grad_norms': torch.stack([params[n].grad.norm() for n in names]),
            'logits_sub': logits.detach().permute(0, 2, 1)[:, ::C.TRAIN_LOGIT_STRIDE].contiguous()}


def nar_big():
    kw, sd, batch = C.nar_big_inputs()
    cfg = C.cfg_of(kw)
    out = {}
    for stage in (2, 7):
        logits, p = O.nar_stage_logits(sd, cfg, batch, stage)
        out[f'logits_{stage}'] = logits[:, ::C.NAR_BIG_STRIDE].contiguous()
        out[f'prefix_{stage}'] = torch.tensor(p)
    return out


def nar_full():
    """configs[2] at full size: the oracle re-runs utterances 0 and 1 of the 64 (rows are independent in the NAR forward)."""
    kw, sd, batch = C.nar_full_inputs(rows=C.NAR_FULL_ROWS[:2])
    cfg = C.cfg_of(kw)
    logits, p = O.nar_stage_logits(sd, cfg, batch, C.NAR_FULL_STAGE)
    assert p == min(C.NAR_FULL_FRAMES // 3, 3 * cfg.quantization_factor)
    return {'logits': logits[:, ::C.NAR_FULL_STRIDE].contiguous()}


# golden keys that a runner reproduces only as a prefix (full-size cases trimmed for CPU time)
PREFIX_KEYS = {'ar_generate_full': ('tokens', 'margin'), 'ar_generate_big': ('tokens', 'margin'), 'nar_full': ('logits',)}

ORACLE_RUNNERS = {
    'sampling_filter': sampling_filter, 'ar_generate_full': ar_generate_full, 'ar_generate_big': ar_generate_big,
    'nar_full': nar_full, 'ar_forced_big': ar_forced_big,
    'ar_prefill_full': ar_prefill_full, 'ar_train_full': ar_train_full, 'nar_big': nar_big,
    'masks': masks, 'mha': mha, 'head_dim': head_dim, 'transformer': transformer, 'ar_train': ar_train,
    'ar_train_dropout': ar_train_dropout, 'transformer_dropout': transformer_dropout,
    'ar_generate_tiny': ar_generate_tiny, 'ar_generate_mid': ar_generate_mid,
    'ar_generate_eos': ar_generate_eos, 'nar': nar, 'sampling': sampling,
}
