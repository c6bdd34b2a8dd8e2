"""CPU ORACLE for the AR + NAR codec-token transformer path.  TEST INFRASTRUCTURE ONLY.

This file restates, as plain functions over a `state_dict` on torch-CPU fp32, the algorithm of the
reference's hot path (KubiakJakub01/Valle2 @ 2024-10-22).  It is the checker that `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg compare the HIP path against; nothing
under `valle2_amd/` may import it, and the product fails loudly without its HIP library instead of
falling back to this file.

Pinning: `tests/golden/*.npz` were produced by importing the real reference in the build container
(`tests/golden/gen_golden.py`, committed) and `tests/test_oracle_golden.py` checks every function
below against them (bit-exact on the generating box; 2e-5 elsewhere because CPU BLAS/oneDNN kernels
differ per micro-architecture).  The op *sequence* deliberately follows the reference (per-step
`torch.cat` KV growth, full-sequence re-embedding, materialised (B,h,T,T) masks) so that (a) results
are bit-identical to it on the same torch build and (b) its timing is a fair CPU baseline.

Third-party arithmetic restated here because the package is absent/changed in this image:
`transformers==4.38.2` `top_k_top_p_filtering` (call site valle/models/utils.py:5,63) — see
`_top_k_top_p_filter`.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

NEG_INF = -float('inf')


# --------------------------------------------------------------------------------------------
# embeddings / position (valle/models/modules.py:11-80)
# --------------------------------------------------------------------------------------------
def positional_table(d_model: int, max_len: int = 5000) -> torch.Tensor:
    """valle/models/modules.py:60-66 — (max_len, 1, d) sinusoid buffer."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).transpose(0, 1)


class Dropout:
    """Train-mode dropout of the path (valle/models/modules.py:56-58,80 PositionalEncoding p = 0.1 always;
    :219 FeedForward, :277-278 EncoderLayer.dropout1/2 with p = config.dropout, valle/config.py:26), in two modes:

      Dropout(p_layers, p_pe)             the reference's own arithmetic: every site calls F.dropout in the reference's
                                          order, so under one torch.manual_seed the draws — and hence the results — are
                                          the reference's bit for bit.  The multiplier tensors (0 or 1/(1-p)) each site
                                          used are kept in `.used[name]` (drawn as F.dropout(ones): same generator
                                          consumption, and x * F.dropout(ones) is exactly what F.dropout(x) computes on
                                          the CPU: noise = bernoulli(1-p) / (1-p), out = x * noise).
      Dropout(..., masks={name: keep})    the SAME sites with GIVEN keep fields (uint8/bool, (rows..., cols) in the
                                          layout of the tensor they multiply): how the HIP path's counter-based fields
                                          (vh_dropout_mask) are replayed on the CPU.

    Site names: 'tokens_position_emb.dropout', 'audio_position_emb.dropout', 'layer{i}.dropout1',
    'layer{i}.ffn.dropout', 'layer{i}.dropout2'."""

    def __init__(self, p_layers: float, p_pe: float = 0.1, masks=None):
        self.p_layers, self.p_pe, self.masks, self.used = p_layers, p_pe, masks, {}

    def __call__(self, name: str, x: torch.Tensor, p: float) -> torch.Tensor:
        if p == 0:
            return x                                         # F.dropout(p=0) returns its input and draws nothing
        if self.masks is not None:
            noise = self.masks[name].reshape(x.shape).to(x.dtype) * (1.0 / (1.0 - p))
        else:
            noise = F.dropout(torch.ones_like(x), p, True)          # (same strides as x: same draw order)
        self.used[name] = noise
        return x * noise


def embed(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """valle/models/modules.py:33-37 — row gather (TokenEmbedding's dropout has p = 0: the identity, no draw)."""
    return F.embedding(ids, table)


def add_position(x: torch.Tensor, pe: torch.Tensor, drop: Dropout | None = None, name: str = '') -> torch.Tensor:
    """valle/models/modules.py:78-80 — x (B,T,d) + pe[:T] broadcast over batch, then dropout (train mode) on the
    (T,B,d) tensor; a given keep field is in (B,T,d) layout."""
    xt = x.permute(1, 0, 2)
    xt = xt + pe[: xt.size(0), :]
    if drop is not None:
        if drop.masks is not None and drop.p_pe:
            drop = Dropout(drop.p_layers, drop.p_pe, {name: drop.masks[name].reshape(x.shape).permute(1, 0, 2)})
        xt = drop(name, xt, drop.p_pe)
    return xt.permute(1, 0, 2)


# --------------------------------------------------------------------------------------------
# masks (valle/models/utils.py:8-43, valle/models/modules.py:175-207)
# --------------------------------------------------------------------------------------------
def build_pad_mask(lens: torch.Tensor) -> torch.Tensor:
    """valle/models/utils.py:8-14 — (B, max_len) bool, True = padded."""
    max_len = int(lens.max().item())
    return torch.arange(max_len).unsqueeze(0).expand(len(lens), -1) >= lens.unsqueeze(1)


def build_attn_mask(x_len: int, y_len: int) -> torch.Tensor:
    """valle/models/utils.py:17-43 — prefix-LM mask, True = masked.
    Rows <x_len (text) see text only; rows >=x_len (audio) see all text + causal audio."""
    top = torch.cat((torch.zeros(x_len, x_len, dtype=torch.bool),
                     torch.ones(x_len, y_len, dtype=torch.bool)), dim=1)
    bottom = torch.cat((torch.zeros(y_len, x_len, dtype=torch.bool),
                        torch.triu(torch.ones(y_len, y_len, dtype=torch.bool), diagonal=1)), dim=1)
    return torch.cat((top, bottom), dim=0)


def merge_masks(batch_size: int, n_heads: int, attn_mask, key_padding_mask):
    """valle/models/modules.py:175-207 — returns the SUM tensor (B,h,T,T) (or (B,1,T,T) for a 3-D
    attn_mask); key padding is only merged when an attn_mask is present (defect D6)."""
    if attn_mask is None:
        return None
    if attn_mask.dim() == 3:
        merged = attn_mask.unsqueeze(1)
    else:
        merged = attn_mask.unsqueeze(0).unsqueeze(0).expand(batch_size, n_heads, -1, -1)
    if key_padding_mask is not None:
        kp = key_padding_mask.unsqueeze(1).unsqueeze(1).expand(batch_size, n_heads, 1, -1)
        merged = merged + kp
    return merged


# --------------------------------------------------------------------------------------------
# transformer blocks (valle/models/modules.py:83-352)
# --------------------------------------------------------------------------------------------
def layer_norm(sd, p, x):
    """nn.LayerNorm(d), eps 1e-5 (valle/models/modules.py:284)."""
    return F.layer_norm(x, (x.shape[-1],), sd[p + 'weight'], sd[p + 'bias'], 1e-5)


def adaptive_layer_norm(sd, p, x, embedding):
    """valle/models/modules.py:93-99 — [w,b] = split(Linear(d,2d)(emb)); w * LN(x) + b."""
    d = x.shape[-1]
    wb = F.linear(embedding, sd[p + 'project_layer.weight'], sd[p + 'project_layer.bias'])
    weight, bias = torch.split(wb, d, dim=-1)
    return weight * layer_norm(sd, p + 'norm.', x) + bias


def _norm(sd, p, x, norm_kind, embedding):
    if norm_kind == 'LayerNorm':
        return layer_norm(sd, p, x)
    return adaptive_layer_norm(sd, p, x, embedding)


def multi_head_attention(sd, p, x, n_heads, attn_mask=None, padding_mask=None, kv_cache=None,
                         use_cache=False):
    """valle/models/modules.py:117-173 — qkv GEMM (no bias) → heads → KV cat → mask merge/negate →
    SDPA (scale 1/sqrt(hd)) → merge heads → out GEMM (+bias).  Returns (out, (k,v)|None) with k,v
    of shape (B,h,S,hd)."""
    b, n, d = x.shape
    hd = d // n_heads
    q, k, v = F.linear(x, sd[p + 'qkv.weight']).chunk(3, dim=-1)
    q, k, v = (t.view(b, n, n_heads, hd).permute(0, 2, 1, 3) for t in (q, k, v))
    kv = None
    if use_cache and kv_cache is not None:
        k = torch.cat([kv_cache[0], k], dim=-2)
        v = torch.cat([kv_cache[1], v], dim=-2)
    if use_cache:
        kv = (k, v)
    if attn_mask is not None:
        attn_mask = ~merge_masks(b, n_heads, attn_mask, padding_mask).to(dtype=torch.bool)
    attn = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
    out = attn.permute(0, 2, 1, 3).reshape(b, n, d)
    return F.linear(out, sd[p + 'out.weight'], sd[p + 'out.bias']), kv


def feed_forward(sd, p, x, drop=None, site=''):
    """valle/models/modules.py:215-221 — Linear → exact-erf GELU → dropout → Linear."""
    hidden = F.gelu(F.linear(x, sd[p + 'linear_1.weight'], sd[p + 'linear_1.bias']))
    if drop is not None:
        hidden = drop(site + 'ffn.dropout', hidden, drop.p_layers)
    return F.linear(hidden, sd[p + 'linear_2.weight'], sd[p + 'linear_2.bias'])


def encoder_layer(sd, p, x, cfg, padding_mask=None, attn_mask=None, embedding=None, kv_cache=None,
                  use_cache=False, drop=None, site=''):
    """valle/models/modules.py:240-280 — pre-norm residual block; `drop` (train mode): dropout1 on the attention
    branch, FeedForward's dropout, dropout2 on the FeedForward branch (:219, :277-278)."""
    a, kv = multi_head_attention(sd, p + 'self_attn.', _norm(sd, p + 'norm1.', x, cfg.norm, embedding),
                                 cfg.n_heads, attn_mask=attn_mask, padding_mask=padding_mask,
                                 kv_cache=kv_cache, use_cache=use_cache)
    if drop is not None:
        a = drop(site + 'dropout1', a, drop.p_layers)
    x = x + a
    f = feed_forward(sd, p + 'ffn.', _norm(sd, p + 'norm2.', x, cfg.norm, embedding), drop, site)
    if drop is not None:
        f = drop(site + 'dropout2', f, drop.p_layers)
    x = x + f
    return x, kv


def transformer(sd, p, x, cfg, padding_mask=None, attn_mask=None, embedding=None, kv_cache=None,
                use_cache=False, drop=None):
    """valle/models/modules.py:305-352 — L layers; with a cache keep the last row and drop the
    mask; collect the per-layer (k,v) tuple when use_cache."""
    new_kv = ()
    if use_cache and kv_cache is not None:
        x = x[:, -1:]
        attn_mask = None
    else:
        kv_cache = (None,) * cfg.num_layers
    for i in range(cfg.num_layers):
        x, kv = encoder_layer(sd, f'{p}layers.{i}.', x, cfg, padding_mask=padding_mask,
                              attn_mask=attn_mask, embedding=embedding, kv_cache=kv_cache[i],
                              use_cache=use_cache, drop=drop, site=f'layer{i}.')
        if use_cache:
            new_kv = new_kv + (kv,)
    return x, new_kv


# --------------------------------------------------------------------------------------------
# sampling (valle/models/utils.py:46-88 + transformers 4.38.2 top_k_top_p_filtering)
# --------------------------------------------------------------------------------------------
def _top_k_top_p_filter(logits, top_k=0, top_p=1.0, min_tokens_to_keep=1):
    """Published algorithm of transformers==4.38.2 `top_k_top_p_filtering` (TopKLogitsWarper then
    TopPLogitsWarper, filter value -inf): top-k removes scores strictly below the k-th largest
    (ties kept); top-p sorts ascending and removes entries whose cumulative probability is
    <= 1 - top_p, always keeping the last `min_tokens_to_keep`."""
    if top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        kth = torch.topk(logits, k)[0][..., -1, None]
        logits = logits.masked_fill(logits < kth, NEG_INF)
    if 0 <= top_p <= 1.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove_sorted = cum <= (1 - top_p)
        remove_sorted[..., -min_tokens_to_keep:] = 0
        remove = remove_sorted.scatter(1, sorted_idx, remove_sorted)
        logits = logits.masked_fill(remove, NEG_INF)
    return logits


def topk_sampling(logits, top_k=50, tok_p=1.0, temperature=1.0):
    """valle/models/utils.py:46-68 — returns (token (B,1) int64, logprob (B,))."""
    if temperature is not None:
        logits = logits / temperature
    logits = _top_k_top_p_filter(logits, top_k=top_k, top_p=tok_p)
    token = torch.multinomial(F.softmax(logits, dim=-1), num_samples=1)
    logprobs = F.log_softmax(logits, dim=-1)
    return token, logprobs[torch.arange(logits.shape[0]), token.squeeze(1)]


def get_best_beam(x, sum_logprobs, stop_token, length_penalty=1.0):
    """valle/models/utils.py:71-88 — beam with the best length-normalised log-prob, EOS stripped."""
    length = torch.sum(x != stop_token, dim=-1)
    best = x[torch.argmax(sum_logprobs / length**length_penalty), :]
    return best[best != stop_token]


# --------------------------------------------------------------------------------------------
# ValleAR (valle/models/valle_ar.py)
# --------------------------------------------------------------------------------------------
def ar_logits(sd, cfg, batch, drop=None):
    """valle/models/valle_ar.py:54-83 — teacher-forced forward; returns logits (B, V_a+1, Ty).  `drop`: a Dropout
    (train mode); None = eval."""
    tokens_lens, codes_lens = batch['tokens_lens'], batch['codes_lens']
    tx, ty = int(max(tokens_lens)), int(max(codes_lens))
    tokens = add_position(embed(sd['tokens_emb.word_embeddings.weight'], batch['tokens']),
                          sd['tokens_position_emb.pe'], drop, 'tokens_position_emb.dropout')
    codes = add_position(embed(sd['audio_emb.word_embeddings.weight'], batch['codes']),
                         sd['audio_position_emb.pe'], drop, 'audio_position_emb.dropout')
    padding_mask = F.pad(build_pad_mask(codes_lens), (tx, 0), value=False)
    attn_mask = build_attn_mask(tx, ty)
    out, _ = transformer(sd, 'transformer.', torch.cat((tokens, codes), dim=1), cfg,
                         padding_mask=padding_mask, attn_mask=attn_mask, drop=drop)
    return F.linear(out[:, tx:], sd['proj.weight']).permute(0, 2, 1)


def ar_training_loss(sd, cfg, batch, drop=None):
    """valle/models/valle_ar.py:86 — mean CE over ALL (B,Ty) positions, pads included."""
    return F.cross_entropy(ar_logits(sd, cfg, batch, drop), batch['target'])


def ar_generate(sd, cfg, prompt_tokens, prompt_codes, target_tokens=None, trace=None):
    """valle/models/valle_ar.py:92-180 — the decode hot loop (use_kv_cache=True only, defect D2).
    `trace`, if a dict, receives per-step 'logits' (B,V) of beam rows, 'tokens' and 'margin'
    (top-1 minus top-2 logit of row 0) so tests can tell a true divergence from a near-tie."""
    assert prompt_tokens.dim() == 1 and prompt_codes.dim() == 2
    assert cfg.use_kv_cache, 'reference generate() only works with use_kv_cache=True (D2)'
    eos, bos, beams = cfg.num_audio_tokens, cfg.num_audio_tokens + 1, cfg.num_beams
    codes0 = F.pad(prompt_codes[..., 0], (1, 0), value=bos).unsqueeze(0)
    prompt_len = codes0.shape[1]
    text = prompt_tokens if target_tokens is None else torch.cat((prompt_tokens, target_tokens))
    tokens_len = text.shape[0]
    tokens = add_position(embed(sd['tokens_emb.word_embeddings.weight'], text.unsqueeze(0)),
                          sd['tokens_position_emb.pe'])
    attn_mask = build_attn_mask(tokens_len, prompt_len)
    kv_cache = None
    sum_logprobs = torch.zeros(beams)
    tokens = tokens.repeat(beams, 1, 1)
    codes0 = codes0.repeat(beams, 1)
    if trace is not None:
        trace.update(logits=[], tokens=[], margin=[])
    for _ in range(cfg.max_audio_len):
        codes = add_position(embed(sd['audio_emb.word_embeddings.weight'], codes0),
                             sd['audio_position_emb.pe'])
        y, kv_cache = transformer(sd, 'transformer.', torch.cat([tokens, codes], dim=1), cfg,
                                  attn_mask=attn_mask, kv_cache=kv_cache, use_cache=True)
        logits = F.linear(y, sd['proj.weight'])[:, -1]
        samples, lp = topk_sampling(logits, top_k=cfg.top_k, tok_p=cfg.tok_p,
                                    temperature=cfg.temperature)
        if trace is not None:
            top2 = torch.topk(logits[0], 2)[0]
            trace['logits'].append(logits.clone())
            trace['margin'].append(float(top2[0] - top2[1]))
        sum_logprobs += lp * (codes0[:, -1] != eos)
        samples[codes0[:, -1] == eos] = eos
        if trace is not None:
            trace['tokens'].append(samples[:, 0].clone())
        if (samples[:, -1] == eos).all():
            break
        codes0 = torch.cat([codes0, samples], dim=1)
    best = get_best_beam(codes0, sum_logprobs, eos, cfg.length_penalty)
    best = best[prompt_len:]
    return best[best != eos]


# --------------------------------------------------------------------------------------------
# ValleNAR (valle/models/valle_nar.py) — `_prepare_audio_codes` follows the reference exactly;
# training_step/generate raise in the reference (defects D4/D5), so `nar_*` below implement the
# INTENDED algorithm of SURVEY.md §3.4 and are pinned at sub-expression level.
# --------------------------------------------------------------------------------------------
def nar_prepare_audio_codes(sd, cfg, codes, nar_stage):
    """valle/models/valle_nar.py:167-188 — prefix (first min(T//3, 3*qf) frames) embedded with all
    Q codebooks, the rest with codebooks < nar_stage; returns ((B,T,d), prefix_len)."""
    _, codes_len, q = codes.shape
    qf = cfg.sampling_rate // cfg.polling_factor
    prefix_len = min(codes_len // 3, 3 * qf)
    tab = [sd[f'codes_embs.{j}.word_embeddings.weight'] for j in range(q)]
    prompt = embed(tab[0], codes[:, :prefix_len, 0])
    rest = embed(tab[0], codes[:, prefix_len:, 0])
    for j in range(1, q):
        prompt += embed(tab[j], codes[:, :prefix_len, j])
        if j < nar_stage:
            rest += embed(tab[j], codes[:, prefix_len:, j])
    return torch.concat((prompt, rest), dim=1), prefix_len


def nar_stage_logits(sd, cfg, batch, stage, drop=None):
    """Intended forward of valle/models/valle_nar.py:71-100 for a given stage (1..Q-1): logits
    (B, T-prefix, V_a) for codebook `stage` of the non-prefix frames.  Key padding is NOT applied
    (attn_mask is None → defect D6 drops it), matching what the reference's Transformer does."""
    tx = int(batch['tokens_lens'].max())
    tokens = add_position(embed(sd['tokens_emb.word_embeddings.weight'], batch['tokens']),
                          sd['tokens_position_emb.pe'], drop, 'tokens_position_emb.dropout')
    y, prefix_len = nar_prepare_audio_codes(sd, cfg, batch['codes'], stage)
    y = add_position(y, sd['audio_position_emb.pe'], drop, 'audio_position_emb.dropout')
    pad = F.pad(build_pad_mask(batch['codes_lens']), (tx, 0), value=False)
    z, _ = transformer(sd, 'transformer.', torch.cat([tokens, y], dim=1), cfg, padding_mask=pad,
                       embedding=sd[f'stage_embs.{stage - 1}.word_embeddings.weight'], drop=drop)
    return F.linear(z[:, tx + prefix_len:], sd[f'proj_layers.{stage - 1}.weight']), prefix_len


def nar_training_loss(sd, cfg, batch, stage, drop=None):
    """Intended loss (valle_nar.py:81,103): CE of stage logits vs raw ids codes[:, prefix:, stage],
    mean over all positions (same convention as the AR loss)."""
    logits, prefix_len = nar_stage_logits(sd, cfg, batch, stage, drop)
    return F.cross_entropy(logits.permute(0, 2, 1), batch['codes'][:, prefix_len:, stage])


def nar_generate(sd, cfg, prompt_tokens, prompt_codes, target_tokens, target_codes_first_layer,
                 greedy=True, generator=None):
    """Intended algorithm of valle/models/valle_nar.py:107-165: for stage n the
    target frames carry sum_{j<n} codes_embs[j](out[j]); returns (Ty, Q) int64.  greedy=True takes
    the argmax (build-defined, for parity tests); otherwise Categorical sampling as the reference."""
    q = cfg.num_quantizers
    tab = [sd[f'codes_embs.{j}.word_embeddings.weight'] for j in range(q)]
    emb_prompt = sum(embed(tab[j], prompt_codes[:, j]) for j in range(q))
    text = torch.cat([prompt_tokens, target_tokens]).unsqueeze(0)
    tx, tc = text.shape[1], prompt_codes.shape[0]
    tokens = add_position(embed(sd['tokens_emb.word_embeddings.weight'], text),
                          sd['tokens_position_emb.pe'])
    out = [target_codes_first_layer]
    emb_out = torch.zeros(target_codes_first_layer.shape[0], cfg.d_model)
    for n in range(1, q):
        emb_out = emb_out + embed(tab[n - 1], out[n - 1])
        codes = add_position(torch.cat([emb_prompt, emb_out], dim=0).unsqueeze(0),
                             sd['audio_position_emb.pe'])
        z, _ = transformer(sd, 'transformer.', torch.cat([tokens, codes], dim=1), cfg,
                           embedding=sd[f'stage_embs.{n - 1}.word_embeddings.weight'])
        logits = F.linear(z[0, tx + tc:], sd[f'proj_layers.{n - 1}.weight'])
        if greedy:
            out.append(torch.argmax(logits, dim=-1))
        else:
            probs = F.softmax(logits / cfg.temperature, dim=-1)
            out.append(torch.multinomial(probs, 1, generator=generator).squeeze(1))
    return torch.stack(out, dim=1)
