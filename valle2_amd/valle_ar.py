"""ValleAR with the reference's constructor / training_step / generate / configure_optimizers
signatures and state_dict keys (valle/models/valle_ar.py:14-194), running on MI355X.

`generate()` does not walk the module tree per step as the reference does: the prompt is embedded
and prefilled once by the native forward composite (analytic prefix-LM mask, K/V written straight
into a preallocated cache) and every further token is one replay of a hipGraph holding the whole
decode step (engine.ArDecoder).  EOS is polled every `EOS_POLL` steps instead of a host sync per
step (valle_ar.py:169-170).
"""
from __future__ import annotations

import os
import threading
import time

import torch
import torch.nn as nn
from torch import optim

from . import _lib, dropout, kernels
from .engine import (ArDecoder, ForwardScratch, ForwardScratch16, KVCache, StepSampler, perf_forward_supported,
                     shared_prompt_fits,
                     transformer_forward, transformer_forward_bf16)
from .modules import PositionalEncoding, TokenEmbedding, Transformer, _on_device, device_mirror
from .utils import get_best_beam

try:  # the reference subclasses lightning.LightningModule; lightning is optional here
    import lightning as L
    _Base = L.LightningModule
except Exception:  # pragma: no cover - lightning is absent in this image
    class _Base(nn.Module):
        def log(self, *args, **kwargs):
            return None

EOS_POLL = 32
MAX_DECODE_ROWS = 64      # rows per decode launch (vh_ar_decoder: 1..64)
# generate(): the beams of one utterance share its prompt K/V (read once per step for all beams).  VALLE2_SHARED_PROMPT=0
# decodes the beams as independent rows (round 4's form: the A/B arm, and what generate_batch does for distinct rows).
SHARED_PROMPT = os.environ.get('VALLE2_SHARED_PROMPT', '1') != '0'


class _Run:
    """Shapes and modes of one generate_batch call, handed between its helpers."""


class _DecodeSlot:
    """Everything of a generate_batch call that a captured decode graph points at, kept per SHAPE on the model so that the
    next call of the same shape neither allocates, nor builds a decoder, nor captures (~1.7 ms of capture +
    the construction per call, which a 2 ms prompt pass no longer hides): the token buffer, the K/V caches, the per-row
    counters and the ArDecoder with its graphs and workspaces.  A slot is used by one call at a time (`busy`)."""

    def __init__(self):
        self.codes = self.cache = self.prefix = self.cache_len = self.audio_pos = self.pos_base = self.dec = None
        self.busy = False
        self.uses = 0

    def close(self):
        if self.dec is not None:
            self.dec.close()
            self.dec = None


_DECODER_ENV = ('VALLE2_HEAD_FUSED', 'VALLE2_SHARED_SPLIT', 'VALLE2_FOLD_LN', 'VALLE2_DECODE_W16')   # environment knobs read when a decoder is built
DECODER_SLOTS = int(os.environ.get('VALLE2_DECODER_SLOTS', '2'))    # decoders kept per model (0: build one per call, as before)
_SLOT_LOCK = threading.Lock()


class ValleAR(_Base):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.tokens_emb = TokenEmbedding(config.vocab_size, config.d_model)
        self.audio_emb = TokenEmbedding(config.num_audio_tokens + 2, config.d_model)
        self.tokens_position_emb = PositionalEncoding(config.d_model)
        self.audio_position_emb = PositionalEncoding(config.d_model)
        self.transformer = Transformer(config)
        self.proj = nn.Linear(config.d_model, config.num_audio_tokens + 1, bias=False)
        self.last_generate_stats: dict = {}

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def eos_token(self):
        return self.config.num_audio_tokens

    @property
    def bos_token(self):
        return self.config.num_audio_tokens + 1

    # ------------------------------------------------------------------------------------
    def _require_layernorm(self):
        if self.config.norm != 'LayerNorm':
            # reference defect D3: with AdaptiveLayerNorm the AR model passes embedding=None and
            # dies in Linear(None) with a TypeError; keep the failure, say why.
            raise TypeError("ValleAR needs norm='LayerNorm' (AdaptiveLayerNorm has no stage "
                            'embedding in the AR model; the reference raises TypeError too)')

    def _embed_rows(self, text_ids, codes_ids, x, x_t0=0):
        """x[:, x_t0:x_t0+Tx] = tokens_emb + PE; x[:, x_t0+Tx:] = audio_emb + PE (positions restart
        at 0 for the audio stream, valle_ar.py:61-66)."""
        tx = text_ids.shape[1]
        kernels.embed_sum_pe(text_ids, [self.tokens_emb.weight.detach()], self.tokens_position_emb.pe,
                             0, x, out_t0=x_t0)
        kernels.embed_sum_pe(codes_ids, [self.audio_emb.weight.detach()], self.audio_position_emb.pe,
                             0, x, out_t0=x_t0 + tx)

    @_on_device
    def forward_logits(self, batch, perf_mode: bool = False):
        """Teacher-forced logits (B, Ty, V_a+1) of valle_ar.py:54-83 (row-major, before the
        reference's rearrange to (B, V, Ty)).  A model that lives on the CPU computes through its
        device mirror (modules.device_mirror); the logits stay on the HIP device.
        perf_mode=True (opt-in, SECONDARY): the stack on the bf16 matrix cores (engine.transformer_forward_bf16); logits
        agree with the reference to 5e-2 instead of 2e-4."""
        self._require_layernorm()
        dev = self.device
        tokens = kernels.ids_to_device(batch['tokens'], dev, self.config.vocab_size, 'tokens')
        codes = kernels.ids_to_device(batch['codes'], dev, self.config.num_audio_tokens + 2, 'codes')
        codes_lens = batch['codes_lens']
        tx, ty = int(max(batch['tokens_lens'])), int(max(codes_lens))
        b = tokens.shape[0]
        d = self.config.d_model
        x = torch.empty(b, tx + ty, d, device=dev, dtype=torch.float32)
        self._embed_rows(tokens[:, :tx], codes[:, :ty], x)
        # key padding covers audio only; text padding is NOT masked (valle_ar.py:69-73)
        kv_len = _lib.to_device_async(codes_lens.to(torch.int64) + tx, dev, torch.int32)
        if perf_mode:
            cache = KVCache(self.config.num_layers, b, self.config.n_heads, tx + ty, dev, dtype=kernels.H16)
            transformer_forward_bf16(self.transformer, x, cache, mode=kernels.MASK_PREFIX, x_len=tx, kv_len=kv_len)
        else:
            cache = KVCache(self.config.num_layers, b, self.config.n_heads, tx + ty, dev)
            transformer_forward(self.transformer, x, cache, mode=kernels.MASK_PREFIX, x_len=tx, kv_len=kv_len)
        out = x[:, tx:].reshape(b * ty, d)
        logits = kernels.linear(out, self.proj.weight.detach())
        return logits.reshape(b, ty, -1)

    def _logits_with_graph(self, batch):
        """The same forward as `forward_logits`, composed from autograd Functions so that
        `loss.backward()` reaches every parameter (valle_ar.py:61-83)."""
        from . import autograd as A
        self._require_layernorm()
        dev = self.device
        if dev.type != 'cuda':
            raise _lib.VhError('ValleAR.training_step with gradients needs the model on its HIP device '
                               '(model.to("cuda")): gradients cannot flow into a CPU copy of the parameters')
        tokens = kernels.ids_to_device(batch['tokens'], dev, self.config.vocab_size, 'tokens')
        codes = kernels.ids_to_device(batch['codes'], dev, self.config.num_audio_tokens + 2, 'codes')
        codes_lens = batch['codes_lens']
        tx, ty = int(max(batch['tokens_lens'])), int(max(codes_lens))
        b, d = tokens.shape[0], self.config.d_model
        # PE dropout p = 0.1 is live in train mode whatever config.dropout says (D9).  Both streams' embeddings are written
        # into ONE buffer (no torch.cat, no strided copies of its gradient) and each part's dropout is a field applied by
        # the gather kernel itself before it stores the row (dropout.py) — the backward regenerates it in the scatter
        seed = dropout.seed_if(dropout.live(self.tokens_position_emb.dropout), dropout.live(self.audio_position_emb.dropout))
        dr_t = dropout.spec(seed, dropout.site(dropout.PE_TEXT), dropout.live(self.tokens_position_emb.dropout))
        dr_a = dropout.spec(seed, dropout.site(dropout.PE_AUDIO), dropout.live(self.audio_position_emb.dropout))
        dropout.record('tokens_position_emb.dropout', dr_t, b * (tx + ty), d)
        dropout.record('audio_position_emb.dropout', dr_a, b * (tx + ty), d)
        x = A.EmbedConcatFn.apply([(tokens[:, :tx], self.tokens_position_emb.pe, 0, [0], dr_t),
                                   (codes[:, :ty], self.audio_position_emb.pe, 0, [1], dr_a)],
                                  self.tokens_emb.weight, self.audio_emb.weight)
        x = x.reshape(b * (tx + ty), d)
        kv_len = _lib.to_device_async(codes_lens.to(torch.int64) + tx, dev, torch.int32)
        spec = dict(mode=kernels.MASK_PREFIX, x_len=tx, kv_len=kv_len)
        x = A.transformer_train(self.transformer, x, b, tx + ty, spec)
        out = x.view(b, tx + ty, d)[:, tx:].reshape(b * ty, d)
        return A.linear(out, self.proj.weight).reshape(b, ty, -1)

    def training_step(self, batch, **kwargs):
        """valle_ar.py:43-90: mean cross entropy over ALL (B, Ty) positions, pads included.  With
        grad mode on the loss carries a full autograd graph (hand-written HIP kernels forward and
        backward: tile GEMMs, TN weight-gradient GEMM, flash attention backward, row kernels —
        valle2_amd/autograd.py); under no_grad it takes the fused inference kernels."""
        from . import autograd as A
        if torch.is_grad_enabled():
            logits = self._logits_with_graph(batch)
        else:
            logits = self.forward_logits(batch)
        target = kernels.ids_to_device(batch['target'], logits.device, self.config.num_audio_tokens + 1, 'target')
        rows = logits.shape[0] * logits.shape[1]
        loss = A.CrossEntropyFn.apply(logits.reshape(rows, -1), target[:, : logits.shape[1]].reshape(rows))
        self.log('train/loss', loss)
        return loss

    # ------------------------------------------------------------------------------------
    @_on_device
    @torch.inference_mode()
    def generate(self, prompt_tokens, prompt_codes, target_tokens=None):
        """valle_ar.py:92-180 — one utterance replicated over `num_beams` rows; returns the 1-D
        int64 first-codebook tokens of the best beam with EOS stripped.

        The beams share one prompt, so (SHARED_PROMPT, default on) the prompt pass runs for ONE row and its K/V are read
        once per decode step for all beams (`generate_batch(..., shared_prompt=True)`); the beams themselves — their
        sampled tokens, their own K/V rows, the per-beam log-probabilities — are never deduplicated."""
        assert prompt_tokens.dim() == 1, 'Prompt tokens should be 1D tensor.'
        assert prompt_codes.dim() == 2, 'Prompt codes should be 2D tensor.'
        if target_tokens is not None:
            assert target_tokens.dim() == 1, 'Target tokens should be 1D tensor.'
        beams = self.config.num_beams
        text = prompt_tokens if target_tokens is None else torch.cat((prompt_tokens, target_tokens), dim=0)
        shared = SHARED_PROMPT and self.config.use_kv_cache and self.config.d_model == self.config.n_heads * kernels.HEAD_DIM
        # (a prompt beyond the shared kernel's record bound — 7680 keys at 4 beams x 8 heads — decodes as independent rows)
        shared = shared and shared_prompt_fits(beams, self.config.n_heads, int(text.shape[0]) + int(prompt_codes.shape[0]) + 1)   # + BOS
        rows = self.generate_batch([text] * beams, [prompt_codes[..., 0]] * beams, shared_prompt=shared and beams <= MAX_DECODE_ROWS)
        # beams → one sequence (valle_ar.py:174-180); with top_k=1 every log-prob is exactly 0
        sum_logprobs = self.last_generate_stats['sum_logprobs']
        prompt_len = prompt_codes.shape[0] + 1
        best = get_best_beam(rows, sum_logprobs, self.eos_token, self.config.length_penalty)
        best = best[prompt_len:]
        return best[best != self.eos_token]

    def _generate_in_groups(self, texts, first_codes, max_new, use_graph, perf_mode):
        """More rows than one decode launch serves (64: 4 MFMA row tiles): consecutive groups of 64 rows; rows are
        independent, so the result is what one pass would give."""
        B, dev = len(texts), self.device
        parts, stats = [], []
        for r0 in range(0, B, MAX_DECODE_ROWS):
            parts.append(self.generate_batch(texts[r0:r0 + MAX_DECODE_ROWS], first_codes[r0:r0 + MAX_DECODE_ROWS],
                                             max_new=max_new, use_graph=use_graph, perf_mode=perf_mode))
            stats.append(self.last_generate_stats)
        width = max(p.shape[1] for p in parts)
        out = torch.full((B, width), self.eos_token, device=dev, dtype=torch.int64)
        r = 0
        for p in parts:
            out[r:r + p.shape[0], :p.shape[1]] = p
            r += p.shape[0]
        merged = dict(stats[-1])
        merged['prompt_lens'] = [x for st in stats for x in st['prompt_lens']]
        merged['sum_logprobs'] = torch.cat([st['sum_logprobs'] for st in stats])
        merged['tokens_appended'] = max(st['tokens_appended'] for st in stats)
        self.last_generate_stats = merged
        return out

    def _weights_key(self):
        """Changes whenever a pointer or a value the decoder's tables were built from may have changed."""
        from . import engine
        return (engine._WEIGHTS_EPOCH,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _acquire_slot(self, key):
        """The free slot of this shape (LRU order), or a new one (the oldest free slot beyond DECODER_SLOTS is dropped).
        None when slots are off."""
        if DECODER_SLOTS <= 0:
            return None
        with _SLOT_LOCK:
            slots = self.__dict__.setdefault('_decode_slots', {})
            slot = slots.pop(key, None)
            if slot is not None and slot.busy:              # another host thread is decoding this shape right now
                slots[key] = slot
                return None
            if slot is None:
                slot = _DecodeSlot()
                free = [k for k, v in slots.items() if not v.busy]
                while len(slots) >= DECODER_SLOTS and free:
                    slots.pop(free.pop(0)).close()
            slot.busy = True
            slots[key] = slot                                # most recently used last
            return slot

    def _release_slot(self, key, slot, ok):
        if slot is None:
            return
        with _SLOT_LOCK:
            slot.busy = False
            if not ok:                                       # a failed call leaves nothing behind
                self.__dict__.get('_decode_slots', {}).pop(key, None)
                slot.close()

    def release_decoders(self):
        """Free the decoders (graphs, K/V caches, workspaces) kept from earlier generate() calls."""
        with _SLOT_LOCK:
            for slot in self.__dict__.pop('_decode_slots', {}).values():
                if not slot.busy:
                    slot.close()

    def _prompt_pass(self, run, texts, first_codes, codes):
        """Step 0 (valle_ar.py:143-155 at kv_cache=None): embed and run the whole prompt.  Row b is laid out
        [text_b | BOS + prompt_b | padding]; the prefix-LM mask takes per-row lengths.  Returns the K/V cache the decode
        steps continue on (None without one), the shared-prompt prefix cache (or None) and the last hidden row of every
        decode row.  `run`: the _Run record of generate_batch (shapes, modes)."""
        cfg, dev, d = self.config, self.device, self.config.d_model
        B, s0, s_max = run.B, run.s0, run.s_max
        i32 = dict(device=dev, dtype=torch.int32)
        prefix = None
        slot = getattr(run, 'slot', None)
        if slot is not None and slot.cache is not None:
            cache, prefix = slot.cache, slot.prefix          # the buffers this shape's captured graphs point at
        elif run.any_head_dim:
            cache = None
        elif run.shared:
            # ONE row through the prompt pass: its K/V are the prefix every beam reads; the beams' cache holds generated rows only
            prefix = KVCache(cfg.num_layers, 1, cfg.n_heads, (s0 + 31) // 32 * 32, dev)
            cache = KVCache(cfg.num_layers, B, cfg.n_heads, (run.max_new + 1 + 31) // 32 * 32, dev)
        elif run.perf_prefill:
            cache = KVCache(cfg.num_layers, B, cfg.n_heads, s_max, dev, dtype=kernels.H16)
        else:
            cache = KVCache(cfg.num_layers, B, cfg.n_heads, s0 if run.perf_mode else s_max, dev)
        rows = 1 if run.shared else B
        if not run.ragged:
            run.text_ids = torch.stack(texts[:rows])
            codes[:, 1:run.pl_max] = torch.stack(first_codes)
            x = torch.empty(rows, s0, d, device=dev, dtype=torch.float32)
            self._embed_rows(run.text_ids, codes[:rows, :run.pl_max], x)
            run.fwd = dict(x_len=run.txs[0])
        else:
            x = torch.zeros(B, s0, d, device=dev, dtype=torch.float32)
            for b in range(B):
                codes[b, 1:run.pls[b]] = first_codes[b]
                self._embed_rows(texts[b].unsqueeze(0), codes[b:b + 1, :run.pls[b]], x[b:b + 1])
            run.lens = torch.tensor([t + p for t, p in zip(run.txs, run.pls)], **i32)
            run.fwd = dict(x_len_dev=torch.tensor(run.txs, **i32), kv_len=run.lens)
        if run.perf_prefill:
            transformer_forward_bf16(self.transformer, x, cache, mode=kernels.MASK_PREFIX,
                                     scratch=ForwardScratch16(B * s0, d, cfg.dim_feedforward, dev), **run.fwd)
        else:
            scratch = None if run.any_head_dim else ForwardScratch(rows * s0, d, cfg.dim_feedforward, dev)
            transformer_forward(self.transformer, x, prefix if run.shared else cache, mode=kernels.MASK_PREFIX,
                                scratch=scratch, **run.fwd)
        if run.ragged:
            last = x[torch.arange(B, device=dev), run.lens.long() - 1]
        elif run.shared:
            last = x[:, -1].expand(B, d)                  # every beam starts from the one prompt row's last hidden state
        else:
            last = x[:, -1]
        if run.perf_mode and not run.perf_prefill:
            cache = cache.narrowed(s_max)                 # fp32 prompt K/V -> the bf16 cache of the decode steps
        return cache, prefix, last.contiguous()

    def _decode_forced(self, run, dec, codes, forced, keep_logits):
        """TEACHER FORCING (tolerance tests): after every step replace the sampled token and its embedding by the given one;
        returns the logits the head produced at the steps listed in keep_logits."""
        dev, d = self.device, self.config.d_model
        forced = forced.to(dev)
        if run.ragged or forced.numel() < run.max_new:
            raise ValueError('forced: one token per step, equal-length rows')
        pe, kept = self.audio_position_emb.pe, {}
        for t in range(run.max_new):
            if t:
                dec.run(1)
            if t in keep_logits:
                kept[t] = dec.logits[:, : dec.V].clone()
            codes[:, run.pl_max + t] = forced[t]
            kernels.embed_sum_pe(codes[:, run.pl_max + t:run.pl_max + t + 1], [self.audio_emb.weight.detach()], pe,
                                 run.pl_max + t, dec.x.view(run.B, 1, d))
        return kept

    def _decode_recompute(self, run, dec, texts, codes, cache):
        """config.use_kv_cache = False (valle_ar.py:132,150-155 — the reference's branch raises, D2; build-defined here as
        what the flag says): every step embeds the WHOLE sequence again and runs the full stack over it under the prefix
        mask — no state is carried from step to step except the tokens — and samples from its last row with the same head /
        sample kernels.  O(S^2) per token; it exists so that the flag works and as an independent check of the cached
        decoder (same tokens, tests/test_models_gpu.py).  Returns the number of steps run."""
        cfg, dev, d = self.config, self.device, self.config.d_model
        B, s0 = run.B, run.s0
        scratch = None if run.any_head_dim else ForwardScratch(B * (s0 + run.max_new), d, cfg.dim_feedforward, dev)
        rows_idx = torch.arange(B, device=dev)
        done = 1
        while done < run.max_new:
            t = done
            xs = (torch.zeros if run.ragged else torch.empty)(B, s0 + t, d, device=dev, dtype=torch.float32)
            if not run.ragged:
                self._embed_rows(run.text_ids, codes[:, :run.pl_max + t], xs)
                transformer_forward(self.transformer, xs, cache, mode=kernels.MASK_PREFIX, scratch=scratch, **run.fwd)
                last = xs[:, -1]
            else:
                for b in range(B):
                    self._embed_rows(texts[b].unsqueeze(0), codes[b:b + 1, :run.pls[b] + t], xs[b:b + 1])
                transformer_forward(self.transformer, xs, cache, mode=kernels.MASK_PREFIX, scratch=scratch,
                                    x_len_dev=run.fwd['x_len_dev'], kv_len=run.lens + t)
                last = xs[rows_idx, run.lens.long() + t - 1]
            dec.sample_from(last.contiguous())
            done += 1
            if done % EOS_POLL == 0 and bool((dec.eos_count[:done] == B).any()):
                break
        return done

    @staticmethod
    def _decode_cached(run, dec, done):
        """Steps done .. max_new-1 on the cached decoder, EOS polled every EOS_POLL steps (valle_ar.py:169-170 breaks when
        every beam has emitted EOS).  Returns (steps run, step at which every row had finished or None)."""
        while done < run.max_new:
            n = min(EOS_POLL, run.max_new - done)
            dec.run(n)
            done += n
            full = (dec.eos_count[:done] == run.B).nonzero()
            if full.numel():
                return done, int(full[0])
        return done, None

    @_on_device
    @torch.inference_mode()
    def generate_batch(self, texts, first_codes, max_new=None, use_graph=True, profile_attn=False, perf_mode=False,
                       forced=None, keep_logits=(), shared_prompt=False):
        """Batched greedy decoding of B independent rows (extension; `generate` is built on it).
        texts[b]: 1-D int64 text ids; first_codes[b]: 1-D int64 first-codebook prompt (no BOS).
        Rows may differ in text and prompt length.  Returns codes (B, max_prompt_len + n_new) int64
        on the device: row b holds BOS + prompt_b + its n_new generated tokens from index 0 (finished
        rows and the tail of shorter rows are EOS-filled; `last_generate_stats['prompt_lens'][b]` is
        where row b's generated tokens start).
        profile_attn=True runs the steps eagerly with HIP events around every decode-attention
        launch and leaves their mean duration in `last_generate_stats` (measurement only).
        perf_mode=True (opt-in): the prompt pass runs on the bf16 matrix cores (bf16 operands, fp32
        accumulators and residual stream; engine.transformer_forward_bf16) and writes its K/V straight into the bf16 cache
        the decode steps stream; the decode steps' weights and arithmetic stay fp32.  perf_mode='kv': only the cache is bf16
        (the fp32 prompt pass, its K/V narrowed once — round 3's form).  Greedy tokens are then NOT guaranteed to be the
        reference's (teacher-forced logits agree to 5e-2).
        forced (max_new,) int64 + keep_logits (step indices): TEACHER FORCING for the tolerance tests — step t appends
        forced[t] whatever the head says (steps run eagerly, one at a time) and the logits (B, V) the head produced at
        the steps listed in keep_logits are left in `last_generate_stats['logits']`.
        shared_prompt=True: the caller vouches that every row has the SAME text and prompt (the beams of one utterance,
        valle_ar.py:135-138; checked: equal lengths and equal ids) — the prompt pass then runs for one row and every decode
        step reads the prompt's K/V once for all rows (vh_attn_decode_shared); rows still sample, append and score
        independently.  fp32 (no perf_mode), cached decoder only."""
        self._require_layernorm()
        cfg = self.config
        dev = self.device
        B = len(texts)
        if B == 0 or len(first_codes) != B:
            raise ValueError('generate_batch: texts and first_codes must be non-empty lists of equal length')
        run = _Run()
        run.B, run.max_new = B, cfg.max_audio_len if max_new is None else max_new
        # a head width other than 64 (modules.py:109-111 allows it; no configuration of the path has it): the native
        # decoder and its KV-cache kernels are built for 64, so such a model decodes by recomputation on the general kernels
        run.any_head_dim = cfg.d_model != cfg.n_heads * kernels.HEAD_DIM
        no_cache = not cfg.use_kv_cache or run.any_head_dim
        if no_cache and (perf_mode or profile_attn or forced is not None or shared_prompt):
            raise ValueError('use_kv_cache=False (or a head width other than 64) recomputes every step from scratch: '
                             'perf_mode / profile_attn / forced / shared_prompt belong to the cached decoder')
        if B > MAX_DECODE_ROWS:
            if shared_prompt or forced is not None:
                raise ValueError(f'shared_prompt / forced serve at most {MAX_DECODE_ROWS} rows')
            return self._generate_in_groups(texts, first_codes, run.max_new, use_graph, perf_mode)
        run.txs = [int(t.shape[0]) for t in texts]
        run.pls = [int(c.shape[0]) + 1 for c in first_codes]              # BOS + prompt
        run.ragged = len(set(run.txs)) > 1 or len(set(run.pls)) > 1
        tx_max, run.pl_max = max(run.txs), max(run.pls)
        run.s0 = max(t + p for t, p in zip(run.txs, run.pls))               # longest row's context
        run.s_max = (run.s0 + run.max_new + 31) // 32 * 32   # whole 32-key chunks per (row, head) block (the ring kernel reads ahead in chunks of 32 keys)
        if run.pl_max + run.max_new > self.audio_position_emb.pe.shape[0] or tx_max > self.tokens_position_emb.pe.shape[0]:
            raise _lib.VhError('sequence exceeds the positional table (max_len 5000)')
        run.perf_mode = perf_mode
        run.perf_prefill = bool(perf_mode) and perf_mode != 'kv' and perf_forward_supported(cfg)
        run.shared = bool(shared_prompt)
        if run.shared and (perf_mode or run.ragged):
            raise ValueError('shared_prompt: identical rows, fp32')
        t_host0 = time.perf_counter()
        # a decoder per shape survives the call (graphs, caches, counters: _DecodeSlot) unless the call is one of the
        # measurement / test forms that drive the decoder by hand
        slot_key = slot = None
        if not (no_cache or forced is not None or profile_attn or (perf_mode and not run.perf_prefill)):
            slot_key = (B, run.s0 if run.shared else None, run.s_max, run.pl_max + run.max_new, run.max_new, run.shared,
                        run.perf_prefill, bool(use_graph), int(cfg.top_k), float(cfg.tok_p), float(cfg.temperature),
                        str(dev), _lib.TUNING_EPOCH, tuple(os.environ.get(k) for k in _DECODER_ENV), self._weights_key())
            slot = self._acquire_slot(slot_key)
        run.slot = slot
        reuse = slot is not None and slot.dec is not None
        if reuse:
            codes = slot.codes
            codes.fill_(self.eos_token)
        else:
            codes = torch.full((B, run.pl_max + run.max_new), self.eos_token, device=dev, dtype=torch.int64)
        codes[:, 0] = self.bos_token                                   # valle_ar.py:115-117
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(3)]   # prefill | decode phase times
        marks[0].record()
        texts = [kernels.ids_to_device(t, dev, cfg.vocab_size, 'text ids') for t in texts]
        first_codes = [kernels.ids_to_device(c, dev, cfg.num_audio_tokens, 'prompt codes') for c in first_codes]
        if run.shared and any(t is not texts[0] and not torch.equal(t, texts[0]) for t in texts[1:]) or \
                run.shared and any(c is not first_codes[0] and not torch.equal(c, first_codes[0]) for c in first_codes[1:]):
            raise ValueError('shared_prompt: every row must carry the same text and prompt ids')
        # (the decode loop's small state goes up BEFORE the prompt pass is enqueued: a host->device copy behind it
        # would hold the host until the pass has finished, and the decoder is built and captured during the pass)
        # cache_len: rows in the cache the decode steps append to (+1 by the sample step); shared prompt: generated rows only
        first_len = [-1] * B if run.shared else [t + p - 1 for t, p in zip(run.txs, run.pls)]
        if reuse:
            cache_len, audio_pos, pos_base = slot.cache_len, slot.audio_pos, slot.pos_base
            cache_len.copy_(torch.tensor(first_len, dtype=torch.int32), non_blocking=True)
            audio_pos.copy_(torch.tensor(run.pls, dtype=torch.int32), non_blocking=True)
            pos_base.copy_(audio_pos)
        else:
            cache_len = _lib.to_device_async(torch.tensor(first_len, dtype=torch.int32), dev)
            audio_pos = _lib.to_device_async(torch.tensor(run.pls, dtype=torch.int32), dev)
            pos_base = audio_pos.clone()
        # sampling seed drawn from torch's generator, so torch.manual_seed() makes a run repeatable
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if cfg.top_k != 1 else 0
        if reuse:
            slot.dec.reset(seed)                          # (before the prompt pass is enqueued: its copies do not queue behind it)
        t_host1 = time.perf_counter()
        ok = False
        try:
            cache, prefix, last = self._prompt_pass(run, texts, first_codes, codes)
        except BaseException:
            self._release_slot(slot_key, slot, False)
            raise
        t_host2 = time.perf_counter()
        if reuse:
            dec = slot.dec
        elif run.any_head_dim:
            dec = StepSampler(self, B, codes, cache_len, audio_pos, pos_base, seed=seed)
        else:
            try:
                dec = ArDecoder(self, B, cache.s_max, codes, cache, cache_len, audio_pos, pos_base,
                                use_graph=use_graph and not no_cache, seed=seed, prefix=prefix, prefix_len=run.s0)
            except BaseException:
                self._release_slot(slot_key, slot, False)
                raise
            if slot is not None:
                slot.codes, slot.cache, slot.prefix, slot.dec = codes, cache, prefix, dec
                slot.cache_len, slot.audio_pos, slot.pos_base = cache_len, audio_pos, pos_base
        if slot is not None:
            slot.uses += 1
        try:
            dec.capture()                                 # (a no-op without a graph: the no-cache path only borrows the sampler)
            t_host3 = time.perf_counter()
            dec.sample_from(last)
            marks[1].record()
            del last
            kept, done, stop = {}, 1, None
            attn_ms = attn_floor_ms = attn_kernel_ms = None
            if forced is not None:
                kept, done = self._decode_forced(run, dec, codes, forced, keep_logits), run.max_new
            elif no_cache:
                done = self._decode_recompute(run, dec, texts, codes, cache)
            elif profile_attn and run.max_new > 1:
                attn_ms, attn_floor_ms, attn_kernel_ms = dec.profile_attn(run.max_new - 1)
                done = run.max_new
            else:
                done, stop = self._decode_cached(run, dec, done)
            marks[2].record()
            if stop is None:
                full = (dec.eos_count[:done] == B).nonzero()
                stop = int(full[0]) if full.numel() else None
            n_new = run.max_new if stop is None else stop     # the all-EOS step is not appended (:169-171)
            marks[2].synchronize()
            t_host4 = time.perf_counter()
            _lib.raise_device_errors(dev)                 # ids that were already on the device: checked in-kernel
            self.last_generate_stats = {'steps_run': done, 'tokens_appended': n_new, 'n_split': dec.n_split,
                                        'ffn_fused': dec.ffn_ws is not None and cfg.d_model <= 512, 'kv_bf16': dec.kv_bf16,
                                        'decode_w16': bool(getattr(dec, 'w16', False)),
                                        'head_fused': dec.head_ws is not None,
                                        'prefill_bf16': run.perf_prefill, 'shared_prompt': run.shared, 'logits': kept,
                                        'prefill_ms': marks[0].elapsed_time(marks[1]),
                                        'decode_ms': marks[1].elapsed_time(marks[2]),
                                        'attn_mean_ms': attn_ms, 'attn_floor_ms': attn_floor_ms,
                                        'attn_kernel_ms': attn_kernel_ms, 's0': run.s0,
                                        'prompt_lens': run.pls,
                                        'sum_logprobs': dec.sum_logprobs.clone(),
                                        # host time this call spent OUTSIDE enqueueing the prompt pass and the replays and
                                        # waiting for them: set-up of the call's state + building / capturing the decoder
                                        # (nothing on a reused slot) + the tail after the last step has finished
                                        'decoder_reused': bool(reuse), 'slot_uses': slot.uses if slot is not None else 0,
                                        'host_setup_ms': (t_host1 - t_host0) * 1e3,
                                        'host_decoder_ms': (t_host3 - t_host2) * 1e3}
            out_codes = codes[:, : run.pl_max + n_new].clone()
            self.last_generate_stats['host_tail_ms'] = (time.perf_counter() - t_host4) * 1e3
            self.last_generate_stats['host_outside_ms'] = (self.last_generate_stats['host_setup_ms']
                                                           + self.last_generate_stats['host_decoder_ms']
                                                           + self.last_generate_stats['host_tail_ms'])
            ok = True
            return out_codes
        finally:
            if slot is None or dec is not slot.dec:
                dec.close()
            self._release_slot(slot_key, slot, ok)

    def configure_optimizers(self):
        """valle_ar.py:182-194"""
        from .optim import FlatAdamW           # AdamW(fused=True) as one flat HIP pass (+ clip, + 1/world)
        optimizer = FlatAdamW(self.parameters(), lr=self.config.lr, betas=self.config.betas,
                              weight_decay=self.config.weight_decay)
        scheduler = optim.lr_scheduler.CosineAnnealingWarmRestarts(optimizer, self.config.lr_warmup)
        return {'optimizer': optimizer, 'lr_scheduler': scheduler}
