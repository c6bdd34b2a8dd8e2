"""ValleNAR with the reference's constructor / method signatures and state_dict keys
(valle/models/valle_nar.py:17-188), running on MI355X.

`_prepare_audio_codes` follows the reference exactly.  `training_step` and `generate` RAISE in the
reference (defects D4/D5); they implement the intended algorithm of SURVEY.md §3.4
here: stage n predicts codebook n from the sum of codebooks < n (all Q for the acoustic prompt),
AdaLN conditioned on stage_embs[n-1], full attention, head proj_layers[n-1].
"""
from __future__ import annotations

import random

import torch
import torch.nn as nn

from . import _lib, dropout, kernels
from .engine import ForwardScratch, ForwardScratch16, KVCache, head16, transformer_forward, transformer_forward_bf16
from .modules import PositionalEncoding, TokenEmbedding, Transformer, _on_device
from .valle_ar import _Base


class ValleNAR(_Base):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.eos_token = config.num_audio_tokens
        self.bos_token = config.num_audio_tokens + 1
        self.tokens_emb = TokenEmbedding(config.vocab_size, config.d_model)
        self.codes_embs = nn.ModuleList(
            [TokenEmbedding(config.num_audio_tokens, config.d_model) for _ in range(config.num_quantizers)])
        self.tokens_position_emb = PositionalEncoding(config.d_model)
        self.audio_position_emb = PositionalEncoding(config.d_model)
        self.stage_embs = nn.ModuleList(
            [TokenEmbedding(1, config.d_model) for _ in range(config.num_quantizers - 1)])
        self.transformer = Transformer(config)
        self.proj_layers = nn.ModuleList(
            [nn.Linear(config.d_model, config.num_audio_tokens, bias=False)
             for _ in range(config.num_quantizers - 1)])

    @property
    def device(self):
        return next(self.parameters()).device

    def _dev(self):
        dev = self.device
        if dev.type != 'cuda':     # inference entry points never get here (they run on the device mirror)
            raise _lib.VhError('ValleNAR.training_step with gradients needs the model on its HIP device '
                               '(model.to("cuda")): gradients cannot flow into a CPU copy of the parameters')
        return dev

    def _tables(self, n):
        return [self.codes_embs[j].weight.detach() for j in range(n)]

    def prefix_len_of(self, codes_len: int) -> int:
        """3 s of audio or a third of it, whichever is shorter (valle_nar.py:179)."""
        return min(codes_len // 3, 3 * self.config.quantization_factor)

    def _embed_audio(self, codes, nar_stage, out, out_t0, pe):
        """out[:, out_t0 + t] = sum_j codes_embs[j](codes[:, t, j]) (+ pe[t]); all Q codebooks for
        t < prefix, codebooks < nar_stage after (valle_nar.py:180-185).  One gather kernel each."""
        _, t, q = codes.shape
        p = self.prefix_len_of(t)
        if p:
            kernels.embed_sum_pe(codes[:, :p], self._tables(q), pe, 0, out, out_t0=out_t0)
        if t > p:
            kernels.embed_sum_pe(codes[:, p:], self._tables(max(1, min(nar_stage, q))), pe, p, out,
                                 out_t0=out_t0 + p)
        return p

    @_on_device
    def _prepare_audio_codes(self, codes: torch.Tensor, nar_stage: int):
        """valle_nar.py:167-188 → ((B, T, d) embedding sum without position, prefix_len)."""
        dev = self._dev()
        codes = kernels.ids_to_device(codes, dev, self.config.num_audio_tokens, 'codes')
        b, t, _ = codes.shape
        y = torch.empty(b, t, self.config.d_model, device=dev, dtype=torch.float32)
        p = self._embed_audio(codes, nar_stage, y, 0, None)
        return y, p

    @_on_device
    def stage_logits(self, batch, stage: int, perf_mode: bool = False):
        """Intended forward of valle_nar.py:71-100 for `stage` in 1..Q-1: logits (B, T-prefix, V_a)
        of codebook `stage` for the non-prefix frames.  Full attention; key padding is dropped
        exactly as the reference's Transformer does when attn_mask is None (defect D6).
        perf_mode=True (opt-in, SECONDARY — SURVEY section 7): the stack's products run on the bf16 matrix cores
        (engine.transformer_forward_bf16); logits agree with the reference to 5e-2, not to the parity path's 2e-4."""
        dev = self._dev()
        cfg = self.config
        tokens = kernels.ids_to_device(batch['tokens'], dev, cfg.vocab_size, 'tokens')
        codes = kernels.ids_to_device(batch['codes'], dev, cfg.num_audio_tokens, 'codes (EOS/BOS have no codebook row)')
        tx = int(batch['tokens_lens'].max())
        b, t, _ = codes.shape
        d = cfg.d_model
        x = torch.empty(b, tx + t, d, device=dev, dtype=torch.float32)
        kernels.embed_sum_pe(tokens[:, :tx], [self.tokens_emb.weight.detach()],
                             self.tokens_position_emb.pe, 0, x)
        p = self._embed_audio(codes, stage, x, tx, self.audio_position_emb.pe)
        if perf_mode:
            cache = KVCache(cfg.num_layers, b, cfg.n_heads, tx + t, dev, dtype=kernels.H16)
            transformer_forward_bf16(self.transformer, x, cache, mode=kernels.MASK_FULL,
                                     embedding=self.stage_embs[stage - 1].weight)
        else:
            cache = KVCache(cfg.num_layers, b, cfg.n_heads, tx + t, dev)
            transformer_forward(self.transformer, x, cache, mode=kernels.MASK_FULL,
                                embedding=self.stage_embs[stage - 1].weight)  # (the parameter itself: adaln_table keys on it)
        z = x[:, tx + p:].reshape(b * (t - p), d)
        logits = self._head(z, stage, perf_mode)
        return logits.reshape(b, t - p, -1), p

    def _head(self, z, stage: int, perf_mode):
        """The stage's projection (valle_nar.py:97-98) over the packed target frames z (rows, d); in perf mode on the 16-bit
        matrix cores like the stack's products (fp32 accumulate, fp32 logits) when the tile GEMM takes the shape."""
        w = self.proj_layers[stage - 1].weight
        w16 = head16(self, w) if perf_mode else None
        if w16 is None:
            return kernels.linear(z, w.detach())
        return kernels.linear_bf16(kernels.to_bf16(z), w16)

    def _stage_logits_with_graph(self, batch, stage: int):
        """`stage_logits` composed from autograd Functions (training path)."""
        from . import autograd as A
        dev = self._dev()
        cfg = self.config
        tokens = kernels.ids_to_device(batch['tokens'], dev, cfg.vocab_size, 'tokens')
        codes = kernels.ids_to_device(batch['codes'], dev, cfg.num_audio_tokens, 'codes (EOS/BOS have no codebook row)')
        tx = int(batch['tokens_lens'].max())
        b, t, q = codes.shape
        d = cfg.d_model
        p = self.prefix_len_of(t)
        tabs = [e.weight for e in self.codes_embs]
        n_stage = max(1, min(stage, q))
        # text | prefix frames (all codebooks) | target frames (codebooks < stage) written into ONE buffer; a codebook
        # table that two parts read receives one gradient (no torch.cat, no strided copies, no gradient adds); the PE
        # dropouts (p = 0.1 in train mode, D9) are fields applied by the gather kernel before it stores a row
        seed = dropout.seed_if(dropout.live(self.tokens_position_emb.dropout), dropout.live(self.audio_position_emb.dropout))
        dr_t = dropout.spec(seed, dropout.site(dropout.PE_TEXT), dropout.live(self.tokens_position_emb.dropout))
        dr_a = dropout.spec(seed, dropout.site(dropout.PE_AUDIO), dropout.live(self.audio_position_emb.dropout))
        dropout.record('tokens_position_emb.dropout', dr_t, b * (tx + t), d)
        dropout.record('audio_position_emb.dropout', dr_a, b * (tx + t), d)
        spec = [(tokens[:, :tx], self.tokens_position_emb.pe, 0, [0], dr_t)]
        if p:
            spec.append((codes[:, :p], self.audio_position_emb.pe, 0, list(range(1, 1 + q)), dr_a))
        if t > p:
            spec.append((codes[:, p:], self.audio_position_emb.pe, p, list(range(1, 1 + n_stage)), dr_a))
        x = A.EmbedConcatFn.apply(spec, self.tokens_emb.weight, *tabs).reshape(b * (tx + t), d)
        x = A.transformer_train(self.transformer, x, b, tx + t, dict(mode=kernels.MASK_FULL),
                                embedding=self.stage_embs[stage - 1].weight)
        z = x.view(b, tx + t, d)[:, tx + p:].reshape(b * (t - p), d)
        return A.linear(z, self.proj_layers[stage - 1].weight).reshape(b, t - p, -1), p

    def training_step(self, batch, **kwargs):
        """Intended loss: CE of the stage's logits against the raw ids codes[:, prefix:, stage]
        (the reference slices the embedded tensor, valle_nar.py:81, and raises); mean over all
        positions.  `stage=` pins the stage (the reference draws it with random.randint, :76)."""
        from . import autograd as A
        stage = kwargs.get('stage') or random.randint(1, self.config.num_quantizers - 1)
        if torch.is_grad_enabled():
            logits, p = self._stage_logits_with_graph(batch, stage)
        else:
            logits, p = self.stage_logits(batch, stage)
        target = kernels.ids_to_device(batch['codes'][:, p:, stage], logits.device, self.config.num_audio_tokens, 'codes')
        rows = logits.shape[0] * logits.shape[1]
        return A.CrossEntropyFn.apply(logits.reshape(rows, -1), target.reshape(rows))

    def configure_optimizers(self):
        """The reference's ValleNAR has no configure_optimizers; same recipe as AR."""
        from torch import optim
        from .optim import FlatAdamW
        optimizer = FlatAdamW(self.parameters(), lr=self.config.lr, betas=self.config.betas,
                              weight_decay=self.config.weight_decay)
        scheduler = optim.lr_scheduler.CosineAnnealingWarmRestarts(optimizer, self.config.lr_warmup)
        return {'optimizer': optimizer, 'lr_scheduler': scheduler}

    @_on_device
    @torch.inference_mode()
    def generate(self, prompt_tokens, prompt_codes, target_tokens, target_codes_first_layer,
                 greedy: bool = False):
        """valle_nar.py:107-165 (intended algorithm) → codes (Ty, Q) int64.  The reference samples
        from Categorical(logits / temperature) (:160); greedy=True takes the arg-max instead,
        which is what the parity tests pin.  One utterance of `generate_batch`."""
        text = torch.cat([prompt_tokens, target_tokens], dim=0)
        return self.generate_batch([text], [prompt_codes], [target_codes_first_layer], greedy=greedy)[0]

    @_on_device
    @torch.inference_mode()
    def generate_batch(self, texts, prompt_codes, first_layers, greedy: bool = False, seed=None, perf_mode: bool = False):
        """Batched NAR decoding of B independent utterances (extension; `generate` is built on it).
        texts[b]: 1-D int64 text ids (prompt + target text); prompt_codes[b]: (Tc_b, Q) int64 acoustic
        prompt; first_layers[b]: (Ty_b,) int64 first-codebook codes of the target (the AR model's
        output).  Rows may differ in every length.  Returns a list of (Ty_b, Q) int64 device tensors.

        Row b is laid out [text_b | prompt_b | target_b | padding]; attention is full over the row's
        own length (per-row key length in the kernel, nothing materialised).  Per stage n = 1..Q-1
        (valle_nar.py:142-160): the target frames carry sum_{j<n} codes_embs[j](out[j]) + PE, one stack
        forward with AdaLN on stage_embs[n-1], head proj_layers[n-1] on the target frames of all rows at
        once, and codebook n is drawn on the device by `vh_categorical_rows` (Categorical(logits /
        temperature), or the arg-max when greedy).
        perf_mode=True: the seven stack forwards on the bf16 matrix cores (see `stage_logits`); the drawn codes are then not
        guaranteed to be the parity path's."""
        dev = self._dev()
        cfg = self.config
        q, d = cfg.num_quantizers, cfg.d_model
        B = len(texts)
        if B == 0 or len(prompt_codes) != B or len(first_layers) != B:
            raise ValueError('generate_batch: texts, prompt_codes and first_layers must be non-empty lists of equal length')
        texts = [kernels.ids_to_device(t, dev, cfg.vocab_size, 'text ids') for t in texts]
        pcs = [kernels.ids_to_device(c, dev, cfg.num_audio_tokens, 'prompt codes') for c in prompt_codes]
        firsts = [kernels.ids_to_device(f, dev, cfg.num_audio_tokens, 'first-layer codes') for f in first_layers]
        for t, c, f in zip(texts, pcs, firsts):
            if t.dim() != 1 or c.dim() != 2 or c.shape[1] != q or f.dim() != 1:
                raise ValueError('generate_batch: expected text (Tx,), prompt codes (Tc, Q), first layer (Ty,)')
        txs, tcs, tys = [t.shape[0] for t in texts], [c.shape[0] for c in pcs], [f.shape[0] for f in firsts]
        lens = [a + b + c for a, b, c in zip(txs, tcs, tys)]
        total = max(lens)
        pe_a, pe_t = self.audio_position_emb.pe, self.tokens_position_emb.pe
        if max(tc + ty for tc, ty in zip(tcs, tys)) > pe_a.shape[0] or max(txs) > pe_t.shape[0]:
            raise _lib.VhError('sequence exceeds the positional table (max_len 5000)')
        ty_max = max(tys)
        pad = torch.nn.utils.rnn.pad_sequence                      # index plumbing: ragged id lists -> padded id tensors
        i32 = lambda v: _lib.to_device_async(torch.tensor(v, dtype=torch.int32), dev)   # noqa: E731
        out = torch.zeros(B, ty_max, q, device=dev, dtype=torch.int64)
        out[:, :, 0] = pad(firsts, batch_first=True)
        tx_d, tc_d, ty_d = i32(txs), i32(tcs), i32(tys)
        t0_target = i32([a + c for a, c in zip(txs, tcs)])
        valid = torch.arange(ty_max, device=dev)[None, :] < ty_d[:, None]          # (B, ty_max) target frames that exist
        # text and acoustic-prompt embeddings do not change from stage to stage: build them once, ONE launch each for
        # the whole ragged batch (per-row lengths and offsets travel to the kernel)
        base = torch.zeros(B, total, d, device=dev, dtype=torch.float32)
        kernels.embed_sum_pe(pad(texts, batch_first=True), [self.tokens_emb.weight.detach()], pe_t, 0, base, lens=tx_d,
                             row_t0=torch.zeros_like(tx_d), max_pos=max(txs))
        if max(tcs):
            kernels.embed_sum_pe(pad(pcs, batch_first=True), self._tables(q), pe_a, 0, base, lens=tc_d, row_t0=tx_d,
                                 max_pos=max(tcs))
        kv_len = i32(lens) if len(set(lens)) > 1 else None
        # flat row indices of every target frame, and where each row's run starts in the packed logits
        idx = torch.cat([torch.arange(tys[b], device=dev) + (b * total + txs[b] + tcs[b]) for b in range(B)])
        starts = [0]
        for ty in tys:
            starts.append(starts[-1] + ty)
        cache = KVCache(cfg.num_layers, B, cfg.n_heads, total, dev, dtype=kernels.H16 if perf_mode else torch.float32)
        scratch = (ForwardScratch16 if perf_mode else ForwardScratch)(B * total, d, cfg.dim_feedforward, dev)
        forward = transformer_forward_bf16 if perf_mode else transformer_forward
        x = torch.empty_like(base)
        if seed is None:       # drawn from torch's generator, so torch.manual_seed() makes a run repeatable
            seed = 0 if greedy else int(torch.randint(0, 2 ** 62, (1,)).item())
        toks = torch.empty(starts[-1], device=dev, dtype=torch.int64)
        for n in range(1, q):
            x.copy_(base)
            kernels.embed_sum_pe(out, self._tables(n), pe_a, 0, x, lens=ty_d, row_pos0=tc_d, row_t0=t0_target,
                                 max_pos=max(tc + ty for tc, ty in zip(tcs, tys)))
            forward(self.transformer, x, cache, mode=kernels.MASK_FULL, kv_len=kv_len,
                    embedding=self.stage_embs[n - 1].weight, scratch=scratch)                # (cached AdaLN table per stage)
            z = x.view(B * total, d).index_select(0, idx)                      # target frames of every row
            logits = self._head(z, n, perf_mode)
            kernels.categorical_rows(logits, toks, temperature=cfg.temperature, greedy=greedy, seed=seed,
                                     stream_id=n)
            out[:, :, n][valid] = toks                                          # packed row-major -> (B, ty_max)
        _lib.raise_device_errors(dev)
        return [out[b, :tys[b]].clone() for b in range(B)]
