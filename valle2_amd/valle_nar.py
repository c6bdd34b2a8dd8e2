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

from . import _lib, kernels
from .engine import ForwardScratch, KVCache, transformer_forward
from .modules import PositionalEncoding, TokenEmbedding, Transformer
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
        if dev.type != 'cuda':
            raise _lib.VhError('ValleNAR is on the CPU; move it to a HIP device (no CPU fallback)')
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

    def _prepare_audio_codes(self, codes: torch.Tensor, nar_stage: int):
        """valle_nar.py:167-188 → ((B, T, d) embedding sum without position, prefix_len)."""
        dev = self._dev()
        codes = codes.to(dev)
        b, t, _ = codes.shape
        y = torch.empty(b, t, self.config.d_model, device=dev, dtype=torch.float32)
        p = self._embed_audio(codes, nar_stage, y, 0, None)
        return y, p

    def stage_logits(self, batch, stage: int):
        """Intended forward of valle_nar.py:71-100 for `stage` in 1..Q-1: logits (B, T-prefix, V_a)
        of codebook `stage` for the non-prefix frames.  Full attention; key padding is dropped
        exactly as the reference's Transformer does when attn_mask is None (defect D6)."""
        dev = self._dev()
        cfg = self.config
        tokens, codes = batch['tokens'].to(dev), batch['codes'].to(dev)
        tx = int(batch['tokens_lens'].max())
        b, t, _ = codes.shape
        d = cfg.d_model
        x = torch.empty(b, tx + t, d, device=dev, dtype=torch.float32)
        kernels.embed_sum_pe(tokens[:, :tx], [self.tokens_emb.weight.detach()],
                             self.tokens_position_emb.pe, 0, x)
        p = self._embed_audio(codes, stage, x, tx, self.audio_position_emb.pe)
        cache = KVCache(cfg.num_layers, b, cfg.n_heads, tx + t, dev)
        transformer_forward(self.transformer, x, cache, mode=kernels.MASK_FULL,
                            embedding=self.stage_embs[stage - 1].weight.detach())
        z = x[:, tx + p:].reshape(b * (t - p), d)
        logits = kernels.linear(z, self.proj_layers[stage - 1].weight.detach())
        return logits.reshape(b, t - p, -1), p

    def _stage_logits_with_graph(self, batch, stage: int):
        """`stage_logits` composed from autograd Functions (training path)."""
        from . import autograd as A
        dev = self._dev()
        cfg = self.config
        tokens, codes = batch['tokens'].to(dev), batch['codes'].to(dev)
        tx = int(batch['tokens_lens'].max())
        b, t, q = codes.shape
        d = cfg.d_model
        p = self.prefix_len_of(t)
        parts = [A.EmbedSumPeFn.apply(tokens[:, :tx], self.tokens_position_emb.pe, 0, self.tokens_emb.weight)]
        tabs = [e.weight for e in self.codes_embs]
        if p:
            parts.append(A.EmbedSumPeFn.apply(codes[:, :p], self.audio_position_emb.pe, 0, *tabs))
        if t > p:
            parts.append(A.EmbedSumPeFn.apply(codes[:, p:], self.audio_position_emb.pe, p,
                                              *tabs[: max(1, min(stage, q))]))
        drops = [self.tokens_position_emb.dropout] + [self.audio_position_emb.dropout] * (len(parts) - 1)
        parts = [dr(x) if (dr.training and dr.p > 0) else x for dr, x in zip(drops, parts)]
        x = torch.cat(parts, dim=1).reshape(b * (tx + t), d)
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
        target = batch['codes'][:, p:, stage].to(logits.device)
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

    @torch.inference_mode()
    def generate(self, prompt_tokens, prompt_codes, target_tokens, target_codes_first_layer,
                 greedy: bool = False):
        """valle_nar.py:107-165 (intended algorithm) → codes (Ty, Q) int64.  The reference samples
        from Categorical(logits / temperature) (:160); greedy=True takes the arg-max instead,
        which is what the parity tests pin."""
        dev = self._dev()
        cfg = self.config
        q = cfg.num_quantizers
        d = cfg.d_model
        text = torch.cat([prompt_tokens, target_tokens], dim=0).to(dev).unsqueeze(0)
        pc = prompt_codes.to(dev).unsqueeze(0)                      # (1, Tc, Q)
        tx, tc, ty = text.shape[1], pc.shape[1], target_codes_first_layer.shape[0]
        out = torch.zeros(1, ty, q, device=dev, dtype=torch.int64)
        out[0, :, 0] = target_codes_first_layer.to(dev)
        total = tx + tc + ty
        cache = KVCache(cfg.num_layers, 1, cfg.n_heads, total, dev)
        scratch = ForwardScratch(total, d, cfg.dim_feedforward, dev)
        x = torch.empty(1, total, d, device=dev, dtype=torch.float32)
        pe_a, pe_t = self.audio_position_emb.pe, self.tokens_position_emb.pe
        for n in range(1, q):
            kernels.embed_sum_pe(text, [self.tokens_emb.weight.detach()], pe_t, 0, x)
            kernels.embed_sum_pe(pc, self._tables(q), pe_a, 0, x, out_t0=tx)
            kernels.embed_sum_pe(out, self._tables(n), pe_a, tc, x, out_t0=tx + tc)
            transformer_forward(self.transformer, x, cache, mode=kernels.MASK_FULL,
                                embedding=self.stage_embs[n - 1].weight.detach(), scratch=scratch)
            logits = kernels.linear(x[0, tx + tc:], self.proj_layers[n - 1].weight.detach())
            if greedy:
                out[0, :, n] = torch.argmax(logits, dim=-1)
            else:
                probs = torch.softmax(logits / cfg.temperature, dim=-1)
                out[0, :, n] = torch.multinomial(probs, 1).squeeze(1)
        return out[0]
