"""Soak of the perf-mode kernels reworked in round 6 (developer tool): the LDS-DMA ring of attn16_kernel is only correct if its counted
waits and its one barrier per tile order every request against every read — a miss shows up as a run that differs from the first.
  (a) vh_attn_rows_bf16 alone, N launches on each of several shapes (full / prefix masks, ragged key lengths, T not a multiple of the
      tile, long context), every output bit-identical to the first launch's;
  (b) the configs[1] prompt pass + 32 decode steps in perf mode, N generates, identical tokens;
  (c) one configs[2] NAR stage in perf mode, N forwards, bit-identical logits.
usage: soak_perf_mode.py [n=40]"""
import os
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, get_model_class, synth  # noqa: E402
from valle2_amd import kernels as K  # noqa: E402


def main(n=40):
    g = torch.Generator().manual_seed(0)
    H16 = K.H16
    for B, h, T, mode in ((64, 8, 1024, 'full'), (32, 8, 1024, 'prefix'), (5, 3, 777, 'prefix'), (7, 2, 333, 'full'), (2, 16, 2875, 'full')):
        d = 64 * h
        q = (torch.randn(B * T, d, generator=g) * K.Q16_PRESCALE).to(H16).cuda()
        k = torch.randn(B, h, T, 64, generator=g).to(H16).cuda()
        v = torch.randn(B, h, T, 64, generator=g).to(H16).cuda()
        kvl = torch.tensor([T - (37 * i) % (T // 2) for i in range(B)], dtype=torch.int32).cuda()
        kw = dict(mode=K.MASK_PREFIX, x_len=T // 4, kv_len=kvl) if mode == 'prefix' else dict(mode=K.MASK_FULL, kv_len=kvl)
        ref, bad = None, 0
        for i in range(5 * n):
            out = torch.empty(B * T, d, device='cuda', dtype=H16)
            K.attn_rows_bf16(q, k, v, out, B, h, T, T, **kw)
            if ref is None:
                ref = out.clone()
            elif not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
                bad += 1
        print(f'attn_rows_bf16 B={B} h={h} T={T} {mode}: {5 * n} launches, {bad} differing from the first', flush=True)
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm', top_k=1, num_beams=32,
                      max_audio_len=32)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    utts = [synth.synth_utterance(cfg, 128, 128, 767, seed=1234 + u) for u in range(32)]
    texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
    firsts = [u[1][:, 0].cuda() for u in utts]
    ref, bad = None, 0
    for i in range(n):
        out = m.generate_batch(texts, firsts, perf_mode=True)
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out, ref):
            bad += 1
    print(f'perf-mode generate_batch (prompt pass 32 x 1024 + 31 steps): {n} runs, {bad} differing from the first', flush=True)
    del m
    ncfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='AdaptiveLayerNorm')
    nar = get_model_class('ValleNAR')(ncfg)
    nar.load_state_dict(synth.make_state_dict(ncfg, 'ValleNAR', seed=0, rich=False))
    nar = nar.cuda().eval()
    batch = synth.synth_nar_batch(ncfg, 64, n_tokens=256, n_frames=768, seed=5)
    ref, bad = None, 0
    for i in range(n):
        logits, _ = nar.stage_logits(batch, 1 + i % 7, perf_mode=True)
        if i < 7:
            ref = ref or {}
            ref[i] = logits.clone()
        elif not torch.equal(logits, ref[i % 7]):
            bad += 1
    print(f'perf-mode NAR stage_logits (64 x 1024, stages 1..7 in turn): {n} forwards, {bad} differing from the same stage\'s first', flush=True)


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:]])
