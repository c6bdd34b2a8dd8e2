"""Per-generate decode-step time of the pipelined form right after switching to it (developer tool)."""
import os
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, engine, get_model_class, synth  # noqa: E402

cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm',
                  top_k=1, num_beams=32, max_audio_len=512)
sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
m = get_model_class('ValleAR')(cfg)
m.load_state_dict(sd)
m = m.cuda().eval()
utts = [synth.synth_utterance(cfg, 128, 128, 767, seed=1234 + u) for u in range(32)]
texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
firsts = [u[1][:, 0].cuda() for u in utts]
for pipe, n in ((False, int(sys.argv[1]) if len(sys.argv) > 1 else 3), (True, 8), (False, 2), (True, 3)):
    engine.PIPELINED_ATTENTION = pipe
    for i in range(n):
        m.generate_batch(texts, firsts)
        torch.cuda.synchronize()
        print(f'pipe={pipe} generate {i}: decode step {m.last_generate_stats["decode_ms"] / 511 * 1e3:8.1f} us', flush=True)
