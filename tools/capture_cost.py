"""Host time of the per-generate decoder set-up at configs[1]: ArDecoder construction (buffers, LayerNorm folding, layer table) and
graph capture (developer tool)."""
import os
import sys
import tempfile
import time
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
t = {}
init, run = engine.ArDecoder.__init__, engine.ArDecoder.run


def timed_init(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    init(self, *a, **k)
    torch.cuda.synchronize(); t.setdefault('init', []).append(time.perf_counter() - t0)


def timed_run(self, n):
    first = not self._captured
    if first:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    run(self, n)
    if first:
        t.setdefault('first_run_enqueue', []).append(time.perf_counter() - t0)


engine.ArDecoder.__init__, engine.ArDecoder.run = timed_init, timed_run
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.generate_batch(texts, firsts)
    torch.cuda.synchronize(); t.setdefault('generate', []).append(time.perf_counter() - t0)
for k, v in t.items():
    print(f'{k:20s}', ' '.join(f'{x * 1e3:8.2f}' for x in v), 'ms')
st = m.last_generate_stats
print('prefill_ms', st['prefill_ms'], 'decode_ms', st['decode_ms'], 'per step', st['decode_ms'] / 511 * 1e3)
