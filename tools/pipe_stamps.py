"""Where a pipelined decode step spends its time (vh_attn_decode_pipe diagnostic stamps): for the LAST decode step,
per layer, when workgroup 0 of the attention launch started, when its q / k / v pairs arrived and when it ended —
relative to layer 0's start (developer tool).  usage: pipe_stamps.py [rows] [graph 0|1] [pipe_mode]"""
import os
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, _lib, engine, get_model_class, synth  # noqa: E402


def main(rows=32, graph=1, mode=0):
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm',
                      top_k=1, num_beams=rows, max_audio_len=512)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    utts = [synth.synth_utterance(cfg, 128, 128, 767, seed=1234 + u) for u in range(rows)]
    texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
    firsts = [u[1][:, 0].cuda() for u in utts]
    engine.PIPELINED_ATTENTION = True
    _lib.lib().vh_set_tuning(7, mode)
    keep = {}
    orig = engine.ArDecoder.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.pipe_err[1] = 1
        keep['dec'] = self
    engine.ArDecoder.__init__ = init
    close = engine.ArDecoder.close
    engine.ArDecoder.close = lambda self: None
    m.generate_batch(texts, firsts, use_graph=bool(graph))
    torch.cuda.synchronize()
    dec = keep['dec']
    print('decode step', m.last_generate_stats['decode_ms'] / 511 * 1e3, 'us; error word', hex(int(dec.pipe_err[0].item()) & 0xffffffff))
    st = dec.pipe_err[16:16 + 8 * 12].view(torch.int64).view(12, 4).cpu().double() / 100.0   # us (100 MHz)
    t0 = st[0, 0]
    prev_end = None
    for l in range(12):
        s, q, e = (float(st[l, k] - t0) for k in range(3))
        gap = '' if prev_end is None else f'  start - prev end {s - prev_end:6.2f}  q - prev end {q - prev_end:6.2f}'
        print(f'layer {l:2d}: start {s:7.2f}  inputs {q:7.2f}  end {e:7.2f}   wait {q - s:6.2f}  after q {e - q:6.2f}{gap}')
        prev_end = e
    close(dec)


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:]])
