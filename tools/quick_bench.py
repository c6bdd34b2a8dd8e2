"""Developer timing script (not the graded bench): config-2 shaped AR generate, phase timings."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import ConfigValle, get_model_class, synth  # noqa: E402


def main(beams=32, text=256, frames=767, new=512, layers=12, d=512, graph=True):
    import os
    import tempfile
    os.chdir(tempfile.mkdtemp())
    cfg = ConfigValle(d_model=d, n_heads=d // 64, dim_feedforward=4 * d, num_layers=layers, dropout=0.0,
                      norm='LayerNorm', num_beams=beams, top_k=1, max_audio_len=new)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.to('cuda').eval()
    pt, pc, tt = synth.synth_utterance(cfg, text // 2, text - text // 2, frames, seed=1234)
    pt, pc, tt = pt.cuda(), pc.cuda(), tt.cuda()
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        texts = [torch.cat([pt, tt])] * beams
        rows = m.generate_batch(texts, [pc[:, 0]] * beams, use_graph=graph)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f'iter {it}: {dt*1e3:.1f} ms  {beams*new/dt:.0f} tok/s  stats={m.last_generate_stats} '
              f'eos_in_out={(rows[:, frames+1:] == cfg.num_audio_tokens).sum().item()}', flush=True)


if __name__ == '__main__':
    kw = {}
    for a in sys.argv[1:]:
        k, v = a.split('=')
        kw[k] = int(v)
    main(**kw)
