"""Stage / barrier timing inside the persistent decode-chain kernel (vh_decode_chain diagnostic stamps).
Runs one eager decode of configs[1]'s shape and prints, for the LAST chain launch, per phase: the median over
workgroups and the slowest workgroup (developer tool)."""
import os
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, engine, get_model_class, synth  # noqa: E402


def main(rows=32):
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm',
                      top_k=1, num_beams=rows, max_audio_len=6)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    utts = [synth.synth_utterance(cfg, 128, 128, 767, seed=1234 + u) for u in range(rows)]
    texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]
    firsts = [u[1][:, 0].cuda() for u in utts]
    keep = {}
    orig = engine.ArDecoder.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        if self.chain:
            self.chain_sync[2] = 1
        keep['dec'] = self
    engine.ArDecoder.__init__ = init
    close = engine.ArDecoder.close
    engine.ArDecoder.close = lambda self: None          # keep the sync block readable
    try:
        m.generate_batch(texts, firsts, use_graph=False)
    except Exception as e:
        print('generate raised:', str(e)[:80])
    torch.cuda.synchronize()
    dec = keep['dec']
    w = int(dec.chain_sync[1].item()) & 0xffffffff
    if w:
        t = (w >> 8) & 0xfffff
        print(f'error word {w:#x}: reduce={bool(w & 0x40000000)} wg={w & 0xff} stage_tag={t & 7} layer={(t >> 3) & 63} epoch~={(t >> 9)}')
    st = dec.chain_sync[64:].view(torch.int64).view(256, 16)[:, :6].cpu().double() / 100.0     # us (100 MHz)
    names = ['S1 out-proj', 'S2 lin1 (+wait)', 'S3 lin2 slices (+wait)', 'S4 reduce (+wait)', 'S5 qkv/head (+wait)']
    t0 = st[:, 0].min()
    print(f'workgroup start spread: {float(st[:, 0].max() - t0):.2f} us')
    for i, n in enumerate(names):
        d = st[:, i + 1] - st[:, i]
        print(f'{n:16s} median {float(d.median()):6.2f} us   max {float(d.max()):6.2f} us   '
              f'(exit at {float(st[:, i + 1].max() - t0):6.2f} us)')
    close(dec)


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:]])
