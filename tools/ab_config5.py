"""Developer tool: vh_ffn_decode against the three-launch FeedForward at configs[4] shapes (24L/1024d, 8 rows)."""
import sys, time, torch, os, tempfile
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, get_model_class, synth, _lib
cfg = ConfigValle(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0, norm='LayerNorm', num_beams=8, top_k=1, max_audio_len=256)
sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
m = get_model_class('ValleAR')(cfg); m.load_state_dict(sd); m = m.to('cuda').eval()
for frames in (225, 2250):
    utts = [synth.synth_utterance(cfg, 200, 200, frames, seed=7 + u) for u in range(8)]
    texts = [torch.cat([u[0], u[2]]).cuda() for u in utts]; firsts = [u[1][:, 0].cuda() for u in utts]
    outs = {}
    for rnd in range(3):
        for knob in (2, 1):
            _lib.lib().vh_set_tuning(5, knob)
            out = m.generate_batch(texts, firsts); torch.cuda.synchronize()
            outs.setdefault(knob, []).append(m.last_generate_stats['decode_ms'] / 255 * 1e3)
            last = out if knob == 2 else last
            if knob == 1: same = bool(torch.equal(out, last))
    print(f'frames {frames}: fused {min(outs[2]):.1f} us/step, three launches {min(outs[1]):.1f} us/step, same tokens {same}', flush=True)
_lib.lib().vh_set_tuning(5, 0)
