"""Developer tool: generate() at the reference's generation defaults (num_beams=4, top_k=50, max_audio_len=1024) on the
12L/512d model, once warm + `reps` timed — the command profiles/r4_default_generate_kernel_stats.md is a rocprofv3
--kernel-trace --stats summary of.   usage: python tools/default_generate.py [reps=2]"""
import json
import os
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.chdir(tempfile.mkdtemp())
import torch  # noqa: E402

import bench  # noqa: E402
from valle2_amd import ConfigValle, synth  # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda', 0)
    cfg = ConfigValle(**bench.AR)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    utt = synth.synth_utterance(cfg, bench.TEXT // 2, bench.TEXT // 2, bench.FRAMES, seed=1234)
    print(json.dumps(bench.default_generate_leg(dev, sd, utt), indent=1))
