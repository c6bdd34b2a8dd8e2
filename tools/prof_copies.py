"""Developer tool: which call sites of one NAR training step issue copies (torch profiler with stacks).
python tools/prof_copies.py"""
import sys, os, tempfile
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
os.chdir(tempfile.mkdtemp())
from valle2_amd import ConfigValle, get_model_class, synth
from torch.profiler import profile, ProfilerActivity
cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='AdaptiveLayerNorm', batch_size=16)
torch.manual_seed(0)
model = get_model_class('ValleNAR')(cfg).cuda().train()
opt = model.configure_optimizers()['optimizer']
batches = []
for i in range(4):
    b = synth.synth_nar_batch(cfg, 16, n_tokens=80, n_frames=560, seed=100 + i)
    batches.append({k: (v if k.endswith('_lens') else v.cuda()) for k, v in b.items()})
def step(b):
    loss = model.training_step(b, stage=3); loss.backward(); opt.step(max_norm=1.0, zero_grad=True)
for b in batches[:3]: step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(batches[3]); torch.cuda.synchronize()
ev = [e for e in prof.events() if 'copy' in e.name.lower() or 'Memcpy' in e.name]
from collections import Counter
c = Counter()
for e in ev:
    st = [s for s in (e.stack or []) if 'valle2_amd' in s]
    c[(e.name, st[0] if st else '-')] += 1
for k, v in c.most_common(25): print(v, k)
