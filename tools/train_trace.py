"""Developer tool: ONE AR training step at configs[3] under `rocprofv3 --kernel-trace` -> per-kernel counts, busy time and
the idle gaps between consecutive kernels on the stream.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- python3 tools/train_trace.py run
                 python3 tools/train_trace.py report OUT"""
import csv
import glob
import os
import sys
import tempfile
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def run(model_name='ValleAR'):
    import torch
    os.chdir(tempfile.mkdtemp())
    from valle2_amd import ConfigValle, get_model_class, synth
    norm = 'LayerNorm' if model_name == 'ValleAR' else 'AdaptiveLayerNorm'
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm=norm, batch_size=16)
    torch.manual_seed(0)
    model = get_model_class(model_name)(cfg).cuda().train()
    opt = model.configure_optimizers()['optimizer']
    batches = []                       # in HBM before the loop, as in bench.py: the host then runs ahead of the GPU and
    for i in range(5):                 # the traced (last) step shows the stream as the training loop sees it
        if model_name == 'ValleAR':
            batch = synth.synth_ar_batch(cfg, 16, seed=100 + i)
        else:
            batch = synth.synth_nar_batch(cfg, 16, n_tokens=80, n_frames=560, seed=100 + i)
        batches.append({k: (v if k.endswith('_lens') else v.cuda()) for k, v in batch.items()})
    torch.cuda.synchronize()
    for i, batch in enumerate(batches):
        loss = model.training_step(batch, **({'stage': 3} if model_name == 'ValleNAR' else {}))
        loss.backward()
        opt.step(max_norm=1.0, zero_grad=True)
    torch.cuda.synchronize()


def short(name):
    return name.split('(')[0].replace('void ', '')[:56]


def report(src):
    rows = list(csv.DictReader(open(glob.glob(f'{src}/**/*kernel_trace.csv', recursive=True)[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # the last step = everything after the last AdamW launch but one (steps end with the optimizer's update kernel)
    ends = [i for i, r in enumerate(rows) if 'adamw_flat_kernel' in r['Kernel_Name']]
    rows = rows[ends[-2] + 1:ends[-1] + 1]
    busy = defaultdict(lambda: [0, 0.0])
    gap_after = defaultdict(lambda: [0, 0.0])
    total_gap = 0.0
    for i, r in enumerate(rows):
        k = short(r['Kernel_Name'])
        busy[k][0] += 1
        busy[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if i:
            g = (int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3
            total_gap += max(g, 0.0)
            gap_after[short(rows[i - 1]['Kernel_Name'])][0] += 1
            gap_after[short(rows[i - 1]['Kernel_Name'])][1] += max(g, 0.0)
    span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
    print(f'{len(rows)} launches, span {span / 1e3:.2f} ms, busy {sum(v[1] for v in busy.values()) / 1e3:.2f} ms, '
          f'idle gaps {total_gap / 1e3:.2f} ms')
    print('| kernel | launches | busy us | mean us | idle after it, us | mean gap us |')
    print('|---|---|---|---|---|---|')
    for k, (n, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
        ga = gap_after.get(k, [0, 0.0])
        print(f'| {k} | {n} | {t:.0f} | {t / n:.1f} | {ga[1]:.0f} | {ga[1] / max(1, ga[0]):.1f} |')


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(*sys.argv[2:3])
    else:
        report(sys.argv[2])
