"""The perf-mode prompt pass of configs[1] as generate_batch() runs it, under `rocprofv3 --kernel-trace`: which kernels run between the
first embedding launch and the first decode step, how long each takes and how long the device idles between them.

    rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- python3 tools/trace_prefill.py run
    python3 tools/trace_prefill.py report OUT
"""
import csv
import glob
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def run():
    import torch
    from valle2_amd import ConfigValle, get_model_class, synth
    dev = 'cuda'
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm', num_beams=32, top_k=1,
                      max_audio_len=4)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    utts = [synth.synth_utterance(cfg, 128, 128, 767, seed=1234 + u) for u in range(32)]
    texts = [torch.cat([u[0], u[2]]).to(dev) for u in utts]
    firsts = [u[1][:, 0].to(dev) for u in utts]
    for _ in range(3):
        m.generate_batch(texts, firsts, perf_mode=True)
    torch.cuda.synchronize()
    print('prefill_ms', m.last_generate_stats['prefill_ms'])


def report(src):
    f = glob.glob(f'{src}/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    # the last generate: from the last embed_sum_pe launch before the last block of kernels
    names = [r['Kernel_Name'] for r in rows]
    starts = [i for i, n in enumerate(names) if 'embed_sum_pe' in n]
    # the last prompt pass begins at the embedding launch that is followed by a layernorm16
    i0 = max(i for i in starts if any('layernorm16' in n for n in names[i:i + 6]))
    i1 = next(i for i in range(i0, len(rows)) if 'attn_decode' in names[i])
    seg = rows[i0:i1]
    t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    agg = {}
    for r in seg:
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print(f'prompt pass of the last generate: {len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, '
          f'idle between kernels {(t1 - t0 - busy) / 1e3:.1f} us')
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'  {k:50s} x{n:3d}  {t / 1e3:9.1f} us  ({t / n / 1e3:7.1f} us each)')


if __name__ == '__main__':
    run() if sys.argv[1] == 'run' else report(sys.argv[2])
