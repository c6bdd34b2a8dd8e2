"""Condense rocprofv3 CSV output (kernel trace / stats / PMC counter collection) into the small
summaries that are committed under profiles/.  Runs on the GPU box right after the profiler.

usage: summarize_prof.py <rocprof_out_dir> <summary_out.md> [title]
"""
import csv
import glob
import statistics
import sys
from collections import defaultdict


def short(name):
    return name.split('(')[0].replace('void ', '')[:60]


def main(src, dst, title=''):
    out = [f'# {title or src}', '']
    stats = glob.glob(f'{src}/**/*kernel_stats.csv', recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        out += ['## rocprofv3 --kernel-trace --stats (per kernel, whole command)', '',
                '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
        for r in rows[:25]:
            out.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                       f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | "
                       f"{float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
        out.append('')
    traces = glob.glob(f'{src}/**/*kernel_trace.csv', recursive=True)
    if traces:
        agg = defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(traces[0])):
            key = (short(r['Kernel_Name']), r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
            agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
            meta[key] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'])
        out += ['## kernel trace grouped by (kernel, grid, workgroup)', '',
                '| kernel | grid x,y (threads) | wg | n | avg us | median us | total ms | vgpr/agpr/sgpr | lds B |',
                '|---|---|---|---|---|---|---|---|---|']
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:30]:
            m = meta[k]
            out.append(f'| {k[0]} | {k[1]},{k[2]} | {k[3]} | {len(v)} | {sum(v) / len(v):.2f} | '
                       f'{statistics.median(v):.2f} | {sum(v) / 1e3:.2f} | {m[0]}/{m[1]}/{m[2]} | {m[3]} |')
        out.append('')
    counters = glob.glob(f'{src}/**/*counter_collection.csv', recursive=True)
    if counters:
        agg = defaultdict(list)
        for path in counters:                                # one file per --pmc pass
            for r in csv.DictReader(open(path)):
                agg[(short(r['Kernel_Name']), r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''),
                     r['Counter_Name'])].append(float(r['Counter_Value']))
        out += ['## PMC counters per dispatch (mean over dispatches)', '',
                '| kernel | grid | counter | dispatches | mean | min | max |', '|---|---|---|---|---|---|---|']
        for k, v in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1], kv[0][2]))[:120]:
            out.append(f'| {k[0]} | {k[1]} | {k[2]} | {len(v)} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |')
        out.append('')
        # derived: MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the 1024 SIMDs) over the
        # dispatch's shader cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs) x 1024 SIMDs — MI355X_MICROARCH.md
        util = []
        for (kern, grid, name), v in agg.items():
            if name != 'SQ_VALU_MFMA_BUSY_CYCLES':
                continue
            act = agg.get((kern, grid, 'GRBM_GUI_ACTIVE'))
            if act and sum(act) > 0 and sum(v) > 0:
                cyc = sum(act) / len(act) / 8.0
                util.append((sum(v) / len(v) / (cyc * 1024.0), kern, grid, len(v), cyc))
        if util:
            out += ['## MFMA pipe utilisation per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))', '',
                    '| kernel | grid | dispatches | shader cycles per dispatch | MFMA busy |', '|---|---|---|---|---|']
            for u, kern, grid, n, cyc in sorted(util, reverse=True):
                out.append(f'| {kern} | {grid} | {n} | {cyc:.0f} | {100 * u:.1f} % |')
            out.append('')
    open(dst, 'w').write('\n'.join(out) + '\n')
    print(f'wrote {dst}')


if __name__ == '__main__':
    main(*sys.argv[1:4])
