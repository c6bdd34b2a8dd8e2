"""Timeline of one decode step from a rocprofv3 --kernel-trace CSV: per kernel start offset, duration, and the
gap to the previous kernel's end (developer tool).  usage: step_timeline.py <kernel_trace.csv> [step_index]"""
import csv
import sys


def main(path, step=300):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a decode step ends with greedy_step_kernel; find the step-th occurrence
    ends = [i for i, r in enumerate(rows) if 'greedy_step' in r['Kernel_Name']]
    lo, hi = ends[step] + 1, ends[step + 1] + 1
    t0 = int(rows[lo]['Start_Timestamp'])
    prev_end = None
    tot_k = tot_gap = 0.0
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
        print(f'{(s - t0) / 1e3:8.2f} us  dur {(e - s) / 1e3:6.2f}  gap {gap:6.2f}  {name}')
        tot_k += (e - s) / 1e3
        tot_gap += gap
        prev_end = e
    print(f'step: {(prev_end - t0) / 1e3:.1f} us; kernels {tot_k:.1f} us; gaps {tot_gap:.1f} us ({hi - lo} launches)')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300)
