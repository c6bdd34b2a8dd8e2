"""Analyse a rocprofv3 kernel trace: how much do kernels overlap in time? (diagnostic)"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:], r['Queue_Id']) for r in rows]
ev.sort()
# take the last 4000 kernels (steady-state decode)
ev = ev[-6000:-500]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = sum(e[1] - e[0] for e in ev)
print(f'span {(t1-t0)/1e3:.0f} us, sum of kernel durations {busy/1e3:.0f} us, ratio {busy/(t1-t0):.2f}, queues {sorted(set(e[3] for e in ev))}')
# union coverage
cov, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e, *_ in ev[1:]:
    if s > cur_e:
        cov += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print(f'time with >=1 kernel running: {cov/1e3:.0f} us ({cov/(t1-t0):.2f} of span)')
for e in ev[1000:1040]:
    print(f'{(e[0]-t0)/1e3:10.1f} {(e[1]-e[0])/1e3:7.1f} q={e[3]} {e[2]}')
