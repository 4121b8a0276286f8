"""Timeline of the big kernels of the LAST process() call in a rocprofv3 kernel trace: start, duration, gap to the previous
one, and what else ran in the gaps."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = []
with open(path) as fp:
    for r in csv.DictReader(fp):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]))
rows.sort()
big = [r for r in rows if 'k_georef_rows' in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
last = big[-n:]
t0 = last[0][0]
prev_end = None
for s, e, name in last:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('start %8.1f us  dur %7.1f us  gap before %6.1f us' % ((s - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end = e
print('span %.1f us, kernels %.1f us' % ((last[-1][1] - t0) / 1e3, sum(e - s for s, e, _ in last) / 1e3))
after = [r for r in rows if r[0] >= last[-1][1]]
print('after the last big kernel:', [(round((s - last[-1][1]) / 1e3, 1), round((e - s) / 1e3, 1), nm[:16]) for s, e, nm in after[:8]])
