"""Summarise a rocprofv3 kernel trace csv: per-kernel totals and idle gaps between kernels (all streams merged)."""
import csv, glob, sys
from collections import defaultdict
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = []
with open(path) as fp:
    for r in csv.DictReader(fp):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]))
rows.sort()
# steady state: from the 10th to the last launch of the dominant kernel
big = [i for i, r in enumerate(rows) if 'k_georef_rows' in r[2]]
lo, hi = big[len(big) // 3], big[-1]
sel = rows[lo:hi]
t0, t1 = sel[0][0], rows[hi][0]
nfr = sum(1 for r in sel if 'k_georef_rows' in r[2])
tot = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    tot[n][0] += e - s
    tot[n][1] += 1
busy, cur_s, cur_e = 0, None, None
for s, e, n in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('frames', nfr, 'wall/frame us', (t1 - t0) / nfr / 1e3, 'gpu busy/frame us', busy / nfr / 1e3)
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print('%-62s n/frame %.2f  avg us %8.2f  us/frame %8.2f' % (n, c / nfr, d / c / 1e3, d / nfr / 1e3))
