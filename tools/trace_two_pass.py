"""All kernels of the last stretch of a rocprofv3 kernel trace as a timeline (start, duration, name): how the big kernel and
the binning kernel of the two-pass plan overlap.  usage: trace_two_pass.py <trace dir> [n kernels]"""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = []
with open(path) as fp:
    for r in csv.DictReader(fp):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:28], r.get('Queue_Id', '?')))
rows = [r for r in rows if not r[2].startswith('at::') and 'rocclr' not in r[2]]
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
last = rows[-n:]
t0 = last[0][0]
for s, e, name, q in last:
    print('%9.1f .. %9.1f  (%7.1f us)  q%-3s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
