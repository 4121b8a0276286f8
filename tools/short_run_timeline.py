"""The driver's invocation (20 frames): kernel trace of the timed region — every launch of the big kernel with its frames,
duration and the gap to the one before, what runs after the last one — next to the host's own clock.
usage (on the GPU box): rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --cpu-rows 0
--no-variants;  python3 tools/short_run_timeline.py DIR"""
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
big = [i for i, r in enumerate(rows) if 'k_georef_rows' in r['Kernel_Name']]
one = min(int(rows[i]['Grid_Size_X']) for i in big)
# the timed region: walk back from the last big kernel until 20 frames are covered
covered, first = 0, None
for i in reversed(big):
    covered += int(round(int(rows[i]['Grid_Size_X']) / float(one)))
    first = i
    if covered >= int(sys.argv[2]) if len(sys.argv) > 2 else covered >= 20:
        break
t0 = int(rows[first]['Start_Timestamp'])
prev_end = None
busy = 0
stop = big[-1] + 6
for r in rows[first:stop]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:60]
    is_big = 'k_georef_rows' in r['Kernel_Name']
    if is_big:
        busy += e - s
    print('%9.1f us  +%7.1f us  %-62s %s' % ((s - t0) / 1e3, (e - s) / 1e3, name,
                                             ('frames %d  gap %.1f us' % (round(int(r['Grid_Size_X']) / float(one)), (s - prev_end) / 1e3 if prev_end else 0)) if is_big else ''))
    if is_big:
        prev_end = e
last = max(int(r['End_Timestamp']) for r in rows[first:stop])
print('span first big kernel start -> last kernel end: %.1f us; big kernels %.1f us' % ((last - t0) / 1e3, busy / 1e3))
