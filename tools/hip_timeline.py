"""
The host's HIP calls around the start and the end of the last 20-frame call in a trace made with
    rocprofv3 --kernel-trace --hip-trace -d DIR -o t -- python3 tools/host_overhead_probe.py
usage: python tools/hip_timeline.py DIR/t_results.db [frames]   (run it on the GPU box: the database is large)
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ker = list(db.execute("select name, start, end, grid_x, stream_id from kernels order by start"))
    big = [(s, e, g) for n, s, e, g, _ in ker if 'k_georef_rows' in n]
    per_frame = min(g for _, _, g in big)
    calls, cur = [], [big[0]]
    for b in big[1:]:
        if b[0] - cur[-1][1] > 60000:
            calls.append(cur)
            cur = [b]
        else:
            cur.append(b)
    calls.append(cur)
    calls = [c for c in calls if sum(b[2] // per_frame for b in c) == steps]
    c = calls[-2]                      # (the last one runs under cProfile)
    t0, t1 = c[0][0], c[-1][1]
    api = list(db.execute("select name, start, end, tid from regions where start > ? and start < ? order by start", (t0 - 600000, t1 + 400000)))
    print('first big kernel starts at 0, last big kernel ends at %.1f us' % ((t1 - t0) / 1e3))
    print('-- host calls before the first big kernel starts (and 30 us after)')
    for n, s, e, tid in api:
        if s < t0 + 30000:
            print('%9.1f .. %9.1f  %-34s tid %s' % ((s - t0) / 1e3, (e - t0) / 1e3, n, tid))
    print('-- kernels from 100 us before the last big kernel ends')
    for n, s, e, g, st in ker:
        if t1 - 100000 < s < t1 + 400000:
            print('%9.1f .. %9.1f  %-34s grid %d stream %s' % ((s - t1) / 1e3, (e - t1) / 1e3, n.split('(')[0][-34:], g, st))
    print('-- host calls from 20 us before the last big kernel ends (relative to its end)')
    for n, s, e, tid in api:
        if s > t1 - 20000:
            print('%9.1f .. %9.1f  %-34s tid %s' % ((s - t1) / 1e3, (e - t1) / 1e3, n, tid))


if __name__ == '__main__':
    main()
