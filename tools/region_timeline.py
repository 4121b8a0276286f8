"""
Where a short timed region's time goes on the GPU side: reads the database of
    rocprofv3 --kernel-trace -d DIR -o t -- python3 bench.py --steps 20 --warmup 5 --no-variants
and prints, for every timed region (a call of 1 + 19 frames' launches: the first launch carries one frame), the big kernel's
busy time, the gaps between its launches, the coarse pre-passes of the frames that have no finished neighbour yet (a repeated
region starts the sequence again) and the tail after the last big kernel (the finalise kernels of the last launch).
usage: python tools/region_timeline.py gpurun_out/r6_trace20/t_results.db [frames_per_region]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = list(db.execute("select name, start, end, grid_x from kernels order by start"))
    big = [(s, e, g) for n, s, e, g in rows if 'k_georef_rows' in n]
    per_frame = min(g for _, _, g in big)
    other = [(n, s, e) for n, s, e, g in rows if 'k_georef_rows' not in n]
    calls, cur = [], [big[0]]
    for b in big[1:]:
        if b[0] - cur[-1][1] > 60000:
            calls.append(cur)
            cur = [b]
        else:
            cur.append(b)
    calls.append(cur)
    regions = [c for c in calls if sum(b[2] // per_frame for b in c) == steps]
    print('%d calls, %d of them with %d frames; the last 7:' % (len(calls), len(regions), steps))
    print('%8s %8s %8s %8s %8s %8s %8s  %s' % ('span_us', 'busy_us', 'gaps_us', 'tail_us', 'coarse', 'n_launch', 'us/frame', 'frames per launch'))
    for c in regions[-7:]:
        t0, t1 = c[0][0], c[-1][1]
        busy = sum(e - s for s, e, _ in c)
        gaps = sum(c[i + 1][0] - c[i][1] for i in range(len(c) - 1))
        tail = max([e for n, s, e in other if t0 <= s <= t1 + 200000] + [t1]) - t1
        coarse = sum(1 for n, s, e in other if 'k_coarse_bbox' in n and t0 - 100000 <= s <= t1)
        print('%8.1f %8.1f %8.1f %8.1f %8d %8d %8.2f  %s' % ((t1 - t0) / 1e3, busy / 1e3, gaps / 1e3, tail / 1e3, coarse, len(c),
                                                             busy / 1e3 / steps, [b[2] // per_frame for b in c]))


if __name__ == '__main__':
    main()
