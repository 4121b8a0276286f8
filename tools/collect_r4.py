"""Copies the round-4 profile set (tools/profile_r4.sh -> gpurun_out/r4/final) into profiles/r4/ and cuts the per-launch extract
of the timed region out of the kernel traces (the launches of k_georef_rows after the spin-up: the last ceil(192 / 3) + 1)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, 'gpurun_out', 'r4', 'final'), os.path.join(ROOT, 'profiles', 'r4')
os.makedirs(DST, exist_ok=True)
for name in sorted(os.listdir(SRC)):
    if name.endswith('.json'):
        # (a line of its own: whatever a library printed before it is dropped)
        lines = [ln for ln in open(os.path.join(SRC, name)).read().splitlines() if ln.startswith('{"metric"')]
        if lines:
            open(os.path.join(DST, name), 'w').write(lines[-1] + '\n')
    elif name == 'e_pmc_summary_per_launch.txt':
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, name))
shutil.copy(os.path.join(ROOT, 'tools', 'profile_r4.sh'), os.path.join(DST, 'a_cmd.sh'))


def extract(stats_dir, out_csv, frames):
    traces = glob.glob(os.path.join(SRC, stats_dir, '**', '*kernel_trace.csv'), recursive=True)
    stats = glob.glob(os.path.join(SRC, stats_dir, '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(DST, out_csv.replace('timed_region_launches', 'kernel_stats')))
    if not traces:
        return None
    rows = [r for r in csv.DictReader(open(traces[0])) if 'k_georef_rows' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    one = min(int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']) for r in rows)
    # walk back from the end until `frames` frames are covered
    picked, covered = [], 0
    for r in reversed(rows):
        g = int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size'])
        n = int(round(g / float(one)))
        picked.append((r, n))
        covered += n
        if covered >= frames:
            break
    picked.reverse()
    total = 0
    with open(os.path.join(DST, out_csv), 'w') as fp:
        fp.write('dispatch_id,kernel,grid_size,frames,duration_ns,gap_to_previous_ns\n')
        prev_end = None
        for r, n in picked:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            total += e - s
            fp.write('%s,"%s",%s,%d,%d,%s\n' % (r.get('Dispatch_Id', ''), r['Kernel_Name'].split('(')[0], r.get('Grid_Size_X', r.get('Grid_Size')), n,
                                                e - s, '' if prev_end is None else s - prev_end))
            prev_end = e
    return covered, total


for stats_dir, out_csv in (('b_stats', 'b_timed_region_launches.csv'), ('b_stats_magnetic', 'b_timed_region_launches_magnetic.csv')):
    got = extract(stats_dir, out_csv, 192)
    if got:
        print('%s: %d launch-frames, %.3f ms in the kernel = %.1f us per frame' % (out_csv, got[0], got[1] / 1e6, got[1] / 1e3 / got[0]))
for name in ('a_bench_default_n1', 'a3_bench_driver_command_steps20', 'c_bench_magnetic_n1', 'c_bench_magnetic_nine_arrays_n1'):
    p = os.path.join(DST, name + '.json')
    if os.path.exists(p):
        d = json.load(open(p))
        print(name, '%.0f Mpx/s' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel us/frame %.1f' % (d['kernels']['k_georef_rows']['ms'] * 1e3),
              'frac %.3f' % d['roofline']['frac'])
