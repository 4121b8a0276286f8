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

# PMC per FRAME: every counter's sum over the launches of the fused kernel divided by the frames those launches covered
# (a launch covers grid size / one frame's grid size frames)
import collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(SRC, 'e_pmc', '**', '*counter_collection.csv'), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if 'k_georef_rows' in r['Kernel_Name']]
    if not rows:
        continue
    one = min(int(r['Grid_Size']) for r in rows)
    for r in rows:
        a = acc[r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += int(round(int(r['Grid_Size']) / float(one)))
if acc:
    with open(os.path.join(DST, 'e_pmc_per_frame.txt'), 'w') as fp:
        fp.write('k_georef_rows<true, false, 0, 2> (bench.py default workload), PMC per FRAME (sum over launches / frames covered)\n')
        for k in sorted(acc):
            fp.write('   %-28s %.5g  (frames %d)\n' % (k, acc[k][0] / acc[k][1], acc[k][1]))
        w, f = acc.get('WRITE_SIZE'), acc.get('FETCH_SIZE')
        if w and f:
            hbm = (w[0] / w[1] + 2 * f[0] / f[1]) * 1024
            fp.write('HBM traffic per frame: WRITE_SIZE + 2 x FETCH_SIZE (gfx950 half count) = %.1f MB\n' % (hbm / 1e6))
        v, g = acc.get('SQ_ACTIVE_INST_VALU'), acc.get('GRBM_GUI_ACTIVE')
        if v and g:
            fp.write('VALU busy = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024) = %.3f\n' % (4 * (v[0] / v[1]) / ((g[0] / g[1]) / 8 * 1024)))
        n, fma, mul, add = (acc.get(k) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64'))
        if n and fma and mul and add:
            per = n[0] / n[1]
            fp.write('VALU instructions per frame %.4g: FP64 fma %.1f %%, mul %.1f %%, add %.1f %%\n' % (
                per, 100 * fma[0] / fma[1] / per, 100 * mul[0] / mul[1] / per, 100 * add[0] / add[1] / per))
    print(open(os.path.join(DST, 'e_pmc_per_frame.txt')).read())
