"""Copies the round-6 profile set (tools/profile_r6.sh -> gpurun_out/r6/final) into profiles/r6/, cuts the per-launch extract of
the timed region out of the kernel trace, sums the PMC passes per FRAME and refreshes profiles/traffic.json — every entry with
the hash of the kernel sources it was measured on (bench.py reports an entry's traffic only when that hash is the running build's)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC, DST = os.path.join(ROOT, 'gpurun_out', 'r6', 'final'), os.path.join(ROOT, 'profiles', 'r6')
os.makedirs(DST, exist_ok=True)
for name in sorted(os.listdir(SRC)):
    if name.endswith('.json'):
        lines = [ln for ln in open(os.path.join(SRC, name)).read().splitlines() if ln.startswith('{"metric"')]
        if lines:
            open(os.path.join(DST, name), 'w').write(lines[-1] + '\n')
    elif name in ('e_pmc_summary_per_launch.txt', 'k_class_api.txt', 'n_cubic_full_size.txt'):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, name))
shutil.copy(os.path.join(ROOT, 'tools', 'profile_r6.sh'), os.path.join(DST, 'a_cmd.sh'))


def grid_of(r):
    return int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size'])


def extract(stats_dir, out_csv, frames):
    traces = glob.glob(os.path.join(SRC, stats_dir, '**', '*kernel_trace.csv'), recursive=True)
    stats = glob.glob(os.path.join(SRC, stats_dir, '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(DST, out_csv.replace('timed_region_launches', 'kernel_stats')))
    if not traces:
        return None
    rows = [r for r in csv.DictReader(open(traces[0])) if 'k_georef_rows' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    one = min(grid_of(r) for r in rows)
    picked, covered = [], 0
    for r in reversed(rows):
        n = int(round(grid_of(r) / float(one)))
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
            fp.write('%s,"%s",%s,%d,%d,%s\n' % (r.get('Dispatch_Id', ''), r['Kernel_Name'].split('(')[0], grid_of(r), n, e - s,
                                                '' if prev_end is None else s - prev_end))
            prev_end = e
    return covered, total


got = extract('b_stats', 'b_timed_region_launches.csv', 192)
if got:
    print('b_timed_region_launches.csv: %d launch-frames, %.3f ms in the kernel = %.1f us per frame' % (got[0], got[1] / 1e6, got[1] / 1e3 / got[0]))
for name in ('a_bench_default_n1', 'a3_bench_driver_command_steps20', 'c_bench_magnetic_n1', 'c_bench_two-pass_n1', 'c_bench_upload_n1'):
    p = os.path.join(DST, name + '.json')
    if os.path.exists(p):
        d = json.load(open(p))
        print(name, '%.0f Mpx/s' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel us/frame %.1f' % (d['kernels']['k_georef_rows']['ms'] * 1e3),
              'frac %.3f' % d['roofline']['frac'])


def per_frame(pmc_dir, kernel_part):
    """{counter: value per frame} of the kernel whose name contains `kernel_part`, over all counter sets under pmc_dir."""
    acc = collections.defaultdict(lambda: [0.0, 0])
    name = None
    for path in glob.glob(os.path.join(SRC, pmc_dir, '**', '*counter_collection.csv'), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if kernel_part in r['Kernel_Name']]
        if not rows:
            continue
        name = rows[0]['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        one = min(int(r['Grid_Size']) for r in rows)
        for r in rows:
            a = acc[r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += int(round(int(r['Grid_Size']) / float(one)))
    return name, {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def summary(fp, title, vals, frames):
    fp.write(title + '\n')
    for k in sorted(vals):
        fp.write('   %-28s %.5g  (frames %d)\n' % (k, vals[k], frames[k]))
    out = {}
    if 'WRITE_SIZE' in vals and 'FETCH_SIZE' in vals:
        out['write_kib'], out['fetch_kib'] = vals['WRITE_SIZE'], vals['FETCH_SIZE']
        out['fetch_correction'] = 2.0
        out['hbm_bytes'] = int(round((vals['WRITE_SIZE'] + 2 * vals['FETCH_SIZE']) * 1024))
        fp.write('HBM traffic per frame: WRITE_SIZE + 2 x FETCH_SIZE (gfx950 half count) = %.1f MB\n' % (out['hbm_bytes'] / 1e6))
    if 'SQ_ACTIVE_INST_VALU' in vals and 'GRBM_GUI_ACTIVE' in vals:
        out['valu_busy'] = round(4 * vals['SQ_ACTIVE_INST_VALU'] / (vals['GRBM_GUI_ACTIVE'] / 8 * 1024), 3)
        fp.write('VALU busy = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024) = %.3f\n' % out['valu_busy'])
    if 'SQ_INSTS_VALU' in vals:
        out['valu_insts'] = vals['SQ_INSTS_VALU']
    if all(k in vals for k in ('SQ_INSTS_VALU', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64')):
        per = vals['SQ_INSTS_VALU']
        fp.write('VALU instructions per frame %.4g: FP64 fma %.1f %%, mul %.1f %%, add %.1f %%\n' % (
            per, 100 * vals['SQ_INSTS_VALU_FMA_F64'] / per, 100 * vals['SQ_INSTS_VALU_MUL_F64'] / per, 100 * vals['SQ_INSTS_VALU_ADD_F64'] / per))
    return out


import bench
sha = bench.kernel_sources_sha16()
tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
traffic = json.load(open(tpath))
updates = {}
name, vals, frames = per_frame('e_pmc', 'k_georef_rows')
if vals:
    with open(os.path.join(DST, 'e_pmc_per_frame.txt'), 'w') as fp:
        updates['k_georef_rows_fused'] = summary(fp, '%s (bench.py default workload), PMC per FRAME (sum over launches / frames covered)' % name, vals, frames)
        updates['k_georef_rows_fused']['source'] = 'profiles/r6/e_pmc_per_frame.txt (tools/collect_r6.py from the PMC passes of tools/profile_r6.sh 1)'
    print(open(os.path.join(DST, 'e_pmc_per_frame.txt')).read())
with open(os.path.join(DST, 'e_pmc_variants.txt'), 'w') as fp:
    for pmc_dir, part, key, what in (('p_pmc_two', 'k_georef_rows', 'k_georef_rows', 'georef only (bench.py --plan two-pass --streams 1, one frame per launch)'),
                                     ('p_pmc_two', 'k_bin_frame', 'k_bin_frame', 'k_bin_frame (the same runs)'),
                                     ('p_pmc_magonly', 'k_georef_rows', 'k_georef_rows_fused_mag_only', 'MLat/MLT only (bench.py --magnetic, one frame per launch)'),
                                     ('p_pmc_nine', 'k_georef_rows', 'k_georef_rows_fused_mag', 'nine arrays (bench.py --magnetic --nine-arrays, one frame per launch)')):
        name, vals, frames = per_frame(pmc_dir, part)
        if not vals:
            continue
        updates[key] = summary(fp, '== %s: %s, PMC per FRAME' % (name, what), vals, frames)
        updates[key]['source'] = 'profiles/r6/e_pmc_variants.txt (tools/profile_r6.sh 2: %s)' % what
if os.path.getsize(os.path.join(DST, 'e_pmc_variants.txt')):
    print(open(os.path.join(DST, 'e_pmc_variants.txt')).read())
for key, u in updates.items():
    if 'hbm_bytes' in u:
        u['sources_sha16'] = sha
        traffic[key] = u
traffic['_comment'] = ('HBM bytes per FRAME from rocprofv3 PMC passes (4240x2832 frame): WRITE_SIZE + 2 x FETCH_SIZE in KiB (per MI355X_MICROARCH.md '
                       'FETCH_SIZE counts half of the bytes of wide coalesced streaming reads on gfx950: read sides are doubled; 8-B-per-lane stores '
                       'match the byte count exactly).  valu_busy = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) from the same '
                       'runs.  sources_sha16: bench.kernel_sources_sha16() of the build that was measured — bench.py reports an entry only for '
                       'that build.  Entries without it date from rounds 1-4.')
json.dump(traffic, open(tpath, 'w'), indent=1)
print('traffic.json entries refreshed for sources', sha, ':', sorted(k for k, u in updates.items() if 'hbm_bytes' in u))
