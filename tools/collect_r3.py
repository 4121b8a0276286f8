"""Copy the results of tools/profile_r3.sh (gpurun_out/r3/final/) into profiles/r3/, cut the per-launch extract of the timed
region out of the kernel trace (so that the per-frame kernel time can be recomputed from tracked files alone), refresh
profiles/traffic.json from the PMC passes and print the figures profiles/README.md quotes.  Run here after the gpurun call."""
import csv, glob, json, os, re, shutil
O, P = 'gpurun_out/r3/final', 'profiles/r3'
os.makedirs(P, exist_ok=True)
names = ['a_bench_default_n1', 'a3_bench_driver_command_steps20', 'b_bench_under_rocprof', 'g_bench_configs4_one_rank_rccl_steps32'] + \
        ['c_bench_%s_n1' % v for v in ('exact', 'magnetic', 'upload', 'two-pass')]
lines = {}
for n in names:
    try:
        line = open(os.path.join(O, n + '.json')).read().strip().splitlines()[-1]
        lines[n] = json.loads(line)
        open(os.path.join(P, n + '.json'), 'w').write(line + '\n')
    except Exception as e:
        print('missing', n, e)
newest = lambda pat: sorted(glob.glob(os.path.join(O, pat), recursive=True), key=os.path.getmtime)[-1]
shutil.copy(newest('b_stats/**/*kernel_stats.csv'), os.path.join(P, 'b_kernel_stats_bench_default.csv'))
shutil.copy(os.path.join(O, 'e_pmc_summary_fused_kernel.txt'), os.path.join(P, 'e_pmc_summary_per_launch.txt'))
with open(os.path.join(P, 'g_timed_region_breakdown.txt'), 'w') as fp:
    fp.write(''.join(l for l in open(os.path.join(O, 'g_bench_configs4_one_rank_rccl.err')) if 'timed region' in l))
shutil.copy('tools/profile_r3.sh', os.path.join(P, 'a_cmd.sh'))
# ---- per-launch extract of the timed region ------------------------------------------------------------------------------
rows = []
with open(newest('b_stats/**/*kernel_trace.csv')) as fp:
    for r in csv.DictReader(fp):
        if 'k_georef_rows' in r['Kernel_Name']:
            rows.append(r)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
b = lines['b_bench_under_rocprof']
steps = b['steps']
n_last = 1 + (steps - 1 + 2) // 3              # the first launch of a process() call carries one frame, the others three
wg_per_frame = None
out = []
for r in rows[-n_last:]:
    grid = int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0)
    wgsz = int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or 256)
    out.append((int(r['Dispatch_Id']), grid, wgsz, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
one = min(g for _, g, _, _, _ in out)
with open(os.path.join(P, 'b_timed_region_launches.csv'), 'w') as fp:
    fp.write('# the %d launches of k_georef_rows<true,false,0,2> in the timed region of b_bench_under_rocprof.json (%d frames), cut from the\n'
             '# kernel trace of the same rocprofv3 run; frames = grid size / grid size of a one-frame launch; gap = start - end of the launch before\n' % (n_last, steps))
    fp.write('dispatch_id,grid_size,workgroup_size,frames,duration_ns,gap_before_ns\n')
    prev_end = None
    for d, g, w, s, e in out:
        fp.write('%d,%d,%d,%d,%d,%s\n' % (d, g, w, round(g / one), e - s, '' if prev_end is None else s - prev_end))
        prev_end = e
tot = sum(e - s for _, _, _, s, e in out)
frames = sum(round(g / one) for _, g, _, _, _ in out)
print('timed region: %d launches, %d frames, %.3f ms of kernel = %.1f us per frame; live (HIP events): %.1f us' % (
    n_last, frames, tot / 1e6, tot / frames / 1e3, b['kernels']['k_georef_rows']['ms'] * 1e3))
# ---- traffic.json and the instruction mix from the PMC passes (per FRAME: a launch's grid size says how many it carries) -----
per_frame = {}
for path in sorted(glob.glob(os.path.join(O, 'e_pmc', '**', '*counter_collection.csv'), recursive=True)):
    acc = {}
    rows_ = [r for r in csv.DictReader(open(path)) if 'k_georef_rows<true, false, 0, 2>' in r['Kernel_Name'].replace('(anonymous namespace)::', '')]
    if not rows_:
        continue
    one_frame = min(int(r['Grid_Size']) for r in rows_)
    for r in rows_:
        a = acc.setdefault(r['Counter_Name'], [0.0, 0.0])
        a[0] += float(r['Counter_Value'])
        a[1] += round(int(r['Grid_Size']) / one_frame)
    for c, (v, f) in acc.items():
        per_frame[c] = v / f
val = lambda c: per_frame[c]
wr, fe = val('WRITE_SIZE'), val('FETCH_SIZE')
busy = 4 * val('SQ_ACTIVE_INST_VALU') / (val('GRBM_GUI_ACTIVE') / 8 * 1024)
tr = json.load(open('profiles/traffic.json'))
tr['k_georef_rows_fused'] = dict(fetch_kib=fe, write_kib=wr, fetch_correction=2.0, hbm_bytes=int((wr + 2 * fe) * 1024),
                                 valu_busy=round(busy, 3), source='profiles/r3/e_pmc_per_frame.txt (tools/collect_r3.py from the PMC passes of tools/profile_r3.sh)')
json.dump(tr, open('profiles/traffic.json', 'w'), indent=1)
mix = {k: val(k) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_TRANS_F64',
                           'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_INT64', 'SQ_INSTS_VALU_CVT')}
arith = sum(mix[k] for k in mix if k.endswith('F64'))
with open(os.path.join(P, 'e_pmc_per_frame.txt'), 'w') as fp:
    fp.write('# k_georef_rows<true,false,0,2> (fused, uint16 image), counters per FRAME of the bench loop (4240 x 2832): every PMC pass of\n'
             '# tools/profile_r3.sh sees launches of one, two and three frames (grid size / grid size of a one-frame launch)\n')
    for c in sorted(per_frame):
        fp.write('%-28s %.6g\n' % (c, per_frame[c]))
    t1 = 'traffic per frame: WRITE_SIZE %.0f KiB + FETCH_SIZE %.0f KiB x 2 (gfx950 half count) = %.1f MB; VALU busy %.3f' % (
        wr, fe, (wr + 2 * fe) * 1024 / 1e6, busy)
    t2 = 'VALU instructions per frame %.3g: FP64 add %.1f %% mul %.1f %% fma %.1f %% transcendental %.1f %% | int %.1f %% cvt %.1f %% | rest (moves, compares, selects, DPP) %.1f %%' % (
        mix['SQ_INSTS_VALU'], *[100 * mix[k] / mix['SQ_INSTS_VALU'] for k in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_TRANS_F64')],
        100 * (mix['SQ_INSTS_VALU_INT32'] + mix['SQ_INSTS_VALU_INT64']) / mix['SQ_INSTS_VALU'], 100 * mix['SQ_INSTS_VALU_CVT'] / mix['SQ_INSTS_VALU'],
        100 * (1 - (arith + mix['SQ_INSTS_VALU_INT32'] + mix['SQ_INSTS_VALU_INT64'] + mix['SQ_INSTS_VALU_CVT']) / mix['SQ_INSTS_VALU']))
    fp.write('# ' + t1 + '\n# ' + t2 + '\n')
print(t1)
print(t2)
for n in names:
    d = lines.get(n)
    if d is None:
        continue
    k = d['kernels']['k_georef_rows']
    print('%-42s %6d Mpx/s  %.4f ms/frame  kernel %s us  frac %.3f' % (
        n, d['value'], d['ms_per_step'], round(k['ms'] * 1e3, 1) if isinstance(k, dict) else '-', d['roofline']['frac']))
d = lines['a_bench_default_n1']
print('variants', {k: (round(v['Mpixels_per_s']), round(v['kernel_ms_per_frame'] * 1e3, 1)) for k, v in d['variants'].items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('n_process', {}).get('value'), 'copy', round(d['roofline']['measured_copy_GBs']),
      'fill', round(d['roofline']['measured_fill_GBs']), 'parity', d['parity']['ok'], d['parity']['max_abs_dlat_dlon_deg'])
print(open(os.path.join(P, 'g_timed_region_breakdown.txt')).read().strip())
