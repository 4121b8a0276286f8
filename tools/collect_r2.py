"""Copy the results of tools/profile_r2.sh (gpurun_out/r2/) into profiles/r2/ and print the figures profiles/README.md
quotes (run here after the gpurun call)."""
import csv, glob, json, os, shutil
O, P = 'gpurun_out/r2', 'profiles/r2'
names = ['a_bench_default_n1', 'b_bench_under_rocprof', 'g_bench_one_rank_rccl'] + \
        ['c_bench_%s_n1' % v for v in ('exact', 'magnetic', 'no-hints', 'shared-image', 'upload', 'two-pass')]
for n in names:
    line = open(os.path.join(O, n + '.json')).read().strip().splitlines()[-1]
    open(os.path.join(P, n + '.json'), 'w').write(line + '\n')
newest = lambda pat: sorted(glob.glob(os.path.join(O, pat)), key=os.path.getmtime)[-1]
shutil.copy(newest('b_stats/*/*kernel_stats.csv'), os.path.join(P, 'b_kernel_stats_bench_default.csv'))
shutil.copy(newest('d_stats_exact/*/*kernel_stats.csv'), os.path.join(P, 'd_kernel_stats_bench_exact.csv'))
shutil.copy(newest('d_stats_magnetic/*/*kernel_stats.csv'), os.path.join(P, 'd_kernel_stats_bench_magnetic.csv'))
shutil.copy(os.path.join(O, 'p_pole_frames.txt'), os.path.join(P, 'p_pole_frames.txt'))
with open(os.path.join(P, 'g_timed_region_breakdown.txt'), 'w') as fp:
    fp.write(''.join(l for l in open(os.path.join(O, 'g_bench_one_rank_rccl.err')) if 'timed region' in l))
for n in names:
    d = json.load(open(os.path.join(P, n + '.json')))
    k = d['kernels']['k_georef_rows']
    print('%-28s %6d Mpx/s  %.4f ms/frame  kernel %s us  frac %.3f' % (
        n, d['value'], d['ms_per_step'], round(k['ms'] * 1e3, 1) if isinstance(k, dict) else '-', d['roofline']['frac']))
d = json.load(open(os.path.join(P, 'a_bench_default_n1.json')))
print('variants', {k: (round(v['Mpixels_per_s']), round(v['kernel_ms_per_frame'] * 1e3, 1)) for k, v in d['variants'].items()})
print('cpu', d['cpu_baseline']['value'], 'copy', round(d['roofline']['measured_copy_GBs']), 'fill', round(d['roofline']['measured_fill_GBs']),
      'parity', d['parity']['ok'], d['parity']['max_abs_dlat_dlon_deg'])
rows = []
with open(newest('b_stats/*/*kernel_trace.csv')) as fp:
    for r in csv.DictReader(fp):
        if 'k_georef_rows' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
rows.sort()
b = json.load(open(os.path.join(P, 'b_bench_under_rocprof.json')))
steps = b['steps']
n_last = 1 + (steps - 1 + 2) // 3
tot = sum(e - s for s, e in rows[-n_last:])
print('trace: %d launches %.1f ms in all; timed region: %d launches %.2f ms = %.1f us/frame; live %.1f us' % (
    len(rows), sum(e - s for s, e in rows) / 1e6, n_last, tot / 1e6, tot / steps / 1e3, b['kernels']['k_georef_rows']['ms'] * 1e3))
print(open(os.path.join(P, 'g_timed_region_breakdown.txt')).read().strip())
print(open(os.path.join(P, 'p_pole_frames.txt')).read().strip())
