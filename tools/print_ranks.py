"""Summary of an N-rank bench line (the ranks' own records).  usage: print_ranks.py line.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print('n_gpus %d  %.0f Mpixel/s  %.4f ms per step (regions %s)  ranks %s  distinct devices %s  loop %s' % (
    d['n_gpus'], d['value'], d['ms_per_step'], ' '.join('%.2f' % t for t in d.get('regions_ms', [])), d.get('ranks'),
    d.get('distinct_devices'), d['config']['frame_loop']))
for r in d.get('per_rank', []):
    print('  rank %d pid %d: own %.2f ms (process %.2f, gather %.2f, fence %.2f), kernel %.1f us per frame, host threads %s' % (
        r['rank'], r['pid'], r['elapsed_ms'], r['process_ms'], r['gather_ms'], r['closing_fence_ms'], r['kernel_us_per_frame'],
        {k: r['host_threads'][k] for k in ('cores_available', 'local_ranks', 'share', 'copy_threads', 'triangulator_threads')}))
up = d.get('variants', {}).get('upload')
if up:
    print('  upload: %.0f Mpixel/s, %.3f ms per frame, %s, PCIe GB/s per rank %s' % (
        up['Mpixels_per_s'], up['ms_per_frame'], up['frame_loop'], [round(v, 1) for v in up['pcie_GBs_per_rank']]))
