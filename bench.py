#!/usr/bin/env python
"""
bench.py — georef + resample throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one synthetic 4240x2832 ISS-like frame through the whole hot path on one GPU:
fused georeferencing (WCS -> ray -> inflated-WGS84 hit -> geodetic lat/lon of corners and centres,
elevation), maskedByElevation(10), bounding box, 0.1 deg plate-carree grid, binned mean of the
uint16 RGB image + elevation (BASELINE.json configs[2]; configs[1] is its first kernel).  The image
is resident in HBM before the timed region; per-frame host set-up (matrices, grid) is inside it.
With N > 1 every rank processes its own frames (weak scaling) and the per-frame grids are gathered
on rank 0 over RCCL inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 4240, 2832
ALTITUDE, MIN_ELEV, PPD = 110, 10.0, 10
TIMING_EVERY = 1        # events ride on the dispatch packets (hipExtLaunchKernelGGL): every launch is timed
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def algorithmic_bytes(width, height, nchan=3, pix_bytes=2):
    """SURVEY.md §8d contract figures (f64 coordinates)."""
    nc, npx = (width + 1) * (height + 1), width * height
    return dict(image=nchan * pix_bytes * npx,
                mag=16 * nc + 16 * npx,                  # mlat, mlt corners + mlat_c, mlt_c centres written
                georef=16 * nc + 24 * npx,              # WCS-fused: lat, lon corners + latC, lonC, elev written
                georef_dirs_in=40 * nc + 24 * npx,      # + 24 B/corner direction read (contract row "directions-in")
                resample=(24 + nchan * pix_bytes) * npx)


def measured_copy_gbs(device):
    """Device-to-device copy rate (read + write bytes per second) of this GPU, for reading the roofline fraction
    against something achievable beside the 8 TB/s spec figure.  Outside the timed region."""
    import torch
    n = 1 << 27                                    # 1 GiB of float64 each way
    a = torch.empty(n, dtype=torch.float64, device=device).fill_(1.0)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize()
    return 10 * 2 * n * 8 / (time.perf_counter() - t0) / 1e9


def cpu_baseline(sample_rows, frames=4):
    """Oracle (NumPy restatement of the reference) on the same workload, 1 core: `frames` frames of the synthetic
    sequence (rows [0, sample_rows) of each), about 12 s of CPU work at the full frame height."""
    from oracle import ref_numpy as O
    from auromat_amd.synthetic import sequence_frame, frame_image
    from auromat_amd.coordinates import transform as T
    t_geo = t_res = 0.0
    for k in range(frames):
        hdr, cam, t, seed = sequence_frame(k, WIDTH, HEIGHT)
        # crop: same pixels as the top `sample_rows` rows of the full frame
        hdr = dict(hdr, IMAGEH=sample_rows)
        img = frame_image(WIDTH, HEIGHT, seed=seed)[:sample_rows]
        t0 = time.time()
        et = T.date2es(t)
        g = O.georef_frame(hdr, ALTITUDE, cam, O.mat_j2000_to_geo(et), None, fast=True)
        t1 = time.time()
        corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), MIN_ELEV)
        bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
        data = np.dstack((img.astype(np.float64), g['elev']))
        data[center_mask] = np.nan
        O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), ALTITUDE,
                        data, None, bbox, (PPD, PPD), disc, False)
        t2 = time.time()
        t_geo += t1 - t0
        t_res += t2 - t1
        del g, data
    npx = WIDTH * sample_rows * frames
    return dict(value=npx / 1e6 / (t_geo + t_res), unit='Mpixels/s', cores=1, kind='port',
                sample='%d frames of the sequence, rows 0..%d of %dx%d each (%.1f Mpx): georef %.1f s + mask/resample '
                       '%.1f s, NumPy, 1 thread' % (frames, sample_rows, WIDTH, HEIGHT, npx / 1e6, t_geo, t_res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--cpu-rows', type=int, default=2832, help='rows of the frame the CPU baseline processes (0 = skip)')
    ap.add_argument('--exact', action='store_true', help='exact centre rays instead of fast centres')
    ap.add_argument('--magnetic', action='store_true',
                    help='configs[3] instead of configs[2]: MLat/MLT outputs and the (MLat, SM longitude) grid of '
                         'resampleMLatMLT; the three altitude shells 100/110/120 km alternate frame by frame')
    ap.add_argument('--plan', default='fused', choices=('fused', 'two-pass'),
                    help='fused: binning inside the georeferencing kernel (superset grid + crop); '
                         'two-pass: separate binning kernel that re-reads the centre arrays')
    ap.add_argument('--no-hints', action='store_true',
                    help='run the coarse bounding-box pre-pass for every frame instead of re-using the previous exact box')
    ap.add_argument('--batch', type=int, default=3, choices=(1, 2, 3),
                    help='frames per launch of the big kernel in the fused plan (amt_pipe_launch_many)')
    ap.add_argument('--upload', action='store_true',
                    help='PCIe-inclusive variant (never the headline value): every frame brings its own image from '
                         'pinned host memory, uploaded on a copy stream beside the previous frames\' kernels')
    ap.add_argument('--streams', type=int, default=2, choices=(1, 2),
                    help='2: bin frame k beside the ray casting of frame k+1 on a second HIP stream')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    if rank != 0:
        # only rank 0 reports; libraries that write to stdout (RCCL prints a version banner with C stdio) must not
        # interleave with its JSON line in the launcher's merged output
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    torch.cuda.set_device(local_rank)
    # AMT_BENCH_FORCE_DIST=1 exercises the RCCL gather path with a single rank (boxes with one GPU)
    use_dist = world > 1 or bool(os.environ.get('AMT_BENCH_FORCE_DIST'))
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.sequence import gather_device
    from auromat_amd.synthetic import sequence_frame, frame_image

    # The product's own sequence loop (auromat_amd/pipeline.py): two frame buffers, frame k+1 is georeferenced
    # while the host finishes frame k and prepares frame k+2.  plan=fused: binning inside the georeferencing
    # kernel via the native frame driver; plan=two-pass: separate binning kernel, on a second HIP stream beside
    # the next frame's ray casting when --streams 2.
    fast = not args.exact
    total = args.warmup + args.steps
    fused = args.plan == 'fused'
    seq = SequencePipeline(WIDTH, HEIGHT, altitude=ALTITUDE, fast=fast, min_elevation=MIN_ELEV, pxPerDeg=PPD,
                           plan='single-pass' if fused else 'two-pass', bin_stream=args.streams == 2,
                           # resident before the timed region (unless --upload brings one per frame)
                           shared_image=None if args.upload else frame_image(WIDTH, HEIGHT, seed=rank),
                           magnetic=args.magnetic, batch=args.batch)
    seq.use_hints = not args.no_hints
    ctx = seq.ctx
    # the synthetic sequence (what a reader would hand over: WCS cards, camera position, time) exists before the
    # timed region, like the image; everything derived from it (matrices, grids) is computed inside
    frames = [sequence_frame(rank * total + k, WIDTH, HEIGHT)[:3] + (None,) for k in range(total)]
    if args.upload:
        # four distinct images in pinned host memory (uint16 bits as int16, the frame buffer's layout), cycled
        host_imgs = [torch.from_numpy(frame_image(WIDTH, HEIGHT, seed=100 + i).view(np.int16)).pin_memory()
                     for i in range(4)]
        frames = [f[:3] + (host_imgs[k % 4],) for k, f in enumerate(frames)]
    if args.magnetic:
        # per-frame shells: the matrices are made here (outside the timed region only in this variant, because the
        # sequence loop takes one altitude); 100 / 110 / 120 km alternate
        from auromat_amd.mapping.astrometry import frame_params
        frames = [(frame_params(h, (100, 110, 120)[k % 3], c, t, fast, magnetic=True), c, t, im)
                  for k, (h, c, t, im) in enumerate(frames)]

    warm = seq.process(frames[:args.warmup])
    if use_dist and warm:
        # first-call costs of the gather path (RCCL channels, torch.cat, allocations) belong to the warm-up as well
        gather_device(warm, [rank * total + k for k in range(len(warm))], ctx.device)
    del warm
    ctx.timing_enable(TIMING_EVERY)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    results = seq.process(frames[args.warmup:])
    plans = list(seq.plans)
    hinted = seq.hinted
    gathered = None
    if use_dist:
        # device-to-device over xGMI; rank 0 unpacks to the host after the timed region
        gathered = gather_device(results, [rank * total + args.warmup + k for k in range(args.steps)], ctx.device)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if rank == 0:
            assert gathered.n_frames == world * args.steps and len(gathered.unpack()) == world * args.steps

    # kernel durations measured live over the timed region: HIP events recorded by the library directly
    # around each k_georef_rows / k_bin_frame launch, on the stream they run on
    g_total, g_n = ctx.timing_read(0)
    b_total, b_n = ctx.timing_read(1)
    n_timed = (args.steps + TIMING_EVERY - 1) // TIMING_EVERY
    assert g_n == n_timed and b_n in (0, n_timed), (g_n, b_n)
    georef_ms, bin_ms = g_total / g_n, (b_total / b_n if b_n else 0.0)
    ctx.timing_enable(False)

    if rank == 0:
        npx = WIDTH * HEIGHT
        copy_gbs = measured_copy_gbs(ctx.device)
        ab = algorithmic_bytes(WIDTH, HEIGHT)
        info = ctx.device_info()
        res = results[-1]
        traffic = {}
        try:
            with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as fp:
                traffic = json.load(fp)
        except (IOError, ValueError):
            pass

        def frac(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

        if fused:
            # One kernel does both stages.  Algorithmic bytes = SURVEY.md 8d: "georef, WCS-fused variant" (16 B per
            # corner + 24 B per pixel written) + "resample-mean" (30 B per pixel read) = 840.6 MB per frame (MLat/MLT
            # config: the same 16 Nc + 54 Np).  The fused kernel never re-reads the centre arrays, so it MOVES less:
            # 480.4 MB written + the 72.0 MB image read = 552.5 MB (`bytes_moved_min`; `traffic` is what PMC counted).
            kname, tkey = 'k_georef_rows<BIN> (amt_georef_frame with fused binning, via amt_pipe_launch)', 'k_georef_rows_fused'
            kbytes = ab['georef'] + ab['resample']
            moved = ab['georef'] + ab['image']
        else:
            kname, tkey = 'k_georef_rows (amt_georef_frame)', 'k_georef_rows'
            kbytes = moved = ab['georef']
        if args.magnetic:
            moved += ab['mag']           # this variant writes MLat/MLT beside lat/lon (more than the contract counts)
            tkey += '_mag'               # profiles/r1/k_pmc_summary_magnetic.txt
        achieved = kbytes / (georef_ms * 1e-3) / 1e9
        fpl = seq.batch if fused else 1
        out = {
            'metric': 'Mpixels/s georef+resample, 4240x2832 frame',
            'value': world * args.steps * npx / 1e6 / elapsed,
            'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic' + (', image of every frame uploaded from pinned host memory' if args.upload else ''),
            'config': {'workload': ('configs[3]: as configs[2] on shells 100/110/120 km + MLat/MLT of corners and centres, '
                                    'mean-resample on the 0.1 deg (MLat, SM longitude) grid (resampleMLatMLT)'
                                    if args.magnetic else
                                    'configs[2]: 4240x2832 ISS-like frame, WCS ray cast + WGS84(+110 km) '
                                    'intersection + geodetic transform + elevation (%s centres), '
                                    'maskedByElevation(10), mean-resample to 0.1 deg plate-carree, uint16 RGB'
                                    % ('fast' if fast else 'exact')),
                       'frame': [WIDTH, HEIGHT], 'px_per_deg': PPD, 'grid': list(res['mean'].shape),
                       'plan': args.plan, 'frames_per_launch': seq.batch, 'frames_without_prepass': hinted,
                       'single_pass_frames': sum(1 for q in plans if q == 'single-pass'),
                       'parallelism': 'frames sharded over %d GPU(s), RCCL gather of grids' % world,
                       'device': info['name']},
            # dominant kernel.  It is FP64-VALU bound (about 400 VALU instructions per pixel row and lane, see
            # DESIGN.md and profiles/), so the HBM fraction understates how busy the chip is.
            'roofline': {'bound': 'hbm', 'kernel': kname, 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'measured_copy_GBs': copy_gbs, 'fp64_vector_peak_TFLOPs': 78.6,
                         # one launch covers `frames_per_launch` frames: bytes, traffic and duration are per launch
                         'frames_per_launch': fpl,
                         'traffic': (traffic.get(tkey, {}).get('hbm_bytes') or 0) * fpl or None,
                         'algorithmic_bytes': kbytes * fpl, 'ms_per_launch': georef_ms * fpl,
                         # the bytes this kernel has to move at the very least, and the same fraction on that basis
                         'bytes_moved_min': moved * fpl, 'frac_bytes_moved_min': frac(moved, georef_ms),
                         'frames_timed': g_n, 'valu_busy': traffic.get(tkey, {}).get('valu_busy')},
            'kernels': {
                'k_georef_rows': {'ms': georef_ms, 'algorithmic_bytes': kbytes, 'frac_hbm_peak': frac(kbytes, georef_ms),
                                  'bytes_moved_min': moved},
                'k_bin_frame': ({'ms': bin_ms, 'algorithmic_bytes': ab['resample'],
                                 'frac_hbm_peak': frac(ab['resample'], bin_ms),
                                 'traffic': traffic.get('k_bin_frame', {}).get('hbm_bytes')} if bin_ms else
                                'not launched: binning is fused into k_georef_rows (plan=fused)'),
                # SURVEY.md 8d contract figures for the whole pipeline against the time of the kernel(s) that do it
                'pipeline_frac_wcs_fused_840.6MB': frac(ab['georef'] + ab['resample'], georef_ms + bin_ms),
                'pipeline_frac_directions_in_1129.0MB': frac(ab['georef_dirs_in'] + ab['resample'], georef_ms + bin_ms),
            },
        }
        if world == 1 and args.cpu_rows > 0:
            out['cpu_baseline'] = cpu_baseline(min(args.cpu_rows, HEIGHT))
        else:
            out['cpu_baseline'] = None
        # the JSON line is the LAST thing on stdout: flush what C libraries have buffered there (RCCL's banner),
        # print, and close the descriptor for whatever they print while shutting down
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
