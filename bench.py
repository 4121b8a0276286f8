#!/usr/bin/env python
"""
bench.py — georef + resample throughput on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 and no WORLD_SIZE in the environment this process touches no GPU: it starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py`` as a child,
passes rank 0's JSON line through and exits with the child's code.  Under torch.distributed.run (the driver's own
launch form) every rank runs :func:`main` directly.

One step = one synthetic 4240x2832 ISS-like frame through the whole hot path on one GPU:
fused georeferencing (WCS -> ray -> inflated-WGS84 hit -> geodetic lat/lon of corners and centres,
elevation), maskedByElevation(10), bounding box, 0.1 deg plate-carree grid, binned mean of the
uint16 RGB image + elevation (BASELINE.json configs[2]; configs[1] is its first kernel).  Every frame has its OWN
image, resident in HBM before the timed region (pointing, time, camera and image differ frame by
frame); per-frame host set-up (matrices, grid) is inside the timed region.  With N > 1 every rank processes K frames
of its own (weak scaling; K defaults to 192 for every N; `--gpus 8 --steps 32` is the 256 frames of configs[4]) and the per-frame grids are
gathered on rank 0 over RCCL inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 4240, 2832
ALTITUDE, MIN_ELEV, PPD = 110, 10.0, 10
SHELLS = (100, 110, 120)  # configs[3]
TIMING_EVERY = 1        # events ride on the dispatch packets (hipExtLaunchKernelGGL): every launch is timed
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
LAUNCH_STREAMS = 1      # --launch-streams
MAX_RESIDENT_IMAGES = 128  # distinct images kept in HBM (72 MB each); longer runs cycle through them
# the pipelines' own coordinate buffers in strip-padded rows (amt_georef_out.row_layout; AMT_PADDED_ROWS=0: contiguous, A/B runs)
PADDED = os.environ.get('AMT_PADDED_ROWS', '1') != '0'
# The K-step timed region (barrier + synchronize, K steps, barrier + synchronize) runs REPEATS times back to back on the same frames
# in the same warm state, and the line reports the MEDIAN region: a 20-step region lasts 3 ms, and single regions on one box
# differ by more than the effect of a round's optimisations (-5.6 % between two rounds with no kernel change)
REPEATS = 7


def median_index(values):
    order = sorted(range(len(values)), key=lambda i: values[i])
    return order[len(order) // 2]


def algorithmic_bytes(width, height, nchan=3, pix_bytes=2):
    """SURVEY.md §8d contract figures (f64 coordinates)."""
    nc, npx = (width + 1) * (height + 1), width * height
    return dict(image=nchan * pix_bytes * npx,
                mag=16 * nc + 16 * npx,                  # mlat, mlt corners + mlat_c, mlt_c centres written
                georef=16 * nc + 24 * npx,              # WCS-fused: lat, lon corners + latC, lonC, elev written
                georef_dirs_in=40 * nc + 24 * npx,      # + 24 B/corner direction read (contract row "directions-in")
                resample=(24 + nchan * pix_bytes) * npx,
                # SURVEY 8d "MLat/MLT + resample per shell": 40 Nc + 54 Np
                mag_shell=40 * nc + 54 * npx)


def kernel_sources_sha16():
    """Identifies the kernel sources a PMC pass was made on (profiles/traffic.json records it per entry): traffic counted on
    another version of the kernels is not reported as this run's."""
    import hashlib
    h = hashlib.sha256()
    for name in ('amt_georef.hip', 'amt_common.h', 'amt_binning.hip'):
        with open(os.path.join(ROOT, 'auromat_amd', 'csrc', name), 'rb') as fp:
            h.update(fp.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='frames per rank in the timed region (default 192 for every N: weak scaling; BASELINE.json configs[4], the '
                         '256-frame job on 8 GPUs, is --gpus 8 --steps 32)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--cpu-rows', type=int, default=2832, help='rows of the frame the CPU baseline processes (0 = skip)')
    ap.add_argument('--exact', action='store_true', help='exact centre rays instead of fast centres')
    ap.add_argument('--magnetic', action='store_true',
                    help='configs[3] instead of configs[2]: MLat/MLT outputs and the (MLat, SM longitude) grid of '
                         'resampleMLatMLT; the three altitude shells 100/110/120 km alternate frame by frame')
    ap.add_argument('--nine-arrays', action='store_true',
                    help='with --magnetic: also write lat, lon, lat_c, lon_c (nine per-pixel arrays instead of the five '
                         'BASELINE.md\'s configs[3] row lists; the kernel then runs both coordinate chains)')
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
    ap.add_argument('--shared-image', action='store_true',
                    help='round-1 conditions: one resident image shared by all frames (A/B only)')
    ap.add_argument('--streams', type=int, default=2, choices=(1, 2),
                    help='2: bin frame k beside the ray casting of frame k+1 on a second HIP stream (two-pass plan)')
    ap.add_argument('--launch-streams', type=int, default=1, choices=(1, 2),
                    help='fused plan: 2 = consecutive launches of the big kernel alternate between two HIP streams')
    ap.add_argument('--no-variants', action='store_true',
                    help='skip the short extra runs (exact centres, configs[3]) whose figures the JSON line carries as '
                         '"variants" (they run on rank 0 at N = 1 only, after the timed region)')
    ap.add_argument('--spinup-ms', type=float, default=400.0,
                    help='untimed: before the W warm-up steps the same loop runs for about this long, so that the clocks '
                         'and the memory system are in their sustained state when the timed region starts (a cold chip '
                         'runs its first few dozen launches 3-5 %% slower; MI355X_MICROARCH.md asks for 2 s of back-to-back '
                         'launches before trusting an in-kernel clock); 0 disables')
    ap.add_argument('--cpu-procs', type=int, default=16,
                    help='processes of the N-process CPU baseline (cpu_baseline.n_process: one oracle process per frame, the '
                         'reference\'s only data parallelism, SURVEY 8d (ii)); 0 = skip')
    ap.add_argument('--cpu-worker', type=int, default=None, help=argparse.SUPPRESS)      # internal: one such process
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / reporting path only: gloo on the CPU, no GPU, a step is a sleep (tests)')
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 192
    return args


# ------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_children(argv):
    """
    Start the N ranks as a child `python -m torch.distributed.run` (this process has made no GPU call and makes none),
    hand rank 0's JSON line on and return the child's exit code (non-zero when any rank failed: the launcher tears the
    others down).
    """
    args = parse_args(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, universal_newlines=True)
    line = None
    for out in child.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)       # banners of libraries and the launcher: not on our stdout
    rc = child.wait()
    if rc != 0:
        sys.stderr.write('bench.py: the %d-rank child run failed with exit code %d\n' % (args.gpus, rc))
        return rc
    if line is None:
        sys.stderr.write('bench.py: the child run printed no result line\n')
        return 1
    print(line)
    sys.stdout.flush()
    return 0


def dry_run(args, world, rank):
    """The reporting path without a GPU (gloo): barrier, K sleeps as steps, max over ranks, one JSON line."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if os.environ.get('AMT_BENCH_DRYRUN_FAIL_RANK') == str(rank):
        raise SystemExit(3)
    elapsed = time.perf_counter() - t0
    records, ranks, backend = None, 1, None
    if world > 1:
        own = elapsed
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        records = [None] * world
        dist.all_gather_object(records, dict(rank=rank, host=socket.gethostname(), pid=os.getpid(), elapsed_ms=own * 1e3,
                                             frames=args.steps))
        ranks, backend = dist.get_world_size(), dist.get_backend()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'dry run (no GPU work)', 'value': world * args.steps / elapsed, 'unit': 'steps/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'dry_run': True,
                          'ranks': ranks, 'backend': backend, 'per_rank': records}))
        sys.stdout.flush()


# ------------------------------------------------------------------------------------------------------------
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


def measured_fill_gbs(device):
    """Write-only rate of this GPU (torch.fill_ of five arrays of the frame's size, the kernel's own 480 MB of output):
    what a kernel that only writes can reach here — the dominant kernel's traffic is 86 % writes.  Outside the timed region."""
    import torch
    bufs = [torch.empty((HEIGHT + 1) * (WIDTH + 1), dtype=torch.float64, device=device) for _ in range(5)]
    for b in bufs:
        b.fill_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for b in bufs:
            b.fill_(1.0)
    torch.cuda.synchronize()
    return 10 * sum(b.numel() for b in bufs) * 8 / (time.perf_counter() - t0) / 1e9


def device_description(device):
    """Name, PCI address and uuid of a torch device (the N-rank line lists one per rank)."""
    import torch
    props = torch.cuda.get_device_properties(device)
    out = {'index': device.index, 'name': props.name}
    dom, bus, dev = (getattr(props, k, None) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id'))
    if bus is not None:
        out['pci_bus_id'] = '%04x:%02x:%02x.0' % (dom or 0, bus, dev or 0)
    uuid = getattr(props, 'uuid', None)
    if uuid is not None:
        out['uuid'] = str(uuid)
    return out


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fp:
            for ln in fp:
                if ln.startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except IOError:
        pass
    return 'unknown'


def cpu_baseline(sample_rows, frames=4, parity=None):
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
        res = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']),
                              ALTITUDE, data, None, bbox, (PPD, PPD), disc, False)
        t2 = time.time()
        t_geo += t1 - t0
        t_res += t2 - t1
        if k == 0 and parity is not None:
            parity.update(check_against_oracle(hdr, cam, t, img, g, res))         # the oracle as the checker (untimed)
        del g, data, res
    npx = WIDTH * sample_rows * frames
    return dict(value=npx / 1e6 / (t_geo + t_res), unit='Mpixels/s', cores=1, kind='port', cpu=cpu_model(),
                host_cores=os.cpu_count(),
                sample='%d frames of the sequence, rows 0..%d of %dx%d each (%.1f Mpx): georef %.1f s + mask/resample '
                       '%.1f s, NumPy, 1 thread (the reference is single-threaded NumPy; one process per frame would '
                       'scale it by the core count at best)' % (frames, sample_rows, WIDTH, HEIGHT, npx / 1e6, t_geo, t_res))


def cpu_frame(k, sample_rows):
    """Frame k of the synthetic sequence (rows [0, sample_rows)) through the oracle -> (seconds georef, seconds mask + resample)."""
    from oracle import ref_numpy as O
    from auromat_amd.synthetic import sequence_frame, frame_image
    from auromat_amd.coordinates import transform as T
    hdr, cam, t, seed = sequence_frame(k, WIDTH, HEIGHT)
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
    O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']),
                    ALTITUDE, data, None, bbox, (PPD, PPD), disc, False)
    return t0, t1, time.time()


def cpu_worker(k, sample_rows):
    """`bench.py --cpu-worker k --cpu-rows R`: one process of the N-process CPU baseline; touches no GPU."""
    t0, t1, t2 = cpu_frame(k, sample_rows)
    print(json.dumps({'frame': k, 'start': t0, 'georef_done': t1, 'end': t2}))


def cpu_baseline_n_process(sample_rows, procs):
    """The reference's only data parallelism is one process per frame (a plain map over the frames, mapping/spacecraft.py:
    326-332): `procs` processes, each one frame (rows [0, sample_rows)) through the oracle, started together
    as children.  The span runs from the first worker's start of compute to the last one's end (interpreter start-up and
    the image generation are left out, as in the one-core figure)."""
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-rows', str(sample_rows), '--cpu-worker']
    children = [subprocess.Popen(cmd + [str(k)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, universal_newlines=True)
                for k in range(procs)]
    rows = []
    for c in children:
        o, _ = c.communicate()
        if c.returncode != 0:
            return {'error': 'a worker exited with code %d' % c.returncode}
        rows.append(json.loads(o.strip().splitlines()[-1]))
    span = max(r['end'] for r in rows) - min(r['start'] for r in rows)
    per_proc = sum(r['end'] - r['start'] for r in rows) / len(rows)
    npx = WIDTH * sample_rows * procs
    return dict(value=npx / 1e6 / span, unit='Mpixels/s', cores=procs, processes=procs, seconds=span,
                mean_seconds_per_process=per_proc,
                sample='%d processes, one frame of the sequence each (rows 0..%d of %dx%d: %.1f Mpx in all), NumPy oracle, '
                       '1 thread per process; span from the first start of compute to the last end' % (
                           procs, sample_rows, WIDTH, HEIGHT, npx / 1e6))


def sky_rows_note():
    import ctypes as C
    from auromat_amd import _native
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.synthetic import sequence_frame
    hdr, cam, t, _ = sequence_frame(0, WIDTH, HEIGHT)
    p = frame_params(hdr, ALTITUDE, cam, t, True)
    o = [C.c_int32(0) for _ in range(4)]
    _native.lib().amt_georef_sky_rows(C.byref(p), *[C.byref(v) for v in o])
    rows, n, top, bottom = [v.value for v in o]
    return '%d of %d rows of work items (%d pixel rows each) of frame 0 are NaN fills (amt_georef_sky_rows)' % (top + n - bottom, n, rows)


def check_against_oracle(hdr, cam, t, img, g, res):
    """BASELINE.json's second figure, max |dlat, dlon| vs ref: the HIP path (single-pass plan) on the frame the CPU
    baseline has just computed with the oracle — coordinate arrays, NaN patterns and the resampled grid."""
    from auromat_amd.pipeline import FramePipeline
    h = int(hdr['IMAGEH'])
    pipe = FramePipeline(WIDTH, h)
    got = pipe.run(hdr, ALTITUDE, cam, t, img=img, fast=True, min_elevation=MIN_ELEV, pxPerDeg=PPD, fuse=True)
    arrays = pipe.host_arrays()
    out = {'frame': [WIDTH, h], 'plan': pipe.last_plan, 'checker': 'oracle/ref_numpy.py (NumPy restatement of the reference)'}
    worst = 0.0
    same_nan = True
    for name in ('lat', 'lon', 'lat_c', 'lon_c'):
        a, b = arrays[name], g[name]
        same_nan = same_nan and bool(np.array_equal(np.isnan(a), np.isnan(b)))
        d = np.abs(a - b)
        if name.startswith('lon'):
            d = np.minimum(d, 360.0 - d)
        worst = max(worst, float(np.nanmax(d)))
    out['max_abs_dlat_dlon_deg'] = worst
    out['max_abs_delevation_deg'] = float(np.nanmax(np.abs(arrays['elev'] - g['elev'])))
    out['same_nan_pattern'] = same_nan
    want = res['data']
    mask = np.isnan(want[..., 0])
    same_shape = got['mean'].shape == want.shape
    out['grid'] = list(want.shape)
    out['grid_mask_cells_differing'] = int((got['mask'] != mask).sum()) if same_shape else -1
    ok = ~mask & ~got['mask'] if same_shape else None
    out['grid_rgb_cells_differing'] = int((np.abs(got['mean'][..., :3][ok] - want[..., :3][ok]).max(axis=-1) > 0).sum()) if same_shape else -1
    out['tolerance_deg'] = 1e-6
    out['ok'] = bool(same_nan and worst <= 1e-6 and out['max_abs_delevation_deg'] <= 1e-6 and same_shape and
                     out['grid_mask_cells_differing'] == 0 and out['grid_rgb_cells_differing'] == 0)
    del pipe
    return out


def resident_images(device, n, first_seed):
    """n distinct uint16 RGB images (bits as int16, the frame buffers' layout) generated in HBM, seeded per frame."""
    import torch
    out = []
    for k in range(n):
        g = torch.Generator(device=device)
        g.manual_seed(1000003 * (first_seed + k) + 17)
        out.append(torch.randint(0, 65535, (HEIGHT, WIDTH, 3), generator=g, device=device, dtype=torch.int32)
                   .to(torch.int16))
    return out


def timed_run(frames, warmup, steps, fast, plan, magnetic, batch, streams, use_hints, shared_image, own_buffers,
              fence, after=None, keep_coordinates=True, spinup_ms=0.0, geodetic_arrays=None, repeats=None):
    """
    W untimed frames, then the timed region of K frames `repeats` times, through a fresh SequencePipeline.  Returns
    dict(regions, seq, results, extra, ...): `regions` holds per region dict(elapsed, georef_ms, bin_ms, plans, hinted, ...) —
    georef_ms / bin_ms are per FRAME, from HIP events on the dispatch packets —, results / extra are the last region's.
    `after(results)` runs inside the timed region (the gather of the N > 1 runs).
    """
    repeats = REPEATS if repeats is None else repeats
    from auromat_amd.pipeline import NativeResults, SequencePipeline
    seq = SequencePipeline(WIDTH, HEIGHT, altitude=ALTITUDE, fast=fast, min_elevation=MIN_ELEV, pxPerDeg=PPD,
                           plan='single-pass' if plan == 'fused' else 'two-pass', bin_stream=streams == 2,
                           shared_image=shared_image, magnetic=magnetic, batch=batch, own_image_buffers=own_buffers,
                           keep_coordinates=keep_coordinates, launch_streams=LAUNCH_STREAMS, geodetic_arrays=geodetic_arrays)
    seq.use_hints = use_hints
    ctx = seq.ctx
    spun = 0
    if after is not None and warmup + steps > 0:
        # the first gather (RCCL channels, its buffers at full size) comes BEFORE the spin-up: its host synchronisations
        # leave the GPU idle, and a short timed region right behind it runs its first launches 10 % slower
        first = seq.process(frames[:max(warmup, 1)])
        after(first, True, None)
        del first
    if spinup_ms > 0 and warmup + steps > 0:
        import torch
        t_end = time.perf_counter() + spinup_ms * 1e-3
        while time.perf_counter() < t_end:
            seq.process(frames[:max(warmup, 2 * batch)])
            torch.cuda.synchronize()
            spun += max(warmup, 2 * batch)
    warm = seq.process(frames[:warmup])
    del warm
    regions = []
    results = extra = None
    n_timed = (steps + TIMING_EVERY - 1) // TIMING_EVERY
    for _ in range(repeats):
        results = extra = None                  # (the previous region's arenas go back to the allocator)
        if regions and warmup > 0:
            # every region starts from the state the first one starts from — behind the W warm-up frames, the timed frames'
            # predecessors in the sequence — instead of behind its own last frame (whose successor the first timed frame is
            # not: six frames of such a region run the coarse pre-pass, profiles/r6/b2_region_timeline_driver_command.txt)
            warm = seq.process(frames[:warmup])
            del warm
        ctx.timing_enable(TIMING_EVERY)
        fence()
        t0 = time.perf_counter()
        error = None
        try:
            timed = frames[warmup:warmup + steps]
            if os.environ.get('AMT_BENCH_FAIL_RANK') == os.environ.get('RANK', '0') and len(regions) == 1:
                # (tests: this rank's pipeline fails in the middle of its second timed region — a frame dated beyond the IGRF
                # table, which the library's frame loop refuses — while the other ranks go on to the gather)
                from datetime import datetime
                bad = list(timed[len(timed) // 2])
                bad[2] = datetime(2031, 1, 1)
                timed = timed[:len(timed) // 2] + [tuple(bad)] + timed[len(timed) // 2 + 1:]
            results = seq.process(timed)
        except Exception as e:              # (a HIP error, no memory, a date outside the IGRF table ...)
            if after is None:
                raise
            error, results = e, None
        t_proc = time.perf_counter()
        plans, hinted = list(seq.plans), seq.hinted
        # a rank that failed still takes part in the gather (it sends a buffer that says so) and in the closing fence, where
        # every rank learns of it and raises: the job ends with a non-zero code, nobody waits in a collective
        extra = after(results, False, error) if after is not None else None
        t_after = time.perf_counter()
        fence(error)
        elapsed = time.perf_counter() - t0
        if os.environ.get('AMT_BENCH_DEBUG'):
            sys.stderr.write('timed region: process %.3f ms, after %.3f ms, fence %.3f ms\n' % (
                (t_proc - t0) * 1e3, (t_after - t_proc) * 1e3, (t0 + elapsed - t_after) * 1e3))
        g_total, g_n = ctx.timing_read(0)
        b_total, b_n = ctx.timing_read(1)
        ctx.timing_enable(False)
        assert g_n == n_timed and b_n in (0, n_timed), (g_n, b_n)
        regions.append(dict(elapsed=elapsed, georef_ms=g_total / g_n, bin_ms=(b_total / b_n if b_n else 0.0), plans=plans,
                            hinted=hinted, uploaded_bytes=seq.uploaded_bytes, process_ms=(t_proc - t0) * 1e3,
                            after_ms=(t_after - t_proc) * 1e3, fence_ms=(t0 + elapsed - t_after) * 1e3))
    return dict(regions=regions, seq=seq, results=results, extra=extra, spinup_frames=spun,
                native_loop=isinstance(results, NativeResults), variant=ctx.last_variant())


def pick_region(run, use_dist=False, cdev=None):
    """The median region of a timed_run: with N ranks the region's time is the MAX over the ranks, per region, and every rank
    picks the same one -> (index, elapsed of that region as the job sees it, the rank's own record of it)."""
    own = [r['elapsed'] for r in run['regions']]
    seen = own
    if use_dist:
        import torch
        import torch.distributed as dist
        tmax = torch.tensor(own, dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        seen = [float(v) for v in tmax.tolist()]
    m = median_index(seen)
    return m, seen, run['regions'][m]


def directions_in_run(imgs, first, warmup, steps, fence, n_dirs=6, batch=3, repeats=None):
    """
    The pipeline in the form north_star words it — "coalesced HBM reads of the (H+1) x (W+1) corner arrays": every frame's
    corner directions ((H+1, W+1, 3) float64, J2000; reference astrometry.py:49-64 cameraToPixelCornerDirection) are resident
    in HBM like its image, and ONE kernel (k_georef_rows<DIRS_IN, BIN>) reads them, intersects the shell, writes the five
    coordinate arrays and bins — `batch` frames per launch like the headline (amt_pipe_launch_dirs_many), two batches in
    flight.  `n_dirs` distinct frames (direction arrays are 288 MB each) are cycled, each time with another image; a frame's
    estimate is its own exact box from the cycle before (the first cycle, inside the warm-up, runs the coarse pre-pass on the
    direction array).  The timed region runs `repeats` times; the median region is reported.
    """
    import torch
    from auromat_amd.coordinates.wcs import pix2world
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import sequence_frame
    repeats = REPEATS if repeats is None else repeats
    pipes = [FramePipeline(WIDTH, HEIGHT, alloc_image=False, padded=PADDED) for _ in range(2 * batch)]
    ctx = pipes[0].ctx
    for q in pipes:
        q.defer_join = True
    frames = []
    for k in range(n_dirs):
        hdr, cam, t, _ = sequence_frame(first + k, WIDTH, HEIGHT)
        frames.append((frame_params(hdr, ALTITUDE, cam, t, True, magnetic=False), cam, t,
                       pix2world(hdr, WIDTH, HEIGHT, corner=True, ascartesian=True, device=ctx.device)))
    hints = [None] * n_dirs
    plans = []

    def launch(k0, n):
        qs = [pipes[(k0 + i) % len(pipes)] for i in range(n)]
        ps, ds = [], []
        for i, q in enumerate(qs):
            p, cam, t, dirs = frames[(k0 + i) % n_dirs]
            q.use_image(imgs[(k0 + i) % len(imgs)])
            q.start_coarse(p, MIN_ELEV, False, hint=hints[(k0 + i) % n_dirs], dirs=dirs)
            ps.append(p)
            ds.append(dirs)
        FramePipeline.georef_many(qs, ps, ALTITUDE, MIN_ELEV, (PPD, PPD), False, dirs=ds, pole_in_view=0)

    def finish(k0, n):
        qs = [pipes[(k0 + i) % len(pipes)] for i in range(n)]
        ready = [q.fused_ready((PPD, PPD), False) for q in qs]
        if all(r is not None for r in ready):
            out = FramePipeline.finalize_many(qs, ready, (PPD, PPD), True)
        else:
            out = [q.resample((PPD, PPD), keep_on_device=True) for q in qs]
        for i, q in enumerate(qs):
            plans.append(q.last_plan)
            if q.last_plan == 'single-pass':
                hints[(k0 + i) % n_dirs] = list(q._fused['result'].bbox)
        return out

    def run(n_frames):
        out, k, prev = [], 0, None
        while k < n_frames:
            n = min(batch, n_frames - k)
            launch(k, n)
            if prev is not None:
                out.extend(finish(*prev))
            prev = (k, n)
            k += n
        if prev is not None:
            out.extend(finish(*prev))
        for q in pipes:
            q.join()
        return out

    run(max(warmup, 2 * n_dirs))                 # every frame of the cycle has its hint now
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        run(n_dirs)
        torch.cuda.synchronize()
    regions = []
    results = None
    for _ in range(repeats):
        results = None
        del plans[:]
        ctx.timing_enable(1)
        fence()
        t0 = time.perf_counter()
        results = run(steps)
        fence()
        elapsed = time.perf_counter() - t0
        g_total, g_n = ctx.timing_read(0)
        ctx.timing_enable(False)
        assert g_n == steps and len(results) == steps, (g_n, len(results))
        regions.append(dict(elapsed=elapsed, georef_ms=g_total / g_n, plans=list(plans)))
    variant = ctx.last_variant()
    grid = list(results[-1]['mean'].shape)
    del results, frames
    for q in pipes:
        del q
    m = median_index([r['elapsed'] for r in regions])
    return dict(elapsed=regions[m]['elapsed'], georef_ms=regions[m]['georef_ms'], plans=regions[m]['plans'], variant=variant,
                grid=grid, repeats=repeats, georef_ms_min=min(r['georef_ms'] for r in regions))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    global LAUNCH_STREAMS
    LAUNCH_STREAMS = args.launch_streams
    if args.cpu_worker is not None:
        return cpu_worker(args.cpu_worker, min(args.cpu_rows, HEIGHT))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_children(argv))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    if rank != 0:
        # only rank 0 reports; libraries that write to stdout (RCCL prints a version banner with C stdio) must not
        # interleave with its JSON line in the launcher's merged output
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist
    # Rehearsal of the N-rank run on a box with ONE GPU (tests/test_gpu_bench_ranks.py): AMT_BENCH_ONE_GPU=1 puts every rank on
    # cuda:0 and AMT_BENCH_BACKEND=gloo takes the collectives through the host (RCCL refuses two ranks on one device) — the
    # sharding, the capacity agreement, the gather, the barriers and the max over the ranks are the code of the real run
    backend = os.environ.get('AMT_BENCH_BACKEND', 'nccl')
    if os.environ.get('AMT_BENCH_ONE_GPU'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # AMT_BENCH_FORCE_DIST=1 exercises the RCCL gather path with a single rank (boxes with one GPU)
    use_dist = world > 1 or bool(os.environ.get('AMT_BENCH_FORCE_DIST'))
    def init_dist():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        # (a rank that dies without a word — killed, a fault — leaves the others in a collective: they give up after this long
        # and the launcher, which tears the job down at the first rank that exits with an error, ends it sooner)
        from datetime import timedelta
        limit = timedelta(seconds=int(os.environ.get('AMT_DIST_TIMEOUT_S', '600')))
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank), timeout=limit)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)

    if use_dist:
        init_dist()

    from auromat_amd._native import Context, host_threads_report
    from auromat_amd.sequence import agree_capacity, agree_ok, collective_device, gather_device
    from auromat_amd.synthetic import sequence_frame, frame_image

    ctx = Context.current()
    device = ctx.device
    cdev = collective_device(device) if use_dist else device      # where the collectives' tensors live (the GPU for RCCL)
    fast = not args.exact
    total = args.warmup + args.steps
    first = rank * total                    # this rank's block of the synthetic sequence
    # the synthetic sequence (what a reader would hand over: WCS cards, camera position, time, image) exists before
    # the timed region; everything derived from it (matrices, grids) is computed inside
    shared = frame_image(WIDTH, HEIGHT, seed=rank) if args.shared_image else None
    if args.upload:
        # four distinct images in pinned host memory (uint16 bits as int16, the frame buffer's layout), cycled
        imgs = [torch.from_numpy(frame_image(WIDTH, HEIGHT, seed=100 + i).view(np.int16)).pin_memory() for i in range(4)]
    elif shared is None:
        imgs = resident_images(device, min(total, MAX_RESIDENT_IMAGES), first)
    else:
        imgs = [None]

    def make_frames(n, magnetic):
        fr = []
        for k in range(n):
            hdr, cam, t, _ = sequence_frame(first + k, WIDTH, HEIGHT)
            fr.append((hdr, cam, t, imgs[k % len(imgs)], SHELLS[k % 3] if magnetic else None))
        return fr

    def fence(error=None):
        """synchronize + barrier + synchronize.  With a process group the barrier is an all_reduce(MAX) of one flag — what a
        barrier is made of anyway — that carries this rank's failure: when any rank failed, every rank raises here."""
        torch.cuda.synchronize()
        if use_dist and dist.is_initialized():
            flag = torch.tensor([0 if error is None else 1], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()
            if error is not None:
                raise error
            if int(flag.item()):
                from auromat_amd.sequence import SequenceError
                raise SequenceError(['?'], ['another rank failed inside the timed region (its own message is on its stderr)'])
        elif error is not None:
            raise error
        torch.cuda.synchronize()

    gather_state = {}

    def gather(results, warm, error=None):
        # device-to-device over xGMI; rank 0 unpacks to the host after the timed region.  The first call's costs
        # (RCCL channels, allocations) belong to the warm-up.
        if not use_dist:
            return None
        if error is not None:
            from auromat_amd.sequence import _FailedRank
            return gather_device(_FailedRank(), [], cdev, capacity=gather_state.get('capacity'))
        base = first + (0 if warm else args.warmup)
        if warm:
            # as many frames as the timed gather will carry, so that its buffers (payload, padded send and receive
            # buffers) come out of the caching allocator instead of hipMalloc inside the timed region; and the ranks
            # agree on a capacity (the longest payload + 25 %) here, so that the timed gather is ONE collective
            # without a size exchange and its host synchronisation
            results = (results * (args.steps // max(len(results), 1) + 1))[:args.steps]
            gather_state['capacity'] = agree_capacity(results, [base + k for k in range(len(results))], cdev)
        return gather_device(results, [base + k for k in range(len(results))], cdev, capacity=gather_state.get('capacity'))

    run = timed_run(make_frames(total, args.magnetic), args.warmup, args.steps, fast, args.plan, args.magnetic,
                    args.batch, args.streams, not args.no_hints, shared, own_buffers=args.upload, fence=fence,
                    after=gather, spinup_ms=args.spinup_ms, geodetic_arrays=args.nine_arrays)
    # the median of the REPEATS timed regions (each one: barrier + synchronize, K steps, gather, barrier + synchronize; with N
    # ranks a region's time is the maximum over the ranks)
    m_region, region_times, reg = pick_region(run, use_dist, cdev)
    elapsed = region_times[m_region]
    rank_records = None
    if use_dist:
        own_elapsed = reg['elapsed']
        # what every rank saw, so that the line itself shows that N ranks on N distinct devices took part (after the timed
        # region; all_gather_object is a collective of its own on every backend)
        whole = run['results'].payload() if hasattr(run['results'], 'payload') else None
        cap = gather_state.get('capacity')
        from auromat_amd.sequence import DESC_LEN
        mine = dict(rank=rank, local_rank=int(os.environ.get('LOCAL_RANK', '0')), host=socket.gethostname(), pid=os.getpid(),
                    device=device_description(device), frames=args.steps, first_frame=first + args.warmup,
                    elapsed_ms=own_elapsed * 1e3, process_ms=reg['process_ms'], gather_ms=reg['after_ms'],
                    closing_fence_ms=reg['fence_ms'], kernel_us_per_frame=reg['georef_ms'] * 1e3,
                    regions_elapsed_ms=[r['elapsed'] * 1e3 for r in run['regions']],
                    single_pass_frames=sum(1 for q in reg['plans'] if q == 'single-pass'),
                    payload_bytes=None if whole is None else int(whole[1]) * 8,
                    gather_bytes=None if cap is None else (cap[0] * DESC_LEN + cap[1] + 2) * 8)
        mine['host_threads'] = host_threads_report()
        # a rank whose grids did not fit the capacity agreed in the warm-up said so in its buffer: the destination reads that
        # now, and every rank ends with it (agree_ok raises SequenceError on all of them)
        problem = None
        if rank == 0:
            try:
                run['extra'].sizes
            except ValueError as e:
                problem = e
        agree_ok(problem, cdev)
        rank_records = [None] * world
        dist.all_gather_object(rank_records, mine)
        if rank == 0:
            gathered = run['extra']
            assert gathered.n_frames == world * args.steps and len(gathered.unpack()) == world * args.steps

    upload = None
    host_imgs = None
    want_upload = not args.no_variants and not args.upload and shared is None and args.plan == 'fused' and not args.exact
    if want_upload:
        # The PCIe-inclusive form of the same job (never the headline value): every rank streams its frames'
        # images from its OWN pinned host buffers (four distinct 72 MB images, cycled), uploaded on a copy stream beside the
        # previous frames' kernels, instead of finding them resident in HBM — what a real 256-frame run does, and what
        # eight ranks do to the host's memory system at once.  Same loop, gather included, barrier + max over the ranks.
        u_steps = min(args.steps, 96)
        try:
            host_imgs = [torch.from_numpy(frame_image(WIDTH, HEIGHT, seed=100 + 4 * rank + i).view(np.int16)).pin_memory() for i in range(4)]
        except RuntimeError:
            host_imgs = None                  # (page-locking 288 MB failed on this rank)
        if use_dist:
            # every rank or none: a rank that could not page-lock its images must not leave the others waiting in the gather
            ok_all = torch.tensor([0 if host_imgs is None else 1], dtype=torch.int32, device=cdev)
            dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
            if int(ok_all.item()) == 0:
                host_imgs = None
    if want_upload and host_imgs is not None:
        u_frames = []
        for k in range(args.warmup + u_steps):
            hdr, cam, t, _ = sequence_frame(first + k, WIDTH, HEIGHT)
            u_frames.append((hdr, cam, t, host_imgs[k % 4], SHELLS[k % 3] if args.magnetic else None))
        u_first = first
        saved_steps = args.steps
        args.steps = u_steps                  # (the gather's warm-up sizes its buffers from args.steps)
        u = timed_run(u_frames, args.warmup, u_steps, fast, 'fused', args.magnetic, args.batch, args.streams, not args.no_hints,
                      None, own_buffers=True, fence=fence, after=gather, spinup_ms=min(args.spinup_ms, 100.0),
                      geodetic_arrays=args.nine_arrays)
        args.steps = saved_steps
        u_m, u_times, u_reg = pick_region(u, use_dist, cdev)
        u_own = u_reg['elapsed']
        u_elapsed = u_times[u_m]
        # what crossed the link, as the loop that sent it counted it: only the rows of each image that can be binned (inside the
        # limb and above min_elevation: amt_georef_image_rows / amt_run_result.uploaded_bytes), not the 72 MB
        u_bytes = u_reg['uploaded_bytes']
        u_rates = [u_bytes / u_own / 1e9]
        if use_dist:
            rates = [None] * world
            dist.all_gather_object(rates, u_rates[0])
            u_rates = rates
        upload = {'Mpixels_per_s': world * u_steps * WIDTH * HEIGHT / 1e6 / u_elapsed, 'ms_per_frame': u_elapsed / u_steps * 1e3,
                  'frames_per_rank': u_steps, 'kernel_ms_per_frame': u_reg['georef_ms'],
                  'repeats': len(u_times), 'ms_per_frame_min': min(u_times) / u_steps * 1e3, 'ms_per_frame_max': max(u_times) / u_steps * 1e3,
                  'single_pass_frames': sum(1 for q in u_reg['plans'] if q == 'single-pass'),
                  'image_bytes_per_frame': WIDTH * HEIGHT * 6, 'uploaded_bytes_per_frame': u_bytes / u_steps,
                  'pcie_GBs_per_rank': u_rates, 'pcie_GBs_total': float(sum(u_rates)),
                  'frame_loop': 'library (amt_run_push with amt_run_frame.img_host)' if u['native_loop'] else 'python',
                  'source': 'four distinct uint16 RGB images per rank in page-locked host memory, cycled; one upload per frame on the '
                            "runner's copy stream, one batch ahead of the frame's launch — the rows of the image that can be binned "
                            '(a ray hits the shell at an elevation >= min_elevation), the only ones a kernel needs —, gather of the '
                            'grids included'}
        del u, u_frames, host_imgs
        import gc
        gc.collect()

    if rank == 0:
        fused = args.plan == 'fused'
        seq, results, plans = run['seq'], run['results'], reg['plans']
        georef_ms, bin_ms = reg['georef_ms'], reg['bin_ms']
        npx = WIDTH * HEIGHT
        copy_gbs = measured_copy_gbs(device)
        fill_gbs = measured_fill_gbs(device)
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
            # BASELINE.md's configs[3] row: mlat, mlt (corners), mlat_c, mlt_c, elev written = 16 Nc + 24 Np, the byte counts of
            # the geodetic row; --nine-arrays writes lat, lon, lat_c, lon_c beside them (more than the contract counts)
            tkey += '_mag' if args.nine_arrays else '_mag_only'
            if args.nine_arrays:
                moved += ab['mag']
        achieved = kbytes / (georef_ms * 1e-3) / 1e9
        fpl = seq.batch if fused else 1
        tr = traffic.get(tkey, {})
        sha = kernel_sources_sha16()
        tr_fresh = bool(tr) and tr.get('sources_sha16') == sha
        out = {
            'metric': 'Mpixels/s georef+resample, 4240x2832 frame',
            'value': world * args.steps * npx / 1e6 / elapsed,
            'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            # value / ms_per_step are those of the MEDIAN of `repeats` timed regions of `steps` steps each, run back to back
            # (ms_per_step x steps = that one region); the spread of the regions beside it.  Every region runs behind the same
            # `warmup` untimed frames (the timed frames' predecessors in the sequence), like the first one
            'repeats': len(region_times), 'untimed_frames_before_every_region': args.warmup, 'ms_per_step_min': min(region_times) / args.steps * 1e3,
            'ms_per_step_max': max(region_times) / args.steps * 1e3,
            'regions_ms': [t * 1e3 for t in region_times],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic; ' + ('image of every frame uploaded from pinned host memory' if args.upload else
                                     'one image shared by all frames, resident in HBM' if shared is not None else
                                     'a distinct image per frame (%d resident in HBM before the timed region)' % len(imgs)),
            'config': {'workload': ('configs[3]: shells 100/110/120 km, MLat/MLT of corners and centres + elevation%s, '
                                    'mean-resample on the 0.1 deg (MLat, SM longitude) grid (resampleMLatMLT)'
                                    % (' + lat/lon of corners and centres (nine arrays)' if args.nine_arrays else '')
                                    if args.magnetic else
                                    'configs[2]: 4240x2832 ISS-like frame, WCS ray cast + WGS84(+110 km) '
                                    'intersection + geodetic transform + elevation (%s centres), '
                                    'maskedByElevation(10), mean-resample to 0.1 deg plate-carree, uint16 RGB'
                                    % ('fast' if fast else 'exact')) +
                                   ('; configs[4] when n_gpus = 8 and steps = 32: 256 frames sharded over 8 GPUs, grids '
                                    'gathered on rank 0' if world > 1 else ''),
                       'frame': [WIDTH, HEIGHT], 'px_per_deg': PPD, 'grid': list(res['mean'].shape),
                       'plan': args.plan, 'row_layout': 'strip-padded' if seq.padded else 'contiguous', 'frames_per_launch': seq.batch, 'frames_without_prepass': reg['hinted'],
                       'single_pass_frames': sum(1 for q in plans if q == 'single-pass'),
                       'frames_total': world * args.steps,
                       # untimed frames run before the W warm-up steps to bring the chip to its sustained state
                       'spinup_frames_untimed': run['spinup_frames'], 'spinup_ms': args.spinup_ms,
                       'parallelism': 'frames sharded over %d GPU(s), %s gather of grids' % (world, 'RCCL' if backend == 'nccl' else backend + ' (rehearsal)'),
                       'frame_loop': 'library (amt_run_*)' if run['native_loop'] else 'python',
                       # transparency: rows of work items whose waves write NaN without casting rays, because the host has
                       # bounded the limb (a conic section in the image) and no ray of them can hit the shell; every
                       # output array is still written in full and is identical to the ray-cast result
                       'sky_item_rows': sky_rows_note(),
                       'kernel_variant': dict(zip(('second', 'bin', 'frames_in_last_launch'), run['variant'])),
                       'host_threads': host_threads_report(), 'device': info['name']},
            # dominant kernel.  What bounds it is the rate at which the memory system takes its stores (every
            # variant moves its bytes at about the same rate whatever it computes; the strip-padded rows of round 6 raised that
            # rate by a tenth); `frac` prices SURVEY 8d's contract bytes, `frac_bytes_moved_min` the bytes the kernel has to move.
            'roofline': {'bound': 'hbm', 'kernel': kname, 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'measured_copy_GBs': copy_gbs, 'measured_fill_GBs': fill_gbs, 'fp64_vector_peak_TFLOPs': 78.6,
                         # one launch covers `frames_per_launch` frames: bytes, traffic and duration are per launch
                         'frames_per_launch': fpl,
                         # NOT measured in this run: PMC counters of an earlier rocprofv3 pass of the same kernel
                         # (null when that pass was made on other kernel sources than the ones this run was built from)
                         'traffic': ((tr.get('hbm_bytes') or 0) * fpl or None) if tr_fresh else None,
                         'traffic_source': ('profiles/traffic.json (%s), per frame x frames_per_launch'
                                            % tr.get('source', 'rocprofv3 --pmc pass, see its _comment')) if tr_fresh else
                                           ('none: the PMC pass recorded in profiles/traffic.json[%s] was made on kernel sources %s, '
                                            'this run on %s' % (tkey, tr.get('sources_sha16', '(unrecorded)'), sha)),
                         'kernel_sources_sha16': sha,
                         'algorithmic_bytes': kbytes * fpl, 'ms_per_launch': georef_ms * fpl,
                         # the bytes this kernel has to move at the very least, and the same fraction on that basis
                         'bytes_moved_min': moved * fpl, 'frac_bytes_moved_min': frac(moved, georef_ms),
                         'frames_timed': args.steps, 'valu_busy': tr.get('valu_busy') if tr_fresh else None,
                         'valu_busy_source': 'profiles/traffic.json'},
            'kernels': {
                'k_georef_rows': {'ms': georef_ms, 'algorithmic_bytes': kbytes, 'frac_hbm_peak': frac(kbytes, georef_ms),
                                  'bytes_moved_min': moved},
                'k_bin_frame': ({'ms': bin_ms, 'algorithmic_bytes': ab['resample'],
                                 'frac_hbm_peak': frac(ab['resample'], bin_ms),
                                 'traffic': traffic.get('k_bin_frame', {}).get('hbm_bytes')
                                 if traffic.get('k_bin_frame', {}).get('sources_sha16') == sha else None} if bin_ms else
                                'not launched: binning is fused into k_georef_rows (plan=fused)'),
                # SURVEY.md 8d contract figures for the whole pipeline against the time of the kernel(s) that do it
                'pipeline_frac_wcs_fused_840.6MB': frac(ab['georef'] + ab['resample'], georef_ms + bin_ms),
                'pipeline_frac_directions_in_1129.0MB': frac(ab['georef_dirs_in'] + ab['resample'], georef_ms + bin_ms),
            },
        }
        if use_dist:
            # the N-rank line describes itself: who took part, on which device, and what each rank measured
            out['ranks'] = dist.get_world_size()
            out['backend'] = dist.get_backend()
            out['per_rank'] = rank_records
            out['distinct_devices'] = len(set((r['host'], r['device'].get('pci_bus_id') or r['device'].get('uuid') or r['device']['index'])
                                              for r in rank_records))
            out['gather_bytes_received'] = sum(r['gather_bytes'] or 0 for r in rank_records)
            out['timed_region_ms'] = {'max_over_ranks': elapsed * 1e3, 'rank0_process': reg['process_ms'],
                                      'rank0_gather': reg['after_ms'], 'rank0_closing_fence': reg['fence_ms']}
        del run, seq, results
        import gc
        gc.collect()        # the pipeline's drivers are freed HERE (hipFree synchronises the device), not whenever the
                            # cycle collector gets to them in the middle of a later timed region
        if world == 1 and not args.no_variants and shared is None and not (args.exact or args.magnetic or args.upload or not fused):
            # the other configurations of SURVEY 8d on the record of the same run: short runs of the same loop
            torch.cuda.empty_cache()
            variants = {}
            nv_w, nv_k = 9, 96
            from auromat_amd.synthetic import pole_frame
            p_hdr, p_cam, p_t = pole_frame(WIDTH, HEIGHT)
            # (configs[3] first: its nine arrays per buffer are the largest allocations of the run, and it measured up to
            # 8 % slower per kernel when it came after the other variants' allocate / free cycles)
            for name, kw in (('configs3_magnetic_3_shells', dict(fast=True, magnetic=True)),
                             # the same with lat, lon, lat_c, lon_c written beside MLat / MLT (nine arrays, both chains)
                             ('configs3_nine_arrays', dict(fast=True, magnetic=True, nine=True)),
                             # the headline workload over 192 frames (the driver's 20-step run times 7 launches)
                             ('configs2_long', dict(fast=True, magnetic=False, frames=192)),
                             ('exact_centres', dict(fast=False, magnetic=False)),
                             # NOT the headline workload: the resampled grids only, no per-pixel coordinate arrays
                             # written (what a convert run needs; SequencePipeline(keep_coordinates=False))
                             ('grids_only_no_coordinate_arrays', dict(fast=True, magnetic=False, keep=False)),
                             # a camera that looks across the geographic pole (the same frame, its own image each time):
                             # the pole plan of the fused kernel, binned in rotated coordinates
                             ('pole_in_view', dict(fast=True, magnetic=False, pole=True))):
                nv_k = kw.get('frames', 96)
                if kw.get('pole'):
                    vframes = [(p_hdr, p_cam, p_t, imgs[k % len(imgs)], None) for k in range(nv_w + nv_k)]
                else:
                    vframes = make_frames(nv_w + nv_k, kw['magnetic'])
                v = timed_run(vframes, nv_w, nv_k, kw['fast'], 'fused', kw['magnetic'],
                              args.batch, args.streams, True, None, own_buffers=False, fence=fence,
                              keep_coordinates=kw.get('keep', True), spinup_ms=args.spinup_ms, geodetic_arrays=kw.get('nine', False))
                vb = ab['mag_shell'] - 24 * (WIDTH + 1) * (HEIGHT + 1) if kw['magnetic'] else ab['georef'] + ab['resample']
                if not kw.get('keep', True):
                    vb = ab['image']            # all it has to move: the image
                v_m, v_times, v_reg = pick_region(v)
                variants[name] = {
                    'ms_per_frame': v_times[v_m] / nv_k * 1e3, 'Mpixels_per_s': nv_k * npx / 1e6 / v_times[v_m],
                    'frames': nv_k, 'repeats': len(v_times), 'kernel_ms_per_frame': v_reg['georef_ms'],
                    'kernel_ms_per_frame_min': min(r['georef_ms'] for r in v['regions']),
                    'single_pass_frames': sum(1 for q in v_reg['plans'] if q == 'single-pass'),
                    'algorithmic_bytes_per_frame': vb, 'frac': frac(vb, v_reg['georef_ms']),
                    'kernel_variant_second': v['variant'][0],
                }
                if kw['magnetic']:
                    # SURVEY 8d config 4 counts 40 Nc + 54 Np per shell (directions read); the kernel generates them
                    variants[name]['frac_contract_3387MB_for_3_shells'] = frac(ab['mag_shell'], v_reg['georef_ms'])
                del v
                gc.collect()
                torch.cuda.empty_cache()
            # BASELINE configs[1]: the intersection + geodetic transform on their own (no image, no resample): the
            # georeferencing kernel without the fused binning, per-frame host set-up inside the timed region as above
            from auromat_amd.pipeline import FramePipeline
            nv_k = 96
            gframes = make_frames(nv_w + nv_k, False)
            for gname, gpad in (('configs1_georef_only', PADDED), ('configs1_georef_only_contiguous_rows', False)):
                if gname.endswith('contiguous_rows') and not PADDED:
                    continue
                # (the second entry: the same kernel writing the reference's contiguous arrays directly — what amt_georef_frame
                # does on a caller's arrays — beside the pipeline's own strip-padded buffers)
                gpipe = FramePipeline(WIDTH, HEIGHT, alloc_image=False, padded=gpad)
                for hdr, cam, t, _, _ in gframes[:nv_w]:
                    gpipe.georef(hdr, ALTITUDE, cam, t, True, MIN_ELEV)
                g_regions = []
                for _ in range(REPEATS):
                    gpipe.ctx.timing_enable(1)
                    fence()
                    t0 = time.perf_counter()
                    for hdr, cam, t, _, _ in gframes[nv_w:]:
                        gpipe.georef(hdr, ALTITUDE, cam, t, True, MIN_ELEV)
                    fence()
                    g_el = time.perf_counter() - t0
                    g_total, g_n = gpipe.ctx.timing_read(0)
                    gpipe.ctx.timing_enable(False)
                    assert g_n == nv_k
                    g_regions.append((g_el, g_total / g_n))
                g_el, g_ms = g_regions[median_index([r[0] for r in g_regions])]
                variants[gname] = {
                    'ms_per_frame': g_el / nv_k * 1e3, 'Mpixels_per_s': nv_k * npx / 1e6 / g_el, 'frames': nv_k, 'repeats': REPEATS,
                    'kernel_ms_per_frame': g_ms, 'kernel_ms_per_frame_min': min(r[1] for r in g_regions),
                    'algorithmic_bytes_per_frame': ab['georef'], 'frac': frac(ab['georef'], g_ms),
                    'row_layout': 'strip-padded' if gpad else 'contiguous'}
                del gpipe
                gc.collect()
                torch.cuda.empty_cache()
            # the directions-in form (SURVEY 8d contract row "directions-in" + "resample-mean" = 1129.0 MB per frame: 24 B per
            # corner read beside the WCS-fused rows' bytes; the kernel has to MOVE 288.4 MB of directions + 552.5 MB)
            nv_k = 96
            dr = directions_in_run(imgs, first, nv_w, nv_k, fence)
            d_contract = ab['georef_dirs_in'] + ab['resample']
            d_moved = ab['georef_dirs_in'] + ab['image']
            variants['directions_in'] = {
                'ms_per_frame': dr['elapsed'] / nv_k * 1e3, 'Mpixels_per_s': nv_k * npx / 1e6 / dr['elapsed'], 'frames': nv_k,
                'kernel_ms_per_frame': dr['georef_ms'], 'single_pass_frames': sum(1 for q in dr['plans'] if q == 'single-pass'),
                'algorithmic_bytes_per_frame': d_contract, 'frac': frac(d_contract, dr['georef_ms']),
                'bytes_moved_min': d_moved, 'frac_bytes_moved_min': frac(d_moved, dr['georef_ms']),
                'repeats': dr['repeats'], 'kernel_ms_per_frame_min': dr['georef_ms_min'],
                'kernel': 'k_georef_rows<FAST, DIRS_IN, 0, BIN=uint16> (amt_pipe_launch_dirs_many), three frames per launch',
                'kernel_variant': dict(zip(('second', 'bin', 'frames_in_last_launch'), dr['variant'])), 'grid': dr['grid'],
                'direction_arrays_resident': 6}
            del dr
            gc.collect()
            torch.cuda.empty_cache()
            out['variants'] = variants
        if upload is not None:
            out.setdefault('variants', {})['upload'] = upload
        if world == 1 and args.cpu_rows > 0:
            parity = {}
            out['cpu_baseline'] = cpu_baseline(min(args.cpu_rows, HEIGHT), parity=parity)
            if args.cpu_procs > 0:
                # (ii) of SURVEY 8d: the same oracle, one process per frame on `procs` host cores (half the rows each: 16
                # whole frames at once would need ~50 GB of NumPy temporaries)
                procs = max(1, min(args.cpu_procs, os.cpu_count() or 1))
                out['cpu_baseline']['n_process'] = cpu_baseline_n_process(min(args.cpu_rows, HEIGHT // 2), procs)
            # max |dlat, dlon| vs ref (BASELINE.json's metric names it beside the throughput)
            out['parity'] = parity
        else:
            out['cpu_baseline'] = None
        # the JSON line is the LAST thing on stdout: flush what C libraries have buffered there (RCCL's banner),
        # print, and close the descriptor for whatever they print while shutting down
        # (RCCL's banner sits in the C library's buffer for stdout: it is flushed to stderr, so that stdout carries the one line)
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
