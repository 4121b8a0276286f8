"""Operator-level entry points at N = 1e7 points (the size of the reference's own test/benchmark.py:30-31) with
device-resident inputs and outputs: time per call (HIP events), bytes moved, fraction of the 8 TB/s HBM peak."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd._native import Context, host3, host9, ptr
from auromat_amd.coordinates import transform as T
from datetime import datetime
N = 10_000_000
ctx = Context.current()
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(1)
lat = (torch.rand(N, generator=g, device=dev, dtype=torch.float64) - 0.5) * 3.0
lon = (torch.rand(N, generator=g, device=dev, dtype=torch.float64) - 0.5) * 6.2
r = 6378.0 + 110.0
xyz = torch.stack((r * torch.cos(lat) * torch.cos(lon), r * torch.cos(lat) * torch.sin(lon), r * torch.sin(lat)), dim=1).contiguous()
x, y, z = xyz[:, 0].contiguous(), xyz[:, 1].contiguous(), xyz[:, 2].contiguous()
cam = np.array([-4809.524217485676, 524.8117887762777, 4729.265809729493])
dirs = xyz - torch.from_numpy(cam).to(dev)
dirs = (dirs / dirs.norm(dim=1, keepdim=True)).contiguous()
o1, o2, o3 = (torch.empty(N, dtype=torch.float64, device=dev) for _ in range(3))
o3v = torch.empty((N, 3), dtype=torch.float64, device=dev)
ob = torch.empty(N, dtype=torch.uint8, device=dev)
m = T.mat_j2000_to_geo(T.date2es(datetime(2012, 1, 25, 9, 26, 55)))
a, b = 6378.137, 6356.752314245179
cases = [
    ('amt_ecef_to_geodetic', 40, lambda: ctx.call('amt_ecef_to_geodetic', ptr(x), ptr(y), ptr(z), N, a, b, ptr(o1), ptr(o2))),
    ('amt_geodetic_to_ecef', 40, lambda: ctx.call('amt_geodetic_to_ecef', ptr(lat), ptr(lon), 110.0, N, a, b, ptr(o1), ptr(o2), ptr(o3))),
    ('amt_intersect_ellipsoid', 48, lambda: ctx.call('amt_intersect_ellipsoid', a + 110, b + 110, host3(cam), ptr(dirs), N, 1, ptr(o3v))),
    ('amt_intersects_ellipsoid', 25, lambda: ctx.call('amt_intersects_ellipsoid', a, b, host3(cam), ptr(dirs), N, 1, ptr(ob))),
    ('amt_rotate_to_latlon', 40, lambda: ctx.call('amt_rotate_to_latlon', host9(m), ptr(xyz), N, a, b, ptr(o1), ptr(o2))),
    ('amt_rotate_to_mlat_mlt', 40, lambda: ctx.call('amt_rotate_to_mlat_mlt', host9(m), ptr(xyz), N, ptr(o1), ptr(o2))),
    ('amt_rotate_vectors', 48, lambda: ctx.call('amt_rotate_vectors', host9(m), ptr(xyz), N, ptr(o3v))),
    ('torch copy (read + write 80 B/pt)', 80, lambda: (o3v.copy_(xyz), None)[1]),
]
for name, bytes_per_pt, fn in cases:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('%-36s %8.1f us  %5.0f MB  %.2f TB/s = %.2f of 8 TB/s' % (name, ms * 1e3, bytes_per_pt * N / 1e6, bytes_per_pt * N / ms / 1e9, bytes_per_pt * N / ms / 1e9 / 8))
