"""method='cubic' on the exact triangulation at full size (4240 x 2832, ~7 M valid pixel centres after maskedByElevation(10)):
where the time goes — host triangulation, neighbour lists, Gauss-Seidel sweeps in scipy's order, point location, element."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd._native import lib, ptr, to_host
from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
from auromat_amd.resample import cached_grid, cubic_exact, outside_outline_mask
from auromat_amd.synthetic import frame_header, frame_image
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4240, 2832)
hdr, cam, t = frame_header(w, h, 'iss030')
m = ArraySpacecraftMapping(hdr, 110, frame_image(w, h, seed=1), cam, t, 'f', fastCenterCalculation=True).maskedByElevation(10)
fd = m.frame()
ctx = fd.ctx
bb = m.boundingBox
grid = cached_grid((10, 10), bb.latSouth, bb.latNorth, bb.lonWest, bb.lonEast)
la, lo = fd.lat_c.reshape(-1), fd.lon_c.reshape(-1)
valid = ~(fd.center_mask_tensor().bool().reshape(-1) | ~(fd.elev.reshape(-1) >= 10.0)) & ~torch.isnan(la)
vals = torch.cat((fd.img.reshape(-1, 3).to(torch.int32).bitwise_and(0xffff).to(torch.float64), fd.elev.reshape(-1, 1)), dim=1)
target_mask = outside_outline_mask(ctx, grid, np.array(m.outline, dtype=np.float64))
print('valid pixels %d of %d, grid %d x %d' % (int(valid.sum()), h * w, grid.ny, grid.nx), flush=True)
# the stages one by one
L = lib()
idx = torch.nonzero(valid).reshape(-1)
n = int(idx.numel())
xy = torch.stack((la[idx], lo[idx]), dim=1).contiguous()
torch.cuda.synchronize(); t0 = time.perf_counter()
xy_host = np.ascontiguousarray(to_host(xy))
t1 = time.perf_counter()
handle = C.c_void_p()
assert L.amt_delaunay_create(xy_host.ctypes.data_as(C.c_void_p), n, C.byref(handle)) == 0
t2 = time.perf_counter()
print('download %.2f s, triangulation %.2f s (%.2f us per point)' % (t1 - t0, t2 - t1, (t2 - t1) / n * 1e6), flush=True)
L.amt_delaunay_destroy(handle)
torch.cuda.synchronize(); t0 = time.perf_counter()
out, sweeps = cubic_exact(ctx, la, lo, valid, vals, h, w, grid, target_mask)
torch.cuda.synchronize(); t1 = time.perf_counter()
print('cubic_exact: %.2f s in all, sweeps per channel %s' % (t1 - t0, sweeps), flush=True)
print('filled cells %d' % int((~torch.isnan(out[:, 0])).sum()))
