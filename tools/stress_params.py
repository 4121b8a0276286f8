"""Single-pass vs two-pass plan over a grid of parameters at full size (exact / fast centres, uint8 / uint16, elevation
thresholds, resolutions, geodetic / magnetic, across the date line)."""
import os, sys, itertools
from datetime import timedelta
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import frame_header, frame_image, sequence_frame
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4240, 2832)
bad = n = 0
pipes = {}
for pointing, shift_min, fast, dtype, min_el, ppd, magnetic in itertools.product(
        ('iss030', 'iss029'), (0, 80), (True, False), (np.uint16, np.uint8), (10.0, None, 20.0), (10, (4, 7), 25), (False, True)):
    if (min_el, ppd) not in ((10.0, 10), (None, (4, 7)), (20.0, 25), (10.0, 25)):
        continue                                  # a slice of the full product
    if shift_min and pointing == 'iss030':
        continue
    hdr, cam, t = frame_header(w, h, pointing)
    t = t - timedelta(minutes=shift_min)
    key = (dtype, magnetic)
    if key not in pipes:
        pipes[key] = FramePipeline(w, h, img_dtype=dtype, with_mag=magnetic)
    pipe = pipes[key]
    img = frame_image(w, h, seed=7, dtype=dtype)
    two = pipe.run(hdr, 110, cam, t, img=img, fast=fast, min_elevation=min_el, pxPerDeg=ppd, magnetic=magnetic, fuse=False)
    one = pipe.run(hdr, 110, cam, t, fast=fast, min_elevation=min_el, pxPerDeg=ppd, magnetic=magnetic, fuse=True)
    n += 1
    ok = all(np.array_equal(one[k], two[k], equal_nan=True) for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'))
    if not ok:
        bad += 1
        d = int((one['count'] != two['count']).sum()) if one['count'].shape == two['count'].shape else -1
        print('MISMATCH', pointing, shift_min, 'fast' if fast else 'exact', dtype.__name__, min_el, ppd, 'mag' if magnetic else 'geo',
              pipe.last_plan, 'cells', d, one['count'].sum(), two['count'].sum())
print('cases', n, 'mismatches', bad)
