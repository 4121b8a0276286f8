"""Single-pass plan against the ORACLE at full size: counts must be identical cell by cell (the reference's
right-most-edge rule included), integer image means identical, for a few frames and resolutions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as O
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import frame_image, sequence_frame
w, h = 4240, 2832
bad = 0
for k, ppd in ((0, 8), (0, 10), (1, 10), (2, 7), (3, 10)):
    hdr, cam, t, seed = sequence_frame(k, w, h)
    img = frame_image(w, h, seed=seed)
    pipe = FramePipeline(w, h)
    one = pipe.run(hdr, 110, cam, t, img=img, pxPerDeg=ppd, fuse=True)
    et = O.date2es(t)
    g = O.georef_frame(hdr, 110, cam, O.mat_j2000_to_geo(et), None, fast=True)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    want = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), 110, data,
                           None, bbox, (ppd, ppd), disc, False)
    same_shape = want['data'].shape == one['mean'].shape
    ndiff = int((want['count'] != one['count']).sum()) if same_shape else -1
    filled = (want['count'] == one['count']) & (want['count'] > 0) if same_shape else None
    img_ok = bool(np.array_equal(one['mean'][..., :3][filled], want['data'][..., :3][filled])) if same_shape else False
    print('frame %d ppd %d plan %s edge pixels %d: shape ok %s, cells with different count %d, sums %d vs %d, image means identical %s'
          % (k, ppd, pipe.last_plan, pipe._fused['result'].edge_pixels, same_shape, ndiff, one['count'].sum(), want['count'].sum(), img_ok))
    bad += (not same_shape) + (ndiff != 0) + (not img_ok)
print('deviations:', bad)
