"""Maximum deviation of the fused kernel's coordinates from the oracle (degrees), small frames + full-size windows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as O
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import frame_header, frame_image

def oracle(hdr, cam, t, alt=110):
    et = O.date2es(t)
    return O.georef_frame(hdr, alt, cam, O.mat_j2000_to_geo(et), O.mat_j2000_to_sm(et), fast=True)

def report(tag, got, ref):
    out = []
    for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt'):
        a, b = got[k], ref[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), (tag, k)
        out.append('%s %.1e' % (k, np.nanmax(np.abs(a - b))))
    print(tag, ' '.join(out))

for w, h, pointing in ((253, 171, 'iss030'), (253, 171, 'iss029'), (640, 427, 'iss030')):
    hdr, cam, t = frame_header(w, h, pointing)
    pipe = FramePipeline(w, h, with_mag=True)
    pipe.run(hdr, 110, cam, t, img=frame_image(w, h, seed=1), pxPerDeg=10)
    report('%dx%d %s' % (w, h, pointing), pipe.host_arrays(), oracle(hdr, cam, t))
w, h = 4240, 2832
for pointing in ('iss030', 'iss029'):
    hdr, cam, t = frame_header(w, h, pointing)
    pipe = FramePipeline(w, h, with_mag=True)
    pipe.run(hdr, 110, cam, t, img=frame_image(w, h, seed=1), pxPerDeg=10, fuse=True)
    got = pipe.host_arrays()
    rows = np.where(~np.isnan(got['lat'][:, w // 2]))[0]
    for x0, y0, ww, wh in ((w // 2 - 100, max(int(rows[0]) - 60, 0), 200, 120), (0, h - 140, 180, 140), (w - 200, h // 2, 200, 150)):
        sub = dict(hdr, IMAGEW=ww, IMAGEH=wh, CRPIX1=hdr['CRPIX1'] - x0, CRPIX2=hdr['CRPIX2'] - y0)
        ref = oracle(sub, cam, t)
        win = {k: (got[k][y0:y0 + wh + 1, x0:x0 + ww + 1] if k in ('lat', 'lon', 'mlat', 'mlt') else got[k][y0:y0 + wh, x0:x0 + ww])
               for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt')}
        report('full %s window (%d,%d)' % (pointing, x0, y0), win, ref)
