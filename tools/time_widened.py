"""Wall time of the widened paths at BASELINE's full frame size: traced outline + centroid, resample(method='nearest'),
and the all-sky mapping at its native 512 x 512 (tools/README.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
from auromat_amd.resample import resample_frame
from auromat_amd.synthetic import frame_header, frame_image

w, h = 4240, 2832
hdr, cam, t = frame_header(w, h, 'iss030')
img = frame_image(w, h, seed=1)
m = ArraySpacecraftMapping(hdr, 110, img, cam, t, 'f', fastCenterCalculation=True).maskedByElevation(10)
fd = m.frame()
bb = m.boundingBox


def timed(label, fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    print('%-44s %8.2f ms' % (label, (time.perf_counter() - t0) / n * 1e3))
    return out


def outline():
    m._host.pop('outline', None)
    m._host.pop('outline_px', None)
    return m.outline


o = timed('traced outline (12 Mpx mask -> polygon)', outline)
print('  outline vertices', len(o))
timed('centroid (cached outline)', lambda: m.centroid)
timed("resample_frame method='mean' (two-pass)", lambda: resample_frame(fd, 110, bb, (10, 10), False, False, keep_on_device=True))
timed("resample_frame method='nearest' 0.1 deg", lambda: resample_frame(fd, 110, bb, (10, 10), False, False, keep_on_device=True, method='nearest', outline=o))
timed("resample_frame method='nearest' 0.04 deg", lambda: resample_frame(fd, 110, bb, (25, 25), False, False, keep_on_device=True, method='nearest', outline=o))

# class API end to end (device arrays -> masked host arrays of the resampled mapping)
from auromat_amd.resample import resample
timed("resample(mapping, pxPerDeg=10) class API, host arrays out", lambda: resample(m, pxPerDeg=10).img, n=3)
timed("resample(mapping, 10, method='nearest') class API", lambda: resample(m, pxPerDeg=10, method='nearest').img, n=3)


def fresh_bbox():
    m.setDirty()
    return m.boundingBox


timed('boundingBox from scratch (reduction + traced outline)', fresh_bbox)
