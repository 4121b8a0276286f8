"""Which plan full-size frames take (single-pass unless the coarse box misjudges the exact one): both synthetic pointings,
thresholds incl. none, fast / exact centres, geodetic / MLat-MLT grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import frame_header, frame_image
W, H = 4240, 2832
img = frame_image(W, H)
for magnetic in (False, True):
    pipe = FramePipeline(W, H, with_mag=magnetic)
    pipe.set_image(img)
    for pointing in ('iss030', 'iss029'):
        for thr in (None, 5.0, 10.0):
            for fast in (True, False):
                hdr, cam, t = frame_header(W, H, pointing)
                res = pipe.run(hdr, 110, cam, t, fast=fast, min_elevation=thr, pxPerDeg=10, fuse=True, magnetic=magnetic)
                print(pointing, 'thr', thr, 'fast' if fast else 'exact', 'magnetic' if magnetic else 'geodetic', '->', pipe.last_plan,
                      res['count'].shape)
