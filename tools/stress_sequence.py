"""Race hunt: long sequences with per-frame images through SequencePipeline (all batch sizes, both plans, MLat/MLT)
against one-frame-at-a-time results; every mismatch is reported.  PIN=1: per-frame images from pinned host memory.
Both result modes: keep_on_device=False (a synchronisation after every frame) and the production default
keep_on_device=True, where nothing synchronises between frames and the results are copied to the host only after
process() has returned — the mode in which a frame buffer can be re-used while a kernel still reads it;
every fifth frame looks at the pole (pole plan of the fused kernel: another kernel variant in the middle of a batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from auromat_amd.pipeline import FramePipeline, SequencePipeline
from auromat_amd.synthetic import frame_image, sequence_frame
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (250, 168)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
from datetime import datetime
from auromat_amd.coordinates import transform as T
tp = datetime(2012, 1, 25, 9, 26, 55)
zen = T.mat_j2000_to_geo(T.date2es(tp)).T.dot([0.0, 0.0, 1.0])
pole_hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0,
            'CRVAL1': np.rad2deg(np.arctan2(-zen[1], -zen[0])) % 360, 'CRVAL2': np.rad2deg(np.arcsin(-zen[2])),
            'CRPIX1': w / 2 + 0.5, 'CRPIX2': h / 2 + 0.5, 'CD1_1': -48.0 / w, 'CD1_2': 0.0, 'CD2_1': 0.0, 'CD2_2': 48.0 / w,
            'IMAGEW': w, 'IMAGEH': h}
frames = []
for k in range(n):
    hdr, cam, t, seed = sequence_frame(k, w, h)
    if k % 5 == 4:
        hdr, cam, t = pole_hdr, zen * (6356.75 + 400.0), tp
    frames.append((hdr, cam, t, frame_image(w, h, seed=seed)))
seq_frames = frames
if os.environ.get('PIN'):
    # images in pinned host memory: asynchronous uploads on the copy stream (buffers re-used while frames are in flight)
    import torch
    seq_frames = [(hd, cam, t, torch.from_numpy(img.view(np.int16)).pin_memory()) for hd, cam, t, img in frames]
bad = 0
for magnetic in (False, True):
    ref_pipe = FramePipeline(w, h, with_mag=magnetic)
    ref = [ref_pipe.run(hd, 110, cam, t, img=img, pxPerDeg=8, magnetic=magnetic) for hd, cam, t, img in frames]
    for plan, batch in (('single-pass', 1), ('single-pass', 2), ('single-pass', 3), ('two-pass', 1)):
        seq = SequencePipeline(w, h, pxPerDeg=8, plan=plan, batch=batch, magnetic=magnetic)
        for rep in range(4):
            on_device = rep >= 2
            out = seq.process(seq_frames, keep_on_device=on_device)
            if on_device:
                import torch
                out = [{key: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for key, v in r.items()} for r in out]
            for k, (a, b) in enumerate(zip(out, ref)):
                for key in ('mean', 'count', 'img', 'mask'):
                    x = a[key].view(b[key].dtype) if key == 'img' else a[key].astype(b[key].dtype)
                    if not np.array_equal(x, b[key], equal_nan=True):
                        bad += 1
                        print('MISMATCH magnetic=%s plan=%s batch=%d rep=%d frame=%d %s' % (magnetic, plan, batch, rep, k, key))
        print('magnetic=%s %s batch=%d: plans %s, hinted %d' % (magnetic, plan, batch, sorted(set(seq.plans)), seq.hinted))
print('mismatches:', bad)
sys.exit(1 if bad else 0)
