"""Race hunt: long sequences with per-frame images through SequencePipeline (all batch sizes, both plans, MLat/MLT)
against one-frame-at-a-time results; every mismatch is reported.  PIN=1: per-frame images from pinned host memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from auromat_amd.pipeline import FramePipeline, SequencePipeline
from auromat_amd.synthetic import frame_image, sequence_frame
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (250, 168)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
frames = []
for k in range(n):
    hdr, cam, t, seed = sequence_frame(k, w, h)
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
        for rep in range(2):
            out = seq.process(seq_frames, keep_on_device=False)
            for k, (a, b) in enumerate(zip(out, ref)):
                for key in ('mean', 'count', 'img', 'mask'):
                    if not np.array_equal(a[key], b[key], equal_nan=True):
                        bad += 1
                        print('MISMATCH magnetic=%s plan=%s batch=%d rep=%d frame=%d %s' % (magnetic, plan, batch, rep, k, key))
        print('magnetic=%s %s batch=%d: plans %s, hinted %d' % (magnetic, plan, batch, sorted(set(seq.plans)), seq.hinted))
print('mismatches:', bad)
sys.exit(1 if bad else 0)
