"""cProfile of the per-frame host work of the pipeline (run on the GPU box)."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
pipe = FramePipeline(W, H)
pipe.set_image(frame_image(W, H))
fuse = sys.argv[1:] != ['two-pass']
def frame(k):
    hdr, cam, t, _ = sequence_frame(k, W, H)
    p = frame_params(hdr, 110, cam, t, True, magnetic=False)
    pipe.georef(None, 110, cam, t, True, 10.0, params=p, fuse_pxPerDeg=(10, 10) if fuse else None)
    return pipe.resample(10, containsPole=False, keep_on_device=True)
for k in range(5): frame(k)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for k in range(5, 45): frame(k)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
