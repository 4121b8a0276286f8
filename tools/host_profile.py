"""cProfile of SequencePipeline.process in the grids-only mode (the one mode the host side bounds): top functions by own time."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
keep = len(sys.argv) > 1 and sys.argv[1] == 'arrays'
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H), keep_coordinates=keep)
frames = [sequence_frame(k, W, H)[:3] + (None,) for k in range(330)]
seq.process(frames[:30])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
seq.process(frames[30:])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
