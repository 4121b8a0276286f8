import time, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame
W,H=4240,2832
dev=torch.device('cuda',0)
imgs=[torch.randint(0,65535,(H,W,3),device=dev,dtype=torch.int32).to(torch.int16) for _ in range(8)]
SH=(100,110,120)
fr=[sequence_frame(k,W,H)[:3]+(imgs[k%8],SH[k%3]) for k in range(105)]
for geo in (True, False, True, False):
    seq=SequencePipeline(W,H,magnetic=True,geodetic_arrays=geo)
    for _ in range(3): seq.process(fr[:9]); torch.cuda.synchronize()
    seq.ctx.timing_enable(1)
    torch.cuda.synchronize(); t0=time.perf_counter()
    r=seq.process(fr[9:]); torch.cuda.synchronize(); el=time.perf_counter()-t0
    ms,n=seq.ctx.timing_read(0); seq.ctx.timing_enable(False)
    print('geodetic_arrays',geo,'variant',seq.ctx.last_variant(),'ms/frame %.4f kernel us/frame %.1f plans'%(el/96*1e3, ms/n*1e3), set(seq.plans), flush=True)
    del seq, r
    import gc; gc.collect(); torch.cuda.empty_cache()
