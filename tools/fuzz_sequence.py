"""Sequence fuzz: frames that do NOT follow each other smoothly (random pointings, times up to 95 min apart — date-line
and pole frames included —, jumps back and forth, repeated frames) through SequencePipeline with every plan / batch
size and hints on, against one frame at a time.  usage: fuzz_sequence.py [sequences] [seed]
PINNED=1: the images are pinned host tensors (the library's loop uploads the rows that can be binned: amt_run_frame.img_host).
RESIDENT=1: the images are device tensors, which sends the sequence through the frame loop in the library (amt_run_*: the
single-pass plan, its hand-backs and the two-pass plan natively) instead of the Python loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from auromat_amd.pipeline import FramePipeline, SequencePipeline
from auromat_amd.synthetic import random_sequence

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
w, h = 250, 168
bad = 0
native = 0
for s in range(nseq):
    frames = random_sequence(rng, w, h)
    n = len(frames)
    magnetic = bool(rng.randint(2))
    ppd = float(rng.choice([4, 8, 10]))
    # a shell of its own for every frame (the altitude travels with the frame through the loop, and a pole frame's cell
    # coordinates depend on it)
    alts = [float(a) for a in rng.choice([95.0, 100.0, 110.0, 117.5, 120.0], n)]
    ref_pipe = FramePipeline(w, h, with_mag=magnetic)
    ref = [ref_pipe.run(hd, alt, cam, t, img=img, pxPerDeg=ppd, magnetic=magnetic) for (hd, cam, t, img), alt in zip(frames, alts)]
    frames = [f + (alt,) for f, alt in zip(frames, alts)]
    for plan, batch in (('single-pass', 1), ('single-pass', 2), ('single-pass', 3), ('two-pass', 1)):
        seq = SequencePipeline(w, h, pxPerDeg=ppd, plan=plan, batch=batch, magnetic=magnetic)
        # the production mode: nothing synchronises between frames, results come to the host after process() returns
        import torch
        from auromat_amd.resample import grid_coordinates
        feed = frames
        if os.environ.get('PINNED'):
            # (round 6) images in page-locked host memory: the library's loop uploads the rows that can be binned (img_host)
            feed = [(hd, cam, t, torch.from_numpy(img.view(np.int16) if img.dtype == np.uint16 else img).pin_memory(), alt)
                    for hd, cam, t, img, alt in frames]
        if os.environ.get('RESIDENT'):
            feed = [(hd, cam, t, torch.from_numpy(img.view(np.int16) if img.dtype == np.uint16 else img).cuda()) for hd, cam, t, img, alt in frames]
            feed = [f + (alt,) for f, alt in zip(feed, alts)]
            seq = SequencePipeline(w, h, pxPerDeg=ppd, plan=plan, batch=batch, magnetic=magnetic, own_image_buffers=False,
                                   img_dtype=frames[0][3].dtype)
        out = seq.process(feed, keep_on_device=True)
        native += type(out).__name__ == 'NativeResults'
        out = [dict({key: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for key, v in r.items()}, **grid_coordinates(r))
               for r in out]
        for i, (a, b) in enumerate(zip(out, ref)):
            for key in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
                x = a[key].view(b[key].dtype) if key == 'img' else (a[key].astype(b[key].dtype) if key == 'mask' else a[key])
                if not np.array_equal(x, b[key], equal_nan=True):
                    bad += 1
                    print('MISMATCH seq %d n %d magnetic %s %s batch %d frame %d %s plan %s' % (s, n, magnetic, plan, batch, i,
                                                                                              key, seq.plans[i]),
                          'pole', b['contains_pole'], 'date line', b['contains_discontinuity'], 'shapes', x.shape, b[key].shape,
                          'cells differing', int((~np.isclose(x, b[key], equal_nan=True)).sum()) if x.shape == b[key].shape else -1,
                          'plans', seq.plans)
                    break
    # round 4, box-first plan: a resolution per frame from its own box (arcsecPerPx) — the sequence loop (Python for host images,
    # the library's for resident ones) against one frame at a time through FramePipeline.run(arcsecPerPx=..., fuse=False)
    arcsec = float(rng.choice([300, 600, 900]))
    ref_bf = []
    for (hd, cam, t, img, alt) in frames:
        try:
            ref_bf.append(ref_pipe.run(hd, alt, cam, t, img=img, arcsecPerPx=arcsec, magnetic=magnetic, fuse=False))
        except (ValueError, AssertionError):
            ref_bf.append(None)                     # no valid pixel / a pole in view: no resolution (as the reference)
    for batch in (1, 3):
        feed = frames
        kw = {}
        if os.environ.get('PINNED'):
            feed = [(hd, cam, t, torch.from_numpy(img.view(np.int16) if img.dtype == np.uint16 else img).pin_memory(), alt)
                    for hd, cam, t, img, alt in frames]
        if os.environ.get('RESIDENT'):
            feed = [(hd, cam, t, torch.from_numpy(img.view(np.int16) if img.dtype == np.uint16 else img).cuda(), alt)
                    for hd, cam, t, img, alt in frames]
            kw = dict(own_image_buffers=False, img_dtype=frames[0][3].dtype)
        seq = SequencePipeline(w, h, arcsecPerPx=arcsec, batch=batch, magnetic=magnetic, **kw)
        out = seq.process(feed, keep_on_device=True)
        native += type(out).__name__ == 'NativeResults'
        for i, (r, b) in enumerate(zip(out, ref_bf)):
            if (r is None) != (b is None):
                bad += 1
                print('BOX-FIRST: frame %d of seq %d present on one side only' % (i, s), seq.plans[i], b is None)
                continue
            if r is None:
                continue
            a = dict({key: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for key, v in r.items()}, **grid_coordinates(r))
            for key in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
                x = a[key].view(b[key].dtype) if key == 'img' else (a[key].astype(b[key].dtype) if key == 'mask' else a[key])
                if not np.array_equal(x, b[key], equal_nan=True):
                    bad += 1
                    print('BOX-FIRST MISMATCH seq %d magnetic %s batch %d frame %d %s plan %s ppd %s / %s' % (
                        s, magnetic, batch, i, key, seq.plans[i], r['pxPerDeg'], b['pxPerDeg']))
                    break
print('sequences', nseq, 'through the library loop', native, 'of', 6 * nseq, 'runs; failures', bad)
sys.exit(1 if bad else 0)
