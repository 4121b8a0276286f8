"""cProfile of `auromat-convert --resample` (reference defaults: 100 arcsec per pixel, MLat/MLT grid) on 12 full-size .npy frames:
where the 50 ms per frame of a convert run go (file reads, upload, export)."""
import os, sys, time, tempfile, shutil, json, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from auromat_amd.cli.convert import main
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
d = tempfile.mkdtemp()
try:
    for k in range(12):
        hdr, cam, t, seed = sequence_frame(k, W, H)
        np.save(os.path.join(d, 'f%02d.npy' % k), frame_image(W, H, seed=seed))
        hdr = dict(hdr, POSX=float(cam[0]), POSY=float(cam[1]), POSZ=float(cam[2]))
        hdr['DATE-OBS'] = t.strftime('%Y-%m-%dT%H:%M:%S.%f')
        json.dump(hdr, open(os.path.join(d, 'f%02d.json' % k), 'w'))
    args = ['--data', d, '--format', 'netcdf', '--resample', '--min-elevation', '10', '--without-bounds']
    main(args + ['--out', os.path.join(d, 'o0')])
    pr = cProfile.Profile(); pr.enable(); t0 = time.time()
    main(args + ['--out', os.path.join(d, 'o1')])
    el = time.time() - t0; pr.disable()
    print('%.1f ms per frame' % (el / 12 * 1e3))
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
finally:
    shutil.rmtree(d, ignore_errors=True)
