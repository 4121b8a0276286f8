"""
Deterministic synthetic frames for tests and bench.py.

The numbers are *inputs*: the WCS cards, camera position and photo time of the
two real headers the reference ships as test data
(/root/reference/auromat/test/resources/ISS030-E-102170_dc.wcs and
ISS029-E-8492.wcs: `CRVAL*`, `CD*`, `POS*[SHIF]`, `DATE-OBS` + `DATESHIF`),
re-scaled to arbitrary image sizes.  No network, no files.
"""
from datetime import datetime, timedelta

import numpy as np

# name -> (CRVAL1, CRVAL2, CD11, CD12, camera xyz [km, GCRS], photo time); CD21 = -CD12, CD22 = CD11;
# native size 4256 x 2832, CRPIX = (2129, 1417), LONPOLE = 180, LATPOLE = 0
_POINTINGS = {
    # northern hemisphere; SM longitudes do not wrap
    'iss030': (16.0531567459, 23.1148929108, -0.00912247310646, -0.00250608809647,
               (-4809.524217485676, 524.8117887762777, 4729.265809729493),
               datetime(2012, 1, 25, 9, 26, 55, 60000)),
    # southern hemisphere; MLT spans 0..24 h => SM longitude crosses +-180 (discontinuity branch)
    'iss029': (140.917604745, -25.9728268931, 0.0247522916736, 0.00429015193241,
               (3631.092688542516, -2155.308046530645, -5284.393739145378),
               datetime(2011, 9, 18, 11, 54, 56)),
}
NATIVE_W, NATIVE_H = 4256, 2832


def frame_header(width, height, pointing='iss030'):
    """
    WCS header dict (the cards reference wcs.py:80-90 / astrometry.py:260 read)
    showing the same sky as the native 4256x2832 frame but sampled on
    width x height pixels.  Returns (header, cameraPosGCRS, photoTime).
    """
    ra, dec, cd11, cd12, cam, t = _POINTINGS[pointing]
    sx = NATIVE_W / float(width)
    sy = NATIVE_H / float(height)
    # keep pixels square in angle unless the aspect ratio changes
    hdr = {
        'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0,
        'CRVAL1': ra, 'CRVAL2': dec,
        'CRPIX1': width / 2.0 + 0.5, 'CRPIX2': height / 2.0 + 0.5,
        'CD1_1': cd11 * sx, 'CD1_2': cd12 * sy, 'CD2_1': -cd12 * sx, 'CD2_2': cd11 * sy,
        'IMAGEW': int(width), 'IMAGEH': int(height),
    }
    return hdr, np.array(cam, dtype=np.float64), t


def frame_image(width, height, seed=0, dtype=np.uint16):
    """Seeded RGB image, (height, width, 3)."""
    hi = 65535 if np.dtype(dtype) == np.uint16 else 255
    return np.random.RandomState(seed).randint(0, hi, (height, width, 3)).astype(dtype)


def sequence_frame(k, width=4240, height=2832, pointing='iss030'):
    """
    Frame k of a synthetic sequence: pointing drifts by
    0.05 deg of RA per frame, time advances 1 s per frame and the camera moves
    along a great-circle arc at 7.66 km/s.  Returns (header, camera, time, image seed).
    """
    hdr, cam, t = frame_header(width, height, pointing)
    hdr['CRVAL1'] = hdr['CRVAL1'] + 0.05 * k
    r = np.linalg.norm(cam)
    u = cam / r
    # fixed in-plane direction perpendicular to the camera position
    v = np.cross([0.0, 0.0, 1.0], u)
    v /= np.linalg.norm(v)
    ang = 7.66 * k / r
    cam_k = r * (np.cos(ang) * u + np.sin(ang) * v)
    return hdr, cam_k, t + timedelta(seconds=k), k


def random_sequence(rng, width, height, max_frames=40):
    """
    A sequence whose frames do NOT follow each other smoothly (what tools/fuzz_sequence.py feeds to SequencePipeline):
    stretches of sequence_frame(), jumps to other pointings at times up to 95 min apart (date-line and pole frames
    included), repeated frames, jumps back and forth inside the smooth sequence.

    :param rng: np.random.RandomState (consumed in a fixed order: a seed names its sequences)
    :return: list of (header, cameraPosGCRS, time, image)
    """
    n = int(rng.randint(1, max_frames))
    frames = []
    k = 0
    while len(frames) < n:
        mode = rng.randint(4)
        if mode == 0:                                     # a smooth stretch of the synthetic sequence
            for _ in range(int(rng.randint(1, 8))):
                hdr, cam, t, seed = sequence_frame(k, width, height)
                frames.append((hdr, cam, t, frame_image(width, height, seed=seed)))
                k += 1
        elif mode == 1:                                   # a jump somewhere else in time / pointing
            hdr, cam, t = frame_header(width, height, ('iss030', 'iss029')[rng.randint(2)])
            t = t - timedelta(minutes=float(rng.choice([0, 20, 45, 80, 95])))
            frames.append((hdr, cam, t, frame_image(width, height, seed=1000 + len(frames))))
        elif mode == 2 and frames:                        # the same frame again
            frames.append(frames[-1])
        else:
            k = int(rng.randint(0, 300))
    return frames[:n]


def pole_frame(width=4240, height=2832, south=False):
    """
    A frame with a pole in view: camera 400 km above 83 deg latitude looking across the pole, 66 x 53 deg field of view
    (the geometry of tests/golden/pole_frame_*.npz scaled to `width` x `height`).  Returns (header, camera, time).
    """
    from .coordinates import transform as T
    t = datetime(2012, 1, 25, 9, 26, 55, 60000)
    m_geo = np.asarray(T.mat_j2000_to_geo(T.date2es(t)))
    sgn = -1.0 if south else 1.0

    def geo(lat, lon, r):
        la, lo = np.deg2rad(lat), np.deg2rad(lon)
        return r * np.array([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)])

    cam_geo = geo(sgn * 83.0, 30.0, 6360.0 + 400.0)
    bore = m_geo.T.dot(geo(sgn * 87.5, -140.0, 6360.0 + 110.0) - cam_geo)
    bore /= np.linalg.norm(bore)
    s = 200.0 / width
    hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0,
           'CRVAL1': float(np.rad2deg(np.arctan2(bore[1], bore[0])) % 360), 'CRVAL2': float(np.rad2deg(np.arcsin(bore[2]))),
           'CRPIX1': width / 2 + 0.5, 'CRPIX2': height / 2 + 0.5, 'CD1_1': -0.33 * s, 'CD1_2': 0.05 * s, 'CD2_1': 0.05 * s,
           'CD2_2': 0.33 * s, 'IMAGEW': width, 'IMAGEH': height}
    return hdr, m_geo.T.dot(cam_geo), t
