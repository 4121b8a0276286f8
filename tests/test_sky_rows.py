"""
amt_georef_sky_rows (host function, no GPU): the rows of work items whose waves write NaN instead of casting rays.
Sound — no corner of such a band hits the shell, checked against the oracle's ray casting of EVERY corner — and tight
— the bands next to the first / last row with a hit are at most one band away.  Frames: the two real pointings at
several sizes, Earth above instead of below (camera rolled by 180 deg), Earth to the side (rolled by 90 deg), a pole
frame, the zenith (no hit at all) and the nadir (no sky at all).
"""
import ctypes as C

import numpy as np
import pytest


def sky_rows(hdr, cam, t, altitude=110):
    from auromat_amd import _native
    from auromat_amd.mapping.astrometry import frame_params
    p = frame_params(hdr, altitude, cam, t, True)
    out = [C.c_int32(0) for _ in range(4)]
    rc = _native.lib().amt_georef_sky_rows(C.byref(p), *[C.byref(o) for o in out])
    assert rc == 0
    return [o.value for o in out]


def hits_per_corner_row(hdr, cam, t, altitude=110):
    from oracle import ref_numpy as O
    from auromat_amd.coordinates import transform as T
    g = O.georef_frame(hdr, altitude, cam, O.mat_j2000_to_geo(T.date2es(t)), None, fast=True)
    return (~np.isnan(g['lat'])).any(axis=1)


def rolled(hdr, deg):
    """The same boresight with the camera rolled about it: CD -> CD R(deg)."""
    c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    cd = np.array([[hdr['CD1_1'], hdr['CD1_2']], [hdr['CD2_1'], hdr['CD2_2']]]).dot(np.array([[c, -s], [s, c]]))
    return dict(hdr, CD1_1=cd[0, 0], CD1_2=cd[0, 1], CD2_1=cd[1, 0], CD2_2=cd[1, 1])


def look(hdr, v):
    return dict(hdr, CRVAL1=float(np.rad2deg(np.arctan2(v[1], v[0])) % 360), CRVAL2=float(np.rad2deg(np.arcsin(v[2]))))


def cases():
    from auromat_amd.synthetic import frame_header, pole_frame
    out = []
    for pointing in ('iss030', 'iss029'):
        for w, h in ((530, 354), (1060, 708), (333, 1000)):
            hdr, cam, t = frame_header(w, h, pointing)
            out.append(('%s %dx%d' % (pointing, w, h), hdr, cam, t))
        hdr, cam, t = frame_header(640, 420, pointing)
        out.append((pointing + ' rolled 180', rolled(hdr, 180), cam, t))
        out.append((pointing + ' rolled 90', rolled(hdr, 90), cam, t))
        out.append((pointing + ' rolled 37', rolled(hdr, 37), cam, t))
        up = cam / np.linalg.norm(cam)
        out.append((pointing + ' zenith', look(hdr, up), cam, t))
        out.append((pointing + ' nadir', look(hdr, -up), cam, t))
    hdr, cam, t = pole_frame(400, 320)
    out.append(('pole', hdr, cam, t))
    return out


@pytest.mark.parametrize('case', cases(), ids=lambda c: c[0])
def test_sky_rows_are_sound_and_tight(case):
    name, hdr, cam, t = case
    rows, n, top, bottom = sky_rows(hdr, cam, t)
    h = hdr['IMAGEH']
    assert rows == 16 and n == (h + 15) // 16 and 0 <= top <= bottom <= n
    hit = hits_per_corner_row(hdr, cam, t)                 # (h + 1,) any corner of that corner row hits
    for c in list(range(top)) + list(range(bottom, n)):
        assert not hit[c * rows:min((c + 1) * rows, h) + 1].any(), (name, c)
    if not hit.any():
        assert top == n, name                               # a frame of sky is all in the first range
        return
    first, last = np.flatnonzero(hit)[[0, -1]]
    assert top >= first // rows - 1 and bottom <= last // rows + 2, (name, top, bottom, first, last)
    if 'nadir' in name:
        assert (top, bottom) == (0, n)


def test_sky_rows_of_the_bench_frame():
    from auromat_amd.synthetic import sequence_frame
    hdr, cam, t, _ = sequence_frame(0, 4240, 2832)
    rows, n, top, bottom = sky_rows(hdr, cam, t)
    # the limb crosses the bench frame between rows 1140 and 1300 (43 % of the frame is sky)
    assert (rows, n, bottom) == (16, 177, 177) and 60 <= top <= 72, (top, bottom)


# ---- amt_georef_image_rows (round 6): the rows of a host image that have to cross the link --------------------------------------
def image_rows(hdr, cam, t, min_elev, altitude=110, fast=True):
    from auromat_amd import _native
    from auromat_amd.mapping.astrometry import frame_params
    p = frame_params(hdr, altitude, cam, t, fast)
    r0, r1 = C.c_int32(-1), C.c_int32(-1)
    assert _native.lib().amt_georef_image_rows(C.byref(p), float(min_elev), C.byref(r0), C.byref(r1)) == 0
    return r0.value, r1.value


def binned_pixel_rows(hdr, cam, t, min_elev, altitude=110, fast=True):
    """Pixel rows that hold a pixel the binning reads: centre on the shell and elevation >= min_elev (the oracle's, every pixel)."""
    from oracle import ref_numpy as O
    from auromat_amd.coordinates import transform as T
    g = O.georef_frame(hdr, altitude, cam, O.mat_j2000_to_geo(T.date2es(t)), None, fast=fast)
    with np.errstate(invalid='ignore'):
        ok = ~np.isnan(g['lat_c']) & (g['elev'] >= min_elev)
    return ok.any(axis=1)


@pytest.mark.parametrize('case', cases(), ids=lambda c: c[0])
@pytest.mark.parametrize('min_elev', [-np.inf, 0.0, 3.0, 10.0, 25.0, 60.0])
def test_image_rows_are_sound_and_tight(case, min_elev):
    name, hdr, cam, t = case
    h = hdr['IMAGEH']
    for altitude, fast in ((110, True), (300, False)):
        r0, r1 = image_rows(hdr, cam, t, min_elev, altitude, fast)
        assert 0 <= r0 <= r1 <= h
        need = binned_pixel_rows(hdr, cam, t, min_elev, altitude, fast)
        assert not need[:r0].any() and not need[r1:].any(), (name, min_elev, altitude, r0, r1, np.flatnonzero(need)[[0, -1]])
        if not need.any():
            continue
        first, last = np.flatnonzero(need)[[0, -1]]
        # at most three bands of 16 rows more than needed on either side (one for the limb's tolerance, one for the pixel the bands
        # are widened by, one the elevation bound adds)
        assert r0 >= (first // 16 - 3) * 16 and r1 <= (last // 16 + 4) * 16, (name, min_elev, altitude, r0, r1, first, last)
        if not min_elev > 0:
            rows, n, top, bottom = sky_rows(hdr, cam, t, altitude)
            assert (r0, r1) == (top * rows, min(h, bottom * rows))


def test_image_rows_of_the_bench_frame():
    from auromat_amd.synthetic import sequence_frame
    hdr, cam, t, _ = sequence_frame(0, 4240, 2832)
    e0, e1 = image_rows(hdr, cam, t, -np.inf)
    r0, r1 = image_rows(hdr, cam, t, 10.0)
    # the rows a ray can hit: 62 % of the frame; those with a pixel above 10 deg of elevation: fewer
    assert e1 == r1 == 2832 and 0.55 < (e1 - e0) / 2832.0 < 0.66 and r0 > e0 + 200, (e0, e1, r0, r1)
