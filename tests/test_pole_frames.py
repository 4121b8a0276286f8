"""
Camera frames with a pole in view (reference resample.py:176-201,262-273: the data is rotated by +90 deg about x before
binning, the grid coordinates are rotated back afterwards).  Fixtures pole_frame_{north,south}_{fast,exact}.npz come from
the REAL reference (oracle/make_golden.py: pole_frames): a 200x160 frame of a camera 400 km above 83 deg latitude looking
across the pole, maskedByElevation(10) -> _resample(containsPole=True, pxPerDeg=8).

CPU: the oracle against the fixture.  GPU: the single-pass plan (pole plan of the fused kernel) and the two-pass plan
against the fixture and against each other, cell for cell.
"""
from datetime import datetime

import numpy as np
import pytest

from conftest import header_from, load_golden

CASES = [(side, mode) for side in ('north', 'south') for mode in ('fast', 'exact')]


def parse(s):
    return datetime.strptime(str(s), '%Y-%m-%dT%H:%M:%S.%f')


@pytest.mark.parametrize('side,mode', CASES)
def test_oracle_pole_branch_equals_the_reference(side, mode):
    from oracle import ref_numpy as O
    z = load_golden('pole_frame_%s_%s.npz' % (side, mode))
    hdr = header_from(z)
    g = O.georef_frame(hdr, 110.0, z['cam'], z['m_geo'], z['m_sm'], fast=mode == 'fast')
    corner_nan, center_nan = np.isnan(g['lat']), np.isnan(g['lat_c'])
    if mode == 'exact':
        corner_nan, center_nan = O.sanitize_masks(corner_nan, center_nan)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], corner_nan, 10)
    assert np.array_equal(np.where(center_mask, np.nan, g['lat_c']), z['lat_c'], equal_nan=True)
    assert np.array_equal(np.where(center_mask, np.nan, g['lon_c']), z['lon_c'], equal_nan=True)
    outline = np.transpose([g['lat'][~corner_mask], g['lon'][~corner_mask]])
    data = np.dstack((z['img'].astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    res = O.resample_mean(z['lat_c'], z['lon_c'], 110.0, data, outline, z['bbox'], tuple(z['ppd']), True, True)
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c'),
                 ('data', 'out_data')):
        assert np.array_equal(res[a], z[b], equal_nan=True), a


def check_against_fixture(res, z):
    assert res['contains_pole']
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c')):
        assert res[a].shape == z[b].shape, a
        d = np.abs(res[a] - z[b])
        if a.startswith('lon'):
            d = np.minimum(d, 360 - d)
        assert np.max(d) < 1e-9, (a, np.max(d))      # the grid is rotated back on the device
    want = z['out_data']
    assert np.array_equal(res['mask'], np.isnan(want[..., 0]))
    ok = ~res['mask']
    assert ok.sum() > 2000
    assert np.array_equal(res['mean'][..., :3][ok], want[..., :3][ok])                  # exact integer sums / counts
    assert np.max(np.abs(res['mean'][..., 3][ok] - want[..., 3][ok])) < 1e-9            # elevation, fixed point
    assert np.array_equal(res['img'][ok], z['out_img'][ok])


@pytest.mark.gpu
@pytest.mark.parametrize('side,mode', CASES)
def test_single_pass_plan_handles_pole(side, mode):
    """Pole in view: the fused kernel bins in the rotated coordinates; identical to the two-pass plan and the reference."""
    from auromat_amd.pipeline import FramePipeline
    z = load_golden('pole_frame_%s_%s.npz' % (side, mode))
    hdr = header_from(z)
    t = parse(z['time_iso'])
    w, h = hdr['IMAGEW'], hdr['IMAGEH']
    out = {}
    for fuse in (True, False):
        pipe = FramePipeline(w, h)
        res = pipe.run(hdr, 110, z['cam'], t, img=z['img'], fast=mode == 'fast', min_elevation=10, pxPerDeg=8, fuse=fuse)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        check_against_fixture(res, z)
        bb = pipe.bounding_box()
        assert bb.containsPole
        np.testing.assert_allclose([bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast], z['bbox'], rtol=0, atol=1e-9)
        out[fuse] = res
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
        assert np.array_equal(out[True][k], out[False][k], equal_nan=True), k
    # grids only (no coordinate arrays are written at all) and a caller that states the pole itself
    pipe = FramePipeline(w, h, alloc_coords=False)
    res = pipe.run(hdr, 110, z['cam'], t, img=z['img'], fast=mode == 'fast', min_elevation=10, pxPerDeg=8, fuse=True,
                   containsPole=True)
    assert pipe.last_plan == 'single-pass' and pipe.fd.lat is None
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
        assert np.array_equal(res[k], out[False][k], equal_nan=True), k


@pytest.mark.gpu
def test_pole_frames_inside_a_sequence():
    """Pole frames between ordinary ones: the box hints of neighbours are in other coordinates and must not be used."""
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.synthetic import frame_image
    w, h = 200, 160
    z = load_golden('pole_frame_north_fast.npz')
    hdr0 = header_from(z)
    t0 = parse(z['time_iso'])
    cam0 = np.asarray(z['cam'])
    from auromat_amd.pipeline import EmptyFrame
    ref_pipe = FramePipeline(w, h)
    frames, ref = [], []
    a0, d0 = np.deg2rad(hdr0['CRVAL1']), np.deg2rad(hdr0['CRVAL2'])
    bore0 = np.array([np.cos(d0) * np.cos(a0), np.cos(d0) * np.sin(a0), np.sin(d0)])
    nadir = -cam0 / np.linalg.norm(cam0)
    for k in range(10):
        hdr = dict(hdr0)
        # the boresight swings from across the pole down to the nadir (7 deg of latitude away from the pole, which is
        # then outside the frame) and on: some frames contain the pole, others do not
        s_ = 0.17 * k
        b = (1 - s_) * bore0 + s_ * nadir
        b /= np.linalg.norm(b)
        hdr['CRVAL1'] = float(np.rad2deg(np.arctan2(b[1], b[0])) % 360)
        hdr['CRVAL2'] = float(np.rad2deg(np.arcsin(b[2])))
        img = frame_image(w, h, seed=k)
        try:
            ref.append(ref_pipe.run(hdr, 110, cam0, t0, img=img, pxPerDeg=8))
        except EmptyFrame:
            continue
        frames.append((hdr, cam0, t0, img))
    assert len(frames) >= 5
    poles = [r['contains_pole'] for r in ref]
    assert any(poles) and not all(poles)
    for keep_coordinates in (True, False):
        seq = SequencePipeline(w, h, pxPerDeg=8, plan='single-pass', keep_coordinates=keep_coordinates)
        for rep in range(2):
            out = seq.process(frames, keep_on_device=False)
            assert seq.plans == ['single-pass'] * len(frames), seq.plans
            for a, b in zip(out, ref):
                assert a['contains_pole'] == b['contains_pole']
                for key in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
                    assert np.array_equal(a[key], b[key], equal_nan=True), key


# ---- MLat / MLT grids with the geomagnetic pole in view (resampleMLatMLT, mapping.py:1519-1547) ---------------------

@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_oracle_magnetic_pole_branch_equals_the_reference(mode):
    from oracle import ref_numpy as O
    z = load_golden('pole_frame_magnetic_%s.npz' % mode)
    hdr = header_from(z)
    g = O.georef_frame(hdr, 110.0, z['cam'], z['m_geo'], z['m_sm'], fast=mode == 'fast')
    corner_nan, center_nan = np.isnan(g['lat']), np.isnan(g['lat_c'])
    if mode == 'exact':
        corner_nan, center_nan = O.sanitize_masks(corner_nan, center_nan)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], corner_nan, 10)
    lat_c = np.where(center_mask, np.nan, g['mlat_c'])
    lon_c = np.where(center_mask, np.nan, O.mlt_to_sm_lon(g['mlt_c']))
    assert np.array_equal(lat_c, z['lat_c'], equal_nan=True) and np.array_equal(lon_c, z['lon_c'], equal_nan=True)
    outline = np.transpose([g['mlat'][~corner_mask], O.mlt_to_sm_lon(g['mlt'])[~corner_mask]])
    data = np.dstack((z['img'].astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    res = O.resample_mean(lat_c, lon_c, 110.0, data, outline, z['bbox'], tuple(z['ppd']), True, True)
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c'),
                 ('data', 'out_data')):
        assert np.array_equal(res[a], z[b], equal_nan=True), a


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_single_pass_plan_handles_the_magnetic_pole(mode):
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    z = load_golden('pole_frame_magnetic_%s.npz' % mode)
    hdr = header_from(z)
    t = parse(z['time_iso'])
    w, h = hdr['IMAGEW'], hdr['IMAGEH']
    out = {}
    for fuse in (True, False):
        pipe = FramePipeline(w, h, with_mag=True)
        res = pipe.run(hdr, 110, z['cam'], t, img=z['img'], fast=mode == 'fast', min_elevation=10, pxPerDeg=8, fuse=fuse,
                       magnetic=True)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        check_against_fixture(res, z)
        out[fuse] = res
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
        assert np.array_equal(out[True][k], out[False][k], equal_nan=True), k
    # the geodetic grid of the same frame has no pole in it: the plain single-pass plan
    pipe = FramePipeline(w, h, with_mag=True)
    res = pipe.run(hdr, 110, z['cam'], t, img=z['img'], fast=mode == 'fast', min_elevation=10, pxPerDeg=8, fuse=True)
    assert pipe.last_plan == 'single-pass' and not res['contains_pole']
    # and through the sequence loop, the same frame several times (hints carry the rotated box)
    seq = SequencePipeline(w, h, pxPerDeg=8, magnetic=True, fast=mode == 'fast')
    got = seq.process([(hdr, z['cam'], t, z['img'])] * 10, keep_on_device=False)
    assert seq.plans == ['single-pass'] * 10 and seq.hinted >= 1
    for r in got:
        for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
            assert np.array_equal(r[k], out[False][k], equal_nan=True), k


@pytest.mark.gpu
@pytest.mark.parametrize('side,mode', CASES)
def test_geodetic_pole_plan_with_mlat_arrays(side, mode):
    """A pipeline that also writes the MLat / MLT arrays (with_mag) and bins on the GEODETIC grid with the pole in view:
    the kernel variant of the magnetic pole plan with the rotated pair taken from (lat, lon) — single-pass, identical to
    the reference fixture and to the two-pass plan, the MLat / MLT arrays included."""
    from auromat_amd.pipeline import FramePipeline
    z = load_golden('pole_frame_%s_%s.npz' % (side, mode))
    hdr = header_from(z)
    t = parse(z['time_iso'])
    w, h = hdr['IMAGEW'], hdr['IMAGEH']
    pipe = FramePipeline(w, h, with_mag=True)
    two = pipe.run(hdr, 110, z['cam'], t, img=z['img'], fast=mode == 'fast', min_elevation=10, pxPerDeg=8, fuse=False)
    assert pipe.last_plan == 'two-pass'
    arrays_two = pipe.host_arrays()
    one = pipe.run(hdr, 110, z['cam'], t, img=z['img'], fast=mode == 'fast', min_elevation=10, pxPerDeg=8, fuse=True)
    assert pipe.last_plan == 'single-pass'
    check_against_fixture(one, z)
    arrays_one = pipe.host_arrays()
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt', 'mlat_c', 'mlt_c'):
        assert np.array_equal(arrays_one[k], arrays_two[k], equal_nan=True), k
    assert np.isfinite(arrays_one['mlat_c']).sum() > 2000
