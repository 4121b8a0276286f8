"""
GPU parity tests at frame level: awkward frame sizes, the directions-in variant, exact centres end to
end, uint8 images, full-size frames against the reference's strided samples + digests, and
size-independent properties at BASELINE.json's full size (pixel conservation, determinism, linearity
of the sums in the image, agreement of the fused pipeline with the mapping classes).
"""
import os
from datetime import datetime

import numpy as np
import numpy.ma as ma
import pytest

from conftest import header_from, load_golden

pytestmark = pytest.mark.gpu

TOL_DEG = 1e-6


def parse(s):
    return datetime.strptime(str(s), '%Y-%m-%dT%H:%M:%S.%f')


def nan_close(a, b, tol, max_mask_mismatch=0):
    mism = int((np.isnan(a) != np.isnan(b)).sum())
    assert mism <= max_mask_mismatch, 'NaN masks differ in %d places' % mism
    ok = ~np.isnan(a) & ~np.isnan(b)
    err = np.max(np.abs(a[ok] - b[ok]), initial=0.0)
    assert err <= tol, 'max abs error %.3e > %.1e' % (err, tol)


def oracle_frame(hdr, cam, t, fast, alt=110):
    from oracle import ref_numpy as O
    et = O.date2es(t)
    return O.georef_frame(hdr, alt, cam, O.mat_j2000_to_geo(et), O.mat_j2000_to_sm(et), fast=fast)


def oracle_resample(g, img, min_elev, ppd, alt=110):
    from oracle import ref_numpy as O
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), min_elev)
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    res = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), alt,
                          data, None, bbox, (ppd, ppd), disc, False)
    return res, bbox


@pytest.mark.parametrize('width,height,pointing,fast', [
    (253, 171, 'iss030', True),    # not a multiple of 4 / 63 / 16: scalar bin path, partial strips and chunks
    (61, 35, 'iss029', True),      # narrower than one strip
    (130, 97, 'iss030', False),    # exact centres
    (64, 16, 'iss030', True),      # exactly one tile of the LDS variant
])
def test_awkward_sizes_vs_oracle(width, height, pointing, fast):
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    hdr, cam, t = frame_header(width, height, pointing)
    img = frame_image(width, height, seed=5)
    pipe = FramePipeline(width, height, with_mag=True)
    res = pipe.run(hdr, 110, cam, t, img=img, fast=fast, min_elevation=10, pxPerDeg=7)
    got = pipe.host_arrays()
    g = oracle_frame(hdr, cam, t, fast)
    for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlat_c'):
        nan_close(got[k], g[k], TOL_DEG)
    for k in ('mlt', 'mlt_c'):
        nan_close(got[k], g[k], TOL_DEG * 24 / 360)
    if not fast:
        # the reference sanitises exact-mode mappings: centres lose pixels whose corners miss
        from oracle import ref_numpy as O
        cm, ce = O.sanitize_masks(np.isnan(g['lat']), np.isnan(g['lat_c']))
        g = dict(g, lat=np.where(cm, np.nan, g['lat']), lon=np.where(cm, np.nan, g['lon']),
                 lat_c=np.where(ce, np.nan, g['lat_c']), lon_c=np.where(ce, np.nan, g['lon_c']),
                 elev=np.where(ce, np.nan, g['elev']))
    want, bbox = oracle_resample(g, img, 10, 7)
    bb = pipe.bounding_box()
    np.testing.assert_allclose([bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast], bbox, rtol=0, atol=1e-9)
    assert want['data'].shape == res['mean'].shape
    from conftest import assert_counts_equal_up_to_edge_pixels
    from oracle import ref_numpy as O
    _, cmask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    assert_counts_equal_up_to_edge_pixels(want, res['count'], np.where(cmask, np.nan, g['lat_c']), g['lon_c'],
                                          'awkward size %dx%d' % (width, height))
    same = (want['count'] == res['count']) & (want['count'] > 0)
    assert np.array_equal(res['mean'][..., :3][same], want['data'][..., :3][same])


def test_directions_in_variant_matches_wcs_fused():
    """amt_georef_frame_dirs (caller-supplied corner directions) == amt_georef_frame on the same rays."""
    import ctypes as C
    from auromat_amd._native import Context, GeorefOut, to_host
    from auromat_amd.coordinates.wcs import pix2world
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.synthetic import frame_header
    w, h = 200, 150
    hdr, cam, t = frame_header(w, h)
    ctx = Context.current()
    p = frame_params(hdr, 110, cam, t, True)
    dirs = pix2world(hdr, w, h, corner=True, ascartesian=True, device=ctx.device)
    outs = []
    for use_dirs in (False, True):
        o = GeorefOut()
        bufs = dict(lat=ctx.empty((h + 1, w + 1)), lon=ctx.empty((h + 1, w + 1)), lat_c=ctx.empty((h, w)),
                    lon_c=ctx.empty((h, w)), elev=ctx.empty((h, w)))
        for k, v in bufs.items():
            setattr(o, k, v.data_ptr())
        if use_dirs:
            ctx.call('amt_georef_frame_dirs', C.byref(p), C.c_void_p(dirs.data_ptr()), C.byref(o))
        else:
            ctx.call('amt_georef_frame', C.byref(p), C.byref(o))
        outs.append({k: to_host(v) for k, v in bufs.items()})
    for k in outs[0]:
        nan_close(outs[0][k], outs[1][k], 1e-9)


def test_uint8_image_and_generic_mapping_roundtrip():
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.resample import resample
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 240, 160
    hdr, cam, t = frame_header(w, h)
    hdr = dict(hdr, **{'DATE-OBS': t.strftime('%Y-%m-%dT%H:%M:%S.%f'), 'POSX': cam[0], 'POSY': cam[1], 'POSZ': cam[2]})
    img = frame_image(w, h, seed=9, dtype=np.uint8)
    m = getMapping(img, hdr, altitude=110, fastCenterCalculation=True)
    assert m.photoTime == t and np.array_equal(m.cameraPosGCRS, cam)
    mm = m.maskedByElevation(10)
    r = resample(mm, pxPerDeg=5)
    assert r.img.dtype == np.uint8 and r.img.shape[2] == 3
    r.checkPlateCarree()
    r.checkGuarantees()
    g = oracle_frame(hdr, cam, t, True)
    want, _ = oracle_resample(g, img, 10, 5)
    mask = np.isnan(want['data'][..., 0])
    assert np.array_equal(ma.getmaskarray(r.img)[..., 0], mask)
    with np.errstate(invalid='ignore'):
        assert np.array_equal(r.img.data[~mask], np.round(want['data'][..., :3])[~mask].astype(np.uint8))
    # resampling the resampled mapping again at a coarser resolution works on the uploaded GenericMapping
    r2 = resample(r, pxPerDeg=2)
    r2.checkPlateCarree()
    assert r2.img.count() > 0


@pytest.mark.parametrize('name', ['georef_full_iss030_fast.npz', 'georef_full_iss030_exact.npz',
                                  'georef_full_iss029_fast.npz'])
def test_full_size_frames_vs_reference_samples(name):
    """Real 4256x2832 headers: every 32nd row/col and whole-array digests of the reference's arrays."""
    from auromat_amd.pipeline import FramePipeline
    z = load_golden(name)
    hdr = header_from(z)
    step = int(z['step'])
    pipe = FramePipeline(hdr['IMAGEW'], hdr['IMAGEH'], with_mag=True)
    pipe.georef(hdr, float(z['altitude']), z['cam'], parse(z['time_iso']), fast=name.endswith('fast.npz'),
                min_elevation=10)
    got = pipe.host_arrays()
    for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlat_c'):
        nan_close(got[k][::step, ::step], z[k], TOL_DEG)
    for k in ('mlt', 'mlt_c'):
        nan_close(got[k][::step, ::step], z[k], TOL_DEG * 24 / 360)
    for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev'):
        n, s, lo, hi, sabs = z['digest_' + k]
        a = got[k]
        ok = ~np.isnan(a)
        assert abs(int(ok.sum()) - int(n)) <= 2, k          # rays within rounding of tangency
        assert abs(a[ok].min() - lo) < TOL_DEG and abs(a[ok].max() - hi) < TOL_DEG
        assert abs(a[ok].sum() - s) <= 1e-9 * sabs + 400.0  # <= 2 extra/missing values of |x| <= 180


@pytest.fixture(scope='module')
def full_frame():
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 4240, 2832
    hdr, cam, t = frame_header(w, h)
    img = frame_image(w, h, seed=0)
    pipe = FramePipeline(w, h)
    res = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10)
    return dict(pipe=pipe, res=res, hdr=hdr, cam=cam, t=t, img=img)


def test_full_size_properties(full_frame):
    """BASELINE.json size (4240x2832): properties that do not need the oracle."""
    pipe, res, img = full_frame['pipe'], full_frame['res'], full_frame['img']
    a = pipe.host_arrays()
    # fast centres: valid <=> all four corners valid; elevation within [0, 90]
    cn = np.isnan(a['lat'])
    all4 = ~(cn[:-1, :-1] | cn[:-1, 1:] | cn[1:, :-1] | cn[1:, 1:])
    assert np.array_equal(~np.isnan(a['lat_c']), all4)
    assert np.array_equal(np.isnan(a['lat_c']), np.isnan(a['elev']))
    e = a['elev'][~np.isnan(a['elev'])]
    assert e.min() >= 0 and e.max() <= 90
    # conservation: every pixel above the threshold lands in exactly one cell of the (superset) grid,
    # except those in the discarded half-cell ring (reference resample.py:232-237)
    keep = a['elev'] >= 10
    g = res['grid']
    inside = keep & (a['lon_c'] >= g.xrange[0]) & (a['lon_c'] < g.xrange[1]) & \
        (a['lat_c'] >= g.yrange[0]) & (a['lat_c'] < g.yrange[1])
    assert int(res['count'].sum()) == int(inside.sum())
    assert 0 < inside.sum() <= keep.sum()
    # sum of sums: mean * count summed over cells == sum of the binned pixels (exact integers)
    filled = res['count'] > 0
    total = (res['mean'][..., :3][filled] * res['count'][filled][:, None]).sum(axis=0)
    want = img[inside].astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(total, want, rtol=1e-12)
    # bounding box equals the extremes of the corners adjacent to a kept pixel
    kc = np.zeros(cn.shape, bool)
    kc[:-1, :-1] |= keep
    kc[:-1, 1:] |= keep
    kc[1:, :-1] |= keep
    kc[1:, 1:] |= keep
    bb = pipe.bounding_box()
    assert bb.latSouth == a['lat'][kc].min() and bb.latNorth == a['lat'][kc].max()
    assert bb.lonWest == a['lon'][kc].min() and bb.lonEast == a['lon'][kc].max()


def test_full_size_determinism_and_linearity(full_frame):
    pipe, res = full_frame['pipe'], full_frame['res']
    again = pipe.run(full_frame['hdr'], 110, full_frame['cam'], full_frame['t'], fast=True, min_elevation=10, pxPerDeg=10)
    for k in ('mean', 'count', 'img', 'mask'):
        assert np.array_equal(again[k], res[k], equal_nan=True), k     # integer accumulation: bit reproducible
    half = (full_frame['img'] // 2).astype(np.uint16)
    r2 = pipe.run(full_frame['hdr'], 110, full_frame['cam'], full_frame['t'], img=half, fast=True, min_elevation=10,
                  pxPerDeg=10)
    r3 = pipe.run(full_frame['hdr'], 110, full_frame['cam'], full_frame['t'], img=(full_frame['img'] - half),
                  fast=True, min_elevation=10, pxPerDeg=10)
    filled = res['count'] > 0
    np.testing.assert_allclose((r2['mean'] + r3['mean'])[..., :3][filled], res['mean'][..., :3][filled], rtol=1e-14)
    pipe.set_image(full_frame['img'])


def test_mapping_classes_agree_with_fused_pipeline(full_frame):
    """getMapping(...).maskedByElevation(10) -> resample(): the lazy class path equals the fused pipeline."""
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.resample import resample
    m = ArraySpacecraftMapping(full_frame['hdr'], 110, full_frame['img'], full_frame['cam'], full_frame['t'], 'f',
                               fastCenterCalculation=True)
    mm = m.maskedByElevation(10)
    bb, pb = mm.boundingBox, full_frame['pipe'].bounding_box()
    assert (bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast) == (pb.latSouth, pb.lonWest, pb.latNorth, pb.lonEast)
    assert not mm.containsPole and not mm.containsDiscontinuity
    r = resample(mm, pxPerDeg=10)
    res = full_frame['res']
    assert np.array_equal(r.img.data, res['img']) and np.array_equal(ma.getmaskarray(r.img)[..., 0], res['mask'])
    assert np.array_equal(r.elevation.data, res['mean'][..., 3], equal_nan=True)
    assert np.array_equal(r.lats.data, res['lat']) and np.array_equal(r.lonsCenter.data, res['lon_c'])


def test_sequence_single_gpu_equals_frame_by_frame():
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.sequence import run_sequence
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 320, 214
    frames = []
    for k in range(3):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        frames.append((hdr, cam, t, frame_image(w, h, seed=seed)))
    out = run_sequence(frames, w, h, pxPerDeg=5)
    assert [f['index'] for f in out] == [0, 1, 2]
    pipe = FramePipeline(w, h)
    for f, (hdr, cam, t, img) in zip(out, frames):
        res = pipe.run(hdr, 110, cam, t, img=img, pxPerDeg=5)
        assert np.array_equal(f['mean'], res['mean'], equal_nan=True) and np.array_equal(f['count'], res['count'])
        assert f['lat0'] == res['grid'].latCenters[0] and f['lon0'] == res['grid'].lonCenters[0]


def test_pole_detection_camera_model_and_winding():
    """A camera above the pole: host projection test (fused path) and quad winding (generic path) agree."""
    from auromat_amd.mapping.astrometry import frame_params, pole_in_view
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.coordinates import transform as T
    from auromat_amd.resample import resample
    t = datetime(2012, 1, 25, 9, 26, 55)
    w, h = 96, 96
    # camera 400 km above the geographic north pole, looking straight down: boresight = -z_GEO in J2000
    m_geo = T.mat_j2000_to_geo(T.date2es(t))
    zen = m_geo.T.dot([0.0, 0.0, 1.0])
    cam = zen * (6356.75 + 400.0)
    bore = -zen
    ra = np.rad2deg(np.arctan2(bore[1], bore[0])) % 360
    dec = np.rad2deg(np.arcsin(bore[2]))
    hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0, 'CRVAL1': ra, 'CRVAL2': dec,
           'CRPIX1': w / 2 + 0.5, 'CRPIX2': h / 2 + 0.5, 'CD1_1': -0.5, 'CD1_2': 0.0, 'CD2_1': 0.0, 'CD2_2': 0.5,
           'IMAGEW': w, 'IMAGEH': h}
    p = frame_params(hdr, 110, cam, t, True)
    assert pole_in_view(p, 10.0) == 1
    m = ArraySpacecraftMapping(hdr, 110, np.full((h, w, 3), 1000, np.uint16), cam, t, 'pole', fastCenterCalculation=True)
    assert m.containsPole and m.boundingBox.latNorth == 90 and m.boundingBox.lonWest == -180
    r = resample(m, pxPerDeg=2)
    assert r.img.count() > 0 and np.all(r.img.compressed() == 1000)
    assert float(r.latsCenter.max()) > 89.0
    # the usual ISS frames do not contain a pole
    from auromat_amd.synthetic import frame_header
    hdr2, cam2, t2 = frame_header(128, 96)
    assert pole_in_view(frame_params(hdr2, 110, cam2, t2, True), 10.0) == 0


@pytest.mark.parametrize('width,height,pointing', [(256, 170, 'iss030'), (253, 171, 'iss029'), (4240, 2832, 'iss030')])
def test_single_pass_plan_equals_two_pass(width, height, pointing):
    """Binning fused into the georeferencing kernel (superset grid + crop) == separate binning kernel."""
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    hdr, cam, t = frame_header(width, height, pointing)
    img = frame_image(width, height, seed=11)
    pipe = FramePipeline(width, height)
    two = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=False)
    assert pipe.last_plan == 'two-pass'
    one = pipe.run(hdr, 110, cam, t, fast=True, min_elevation=10, pxPerDeg=10, fuse=True)
    assert pipe.last_plan == 'single-pass', 'single-pass plan was not taken'
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    # uint8 image and another resolution
    img8 = frame_image(width, height, seed=12, dtype=np.uint8)
    p8 = FramePipeline(width, height, img_dtype=np.uint8)
    a = p8.run(hdr, 110, cam, t, img=img8, min_elevation=15, pxPerDeg=(4, 7), fuse=False)
    b = p8.run(hdr, 110, cam, t, min_elevation=15, pxPerDeg=(4, 7), fuse=True)
    assert p8.last_plan == 'single-pass'
    for k in ('mean', 'count', 'img', 'mask'):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_single_pass_plan_with_camera_over_the_pole():
    """A camera straight above the pole: the single-pass plan takes the frame (pole plan, tests/test_pole_frames.py) and
    agrees with the two-pass plan."""
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.coordinates import transform as T
    t = datetime(2012, 1, 25, 9, 26, 55)
    w, h = 96, 96
    m_geo = T.mat_j2000_to_geo(T.date2es(t))
    zen = m_geo.T.dot([0.0, 0.0, 1.0])
    cam = zen * (6356.75 + 400.0)
    bore = -zen
    hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0,
           'CRVAL1': np.rad2deg(np.arctan2(bore[1], bore[0])) % 360, 'CRVAL2': np.rad2deg(np.arcsin(bore[2])),
           'CRPIX1': w / 2 + 0.5, 'CRPIX2': h / 2 + 0.5, 'CD1_1': -0.5, 'CD1_2': 0.0, 'CD2_1': 0.0, 'CD2_2': 0.5,
           'IMAGEW': w, 'IMAGEH': h}
    pipe = FramePipeline(w, h)
    img = np.full((h, w, 3), 777, np.uint16)
    res = pipe.run(hdr, 110, cam, t, img=img, min_elevation=10, pxPerDeg=2, fuse=True)
    assert pipe.last_plan == 'single-pass' and res['contains_pole']
    assert np.all(res['img'][~res['mask']] == 777)
    two = FramePipeline(w, h).run(hdr, 110, cam, t, img=img, min_elevation=10, pxPerDeg=2, fuse=False)
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
        assert np.array_equal(res[k], two[k], equal_nan=True), k


def test_sequence_pipeline_plans_agree_with_frame_by_frame():
    """Software-pipelined sequences (both plans, buffers reused every second frame) == one frame at a time."""
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 250, 168
    frames = []
    for k in range(7):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        frames.append((hdr, cam, t, frame_image(w, h, seed=seed)))
    ref_pipe = FramePipeline(w, h)
    ref = [ref_pipe.run(hdr, 110, cam, t, img=img, pxPerDeg=8) for hdr, cam, t, img in frames]
    for plan, streams in (('single-pass', True), ('two-pass', True), ('two-pass', False)):
        seq = SequencePipeline(w, h, pxPerDeg=8, plan=plan, bin_stream=streams)
        for rep in range(2):                       # a second call re-uses the buffers and the driver state
            out = seq.process(frames, keep_on_device=False)
            assert len(out) == len(frames)
            assert seq.plans == [plan] * len(frames)
            for a, b in zip(out, ref):
                for key in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
                    assert np.array_equal(a[key], b[key], equal_nan=True), (plan, key)
    assert SequencePipeline(w, h).process([]) == []
    one = SequencePipeline(w, h, pxPerDeg=8).process(frames[:1], keep_on_device=False)
    assert np.array_equal(one[0]['mean'], ref[0]['mean'], equal_nan=True)


def test_item_order_is_a_pure_scheduling_hint():
    """amt_georef_out.item_order changes the dispatch order of the work items, never the results."""
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 253, 171
    hdr, cam, t = frame_header(w, h, 'iss029')
    pipe = FramePipeline(w, h)
    img = frame_image(w, h, seed=3)
    base = None
    for order in (1, 0, 2, 7):                    # 0 and out of range: decided from the camera model
        pipe._out.item_order = order
        res = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=False)
        arrays = dict(pipe.host_arrays(), mean=res['mean'], count=res['count'])
        if base is None:
            base = arrays
        for k, v in arrays.items():
            assert np.array_equal(v, base[k], equal_nan=True), (order, k)
    pipe._out.item_order = 0


@pytest.mark.parametrize('altitude', [100, 110, 120])
def test_config4_mlat_mlt_three_shells_full_size(altitude):
    """
    BASELINE.json configs[3]: geomagnetic transform + MLat/MLT resample on three altitude shells at full size, against
    the oracle over the WHOLE frame: all nine per-pixel arrays (lat, lon, MLat, MLT of corners and centres, elevation)
    within 1e-6 deg with identical NaN patterns, and the resampled (MLat, SM longitude) grid of every shell — the
    reference's resampleMLatMLT (resample.py:63-71 -> _resample on (mlat, mltToSmLon(mlt))) — cell for cell: mask,
    integer channel means, elevation means to 1e-9 deg, grid coordinates.
    """
    from oracle import ref_numpy as O
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.resample import grid_coordinates, resampleMLatMLT
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 4240, 2832
    hdr, cam, t = frame_header(w, h, 'iss030')
    img = frame_image(w, h, seed=altitude)
    pipe = FramePipeline(w, h, with_mag=True)
    res = pipe.run(hdr, altitude, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, magnetic=True, fuse=True)
    assert pipe.last_plan == 'single-pass'
    got = pipe.host_arrays()
    g = oracle_frame(hdr, cam, t, True, alt=altitude)
    assert int((~np.isnan(g['lat'])).sum()) > 5000000
    for k in ('lat', 'lon', 'mlat', 'lat_c', 'lon_c', 'elev', 'mlat_c'):
        nan_close(got[k], g[k], TOL_DEG)
    for k in ('mlt', 'mlt_c'):
        nan_close(got[k], g[k], TOL_DEG * 24 / 360)
    # the oracle's resampleMLatMLT: the box over the corners that survive maskedByElevation, in (MLat, SM longitude)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    bbox, disc = O.bbox_of_corners(g['mlat'], O.mlt_to_sm_lon(g['mlt']), corner_mask)
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    outline = np.transpose([g['mlat'][~corner_mask], O.mlt_to_sm_lon(g['mlt'])[~corner_mask]])
    want = O.resample_mean(np.where(center_mask, np.nan, g['mlat_c']), np.where(center_mask, np.nan, O.mlt_to_sm_lon(g['mlt_c'])),
                           altitude, data, outline, bbox, (10, 10), disc, False)
    assert res['mean'].shape == want['data'].shape, (res['mean'].shape, want['data'].shape)
    mask = np.isnan(want['data'][..., 0])
    assert np.array_equal(res['mask'], mask) and (~mask).sum() > 2000
    ok = ~mask
    assert np.array_equal(res['mean'][..., :3][ok], want['data'][..., :3][ok])           # exact integer sums / counts
    assert np.max(np.abs(res['mean'][..., 3][ok] - want['data'][..., 3][ok])) < 1e-9    # elevation, fixed-point sums
    c = grid_coordinates(res)
    for a, b in (('lat', 'lat'), ('lon', 'lon'), ('lat_c', 'lat_c'), ('lon_c', 'lon_c')):
        assert np.array_equal(c[a], want[b]), a
    assert pipe.ctx.last_variant()[0] == 1                   # nine arrays: the (lat, lon) + (MLat, MLT) variant
    # the two-pass plan gives the same bits
    two = pipe.run(hdr, altitude, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, magnetic=True, fuse=False)
    for k in ('mean', 'count', 'img', 'mask'):
        assert np.array_equal(two[k], res[k], equal_nan=True), k
    # MLat / MLT-only mode (what resampleMLatMLT consumes, reference resample.py:63-71 + mapping.py:1519-1547; the kernel
    # skips the geodetic half of its arithmetic and its four stores): the same grid and the same five arrays, bit for bit
    lean = FramePipeline(w, h, with_mag=True, with_geo=False)
    res5 = lean.run(hdr, altitude, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, magnetic=True, fuse=True)
    assert lean.last_plan == 'single-pass' and lean.ctx.last_variant()[0] == 4 and lean.fd.lat is None
    for k in ('mean', 'count', 'img', 'mask'):
        assert np.array_equal(res5[k], res[k], equal_nan=True), k
    kept = lean.host_arrays(kept_only=True)
    assert sorted(kept) == ['elev', 'mlat', 'mlat_c', 'mlt', 'mlt_c'] and lean.fd.lat is None
    for k, v in kept.items():
        assert np.array_equal(v, got[k], equal_nan=True), k
    if altitude == 120:
        # ... and a caller that does ask such a pipeline for the geodetic arrays gets them (computed on demand)
        full = lean.host_arrays()
        for k, v in full.items():
            assert np.array_equal(v, got[k], equal_nan=True), k
        again = lean.run(hdr, altitude, cam, t, fast=True, min_elevation=10, pxPerDeg=10, magnetic=True, fuse=True)
        assert lean.ctx.last_variant()[0] == 4 and np.array_equal(again['mean'], res['mean'], equal_nan=True)
    del lean
    if altitude == 110:
        # the reference's own call sequence (resample.py:63-71) gives the same grid
        m = ArraySpacecraftMapping(hdr, altitude, img, cam, t, 'c4', fastCenterCalculation=True).maskedByElevation(10)
        r = resampleMLatMLT(m, pxPerDeg=10)
        assert np.array_equal(r.img.data, res['img']) and np.array_equal(ma.getmaskarray(r.img)[..., 0], res['mask'])


@pytest.mark.parametrize('width,height,pointing,altitude', [(256, 170, 'iss030', 110), (253, 171, 'iss029', 110),
                                                             (4240, 2832, 'iss030', 100), (4240, 2832, 'iss030', 120)])
def test_single_pass_magnetic_equals_two_pass(width, height, pointing, altitude):
    """resampleMLatMLT through the frame driver (MLat / SM-longitude grid fused into the kernel) == two-pass plan."""
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    hdr, cam, t = frame_header(width, height, pointing)
    img = frame_image(width, height, seed=21)
    pipe = FramePipeline(width, height, with_mag=True)
    two = pipe.run(hdr, altitude, cam, t, img=img, min_elevation=10, pxPerDeg=10, magnetic=True, fuse=False)
    arrays_two = pipe.host_arrays()
    one = pipe.run(hdr, altitude, cam, t, min_elevation=10, pxPerDeg=10, magnetic=True, fuse=True)
    assert pipe.last_plan == 'single-pass'        # the southern frame straddles SM longitude 180: shifted binning
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    for k, v in pipe.host_arrays().items():
        assert np.array_equal(v, arrays_two[k], equal_nan=True), k
    # the geodetic box is still available after a magnetic single-pass launch, and equals the geodetic plan's
    geo = FramePipeline(width, height)
    geo.run(hdr, altitude, cam, t, img=img, min_elevation=10, pxPerDeg=10, fuse=True)
    a, b = pipe.bounding_box(), geo.bounding_box()
    assert (a.latSouth, a.lonWest, a.latNorth, a.lonEast) == (b.latSouth, b.lonWest, b.latNorth, b.lonEast)
    if width < 1000:
        seq = SequencePipeline(width, height, altitude=altitude, pxPerDeg=10, magnetic=True)
        out = seq.process([(hdr, cam, t, img)] * 3, keep_on_device=False)
        for r in out:
            assert np.array_equal(r['mean'], two['mean'], equal_nan=True) and np.array_equal(r['count'], two['count'])


@pytest.mark.parametrize('width,height', [(253, 171), (4240, 2832)])
def test_single_pass_plan_across_the_dateline(width, height):
    """A frame that straddles the 180 deg discontinuity is binned with shifted longitudes (resample.py:203-218)."""
    from datetime import timedelta
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    hdr, cam, t = frame_header(width, height, 'iss029')        # longitudes 142 .. 168 E at the fixture's time
    t = t - timedelta(minutes=80)                               # same inertial geometry, the Earth 20 deg further west
    img = frame_image(width, height, seed=31)
    pipe = FramePipeline(width, height)
    two = pipe.run(hdr, 110, cam, t, img=img, min_elevation=10, pxPerDeg=10, fuse=False)
    bb = pipe.bounding_box()
    assert bb.containsDiscontinuity and not bb.containsPole and two['contains_discontinuity']
    one = pipe.run(hdr, 110, cam, t, min_elevation=10, pxPerDeg=10, fuse=True)
    assert pipe.last_plan == 'single-pass' and one['contains_discontinuity']
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    assert one['lon_c'].min() < -170 and one['lon_c'].max() > 170


@pytest.mark.parametrize('magnetic', [False, True])
def test_box_first_plan_across_the_dateline_and_with_a_pole(magnetic):
    """arcsecPerPx on the frame pipeline: a date-line frame takes the box-first plan (its box is (smallest positive,
    largest non-positive) longitude) and equals the two-pass plan at the same px/deg; with a pole in view the reference's
    plateCarreeResolution has no longitude resolution (resample.py:47-61: the box goes all around), here as there."""
    from datetime import timedelta
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.resample import plateCarreeResolution
    from auromat_amd.synthetic import frame_header, frame_image, pole_frame
    w, h = 253, 171
    hdr, cam, t = frame_header(w, h, 'iss029')
    if not magnetic:
        t = t - timedelta(minutes=80)              # geodetic longitudes across +-180 (the SM longitudes cross as they are)
    img = frame_image(w, h, seed=32)
    pipe = FramePipeline(w, h, with_mag=magnetic)
    one = pipe.run(hdr, 110, cam, t, img=img, min_elevation=10, arcsecPerPx=400, magnetic=magnetic, fuse=True)
    assert pipe.last_plan == 'single-pass' and one['contains_discontinuity']
    assert one['pxPerDeg'][0] == 9.0 and 2 < one['pxPerDeg'][1] < 9
    two = pipe.run(hdr, 110, cam, t, min_elevation=10, pxPerDeg=one['pxPerDeg'], magnetic=magnetic, fuse=False)
    assert pipe.last_plan == 'two-pass'
    if not magnetic:
        assert one['pxPerDeg'] == plateCarreeResolution(pipe.bounding_box(), 400)
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon', 'lat_c', 'lon_c'):
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    also = pipe.run(hdr, 110, cam, t, min_elevation=10, arcsecPerPx=400, magnetic=magnetic, fuse=False)
    assert pipe.last_plan == 'two-pass' and also['pxPerDeg'] == one['pxPerDeg']
    assert np.array_equal(also['mean'], one['mean'], equal_nan=True)
    if not magnetic:
        p_hdr, p_cam, p_t = pole_frame(w, h)
        with pytest.raises(AssertionError):
            pipe.run(p_hdr, 110, p_cam, p_t, img=img, min_elevation=10, arcsecPerPx=400, fuse=True)
    # no valid pixel: the reference's ValueError
    up = np.asarray(cam, dtype=np.float64) / np.linalg.norm(cam)                                # looking at the zenith
    z_hdr = dict(hdr, CRVAL1=float(np.rad2deg(np.arctan2(up[1], up[0])) % 360), CRVAL2=float(np.rad2deg(np.arcsin(up[2]))))
    with pytest.raises(ValueError):
        pipe.run(z_hdr, 110, cam, t, img=img, min_elevation=10, arcsecPerPx=400, magnetic=magnetic, fuse=True)


def test_a_frame_handed_back_because_of_a_poor_estimate_is_relaunched_with_its_exact_box():
    """A single-pass launch whose superset grid (laid out from an estimate of the box: here a deliberately wrong one) does not
    hold the exact grid hands the frame back (status 1).  It is launched once more with the exact box as the estimate — no
    per-pixel array is allocated for it in a grids-only pipeline — and gives the grid of an ordinary launch."""
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 400, 270
    hdr, cam, t = frame_header(w, h, 'iss030')
    img = frame_image(w, h, seed=77)
    for magnetic in (False, True):
        pipe = FramePipeline(w, h, alloc_coords=False, with_mag=magnetic)
        ref = pipe.run(hdr, 110, cam, t, img=img, min_elevation=10, pxPerDeg=10, magnetic=magnetic, fuse=True)
        assert pipe.last_plan == 'single-pass' and not pipe._fused.get('retried')
        box = list(pipe._fused['result'].bbox)
        bad = [box[0] + 7, box[1] + 7, box[2] - 9, box[3] - 9, box[4], box[5], 1.0, 0.0]
        p = frame_params(hdr, 110, cam, t, True, magnetic=magnetic)
        pipe.start_coarse(p, 10, magnetic, hint=bad)
        pipe.georef(hdr, 110, cam, t, True, 10, params=p, fuse_pxPerDeg=(10, 10), coarse_started=True, fuse_magnetic=magnetic)
        res = pipe.resample(10, magnetic=magnetic)
        assert pipe.last_plan == 'single-pass' and pipe._fused.get('retried') and pipe.fd.lat is None
        for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
            assert np.array_equal(res[k], ref[k], equal_nan=True), k


def test_staged_copies_between_pageable_memory_and_the_device():
    """amt_upload_staged / amt_download_staged (worker threads, page-locked pieces): odd sizes, sizes below and above one
    piece and above one round of all workers' pieces, repeated use of the staging buffers; to_device / to_host on top of
    them, the bound on page-locked result memory included."""
    import ctypes as C
    import torch
    import auromat_amd._native as N
    ctx = N.Context.current()
    rs = np.random.RandomState(4)
    for nbytes in (1, 4097, (4 << 20) - 1, 4 << 20, (4 << 20) + 5, (37 << 20) + 3, 72 * 1000 * 1000):
        src = rs.randint(0, 256, nbytes).astype(np.uint8)
        dev = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
        ctx.call('amt_upload_staged', C.c_void_p(dev.data_ptr()), C.c_void_p(src.ctypes.data), nbytes)
        assert np.array_equal(dev.cpu().numpy(), src)
        back = np.zeros(nbytes, dtype=np.uint8)
        ctx.call('amt_download_staged', C.c_void_p(back.ctypes.data), C.c_void_p(dev.data_ptr()), nbytes)
        assert np.array_equal(back, src)
    a = rs.rand(1200, 1300)
    t = ctx.to_device(a)
    assert np.array_equal(N.to_host(t), a)
    m = rs.rand(1500, 1400) > 0.5
    assert ctx.to_device(m, np.bool_).dtype == torch.bool and ctx.to_device(m[:10, :10], np.bool_).dtype == torch.bool
    assert np.array_equal(N.to_host(ctx.to_device(m, np.bool_)).astype(bool), m)
    # beyond the limit of page-locked result memory the arrays are pageable ones, equal all the same
    limit = N._PINNED_RESULT_LIMIT
    try:
        N._PINNED_RESULT_LIMIT = 0
        b = N.to_host(t)
        assert np.array_equal(b, a) and b.flags.owndata
        i16 = torch.arange(-3000000, 3000000, dtype=torch.int64, device='cuda').to(torch.int16)
        assert np.array_equal(N.to_host(i16, dtype=np.uint16), i16.cpu().numpy().view(np.uint16))
    finally:
        N._PINNED_RESULT_LIMIT = limit
    held = N._pinned_out['bytes']
    c = N.to_host(t)
    assert N._pinned_out['bytes'] == held + a.nbytes
    del c
    import gc
    gc.collect()
    assert N._pinned_out['bytes'] == held


def test_batched_launch_and_sequence_hints_change_nothing():
    """Two frames per launch of the big kernel (amt_pipe_launch_many) and bounding-box hints from the previous
    frame instead of the coarse pre-pass give the same bits as one frame at a time with pre-passes."""
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 250, 168
    frames = []
    for k in range(7):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        frames.append((hdr, cam, t, frame_image(w, h, seed=seed)))
    base = SequencePipeline(w, h, pxPerDeg=8, batch=1)
    base.use_hints = False
    ref = base.process(frames, keep_on_device=False)
    assert base.hinted == 0 and base.plans == ['single-pass'] * 7
    for batch, hints in ((2, False), (2, True), (1, True), (3, True)):
        seq = SequencePipeline(w, h, pxPerDeg=8, batch=batch)
        seq.use_hints = hints
        out = seq.process(frames, keep_on_device=False)
        assert seq.plans == ['single-pass'] * 7
        if batch < 3:                          # with 3 per launch all 7 frames are prepared before the first finishes
            assert (seq.hinted > 0) == hints
        for a, b in zip(out, ref):
            for key in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
                assert np.array_equal(a[key], b[key], equal_nan=True), (batch, hints, key)
    # a jump in the sequence (other pointing) is not a neighbour: that frame gets a real pre-pass again
    from auromat_amd.synthetic import frame_header
    more = []
    for k in range(7, 12):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        more.append((hdr, cam, t, frame_image(w, h, seed=seed)))
    hdr2, cam2, t2 = frame_header(w, h, 'iss029')
    mixed = frames + [(hdr2, cam2, t2, frames[0][3])] + more
    seq = SequencePipeline(w, h, pxPerDeg=8)
    out = seq.process(mixed, keep_on_device=False)
    assert seq.plans == ['single-pass'] * len(mixed) and 0 < seq.hinted < len(mixed) - 4
    plain = SequencePipeline(w, h, pxPerDeg=8, batch=1)
    plain.use_hints = False
    want = plain.process(mixed, keep_on_device=False)
    for a, b in zip(out, want):
        assert np.array_equal(a['mean'], b['mean'], equal_nan=True) and np.array_equal(a['count'], b['count'])


def test_sequence_with_a_pole_frame_uint8_and_magnetic():
    """Batches whose frames take different plans (a pole frame between ordinary ones), uint8 images, MLat/MLT grids."""
    from auromat_amd.coordinates import transform as T
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 96, 96
    frames = []
    for k in range(6):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        frames.append((hdr, cam, t, frame_image(w, h, seed=seed, dtype=np.uint8)))
    t = datetime(2012, 1, 25, 9, 26, 55)
    zen = T.mat_j2000_to_geo(T.date2es(t)).T.dot([0.0, 0.0, 1.0])
    bore = -zen
    pole_hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0,
                'CRVAL1': np.rad2deg(np.arctan2(bore[1], bore[0])) % 360, 'CRVAL2': np.rad2deg(np.arcsin(bore[2])),
                'CRPIX1': w / 2 + 0.5, 'CRPIX2': h / 2 + 0.5, 'CD1_1': -0.5, 'CD1_2': 0.0, 'CD2_1': 0.0, 'CD2_2': 0.5,
                'IMAGEW': w, 'IMAGEH': h}
    mixed = frames[:3] + [(pole_hdr, zen * (6356.75 + 400.0), t, frames[0][3])] + frames[3:]
    ref_pipe = FramePipeline(w, h, img_dtype=np.uint8)
    ref = [ref_pipe.run(hd, 110, cam, tt, img=img, pxPerDeg=3) for hd, cam, tt, img in mixed]
    for batch in (1, 2, 3):
        seq = SequencePipeline(w, h, img_dtype=np.uint8, pxPerDeg=3, batch=batch)
        out = seq.process(mixed, keep_on_device=False)
        assert seq.plans == ['single-pass'] * 7         # the pole frame too (pole plan of the fused kernel)
        assert out[3]['contains_pole']
        for a, b in zip(out, ref):
            for key in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
                assert np.array_equal(a[key], b[key], equal_nan=True), (batch, key)
            assert a['img'].dtype == np.uint8
    # MLat/MLT grids through the sequence loop, hints included
    mag_ref_pipe = FramePipeline(w, h, img_dtype=np.uint8, with_mag=True)
    mag_ref = [mag_ref_pipe.run(hd, 110, cam, tt, img=img, pxPerDeg=3, magnetic=True) for hd, cam, tt, img in frames]
    seq = SequencePipeline(w, h, img_dtype=np.uint8, pxPerDeg=3, magnetic=True)
    out = seq.process(frames * 2, keep_on_device=False)
    assert seq.hinted > 0
    for a, b in zip(out, mag_ref * 2):
        assert np.array_equal(a['mean'], b['mean'], equal_nan=True) and np.array_equal(a['count'], b['count'])


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_gather_over_rccl_single_rank():
    """run_sequence() with an initialised "nccl" (= RCCL) process group of one rank: the collectives of the gather
    (all_gather of sizes, padded gather of [descriptors | payload]) run on device tensors."""
    import torch
    import torch.distributed as dist
    from auromat_amd.sequence import run_sequence
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 200, 140
    frames = []
    for k in range(4):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        frames.append((hdr, cam, t, frame_image(w, h, seed=seed)))
    plain = run_sequence(frames, w, h, pxPerDeg=6)            # no process group: local packing only
    store = dist.TCPStore('127.0.0.1', _free_port(), 1, True)
    dist.init_process_group('nccl', store=store, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        got = run_sequence(frames, w, h, pxPerDeg=6)
    finally:
        dist.destroy_process_group()
    assert [f['index'] for f in got] == [0, 1, 2, 3]
    for a, b in zip(got, plain):
        assert np.array_equal(a['mean'], b['mean'], equal_nan=True) and np.array_equal(a['count'], b['count'])
        assert (a['lat0'], a['lon0'], a['dlat'], a['dlon']) == (b['lat0'], b['lon0'], b['dlat'], b['dlon'])


def test_direction_array_mapping_for_other_camera_models():
    """A camera model supplied as corner direction vectors (all-sky calibrations, SIP / non-TAN WCS)
    goes through the same kernel; with the TAN model's own directions it reproduces the WCS mapping."""
    from auromat_amd.mapping.astrometry import DirectionArrayMapping, pixelDirection
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.resample import resample
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 257, 173
    hdr, cam, t = frame_header(w, h, 'iss030')
    img = frame_image(w, h, seed=4)
    dirs = pixelDirection(hdr, corner=True)
    assert dirs.shape == (h + 1, w + 1, 3)
    a = DirectionArrayMapping(dirs, 110, img, cam, t, 'dirs')
    b = ArraySpacecraftMapping(hdr, 110, img, cam, t, 'wcs', fastCenterCalculation=True)
    a.checkGuarantees()
    for name in ('lats', 'lons', 'latsCenter', 'lonsCenter', 'elevation'):
        x, y = getattr(a, name), getattr(b, name)
        assert np.array_equal(ma.getmaskarray(x), ma.getmaskarray(y)), name
        assert np.max(np.abs(x.compressed() - y.compressed())) < 1e-9, name
    (mla, mlt), (mlb, mltb) = a.mLatMlt, b.mLatMlt
    assert np.max(np.abs(mla.compressed() - mlb.compressed())) < 1e-9
    assert np.max(np.abs(mlt.compressed() - mltb.compressed())) < 1e-9
    ra, rb = resample(a.maskedByElevation(10), pxPerDeg=10), resample(b.maskedByElevation(10), pxPerDeg=10)
    assert ra.img.shape == rb.img.shape
    assert (ma.getmaskarray(ra.img) != ma.getmaskarray(rb.img)).sum() <= 6
    # a different camera: every direction tilted by a fixed rotation is still a valid model (no WCS involved)
    th = np.deg2rad(0.3)
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    c = DirectionArrayMapping(dirs.dot(rot.T), 110, img, cam, t, 'tilted')
    c.checkGuarantees()
    assert np.nanmax(np.abs(c.lons.filled(np.nan) - b.lons.filled(np.nan))) > 0.01


@pytest.mark.parametrize('pointing', ['iss030', 'iss029', 'pole', 'south-pole'])
def test_direction_array_single_pass_equals_the_array_route(pointing):
    """The single-pass plan on a direction array (amt_pipe_launch_dirs; north_star's "(H+1) x (W+1) corner arrays" form,
    reference astrometry.py:49-64,86-106 + resample.py:301-351): resample() / resampleMLatMLT() of a DirectionArrayMapping
    whose arrays nobody has asked for run ONE kernel and give the bits of the array route (georeference into arrays, mask,
    box, separate binning pass).  A frame with a pole in view has no camera model to say so: its box comes within the guard
    band and the corner quads decide (two-pass plan)."""
    import auromat_amd.resample as R
    from auromat_amd.mapping.astrometry import DirectionArrayMapping, pixelDirection
    from auromat_amd.synthetic import frame_header, frame_image, pole_frame
    w, h = 640, 420
    if pointing in ('pole', 'south-pole'):
        hdr, cam, t = pole_frame(w, h, south=pointing == 'south-pole')
    else:
        hdr, cam, t = frame_header(w, h, pointing)
    img = frame_image(w, h, seed=11)
    dirs = pixelDirection(hdr, corner=True)
    for magnetic in (False, True):
        fn = R.resampleMLatMLT if magnetic else R.resample
        fused = DirectionArrayMapping(dirs, 110, img, cam, t, 'fused').maskedByElevation(10)
        got = fn(fused, pxPerDeg=10)
        plan = R.last_plan
        arrays = DirectionArrayMapping(dirs, 110, img, cam, t, 'arrays')
        arrays.lats                                     # materialise: the array route from here on
        want = fn(arrays.maskedByElevation(10), pxPerDeg=10)
        if pointing in ('iss030', 'iss029'):
            assert plan == 'single-pass', (plan, magnetic)
        elif not magnetic:
            assert plan == 'two-pass', plan             # the geographic pole is in view: decided from the corner quads
        assert fused._frame is None or plan != 'single-pass'
        for name in ('lats', 'lons', 'latsCenter', 'lonsCenter', 'elevation', 'img'):
            a, b = getattr(got, name), getattr(want, name)
            assert a.shape == b.shape, (name, a.shape, b.shape)
            assert np.array_equal(ma.getmaskarray(a), ma.getmaskarray(b)), (name, magnetic)
            if name == 'elevation':
                assert np.max(np.abs(a.compressed() - b.compressed())) < 1e-9
            else:
                assert np.array_equal(a.compressed(), b.compressed()), (name, magnetic)


def test_direction_array_single_pass_full_size_equals_the_camera_model():
    """BASELINE configs[2] in its directions-in form at full size: FramePipeline.run(dirs=pix2world(...)) against the same
    frame through the TAN model — the direction arrays differ from the in-kernel affine model by rounding, so the grids are
    compared cell by cell: same shape, same mask, counts equal but for pixels within rounding of a bin edge."""
    import torch
    from auromat_amd.coordinates.wcs import pix2world
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 4240, 2832
    hdr, cam, t = frame_header(w, h, 'iss030')
    img = frame_image(w, h, seed=5)
    pipe = FramePipeline(w, h, alloc_coords=False)
    dirs = pix2world(hdr, w, h, corner=True, ascartesian=True, device=pipe.ctx.device)
    p = frame_params(hdr, 110, cam, t, True, magnetic=False)
    a = pipe.run(None, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=True, params=p, dirs=dirs)
    assert pipe.last_plan == 'single-pass' and pipe.ctx.last_variant()[1] == 2
    b = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=True)
    assert pipe.last_plan == 'single-pass'
    assert a['mean'].shape == b['mean'].shape and np.array_equal(a['mask'], b['mask'])
    assert int((a['count'] != b['count']).sum()) <= 4
    same = a['count'] == b['count']
    assert np.array_equal(a['mean'][..., :3][same], b['mean'][..., :3][same], equal_nan=True)
    # ... and the two-pass plan on the same directions gives the single-pass bits
    c = pipe.run(None, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=False, params=p, dirs=dirs)
    assert pipe.last_plan == 'two-pass'
    for k in ('mean', 'count', 'img', 'mask'):
        assert np.array_equal(a[k], c[k], equal_nan=True), k
    del dirs
    torch.cuda.empty_cache()


@pytest.mark.parametrize('padded', [False, True])
def test_direction_arrays_three_frames_per_launch_equal_single_launches(padded):
    """amt_pipe_launch_dirs_many (FramePipeline.georef_many(dirs=...), the form bench.py's directions_in leg runs): three
    DIFFERENT frames and images in one launch of k_georef_rows<DIRS_IN, BIN> against one run(dirs=...) per frame — every
    grid and every kept coordinate array bit for bit, in both row layouts; then the entry point's argument checks."""
    import ctypes as C
    import torch
    from auromat_amd._native import NativeError
    from auromat_amd.coordinates.wcs import pix2world
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 1061, 707
    frames, imgs = [], []
    for k in range(3):
        hdr, cam, t, _ = sequence_frame(40 + 9 * k, w, h)
        imgs.append(frame_image(w, h, seed=20 + k))
        frames.append((hdr, cam, t))
    pipes = [FramePipeline(w, h, padded=padded) for _ in range(3)]
    ctx = pipes[0].ctx
    ps = [frame_params(hdr, 110, cam, t, True, magnetic=False) for hdr, cam, t in frames]
    ds = [pix2world(hdr, w, h, corner=True, ascartesian=True, device=ctx.device) for hdr, cam, t in frames]
    single = []
    one = FramePipeline(w, h, padded=False)
    for (hdr, cam, t), p, d, img in zip(frames, ps, ds, imgs):
        r = one.run(None, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=9, fuse=True, params=p, dirs=d)
        assert one.last_plan == 'single-pass'
        r['coords'] = {k: one.fd.host(k).copy() for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev')}
        single.append(r)
    for q, p, d, img in zip(pipes, ps, ds, imgs):
        q.use_image(torch.from_numpy(np.ascontiguousarray(img).view(np.int16)).to(ctx.device))
        q.start_coarse(p, 10, False, dirs=d)
    FramePipeline.georef_many(pipes, ps, 110, 10, (9, 9), False, dirs=ds, pole_in_view=0)
    assert ctx.last_variant()[1] == 2
    ready = [q.fused_ready((9, 9), False) for q in pipes]
    assert all(r is not None for r in ready)
    out = FramePipeline.finalize_many(pipes, ready, (9, 9), False)
    for q, a, b in zip(pipes, out, single):
        assert q.last_plan == 'single-pass'
        for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        for k, v in b['coords'].items():
            assert np.array_equal(q.fd.host(k), v, equal_nan=True), k
    # argument checks: more frames than a launch holds, a driver twice, a NULL direction array
    lib = ctx._lib
    n = 3
    handles = (C.c_void_p * n)(*[q._pipe() for q in pipes])
    pp = (C.c_void_p * n)(*[C.addressof(p) for p in ps])
    oo = (C.c_void_p * n)(*[C.addressof(q._out) for q in pipes])
    ii = (C.c_void_p * n)(*[q.fd.img.data_ptr() for q in pipes])
    dd = (C.c_void_p * n)(*[d.data_ptr() for d in ds])
    code = pipes[0].fd.img_dtype_code
    assert lib.amt_pipe_launch_dirs_many(handles, 99, pp, dd, oo, ii, code, 10.0, 9.0, 9.0, 0, 0) < 0
    assert b'AMT_PIPE_MAX_BATCH' in lib.amt_last_error(ctx.handle)
    twice = (C.c_void_p * n)(handles[0], handles[1], handles[0])
    assert lib.amt_pipe_launch_dirs_many(twice, n, pp, dd, oo, ii, code, 10.0, 9.0, 9.0, 0, 0) < 0
    hole = (C.c_void_p * n)(dd[0], None, dd[2])
    assert lib.amt_pipe_launch_dirs_many(handles, n, pp, hole, oo, ii, code, 10.0, 9.0, 9.0, 0, 0) < 0
    assert lib.amt_pipe_launch_dirs_many(handles, n, pp, None, oo, ii, code, 10.0, 9.0, 9.0, 0, 0) < 0
    for q in pipes:
        q.join()
    ctx.synchronize()


@pytest.mark.parametrize('proj', ['TAN', 'SIN', 'ARC', 'STG', 'ZEA'])
def test_device_generator_for_zenithal_headers_equals_the_numpy_restatement(proj):
    """amt_directions_zenithal (what getMapping runs for a header that is not plain TAN — the reference hands those to
    astropy.wcs, wcs.py:54-56) against coordinates.wcs.zenithal_pix2world, the NumPy restatement the CPU tests pin to the
    projections' laws: corners and centres, a sub-rectangle, with and without SIP polynomials of order 3, LONPOLE != 180."""
    from auromat_amd._native import to_host
    from auromat_amd.coordinates.wcs import zenithal_directions_device, zenithal_pix2world
    from auromat_amd.synthetic import frame_header
    w, h = 300, 210
    hdr, cam, t = frame_header(w, h, 'iss029')
    for sip in (False, True):
        hd = dict(hdr, CTYPE1='RA---' + proj + ('-SIP' if sip else ''), CTYPE2='DEC--' + proj + ('-SIP' if sip else ''), LONPOLE=173.0)
        if sip:
            hd.update(A_ORDER=3, B_ORDER=2, A_2_0=2e-5, A_1_1=-7e-6, A_0_3=3e-8, A_3_0=-2e-8, B_0_2=-1e-5, B_1_1=4e-6, B_2_0=9e-6)
        for corner, sx, sy, ww, hh in ((True, 0, 0, w, h), (False, 0, 0, w, h), (True, 17, 5, 120, 90)):
            want = zenithal_pix2world(hd, ww, hh, startX=sx, startY=sy, corner=corner)
            got = to_host(zenithal_directions_device(hd, ww, hh, startX=sx, startY=sy, corner=corner))
            assert got.shape == want.shape
            assert np.array_equal(np.isnan(got), np.isnan(want))
            # (rounding level; SIN's arccos and ZEA's arcsin amplify it towards the rim of the projection, which this wide
            # frame reaches: 3 % of the SIN frame lies beyond it and is NaN on both sides)
            assert np.nanmax(np.abs(got - want)) < (1e-12 if proj in ('SIN', 'ZEA') else 1e-14), (proj, sip, corner, np.nanmax(np.abs(got - want)))
            n = np.sqrt((got ** 2).sum(axis=-1))
            assert np.nanmax(np.abs(n - 1)) < 1e-15


def test_getMapping_with_a_non_tan_header():
    """A header with another zenithal projection (reference wcs.py:54-56: astropy.wcs) or SIP terms: getMapping builds the
    corner directions on the host (coordinates.wcs.zenithal_pix2world) and hands them to the directions-in kernel — the
    oracle's chain (shell intersection, J2000 -> GEO, Bowring, elevation) on the same directions."""
    from oracle import ref_numpy as O
    from auromat_amd.coordinates.wcs import zenithal_pix2world
    from auromat_amd.mapping.astrometry import DirectionArrayMapping
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.resample import resample
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 200, 140
    hdr, cam, t = frame_header(w, h, 'iss030')
    hdr = dict(hdr, CTYPE1='RA---ARC-SIP', CTYPE2='DEC--ARC-SIP', A_ORDER=2, B_ORDER=2, A_2_0=2e-5, B_0_2=-1e-5,
               **{'DATE-OBS': t.strftime('%Y-%m-%dT%H:%M:%S.%f')})
    img = frame_image(w, h, seed=9)
    m = getMapping(img, hdr, cameraPosGCRS=cam, altitude=110, identifier='arc')
    assert isinstance(m, DirectionArrayMapping)
    m.checkGuarantees()
    dirs = zenithal_pix2world(hdr, w, h, corner=True)
    p = O.inflated_earth_intersection(dirs.reshape(-1, 3), cam, 110).reshape(dirs.shape)
    lat, lon = O.j2000_to_latlon(p.reshape(-1, 3), O.mat_j2000_to_geo(O.date2es(t)))
    lat, lon = lat.reshape(h + 1, w + 1), lon.reshape(h + 1, w + 1)
    assert np.array_equal(ma.getmaskarray(m.lats), np.isnan(lat))
    ok = ~np.isnan(lat)
    assert np.max(np.abs(m.lats.data[ok] - lat[ok])) < TOL_DEG and np.max(np.abs(m.lons.data[ok] - lon[ok])) < TOL_DEG
    # (the ARC frame differs from the TAN one by far more than the tolerance: the generator matters)
    tan = getMapping(img, dict(hdr, CTYPE1='RA---TAN', CTYPE2='DEC--TAN'), cameraPosGCRS=cam, altitude=110, identifier='tan',
                     fastCenterCalculation=True)
    both = ok & ~ma.getmaskarray(tan.lats)
    assert np.max(np.abs(m.lats.data[both] - tan.lats.data[both])) > 1e-3
    r = resample(m.maskedByElevation(10), pxPerDeg=10)
    r.checkGuarantees()
    # a fresh mapping of this kind only remembers the mask (round 5: the single-pass plan reads the direction array,
    # amt_pipe_launch_dirs) and resamples to the same grid
    import auromat_amd.resample as R
    fresh = getMapping(img, hdr, cameraPosGCRS=cam, altitude=110, identifier='arc').maskedByElevation(10)
    assert fresh._frame is None and fresh._lazy_elev == 10.0
    r2 = R.resample(fresh, pxPerDeg=10)
    assert R.last_plan == 'single-pass' and fresh._frame is None
    assert np.array_equal(ma.getmaskarray(r2.img), ma.getmaskarray(r.img)) and np.array_equal(r2.img.filled(0), r.img.filled(0))


def test_single_pass_right_most_edge_rule():
    """
    Pixels that sit on the last edge of the final grid (within the rounding of histogram.py:215-224) belong to the
    last bin.  The fused kernel bins into a superset grid where that edge is an interior one, so it records such
    pixels and the driver resolves them once the final grid is known.  Frame 0 of the synthetic sequence has one
    on the eastern edge at 8 px/deg, frame 3 one on the northern MLat edge; before the fix the single-pass plan lost
    them (count sums differed by one from the two-pass plan and the oracle's rule).
    """
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 4240, 2832
    seen_edge_pixels = 0
    for k, magnetic in ((0, False), (3, True)):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        img = frame_image(w, h, seed=seed)
        pipe = FramePipeline(w, h, with_mag=magnetic)
        two = pipe.run(hdr, 110, cam, t, img=img, pxPerDeg=8, magnetic=magnetic, fuse=False)
        one = pipe.run(hdr, 110, cam, t, pxPerDeg=8, magnetic=magnetic, fuse=True)
        assert pipe.last_plan == 'single-pass'
        seen_edge_pixels += pipe._fused['result'].edge_pixels
        for key in ('mean', 'count', 'img', 'mask'):
            assert np.array_equal(one[key], two[key], equal_nan=True), (k, key)
        # the pixel in question is in the last column / first row of the output
        a = pipe.host_arrays()
        g = one['grid']
        x = (a['mlt_c'] - 12) / (24 / 360) if magnetic else a['lon_c']
        y = a['mlat_c'] if magnetic else a['lat_c']
        keep = a['elev'] >= 10
        beyond = keep & (((x >= g.xrange[1]) & (x < g.xrange[1] + 1e-6) & (y >= g.yrange[0]) & (y < g.yrange[1])) |
                         ((y >= g.yrange[1]) & (y < g.yrange[1] + 1e-6) & (x >= g.xrange[0]) & (x < g.xrange[1])))
        inside = keep & (x >= g.xrange[0]) & (x < g.xrange[1]) & (y >= g.yrange[0]) & (y < g.yrange[1])
        assert beyond.sum() >= 1
        assert int(one['count'].sum()) == int(inside.sum()) + int(beyond.sum())
    assert seen_edge_pixels > 0


@pytest.mark.parametrize('k,ppd', [(0, 8), (3, 10)])
def test_single_pass_vs_oracle_full_size(k, ppd):
    """BASELINE configs[2] at full size against the oracle itself (3 s of NumPy per frame): bin counts identical in
    every cell — the right-most-edge rule included —, exact integer image means, elevation means within 1e-9 deg."""
    from oracle import ref_numpy as O
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 4240, 2832
    hdr, cam, t, seed = sequence_frame(k, w, h)
    img = frame_image(w, h, seed=seed)
    pipe = FramePipeline(w, h)
    one = pipe.run(hdr, 110, cam, t, img=img, pxPerDeg=ppd, fuse=True)
    assert pipe.last_plan == 'single-pass'
    g = oracle_frame(hdr, cam, t, True)
    want, bbox = oracle_resample(g, img, 10, ppd)
    assert want['data'].shape == one['mean'].shape
    assert np.array_equal(want['count'], one['count'])
    filled = want['count'] > 0
    assert np.array_equal(one['mean'][..., :3][filled], want['data'][..., :3][filled])
    assert np.max(np.abs(one['mean'][..., 3][filled] - want['data'][..., 3][filled])) < 1e-9
    assert np.array_equal(np.isnan(one['mean'][..., 0]), ~filled)


@pytest.mark.parametrize('magnetic', [False, True])
def test_exact_centres_without_elevation_threshold(magnetic):
    """Exact centres (fast=False) and no elevation threshold: the limb pixels whose centre ray hits the shell while
    one of their corner rays misses it are dropped by the reference's sanitisation (mapping.py:1093-1101); both plans
    must apply that rule in the binning pass, not only in the bounding box."""
    from oracle import ref_numpy as O
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 1060, 708
    hdr, cam, t = frame_header(w, h, 'iss030')
    img = frame_image(w, h, seed=21, dtype=np.uint8)
    pipe = FramePipeline(w, h, img_dtype=np.uint8, with_mag=magnetic)
    two = pipe.run(hdr, 110, cam, t, img=img, fast=False, min_elevation=None, pxPerDeg=(4, 7), magnetic=magnetic)
    assert pipe.last_plan == 'two-pass'
    one = pipe.run(hdr, 110, cam, t, fast=False, min_elevation=None, pxPerDeg=(4, 7), magnetic=magnetic, fuse=True)
    assert pipe.last_plan == 'single-pass'
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
        assert np.array_equal(one[k], two[k], equal_nan=True), k

    g = oracle_frame(hdr, cam, t, False)
    corner_mask, center_mask = O.sanitize_masks(np.isnan(g['lat']), np.isnan(g['lat_c']), after_masking=False)
    limb = int((center_mask & ~np.isnan(g['lat_c'])).sum())
    assert limb > 0, 'the frame must hold limb pixels that only the sanitisation removes'
    # (pixels beyond the outermost bin edges are not binned — reference resample.py:301-351 —, hence <=)
    assert (~center_mask).sum() - limb < one['count'].sum() <= (~center_mask).sum()
    if magnetic:
        return                                     # the oracle's resample_mean is the geodetic one
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    want = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), 110,
                           data, None, bbox, (4, 7), disc, False)
    assert want['data'].shape == one['mean'].shape
    assert want['count'].sum() == one['count'].sum()
    from conftest import assert_counts_equal_up_to_edge_pixels
    assert_counts_equal_up_to_edge_pixels(want, one['count'], np.where(center_mask, np.nan, g['lat_c']), g['lon_c'], 'exact centres')
    same = (want['count'] == one['count']) & (want['count'] > 0)
    assert np.array_equal(one['mean'][..., :3][same], want['data'][..., :3][same])
    assert np.max(np.abs(one['mean'][..., 3][same] - want['data'][..., 3][same])) < 1e-9


@pytest.mark.parametrize('width,height', [(1, 1), (2, 1), (1, 3), (5, 3), (64, 1), (65, 2), (129, 5), (3, 70)])
def test_degenerate_frame_sizes(width, height):
    """Frames smaller than a wavefront / a row chunk, single rows and columns: coordinates equal the oracle's and the
    single-pass plan (when a grid exists at all) equals the two-pass plan."""
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    # a width x height window in the Earth-looking lower part of a 256 x 170 frame
    hdr, cam, t = frame_header(256, 170, 'iss030')
    ox, oy = 60, 170 - height - 4
    hdr.update(IMAGEW=width, IMAGEH=height, CRPIX1=hdr['CRPIX1'] - ox, CRPIX2=hdr['CRPIX2'] - oy)
    img = frame_image(width, height, seed=6)
    for fast in (True, False):
        pipe = FramePipeline(width, height, with_mag=True)
        pipe.set_image(img)
        pipe.georef(hdr, 110, cam, t, fast=fast, min_elevation=None)
        got = pipe.host_arrays()
        ref = oracle_frame(hdr, cam, t, fast)
        for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt', 'mlat_c', 'mlt_c'):
            assert got[k].shape == ref[k].shape, k
            nan_close(got[k], ref[k], TOL_DEG)
    if width * height < 3:
        return
    # a resolution fine enough that the few pixels still span more than one grid node
    span = max(np.ptp(ref['lat'][~np.isnan(ref['lat'])]), 1e-3)
    ppd = float(min(2000.0, max(10.0, 8.0 / span)))
    assert not np.isnan(ref['lat_c']).any(), 'the window must look at the Earth'
    pipe = FramePipeline(width, height)
    try:
        two = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=None, pxPerDeg=ppd, fuse=False)
    except AssertionError:
        return                                   # the reference asserts nLat, nLon > 1 as well (resample.py:226-227)
    one = pipe.run(hdr, 110, cam, t, fast=True, min_elevation=None, pxPerDeg=ppd, fuse=True)
    for k in ('mean', 'count', 'img', 'mask'):
        assert np.array_equal(one[k], two[k], equal_nan=True), (k, pipe.last_plan)
    assert two['count'].sum() <= width * height


def test_sequence_with_uploaded_images_from_pinned_memory():
    """Per-frame images from page-locked host memory give the same grids as one frame at a time with the WHOLE image resident on
    the device — although only the rows that can be binned cross the link (amt_georef_image_rows: inside the limb, above
    min_elevation), in the library's frame loop (amt_run_frame.img_host: the runner's copy stream, one batch ahead of the launch)
    as in the Python loop; frames that leave the single-pass plan (pole, empty sky) and buffers re-used while older frames are
    still in flight included."""
    import torch
    from auromat_amd.pipeline import FramePipeline, NativeResults, SequencePipeline
    from auromat_amd.synthetic import frame_image, pole_frame, sequence_frame
    w, h, n = 530, 354, 17
    frames, imgs = [], []
    for k in range(n):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        if k == 6:
            hdr, cam, t = pole_frame(w, h)
        if k == 11:
            hdr = dict(hdr, CRVAL2=hdr['CRVAL2'] + 70.0)           # off the limb: no valid pixel
        img = frame_image(w, h, seed=seed)
        imgs.append(img)
        frames.append((hdr, cam, t, torch.from_numpy(img.view(np.int16)).pin_memory()))
    single = FramePipeline(w, h)
    want = []
    for k, (hdr, cam, t, _) in enumerate(frames):
        try:
            want.append(single.run(hdr, 110, cam, t, img=torch.from_numpy(imgs[k].view(np.int16)).cuda(), pxPerDeg=10))
        except ValueError:
            want.append(None)
    assert want[11] is None and want[6]['contains_pole']
    sent = {}
    for plan, native in (('single-pass', True), ('single-pass', False), ('two-pass', False)):
        seq = SequencePipeline(w, h, pxPerDeg=10, plan=plan)
        seq.native = native
        for rep in range(2):
            got = seq.process(frames, keep_on_device=False)
            assert len(got) == n and isinstance(got, NativeResults) == native
            assert native or seq.s_copy is not None
            assert 0 < seq.uploaded_bytes < 0.8 * n * w * h * 6, seq.uploaded_bytes
            sent[(plan, native)] = seq.uploaded_bytes
            for k in range(n):
                assert (got[k] is None) == (want[k] is None), (plan, native, k)
                if want[k] is None:
                    continue
                for key in ('mean', 'count', 'img', 'mask'):
                    assert np.array_equal(got[k][key], want[k][key], equal_nan=True), (plan, native, rep, k, key)
    # the library's loop sends what the Python loop sends, except that the Python loop sends a band of 90 % or more whole
    assert sent[('single-pass', True)] <= sent[('single-pass', False)] == sent[('two-pass', False)]


def test_uploaded_rows_really_are_all_that_is_read():
    """The device image buffer of a slot holds GARBAGE outside the uploaded band (here: the previous frame's pixels, inverted):
    the grids do not change — full-size frame, the fused kernel and the separate binning pass."""
    import torch
    from auromat_amd.pipeline import FramePipeline, earth_rows_of
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h = 1060, 708
    hdr, cam, t, seed = sequence_frame(5, w, h)
    img = frame_image(w, h, seed=seed)
    p = frame_params(hdr, 110, cam, t, True)
    r0, r1 = earth_rows_of(p, h, 10.0)
    assert 0 < r0 < r1 == h
    bad = (~img).copy()
    bad[r0:r1] = img[r0:r1]
    for fuse in (True, False):
        q = FramePipeline(w, h)
        a = q.run(hdr, 110, cam, t, img=torch.from_numpy(img.view(np.int16)).cuda(), pxPerDeg=10, fuse=fuse)
        b = q.run(hdr, 110, cam, t, img=torch.from_numpy(bad.view(np.int16)).cuda(), pxPerDeg=10, fuse=fuse)
        for key in ('mean', 'count', 'img', 'mask'):
            assert np.array_equal(a[key], b[key], equal_nan=True), (fuse, key)
    # ... and one row more would be missed: the band is not loose by accident (a pixel of row r0 + 32 is binned)
    bad[r0:r0 + 48] = ~img[r0:r0 + 48]
    q = FramePipeline(w, h)
    c = q.run(hdr, 110, cam, t, img=torch.from_numpy(bad.view(np.int16)).cuda(), pxPerDeg=10, fuse=True)
    assert not np.array_equal(a['mean'], c['mean'], equal_nan=True)


def test_plain_c_client_of_the_abi(tmp_path):
    """examples/c_abi_demo.c — C99, no Python / torch / HIP headers — built with gcc against the in-tree library and
    run as its own process: same bits as the Python host for the coordinate arrays, the bounding-box reduction and
    the binned grid of the same frame."""
    import ctypes as C
    import subprocess
    from conftest import ROOT
    from auromat_amd._build import LIB_DIR
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    exe = str(tmp_path / 'c_abi_demo')
    build = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-I' + os.path.join(ROOT, 'include'),
                            os.path.join(ROOT, 'examples', 'c_abi_demo.c'), '-L' + LIB_DIR, '-lauromat_hip',
                            '-Wl,-rpath,' + LIB_DIR, '-Wl,-rpath,/opt/rocm/lib', '-lm', '-o', exe],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert build.returncode == 0 and build.stdout.strip() == '', build.stdout
    w, h = 253, 171
    hdr, cam, t = frame_header(w, h, 'iss030')
    img = frame_image(w, h, seed=13)
    params = frame_params(hdr, 110, cam, t, True)
    (tmp_path / 'params.bin').write_bytes(bytes(C.string_at(C.byref(params), C.sizeof(params))))
    (tmp_path / 'img.bin').write_bytes(img.tobytes())
    run = subprocess.run([exe, str(tmp_path / 'params.bin'), str(tmp_path / 'img.bin'), '10', str(tmp_path / 'out.bin')],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=120)
    assert run.returncode == 0, run.stdout
    assert run.stdout.startswith('ok %d x %d frame' % (w, h)), run.stdout
    raw = np.fromfile(str(tmp_path / 'out.bin'), dtype=np.float64)
    nc, npx = (w + 1) * (h + 1), w * h
    pipe = FramePipeline(w, h)
    res = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10)
    got = pipe.host_arrays()
    o = 0
    for k, n, shape in (('lat', nc, (h + 1, w + 1)), ('lon', nc, (h + 1, w + 1)), ('lat_c', npx, (h, w)),
                        ('lon_c', npx, (h, w)), ('elev', npx, (h, w))):
        assert np.array_equal(raw[o:o + n].reshape(shape), got[k], equal_nan=True), k
        o += n
    bbox = raw[o:o + 8]
    o += 8
    ny, nx = int(raw[o]), int(raw[o + 1])
    o += 2
    assert (ny, nx) == res['count'].shape
    assert np.array_equal(raw[o:o + ny * nx].reshape(ny, nx), res['count'])
    o += ny * nx
    assert np.array_equal(raw[o:o + ny * nx * 4].reshape(ny, nx, 4), res['mean'], equal_nan=True)
    assert o + ny * nx * 4 == raw.size
    bb = pipe.bounding_box()
    assert (bbox[0], bbox[1], bbox[2], bbox[3]) == (bb.latSouth, bb.latNorth, bb.lonWest, bb.lonEast)


def test_readme_quick_start():
    """The snippet of README.md, as written there."""
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.resample import resample, resampleMLatMLT
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 253, 171
    wcsHeader, cam, t = frame_header(w, h, 'iss030')
    wcsHeader.update({'DATE-OBS': t.strftime('%Y-%m-%dT%H:%M:%S.%f'), 'POSX': cam[0], 'POSY': cam[1], 'POSZ': cam[2]})
    img = frame_image(w, h, seed=1)
    m = getMapping(img, wcsHeader, altitude=110, fastCenterCalculation=True).maskedByElevation(10)
    geo = resample(m, pxPerDeg=10)
    mag = resampleMLatMLT(m, pxPerDeg=10)
    near = resample(m, arcsecPerPx=100, method='nearest')
    for r in (geo, mag, near):
        r.checkGuarantees()
    assert geo.lats.shape == geo.lons.shape and geo.img.shape[:2] == geo.elevation.shape
    assert geo.boundingBox.latSouth < geo.centroid.lat < geo.boundingBox.latNorth
    assert geo.outline.shape[1] == 2 and geo.isPlateCarree and not mag.isPlateCarree


def test_scratch_memory_is_per_stream_in_two_pass_magnetic_sequences():
    """Regression: in the two-pass MLat/MLT plan frame k's bounding-box reduction (amt_bbox_corners, on the binning
    stream) and frame k+1's georeferencing (on the main stream) both used the context's one scratch buffer; with
    asynchronous uploads the overlap was long enough to corrupt a frame now and then.  Workspaces are per stream."""
    import torch
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h, n = 1060, 708, 60
    frames, imgs = [], []
    for k in range(n):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        imgs.append(frame_image(w, h, seed=seed))
        frames.append((hdr, cam, t, torch.from_numpy(imgs[-1].view(np.int16)).pin_memory()))
    single = FramePipeline(w, h, with_mag=True)
    want = [single.run(hdr, 110, cam, t, img=imgs[k], pxPerDeg=8, magnetic=True)
            for k, (hdr, cam, t, _) in enumerate(frames)]
    seq = SequencePipeline(w, h, pxPerDeg=8, plan='two-pass', magnetic=True)
    for rep in range(3):
        got = seq.process(frames, keep_on_device=False)
        for k in range(n):
            for key in ('mean', 'count', 'img', 'mask'):
                assert np.array_equal(got[k][key], want[k][key], equal_nan=True), (rep, k, key)


def test_differential_fuzz_of_both_plans_against_the_oracle():
    """tools/fuzz_frames.py with a fixed seed: 150 random frames (sizes, pointings, times incl. date-line crossings,
    shells, anisotropic resolutions, fast / exact centres, thresholds incl. none, uint8 / uint16): single-pass ==
    two-pass bit for bit, two-pass vs oracle within 2 cells with exact integer means."""
    import subprocess
    import sys
    from conftest import ROOT
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_frames.py'), '150', '11'],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=500)
    assert run.returncode == 0 and 'failures 0' in run.stdout, run.stdout[-2000:]
