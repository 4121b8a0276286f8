"""
The reference's OWN call form — `resample(mapping, arcsecPerPx=100)` (test/mapping_test.py:24-42; `auromat-convert
--resolution 100`, cli/convert.py:58-130,176-185) — against grids made by the REAL reference's `_resample`
(oracle/make_golden.py: real_frame_arcsec; tests/golden/real_frame_{iss030,iss029}{,_sm}_arcsec100.npz): both frames of the
reference's mapping test, geographic and (MLat, SM longitude) grid, at the (latPxPerDeg, lonPxPerDeg) pair this repository's
plateCarreeResolution gives for the mapping's bounding box.  geographiclib's a12 is absent from the image, so the pair itself is
pinned by two independent formulations (Karney's integrals, Vincenty's series: both values are in the fixture); everything
behind it — a grid whose two axes differ and are no whole number of pixels per degree, the binned means — is the reference's.

CPU: the oracle and the host functions (Python and C++) against the fixtures.  GPU: the box-first plan through the mapping
classes, the frame pipeline and the sequence runner.
"""
import os

import numpy as np
import numpy.ma as ma
import pytest

from conftest import GOLDEN, load_golden, oracle_frame

FRAMES = {
    'iss030': (os.path.join(GOLDEN, 'resources', 'ISS030-E-102170_dc.jpg'), os.path.join(GOLDEN, 'resources', 'ISS030-E-102170_dc.wcs')),
    'iss029': (os.path.join(GOLDEN, 'resources', 'south', 'ISS029-E-8492.jpg'), os.path.join(GOLDEN, 'resources', 'south', 'ISS029-E-8492.wcs')),
}
CASES = [('iss030', False), ('iss030', True), ('iss029', False), ('iss029', True)]


def fixture(tag, magnetic):
    return load_golden('real_frame_%s%s_arcsec100.npz' % (tag, '_sm' if magnetic else ''))


def inputs(tag):
    from auromat_amd.fits import getShiftedSpacecraftPosition, getSpacecraftPosition, readHeader
    from auromat_amd.util.image import loadImage
    jpg, wcs = FRAMES[tag]
    hdr = readHeader(wcs)
    if tag == 'iss030':
        cam, t, _ = getShiftedSpacecraftPosition(hdr)
    else:
        cam, t = getSpacecraftPosition(hdr)
    return hdr, loadImage(jpg), cam, t


def check_image(res_img, res_mask, mean, z):
    want = z['out_data']
    assert res_mask.shape == want.shape[:2], (res_mask.shape, want.shape)
    assert np.array_equal(res_mask, np.isnan(want[..., 0]))
    ok = ~res_mask
    assert ok.sum() > 20000
    if mean is not None:
        assert np.array_equal(mean[..., :3][ok], want[..., :3][ok])                 # exact integer sums / counts
        assert np.max(np.abs(mean[..., 3][ok] - want[..., 3][ok])) < 1e-9           # elevation, fixed point
    assert np.array_equal(res_img[ok], z['out_img'][ok])


@pytest.mark.parametrize('tag,magnetic', CASES)
def test_resolution_pair_of_the_fixture(lib_or_none, tag, magnetic):
    """The px/deg pair the fixture was made with: from the Python restatement, equal to the C++ one the pipelines use to 1e-12
    (the same number of global grid nodes, which is all the layout takes from it), the a12 behind it agreed on by two
    independent formulations to 3e-10, and not a whole number of pixels per degree."""
    from auromat_amd.mapping.mapping import BoundingBox
    from auromat_amd.resample import plateCarreeResolution, plateCarreeResolution_py
    z = fixture(tag, magnetic)
    bb = BoundingBox(*z['bbox'])
    py = plateCarreeResolution_py(bb, 100.0)
    assert py == tuple(z['ppd']) and py[0] == 36.0 and 10 < py[1] < 30 and abs(py[1] - round(py[1])) > 0.05
    assert abs(float(z['a12_karney_integrals']) - float(z['a12_vincenty'])) < 3e-10 * float(z['a12_vincenty'])
    native = plateCarreeResolution(bb, 100.0)
    assert native[0] == py[0] and abs(native[1] - py[1]) < 1e-12 * py[1]
    assert round(native[1] * 360 + 1) == round(py[1] * 360 + 1)


@pytest.fixture
def lib_or_none():
    from auromat_amd import _native
    return _native.lib()


@pytest.mark.parametrize('tag', ['iss030', 'iss029'])
def test_oracle_equals_the_reference_at_the_arcsec_resolution(tag):
    """ONE run of the oracle's frame, both grids: boxes, px/deg pair, grid coordinates and every cell equal the real `_resample`'s."""
    from oracle import ref_numpy as O
    from auromat_amd.mapping.mapping import BoundingBox
    from auromat_amd.resample import plateCarreeResolution_py
    zg, zs = fixture(tag, False), fixture(tag, True)
    hdr, img, cam, t = inputs(tag)
    g = oracle_frame(hdr, 110.0, zg['cam'], zg['m_geo'], zg['m_sm'], fast=True)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    assert int((~center_mask).sum()) == int(zg['n_valid'])
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    sm_lon, sm_lon_c = O.mlt_to_sm_lon(g['mlt']), O.mlt_to_sm_lon(g['mlt_c'])
    for z, la, lo, lac, loc in ((zg, g['lat'], g['lon'], g['lat_c'], g['lon_c']), (zs, g['mlat'], sm_lon, g['mlat_c'], sm_lon_c)):
        bbox, disc = O.bbox_of_corners(la, lo, corner_mask)
        assert np.array_equal(bbox, z['bbox']) and disc == bool(z['contains_discontinuity'])
        ppd = plateCarreeResolution_py(BoundingBox(*bbox), 100.0)
        assert ppd == tuple(z['ppd'])
        outline = np.transpose([la[~corner_mask], lo[~corner_mask]])
        res = O.resample_mean(np.where(center_mask, np.nan, lac), np.where(center_mask, np.nan, loc), 110.0, data, outline, bbox,
                              ppd, disc, False)
        for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c'), ('data', 'out_data')):
            assert np.array_equal(res[a], z[b], equal_nan=True), (tag, a)


@pytest.mark.gpu
@pytest.mark.parametrize('tag,magnetic', CASES)
def test_box_first_plan_equals_the_references_grids(tag, magnetic):
    """`resample(mapping, arcsecPerPx=100)` / `resampleMLatMLT(...)` on a mapping nobody has materialised: the box-first plan
    (box pass, plateCarreeResolution, ONE fused kernel) — the reference's cells, all of them and no others; on the geographic
    grid also its grid coordinates bit for bit (the MLat / MLT result is handed back in geographic coordinates).  Then the frame
    pipeline's result dict (means exact) and the sequence runner's box-first plan, which take the pair from the native host
    function."""
    import auromat_amd.resample as R
    from auromat_amd._native import Context
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    z = fixture(tag, magnetic)
    jpg, wcs = FRAMES[tag]
    mm = getMapping(jpg, wcs, altitude=110, fastCenterCalculation=True).maskedByElevation(10)
    r = (R.resampleMLatMLT if magnetic else R.resample)(mm, arcsecPerPx=100)
    assert R.last_plan == 'single-pass' and Context.current().last_variant()[0] == (4 if magnetic else 0)
    check_image(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
    if not magnetic:
        for name, key in (('lats', 'out_lat'), ('lons', 'out_lon'), ('latsCenter', 'out_lat_c'), ('lonsCenter', 'out_lon_c')):
            assert np.array_equal(getattr(r, name).data, z[key]), name
    hdr, img, cam, t = inputs(tag)
    pipe = FramePipeline(4256, 2832, img_dtype=np.uint8, with_mag=magnetic, with_geo=not magnetic, alloc_coords=False)
    res = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, arcsecPerPx=100, magnetic=magnetic, fuse=True)
    assert pipe.last_plan == 'single-pass' and bool(res['contains_discontinuity']) == bool(z['contains_discontinuity'])
    assert abs(res['pxPerDeg'][1] - z['ppd'][1]) < 1e-9 * z['ppd'][1] and res['pxPerDeg'][0] == 36.0
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c')):
        assert np.array_equal(res[a], z[b]), a
    check_image(res['img'], res['mask'], res['mean'], z)
    seq = SequencePipeline(4256, 2832, img_dtype=np.uint8, altitude=110, fast=True, min_elevation=10, arcsecPerPx=100,
                           magnetic=magnetic, keep_coordinates=False)
    import torch
    dev_img = torch.from_numpy(np.array(img)).to(seq.ctx.device)
    out = seq.process([(hdr, cam, t, dev_img)] * 4, keep_on_device=False)
    assert seq.plans == ['single-pass'] * 4
    for q in out:
        check_image(q['img'], q['mask'], q['mean'], z)
