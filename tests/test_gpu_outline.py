"""
GPU parity tests of the traced outline (reference utils.py:97-225, mapping.py:655-691,758-784):
``amt_mask_outline_links`` + the host link follower against the oracle's restated find_contours, against the
reference's literal vectors (outline_test.py:109-158, stored in tests/golden/known_answers.json), and on masks with
holes, islands, one-pixel spurs and diagonal contacts.
"""
import json
import os

import numpy as np
import numpy.ma as ma
import pytest

from conftest import GOLDEN, header_from, load_golden

pytestmark = pytest.mark.gpu


def known():
    with open(os.path.join(GOLDEN, 'known_answers.json')) as fp:
        return json.load(fp)['outline']


def disc_image(spec):
    n, r = int(spec['n']), float(spec['radius'])
    y, x = np.ogrid[-r: r + 1, -r: r + 1]
    im = np.zeros((n, n), bool)
    disc = x ** 2 + y ** 2 <= r ** 2
    im[:disc.shape[0], :disc.shape[1]] = disc
    im[tuple(spec['removed'])] = False
    return im


def same_polygon(a, b):
    """equal as closed polygons: same vertices in the same cyclic order (the starting point is free)"""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    for s in np.nonzero(np.all(b == a[0], axis=1))[0]:
        if np.array_equal(np.roll(b, -s, axis=0), a):
            return True
    return False


def test_outline_reference_vectors():
    from auromat_amd.utils import outline, polygonArea, polygonCentroid
    ka = known()
    got = outline(disc_image(ka['test_image']))
    assert same_polygon(got, ka['polygon'])
    assert polygonArea(got) == ka['area'] and polygonArea(got, signed=True) == ka['area']
    np.testing.assert_almost_equal(polygonCentroid(ka['centroid_polygon']), ka['centroid'],
                                   decimal=ka['centroid_decimals'])
    np.testing.assert_allclose(polygonCentroid(ka['centroid_polygon']), ka['centroid_ref'], rtol=1e-14)


@pytest.mark.parametrize('seed', range(6))
def test_outline_vs_oracle_on_awkward_masks(seed):
    """Blobs with holes, islands, spurs and diagonal contacts: every contour choice of the reference (biggest by
    area, degenerate ones dropped, 4-connected inside, pixels revisited on spurs)."""
    from oracle import ref_numpy as O
    from auromat_amd.utils import outline
    rng = np.random.RandomState(seed)
    h, w = 37 + 5 * seed, 53 - 3 * seed
    yy, xx = np.mgrid[0:h, 0:w]
    im = ((yy - h / 2) / (h / 2.3)) ** 2 + ((xx - w / 2) / (w / 2.2)) ** 2 <= 1
    im &= rng.rand(h, w) > 0.04                     # holes, some on the rim -> notches and diagonal contacts
    im[h // 2, :] |= (xx[0] > 2) & (xx[0] < w - 2) & (seed % 2 == 0)          # a spur through the middle and beyond
    im[1:4, 1:4] = True                             # an island
    im[2, 2] = seed % 3 == 0
    if seed == 5:
        im[:, :] = False
        im[3, 4] = im[4, 5] = im[5, 4] = True       # three pixels touching only diagonally + a 2x3 block
        im[10:12, 10:13] = True
    want = O.outline(im)
    got = outline(im)
    assert same_polygon(got, want), (got.tolist(), want.tolist())


def test_outline_full_frame_edges_and_single_pixel():
    from auromat_amd.utils import outline
    full = np.ones((5, 7), bool)
    got = outline(full)
    ring = [[x, 0] for x in range(7)] + [[6, y] for y in range(1, 5)] + [[x, 4] for x in range(5, -1, -1)] + \
        [[0, y] for y in range(3, 0, -1)]
    assert same_polygon(got, ring)
    one = np.zeros((4, 4), bool)
    one[2, 1] = True
    assert outline(one).tolist() == [[1, 2]]
    with pytest.raises(ValueError):
        outline(np.zeros((3, 3), bool))


def test_mapping_outline_centroid_known_answer_full_size():
    """outline_test.py:151-158 on the real ISS030-E-102170 header at its native 4256 x 2832: traced outline of the
    valid corners on the device, centroid to 6 decimals of the literal; plus the oracle's outline of the same mask."""
    from oracle import ref_numpy as O
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from datetime import datetime
    ka = known()['mapping_centroid']
    z = load_golden('georef_full_iss030_fast.npz')
    hdr = header_from(z)
    t = datetime.strptime(str(z['time_iso']), '%Y-%m-%dT%H:%M:%S.%f')
    img = np.zeros((hdr['IMAGEH'], hdr['IMAGEW'], 3), np.uint8)
    m = ArraySpacecraftMapping(hdr, float(z['altitude']), img, z['cam'], t, 'iss030', fastCenterCalculation=True)
    c = m.centroid
    np.testing.assert_almost_equal([c.lat, c.lon], ka['expect'], decimal=ka['decimals'])
    el = m.elevation                                         # elevation_test.py:15-22 on the same fixture
    assert 0 <= el.min() <= el.max() <= 90
    outl = m.outline
    assert outl.shape[1] == 2 and len(outl) > 10000
    mask = ~ma.getmaskarray(m.lats)
    want = O.outline(mask)
    lats, lons = m.lats.data, m.lons.data
    want_latlon = np.transpose([lats[want[:, 1], want[:, 0]], lons[want[:, 1], want[:, 0]]])
    assert same_polygon(outl, want_latlon)
    # bounding box = extremes of the outline (mapping.py:699-705)
    bb = m.boundingBox
    assert bb.latSouth == outl[:, 0].min() and bb.latNorth == outl[:, 0].max()
    assert bb.lonWest == outl[:, 1].min() and bb.lonEast == outl[:, 1].max()
    hull = m.outlineConvexHull
    assert 3 <= len(hull) < len(outl)
    # masking by elevation moves the outline inwards and the centroid with it
    m2 = m.maskedByElevation(10)
    assert len(m2.outline) < len(outl)
    c2 = m2.centroid
    assert abs(c2.lat - c.lat) < 5 and abs(c2.lon - c.lon) < 10 and (c2.lat, c2.lon) != (c.lat, c.lon)


def test_centroid_across_the_dateline_and_pole():
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.synthetic import frame_header, frame_image
    from datetime import timedelta
    w, h = 253, 171
    hdr, cam, t = frame_header(w, h, 'iss029')
    m = ArraySpacecraftMapping(hdr, 110, frame_image(w, h, seed=2), cam, t - timedelta(minutes=80), 'x',
                               fastCenterCalculation=True).maskedByElevation(10)
    if m.containsPole:
        with pytest.raises(NotImplementedError):
            m.centroid
        return
    c = m.centroid
    bb = m.boundingBox
    assert bb.latSouth < c.lat < bb.latNorth
    if m.containsDiscontinuity:
        assert c.lon > bb.lonWest or c.lon < bb.lonEast
    else:
        assert bb.lonWest < c.lon < bb.lonEast


def test_outline_links_error_behaviour():
    from auromat_amd._native import Context, NativeError, ptr
    import torch
    ctx = Context.current()
    mask = ctx.zeros((4, 4), torch.uint8)
    count = ctx.zeros((1,), torch.int64)
    with pytest.raises(NativeError):
        ctx.call('amt_mask_outline_links', None, 4, 4, None, 0, ptr(count))
    with pytest.raises(NativeError):
        ctx.call('amt_mask_outline_links', ptr(mask), 0, 4, None, 0, ptr(count))
    ctx.call('amt_mask_outline_links', ptr(mask), 4, 4, None, 0, ptr(count))       # counting only
    assert int(count.cpu()[0]) == 16                                                  # 4 sides x 4 pixels


def test_masked_by_polygon_vs_matplotlib():
    """BaseMapping.maskedByPolygon (mapping.py:866-917): corners inside the polygon (matplotlib's rule), a pixel needs
    all four; checked against the same steps in NumPy on the oracle's arrays, plain and across the date line."""
    import matplotlib.path
    from datetime import timedelta
    from oracle import ref_numpy as O
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 256, 170
    for pointing, shift in (('iss030', 0), ('iss029', 80)):
        hdr, cam, t = frame_header(w, h, pointing)
        t = t - timedelta(minutes=shift)
        m = ArraySpacecraftMapping(hdr, 110, frame_image(w, h, seed=4), cam, t, 'p', fastCenterCalculation=True)
        et = O.date2es(t)
        g = O.georef_frame(hdr, 110, cam, O.mat_j2000_to_geo(et), None, fast=True)
        ok = ~np.isnan(g['lat'])
        lat0, lat1 = np.percentile(g['lat'][ok], [30, 70])
        lon = g['lon'][ok]
        disc = m.containsDiscontinuity
        lons = O.wrap_at(lon + 180, 180) if disc else lon
        lon0, lon1 = np.percentile(lons, [25, 75])
        poly = np.array([[lat0, lon0], [lat1 + 1, lon0 + 0.5], [lat1, lon1], [(lat0 + lat1) / 2, (lon0 + lon1) / 2],
                         [lat0 - 0.5, lon1]])
        if disc:
            poly[:, 1] = O.wrap_at(poly[:, 1] - 180, 180)              # back to true longitudes, straddling +-180
        mm = m.maskedByPolygon(poly)
        mm.checkGuarantees()
        pts = np.transpose([g['lat'].ravel(), (O.wrap_at(g['lon'] + 180, 180) if disc else g['lon']).ravel()])
        p2 = poly.copy()
        if disc:
            p2[:, 1] = O.wrap_at(p2[:, 1] + 180, 180)
        inside = matplotlib.path.Path(p2).contains_points(np.nan_to_num(pts, nan=1e9)).reshape(g['lat'].shape)
        mask = ~inside | ~ok
        center = np.logical_or.reduce((mask[:-1, :-1], mask[1:, :-1], mask[:-1, 1:], mask[1:, 1:]))
        assert 0 < (~center).sum() < center.size
        got = ma.getmaskarray(mm.latsCenter)
        assert int((got != center).sum()) <= 2, int((got != center).sum())      # corners within 1e-11 deg of an edge
    with pytest.raises(ValueError):
        m.maskedByPolygon([[0, 0], [1, 0], [1, 1]])


def test_arc_sec_per_px():
    """BaseMapping.arcSecPerPx (mapping.py:786-843): geodesic arcs of the sides and the diagonal of 1000 sampled
    pixel polygons, the same sample as the reference takes."""
    from oracle import ref_numpy as O
    from auromat_amd.coordinates.geodesic import Location, angularDistance
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 253, 171
    hdr, cam, t = frame_header(w, h, 'iss030')
    m = ArraySpacecraftMapping(hdr, 110, frame_image(w, h, seed=4), cam, t, 'p', fastCenterCalculation=True)
    s = m.arcSecPerPx
    for side in (s.width, s.height, s.diagonal):
        assert 0 < side.min <= side.median <= side.max and side.min <= side.mean <= side.max
    assert s.diagonal.median > s.width.median and s.diagonal.median > s.height.median
    # the sampling rule itself, on the oracle's arrays
    et = O.date2es(t)
    g = O.georef_frame(hdr, 110, cam, O.mat_j2000_to_geo(et), None, fast=True)
    bad = np.isnan(g['lat'])
    has_nans = bad[:-1, :-1] | bad[:-1, 1:] | bad[1:, 1:] | bad[1:, :-1]
    polys = np.nonzero(~has_nans.ravel())[0]
    pick = polys[np.round(np.linspace(0, len(polys) - 1, 1000)).astype(int)]
    i, j = pick // w, pick % w
    widths = [angularDistance(Location(g['lat'][a, b], g['lon'][a, b]), Location(g['lat'][a, b + 1], g['lon'][a, b + 1]))
              for a, b in zip(i, j)]
    assert abs(np.median(widths) * 3600 - s.width.median) < 1e-6
    assert abs(max(widths) * 3600 - s.width.max) < 1e-6


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16])
def test_pixel_polygons_for_drawing(dtype):
    """draw_helpers.generatePolygonsFromMapping (reference draw_helpers.py:34-94, the call of mapping_test.py:16-23), and
    createPolygonsAndColors + filterNanPolygons on the mapping's arrays: every unmasked pixel in row-major order, its four
    corners clockwise from the upper left as (lat, lon), its colour as the mapping's `rgb` gives it — written out here from
    that definition, pixel by pixel."""
    from auromat_amd.draw_helpers import (ColorMode, createPolygonsAndColors, filterNanPolygons,
                                          generatePolygonsFromMapping)
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 253, 171
    hdr, cam, t = frame_header(w, h, 'iss030')
    m = ArraySpacecraftMapping(hdr, 110, frame_image(w, h, seed=8, dtype=dtype), cam, t, 'd',
                               fastCenterCalculation=True).maskedByElevation(10)

    def expected(lats, lons, as_float):
        la, lo = ma.filled(lats, np.nan), ma.filled(lons, np.nan)
        r, c = np.nonzero(~ma.getmaskarray(m.latsCenter))
        verts = np.empty((len(r), 4, 2))
        for k, (dr, dc) in enumerate(((0, 0), (0, 1), (1, 1), (1, 0))):
            verts[:, k, 0], verts[:, k, 1] = la[r + dr, c + dc], lo[r + dr, c + dc]
        rgb = ma.getdata(m.rgb)[r, c]
        assert rgb.dtype == np.uint8
        return verts, (rgb / 255.0 if as_float else rgb)

    for mode in (None, ColorMode.matplotlib):
        wv, wc = expected(m.lats, m.lons, mode is not None)
        verts, colors = generatePolygonsFromMapping(m, mode)
        assert verts.shape == wv.shape == (int((~ma.getmaskarray(m.latsCenter)).sum()), 4, 2)
        assert np.array_equal(verts, wv) and not np.isnan(verts).any()
        assert colors.dtype == wc.dtype and np.array_equal(colors, wc)
        # the two-step form on arrays: a polygon for every pixel, then those with a colour
        av, ac = createPolygonsAndColors(m.lats, m.lons, m.rgb, mode)
        assert av.shape == (w * h, 4, 2) and ac.shape == (w * h, 3) and ma.isMaskedArray(ac)
        fv, fc = filterNanPolygons(av, ac)
        assert np.array_equal(fv, wv) and fc.dtype == wc.dtype and np.array_equal(fc, wc)
    # NaN colours (a float image with holes) are dropped like masked ones
    fv, fc = filterNanPolygons(np.zeros((3, 4, 2)), np.array([[1.0, 1, 1], [np.nan, np.nan, np.nan], [0.5, 0, 0]]))
    assert fv.shape == (2, 4, 2) and np.array_equal(fc, [[1.0, 1, 1], [0.5, 0, 0]])
    # other coordinates (MLat / MLT corners) through coordsFn
    mlat, mlt = m.mLatMlt
    v2, c2 = generatePolygonsFromMapping(m, None, coordsFn=lambda mp: mp.mLatMlt)
    wv, wc = expected(mlat, mlt, False)
    assert np.array_equal(v2, wv) and np.array_equal(c2, wc)


def test_pole_inside_a_hole_of_the_mapping():
    """The device pole test counts unmasked pixels whose corner quad winds around a pole; when the pole sits in a hole
    (or in a part masked by elevation) no pixel sees it, but the reference's rule — the sampled convex hull of the
    outline contains or crosses a pole (mapping.py:705-721) — still applies: the box is degenerate and resampling
    rotates the pole away.  Found by tools/fuzz_mapping.py."""
    from datetime import datetime
    from oracle import ref_numpy as O
    from auromat_amd.mapping.mapping import GenericMapping
    from auromat_amd.resample import resample
    h, w = 16, 18
    lat_g, lon_g = np.meshgrid(np.linspace(4, -4, h + 1), np.linspace(-4.5, 4.5, w + 1), indexing='ij')

    def move(la, lo):        # the patch centred on (0, 0) is tipped over the north pole
        a, o = O.rotate_pole(np.deg2rad(la.ravel()), np.deg2rad(lo.ravel()), 110, angle=-90.3, axis=(0, 1, 0))
        return np.rad2deg(a).reshape(la.shape), np.rad2deg(o).reshape(la.shape)
    lats, lons = move(lat_g, lon_g)
    lats_c, lons_c = move((lat_g[:-1, :-1] + lat_g[1:, 1:]) / 2, (lon_g[:-1, :-1] + lon_g[1:, 1:]) / 2)
    assert lats.max() > 89
    img = np.random.RandomState(2).randint(0, 256, (h, w, 3)).astype(np.uint8)
    elev = np.full((h, w), 45.0)
    whole = GenericMapping(lats, lons, lats_c, lons_c, elev, 110, img, np.array([7000.0, 0, 0]), datetime(2012, 1, 25), 'p')
    assert whole.containsPole
    # mask the 3 x 3 pixels around the pole
    r, c = np.unravel_index(np.argmax(lats_c), lats_c.shape)
    holed_c, holed_o = lats_c.copy(), lons_c.copy()
    holed_c[r - 1:r + 2, c - 1:c + 2] = np.nan
    holed_o[r - 1:r + 2, c - 1:c + 2] = np.nan
    m = GenericMapping(lats, lons, holed_c, holed_o, elev, 110, img, np.array([7000.0, 0, 0]), datetime(2012, 1, 25), 'p')
    assert m.containsPole and m.containsDiscontinuity
    bb = m.boundingBox
    assert (bb.lonWest, bb.lonEast, bb.latNorth) == (-180, 180, 90)
    res = resample(m, pxPerDeg=2)
    res.checkGuarantees()
    assert (~ma.getmaskarray(res.latsCenter)).sum() > 20
