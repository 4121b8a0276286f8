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
