"""
Polygon / outline helpers of the reference's ``auromat.utils`` that sit between georeferencing and resampling
(reference utils.py:97-275): the traced outline of a validity mask, polygon area and centroid, convex hull.

The outline is traced on the device: ``amt_mask_outline_links`` emits, for the whole mask in one pass, the links
of the contour that scikit-image's ``find_contours(mask, 0.99)`` walks (the reference's ``_outline_skimage``,
utils.py:97-140); the host only follows a few thousand links.
"""
import numpy as np

from ._native import Context, ptr, to_host


def contours_from_links(links):
    """
    Closed contours from (key, next key) records (see ``amt_mask_outline_links``): list of int64 arrays of pixel
    indices (key // 4), each without consecutive duplicates (also across the closing point).  The links form
    disjoint cycles; they are ranked with pointer doubling (log2(n) vectorised rounds) instead of being walked.
    """
    links = np.asarray(links, dtype=np.int64).reshape(-1, 2)
    n = len(links)
    if n == 0:
        return []
    order = np.argsort(links[:, 0], kind='stable')
    keys = links[order, 0]
    nxt = np.searchsorted(keys, links[order, 1])
    assert np.all(nxt < n) and np.array_equal(keys[nxt], links[order, 1]), 'dangling contour link'
    rounds = max(1, int(np.ceil(np.log2(n))) + 1)
    # leader of every cycle: its smallest node
    leader = np.arange(n)
    hop = nxt.copy()
    for _ in range(rounds):
        leader = np.minimum(leader, leader[hop])
        hop = hop[hop]
    # steps from every node forward to its leader (the leader absorbs)
    is_leader = leader == np.arange(n)
    dist = np.where(is_leader, 0, 1)
    hop = np.where(is_leader, np.arange(n), nxt)
    for _ in range(rounds):
        dist = dist + dist[hop]
        hop = hop[hop]
    assert np.all(hop == leader), 'contour does not close'
    length = np.zeros(n, dtype=np.int64)
    np.maximum.at(length, leader, dist + 1)                       # the leader's predecessor is length - 1 steps away
    pos = np.where(is_leader, 0, length[leader] - dist)
    walk = np.lexsort((pos, leader))                              # cycle by cycle, in contour order from the leader
    px = keys[walk] // 4
    starts = np.nonzero(is_leader[walk])[0]
    contours = []
    for a, b in zip(starts, list(starts[1:]) + [n]):
        c = px[a:b]
        keep = np.ones(len(c), bool)
        keep[1:] = c[1:] != c[:-1]
        c = c[keep]
        if len(c) > 1 and c[0] == c[-1]:
            c = c[:-1]
        contours.append(c)
    return contours


def outline_of_mask_tensor(ctx, mask, height, width):
    """Outline of a device uint8 mask (1 = masked): (n,2) int array in x,y order, see :func:`outline`."""
    import torch
    count = ctx.zeros((1,), torch.int64)
    capacity = 16 * (height + width) + 64
    while True:
        links = ctx.empty((capacity, 2), torch.int64)
        ctx.call('amt_mask_outline_links', ptr(mask), height, width, ptr(links), capacity, ptr(count))
        n = int(to_host(count, dtype=np.int64)[0])
        if n <= capacity:
            break
        capacity = n
    if n == 0:
        raise ValueError('the mask has no unmasked element')
    contours = contours_from_links(to_host(links[:n], dtype=np.int64))
    polys = [np.transpose([c % width, c // width]) for c in contours]
    if len(polys) > 1:
        # several contours (holes, islands): the biggest one (utils.py:127-137); degenerate ones are dropped
        polys = [p for p in polys if len(p) > 2]
        if not polys:
            raise ValueError('the mask only has degenerate contours')
        polys = [polys[int(np.argmax([polygonArea(p) for p in polys]))]]
    return polys[0]


def outline(im):
    """
    Finds the outline of a binary image, assuming that the inner structure is filled with True's.  The returned
    points are in clockwise order (image coordinates, y down) and can be used as a polygon.  This works for
    concave forms as well (reference utils.py:142-151).

    :param im: shape (h,w), True = inside
    :rtype: ndarray of shape (n,2) in x,y order
    """
    im = np.asarray(im, dtype=bool)
    assert im.ndim == 2
    ctx = Context.current()
    mask = ctx.to_device(np.ascontiguousarray(~im).astype(np.uint8), np.uint8)
    return outline_of_mask_tensor(ctx, mask, im.shape[0], im.shape[1])


def polygonArea(poly, signed=False):
    """Area of an unclosed polygon, (n,2)-array (shoelace formula; reference utils.py:153-171)."""
    p = np.asarray(poly, dtype=np.float64)
    q = np.roll(p, -1, axis=0)
    area = 0.5 * float(np.sum(p[:, 0] * q[:, 1] - q[:, 0] * p[:, 1]))
    return area if signed else abs(area)


def polygonCentroid(poly):
    """
    Centroid (x,y) of an unclosed polygon, (n,2)-array (reference utils.py:173-225).  As in the reference the
    moments are divided by the *unsigned* area: a polygon running against the mathematical orientation gives the
    negated centroid.
    """
    p = np.asarray(poly, dtype=np.float64)
    q = np.roll(p, -1, axis=0)
    cross = p[:, 0] * q[:, 1] - q[:, 0] * p[:, 1]
    area = abs(0.5 * float(np.sum(cross)))
    cx = float(np.sum((p[:, 0] + q[:, 0]) * cross)) / (area * 6.0)
    cy = float(np.sum((p[:, 1] + q[:, 1]) * cross)) / (area * 6.0)
    return (cx, cy)


def withoutConsecutiveDuplicates(arr):
    """Copy of the input where consecutive duplicates on the first dimension are removed (utils.py:235-245)."""
    a = np.asarray(arr)
    if len(a) == 0:
        return a
    keep = np.ones(len(a), bool)
    keep[1:] = np.any(a[1:].reshape(len(a) - 1, -1) != a[:-1].reshape(len(a) - 1, -1), axis=1)
    return a[keep]


def convexHull(points):
    """
    Convex hull spanning the given points, (n,2) -> (m,2), ordered by ``arctan2(dx, dy)`` about the mean of the hull
    vertices as the reference orders them (utils.py:247-276).  Monotone chain instead of the reference's Delaunay
    triangulation; like the boundary of that triangulation it keeps the points that lie ON a hull edge — the pole
    test samples this hull (mapping.py:705-715), and along the straight sides of a frame those points are what makes
    the sampled polygon follow the footprint instead of cutting across it with a few long geodesics.
    """
    pts = np.unique(np.asarray(points).reshape(-1, 2), axis=0)
    assert pts.ndim == 2 and pts.shape[1] == 2
    if len(pts) > 2:
        def half(seq):
            h = []
            for p in seq:
                while len(h) >= 2 and ((h[-1][0] - h[-2][0]) * (p[1] - h[-2][1]) -
                                       (h[-1][1] - h[-2][1]) * (p[0] - h[-2][0])) < 0:
                    h.pop()
                h.append(p)
            return h
        seq = pts.tolist()
        lower, upper = half(seq), half(seq[::-1])
        pts = np.asarray(lower[:-1] + upper[:-1], dtype=pts.dtype)
    centered = pts - pts.mean(axis=0)
    return pts[np.argsort(np.arctan2(centered[:, 0], centered[:, 1]))]


__all__ = ['outline', 'polygonArea', 'polygonCentroid', 'withoutConsecutiveDuplicates', 'convexHull']


def findNearest(a, x):
    """Index of the item of the sorted list `a` that is closest to `x`; the left one when both neighbours are equally
    far (reference utils.py:277-296)."""
    import bisect
    i = bisect.bisect_left(a, x)
    if i == len(a):
        return i - 1
    if i == 0 or a[i] == x:
        return i
    return i - 1 if (x - a[i - 1]) <= (a[i] - x) else i


# ---- small vector helpers of the reference (utils.py:28-75,294-305) ------------------------------------------------------
# NumPy in -> NumPy out, torch device tensor in -> tensor out, like the operator-level mirrors (auromat_amd._ops).  The frame
# kernels do not call them (elevation, masks and the outside-outline test are fused there); they are here for code written
# against the reference's module.

def _staged(*arrays):
    from ._ops import Staged
    return Staged(*arrays)


def vectorLengths(vectors):
    """``np.linalg.norm(vectors, axis=1)`` (reference utils.py:28-31)."""
    import torch
    st = _staged(vectors)
    v = st.inp(vectors)
    return st.result(torch.sqrt((v * v).sum(dim=1)))


def unitVectors(vectors):
    """The unit vectors of an array of vectors (reference utils.py:33-36)."""
    import torch
    st = _staged(vectors)
    v = st.inp(vectors)
    return st.result(v / torch.sqrt((v * v).sum(dim=1))[..., None])


def angleBetween(v1, v2):
    """Angles in radians, in [0, pi], between two unit vector arrays; the dot product is clipped to [-1, 1] before the
    arccosine, as in the reference (utils.py:38-46)."""
    import torch
    st = _staged(v1, v2)
    a, b = st.inp(v1), st.inp(v2)
    return st.result(torch.arccos(torch.clamp((a * b).sum(dim=-1), -1.0, 1.0)))


def signedAngleBetween(v1, v2):
    """Angles in radians, in [-pi, pi], between two 2D vector arrays (reference utils.py:48-56)."""
    import torch
    st = _staged(v1, v2)
    a, b = st.inp(v1), st.inp(v2)
    return st.result(torch.atan2(a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0], a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]))


def pointsInsidePolygon(points, polygon):
    """For each point whether it lies inside the polygon — ``matplotlib.path.Path(polygon).contains_points(points)`` in the
    reference (utils.py:58-74) —: ``amt_points_in_polygon``, the crossing test with Agg's half-open edge rule, which equals
    matplotlib also for points on vertices and edges (tests/test_gpu_nearest.py).

    :param points: shape (n, 2)
    :param polygon: unclosed, shape (m, 2)
    :rtype: boolean array of shape (n,)
    """
    import torch
    st = _staged(points)
    p = st.inp(points)
    ctx = st.ctx
    poly = ctx.to_device(np.ascontiguousarray(polygon, dtype=np.float64))
    x, y = p[:, 0].contiguous(), p[:, 1].contiguous()
    inside = ctx.empty((p.shape[0],), torch.uint8)
    ctx.call('amt_points_in_polygon', ptr(x), ptr(y), x.numel(), ptr(poly), int(poly.shape[0]), ptr(inside))
    if st.on_device:
        return inside.to(torch.bool)
    return to_host(inside).astype(bool)


def extend(instance, new_class):
    """Apply inheritance after object creation (reference utils.py:294-305)."""
    instance.__class__ = type('%s_extended_with_%s' % (instance.__class__.__name__, new_class.__name__),
                              (new_class, instance.__class__), {})
