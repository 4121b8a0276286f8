"""
2-D histograms with lists of weight arrays, on the GPU.

Mirror of the reference's auromat/util/histogram.py (a NumPy ``histogramdd`` fork that bins with
``searchsorted(edges, x, 'right')`` and evaluates several weight arrays in one index pass).  The
bin assignment keeps those semantics exactly, including the rule that values on the right-most
edge (within the reference's rounding tolerance) belong to the last bin (histogram.py:209-224):
the host prepares the edges with ``np.linspace`` like the reference and the kernel compares
against those very doubles.
"""
import ctypes as C

import numpy as np

from .._native import Axis, Context, ptr, to_host


def make_axis(ctx, edges, uniform=False):
    """amt_axis for the given ascending edges. Returns (struct, device tensor to keep alive)."""
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    dedges = np.diff(edges)
    if np.any(dedges <= 0):
        raise ValueError("Found bin edge of size <= 0. Did you specify `bins` with"
                         "non-monotonic sequence?")
    ax = Axis()
    ax.nbin = nbin = len(edges) - 1
    ax.first, ax.last = float(edges[0]), float(edges[-1])
    dev = None
    if uniform:
        # np.linspace builds arange(n)*step + start and stores the end point; when the given edges
        # are exactly that, the kernel evaluates them in registers and no table is uploaded
        step = (edges[-1] - edges[0]) / nbin
        rebuilt = np.arange(0, nbin + 1) * step + edges[0]
        rebuilt[-1] = edges[-1]
        uniform = step > 0 and np.array_equal(rebuilt, edges)
        if uniform:
            ax.step = float(step)
    ax.uniform = 1 if uniform else 0
    if not uniform:
        dev = ctx.to_device(edges)
        ax.edges = dev.data_ptr()
    mindiff = dedges.min()
    if np.isinf(mindiff):
        ax.scale, ax.last_rounded = 1.0, float('nan')       # rule disabled (histogram.py:218)
    else:
        decimal = int(-np.log10(mindiff)) + 6
        ax.scale = float(10.0 ** decimal)
        ax.last_rounded = float(np.around(edges[-1], decimal))
    return ax, dev


def histogramdd(sample, bins=10, range=None, normed=False, weights=None):
    """
    Histogram of 2-D data (reference histogram.py:57-282, restricted to D = 2).

    :param sample: (N,2) array or a sequence of two (N,) arrays
    :param bins: number of bins, (nx, ny), or a sequence of two edge arrays
    :param range: [[xmin, xmax], [ymin, ymax]] used when bin counts are given
    :param weights: (N,) array, or a list of (arrays | None) -> a list of histograms is returned
    :returns: H (or list of H) of shape (nx, ny), and the list of edge arrays
    """
    try:
        N, D = sample.shape
        xs, ys = sample[:, 0], sample[:, 1]
    except (AttributeError, ValueError):
        xs, ys = sample
        xs, ys = np.asarray(xs), np.asarray(ys)
        N, D = len(xs), 2
    if D != 2:
        raise NotImplementedError('only two-dimensional samples are supported')
    as_list = isinstance(weights, (list, tuple))
    if weights is not None and not as_list and np.ndim(weights) != 1:
        raise AttributeError('Weights must be a 1D-array, None, or a list of both')
    wlist = list(weights) if as_list else [weights]
    try:
        if len(bins) != D:
            raise AttributeError('The dimension of bins must be equal to the dimension of the sample x.')
    except TypeError:
        bins = D * [bins]

    edges = []
    for i, v in enumerate((xs, ys)):
        if np.isscalar(bins[i]):
            if bins[i] < 1:
                raise ValueError("Element at index %s in `bins` should be a positive integer." % i)
            if range is None:
                smin, smax = (0.0, 1.0) if N == 0 else (float(np.min(v)), float(np.max(v)))
            else:
                smin, smax = float(range[i][0]), float(range[i][1])
            if smin == smax:
                smin, smax = smin - .5, smax + .5
            edges.append(np.linspace(smin, smax, bins[i] + 1))
        else:
            edges.append(np.asarray(bins[i], float))
    nx, ny = len(edges[0]) - 1, len(edges[1]) - 1

    ctx = Context.current()
    xaxis, xkeep = make_axis(ctx, edges[0], uniform=np.isscalar(bins[0]))
    yaxis, ykeep = make_axis(ctx, edges[1], uniform=np.isscalar(bins[1]))
    x = ctx.to_device(np.ascontiguousarray(xs, dtype=np.float64))
    y = ctx.to_device(np.ascontiguousarray(ys, dtype=np.float64))
    real = [w for w in wlist if w is not None]
    for w in real:
        assert np.shape(w) == (N,)
    wdev = [ctx.to_device(np.ascontiguousarray(w, dtype=np.float64)) for w in real]
    count = ctx.zeros((nx * ny,))
    sums = [ctx.zeros((nx * ny,)) for _ in wdev]
    k = len(wdev)
    wptr = (C.c_void_p * max(k, 1))(*[t.data_ptr() for t in wdev])
    sptr = (C.c_void_p * max(k, 1))(*[t.data_ptr() for t in sums])
    assert k <= 8, 'at most 8 weight arrays per call'
    ctx.call('amt_hist2d_accumulate', ptr(x), ptr(y), N, wptr, k, C.byref(xaxis), C.byref(yaxis), 0,
             ptr(count), sptr)
    count_h = to_host(count).reshape(nx, ny)
    sums_h = iter([to_host(s).reshape(nx, ny) for s in sums])
    hists = [count_h.copy() if w is None else next(sums_h) for w in wlist]
    if normed:
        area = np.diff(edges[0])[:, None] * np.diff(edges[1])[None, :]
        hists = [h / area / h.sum() for h in hists]
    if as_list:
        return hists, edges
    return hists[0], edges


def histogram2d(x, y, bins=10, range=None, normed=False, weights=None):
    """
    Bi-dimensional histogram of two data samples (reference histogram.py:284-417).

    `x` is histogrammed along the first dimension of the result and `y` along the second.
    Weights can be a list of (weight arrays or None), in which case a list of histograms is returned.

    :returns: H (or list of H) with shape (nx, ny), xedges, yedges
    """
    try:
        N = len(bins)
    except TypeError:
        N = 1
    if N != 1 and N != 2:
        xedges = yedges = np.asarray(bins, float)
        bins = [xedges, yedges]
    hist, edges = histogramdd([x, y], bins, range, normed, weights)
    return hist, edges[0], edges[1]
