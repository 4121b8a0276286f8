// Host-side layout of the output grid of resample(method='mean'), in C++.
//
// Same numbers as the Python layout in auromat_amd/resample.py (_Grid / fixedGrid / make_axis), which follows
// the reference's NumPy code operation by operation (auromat/resample.py:220-241,281-299,330-334 and
// auromat/util/histogram.py:186,215-224).  np.linspace(a, b, n)[i] is fl(fl(i*step) + a) with
// step = (b-a)/(n-1) and the last element stored as b; Python's round() on floats and int() are
// round-half-even and truncation.  tests/test_host_cpu.py checks this file against the Python layout on
// thousands of random boxes (amt_grid_layout needs no GPU).
#pragma once

#include <cmath>
#include <cstdint>

#include "../../include/auromat_hip.h"

namespace amt_gl {

struct linspace {
    double start, stop, step;
    int64_t n;
    linspace(double a, double b, int64_t num) : start(a), stop(b), step(num > 1 ? (b - a) / (double)(num - 1) : 0.0), n(num) {}
    // element i exactly as NumPy computes it (two roundings; no fused multiply-add)
    double at(int64_t i) const {
        if (i == n - 1 && n > 1) return stop;
        volatile double m = (double)i * step;
        return m + start;
    }
    // np.argmax(axis > v): first index whose element exceeds v, 0 when there is none
    int64_t first_gt(double v) const {
        int64_t i = (int64_t)std::floor((v - start) / step) + 1;
        if (i < 0) i = 0;
        if (i > n) i = n;
        while (i > 0 && at(i - 1) > v) --i;
        while (i < n && !(at(i) > v)) ++i;
        return i < n ? i : 0;
    }
    // np.argmax(axis >= v)
    int64_t first_ge(double v) const {
        int64_t i = (int64_t)std::floor((v - start) / step);
        if (i < 0) i = 0;
        if (i > n) i = n;
        while (i > 0 && at(i - 1) >= v) --i;
        while (i < n && !(at(i) >= v)) ++i;
        return i < n ? i : 0;
    }
};

inline int64_t py_round(double x) { return (int64_t)std::nearbyint(x); }   // round-half-even (default FP mode)

// histogram axis parameters of np.linspace(first, last, nbin + 1) as make_axis() computes them
inline void fill_axis(amt_axis* ax, double first, double last, int32_t nbin) {
    ax->edges = nullptr;
    ax->nbin = nbin;
    ax->uniform = 1;
    ax->first = first;
    ax->last = last;
    ax->step = (last - first) / (double)nbin;
    const linspace e(first, last, (int64_t)nbin + 1);
    double mindiff = INFINITY;
    for (int32_t i = 0; i < nbin; ++i) {
        const double d = e.at(i + 1) - e.at(i);
        if (d < mindiff) mindiff = d;
    }
    const int decimal = (int)(-std::log10(mindiff)) + 6;          // histogram.py:219
    ax->scale = std::pow(10.0, decimal);
    ax->last_rounded = std::nearbyint(last * ax->scale) / ax->scale;   // np.around(edges[-1], decimal)
}

// fixedGrid + _Grid: returns false when the box gives fewer than one output cell per axis.
inline bool layout(double lat_ppd, double lon_ppd, double lat_min, double lat_max, double lon_min, double lon_max,
                   amt_grid* g) {
    if (!(lat_ppd > 0) || !(lon_ppd > 0)) return false;
    const linspace lat_all(-90.0, 90.0, py_round(lat_ppd * 180 + 1));
    const linspace lon_all(-180.0, 180.0, py_round(lon_ppd * 360 + 1));
    auto wrap = [](int64_t i, int64_t n) { return i < 0 ? i + n : i; };     // Python's negative index
    const double lat_lo = lat_all.at(wrap(lat_all.first_gt(lat_min) - 1, lat_all.n));
    const double lat_hi = lat_all.at(lat_all.first_ge(lat_max));
    const double lon_lo = lon_all.at(wrap(lon_all.first_gt(lon_min) - 1, lon_all.n));
    const double lon_hi = lon_all.at(lon_all.first_ge(lon_max));
    const int64_t n_lat = py_round(lat_ppd * (lat_hi - lat_lo) + 1);
    const int64_t n_lon = py_round(lon_ppd * (lon_hi - lon_lo) + 1);
    g->n_lat_nodes = (int32_t)n_lat;
    g->n_lon_nodes = (int32_t)n_lon;
    g->lat_lo = lat_lo;
    g->lat_hi = lat_hi;
    g->lon_lo = lon_lo;
    g->lon_hi = lon_hi;
    if (n_lat < 3 || n_lon < 3 || n_lat > 65000 || n_lon > 65000) return false;
    const linspace lat_c(lat_hi, lat_lo, n_lat), lon_c(lon_lo, lon_hi, n_lon);    // centres incl. the dropped ends
    g->lat_step = lat_c.step;      // negative: rows run north -> south
    g->lon_step = lon_c.step;
    g->ny = (int32_t)(n_lat - 2);
    g->nx = (int32_t)(n_lon - 2);
    g->lat_center_first = lat_c.at(1);
    g->lat_center_last = lat_c.at(n_lat - 2);
    g->lon_center_first = lon_c.at(1);
    g->lon_center_last = lon_c.at(n_lon - 2);
    // histogram ranges (resample.py:330-334); latitude edges ascend, the output is flipped afterwards
    const double xr0 = g->lon_center_first - g->lon_step / 2, xr1 = g->lon_center_last + g->lon_step / 2;
    const double yr0 = g->lat_center_last + g->lat_step / 2, yr1 = g->lat_center_first - g->lat_step / 2;
    fill_axis(&g->xaxis, xr0, xr1, g->nx);
    fill_axis(&g->yaxis, yr0, yr1, g->ny);
    return true;
}

}  // namespace amt_gl
