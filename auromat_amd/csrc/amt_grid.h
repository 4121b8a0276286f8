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


// ---- plateCarreeResolution (reference auromat/resample.py:36-61) ------------------------------------------------------
// The same arithmetic as auromat_amd/coordinates/geodesic.py angularDistanceOnParallel + auromat_amd/resample.py
// plateCarreeResolution (see there: the a12 of geographiclib's Inverse for two points on one parallel, from Karney's
// integral formulation with 48-point Gauss-Legendre quadrature and a bisection for the azimuth at the node), so that a C
// host — and the sequence runner — derives px/deg from a frame's own bounding box like `resample(arcsecPerPx=...)` does.
inline void gauss_legendre(int n, double* x, double* w) {
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 1;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1, p2 = 0;
            for (int j = 0; j < n; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2 * j + 1) * z * p2 - j * p3) / (j + 1);
            }
            pp = n * (z * p1 - p2) / (z * z - 1);
            const double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-16) break;
        }
        x[i] = -z, x[n - 1 - i] = z;
        w[i] = w[n - 1 - i] = 2 / ((1 - z * z) * pp * pp);
    }
}

// angularDistance(Location(lat, lon0), Location(lat, lon0 + dlon)) in degrees, |dlon| < 180; false when there is no
// symmetric geodesic for that longitude difference (the Python function raises ValueError there)
inline bool angular_distance_on_parallel(double lat, double dlon, double* out) {
    constexpr int kN = 48;
    static double nodes[kN], weights[kN];
    static const bool ready = (gauss_legendre(kN, nodes, weights), true);
    (void)ready;
    dlon = std::fabs(dlon);
    if (dlon == 0) {
        *out = 0;
        return true;
    }
    if (!(dlon < 180)) return false;
    const double f = 1 / 298.257223563;
    const double ep2 = f * (2 - f) / ((1 - f) * (1 - f));
    const double beta = std::atan((1 - f) * std::tan(std::fabs(lat) * M_PI / 180.0));
    const double lam = dlon * M_PI / 180.0;
    const double sb = std::sin(beta);
    if (sb < 1e-12) {
        if (!(dlon <= 180 * (1 - f))) return false;
        *out = dlon / (1 - f);
        return true;
    }
    auto lam_of = [&](double ca0, double* sigma12) {
        const double sa0 = std::sqrt(std::fmax(0.0, 1 - ca0 * ca0));
        const double s_p = ca0 > 0 ? std::asin(std::fmin(1.0, sb / ca0)) : M_PI / 2;
        const double om_p = std::atan2(sa0 * std::sin(s_p), std::cos(s_p));
        const double half = (M_PI / 2 - s_p) / 2, mid = (M_PI / 2 + s_p) / 2;
        const double k2 = ep2 * ca0 * ca0;
        double sum = 0;
        for (int i = 0; i < kN; ++i) {
            const double sn = std::sin(mid + half * nodes[i]);
            sum += weights[i] * (2 - f) / (1 + (1 - f) * std::sqrt(1 + k2 * sn * sn));
        }
        *sigma12 = M_PI - 2 * s_p;
        return 2 * (M_PI / 2 - om_p) - f * sa0 * 2 * (half * sum);
    };
    double lo = std::fmax(sb, 1e-300), hi = 1.0, sig;
    if (lam_of(hi, &sig) < lam) return false;
    for (int it = 0; it < 200; ++it) {
        const double midc = 0.5 * (lo + hi);
        if (lam_of(midc, &sig) < lam) lo = midc; else hi = midc;
        if (hi - lo <= 4e-16 * hi) break;
    }
    lam_of(0.5 * (lo + hi), &sig);
    *out = sig * 180.0 / M_PI;
    return true;
}

// (latPxPerDeg, lonPxPerDeg) for a BoundingBox(latSouth, lonWest, latNorth, lonEast) and a spherical resolution
inline bool plate_carree_resolution(double lat_south, double lon_west, double lat_north, double lon_east, double arcsec_per_px,
                                    double* lat_ppd, double* lon_ppd) {
    if (!(arcsec_per_px > 0)) return false;
    const double deg_per_px = arcsec_per_px / 3600.0;
    const double lat_middle = (lat_north + lat_south) / 2;
    const double lons = lon_west > lon_east ? lon_east + 360 - lon_west : lon_east - lon_west;
    double dist;
    if (!(lons > 0) || !angular_distance_on_parallel(lat_middle, std::fmin(lons, 360 - lons), &dist)) return false;
    *lat_ppd = 1 / deg_per_px;
    *lon_ppd = (dist / deg_per_px) / lons;
    return true;
}

}  // namespace amt_gl
