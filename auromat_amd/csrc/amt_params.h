// Per-frame host scalars of the frame pipeline in C++: ephemeris seconds, the cxform rotation matrices J2000 -> GEO and
// J2000 -> SM (with the IGRF dipole), the WCS Euler matrix.  Ports of auromat_amd/coordinates/{transform,wcs,igrf}.py,
// which are pinned to the reference's doubles (tests/golden/host_scalars.npz; reference transform.py:491-696,
// wcs.py:133-139, igrf.py:25-58): the same operations in the same order; products of 3x3 matrices accumulate with fused
// multiply-adds in the order k = 0, 1, 2, which is what the BLAS behind NumPy's `dot` does for these sizes on the build
// host (tests/test_host_cpu.py compares the two: equal to the last bit on the dates tried, and in any case to 4e-16).
#pragma once
#include <cmath>
#include <cstring>

#include "../../include/auromat_hip.h"

namespace amt_prm {

constexpr double kDeg2Rad = 0.017453292519943295;      // np.deg2rad(x) = x * (pi / 180)
constexpr double kPi = 3.141592653589793;
constexpr double kWgs84A = 6378.137, kWgs84B = 6356.752314245179;      // km, reference geodesic.py:20-21

struct m3 {
    double v[9];
};

inline m3 mul(const m3& a, const m3& b) {
    m3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
#ifdef AMT_PRM_NOFMA
            c.v[3 * i + j] = a.v[3 * i] * b.v[j] + a.v[3 * i + 1] * b.v[3 + j] + a.v[3 * i + 2] * b.v[6 + j];
#else
            double acc = a.v[3 * i] * b.v[j];
            acc = std::fma(a.v[3 * i + 1], b.v[3 + j], acc);
            acc = std::fma(a.v[3 * i + 2], b.v[6 + j], acc);
            c.v[3 * i + j] = acc;
#endif
        }
    return c;
}

inline m3 transpose(const m3& a) {
    m3 t;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t.v[3 * i + j] = a.v[3 * j + i];
    return t;
}

// rotation_matrix(angle, axis)[:3, :3] for the three axis directions that make it agree with cxform's hapgood_matrix
// (transform.py:491-494): X = [-1, 0, 0], Y = [0, 1, 0], Z = [0, 0, -1]
enum axis_id { AX = 0, AY = 1, AZ = 2 };
inline m3 rotation(double angle, axis_id ax) {
    const double s = std::sin(angle), c = std::cos(angle);
    const double d[3] = {ax == AX ? -1.0 : 0.0, ax == AY ? 1.0 : 0.0, ax == AZ ? -1.0 : 0.0};
    const double k = 1.0 - c;
    m3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.v[3 * i + j] = (i == j ? c : 0.0) + (d[i] * d[j]) * k;
    const double a0 = d[0] * s, a1 = d[1] * s, a2 = d[2] * s;
    r.v[0] += 0.0, r.v[1] += -a2, r.v[2] += a1;
    r.v[3] += a2, r.v[4] += 0.0, r.v[5] += -a0;
    r.v[6] += -a1, r.v[7] += a0, r.v[8] += 0.0;
    return r;
}

inline double T0(double et) { return (et / 86400.0) / 36525.0; }

inline double H(double et) {
    const double jd = (et / 86400.0) - 0.5;
    double hh = (jd - (double)(long long)jd) * 24.0;
    if (hh < 0.0) hh += 24.0;
    return hh;
}

inline double lambda0(double et) {
    const double M = 357.528 + 35999.050 * T0(et);
    const double lambd = 280.460 + 36000.772 * T0(et);
    return lambd + (1.915 - 0.0048 * T0(et)) * std::sin(M * kDeg2Rad) + 0.020 * std::sin((2 * M) * kDeg2Rad);
}

inline double epsilon(double et) { return 23.439 - 0.013 * T0(et); }

// IGRF g01, g11, h11 for 1900 ... 2020 (nT; igrf.py:25-58 of the reference; the last entry is extrapolated there)
constexpr int kIgrfYears = 25;
constexpr double kG01[kIgrfYears] = {-31543, -31464, -31354, -31212, -31060, -30926, -30805, -30715, -30654, -30594, -30554, -30500,
                                      -30421, -30334, -30220, -30100, -29992, -29873, -29775, -29692, -29619.4, -29554.63, -29496.5,
                                      -29442, -29390.5};
constexpr double kG11[kIgrfYears] = {-2298, -2298, -2297, -2306, -2317, -2318, -2316, -2306, -2292, -2285, -2250, -2215, -2169,
                                      -2119, -2068, -2013, -1956, -1905, -1848, -1784, -1728.2, -1669.05, -1585.9, -1501, -1410.5};
constexpr double kH11[kIgrfYears] = {5922, 5909, 5898, 5875, 5845, 5817, 5808, 5812, 5821, 5810, 5815, 5820, 5791, 5776, 5737,
                                      5675, 5604, 5500, 5406, 5306, 5186.1, 5077.99, 4944.26, 4797.1, 4664.1};

// false when the date is outside the table (the Python side raises ValueError)
inline bool igrf(double et, double* g01, double* g11, double* h11) {
    const double idx = (et + 3155803200.0) / 157788000.0;
    const double frac = std::fmod(idx, 1.0);
    if (!(idx >= 0) || idx >= kIgrfYears - 1) return false;
    const int lo = (int)std::floor(idx), hi = (int)std::ceil(idx);
    *g01 = kG01[lo] * (1.0 - frac) + kG01[hi] * frac;
    *g11 = kG11[lo] * (1.0 - frac) + kG11[hi] * frac;
    *h11 = kH11[lo] * (1.0 - frac) + kH11[hi] * frac;
    return true;
}

inline m3 mat_P(double et) {
    const double t0 = T0(et);
    m3 m = rotation((-1.0 * (0.64062 * t0 + 0.00030 * t0 * t0)) * kDeg2Rad, AZ);
    m = mul(m, rotation((0.55675 * t0 - 0.00012 * t0 * t0) * kDeg2Rad, AY));
    m = mul(m, rotation((-1.0 * (0.64062 * t0 + 0.00008 * t0 * t0)) * kDeg2Rad, AZ));
    return m;
}

inline m3 mat_T1(double et) {
    const double theta = 100.461 + 36000.770 * T0(et) + 360.0 * (H(et) / 24.0);
    return rotation(theta * kDeg2Rad, AZ);
}

inline m3 mat_T2(double et) { return mul(rotation(lambda0(et) * kDeg2Rad, AZ), rotation(epsilon(et) * kDeg2Rad, AX)); }

// J2000 -> GEO (transform.py:683-686)
inline m3 j2000_to_geo(double et) { return mul(mat_T1(et), mat_P(et)); }

// J2000 -> SM (transform.py:688-691); false when the date is outside the IGRF table
inline bool j2000_to_sm(double et, m3* out) {
    double g01, g11, h11;
    if (!igrf(et, &g01, &g11, &h11)) return false;
    const double lon = std::atan2(h11, g11) + kPi;
    const double lat = kPi / 2 - std::atan((g11 * std::cos(lon) + h11 * std::sin(lon)) / g01);
    const double qg[3] = {std::cos(lat) * std::cos(lon), std::cos(lat) * std::sin(lon), std::sin(lat)};
    const m3 t2 = mat_T2(et), t1 = mat_T1(et);
    const m3 a = mul(t2, transpose(t1));
    double qe[3];
    for (int i = 0; i < 3; ++i) {
        double acc = a.v[3 * i] * qg[0];
        acc = std::fma(a.v[3 * i + 1], qg[1], acc);
        acc = std::fma(a.v[3 * i + 2], qg[2], acc);
        qe[i] = acc;
    }
    const double psi = std::atan2(qe[1] * kDeg2Rad, qe[2] * kDeg2Rad);
    const m3 t3 = rotation(-psi, AX);
    const double mu = std::atan2(qe[0] * kDeg2Rad, std::sqrt(qe[1] * qe[1] + qe[2] * qe[2]) * kDeg2Rad);
    const m3 t4 = rotation(-mu, AY);
    *out = mul(mul(mul(t4, t3), t2), mat_P(et));
    return true;
}

// euler_matrix(ai, aj, ak, 'rzxz')[:3, :3] (wcs.py:133-139 -> the vendored transformations.py:1042-1102)
inline m3 euler_rzxz(double ai, double aj, double ak) {
    const double t = ai;
    ai = ak, ak = t;
    const double si = std::sin(ai), sj = std::sin(aj), sk = std::sin(ak);
    const double ci = std::cos(ai), cj = std::cos(aj), ck = std::cos(ak);
    const double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
    m3 m;
    const int i = 2, j = 0, k = 1;
    m.v[3 * i + i] = cj;
    m.v[3 * i + j] = sj * si;
    m.v[3 * i + k] = sj * ci;
    m.v[3 * j + i] = sj * sk;
    m.v[3 * j + j] = -cj * ss + cc;
    m.v[3 * j + k] = -cj * cs - sc;
    m.v[3 * k + i] = -sj * ck;
    m.v[3 * k + j] = cj * sc + cs;
    m.v[3 * k + k] = cj * cc - ss;
    return m;
}

// The whole amt_frame_params block of a frame.  Returns AMT_OK, or AMT_EINVAL (date outside the IGRF table with want_sm).
inline int frame_params(const amt_run_frame* f, int32_t width, int32_t height, int32_t fast_center, double altitude,
                        int want_sm, amt_frame_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->width = width;
    p->height = height;
    p->fast_center = fast_center ? 1 : 0;
    for (int i = 0; i < 4; ++i) p->cd[i] = f->cd[i];
    p->crpix[0] = f->crpix[0];
    p->crpix[1] = f->crpix[1];
    const m3 rot = euler_rzxz((f->crval[0] + 90) * kDeg2Rad, (90 - f->crval[1]) * kDeg2Rad, (-(f->lonpole - 90)) * kDeg2Rad);
    std::memcpy(p->rot, rot.v, sizeof(rot.v));
    for (int i = 0; i < 3; ++i) p->cam[i] = f->cam[i];
    p->a = kWgs84A + altitude;
    p->b = kWgs84B + altitude;
    p->a0 = kWgs84A;
    p->b0 = kWgs84B;
    const double et = (f->jd - 2451545) * 86400;
    const m3 geo = j2000_to_geo(et);
    std::memcpy(p->m_geo, geo.v, sizeof(geo.v));
    if (want_sm) {
        m3 sm;
        if (!j2000_to_sm(et, &sm)) return AMT_EINVAL;
        std::memcpy(p->m_sm, sm.v, sizeof(sm.v));
    }
    return AMT_OK;
}

}  // namespace amt_prm
