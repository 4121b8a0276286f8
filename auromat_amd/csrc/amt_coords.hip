// Building-block kernels mirroring auromat.coordinates (one thread per point, grid-stride).
// These are the operator-level entry points; the frame pipeline uses the fused kernel in
// amt_georef.hip instead.
#include "amt_common.h"

namespace {

using namespace amt;

constexpr int kBlock = 256;

inline dim3 grid_for(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    const int64_t cap = 256 * 16;  // 256 CUs x 16 resident 256-thread blocks of light kernels
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return dim3(static_cast<unsigned>(blocks));
}

#define AMT_GRID_STRIDE(i, n) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void k_directions_tan(tan_wcs w, int width, int height, int corner, double* __restrict__ out) {
    const int cols = width + corner;
    const int64_t n = (int64_t)(height + corner) * cols;
    const double off = corner ? -0.5 : 0.0;
    AMT_GRID_STRIDE(i, n) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        const vec3 d = tan_direction(w, c + off, r + off);
        out[3 * i + 0] = d.x;
        out[3 * i + 1] = d.y;
        out[3 * i + 2] = d.z;
    }
}

// Zenithal projections other than plain TAN, with SIP distortion (reference wcs.py:54-56 hands such headers to
// astropy.wcs.WCS(header).all_pix2world): Calabretta & Greisen 2002, sections 5.1.1-5.1.7, native spherical angles
// (phi, theta) from the intermediate world coordinates (x, y) in degrees, then the native -> celestial rotation.  The SIP
// polynomials f(u, v) = sum A_p_q u^p v^q (Shupe et al. 2005) act on the pixel offsets from CRPIX.
struct zen_args {
    amt_zenithal_wcs w;
};

__device__ __forceinline__ double sip_eval(const double (&c)[AMT_SIP_MAX][AMT_SIP_MAX], int order, double u, double v) {
    double f = 0, up = 1;
    for (int p = 0; p <= order; ++p) {
        double g = 0, vq = 1;
        for (int q = 0; q <= order - p; ++q) {
            g = fma(c[p][q], vq, g);
            vq *= v;
        }
        f = fma(g, up, f);
        up *= u;
    }
    return f;
}

__global__ void k_directions_zenithal(zen_args A, double* __restrict__ out) {
    const amt_zenithal_wcs& w = A.w;
    const int corner = w.corner ? 1 : 0, cols = w.width + corner;
    const int64_t n = (int64_t)(w.height + corner) * cols;
    const double off = corner ? -0.5 : 0.0, k = kRad2Deg;
    AMT_GRID_STRIDE(i, n) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        double u = (double)c + (w.start_x + off) - w.crpix[0] + 1.0, v = (double)r + (w.start_y + off) - w.crpix[1] + 1.0;
        if (w.sip_order_a > 0 || w.sip_order_b > 0) {
            const double fu = w.sip_order_a > 0 ? sip_eval(w.sip_a, w.sip_order_a, u, v) : 0.0;
            const double fv = w.sip_order_b > 0 ? sip_eval(w.sip_b, w.sip_order_b, u, v) : 0.0;
            u += fu, v += fv;
        }
        const double x = w.cd[0] * u + w.cd[1] * v, y = w.cd[2] * u + w.cd[3] * v;
        const double rr = sqrt(x * x + y * y);
        const double phi = atan2(x, -y);
        double theta;
        switch (w.projection) {
            case 0: theta = atan2(k, rr); break;                                   // TAN
            case 1: theta = acos(rr / k); break;                                   // SIN (no slant)
            case 2: theta = (90.0 - rr) * kDeg2Rad; break;                         // ARC
            case 3: theta = 90.0 * kDeg2Rad - 2.0 * atan(rr / (2.0 * k)); break;   // STG
            default: theta = 90.0 * kDeg2Rad - 2.0 * asin(rr / (2.0 * k)); break;  // ZEA
        }
        const double ct = cos(theta);
        const double nx = ct * cos(phi), ny = ct * sin(phi), nz = sin(theta);
        out[3 * i + 0] = w.rot[0] * nx + w.rot[1] * ny + w.rot[2] * nz;
        out[3 * i + 1] = w.rot[3] * nx + w.rot[4] * ny + w.rot[5] * nz;
        out[3 * i + 2] = w.rot[6] * nx + w.rot[7] * ny + w.rot[8] * nz;
    }
}

__global__ void k_directions_tan_points(tan_wcs w, const double* __restrict__ px, const double* __restrict__ py,
                                        int64_t n, double shift, double* __restrict__ out) {
    AMT_GRID_STRIDE(i, n) {
        const vec3 d = tan_direction(w, px[i] - shift, py[i] - shift);
        out[3 * i + 0] = d.x;
        out[3 * i + 1] = d.y;
        out[3 * i + 2] = d.z;
    }
}

__global__ void k_intersect_ellipsoid(ellipsoid_ray e, const double* __restrict__ dirs, int64_t n,
                                      double* __restrict__ out) {
    AMT_GRID_STRIDE(i, n) {
        const vec3 d = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
        const vec3 p = ray_point(e, d, ray_param(e, d));
        out[3 * i + 0] = p.x;
        out[3 * i + 1] = p.y;
        out[3 * i + 2] = p.z;
    }
}

__global__ void k_intersects_ellipsoid(ellipsoid_ray e, const double* __restrict__ dirs, int64_t n,
                                       uint8_t* __restrict__ out) {
    AMT_GRID_STRIDE(i, n) {
        const vec3 d = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
        // reference intersection.py:165-201: directed -> t >= 0, undirected -> discriminant >= 0
        const double dsx = d.x * e.ia, dsy = d.y * e.ia, dsz = d.z * e.ib;
        const double d_o = dsx * e.osx + dsy * e.osy + dsz * e.osz;
        const double d_d = dsx * dsx + dsy * dsy + dsz * dsz;
        const double disc = d_o * d_o - e.oo * d_d + d_d;
        bool hit;
        if (e.directed) {
            const double root = sqrt(disc);
            const double t = e.inside ? d_o + root : d_o - root;
            hit = t >= 0;
        } else {
            hit = disc >= 0;
        }
        out[i] = hit ? 1 : 0;
    }
}

struct sphere_ray {
    double r2, ox, oy, oz, oo;
    int inside, directed;
};

__global__ void k_intersect_sphere(sphere_ray s, const double* __restrict__ dirs, int64_t n,
                                   double* __restrict__ out) {
    AMT_GRID_STRIDE(i, n) {
        // reference intersection.py:26-48 (unit directions)
        const double dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
        const double dp = dx * s.ox + dy * s.oy + dz * s.oz;
        const double root = sqrt(dp * dp - s.oo + s.r2);
        double t;
        if (s.directed) {
            t = s.inside ? -dp + root : -dp - root;
            if (t < 0) t = NAN;
        } else {
            const double t1 = -dp - root, t2 = -dp + root;
            t = fabs(t1) < fabs(t2) ? t1 : t2;
        }
        out[3 * i + 0] = s.ox + t * dx;
        out[3 * i + 1] = s.oy + t * dy;
        out[3 * i + 2] = s.oz + t * dz;
    }
}

__global__ void k_ecef_to_geodetic(bowring w, const double* __restrict__ x, const double* __restrict__ y,
                                   const double* __restrict__ z, int64_t n, double* __restrict__ lat,
                                   double* __restrict__ lon) {
    AMT_GRID_STRIDE(i, n) {
        double la, lo;
        ecef_to_geodetic(w, x[i], y[i], z[i], la, lo);
        lat[i] = la;
        lon[i] = lo;
    }
}

__global__ void k_geodetic_to_ecef(bowring w, const double* __restrict__ lat, const double* __restrict__ lon,
                                   double h, int64_t n, double* __restrict__ x, double* __restrict__ y,
                                   double* __restrict__ z) {
    AMT_GRID_STRIDE(i, n) {
        double xx, yy, zz;
        geodetic_to_ecef(w, lat[i], lon[i], h, xx, yy, zz);
        x[i] = xx;
        y[i] = yy;
        z[i] = zz;
    }
}

__global__ void k_rotate_to_latlon(mat3 m, bowring w, const double* __restrict__ xyz, int64_t n,
                                   double* __restrict__ lat, double* __restrict__ lon) {
    AMT_GRID_STRIDE(i, n) {
        const vec3 p = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        const vec3 g = mul(m, p);
        double la, lo;
        ecef_to_geodetic(w, g.x, g.y, g.z, la, lo);
        lat[i] = la * kRad2Deg;
        lon[i] = lo * kRad2Deg;
    }
}

__global__ void k_rotate_to_mlat_mlt(mat3 m, const double* __restrict__ xyz, int64_t n, double* __restrict__ mlat,
                                     double* __restrict__ mlt) {
    AMT_GRID_STRIDE(i, n) {
        const vec3 p = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        double a, t;
        sm_to_mlat_mlt(mul(m, p), a, t);
        mlat[i] = a;
        mlt[i] = t;
    }
}

__global__ void k_rotate_vectors(mat3 m, const double* __restrict__ xyz, int64_t n, double* __restrict__ out) {
    AMT_GRID_STRIDE(i, n) {
        const vec3 p = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        const vec3 g = mul(m, p);
        out[3 * i + 0] = g.x;
        out[3 * i + 1] = g.y;
        out[3 * i + 2] = g.z;
    }
}

__global__ void k_latlon_to_mlat_mlt(mat3 m, bowring w, const double* __restrict__ lat,
                                     const double* __restrict__ lon, double h, int64_t n,
                                     double* __restrict__ mlat, double* __restrict__ mlt) {
    AMT_GRID_STRIDE(i, n) {
        vec3 g;
        geodetic_to_ecef(w, lat[i] * kDeg2Rad, lon[i] * kDeg2Rad, h, g.x, g.y, g.z);
        double a, t;
        sm_to_mlat_mlt(mul(m, g), a, t);
        mlat[i] = a;
        mlt[i] = t;
    }
}

__global__ void k_sm_to_latlon(mat3 m, bowring w, const double* __restrict__ smlat, const double* __restrict__ smlon,
                               int64_t n, double* __restrict__ lat, double* __restrict__ lon) {
    AMT_GRID_STRIDE(i, n) {
        // reference transform.py:472-485 (unit sphere)
        double sl, cl, so, co;
        sincos(smlat[i] * kDeg2Rad, &sl, &cl);
        sincos(smlon[i] * kDeg2Rad, &so, &co);
        const vec3 s = {cl * co, cl * so, sl};
        const vec3 g = mul(m, s);
        double la, lo;
        ecef_to_geodetic(w, g.x, g.y, g.z, la, lo);
        lat[i] = la * kRad2Deg;
        lon[i] = lo * kRad2Deg;
    }
}

// reference transform.py:104-127 (r, lat, lon) and :38-63 (x, y, z); r may be NULL (= 1 / not wanted)
__global__ void k_cartesian_to_spherical(const double* __restrict__ x, const double* __restrict__ y,
                                         const double* __restrict__ z, int64_t n, double* __restrict__ r,
                                         double* __restrict__ lat, double* __restrict__ lon) {
    AMT_GRID_STRIDE(i, n) {
        const double xx = x[i], yy = y[i], zz = z[i];
        const double s2 = xx * xx + yy * yy;
        if (r) r[i] = sqrt(s2 + zz * zz);
        lat[i] = atan2(zz, sqrt(s2));
        lon[i] = atan2(yy, xx);
    }
}

__global__ void k_spherical_to_cartesian(const double* __restrict__ r, const double* __restrict__ lat,
                                         const double* __restrict__ lon, int64_t n, double* __restrict__ x,
                                         double* __restrict__ y, double* __restrict__ z) {
    AMT_GRID_STRIDE(i, n) {
        double sl, cl, so, co;
        sincos(lat[i], &sl, &cl);
        sincos(lon[i], &so, &co);
        const double rr = r ? r[i] : 1.0;
        x[i] = rr * cl * co;
        y[i] = rr * cl * so;
        z[i] = rr * sl;
    }
}

template <bool DEG>
__global__ void k_rotate_pole(mat3 m, bowring w, double e2, const double* __restrict__ lat, const double* __restrict__ lon,
                              double altitude, int64_t n, double* __restrict__ olat, double* __restrict__ olon) {
    AMT_GRID_STRIDE(i, n) {
        double la, lo;
        if (DEG)
            rotate_pole_deg(w, m, e2, lat[i], lon[i], altitude, la, lo);
        else
            rotate_pole_rad(w, m, e2, lat[i], lon[i], altitude, la, lo);
        olat[i] = la;
        olon[i] = lo;
    }
}

// All-sky equidistant camera -> az/el -> GEO direction -> shell -> geodetic, reference miracle.py:196-258,314-347.
struct allsky_dev {
    int n;                 // points per side
    double off;            // index offset of this point family (0 for corners)
    double xc, yc, inv_k, rotation;
    mat3 to_geo;
    ellipsoid_ray ray;
    bowring bw;
};

__global__ void k_georef_allsky(allsky_dev A, double* __restrict__ az_out, double* __restrict__ el_out,
                                double* __restrict__ dirs, double* __restrict__ lat, double* __restrict__ lon) {
    constexpr double kDeg = 180.0 / M_PI, kRad = M_PI / 180.0;
    const int64_t total = (int64_t)A.n * A.n;
    AMT_GRID_STRIDE(i, total) {
        const int row = (int)(i / A.n), col = (int)(i - (int64_t)row * A.n);
        // miracle.py:328-343: vector from the zenith pixel, x vertical / y horizontal; north = (-1, 0) is the image top;
        // signedAngleBetween(v, north) = atan2(v0*n1 - v1*n0, v0*n0 + v1*n1) (utils.py:48-56)
        const double v0 = ((double)row + A.off) - A.xc, v1 = ((double)col + A.off) - A.yc;
        double az = (atan2(v0 * 0.0 - v1 * -1.0, v0 * -1.0 + v1 * 0.0) - A.rotation) * kDeg;
        az -= floor(az / 360.0) * 360.0;       // Angle.wrap_at(360 deg): into [0, 360)
        if (az >= 360.0) az -= 360.0;
        if (az < 0.0) az += 360.0;
        const double el = 90.0 - sqrt(v0 * v0 + v1 * v1) * A.inv_k * kDeg;
        if (az_out) az_out[i] = az;
        if (el_out) el_out[i] = el;
        if (!dirs && !lat && !lon) continue;
        // miracle.py:239-258: local (el, -(az-180)) on the unit sphere, then latitude and longitude rotation
        double se, ce, sa, ca;
        sincos(el * kRad, &se, &ce);
        sincos(-(az - 180.0) * kRad, &sa, &ca);
        vec3 local;
        local.x = ce * ca;
        local.y = ce * sa;
        local.z = se;
        const vec3 d = mul(A.to_geo, local);
        if (dirs) {
            dirs[3 * i + 0] = d.x;
            dirs[3 * i + 1] = d.y;
            dirs[3 * i + 2] = d.z;
        }
        if (lat || lon) {
            const vec3 hit = ray_point(A.ray, d, ray_param(A.ray, d));
            double la, lo;
            ecef_to_geodetic(A.bw, hit.x, hit.y, hit.z, la, lo);
            if (lat) lat[i] = la * kDeg;
            if (lon) lon[i] = lo * kDeg;
        }
    }
}

// reference themis.py:224-253 reproject, one thread per coordinate
__global__ void k_reproject_altitude(bowring w, vec3 station, ellipsoid_ray ray, double h_ref,
                                     const double* __restrict__ lat_ref, const double* __restrict__ lon_ref, int64_t n,
                                     double* __restrict__ lat, double* __restrict__ lon) {
    constexpr double kDeg = 180.0 / M_PI, kRad = M_PI / 180.0;
    AMT_GRID_STRIDE(i, n) {
        double x, y, z;
        geodetic_to_ecef(w, lat_ref[i] * kRad, lon_ref[i] * kRad, h_ref, x, y, z);
        vec3 d;
        d.x = x - station.x;
        d.y = y - station.y;
        d.z = z - station.z;
        const vec3 hit = ray_point(ray, d, ray_param(ray, d));
        double la, lo;
        ecef_to_geodetic(w, hit.x, hit.y, hit.z, la, lo);
        lat[i] = la * kDeg;
        lon[i] = lo * kDeg;
    }
}

}  // namespace

extern "C" {

int amt_directions_tan(amt_ctx* ctx, const amt_frame_params* p, int corner, double* out_dirs) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, p && out_dirs, "NULL argument");
    AMT_REQUIRE(ctx, p->width > 0 && p->height > 0, "empty frame");
    corner = corner ? 1 : 0;
    const tan_wcs w = make_tan_wcs(p);
    const int64_t n = (int64_t)(p->height + corner) * (p->width + corner);
    hipLaunchKernelGGL(k_directions_tan, grid_for(n), dim3(kBlock), 0, ctx->stream, w, p->width, p->height, corner,
                       out_dirs);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_directions_zenithal(amt_ctx* ctx, const amt_zenithal_wcs* w, double* out_dirs) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, w && out_dirs, "NULL argument");
    AMT_REQUIRE(ctx, w->width > 0 && w->height > 0, "empty frame");
    AMT_REQUIRE(ctx, w->projection >= 0 && w->projection <= 4, "projection must be 0 TAN, 1 SIN, 2 ARC, 3 STG or 4 ZEA");
    AMT_REQUIRE(ctx, w->sip_order_a >= 0 && w->sip_order_a < AMT_SIP_MAX && w->sip_order_b >= 0 && w->sip_order_b < AMT_SIP_MAX,
                "SIP order out of range");
    zen_args A;
    A.w = *w;
    const int corner = w->corner ? 1 : 0;
    const int64_t n = (int64_t)(w->height + corner) * (w->width + corner);
    hipLaunchKernelGGL(k_directions_zenithal, grid_for(n), dim3(kBlock), 0, ctx->stream, A, out_dirs);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_directions_tan_points(amt_ctx* ctx, const amt_frame_params* p, const double* px, const double* py, int64_t n,
                              int origin, double* out_dirs) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, p && (n == 0 || (px && py && out_dirs)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && (origin == 0 || origin == 1), "bad size or origin");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_directions_tan_points, grid_for(n), dim3(kBlock), 0, ctx->stream, make_tan_wcs(p), px, py, n,
                       (double)origin, out_dirs);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_intersect_ellipsoid(amt_ctx* ctx, double a, double b, const double* origin, const double* dirs,
                            int64_t n, int directed, double* out_xyz) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, origin && (n == 0 || (dirs && out_xyz)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a > 0 && b > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_intersect_ellipsoid, grid_for(n), dim3(kBlock), 0, ctx->stream,
                       make_ray(a, b, origin, directed ? 1 : 0), dirs, n, out_xyz);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_intersects_ellipsoid(amt_ctx* ctx, double a, double b, const double* origin, const double* dirs,
                             int64_t n, int directed, uint8_t* out_hit) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, origin && (n == 0 || (dirs && out_hit)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a > 0 && b > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_intersects_ellipsoid, grid_for(n), dim3(kBlock), 0, ctx->stream,
                       make_ray(a, b, origin, directed ? 1 : 0), dirs, n, out_hit);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_intersect_sphere(amt_ctx* ctx, double radius, const double* origin, const double* dirs, int64_t n,
                         int directed, double* out_xyz) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, origin && (n == 0 || (dirs && out_xyz)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && radius > 0, "bad size or radius");
    if (n == 0) return AMT_OK;
    sphere_ray s;
    s.r2 = radius * radius;
    s.ox = origin[0];
    s.oy = origin[1];
    s.oz = origin[2];
    s.oo = s.ox * s.ox + s.oy * s.oy + s.oz * s.oz;
    s.inside = std::sqrt(s.oo) < radius;
    s.directed = directed ? 1 : 0;
    hipLaunchKernelGGL(k_intersect_sphere, grid_for(n), dim3(kBlock), 0, ctx->stream, s, dirs, n, out_xyz);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_ecef_to_geodetic(amt_ctx* ctx, const double* x, const double* y, const double* z, int64_t n, double a,
                         double b, double* out_lat, double* out_lon) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n == 0 || (x && y && z && out_lat && out_lon), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a > 0 && b > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_ecef_to_geodetic, grid_for(n), dim3(kBlock), 0, ctx->stream, make_bowring(a, b), x, y, z, n,
                       out_lat, out_lon);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_geodetic_to_ecef(amt_ctx* ctx, const double* lat, const double* lon, double h, int64_t n, double a,
                         double b, double* out_x, double* out_y, double* out_z) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n == 0 || (lat && lon && out_x && out_y && out_z), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a > 0 && b > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_geodetic_to_ecef, grid_for(n), dim3(kBlock), 0, ctx->stream, make_bowring(a, b), lat, lon, h,
                       n, out_x, out_y, out_z);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_rotate_to_latlon(amt_ctx* ctx, const double* m, const double* xyz, int64_t n, double a0, double b0,
                         double* out_lat_deg, double* out_lon_deg) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, m && (n == 0 || (xyz && out_lat_deg && out_lon_deg)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a0 > 0 && b0 > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_rotate_to_latlon, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(m),
                       make_bowring(a0, b0), xyz, n, out_lat_deg, out_lon_deg);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_rotate_to_mlat_mlt(amt_ctx* ctx, const double* m, const double* xyz, int64_t n, double* out_mlat_deg,
                           double* out_mlt_h) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, m && (n == 0 || (xyz && out_mlat_deg && out_mlt_h)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0, "negative size");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_rotate_to_mlat_mlt, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(m), xyz, n,
                       out_mlat_deg, out_mlt_h);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_rotate_vectors(amt_ctx* ctx, const double* m, const double* xyz, int64_t n, double* out_xyz) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, m && (n == 0 || (xyz && out_xyz)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0, "negative size");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_rotate_vectors, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(m), xyz, n, out_xyz);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_latlon_to_mlat_mlt(amt_ctx* ctx, const double* m, const double* lat_deg, const double* lon_deg, double h,
                           int64_t n, double a0, double b0, double* out_mlat_deg, double* out_mlt_h) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, m && (n == 0 || (lat_deg && lon_deg && out_mlat_deg && out_mlt_h)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a0 > 0 && b0 > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_latlon_to_mlat_mlt, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(m),
                       make_bowring(a0, b0), lat_deg, lon_deg, h, n, out_mlat_deg, out_mlt_h);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_sm_to_latlon(amt_ctx* ctx, const double* m_sm_to_geo, const double* smlat_deg, const double* smlon_deg,
                     int64_t n, double a0, double b0, double* out_lat_deg, double* out_lon_deg) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, m_sm_to_geo && (n == 0 || (smlat_deg && smlon_deg && out_lat_deg && out_lon_deg)),
                "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a0 > 0 && b0 > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_sm_to_latlon, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(m_sm_to_geo),
                       make_bowring(a0, b0), smlat_deg, smlon_deg, n, out_lat_deg, out_lon_deg);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_cartesian_to_spherical(amt_ctx* ctx, const double* x, const double* y, const double* z, int64_t n,
                               double* out_r, double* out_lat, double* out_lon) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n == 0 || (x && y && z && out_lat && out_lon), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0, "negative size");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_cartesian_to_spherical, grid_for(n), dim3(kBlock), 0, ctx->stream, x, y, z, n, out_r, out_lat,
                       out_lon);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_spherical_to_cartesian(amt_ctx* ctx, const double* r, const double* lat, const double* lon, int64_t n,
                               double* out_x, double* out_y, double* out_z) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n == 0 || (lat && lon && out_x && out_y && out_z), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0, "negative size");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(k_spherical_to_cartesian, grid_for(n), dim3(kBlock), 0, ctx->stream, r, lat, lon, n, out_x,
                       out_y, out_z);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_rotate_pole(amt_ctx* ctx, const double* rot, const double* lat, const double* lon, double altitude,
                    int64_t n, double a0, double b0, double* out_lat, double* out_lon) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, rot && (n == 0 || (lat && lon && out_lat && out_lon)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a0 > 0 && b0 > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    const bowring w = make_bowring(a0, b0);
    hipLaunchKernelGGL(k_rotate_pole<false>, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(rot), w, w.e2a / w.a, lat, lon,
                       altitude, n, out_lat, out_lon);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_rotate_pole_deg(amt_ctx* ctx, const double* rot, const double* lat_deg, const double* lon_deg, double altitude,
                        int64_t n, double a0, double b0, double* out_lat_deg, double* out_lon_deg) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, rot && (n == 0 || (lat_deg && lon_deg && out_lat_deg && out_lon_deg)), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a0 > 0 && b0 > 0, "bad size or axes");
    if (n == 0) return AMT_OK;
    const bowring w = make_bowring(a0, b0);
    hipLaunchKernelGGL(k_rotate_pole<true>, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(rot), w, w.e2a / w.a, lat_deg,
                       lon_deg, altitude, n, out_lat_deg, out_lon_deg);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_georef_allsky(amt_ctx* ctx, const amt_allsky_params* p, int corner, double* az_deg, double* el_deg,
                      double* dirs, double* lat_deg, double* lon_deg) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, p != nullptr, "NULL argument");
    AMT_REQUIRE(ctx, p->size > 0 && p->size < 32768, "bad image size");
    AMT_REQUIRE(ctx, p->k > 0 && p->a > 0 && p->b > 0 && p->a0 > 0 && p->b0 > 0, "bad calibration or axes");
    if (!az_deg && !el_deg && !dirs && !lat_deg && !lon_deg) return AMT_OK;
    allsky_dev A;
    A.n = p->size + (corner ? 1 : 0);
    A.off = corner ? 0.0 : p->center_offset;
    A.xc = p->xc;
    A.yc = p->yc;
    A.inv_k = 1.0 / p->k;
    A.rotation = p->rotation;
    A.to_geo = make_mat3(p->to_geo);
    A.ray = make_ray(p->a, p->b, p->station, 1);
    A.bw = make_bowring(p->a0, p->b0);
    hipLaunchKernelGGL(k_georef_allsky, grid_for((int64_t)A.n * A.n), dim3(kBlock), 0, ctx->stream, A, az_deg, el_deg,
                       dirs, lat_deg, lon_deg);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_reproject_altitude(amt_ctx* ctx, double station_lat_deg, double station_lon_deg, const double* lat_ref_deg,
                           const double* lon_ref_deg, int64_t n, double height_ref, double height_new, double a0,
                           double b0, double* out_lat_deg, double* out_lon_deg) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n == 0 || (lat_ref_deg && lon_ref_deg && out_lat_deg && out_lon_deg), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && a0 > 0 && b0 > 0, "bad size or axes");
    AMT_REQUIRE(ctx, a0 + height_new > 0 && b0 + height_new > 0, "bad height");
    if (n == 0) return AMT_OK;
    // geodetic2EcefZero of the station (transform.py:180-197)
    const double kRad = M_PI / 180.0;
    const double e2 = (a0 * a0 - b0 * b0) / (a0 * a0);
    const double sl = sin(station_lat_deg * kRad), cl = cos(station_lat_deg * kRad);
    const double nn = a0 / sqrt(1 - e2 * sl * sl);
    const double o[3] = {nn * cl * cos(station_lon_deg * kRad), nn * cl * sin(station_lon_deg * kRad),
                         nn * (1 - e2) * sl};
    vec3 station;
    station.x = o[0];
    station.y = o[1];
    station.z = o[2];
    hipLaunchKernelGGL(k_reproject_altitude, grid_for(n), dim3(kBlock), 0, ctx->stream, make_bowring(a0, b0), station,
                       make_ray(a0 + height_new, b0 + height_new, o, 1), height_ref, lat_ref_deg, lon_ref_deg, n,
                       out_lat_deg, out_lon_deg);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

}  // extern "C"
