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

__global__ void k_rotate_pole(mat3 m, bowring w, const double* __restrict__ lat, const double* __restrict__ lon,
                              double altitude, int64_t n, double* __restrict__ olat, double* __restrict__ olon) {
    AMT_GRID_STRIDE(i, n) {
        vec3 g;
        geodetic_to_ecef(w, lat[i], lon[i], altitude, g.x, g.y, g.z);
        const vec3 r = mul(m, g);
        double la, lo;
        ecef_to_geodetic(w, r.x, r.y, r.z, la, lo);
        olat[i] = la;
        olon[i] = lo;
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
    hipLaunchKernelGGL(k_rotate_pole, grid_for(n), dim3(kBlock), 0, ctx->stream, make_mat3(rot), make_bowring(a0, b0),
                       lat, lon, altitude, n, out_lat, out_lon);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

}  // extern "C"
