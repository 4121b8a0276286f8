// Fused per-frame georeferencing kernel (reference auromat/mapping/astrometry.py:49-212).
//
// One workgroup owns a TW x TH tile of pixels:
//   phase 1  every corner of the tile ((TW+1) x (TH+1), one thread per corner, strided):
//            WCS pixel -> unit direction -> ray/ellipsoid hit -> [LDS: P, d, lat, lon]
//            -> J2000->GEO rotation -> Bowring -> lat/lon (deg) written by the tile that owns the corner
//   phase 2  every pixel of the tile: centre point = mean of the 4 corner hits read back from LDS
//            (fast mode) or its own ray cast (exact mode) -> lat/lon, elevation, optional MLat/MLT,
//            and the bounding box of the corners of centres above the elevation threshold.
// Each ray is therefore cast once per tile (1.14x redundancy at 64x8), nothing but the final
// arrays touches HBM, and all per-frame constants travel as kernel arguments (SGPRs).
//
// Algorithmic HBM bytes per frame: 16 B per corner + 24 B per pixel written, nothing read
// ("WCS-fused" row of SURVEY.md §8d); with caller-supplied directions +24 B per corner read.
#include "amt_common.h"

namespace {

using namespace amt;

struct georef_args {
    tan_wcs wcs;
    ellipsoid_ray ray;
    mat3 m_geo;
    mat3 m_sm;
    bowring bw;
    int width, height;
    const double* dirs_in;   // optional (H+1, W+1, 3)
    double* lat;
    double* lon;
    double* lat_c;
    double* lon_c;
    double* elev;
    double* mlat;
    double* mlt;
    double* mlat_c;
    double* mlt_c;
    double* bbox_partials;   // [nblocks][8] or NULL
    double bbox_min_elev;
};

constexpr int kThreads = 256;
constexpr double kInf = __builtin_huge_val();

__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int TW, int TH, bool FAST, bool DIRS_IN>
__global__ __launch_bounds__(kThreads) void k_georef(georef_args A) {
    constexpr int CW = TW + 1, CH = TH + 1, NC = CW * CH;
    __shared__ double sP[3][NC];
    __shared__ double sD[3][NC];
    __shared__ double sLat[NC];
    __shared__ double sLon[NC];
    __shared__ double sRed[8][kThreads / 64];

    const int tiles_x = (A.width + TW - 1) / TW;
    const int tile_y = blockIdx.x / tiles_x;
    const int tile_x = blockIdx.x - tile_y * tiles_x;
    const int x0 = tile_x * TW, y0 = tile_y * TH;
    const int W1 = A.width + 1;
    const bool want_mag = A.mlat != nullptr || A.mlat_c != nullptr;

    // ---- phase 1: corners -------------------------------------------------------------
    for (int c = threadIdx.x; c < NC; c += kThreads) {
        const int cy = c / CW, cx = c - cy * CW;
        const int gx = x0 + cx, gy = y0 + cy;
        vec3 d = {NAN, NAN, NAN}, p = {NAN, NAN, NAN};
        double la = NAN, lo = NAN;
        if (gx <= A.width && gy <= A.height) {
            const int64_t gi = (int64_t)gy * W1 + gx;
            if (DIRS_IN) {
                d.x = A.dirs_in[3 * gi];
                d.y = A.dirs_in[3 * gi + 1];
                d.z = A.dirs_in[3 * gi + 2];
            } else {
                d = tan_direction(A.wcs, gx - 0.5, gy - 0.5);
            }
            const double t = ray_param(A.ray, d);
            const bool hit = t == t;
            const bool owner = (cx < TW || gx == A.width) && (cy < TH || gy == A.height);
            if (hit) {
                p = ray_point(A.ray, d, t);
                const vec3 g = mul(A.m_geo, p);
                ecef_to_geodetic(A.bw, g.x, g.y, g.z, la, lo);
                la *= kRad2Deg;
                lo *= kRad2Deg;
            }
            if (owner) {
                if (A.lat) A.lat[gi] = la;
                if (A.lon) A.lon[gi] = lo;
                if (A.mlat) {
                    double ml = NAN, mt = NAN;
                    if (hit) sm_to_mlat_mlt(mul(A.m_sm, p), ml, mt);
                    A.mlat[gi] = ml;
                    A.mlt[gi] = mt;
                }
            }
        }
        sP[0][c] = p.x;
        sP[1][c] = p.y;
        sP[2][c] = p.z;
        sD[0][c] = d.x;
        sD[1][c] = d.y;
        sD[2][c] = d.z;
        sLat[c] = la;
        sLon[c] = lo;
    }
    __syncthreads();

    // ---- phase 2: centres ---------------------------------------------------------------
    double bla0 = kInf, bla1 = -kInf, blo0 = kInf, blo1 = -kInf, blop = kInf, blon = -kInf, bcnt = 0, bpole = 0;
    for (int q = threadIdx.x; q < TW * TH; q += kThreads) {
        const int py = q / TW, px = q - py * TW;
        const int gx = x0 + px, gy = y0 + py;
        if (gx >= A.width || gy >= A.height) continue;
        const int c00 = py * CW + px, c01 = c00 + 1, c10 = c00 + CW, c11 = c10 + 1;
        vec3 p, d;
        if (FAST) {
            // reference astrometry.py:154-160: ((c00 + c01) + c11) + c10, then /4
            p.x = (((sP[0][c00] + sP[0][c01]) + sP[0][c11]) + sP[0][c10]) * 0.25;
            p.y = (((sP[1][c00] + sP[1][c01]) + sP[1][c11]) + sP[1][c10]) * 0.25;
            p.z = (((sP[2][c00] + sP[2][c01]) + sP[2][c11]) + sP[2][c10]) * 0.25;
            d.x = (((sD[0][c00] + sD[0][c01]) + sD[0][c11]) + sD[0][c10]) * 0.25;
            d.y = (((sD[1][c00] + sD[1][c01]) + sD[1][c11]) + sD[1][c10]) * 0.25;
            d.z = (((sD[2][c00] + sD[2][c01]) + sD[2][c11]) + sD[2][c10]) * 0.25;
        } else {
            d = tan_direction(A.wcs, (double)gx, (double)gy);
            p = ray_point(A.ray, d, ray_param(A.ray, d));
        }
        const int64_t gi = (int64_t)gy * A.width + gx;
        double la = NAN, lo = NAN, el = NAN, ml = NAN, mt = NAN;
        if (p.x == p.x) {
            const vec3 g = mul(A.m_geo, p);
            ecef_to_geodetic(A.bw, g.x, g.y, g.z, la, lo);
            la *= kRad2Deg;
            lo *= kRad2Deg;
            // reference astrometry.py:200-212, utils.py:33-46: angle between -d and P/|P|
            const double inv = rsqrt(dot(p, p));
            double cosang = -(d.x * p.x + d.y * p.y + d.z * p.z) * inv;
            cosang = fmin(1.0, fmax(-1.0, cosang));
            el = 90.0 - acos(cosang) * kRad2Deg;
            if (want_mag) sm_to_mlat_mlt(mul(A.m_sm, p), ml, mt);
        }
        if (A.lat_c) A.lat_c[gi] = la;
        if (A.lon_c) A.lon_c[gi] = lo;
        if (A.elev) A.elev[gi] = el;
        if (A.mlat_c) {
            A.mlat_c[gi] = ml;
            A.mlt_c[gi] = mt;
        }
        if (A.bbox_partials) {
            // corners kept after maskedByElevation + sanitisation = corners of valid centres
            // (reference mapping.py:845-864,1063-1125); in exact mode a centre also needs its 4 corners
            const double a0 = sLat[c00], a1 = sLat[c01], a2 = sLat[c10], a3 = sLat[c11];
            const bool ok = (el >= A.bbox_min_elev) && (FAST || (a0 == a0 && a1 == a1 && a2 == a2 && a3 == a3));
            if (ok) {
                const double o0 = sLon[c00], o1 = sLon[c01], o2 = sLon[c10], o3 = sLon[c11];
                bla0 = fmin(bla0, fmin(fmin(a0, a1), fmin(a2, a3)));
                bla1 = fmax(bla1, fmax(fmax(a0, a1), fmax(a2, a3)));
                blo0 = fmin(blo0, fmin(fmin(o0, o1), fmin(o2, o3)));
                blo1 = fmax(blo1, fmax(fmax(o0, o1), fmax(o2, o3)));
                blop = fmin(blop, fmin(fmin(o0 > 0 ? o0 : kInf, o1 > 0 ? o1 : kInf),
                                       fmin(o2 > 0 ? o2 : kInf, o3 > 0 ? o3 : kInf)));
                blon = fmax(blon, fmax(fmax(o0 > 0 ? -kInf : o0, o1 > 0 ? -kInf : o1),
                                       fmax(o2 > 0 ? -kInf : o2, o3 > 0 ? -kInf : o3)));
                bcnt += 1;
                bpole += quad_winds_pole(o0, o1, o3, o2) ? 1.0 : 0.0;
            }
        }
    }

    if (A.bbox_partials) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        bla0 = wave_min(bla0);
        bla1 = wave_max(bla1);
        blo0 = wave_min(blo0);
        blo1 = wave_max(blo1);
        blop = wave_min(blop);
        blon = wave_max(blon);
        bcnt = wave_sum(bcnt);
        bpole = wave_sum(bpole);
        if (lane == 0) {
            sRed[0][wave] = bla0;
            sRed[1][wave] = bla1;
            sRed[2][wave] = blo0;
            sRed[3][wave] = blo1;
            sRed[4][wave] = blop;
            sRed[5][wave] = blon;
            sRed[6][wave] = bcnt;
            sRed[7][wave] = bpole;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int k = threadIdx.x;
            double v = sRed[k][0];
            for (int w = 1; w < kThreads / 64; ++w) {
                const double o = sRed[k][w];
                v = (k >= 6) ? v + o : ((k == 0 || k == 2 || k == 4) ? fmin(v, o) : fmax(v, o));
            }
            A.bbox_partials[(int64_t)blockIdx.x * 8 + k] = v;
        }
    }
}

// Folds the per-workgroup bbox partials ([n][8]) into bbox[8].
__global__ __launch_bounds__(1024) void k_bbox_fold(const double* __restrict__ partials, int n, double* __restrict__ bbox) {
    __shared__ double sRed[8][16];
    double v[8] = {kInf, -kInf, kInf, -kInf, kInf, -kInf, 0, 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double* q = partials + (int64_t)i * 8;
        v[0] = fmin(v[0], q[0]);
        v[1] = fmax(v[1], q[1]);
        v[2] = fmin(v[2], q[2]);
        v[3] = fmax(v[3], q[3]);
        v[4] = fmin(v[4], q[4]);
        v[5] = fmax(v[5], q[5]);
        v[6] += q[6];
        v[7] += q[7];
    }
    v[0] = wave_min(v[0]);
    v[1] = wave_max(v[1]);
    v[2] = wave_min(v[2]);
    v[3] = wave_max(v[3]);
    v[4] = wave_min(v[4]);
    v[5] = wave_max(v[5]);
    v[6] = wave_sum(v[6]);
    v[7] = wave_sum(v[7]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) sRed[k][wave] = v[k];
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        double r = sRed[k][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            const double o = sRed[k][w];
            r = (k >= 6) ? r + o : ((k == 0 || k == 2 || k == 4) ? fmin(r, o) : fmax(r, o));
        }
        bbox[k] = r;
    }
}

constexpr int kTW = 64, kTH = 8;

int launch_georef(amt_ctx* ctx, const amt_frame_params* p, const double* dirs, const amt_georef_out* out) {
    AMT_REQUIRE(ctx, p && out, "NULL argument");
    AMT_REQUIRE(ctx, p->width > 0 && p->height > 0, "empty frame");
    AMT_REQUIRE(ctx, p->a > 0 && p->b > 0 && p->a0 > 0 && p->b0 > 0, "ellipsoid axes must be positive");
    AMT_REQUIRE(ctx, (out->mlat == nullptr) == (out->mlt == nullptr), "mlat and mlt must be given together");
    AMT_REQUIRE(ctx, (out->mlat_c == nullptr) == (out->mlt_c == nullptr), "mlat_c and mlt_c must be given together");
    AMT_REQUIRE(ctx, dirs == nullptr || p->fast_center, "caller-supplied directions need fast_center");
    georef_args A;
    A.wcs = make_tan_wcs(p);
    A.ray = make_ray(p->a, p->b, p->cam, 1);
    A.m_geo = make_mat3(p->m_geo);
    A.m_sm = make_mat3(p->m_sm);
    A.bw = make_bowring(p->a0, p->b0);
    A.width = p->width;
    A.height = p->height;
    A.dirs_in = dirs;
    A.lat = out->lat;
    A.lon = out->lon;
    A.lat_c = out->lat_c;
    A.lon_c = out->lon_c;
    A.elev = out->elev;
    A.mlat = out->mlat;
    A.mlt = out->mlt;
    A.mlat_c = out->mlat_c;
    A.mlt_c = out->mlt_c;
    A.bbox_min_elev = out->bbox_min_elevation;
    const int tiles_x = (p->width + kTW - 1) / kTW, tiles_y = (p->height + kTH - 1) / kTH;
    const int64_t nblocks = (int64_t)tiles_x * tiles_y;
    AMT_REQUIRE(ctx, nblocks < (1ll << 31), "frame too large");
    A.bbox_partials = nullptr;
    if (out->bbox) {
        A.bbox_partials = static_cast<double*>(amt_workspace(ctx, (size_t)nblocks * 8 * sizeof(double)));
        if (A.bbox_partials == nullptr) {
            ctx->last_error = "amt_georef_frame: workspace allocation failed";
            return AMT_ENOMEM;
        }
    }
    const dim3 grid((unsigned)nblocks), block(kThreads);
    if (dirs) {
        hipLaunchKernelGGL((k_georef<kTW, kTH, true, true>), grid, block, 0, ctx->stream, A);
    } else if (p->fast_center) {
        hipLaunchKernelGGL((k_georef<kTW, kTH, true, false>), grid, block, 0, ctx->stream, A);
    } else {
        hipLaunchKernelGGL((k_georef<kTW, kTH, false, false>), grid, block, 0, ctx->stream, A);
    }
    AMT_LAUNCH_CHECK(ctx);
    if (out->bbox) {
        hipLaunchKernelGGL(k_bbox_fold, dim3(1), dim3(1024), 0, ctx->stream, A.bbox_partials, (int)nblocks, out->bbox);
        AMT_LAUNCH_CHECK(ctx);
    }
    return AMT_OK;
}

}  // namespace

extern "C" {

int amt_georef_frame(amt_ctx* ctx, const amt_frame_params* p, const amt_georef_out* out) {
    AMT_CHECK_CTX(ctx);
    return launch_georef(ctx, p, nullptr, out);
}

int amt_georef_frame_dirs(amt_ctx* ctx, const amt_frame_params* p, const double* corner_dirs,
                          const amt_georef_out* out) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, corner_dirs != nullptr, "corner_dirs is NULL");
    return launch_georef(ctx, p, corner_dirs, out);
}

}  // extern "C"
