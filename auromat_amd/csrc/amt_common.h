// Shared host/device helpers of libauromat_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/auromat_hip.h"

struct amt_copier;                       // staged host <-> device copies (amt_copy.hip)
void amt_copier_destroy(amt_copier* c);

struct amt_ctx {
    int device;
    amt_copier* copier;
    hipStream_t stream;
    bool owns_stream;
    double* scratch;        // small device scratch (counters)
    // grow-only device workspaces (per-block partial reductions, sort buffers), ONE PER STREAM the context has been
    // used on: kernels of different streams run concurrently and must not share scratch memory
    struct workspace {
        hipStream_t stream;
        void* ptr;
        size_t bytes;
        unsigned long long last_use;
    };
    std::vector<workspace> workspaces;
    unsigned long long workspace_clock;
    std::string last_error;
    // side streams shared by all frame drivers of this context (created by the first amt_pipe_create): every
    // extra stream competes for the few hardware queues of the process, and streams that share a hardware queue
    // are served in order, so a kernel can get stuck behind another stream's wait
    hipStream_t aux_pre, aux_tail, aux_fin;     // coarse pre-pass | folds behind a big kernel | crop/finalise
    // optional per-kernel timing (amt_timing_*): event pairs recorded around the dominant kernels
    int last_second, last_bin, last_frames;     // amt_georef_last_variant
    int timing;                       // 0 = off, n = bracket every n-th launch of each kind
    size_t tlaunch[2];                // launches seen per kind since timing was enabled
    int tframes[2];                   // frames covered by the timed launches
    std::vector<hipEvent_t> tev[2];   // [kernel kind] start0, stop0, start1, stop1, ...
    size_t tused[2];
};

constexpr int kTimingMaxLaunches = 4096;

// Event pair for the next launch of kernel kind `kind` (0 georef, 1 bin) when that launch is to be timed,
// else two NULLs.  The events are attached to the dispatch itself (hipExtLaunchKernelGGL start/stop events),
// so timing puts no extra packet on the stream; they are created lazily and reused.
static inline void amt_timing_pair(amt_ctx* ctx, int kind, int frames, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!ctx->timing) return;
    if ((ctx->tlaunch[kind]++ % (size_t)ctx->timing) != 0) return;
    if (ctx->tused[kind] + 2 > (size_t)2 * kTimingMaxLaunches) return;
    while (ctx->tused[kind] + 2 > ctx->tev[kind].size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        ctx->tev[kind].push_back(e);
    }
    *start = ctx->tev[kind][ctx->tused[kind]];
    *stop = ctx->tev[kind][ctx->tused[kind] + 1];
    ctx->tused[kind] += 2;
    ctx->tframes[kind] += frames;      // a launch can cover several frames (amt_pipe_launch_many)
}

// ---- internal interfaces between translation units (not part of the C ABI) ---------------------------
// Where the bounding-box folds of a georeferencing launch run.  The frame driver (amt_pipe.hip) keeps them
// off the context's stream so that the next frame's kernel follows the previous one without a gap.
struct amt_georef_tail {
    hipStream_t stream;        // the folds run here, after `kernel_done` (recorded on the context's stream)
    hipEvent_t kernel_done;
    double* partials;          // buffer of the per-wave partial boxes, amt_georef_partials_bytes() large
    size_t partials_bytes;
};
size_t amt_georef_partials_bytes(const amt_frame_params* p);
int amt_georef_launch(amt_ctx* ctx, const amt_frame_params* p, const double* dirs, const amt_georef_out* out,
                      const amt_georef_tail* tail);
// n <= AMT_MAX_BATCH equally sized frames in ONE launch of the big kernel (falls back to n launches otherwise)
#define AMT_MAX_BATCH 3
int amt_georef_launch_many(amt_ctx* ctx, int n, const amt_frame_params* const* p, const amt_georef_out* const* out,
                           const amt_georef_tail* const* tail);
// the same with caller-supplied corner directions, one array per frame (all frames or none)
int amt_georef_launch_many_dirs(amt_ctx* ctx, int n, const amt_frame_params* const* p, const double* const* dirs,
                                const amt_georef_out* const* out, const amt_georef_tail* const* tail);
// k_apply_bin_events on `stream` (before the finalise kernel): see bin_event
int amt_bin_apply_events_on(amt_ctx* ctx, hipStream_t stream, const void* events, uint32_t* count, uint64_t* acc,
                            int32_t acc_nx, int32_t acc_ny, int32_t off_x, int32_t off_y, int32_t nx, int32_t ny);
// amt_bin_frame_finalize_window on `stream`; clear != 0 also zeroes every cell of the accumulator grid.
int amt_bin_finalize_on(amt_ctx* ctx, hipStream_t stream, uint64_t* acc, int32_t acc_nx, int32_t acc_ny, int32_t off_x,
                        int32_t off_y, int32_t nx, int32_t ny, int32_t nchan, int32_t img_dtype, double* mean,
                        void* out_img, uint8_t* out_mask, double* out_count, int clear);

// one frame of a batched finalise step (amt_pipe_finalize_many -> k_pipe_finish)
struct finish_frame {
    const void* events;        // bin_event records
    unsigned int* count;
    unsigned long long* acc;
    int acc_nx, acc_ny, off_x, off_y, nx, ny;
    unsigned int n_events;
    int pad_;
    double* mean;
    void* img;
    uint8_t* mask;
    double* out_count;
};
struct finish_batch {
    finish_frame f[3];
    int n, pad_;
};
int amt_pipe_finish_on(amt_ctx* ctx, hipStream_t stream, const finish_batch& B, int32_t img_dtype);

// Returns the device workspace of the context's CURRENT stream, at least `bytes` large (grow-only; reallocation
// synchronises that stream).  The Python host switches the context between torch streams (frame k is binned on one
// while frame k+1 is georeferenced on another), so scratch memory is kept per stream.
// At most kMaxWorkspaces streams keep one: a host that keeps creating streams (torch does, per thread) would otherwise
// leak a megabyte per stream ever seen; the least recently used one is synchronised and freed.
constexpr size_t kMaxWorkspaces = 16;
static inline void* amt_workspace(amt_ctx* ctx, size_t bytes) {
    amt_ctx::workspace* w = nullptr;
    for (auto& e : ctx->workspaces)
        if (e.stream == ctx->stream) w = &e;
    if (w == nullptr) {
        if (ctx->workspaces.size() >= kMaxWorkspaces) {
            size_t lru = 0;
            for (size_t i = 1; i < ctx->workspaces.size(); ++i)
                if (ctx->workspaces[i].last_use < ctx->workspaces[lru].last_use) lru = i;
            amt_ctx::workspace& old = ctx->workspaces[lru];
            if (old.ptr) {
                // (the stream may have been destroyed by its owner: then everything on it has completed)
                (void)hipStreamSynchronize(old.stream);
                (void)hipGetLastError();
                (void)hipFree(old.ptr);
            }
            ctx->workspaces.erase(ctx->workspaces.begin() + (long)lru);
        }
        ctx->workspaces.push_back({ctx->stream, nullptr, 0, 0});
        w = &ctx->workspaces.back();
    }
    w->last_use = ++ctx->workspace_clock;
    if (bytes <= w->bytes) return w->ptr;
    if (w->ptr) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(w->ptr);
        w->ptr = nullptr;
        w->bytes = 0;
    }
    const size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
    if (hipMalloc(&w->ptr, cap) != hipSuccess) {
        w->ptr = nullptr;
        return nullptr;
    }
    w->bytes = cap;
    return w->ptr;
}

#define AMT_CHECK_CTX(ctx) \
    if ((ctx) == nullptr) return AMT_EINVAL

#define AMT_REQUIRE(ctx, cond, msg)                           \
    do {                                                      \
        if (!(cond)) {                                        \
            (ctx)->last_error = std::string(__func__) + ": " + (msg); \
            return AMT_EINVAL;                                \
        }                                                     \
    } while (0)

#define AMT_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t err__ = (call);                                                      \
        if (err__ != hipSuccess) {                                                      \
            (ctx)->last_error = std::string(__func__) + ": " #call " -> " + hipGetErrorString(err__); \
            return AMT_EHIP;                                                            \
        }                                                                               \
    } while (0)

#define AMT_LAUNCH_CHECK(ctx) AMT_HIP(ctx, hipGetLastError())

static inline int amt_set_device(amt_ctx* ctx) {
    AMT_HIP(ctx, hipSetDevice(ctx->device));
    return AMT_OK;
}

// ------------------------------------------------------------------------------------------
// device math
// ------------------------------------------------------------------------------------------
namespace amt {

constexpr double kRad2Deg = 57.29577951308232;     // 180/pi, also the TAN projection constant
constexpr double kDeg2Rad = 0.017453292519943295;  // pi/180

struct vec3 {
    double x, y, z;
};

struct mat3 {
    double m[9];
};

__host__ __device__ inline mat3 make_mat3(const double* p) {
    mat3 r;
    for (int i = 0; i < 9; ++i) r.m[i] = p[i];
    return r;
}

__device__ __forceinline__ vec3 mul(const mat3& a, const vec3& v) {
    vec3 r;
    r.x = a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z;
    r.y = a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z;
    r.z = a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z;
    return r;
}

__device__ __forceinline__ double dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Constants of the ray / ellipsoid quadratic that do not depend on the ray
// (reference intersection.py:63-74: origin and direction scaled by 1/a, 1/a, 1/b).
struct ellipsoid_ray {
    double ia, ib;      // 1/a, 1/b
    double osx, osy, osz;   // -origin * (1/a,1/a,1/b)
    double oo;          // |os|^2
    double ox, oy, oz;  // origin
    int inside;         // origin inside the ellipsoid (intersection.py:239-241)
    int directed;
};

__host__ inline ellipsoid_ray make_ray(double a, double b, const double* origin, int directed) {
    ellipsoid_ray r;
    r.ia = 1 / a;
    r.ib = 1 / b;
    r.osx = -origin[0] * r.ia;
    r.osy = -origin[1] * r.ia;
    r.osz = -origin[2] * r.ib;
    r.oo = r.osx * r.osx + r.osy * r.osy + r.osz * r.osz;
    r.ox = origin[0];
    r.oy = origin[1];
    r.oz = origin[2];
    double qx = origin[0] / a, qy = origin[1] / a, qz = origin[2] / b;
    r.inside = (qx * qx + qy * qy + qz * qz) < 1;
    r.directed = directed;
    return r;
}

// Ray parameter of the chosen intersection (units of |d|), NaN for a miss.
// reference intersection.py:76-99.
__device__ __forceinline__ double ray_param(const ellipsoid_ray& e, const vec3& d) {
    const double dsx = d.x * e.ia, dsy = d.y * e.ia, dsz = d.z * e.ib;
    const double d_o = dsx * e.osx + dsy * e.osy + dsz * e.osz;
    const double d_d = dsx * dsx + dsy * dsy + dsz * dsz;
    const double disc = d_o * d_o - e.oo * d_d + d_d;
    const double root = sqrt(disc);  // NaN when the line misses
    double t;
    if (e.directed) {
        t = e.inside ? d_o + root : d_o - root;
        if (t < 0) t = NAN;
    } else {
        const double t1 = d_o - root, t2 = d_o + root;
        t = fabs(t1) < fabs(t2) ? t1 : t2;
    }
    return t / d_d;
}

__device__ __forceinline__ vec3 ray_point(const ellipsoid_ray& e, const vec3& d, double t) {
    vec3 p;
    p.x = d.x * t + e.ox;
    p.y = d.y * t + e.oy;
    p.z = d.z * t + e.oz;
    return p;
}

// Bowring (1985) single-step ECEF -> geodetic, reference transform.py:252-297.
struct bowring {
    double a, b, e2a, d;   // e2a = e^2 * a, d = (a^2 - b^2)/b
};

__host__ __device__ inline bowring make_bowring(double a, double b) {
    bowring w;
    w.a = a;
    w.b = b;
    w.e2a = ((a * a - b * b) / (a * a)) * a;
    w.d = (a * a - b * b) / b;
    return w;
}

__device__ __forceinline__ void ecef_to_geodetic(const bowring& w, double x, double y, double z, double& lat,
                                                 double& lon) {
    // the square roots and quotients from the hardware seeds + one Newton step (4e-15 relative, see fx:: below), the
    // two arctangents from libm: 1e-15 rad, well inside the reference's "at least 11 decimals (in degrees)"
    // (transform.py:205); IEEE sqrt / division cost three times as many instructions and made this the one
    // operator-level kernel below half of the HBM rate
    const double p2 = x * x + y * y;
    const double yp = __builtin_amdgcn_rsq(p2);
    const double hp = 0.5 * yp, gp = p2 * yp;
    const double ep = fma(-hp, gp, 0.5);
    const double p = fma(gp, ep, gp), ip = fma(yp, ep, yp);
    const double r2 = p2 + z * z;
    const double yr = __builtin_amdgcn_rsq(r2);
    const double ir = fma(yr, fma(-0.5 * yr, r2 * yr, 0.5), yr);
    const double tu = (w.b / w.a) * z * fma(w.d, ir, 1.0) * ip;
    const double tu2 = tu * tu;
    const double q = 1.0 + tu2;
    const double yq = __builtin_amdgcn_rsq(q);
    const double c = fma(yq, fma(-0.5 * yq, q * yq, 0.5), yq);
    const double cu3 = c * c * c;
    const double su3 = cu3 * tu2 * tu;
    // p == 0 (on the axis): rsq gives inf and the products NaN; the reference divides by zero there as well
    lat = atan2(z + w.d * su3, p - w.e2a * cu3);
    lon = atan2(y, x);
}

__device__ __forceinline__ void geodetic_to_ecef(const bowring& w, double lat, double lon, double h, double& x,
                                                 double& y, double& z) {
    // reference transform.py:156-178
    const double e2 = w.e2a / w.a;
    double sl, cl, so, co;
    sincos(lat, &sl, &cl);
    sincos(lon, &so, &co);
    // a / sqrt(1 - e^2 sin^2): hardware seed + one Newton step (4e-15 relative) instead of IEEE sqrt and division
    const double q = fma(-e2 * sl, sl, 1.0);
    const double yq = __builtin_amdgcn_rsq(q);
    const double n = w.a * fma(yq, fma(-0.5 * yq, q * yq, 0.5), yq);
    const double nh = n + h;
    x = nh * cl * co;
    y = nh * cl * so;
    z = (n * (1 - e2) + h) * sl;
}

// rotatePole (reference transform.py:301-322) of one point: geodetic (rad) at altitude h -> ECEF -> rotation m -> geodetic
// (rad).  Same arithmetic as geodetic_to_ecef / ecef_to_geodetic above, but with every operation spelled out (no
// contraction): the single-pass kernel re-evaluates the pixels next to a bin edge with it and must then agree with
// k_rotate_pole, which the two-pass plan runs over whole arrays, bit for bit — whatever code surrounds the call.
__device__ __forceinline__ void rotate_pole_rad(const bowring& w, const mat3& m, double e2, double lat, double lon, double h,
                                                double& out_lat, double& out_lon) {
#pragma clang fp contract(off)
    double sl, cl, so, co;
    sincos(lat, &sl, &cl);
    sincos(lon, &so, &co);
    const double q = __builtin_fma(-(e2 * sl), sl, 1.0);
    const double yq = __builtin_amdgcn_rsq(q);
    const double n = w.a * __builtin_fma(yq, __builtin_fma(-0.5 * yq, q * yq, 0.5), yq);
    const double nh = n + h;
    const double gx = (nh * cl) * co, gy = (nh * cl) * so, gz = __builtin_fma(n, 1.0 - e2, h) * sl;
    const double x = __builtin_fma(m.m[2], gz, __builtin_fma(m.m[1], gy, m.m[0] * gx));
    const double y = __builtin_fma(m.m[5], gz, __builtin_fma(m.m[4], gy, m.m[3] * gx));
    const double z = __builtin_fma(m.m[8], gz, __builtin_fma(m.m[7], gy, m.m[6] * gx));
    const double p2 = __builtin_fma(x, x, y * y);
    const double yp = __builtin_amdgcn_rsq(p2);
    const double hp = 0.5 * yp, gp = p2 * yp;
    const double ep = __builtin_fma(-hp, gp, 0.5);
    const double p = __builtin_fma(gp, ep, gp), ip = __builtin_fma(yp, ep, yp);
    const double r2 = __builtin_fma(z, z, p2);
    const double yr = __builtin_amdgcn_rsq(r2);
    const double ir = __builtin_fma(yr, __builtin_fma(-0.5 * yr, r2 * yr, 0.5), yr);
    const double tu = (((w.b / w.a) * z) * __builtin_fma(w.d, ir, 1.0)) * ip;
    const double tu2 = tu * tu;
    const double qq = 1.0 + tu2;
    const double yc = __builtin_amdgcn_rsq(qq);
    const double c = __builtin_fma(yc, __builtin_fma(-0.5 * yc, qq * yc, 0.5), yc);
    const double cu3 = (c * c) * c;
    const double su3 = (cu3 * tu2) * tu;
    out_lat = atan2(__builtin_fma(w.d, su3, z), __builtin_fma(-w.e2a, cu3, p));
    out_lon = atan2(y, x);
}

// ... in degrees (what the resampling code works in; torch.deg2rad / rad2deg are these two products)
__device__ __forceinline__ void rotate_pole_deg(const bowring& w, const mat3& m, double e2, double lat_deg, double lon_deg, double h,
                                                double& out_lat_deg, double& out_lon_deg) {
#pragma clang fp contract(off)
    double la, lo;
    rotate_pole_rad(w, m, e2, lat_deg * kDeg2Rad, lon_deg * kDeg2Rad, h, la, lo);
    out_lat_deg = la * kRad2Deg;
    out_lon_deg = lo * kRad2Deg;
}

// Constants of the pole plan of the fused binning (amt_georef_out.bin_pole): the rotation by +90 deg about x that the
// reference applies to frames with a pole in view (resample.py:176-201), as rotation_matrix() builds it (cos(pi/2) is
// 6.1e-17, not 0), the reference ellipsoid and the mapping altitude.
struct pole_consts {
    bowring w;
    mat3 rot;
    double e2, one_minus_e2, alt, pad_;
};

__host__ inline pole_consts make_pole_consts(double a0, double b0, double altitude) {
    pole_consts c;
    c.w = make_bowring(a0, b0);
    const double cs = std::cos(M_PI / 2), sn = std::sin(M_PI / 2);
    const double r[9] = {1, 0, 0, 0, cs, -sn, 0, sn, cs};
    c.rot = make_mat3(r);
    c.e2 = c.w.e2a / c.w.a;
    c.one_minus_e2 = 1.0 - c.e2;
    c.alt = altitude;
    c.pad_ = 0;
    return c;
}

// SM cartesian -> (mlat deg, mlt h), reference transform.py:104-127,373-386,421-427
__device__ __forceinline__ void sm_to_mlat_mlt(const vec3& s, double& mlat, double& mlt) {
    const double sxy = sqrt(s.x * s.x + s.y * s.y);
    mlat = atan2(s.z, sxy) * kRad2Deg;
    mlt = (atan2(s.y, s.x) * kRad2Deg) * (24.0 / 360.0) + 12.0;
}

// ------------------------------------------------------------------------------------------
// Reduced-cost f64 math for the fused frame kernel.
//
// The frame kernel is FP64-VALU bound (93 % VALU-busy in profiles/r1), and the contract is
// 1e-6 deg, not 0.5 ulp.  IEEE division / sqrt expand to 10-15 VALU instructions each and the
// libm inverse trigonometry carries special-case handling the frame kernel never needs, so
// it uses hardware seeds (v_rcp_f64 / v_rsq_f64: 4.6e-8 / 5.2e-8 relative on gfx950, measured with
// tools/probe_f64_approx.hip) plus ONE Newton step (2.1e-15 / 4.1e-15) and a 9-term odd minimax
// polynomial for atan (1.1e-13 relative).  Inputs must be finite (callers branch on hit/miss first).
// ------------------------------------------------------------------------------------------
namespace fm {

__device__ __forceinline__ double rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}

// (sqrt(x), 1/sqrt(x)) from one v_rsq_f64 + one coupled Newton step; x > 0
__device__ __forceinline__ void sqrt_rsqrt(double x, double& s, double& rs) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    s = fma(g, r, g);
    rs = 2.0 * fma(h, r, h);
}

__device__ __forceinline__ double rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * y;
    const double r = fma(-h, x * y, 0.5);
    return 2.0 * fma(h, r, h);
}

// NaN for x < 0 (a missed ray), as sqrt() would give
__device__ __forceinline__ double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    return fma(g, fma(-h, g, 0.5), g);
}

// atan(t) in DEGREES for |t| <= tan(pi/8): t * P(t^2), least-squares minimax fit (rel. error 1.1e-13)
__device__ __forceinline__ double atan_core_deg(double t) {
    const double s = t * t;
    double p = 1.8906869702149334;
    p = fma(p, s, -3.454838730079989);
    p = fma(p, s, 4.3575716232982611);
    p = fma(p, s, -5.2048471660339564);
    p = fma(p, s, 6.3660331903476619);
    p = fma(p, s, -8.1851078904860763);
    p = fma(p, s, 11.45915587463822);
    p = fma(p, s, -19.098593170990291);
    p = fma(p, s, 57.295779513082323);
    return p * t;
}

constexpr double kTanPi8 = 0.41421356237309503;

// atan2(n, d) in degrees for d > 0 (result in (-90, 90))
__device__ __forceinline__ double atan_pos_deg(double n, double d) {
    const double an = fabs(n);
    const bool flip = an > d;
    const double mx = flip ? an : d, mn = flip ? d : an;
    const bool big = mn > kTanPi8 * mx;
    const double num = big ? mn - mx : mn;
    const double den = big ? mn + mx : mx;
    double r = atan_core_deg(num * rcp(den)) + (big ? 45.0 : 0.0);
    r = flip ? 90.0 - r : r;
    return copysign(r, n);
}

// atan2(y, x) in degrees, all quadrants; (0, 0) gives NaN instead of 0
__device__ __forceinline__ double atan2_deg(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const bool flip = ay > ax;
    const double mx = flip ? ay : ax, mn = flip ? ax : ay;
    const bool big = mn > kTanPi8 * mx;
    const double num = big ? mn - mx : mn;
    const double den = big ? mn + mx : mx;
    double r = atan_core_deg(num * rcp(den)) + (big ? 45.0 : 0.0);
    r = flip ? 90.0 - r : r;
    r = x < 0 ? 180.0 - r : r;
    return copysign(r, y);
}

// asin(c) in degrees, |c| <= 1
__device__ __forceinline__ double asin_deg(double c) {
    const double q = (1.0 - c) * (1.0 + c);
    const double d = q > 0 ? sqrt_pos(q) : 0.0;
    return atan_pos_deg(c, d);
}

}  // namespace fm

// Bowring single step as in ecef_to_geodetic, degrees out, with the reduced-cost primitives:
// 3 rsqrt + 2 rcp instead of 3 sqrt + 4 div + atan + atan2.
struct bowring_fast {
    double b_over_a, d, e2a;
};

__host__ inline bowring_fast make_bowring_fast(double a, double b) {
    bowring_fast w;
    w.b_over_a = b / a;
    w.d = (a * a - b * b) / b;
    w.e2a = ((a * a - b * b) / (a * a)) * a;
    return w;
}

// Numerator and denominator of the Bowring latitude, lat = atan(n / d) with d > 0, and 1 / |(x, y, z)| as a
// by-product (the caller's elevation needs it as well).
__device__ __forceinline__ void bowring_fast_nd(const bowring_fast& w, double x, double y, double z, double& n, double& d,
                                                double& inv_r) {
    const double p2 = x * x + y * y;
    double p, ip;
    fm::sqrt_rsqrt(p2, p, ip);
    inv_r = fm::rsqrt(p2 + z * z);
    const double tu = w.b_over_a * z * fma(w.d, inv_r, 1.0) * ip;
    const double tu2 = tu * tu;
    const double c = fm::rsqrt(1.0 + tu2);
    const double cu3 = c * c * c;
    const double su3 = cu3 * tu2 * tu;
    n = fma(w.d, su3, z);
    d = fma(-w.e2a, cu3, p);
}

// inv_r (optional out): 1 / |(x, y, z)|
__device__ __forceinline__ void ecef_to_geodetic_deg_fast(const bowring_fast& w, double x, double y, double z,
                                                          double& lat_deg, double& lon_deg, double* inv_r = nullptr) {
    double n, d, ir;
    bowring_fast_nd(w, x, y, z, n, d, ir);
    if (inv_r) *inv_r = ir;
    lat_deg = fm::atan_pos_deg(n, d);
    lon_deg = fm::atan2_deg(y, x);
}

// Angle in DEGREES from the plane vector (xr, yr) to (x, y), i.e. atan2(xr y - yr x, xr x + yr y), for the small
// angles between neighbouring pixels: t (1 - t^2/3 + t^4/5 - t^6/7) with t = cross / dot.  `ok` is false (and the
// value meaningless) when |t| > 0.03 (1.7 deg; the next term is then < 7e-14 relative), when dot <= 0 or for NaN:
// the caller then evaluates the full arctangent.  One reciprocal + 11 multiply-adds instead of ~30 instructions.
__device__ __forceinline__ double small_angle_deg(double xr, double yr, double x, double y, bool& ok) {
    const double cross = xr * y - yr * x, dot = xr * x + yr * y;
    const double t = cross * fm::rcp(dot);
    const double s = t * t;
    ok = s <= 9.0e-4 && dot > 0.0;
    double q = -8.1851113590117602;              // -(180/pi)/7
    q = fma(q, s, 11.459155902616464);           //  (180/pi)/5
    q = fma(q, s, -19.098593171027442);          // -(180/pi)/3
    q = fma(q, s, 57.295779513082323);           //   180/pi
    return q * t;
}

// ------------------------------------------------------------------------------------------
// fx:: the arithmetic of the row-marching frame kernel, written operation by operation.
//
// k_georef_rows evaluates a row either on a speculative straight-line path (every lane hits the shell, every
// small angle applies, ...) or on the general, divergent path.  Both must give the same BITS (the two execution
// plans of the pipeline are tested for bit-identical grids, and a row may take either path depending on the
// kernel variant), so the arithmetic lives here once: explicit fma / mul / add in a fixed order with contraction
// off — the compiler can schedule these but not re-associate or fuse them differently in different contexts.
// Issue costs on gfx950 (tools/valu_rates.hip): f64 fma/mul/add 3.0 cycles per wave and SIMD, v_rcp/rsq_f64 11,
// 32-bit moves and integer ops 2, v_cndmask 3.1, DPP move 3.2, LDS f64 atomic 15.
// ------------------------------------------------------------------------------------------
namespace fx {
__device__ __forceinline__ double rcp_n(double x) {              // 1 / x, 2e-15 relative
#pragma clang fp contract(off)
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

__device__ __forceinline__ double rsqrt_n(double x) {            // 1 / sqrt(x), x > 0, 4e-15 relative
#pragma clang fp contract(off)
    const double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * y, g = x * y;
    const double e = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(y, e, y);
}

__device__ __forceinline__ double sqrt_n(double x) {             // sqrt(x); NaN for x < 0 (a missed ray)
#pragma clang fp contract(off)
    const double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * y, g = x * y;
    const double e = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, e, g);
}

__device__ __forceinline__ void sqrt_rsqrt_n(double x, double& s, double& rs) {
#pragma clang fp contract(off)
    const double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * y, g = x * y;
    const double e = __builtin_fma(-h, g, 0.5);
    s = __builtin_fma(g, e, g);
    rs = __builtin_fma(y, e, y);
}

__device__ __forceinline__ double dot3(double ax, double ay, double az, double bx, double by, double bz) {
#pragma clang fp contract(off)
    return __builtin_fma(az, bz, __builtin_fma(ay, by, ax * bx));
}

// Bowring numerator / denominator (lat = atan(n / d), d > 0) and 1 / |(x, y, z)|: 27 operations + 3 v_rsq_f64
__device__ __forceinline__ void bowring_nd(const bowring_fast& w, double x, double y, double z, double& n, double& d,
                                           double& inv_r) {
#pragma clang fp contract(off)
    const double p2 = __builtin_fma(x, x, y * y);
    double p, ip;
    sqrt_rsqrt_n(p2, p, ip);
    inv_r = rsqrt_n(__builtin_fma(z, z, p2));
    const double k = __builtin_fma(w.d, inv_r, 1.0);
    const double tu = ((w.b_over_a * z) * k) * ip;
    const double tu2 = tu * tu;
    const double c = rsqrt_n(__builtin_fma(tu, tu, 1.0));
    const double cu3 = (c * c) * c;
    const double su3 = (cu3 * tu2) * tu;
    n = __builtin_fma(w.d, su3, z);
    d = __builtin_fma(-w.e2a, cu3, p);
}

// Polynomial coefficients of the row kernel.  They travel in the kernel-argument segment and are read through the
// scalar cache where they are used: as literals the compiler keeps all of them in VGPRs across the row loop (64-bit
// literals cannot be encoded in VOP3), which cost ~40 of the 128 registers.
struct math_table {
    double small4[4];       // (180/pi) {-1/7, 1/5, -1/3, 1}: atan(t) / t in degrees for t^2 <= 9e-4
    double atan9[9];        // atan(t) / t in degrees for |t| <= tan(pi/8), highest power first (fm::atan_core_deg)
    double pad_[3];
};

__host__ inline math_table make_math_table() {
    math_table t = {{-8.1851113590117602, 11.459155902616464, -19.098593171027442, 57.295779513082323},
                    {1.8906869702149334, -3.454838730079989, 4.3575716232982611, -5.2048471660339564, 6.3660331903476619,
                     -8.1851078904860763, 11.45915587463822, -19.098593170990291, 57.295779513082323},
                    {0, 0, 0}};
    return t;
}

// t (1 - s/3 + s^2/5 - s^3/7) in degrees, s = t^2 <= 9e-4 (next term < 7e-14 relative)
__device__ __forceinline__ double small_atan_deg(double t, double s, const double (&c)[4]) {
#pragma clang fp contract(off)
    double q = __builtin_fma(c[0], s, c[1]);
    q = __builtin_fma(q, s, c[2]);
    q = __builtin_fma(q, s, c[3]);
    return q * t;
}

// The two small angles (degrees) a corner or centre needs, with ONE reciprocal: from the plane vector (ar, br) to
// (a, b) — Bowring (d, n) pairs: the step in latitude — and from (xr, yr) to (x, y) — GEO (x, y): the step in
// longitude.  `ok` is false (values meaningless) when either angle is above 1.7 deg, a dot product is not positive,
// or anything is NaN: the caller then evaluates the full arctangents.
__device__ __forceinline__ void small_angles(double ar, double br, double a, double b, double xr, double yr, double x,
                                             double y, const double (&c)[4], double& dlat, double& dlon, bool& ok) {
#pragma clang fp contract(off)
    const double cross_a = __builtin_fma(ar, b, -(br * a)), dot_a = __builtin_fma(ar, a, br * b);
    const double cross_o = __builtin_fma(xr, y, -(yr * x)), dot_o = __builtin_fma(xr, x, yr * y);
    const double r = rcp_n(dot_a * dot_o);
    const double ta = (cross_a * dot_o) * r, to = (cross_o * dot_a) * r;
    const double sa = ta * ta, so = to * to;
    ok = sa <= 9.0e-4 && so <= 9.0e-4 && dot_a > 0.0 && dot_o > 0.0;
    dlat = small_atan_deg(ta, sa, c);
    dlon = small_atan_deg(to, so, c);
}

// atan(t) in DEGREES for |t| <= tan(pi/8): t * P(t^2), 1.1e-13 relative
__device__ __forceinline__ double atan_core_deg(double t, const double (&c)[9]) {
#pragma clang fp contract(off)
    const double s = t * t;
    double p = __builtin_fma(c[0], s, c[1]);
    p = __builtin_fma(p, s, c[2]);
    p = __builtin_fma(p, s, c[3]);
    p = __builtin_fma(p, s, c[4]);
    p = __builtin_fma(p, s, c[5]);
    p = __builtin_fma(p, s, c[6]);
    p = __builtin_fma(p, s, c[7]);
    p = __builtin_fma(p, s, c[8]);
    return p * t;
}

// atan2(n, d) in degrees for d > 0 (result in (-90, 90))
__device__ __forceinline__ double atan_pos_deg(double n, double d, const double (&c)[9]) {
#pragma clang fp contract(off)
    const double an = fabs(n);
    const bool flip = an > d;
    const double mx = flip ? an : d, mn = flip ? d : an;
    const bool big = mn > fm::kTanPi8 * mx;
    const double num = big ? mn - mx : mn;
    const double den = big ? mn + mx : mx;
    double r = atan_core_deg(num * rcp_n(den), c) + (big ? 45.0 : 0.0);
    r = flip ? 90.0 - r : r;
    return copysign(r, n);
}

// atan2(y, x) in degrees, all quadrants; (0, 0) gives NaN instead of 0
__device__ __forceinline__ double atan2_deg(double y, double x, const double (&c)[9]) {
#pragma clang fp contract(off)
    const double ax = fabs(x), ay = fabs(y);
    const bool flip = ay > ax;
    const double mx = flip ? ay : ax, mn = flip ? ax : ay;
    const bool big = mn > fm::kTanPi8 * mx;
    const double num = big ? mn - mx : mn;
    const double den = big ? mn + mx : mx;
    double r = atan_core_deg(num * rcp_n(den), c) + (big ? 45.0 : 0.0);
    r = flip ? 90.0 - r : r;
    r = x < 0 ? 180.0 - r : r;
    return copysign(r, y);
}

constexpr double kSin45 = 0.7071;       // below sin(45 deg): c / (1 + sqrt(1 - c^2)) stays within tan(pi/8)

// asin(c) in degrees for |c| <= kSin45 as 2 atan(c / (1 + sqrt(1 - c^2))): no range reduction, no selects
// (21 operations + v_rsq + v_rcp; the general form below costs three times that)
__device__ __forceinline__ double asin_deg_low(double c, const double (&k)[9]) {
#pragma clang fp contract(off)
    const double sq = sqrt_n(__builtin_fma(-c, c, 1.0));
    const double t = c * rcp_n(1.0 + sq);
    return 2.0 * atan_core_deg(t, k);
}

// asin(c) in degrees, |c| <= 1, any elevation
__device__ __forceinline__ double asin_deg_any(double c, const double (&k)[9]) {
#pragma clang fp contract(off)
    const double q = (1.0 - c) * (1.0 + c);
    const double d = q > 0 ? sqrt_n(q) : 0.0;
    return atan_pos_deg(c, d, k);
}

__device__ __forceinline__ double asin_deg(double c, const double (&k)[9]) {
    // (a NaN takes the short form and stays NaN)
    return fabs(c) > kSin45 ? asin_deg_any(c, k) : asin_deg_low(c, k);
}

}  // namespace fx

// True when the closed longitude path o00 -> o01 -> o11 -> o10 -> o00 (corner quad of one pixel, degrees)
// winds once around a geographic pole: the wrapped longitude steps then sum to +-360 instead of 0.
__device__ __forceinline__ bool quad_winds_pole(double o00, double o01, double o11, double o10) {
    auto step = [](double a, double b) {
        const double d = b - a;
        return d - 360.0 * rint(d * (1.0 / 360.0));
    };
    const double w = step(o00, o01) + step(o01, o11) + step(o11, o10) + step(o10, o00);
    return fabs(w) > 180.0;
}

// TAN (gnomonic) WCS camera model.
struct tan_wcs {
    double cd[4];
    double crpix[2];
    mat3 rot;
};

__host__ inline tan_wcs make_tan_wcs(const amt_frame_params* p) {
    tan_wcs w;
    for (int i = 0; i < 4; ++i) w.cd[i] = p->cd[i];
    w.crpix[0] = p->crpix[0];
    w.crpix[1] = p->crpix[1];
    w.rot = make_mat3(p->rot);
    return w;
}

// Unit direction of pixel coordinate (x, y) [0-based, pixel centres at integers].
// reference wcs.py:93-142 evaluates atan2/atan and then cos/sin of those angles; algebraically the
// native unit vector is (-Y, X, 180/pi) / sqrt(X^2 + Y^2 + (180/pi)^2) with (X, Y) = CD (p - CRPIX + 1),
// which needs one rsqrt and no trigonometry.
__device__ __forceinline__ vec3 tan_direction(const tan_wcs& w, double x, double y) {
    const double px = x - w.crpix[0] + 1.0;
    const double py = y - w.crpix[1] + 1.0;
    const double X = w.cd[0] * px + w.cd[1] * py;
    const double Y = w.cd[2] * px + w.cd[3] * py;
    const double inv = rsqrt(X * X + Y * Y + kRad2Deg * kRad2Deg);
    vec3 v = {-Y * inv, X * inv, kRad2Deg * inv};
    return mul(w.rot, v);
}

// Same with the reduced-cost reciprocal square root (|d| = 1 to 4e-15), for the fused frame kernel.
__device__ __forceinline__ vec3 tan_direction_fast(const tan_wcs& w, double x, double y) {
    const double px = x - w.crpix[0] + 1.0;
    const double py = y - w.crpix[1] + 1.0;
    const double X = w.cd[0] * px + w.cd[1] * py;
    const double Y = w.cd[2] * px + w.cd[3] * py;
    const double inv = fm::rsqrt(X * X + Y * Y + kRad2Deg * kRad2Deg);
    vec3 v = {-Y * inv, X * inv, kRad2Deg * inv};
    return mul(w.rot, v);
}

// astropy Angle.wrap_at(180 deg) of (v + 180), reference resample.py:212-218
__device__ __forceinline__ double wrap180_shifted(double v) {
    double a = v + 180.0;
    const double wraps = floor((a + 180.0) / 360.0);
    a = a - wraps * 360.0;
    if (a >= 180.0) a -= 360.0;
    if (a < -180.0) a += 360.0;
    return a;
}

// ------------------------------------------------------------------------------------------
// Histogram axis with the reference's bin-edge semantics (auromat/util/histogram.py:178-224)
// ------------------------------------------------------------------------------------------
struct axis_dev {
    const double* edges;
    int nbin;
    int uniform;
    double scale, last_rounded;
    double e0, e_last, step, inv_step;
};

// Edge i of a uniform axis, bit-identical to np.linspace(e0, e_last, nbin+1)[i]:
// arange(num)*step + start with two roundings (contraction to an FMA would change the last bit),
// and the end point stored exactly (numpy/_core/function_base.py: y = y*step + start; y[-1] = stop).
__device__ __forceinline__ double linspace_edge(const axis_dev& ax, int i) {
#pragma clang fp contract(off)
    const double m = (double)i * ax.step;
    const double e = m + ax.e0;
    return i >= ax.nbin ? ax.e_last : e;
}

// searchsorted(edges, v, 'right') with the right-most-edge rule of histogram.py:209-224.
// Returns 0..nbin+1; 0 and nbin+1 are outliers (NaN sorts to the end like NumPy does).
// UNIFORM_ONLY: the caller guarantees ax.uniform (drops the bisection over an edge table from the code).
template <bool UNIFORM_ONLY = false>
__device__ __forceinline__ int bin_index(const axis_dev& ax, double v) {
    if (!(v == v)) return ax.nbin + 1;
    if (v < ax.e0) return 0;
    if (v >= ax.e_last) {
        const bool on_edge = rint(v * ax.scale) / ax.scale == ax.last_rounded;
        return on_edge ? ax.nbin : ax.nbin + 1;
    }
    int g;
    if (UNIFORM_ONLY || ax.uniform) {
        g = (int)((v - ax.e0) * ax.inv_step);
        g = g < 0 ? 0 : (g > ax.nbin - 1 ? ax.nbin - 1 : g);
        // the guess is almost always right and otherwise off by one: two edge evaluations on the common path,
        // the loops (bounded by e0 <= v < e_last) only when it was wrong
        if (v < linspace_edge(ax, g)) {
            do { --g; } while (v < linspace_edge(ax, g));
        } else if (v >= linspace_edge(ax, g + 1)) {
            do { ++g; } while (v >= linspace_edge(ax, g + 1));
        }
    } else {
        int lo = 0, hi = ax.nbin;             // invariant: edges[lo] <= v < edges[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (v >= ax.edges[mid]) lo = mid; else hi = mid;
        }
        g = lo;
    }
    return g + 1;
}

// The five scalars of a uniform axis that the common path of the fused binning needs.
struct axis_lin {
    double e0, e_last, step, inv_step;
    double margin;      // fractional positions below this (just above an edge) take the exact path, see bin_fast
    int nbin, pad;
};

// A pixel that sits on a bin edge in the sense of the reference's right-most-edge rule (histogram.py:215-224:
// v >= edge and around(v, decimal) == around(edge, decimal)).  For an interior edge that changes nothing, but the
// fused kernel bins into a superset of the final grid: if the edge turns out to be the LAST edge of the final
// grid, the reference counts the pixel into the last bin.  Such pixels (a few dozen per frame) are therefore not
// binned by the kernel but recorded, and resolved when the final grid is known (k_apply_bin_events).
struct bin_event {
    int bx, by;                 // 1-based superset bins by plain searchsorted(edges, v, 'right')
    unsigned int flags;         // 1: on the lower x edge of bx, 2: on the lower y edge of by
    unsigned int c0, c1, c2;    // image channels
    long long el;               // elevation, 31.32 fixed point
};
static_assert(sizeof(bin_event) == 32, "bin_event layout");

inline axis_lin make_axis_lin(const axis_dev& a) {
    axis_lin l;
    l.e0 = a.e0;
    l.e_last = a.e_last;
    l.step = a.step;
    l.inv_step = a.inv_step;
    l.nbin = a.nbin;
    l.pad = 0;
    // width of the on-edge zone of the rule above: one unit of the rounded decimal, in bins (<= 1e-5), with slack
    const double zone = a.scale > 0 && a.step > 0 ? 1.5 / (a.scale * a.step) : 0.0;
    l.margin = zone > 1e-7 ? zone : 1e-7;
    return l;
}

// Common path of bin_index for a uniform axis.  t = (v - e0) / step locates v to ~1e-12 bins (two roundings of
// the expression, and the edges np.linspace produces differ from e0 + i step by rounding only), so whenever the
// fractional part of t keeps away from 0 (by ax.margin: the on-edge zone of the right-most-edge rule, <= 1e-5) and
// from 1 (by 1e-7) the bin floor(t) is certain and no edge needs evaluating.  Everything else — close to an
// edge, outside the axis, NaN — sets `slow` and is decided exactly by bin_index (a few dozen pixels per frame).
// Returns the 1-based bin, 0 when `slow` is set.
__device__ __forceinline__ int bin_fast(double e0, double inv_step, double margin, int nbin, double v, bool& slow) {
    const double t = (v - e0) * inv_step;
    const double fl = floor(t);
    const double fr = t - fl;
    const bool sure = fr > margin && fr < 1.0 - 1e-7 && fl >= 0.0 && fl < (double)nbin;     // false for NaN
    slow = !sure;
    return sure ? (int)fl + 1 : 0;
}
__device__ __forceinline__ int bin_fast(const axis_lin& ax, double v, bool& slow) {
    return bin_fast(ax.e0, ax.inv_step, ax.margin, ax.nbin, v, slow);
}

// v lies in bin b (1-based, by plain searchsorted): does it sit on that bin's lower edge in the sense of the
// right-most-edge rule?
__device__ __forceinline__ bool on_lower_edge(const axis_dev& ax, int b, double v) {
    const double e = linspace_edge(ax, b - 1);
    return rint(v * ax.scale) / ax.scale == rint(e * ax.scale) / ax.scale;
}

constexpr double kFix = 4294967296.0;   // 2^32: elevation sums are kept in signed 31.32 fixed point

// round-to-nearest-even of v * 2^32 as a 64-bit integer for |v * 2^32| < 2^51 (elevations: < 2^39), the value
// __double2ll_rn(v * kFix) gives: the product is exact, and adding 1.5 * 2^52 rounds it to an integer in the low mantissa
// bits, of which the constant's own integer is subtracted — one fused multiply-add and one 32-bit subtraction (the
// constant's low word is zero) instead of the seven instructions of the library conversion.
__device__ __forceinline__ long long to_fix32(double v) {
    constexpr double kMagic = 6755399441055744.0;       // 2^52 + 2^51
    const double t = __builtin_fma(v, kFix, kMagic);
    return __double_as_longlong(t) - __double_as_longlong(kMagic);
}

inline void make_axis(const amt_axis* a, axis_dev* out) {
    out->edges = a->edges;
    out->nbin = a->nbin;
    out->uniform = a->uniform;
    out->scale = a->scale;
    out->last_rounded = a->last_rounded;
    out->e0 = a->first;
    out->e_last = a->last;
    out->step = a->uniform ? a->step : 0.0;
    out->inv_step = a->nbin / (a->last - a->first);
}

inline bool axis_ok(const amt_axis* a) {
    if (a == nullptr || a->nbin <= 0 || !(a->last > a->first)) return false;
    if (a->uniform) return a->step > 0;
    return a->edges != nullptr;
}


}  // namespace amt
