// Fused per-frame georeferencing kernel (reference auromat/mapping/astrometry.py:49-212).
//
// One workgroup owns a TW x TH tile of pixels:
//   phase 1  every corner of the tile ((TW+1) x (TH+1), one thread per corner, strided):
//            WCS pixel -> unit direction -> ray/ellipsoid hit -> [LDS: P, d]
//            -> J2000->GEO rotation -> Bowring -> lat/lon (deg) written by the tile that owns the corner
//   phase 2  every pixel of the tile: centre point = mean of the 4 corner hits read back from LDS
//            (fast mode) or its own ray cast (exact mode) -> lat/lon, elevation, optional MLat/MLT;
//            leaves a "valid above the elevation threshold" flag per pixel in LDS
//   phase 3  (bounding box requested) every corner again, from registers: a corner counts when one of
//            its in-tile neighbour centres is valid -> min/max of lat/lon -> one partial per workgroup
// Each ray is cast once per tile (1.08x redundancy at 64x16), nothing but the final arrays touches
// HBM, and all per-frame constants travel as kernel arguments (SGPRs), not through LDS.
//
// The kernel is FP64-VALU bound; see the fm:: helpers in amt_common.h for the arithmetic budget.
// Algorithmic HBM bytes per frame: 16 B per corner + 24 B per pixel written, nothing read
// ("WCS-fused" row of SURVEY.md §8d); with caller-supplied directions +24 B per corner read.
#include "amt_common.h"

namespace {

using namespace amt;

// The ray / shell intersection with everything expressed in the GEO frame (k_georef_rows).  The shell is aligned
// with the J2000 axes (reference intersection.py:63-74 scales J2000 components by 1/a, 1/a, 1/b), so in GEO
// coordinates it is the quadric x^T Q x = 1 with Q = M diag(1/a^2, 1/a^2, 1/b^2) M^T = q_a I + q_d k k^T, where
// q_a = 1/a^2, q_d = 1/b^2 - 1/a^2 and k = M e_z is the J2000 pole in GEO coordinates.  Intersecting there gives
// the hit in GEO coordinates: the 3x3 rotation of every corner point and of every centre point (18 multiply-adds
// per pixel) disappears, and the centre of the fast mode, a mean of corner hits, needs none either because
// rotations are linear.  With the rank-one form the quadratic's coefficients of a ray o + t u are
//   a2 = q_a (u.u) + q_d (k.u)^2,   -b/2 = -(q_a (u.o) + q_d (k.u)(k.o)),   c = o^T Q o - 1
// (11 constants, 14 multiply-adds; u need not be normalised).
struct shell_ray {
    double qa, qd;                          // 1/a^2, 1/b^2 - 1/a^2
    double kx, ky, kz;                      // J2000 z axis in GEO
    double qd_ko;                           // q_d (k.o)
    double c0;                              // o^T Q o - 1
    double ox, oy, oz;                      // camera in GEO
    double root_sign;                       // +1: camera inside the shell (far root), -1: outside (intersection.py:84-91,239-241)
    double pad_;
};

__host__ inline void mat_mul3(const double* a, const double* b, double* out) {     // out = a b (row major 3x3)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

__host__ inline shell_ray make_shell_ray(double a, double b, const double* cam, const double* m) {
    shell_ray r;
    r.qa = 1 / (a * a);
    r.qd = 1 / (b * b) - 1 / (a * a);
    r.kx = m[2], r.ky = m[5], r.kz = m[8];
    r.ox = m[0] * cam[0] + m[1] * cam[1] + m[2] * cam[2];
    r.oy = m[3] * cam[0] + m[4] * cam[1] + m[5] * cam[2];
    r.oz = m[6] * cam[0] + m[7] * cam[1] + m[8] * cam[2];
    r.qd_ko = r.qd * cam[2];                // k.o = (M e_z).(M cam) = cam_z
    const double qx = cam[0] / a, qy = cam[1] / a, qz = cam[2] / b;
    const double oo = qx * qx + qy * qy + qz * qz;       // axis-aligned form: exact decision, and a better c0
    r.c0 = oo - 1;
    r.root_sign = oo < 1 ? 1.0 : -1.0;
    r.pad_ = 0;
    return r;
}

// Affine form of the TAN camera model in the GEO frame.  reference wcs.py:93-142 evaluates atan2 / atan and then
// cos / sin of those angles; algebraically the un-normalised native vector is (-Y, X, 180/pi) with (X, Y) = CD (p -
// CRPIX + 1), i.e. after the rotation w = u0 + px ux + py uy: the direction of a pixel is three multiply-adds and
// one normalisation.  px = column + cx, py = row + cy for CORNER (column - 1/2, row - 1/2); centres add 1/2.
struct affine_cam {
    double u0[3], ux[3], uy[3];
    double cx, cy;
    double pad_;
};

__host__ inline affine_cam make_affine_cam(const tan_wcs& w) {
    affine_cam c;
    const double* r = w.rot.m;
    for (int i = 0; i < 3; ++i) {
        const double r0 = r[3 * i], r1 = r[3 * i + 1], r2 = r[3 * i + 2];
        c.u0[i] = kRad2Deg * r2;
        c.ux[i] = w.cd[0] * r1 - w.cd[2] * r0;
        c.uy[i] = w.cd[1] * r1 - w.cd[3] * r0;
    }
    c.cx = 0.5 - w.crpix[0];
    c.cy = 0.5 - w.crpix[1];
    c.pad_ = 0;
    return c;
}

struct georef_args {
    tan_wcs wcs;
    ellipsoid_ray ray;
    mat3 m_geo;
    mat3 m_sm;
    bowring_fast bw;
    // k_georef_rows works in the GEO frame: camera model rotated into GEO, shell as a general quadric, GEO -> SM
    affine_cam cam;
    shell_ray sray;
    mat3 m_geo_sm;
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
    // fused binning (row-marching kernel only)
    axis_dev bax, bay;       // full descriptions (rare paths: right-most edge rule, wrong first guess)
    axis_lin bxl, byl;       // what the common path needs: 5 scalars per axis
    const void* bin_img;
    unsigned long long* bin_acc;
    int bin_lon_wrap, bin_magnetic;
    int item_order, chunk_stride;       // amt_georef_out.item_order; stride of the interleaved chunk order
    // item_order 4 (two fronts): rows of work items [front_split, n) — the Earth side — are visited from front_split
    // on, rows [0, front_split) — the sky side — from front_split - 1 backwards, front_e of the one for every front_s
    // of the other; front_flip mirrors the frame first (Earth above the limb)
    int front_split, front_e, front_s, front_flip;
    // rows of work items that cannot see the shell (sky_bands(): rows [0, sky_top_end) and [sky_bottom_begin, n)); their
    // waves write NaN and cast no ray
    int sky_top_end, sky_bottom_begin;
    int bin_pole;                       // amt_georef_out.bin_pole: bin (and box) in the coordinates rotated by 90 deg about x
    int row_layout;                     // amt_georef_out.row_layout: 0 = contiguous rows, 1 = strip-padded rows (k_georef_rows only)
    pole_consts pole;
    bin_event* bin_events;      // optional list for on-edge pixels (amt_georef_out.bin_events)
    unsigned int* bin_event_count;
    long long bin_event_cap;
};

// Frames of one launch of k_georef_rows (the kernel-argument segment holds their constants side by side)
constexpr int kMaxBatch = 3;
struct georef_batch {
    georef_args f[kMaxBatch];
    fx::math_table math;
};
static_assert(sizeof(georef_args) % 8 == 0 && sizeof(georef_batch) <= 4096, "kernel-argument segment");

constexpr int kThreads = 256;
// Workgroup of the row-marching kernel.  Its waves never synchronise with each other (each has private LDS), so
// the size only sets the granularity at which the dispatcher places and retires work.
#ifndef AMT_ROWS_THREADS
#define AMT_ROWS_THREADS 256
#endif
constexpr int kRowsThreads = AMT_ROWS_THREADS;
constexpr double kInf = __builtin_huge_val();

// Ray parameter with the reduced-cost primitives (same algebra as amt::ray_param, directed, camera
// outside or inside decided on the host).  NaN for a miss.
__device__ __forceinline__ double ray_param_fast(const ellipsoid_ray& e, const vec3& d) {
    const double dsx = d.x * e.ia, dsy = d.y * e.ia, dsz = d.z * e.ib;
    const double d_o = dsx * e.osx + dsy * e.osy + dsz * e.osz;
    const double d_d = dsx * dsx + dsy * dsy + dsz * dsz;
    const double disc = d_o * d_o - e.oo * d_d + d_d;
    const double root = fm::sqrt_pos(disc);          // NaN when the line misses
    double t = e.inside ? d_o + root : d_o - root;
    if (t < 0) t = NAN;
    return t * fm::rcp(d_d);
}

__device__ __forceinline__ void sm_to_mlat_mlt_fast(const vec3& s, double& mlat, double& mlt) {
    const double q = s.x * s.x + s.y * s.y;
    mlat = fm::atan_pos_deg(s.z, q > 0 ? fm::sqrt_pos(q) : 0.0);
    mlt = fm::atan2_deg(s.y, s.x) * (24.0 / 360.0) + 12.0;
}

// v[0..5]: min/max slots (even = min, odd = max), v[6..7]: sums -> out[0..7]
template <int NT>
__device__ __forceinline__ void block_reduce8(double (&v)[8], double* __restrict__ out, double (*sRed)[NT / 64]) {
    for (int o = 32; o > 0; o >>= 1) {
        v[0] = fmin(v[0], __shfl_xor(v[0], o));
        v[1] = fmax(v[1], __shfl_xor(v[1], o));
        v[2] = fmin(v[2], __shfl_xor(v[2], o));
        v[3] = fmax(v[3], __shfl_xor(v[3], o));
        v[4] = fmin(v[4], __shfl_xor(v[4], o));
        v[5] = fmax(v[5], __shfl_xor(v[5], o));
        v[6] += __shfl_xor(v[6], o);
        v[7] += __shfl_xor(v[7], o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) sRed[k][wave] = v[k];
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        double r = sRed[k][0];
        for (int w = 1; w < NT / 64; ++w) {
            const double o = sRed[k][w];
            r = (k >= 6) ? r + o : ((k & 1) ? fmax(r, o) : fmin(r, o));
        }
        out[k] = r;
    }
}

template <int TW, int TH, bool FAST, bool DIRS_IN, bool MAG>
__global__ __launch_bounds__(kThreads) void k_georef(georef_args A) {
    constexpr int CW = TW + 1, CH = TH + 1, NC = CW * CH;
    constexpr int NCI = (NC + kThreads - 1) / kThreads;       // corner iterations per thread
    constexpr int NPI = TW * TH / kThreads;                   // pixel iterations per thread
    static_assert(TW * TH % kThreads == 0, "tile must be a multiple of the workgroup");
    __shared__ double sP[3][NC];
    __shared__ double sD[3][NC];
    __shared__ unsigned char sValid[TW * TH];
    __shared__ double sRed[8][kThreads / 64];

    const int tiles_x = (A.width + TW - 1) / TW;
    const int tile_y = blockIdx.x / tiles_x;
    const int tile_x = blockIdx.x - tile_y * tiles_x;
    const int x0 = tile_x * TW, y0 = tile_y * TH;
    const int W1 = A.width + 1;
    const bool want_bbox = A.bbox_partials != nullptr;

    // ---- phase 1: corners -------------------------------------------------------------
    double cla[NCI], clo[NCI];
#pragma unroll
    for (int it = 0; it < NCI; ++it) {
        const int c = threadIdx.x + it * kThreads;
        cla[it] = NAN;
        clo[it] = NAN;
        if (c >= NC) continue;
        const int cy = c / CW, cx = c - cy * CW;
        const int gx = x0 + cx, gy = y0 + cy;
        vec3 d = {NAN, NAN, NAN}, p = {NAN, NAN, NAN};
        if (gx <= A.width && gy <= A.height) {
            const int64_t gi = (int64_t)gy * W1 + gx;
            if (DIRS_IN) {
                d.x = A.dirs_in[3 * gi];
                d.y = A.dirs_in[3 * gi + 1];
                d.z = A.dirs_in[3 * gi + 2];
            } else {
                d = tan_direction_fast(A.wcs, gx - 0.5, gy - 0.5);
            }
            const double t = ray_param_fast(A.ray, d);
            const bool hit = t == t;
            const bool owner = (cx < TW || gx == A.width) && (cy < TH || gy == A.height);
            double la = NAN, lo = NAN;
            if (hit) {
                p = ray_point(A.ray, d, t);
                const vec3 g = mul(A.m_geo, p);
                ecef_to_geodetic_deg_fast(A.bw, g.x, g.y, g.z, la, lo);
            }
            cla[it] = la;
            clo[it] = lo;
            if (owner) {
                if (A.lat) A.lat[gi] = la;
                if (A.lon) A.lon[gi] = lo;
                if (MAG && A.mlat) {
                    double ml = NAN, mt = NAN;
                    if (hit) sm_to_mlat_mlt_fast(mul(A.m_sm, p), ml, mt);
                    A.mlat[gi] = ml;
                    A.mlt[gi] = mt;
                }
            }
        }
        sP[0][c] = p.x;
        sP[1][c] = p.y;
        sP[2][c] = p.z;
        if (FAST) {
            sD[0][c] = d.x;
            sD[1][c] = d.y;
            sD[2][c] = d.z;
        }
    }
    __syncthreads();

    // ---- phase 2: centres ---------------------------------------------------------------
    double nvalid = 0;
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
        const int q = threadIdx.x + it * kThreads;
        const int py = q / TW, px = q - py * TW;
        const int gx = x0 + px, gy = y0 + py;
        bool valid = false;
        if (gx < A.width && gy < A.height) {
            const int c00 = py * CW + px, c01 = c00 + 1, c10 = c00 + CW, c11 = c10 + 1;
            vec3 p, d;
            bool corners_ok = true;
            if (FAST) {
                // reference astrometry.py:154-160: ((c00 + c01) + c11) + c10, then /4
                p.x = (((sP[0][c00] + sP[0][c01]) + sP[0][c11]) + sP[0][c10]) * 0.25;
                p.y = (((sP[1][c00] + sP[1][c01]) + sP[1][c11]) + sP[1][c10]) * 0.25;
                p.z = (((sP[2][c00] + sP[2][c01]) + sP[2][c11]) + sP[2][c10]) * 0.25;
                d.x = (((sD[0][c00] + sD[0][c01]) + sD[0][c11]) + sD[0][c10]) * 0.25;
                d.y = (((sD[1][c00] + sD[1][c01]) + sD[1][c11]) + sD[1][c10]) * 0.25;
                d.z = (((sD[2][c00] + sD[2][c01]) + sD[2][c11]) + sD[2][c10]) * 0.25;
            } else {
                d = tan_direction_fast(A.wcs, (double)gx, (double)gy);
                p = ray_point(A.ray, d, ray_param_fast(A.ray, d));
                if (want_bbox) {
                    // after sanitisation a centre also needs its 4 corners (reference mapping.py:1093-1101)
                    const double s4 = (sP[0][c00] + sP[0][c01]) + (sP[0][c10] + sP[0][c11]);
                    corners_ok = s4 == s4;
                }
            }
            const int64_t gi = (int64_t)gy * A.width + gx;
            double la = NAN, lo = NAN, el = NAN, ml = NAN, mt = NAN;
            if (p.x == p.x) {
                const vec3 g = mul(A.m_geo, p);
                ecef_to_geodetic_deg_fast(A.bw, g.x, g.y, g.z, la, lo);
                // reference astrometry.py:200-212, utils.py:33-46: 90 - angle(-d, P/|P|) = asin(-d.P/|P|)
                double c = -(d.x * p.x + d.y * p.y + d.z * p.z) * fm::rsqrt(dot(p, p));
                c = fmin(1.0, fmax(-1.0, c));
                el = fm::asin_deg(c);
                if (MAG && A.mlat_c) sm_to_mlat_mlt_fast(mul(A.m_sm, p), ml, mt);
            }
            if (A.lat_c) A.lat_c[gi] = la;
            if (A.lon_c) A.lon_c[gi] = lo;
            if (A.elev) A.elev[gi] = el;
            if (MAG && A.mlat_c) {
                A.mlat_c[gi] = ml;
                A.mlt_c[gi] = mt;
            }
            valid = (el >= A.bbox_min_elev) && corners_ok;
        }
        if (want_bbox) {
            sValid[q] = valid ? 1 : 0;
            nvalid += valid ? 1.0 : 0.0;
        }
    }

    // ---- phase 3: bounding box of the corners that keep a valid neighbour centre ----------------
    // (reference mapping.py:845-864 maskedByElevation + 1063-1125 sanitisation + 693-743 boundingBox)
    if (want_bbox) {
        __syncthreads();
        double v[8] = {kInf, -kInf, kInf, -kInf, kInf, -kInf, nvalid, 0};
#pragma unroll
        for (int it = 0; it < NCI; ++it) {
            const int c = threadIdx.x + it * kThreads;
            if (c >= NC) continue;
            const int cy = c / CW, cx = c - cy * CW;
            bool keep = false;
            if (cy > 0 && cx > 0) keep |= sValid[(cy - 1) * TW + cx - 1] != 0;
            if (cy > 0 && cx < TW) keep |= sValid[(cy - 1) * TW + cx] != 0;
            if (cy < TH && cx > 0) keep |= sValid[cy * TW + cx - 1] != 0;
            if (cy < TH && cx < TW) keep |= sValid[cy * TW + cx] != 0;
            const double la = cla[it], lo = clo[it];
            if (keep && la == la) {
                v[0] = fmin(v[0], la);
                v[1] = fmax(v[1], la);
                v[2] = fmin(v[2], lo);
                v[3] = fmax(v[3], lo);
                if (lo > 0) v[4] = fmin(v[4], lo); else v[5] = fmax(v[5], lo);
            }
        }
        block_reduce8<kThreads>(v, A.bbox_partials + (int64_t)blockIdx.x * 8, sRed);
    }
}

// ------------------------------------------------------------------------------------------
// Row-marching variant: no workgroup barriers.
//
// One WAVE owns a strip of 63 pixel columns (64 corner columns, one per lane) and marches down
// `rows` pixel rows.  Each lane casts the ray of its corner column once per corner row and keeps the
// previous row's hit in registers; a pixel's four corners are then (own previous, own current) plus
// the same two of lane+1, fetched with DPP wave shifts.  Redundancy: 64/63 horizontally, (rows+1)/rows
// vertically.  The bounding box is accumulated from the corner side with a one-row delay (a corner
// row is final once the centre row below it has been classified).
//
// All arithmetic is in fx:: (amt_common.h): explicit fma / mul / add in a fixed order, so that every variant of the
// kernel gives the same bits (the two execution plans of the pipeline are tested for bit-identical grids).
// What round 2 measured about this kernel (profiles/NOTEBOOK_r1-r3.md 4.1a): it is not FP64-issue bound any more; straight-line
// "speculative" and per-phase wave-uniform variants of the row step (no NaN presets, no exec-mask regions) were
// built and were SLOWER than this plain divergent form (register pressure, code size), so they were dropped.
// ------------------------------------------------------------------------------------------
constexpr int kDppWaveShl1 = 0x130;   // lane i <- lane i+1
constexpr int kDppWaveShr1 = 0x138;   // lane i <- lane i-1

// (bound_ctrl: a lane without a source lane — lane 63 — reads 0, so the destination needs no initial value and the
// compiler emits no v_mov_b32 0 in front of every DPP move; lane 63's centres are never used: it owns no pixel)
__device__ __forceinline__ double from_next_lane(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), kDppWaveShl1, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), kDppWaveShl1, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int from_next_lane(int v) {
    return __builtin_amdgcn_mov_dpp(v, kDppWaveShl1, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_prev_lane(int v) {   // lane 0 receives 0
    return __builtin_amdgcn_update_dpp(0, v, kDppWaveShr1, 0xf, 0xf, false);
}

// v, or NaN where `keep` is false: ONE v_cndmask_b32 on the high word (0x7ff80000 over any low word is a quiet NaN)
__device__ __forceinline__ double nan_unless(bool keep, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned int hi = keep ? (unsigned int)(b >> 32) : 0x7ff80000u;
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (b & 0xffffffffull)));
}

// NaN in a register pair the optimiser cannot see through: assignments from it stay in the block they are written in
__device__ __forceinline__ double opaque_nan() {
    double n = __builtin_nan("");
    asm volatile("" : "+v"(n));
    return n;
}

// Scalar re-loads of the per-frame constants.  The ~60 constant doubles plus the 64-bit literals of the
// polynomials exceed the 102 SGPRs of a wave; held for the whole loop they are spilled to VGPR lanes
// (v_writelane / v_readlane, 12 % of the kernel's VALU instructions in the first version).  Instead each
// phase of the loop body re-reads the few blocks it needs from the kernel-argument segment through the
// scalar cache; the empty asm makes the base pointer opaque so that the loads cannot be hoisted.
typedef const __attribute__((address_space(4))) unsigned long long* karg_ptr;

// word_offset: where this wave's georef_args starts inside the kernel-argument segment (frames of a batch)
__device__ __forceinline__ karg_ptr karg_fresh(int word_offset = 0) {
    karg_ptr p = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + word_offset;
    asm volatile("" : "+s"(p));
    return p;
}

template <typename T>
__device__ __forceinline__ T karg_load(karg_ptr base, size_t byte_offset) {
    static_assert(sizeof(T) % 8 == 0, "constant blocks are multiples of 8 bytes");
    union {
        T value;
        unsigned long long words[sizeof(T) / 8];
    } u;
    karg_ptr q = base + byte_offset / 8;
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 8; ++i) u.words[i] = q[i];
    return u.value;
}

// Un-normalised direction (GEO frame) of the corner / centre ray from the affine camera model: u = ub + py * uy
// (ub = u0 + px * ux is per lane and fixed for a work item).
__device__ __forceinline__ vec3 affine_ray(double ubx, double uby, double ubz, double uyx, double uyy, double uyz, double py) {
#pragma clang fp contract(off)
    vec3 u = {__builtin_fma(py, uyx, ubx), __builtin_fma(py, uyy, uby), __builtin_fma(py, uyz, ubz)};
    return u;
}

// Ray parameter t >= 0 (in units of |u|) of the first hit of x = o + t u with the shell, NaN or negative for no hit.
// uu = u.u.  14 + 2 + 5 + 1 + 4 operations, v_rsq_f64 + v_rcp_f64.
__device__ __forceinline__ double shell_t(const shell_ray& e, const vec3& u, double uu) {
#pragma clang fp contract(off)
    const double ku = fx::dot3(u.x, u.y, u.z, e.kx, e.ky, e.kz);
    const double uo = fx::dot3(u.x, u.y, u.z, e.ox, e.oy, e.oz);
    const double a2 = __builtin_fma(e.qd * ku, ku, e.qa * uu);
    const double nb = -__builtin_fma(e.qa, uo, e.qd_ko * ku);              // = d_o of the axis-aligned form
    const double disc = __builtin_fma(nb, nb, -(a2 * e.c0));
    const double root = fx::sqrt_n(disc);                                   // NaN when the line misses
    const double t = __builtin_fma(e.root_sign, root, nb);                  // nearer root from outside the shell
    return t * fx::rcp_n(a2);
}

__device__ __forceinline__ vec3 shell_point(const shell_ray& e, const vec3& u, double t) {
#pragma clang fp contract(off)
    vec3 p = {__builtin_fma(u.x, t, e.ox), __builtin_fma(u.y, t, e.oy), __builtin_fma(u.z, t, e.oz)};
    return p;
}

__device__ __forceinline__ vec3 rotate3(const mat3& a, const vec3& v) {
#pragma clang fp contract(off)
    vec3 r = {__builtin_fma(a.m[2], v.z, __builtin_fma(a.m[1], v.y, a.m[0] * v.x)),
              __builtin_fma(a.m[5], v.z, __builtin_fma(a.m[4], v.y, a.m[3] * v.x)),
              __builtin_fma(a.m[8], v.z, __builtin_fma(a.m[7], v.y, a.m[6] * v.x))};
    return r;
}

// Fused binning: every wave keeps a private kBinW x kBinW-cell window of the output grid in LDS, anchored
// at the cell of its first kept pixel (a 63 x 16-pixel strip spans a few cells only).  Pixels add to it with
// LDS atomics as soon as their coordinates exist; pixels outside the window go straight to global atomics;
// the window is flushed with global 64-bit integer atomics when the wave is done.
constexpr int kBinW = 8, kBinCells = kBinW * kBinW;

// BIN: 0 = no fused binning, 1 = uint8 RGB image, 2 = uint16 RGB image
// Minimum waves per SIMD the register allocator must reach: 5 for the georef-only variants, 4 for the fused ones.
#ifndef AMT_ROWS_MIN_WAVES
#define AMT_ROWS_MIN_WAVES 4
#endif
#ifndef AMT_ROWS_MIN_WAVES_BIN
#define AMT_ROWS_MIN_WAVES_BIN 4
#endif
// SECOND: the second pair of angles a point gets besides (lat, lon): 0 none, 1 (MLat, SM longitude) — outputs mlat / mlt
// and, with bin_magnetic, binning and box —, 2 the pole plan (bin_pole): (lat, lon) rotated by 90 deg about x, for binning
// and box only, 3 the pole plan of an MLat / MLT frame (bin_magnetic + bin_pole): (MLat, SM longitude) as in 1 and, for
// binning and box, a third pair: those two rotated by 90 deg about x as if they were geodetic (what resampleMLatMLT does
// with a magnetic pole in view, mapping.py:1519-1547 -> resample.py:176-201)
#ifndef AMT_ROWS_MIN_WAVES_MAG
#define AMT_ROWS_MIN_WAVES_MAG 3        // fused MLat / MLT variants: 271 us per frame at 3 waves (no spills), 278 at 4 (128 VGPRs, 2-3 spills)
#endif
#ifndef AMT_ROWS_MIN_WAVES_POLE
#define AMT_ROWS_MIN_WAVES_POLE 3       // the pole variants need ~150 VGPRs; at 4 waves (128) they spill 40
#endif
// SECOND = 4, "MLat / MLT only": what resampleMLatMLT consumes and nothing else (reference resample.py:63-71,
// mapping.py:1519-1547: mLatMlt, mLatMltCenter, elevation, image) — ray -> shell -> SM rotation -> (MLat, SM longitude)
// small angles -> elevation -> bin on the (MLat, SM longitude) grid.  No Bowring step, no geodetic small angles, no geodetic
// box, no lat / lon / lat_c / lon_c stores (those outputs must be NULL); MLat / MLT / elevation come out bit-identical to
// SECOND = 1 (their arithmetic never touched the geodetic pair).  Fused binning with the camera model only.
#ifndef AMT_ROWS_MIN_WAVES_MAGONLY
#define AMT_ROWS_MIN_WAVES_MAGONLY 4
#endif
// item_order 4: the c-th row of work items in dispatch order -> its row of chunks.  The launch works on two fronts that
// start at the limb and move apart: the Earth side (VALU-bound rows) and the sky side (rows that only store NaN), so that
// both kinds are in flight together for most of the launch while each front keeps writing neighbouring rows (an order
// that scatters the rows in flight over the whole frame, item_order 3, costs the stores their locality: a frame of sky
// alone takes 113 instead of 96 us).  A bijection of [0, n) for any split in [0, n] and e, s >= 1.
__host__ __device__ inline int two_front_chunk(int c, int n, int split, int e, int s, int flip) {
    const int n_e = n - split, n_s = split;
    const int full = min(n_e / e, n_s / s), per = e + s;
    int chunk;
    if (c < full * per) {
        const int q = c / per, r = c - q * per;
        chunk = r < e ? split + q * e + r : split - 1 - (q * s + (r - e));
    } else {
        const int r = c - full * per, rem_e = n_e - full * e;
        chunk = r < rem_e ? split + full * e + r : split - 1 - (full * s + (r - rem_e));
    }
    return flip ? n - 1 - chunk : chunk;
}

template <bool FAST, bool DIRS_IN, int SECOND, int BIN>
__global__ __launch_bounds__(kRowsThreads, SECOND == 4 ? AMT_ROWS_MIN_WAVES_MAGONLY : (SECOND >= 2 ? AMT_ROWS_MIN_WAVES_POLE : (SECOND == 1 && BIN ? AMT_ROWS_MIN_WAVES_MAG : (BIN ? AMT_ROWS_MIN_WAVES_BIN : AMT_ROWS_MIN_WAVES)))) void k_georef_rows(georef_batch B, int rows_per_chunk, int strips_x,
                                                           int n_items, int n_frames) {
    constexpr bool MAG = SECOND != 0, kPole = SECOND == 2, kMagPole = SECOND == 3, kMagOnly = SECOND == 4;
    static_assert(!(kPole || kMagPole) || (BIN != 0 && !DIRS_IN), "the pole plans exist for fused binning with the camera model only");
    static_assert(!kMagOnly || !DIRS_IN, "the MLat / MLT-only mode exists for the camera model only");
    constexpr int kBinWaves = BIN ? kRowsThreads / 64 : 1, kBinSlots = BIN ? kBinCells : 1;
    __shared__ unsigned int sCnt[kBinWaves][kBinSlots];
    __shared__ unsigned int sCh[kBinWaves][3][kBinSlots];
    __shared__ unsigned long long sEl[kBinWaves][kBinSlots];
    // per-lane bounding-box accumulators live in LDS (ds_min_f64 / ds_max_f64 on the lane's own slots: no
    // conflicts, no return value to wait for) instead of 12 VGPRs that would be live across the whole loop
    __shared__ double sBox[kRowsThreads / 64][6][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: keep it scalar
    const int item_all = blockIdx.x * (kRowsThreads / 64) + wave;           // one work item per wave
    if (item_all >= n_items * n_frames) return;                             // wave-uniform
    // several equally sized frames can share one launch: n_items items each, frame after frame
    const int frame = item_all / n_items;
    const int item = item_all - frame * n_items;
    const georef_args& A = B.f[frame];
    const int koff = frame * (int)(sizeof(georef_args) / 8);                // this frame's block of constants
    int bin_ax0 = 0, bin_ay0 = 0;
    bool bin_anchor = false;                                                // wave-uniform
    if (BIN) {
#pragma unroll
        for (int i = lane; i < kBinCells; i += 64) {
            sCnt[wave][i] = 0;
            sEl[wave][i] = 0;
            sCh[wave][0][i] = 0;
            sCh[wave][1][i] = 0;
            sCh[wave][2][i] = 0;
        }
    }
    int run_key = 0;                                    // (bin_x << 16 | bin_y) of the lane's current run, 0 = none
    unsigned int run_cnt = 0, run_c0 = 0, run_c1 = 0, run_c2 = 0;
    long long run_el = 0;
    auto bin_flush = [&](int key, unsigned int cnt, unsigned int c0, unsigned int c1, unsigned int c2, long long el) {
        const int bx = key >> 16, by = key & 0xffff;
        const int dx = bx - bin_ax0, dy = by - bin_ay0;
        if (dx >= 0 && dx < kBinW && dy >= 0 && dy < kBinW) {
            const int slot = dx * kBinW + dy;
            atomicAdd(&sCnt[wave][slot], cnt);
            atomicAdd(&sCh[wave][0][slot], c0);
            atomicAdd(&sCh[wave][1][slot], c1);
            atomicAdd(&sCh[wave][2][slot], c2);
            atomicAdd(&sEl[wave][slot], (unsigned long long)el);
        } else {
            // outside the wave's window (very fine grids): straight to the global accumulators
            karg_ptr K = karg_fresh(koff);
            const int nby = karg_load<axis_lin>(K, offsetof(georef_args, byl)).nbin;
            const int64_t ncell = (int64_t)karg_load<axis_lin>(K, offsetof(georef_args, bxl)).nbin * nby;
            unsigned long long* acc = karg_load<unsigned long long*>(K, offsetof(georef_args, bin_acc));
            const int64_t cell = (int64_t)(bx - 1) * nby + (by - 1);
            atomicAdd(&acc[cell], (unsigned long long)cnt);
            atomicAdd(&acc[ncell + cell], (unsigned long long)c0);
            atomicAdd(&acc[2 * ncell + cell], (unsigned long long)c1);
            atomicAdd(&acc[3 * ncell + cell], (unsigned long long)c2);
            atomicAdd(&acc[4 * ncell + cell], (unsigned long long)el);
        }
    };
    // item -> (chunk of rows, strip of columns); the order only decides which items the dispatcher starts first
    const int chunks_y = n_items / strips_x;
    int chunk, strip;
    chunk = item / strips_x;
    strip = item - chunk * strips_x;
    if (A.item_order == 2) chunk = chunks_y - 1 - chunk;
    if (A.item_order == 3) chunk = (int)(((long long)chunk * A.chunk_stride) % chunks_y);
    if (A.item_order == 4) chunk = two_front_chunk(chunk, chunks_y, A.front_split, A.front_e, A.front_s, A.front_flip);
    const int x0 = strip * 63, y0 = chunk * rows_per_chunk;
    const int rows = min(rows_per_chunk, A.height - y0);
    const int gx = x0 + lane;
    const int W1 = A.width + 1;
    const bool col_ok = gx <= A.width;                 // this lane's corner column exists
    const bool px_ok = lane < 63 && gx < A.width;      // this lane's pixel column exists (and is owned)
    const bool want_bbox = A.bbox_partials != nullptr;

    if (!DIRS_IN && (chunk < A.sky_top_end || chunk >= A.sky_bottom_begin)) {      // wave-uniform
        // No ray of this item reaches the shell (the host has bounded the conic section that the limb is in the image,
        // sky_bands()): its part of every output array is NaN, nothing is binned, its box partial is the empty one.
        // The rows of a whole row of items are one contiguous range of each array, so the strips_x waves of the row
        // share it as contiguous pieces written with 16-byte stores (the pattern of a plain fill: 6 TB/s where the
        // 504-byte runs of the row-marching pattern reach 5).
        const int n_corner_rows = rows + (y0 + rows == A.height ? 1 : 0);           // the image's last corner row too
        auto fill_nan = [&](double* base, int64_t first, int64_t count) {
            if (base == nullptr) return;
            int64_t a = first + count * strip / strips_x, b = first + count * (strip + 1) / strips_x;
            // 16-byte alignment of the pairs, from the address itself: an output array may start at any multiple of 8 bytes
            // (a row slice of a larger array handed to the C API)
            if ((reinterpret_cast<uintptr_t>(base + a) & 15) != 0 && a < b) {
                if (lane == 0) base[a] = NAN;
                ++a;
            }
            const double2 two = {NAN, NAN};
            for (int64_t i = a + 2 * lane; i + 1 < b; i += 128) *reinterpret_cast<double2*>(base + i) = two;
            if (((b - a) & 1) && lane == 0 && b > a) base[b - 1] = NAN;
        };
        const bool mag_out = MAG && !kPole;
        // (strip-padded rows: the pad columns are part of the contiguous range and become NaN too)
        const int64_t cpitch = A.row_layout ? (int64_t)strips_x * 64 : W1, ppitch = A.row_layout ? (int64_t)strips_x * 64 : A.width;
        const int64_t c0 = (int64_t)y0 * cpitch, cn = (int64_t)n_corner_rows * cpitch;
        const int64_t p0 = (int64_t)y0 * ppitch, pn = (int64_t)rows * ppitch;
        fill_nan(A.lat, c0, cn);
        fill_nan(A.lon, c0, cn);
        fill_nan(A.lat_c, p0, pn);
        fill_nan(A.lon_c, p0, pn);
        fill_nan(A.elev, p0, pn);
        if (mag_out) {
            fill_nan(A.mlat, c0, cn);
            fill_nan(A.mlt, c0, cn);
            fill_nan(A.mlat_c, p0, pn);
            fill_nan(A.mlt_c, p0, pn);
        }
        if (want_bbox && lane < 8)
            A.bbox_partials[(int64_t)item * 8 + lane] = lane >= 6 ? 0.0 : ((lane & 1) ? -kInf : kInf);
        return;
    }

    // The constants of the common path are read ONCE, before the row loop, and stay in SGPRs (the compiler parks what
    // does not fit in VGPR lanes, a v_readlane per use).  Re-reading them from the kernel-argument segment in every
    // phase of every row, as the first version did to save registers, made the scalar data cache the bottleneck of
    // the whole kernel: an s_load that hits costs ~320 cycles of latency with 4 waves per SIMD and the cache serves
    // a CU about one request per 20 cycles (tools/sload_latency.hip), and a row made 23 of them.
    // (Each value is passed through an empty asm: the compiler must then keep that very value alive; otherwise it
    // prefers to re-load "invariant" kernel arguments inside the loop whenever SGPRs run short.)
    // AMT_ROWS_PIN (A/B builds): 0 = nothing is pinned, 1 = what a row that misses the shell needs (camera model, ray),
    // 2 = everything
#ifndef AMT_ROWS_PIN
#define AMT_ROWS_PIN 1
#endif
    auto pin = [](auto v) {
        if (AMT_ROWS_PIN >= 1) asm volatile("" : "+s"(v));
        return v;
    };
    auto pin2 = [](auto v) {
        if (AMT_ROWS_PIN >= 2) asm volatile("" : "+s"(v));
        return v;
    };
    affine_cam cm = A.cam;
    cm.uy[0] = pin(cm.uy[0]), cm.uy[1] = pin(cm.uy[1]), cm.uy[2] = pin(cm.uy[2]), cm.cy = pin(cm.cy);
    shell_ray ray = A.sray;
    ray.qa = pin(ray.qa), ray.qd = pin(ray.qd), ray.kx = pin(ray.kx), ray.ky = pin(ray.ky), ray.kz = pin(ray.kz);
    ray.qd_ko = pin(ray.qd_ko), ray.c0 = pin(ray.c0), ray.ox = pin(ray.ox), ray.oy = pin(ray.oy), ray.oz = pin(ray.oz);
    ray.root_sign = pin(ray.root_sign);
    bowring_fast bw = A.bw;
    bw.b_over_a = pin2(bw.b_over_a), bw.d = pin2(bw.d), bw.e2a = pin2(bw.e2a);
    const double min_elev = pin2(A.bbox_min_elev);
    double* const out_lat = pin2(A.lat);
    double* const out_lon = pin2(A.lon);
    double* const out_lat_c = pin2(A.lat_c);
    double* const out_lon_c = pin2(A.lon_c);
    double* const out_elev = pin2(A.elev);
    const unsigned char* const img_base = pin2(static_cast<const unsigned char*>(A.bin_img));
    const double bx_e0 = pin2(A.bxl.e0), bx_inv = pin2(A.bxl.inv_step), bx_margin = pin2(A.bxl.margin), by_e0 = pin2(A.byl.e0),
                 by_inv = pin2(A.byl.inv_step), by_margin = pin2(A.byl.margin);
    const int bx_n = pin2(A.bxl.nbin), by_n = pin2(A.byl.nbin);
    const bool lon_wrap = pin2(A.bin_lon_wrap) != 0;
    const double sm0 = pin2(B.math.small4[0]), sm1 = pin2(B.math.small4[1]), sm2 = pin2(B.math.small4[2]), sm3 = pin2(B.math.small4[3]);
    const int frame_h = pin2(A.height);
    // State of the previous corner row.  The row loop is unrolled by two with the roles of S0 / S1 swapped, so
    // that no prev <- cur register moves are needed.
    struct row_state {
        vec3 p, d;
        double la, lo;
        double bn, bd;      // Bowring numerator / denominator of the corner: lat = atan(bn / bd)
        double bla, blo;    // what the bounding box is reduced over when that is not (la, lo): MLat / SM longitude
        vec3 s;             // MAG: the corner in SM coordinates, |(s.x, s.y)|, MLat and SM longitude (degrees)
        double sxy, ml, sl;
        vec3 r;             // kMagPole: the rotated point of (MLat, SM longitude) as pole_point gives it, its angles
        double rxy, rla, rlo;
        int flag;
    };
    row_state S0 = {{NAN, NAN, NAN}, {NAN, NAN, NAN}, NAN, NAN, NAN, NAN, NAN, NAN, {NAN, NAN, NAN}, NAN, NAN, NAN,
                    {NAN, NAN, NAN}, NAN, NAN, NAN, 0}, S1 = S0;
    // Neighbouring pixels differ by a fraction of a degree, so latitude and longitude of a corner are taken as
    // the previous row's plus a small angle (fx::small_angles: one reciprocal and two 4-term series instead of two
    // range-reduced 9-term arctangents), and a centre's as its corner's plus a small angle.  The full arctangent
    // runs where that does not apply: first row of a chunk, previous row missed the shell, steps above 1.7 deg at
    // the limb, within 2 deg of the date line.
    constexpr bool kMagBox = (MAG && BIN != 0) || kMagOnly;      // the box may be asked for in (MLat, SM longitude): bla / blo
    // Pole plan (bin_pole; SECOND = 2, the machinery of the MLat / MLT variants on another second pair of angles): the reference rotates a
    // frame with a pole in view by +90 deg about x before binning (resample.py:176-201: geodetic -> ECEF at the mapping
    // altitude -> rotation -> geodetic).  Here: the point is rebuilt from the Bowring numerator / denominator of its
    // latitude and its (x, y) — no trigonometry —, turned (x, y, z) -> (x, -z, y), and its rotated latitude /
    // longitude follow as the previous point's plus two small angles, exactly like MLat / SM longitude.  That locates a
    // pixel to ~1e-11 deg; the few pixels within 1e-7 bins of an edge are re-evaluated with rotate_pole_deg, the very
    // function the two-pass plan uses.
    constexpr bool pole_bin = kPole;
    const bool geo_pole = kMagPole && !A.bin_magnetic;      // wave-uniform: SECOND = 3 on a geodetic grid (see the corner step)
    // a centre's coordinates are computed when somebody takes them: the fused binning, or an output array.  The box pass of
    // the box-first plan (amt_pipe_launch_box: no output, no binning) needs a centre's elevation only — the box is reduced
    // over the corners — and skips a quarter of the arithmetic (wave-uniform)
    const bool centre_coords = BIN != 0 || A.lat_c != nullptr || A.lon_c != nullptr || (MAG && A.mlat_c != nullptr);
    int n_valid = 0;
    auto box_add = [&](double la_v, double lo_v) {
        __hip_atomic_fetch_min(&sBox[wave][0][lane], la_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(&sBox[wave][1][lane], la_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_min(&sBox[wave][2][lane], lo_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(&sBox[wave][3][lane], lo_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lo_v > 0)
            __hip_atomic_fetch_min(&sBox[wave][4][lane], lo_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            __hip_atomic_fetch_max(&sBox[wave][5][lane], lo_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (want_bbox) {
#pragma unroll
        for (int k = 0; k < 6; ++k) sBox[wave][k][lane] = (k & 1) ? -kInf : kInf;
    }
    // Image pixels are loaded ahead, two rows per wait: loads and stores share one counter and the compiler waits
    // for all of them when both kinds are in flight, so every wait for a pixel also waits for the wave's own
    // earlier stores.  The even step of each pair of rows waits once (right before its first store, see below),
    // takes its own pixel (A), sets the odd step's pixel (B) aside and issues the loads for the next pair.
    unsigned int rawA0 = 0, rawA1 = 0, rawB0 = 0, rawB1 = 0;      // raw words as loaded; unpacked only after the wait
    unsigned int curB0 = 0, curB1 = 0;
    int shiftA = 0, shiftB = 0, curShiftB = 0;
    // byte offsets into the image fit 32 bits (the host checks): half the address arithmetic of 64-bit indices
    const unsigned int img_last = (unsigned int)((int64_t)A.width * A.height * (BIN == 1 ? 3 : 6) - 8);
    auto load_pixel = [&](unsigned int byte_offset, unsigned int& w0, unsigned int& w1, int& shift) {
        // one aligned 8-byte load covers the 3 (uint8) or 6 (uint16) bytes of the pixel wherever it starts; the
        // address is clamped so that the load never reaches past the image
        unsigned int a = byte_offset & ~3u;
        a = a > img_last ? img_last : a;
        shift = (int)(byte_offset - a) * 8;
        uint2 w2;
        __builtin_memcpy(&w2, img_base + a, 8);
        w0 = w2.x;
        w1 = w2.y;
    };
    auto unpack_pixel = [&](unsigned int w0, unsigned int w1, int shift, unsigned int& c0, unsigned int& c1, unsigned int& c2) {
        if (BIN == 1) {
            const unsigned int w = (unsigned int)((((unsigned long long)w1 << 32) | w0) >> shift);   // shift <= 40
            c0 = w & 0xffu, c1 = (w >> 8) & 0xffu, c2 = (w >> 16) & 0xffu;
        } else {
            // shift is 0 or 16 (or 16 at the clamped end): two 32-bit funnel shifts
            const unsigned int lo = __builtin_amdgcn_alignbit(w1, w0, (unsigned int)shift), hi = w1 >> shift;
            c0 = lo & 0xffffu, c1 = lo >> 16, c2 = hi & 0xffffu;
        }
    };

    // byte offsets of this lane's corner (row gy) and pixel (row gy-1) inside their arrays, advanced by one row per
    // step.  They fit 32 bits (the host checks), so every store is `base in SGPRs + 32-bit lane offset`.
    // Strip-padded rows (amt_georef_out.row_layout = AMT_ROWS_STRIP_PADDED; round 6): strip s of a row lies at doubles
    // [64 s, 64 s + 64) of a row of 64 strips_x doubles and ALL 64 lanes store — a corner array's 64th value is the next strip's
    // first corner (the same ray), a pixel array's a pad —, so that every run a wave writes is 512 bytes on a 512-byte
    // boundary: whole 128-byte lines only.  With contiguous rows a run is 504 bytes from an odd multiple of 8 and its first and
    // last line are shared with the neighbour strips' waves, which write their parts at another time: the memory system
    // takes such partial lines at well under the rate of whole ones (tools/store_pattern.hip, profiles/r6/).
    const bool padded = A.row_layout != 0;                                       // wave-uniform
    const unsigned int pitch_corner = padded ? (unsigned int)strips_x * 512u : (unsigned int)W1 * 8u;
    const unsigned int pitch_pixel = padded ? (unsigned int)strips_x * 512u : (unsigned int)A.width * 8u;
    const unsigned int col_bytes = padded ? (unsigned int)(strip * 64 + lane) * 8u : (unsigned int)gx * 8u;
    unsigned int off_corner = (unsigned int)y0 * pitch_corner + col_bytes;
    unsigned int off_pixel = (unsigned int)(y0 - 1) * pitch_pixel + col_bytes;  // row -1 wraps; used from row 0 on only
    // lanes that store: with contiguous rows the owners of a column (lane 63's corner column belongs to the next strip unless
    // it is the last, its pixel column does not exist); with strip-padded rows every lane
    const bool st_corner = padded || (col_ok && (lane < 63 || gx == A.width));
    const bool st_pixel = padded || px_ok;
    auto at = [](double* base, unsigned int byte_offset) -> double& {
        return *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_offset);
    };
    // byte offset of pixel (gy-1, gx) in the image and of one image row (unsigned wrap-around for row -1 is harmless:
    // only offset + k rows with k >= 1 is ever used)
    const unsigned int img_row = (unsigned int)A.width * (BIN == 1 ? 3u : 6u);
    unsigned int img_off = (unsigned int)((y0 - 1) * A.width + gx) * (BIN == 1 ? 3u : 6u);
    // this lane's share of the affine camera model (see affine_ray): corner column gx - 1/2; the centre of pixel
    // (row - 1, gx) is the corner (row, gx) plus half a column minus half a row
    double ubx = 0, uby = 0, ubz = 0;
    if (!DIRS_IN || !FAST) {
#pragma clang fp contract(off)
        const double px = (double)gx + cm.cx;
        ubx = __builtin_fma(px, cm.ux[0], cm.u0[0]), uby = __builtin_fma(px, cm.ux[1], cm.u0[1]);
        ubz = __builtin_fma(px, cm.ux[2], cm.u0[2]);
    }
    const double hcx = 0.5 * (cm.ux[0] - cm.uy[0]), hcy = 0.5 * (cm.ux[1] - cm.uy[1]), hcz = 0.5 * (cm.ux[2] - cm.uy[2]);
    // directions-in variant: the corner direction of the row after the current one
    double dj0 = NAN, dj1 = NAN, dj2 = NAN;
    if (DIRS_IN && col_ok) {
        const double* q = A.dirs_in + 3 * ((int64_t)y0 * W1 + gx);
        dj0 = q[0], dj1 = q[1], dj2 = q[2];
    }

    // ---- pieces shared by the two paths ---------------------------------------------------------------------
    typedef double coef4[4];
    typedef double coef9[9];
    struct c4 { coef4 c; };
    struct c9 { coef9 c; };
    auto small_table = [&]() {
        c4 k = {{sm0, sm1, sm2, sm3}};
        return k;
    };
    auto atan_table = [&]() { return karg_load<c9>(karg_fresh(), offsetof(georef_batch, math) + offsetof(fx::math_table, atan9)); };
    auto full_angles = [&](double n, double d, double x, double y, double& la_v, double& lo_v) {
        const c9 k = atan_table();
        la_v = fx::atan_pos_deg(n, d, k.c);
        lo_v = fx::atan2_deg(y, x, k.c);
    };
    // GEO point -> SM cartesian, |(x, y)|; then (MLat deg, SM longitude deg) as the reference point's plus two small
    // angles (like latitude / longitude), the full arctangents where that does not apply.  MLT = SM longitude *
    // 24/360 + 12 (reference transform.py:104-127,373-386,421-427).
    auto sm_point = [&](const vec3& pt, vec3& sv, double& sxy) {
#pragma clang fp contract(off)
        sv = rotate3(karg_load<mat3>(karg_fresh(koff), offsetof(georef_args, m_geo_sm)), pt);
        const double q = __builtin_fma(sv.x, sv.x, sv.y * sv.y);
        sxy = q > 0 ? fx::sqrt_n(q) : 0.0;
    };
    // pole plan: geodetic latitude atan(n / d) and planar position (x, y) at the mapping altitude -> the rotated point's
    // (x, y, Bowring numerator) in `sv` and its Bowring denominator in `sxy`, which is what sm_angles works on
    auto pole_point = [&](double n, double d, double x, double y, vec3& sv, double& sxy) {
#pragma clang fp contract(off)
        const pole_consts pc = karg_load<pole_consts>(karg_fresh(koff), offsetof(georef_args, pole));
        const double rs = fx::rsqrt_n(__builtin_fma(n, n, d * d));
        const double sphi = n * rs, cphi = d * rs;
        const double q = __builtin_fma(x, x, y * y);
        const double ip = q > 0 ? fx::rsqrt_n(q) : 0.0;
        const double big_n = pc.w.a * fx::rsqrt_n(__builtin_fma(-(pc.e2 * sphi), sphi, 1.0));
        const double rc = ((big_n + pc.alt) * cphi) * ip;
        const double gx = rc * x, gy = rc * y, gz = __builtin_fma(big_n, pc.one_minus_e2, pc.alt) * sphi;
        double rn, rd, ir;
        fx::bowring_nd(bw, gx, -gz, gy, rn, rd, ir);
        sv.x = gx, sv.y = -gz, sv.z = rn;
        sxy = rd;
    };
    // `live`: the point exists (lanes without one carry NaN through the small-angle path and must not ask for the full one)
    auto sm_angles = [&](const vec3& ref, double ref_sxy, double ref_ml, double ref_sl, const vec3& sv, double sxy, bool live,
                         double& ml, double& sl) {
        double dml, dsl;
        bool ok;
        fx::small_angles(ref_sxy, ref.z, sxy, sv.z, ref.x, ref.y, sv.x, sv.y, small_table().c, dml, dsl, ok);
        ok = ok && fabs(ref_sl) < 178.0;
        ml = ref_ml + dml;
        sl = ref_sl + dsl;
        if (live && !ok) {
            const c9 k = atan_table();
            ml = fx::atan_pos_deg(sv.z, sxy, k.c);
            sl = fx::atan2_deg(sv.y, sv.x, k.c);
        }
    };
    // un-normalised direction of this lane's corner of row gy (GEO frame)
    auto corner_ray = [&](int gy, const vec3& dj) -> vec3 {
        if (DIRS_IN) return rotate3(karg_load<mat3>(karg_fresh(koff), offsetof(georef_args, m_geo)), dj);       // J2000 -> GEO
        return affine_ray(ubx, uby, ubz, cm.uy[0], cm.uy[1], cm.uy[2], (double)gy + cm.cy);
    };
    // the image pixel of this step's centre row (see the comment at rawA0)
    auto take_pixel = [&](int r, bool even, unsigned int& ch0, unsigned int& ch1, unsigned int& ch2) {
        if (even) {
            asm volatile("" : "+v"(rawA0), "+v"(rawA1), "+v"(rawB0), "+v"(rawB1));
            unpack_pixel(rawA0, rawA1, shiftA, ch0, ch1, ch2);      // pixel row gy-1: this step's centres
            curB0 = rawB0, curB1 = rawB1, curShiftB = shiftB;       // pixel row gy: the odd step's centres
            // next pair: pixel rows gy+1 (even step r+2) and gy+2 (odd step r+3)
            if (px_ok && r + 1 < rows) load_pixel(img_off + 2 * img_row, rawA0, rawA1, shiftA);
            if (px_ok && r + 2 < rows) load_pixel(img_off + 3 * img_row, rawB0, rawB1, shiftB);
        } else {
            unpack_pixel(curB0, curB1, curShiftB, ch0, ch1, ch2);
        }
    };
    // a pixel with its 1-based bins (0 = not binned) joins the lane's run; the run goes to the wave's LDS window
    // when the cell changes
    auto bin_account = [&](int bin_x, int bin_y, unsigned int ch0, unsigned int ch1, unsigned int ch2, long long el_fix) {
        const unsigned long long m = __ballot(bin_x > 0);
        if (m && !bin_anchor) {          // first kept pixel of this wave: centre the window on its cell
            const int src = __builtin_ctzll(m);
            bin_ax0 = __shfl(bin_x, src) - kBinW / 2;
            bin_ay0 = __shfl(bin_y, src) - kBinW / 2;
            bin_anchor = true;
        }
        // A lane walks down one pixel column: consecutive rows mostly stay in one cell, so the lane sums
        // that run in registers and only touches the (conflict-prone) LDS window when the cell changes.
        if (bin_x > 0) {
            const int key = (bin_x << 16) | bin_y;
            if (key != run_key) {
                if (run_key) bin_flush(run_key, run_cnt, run_c0, run_c1, run_c2, run_el);
                run_key = key;
                run_cnt = 0;
                run_c0 = run_c1 = run_c2 = 0;
                run_el = 0;
            }
            run_cnt += 1;
            run_c0 += ch0;
            run_c1 += ch1;
            run_c2 += ch2;
            run_el += el_fix;
        }
    };
    // bins of a valid pixel at (bxv, byv): common path without edges; anything near an edge or outside the axes
    // sets `slow` (decided exactly by bin_slow)
    auto bin_common = [&](double bxv, double byv, int& bx, int& by, bool& slow) {
        bool slow_x, slow_y;
        bx = bin_fast(bx_e0, bx_inv, bx_margin, bx_n, bxv, slow_x);
        by = bin_fast(by_e0, by_inv, by_margin, by_n, byv, slow_y);
        slow = slow_x || slow_y;
    };
    auto bin_slow = [&](double bxv, double byv, int& bx, int& by, unsigned int& edge_flags) {
        karg_ptr K = karg_fresh(koff);
        const axis_dev ax = karg_load<axis_dev>(K, offsetof(georef_args, bax));
        bx = bin_index<true>(ax, bxv);
        bx = bx > ax.nbin ? 0 : bx;
        if (bx > 0 && on_lower_edge(ax, bx, bxv)) edge_flags |= 1u;
        const axis_dev ay = karg_load<axis_dev>(K, offsetof(georef_args, bay));
        by = bin_index<true>(ay, byv);
        by = by > ay.nbin ? 0 : by;
        if (by > 0 && on_lower_edge(ay, by, byv)) edge_flags |= 2u;
    };

    // ---- one corner row + the centre row above it --------------------------------------------------------------
    // There is no lane-divergent region around the arithmetic (round 5).  A ray that misses the shell gets the parameter
    // NaN, and the NaN runs through the whole chain — point, Bowring pair, small angles, sums, elevation — so a miss comes
    // out as NaN without a "preset NaN, overwrite under the exec mask" pair per value (the earlier form spent 34 + 20
    // v_mov_b32 per row step on those presets, a sixth of its VALU instructions).  What NaN does NOT pass through by
    // itself is guarded: the comparisons that select the full arctangents (`live`), v_min / v_max of the elevation clamp
    // (nan_unless), the `q > 0 ? sqrt : 0` forms (their products with the NaN point are NaN again).  Rows of a wave none of
    // whose lanes hits take a wave-uniform shortcut.
    struct corner_vals {
        vec3 p;
        double la, lo, bn, bd, bla, blo;
        vec3 sv;
        double sxy, sml, ssl;
        vec3 rv;
        double rxy, rla, rlo;
    };
    auto step = [&](const int r, const bool even, const vec3& dj, const row_state& prev, row_state& cur) {
        const int gy = y0 + r;
        unsigned int ch0 = 0, ch1 = 0, ch2 = 0;          // image pixel (gy-1, gx)
        // ---- corner (gy, gx) ------------------------------------------------------------------
        // (lanes beyond the last corner column: the camera model extrapolates, caller-supplied directions are NaN there;
        // either way they count as misses and own nothing)
        const shell_ray& ry = ray;
        const vec3 u = corner_ray(gy, dj);
        const double uu = fx::dot3(u.x, u.y, u.z, u.x, u.y, u.z);
        const double t = shell_t(ry, u, uu);
        const bool hit = col_ok && t >= 0.0;
        // unit direction (what the centre's elevation averages, reference astrometry.py:154-160); caller-supplied
        // directions are used as they are, like the reference does
        vec3 d;
        if (DIRS_IN) {
            d = u;
        } else {
#pragma clang fp contract(off)
            const double rs = fx::rsqrt_n(uu);
            d.x = u.x * rs, d.y = u.y * rs, d.z = u.z * rs;
        }
        corner_vals c;
        // with bin_magnetic the bounding box is reduced over (MLat, SM longitude) as well
        const bool magbox = MAG && (kMagOnly || (BIN && !kPole && A.bin_magnetic));
        if (__builtin_amdgcn_ballot_w64(hit) != 0) {     // wave-uniform
            c.p = shell_point(ry, u, nan_unless(hit, t));         // already in GEO; NaN for a miss
            c.la = c.lo = c.bn = c.bd = NAN;
            if (!kMagOnly) {
                double ir;
                fx::bowring_nd(bw, c.p.x, c.p.y, c.p.z, c.bn, c.bd, ir);
                double dla, dlo;
                bool ok;
                fx::small_angles(prev.bd, prev.bn, c.bd, c.bn, prev.p.x, prev.p.y, c.p.x, c.p.y, small_table().c, dla, dlo, ok);
                ok = ok && fabs(prev.lo) < 178.0;
                c.la = prev.la + dla;
                c.lo = prev.lo + dlo;
                if (hit && !ok) full_angles(c.bn, c.bd, c.p.x, c.p.y, c.la, c.lo);
            }
            c.sv = {NAN, NAN, NAN}, c.sxy = c.sml = c.ssl = NAN;
            c.rv = {NAN, NAN, NAN}, c.rxy = c.rla = c.rlo = NAN;
            if (MAG) {
                if (pole_bin) pole_point(c.bn, c.bd, c.p.x, c.p.y, c.sv, c.sxy); else sm_point(c.p, c.sv, c.sxy);
                sm_angles(prev.s, prev.sxy, prev.ml, prev.sl, c.sv, c.sxy, hit, c.sml, c.ssl);
                if (kMagPole) {
                    // MLat = atan(s.z / |s.xy|), SM longitude = atan2(s.y, s.x): the same construction on the SM vector;
                    // without bin_magnetic (a geodetic grid with the pole in view whose caller also wants the MLat /
                    // MLT arrays) the rotated pair is that of (lat, lon), as in the SECOND = 2 variant
                    if (geo_pole) pole_point(c.bn, c.bd, c.p.x, c.p.y, c.rv, c.rxy); else pole_point(c.sv.z, c.sxy, c.sv.x, c.sv.y, c.rv, c.rxy);
                    sm_angles(prev.r, prev.rxy, prev.rla, prev.rlo, c.rv, c.rxy, hit, c.rla, c.rlo);
                }
            }
        } else {
            // a row of sky inside a chunk that sees the Earth: everything is NaN (taken from an opaque register pair, so
            // that these assignments stay in this rarely taken block instead of becoming presets in front of the branch)
            const double n = opaque_nan();
            c.p = {n, n, n}, c.la = c.lo = c.bn = c.bd = n;
            c.sv = {n, n, n}, c.sxy = c.sml = c.ssl = n;
            c.rv = {n, n, n}, c.rxy = c.rla = c.rlo = n;
        }
        const double mt_corner = MAG ? c.ssl * (24.0 / 360.0) + 12.0 : NAN;
        c.bla = c.blo = NAN;
        if (MAG) {
            if (magbox) {
                c.bla = c.sml;
                c.blo = (mt_corner - 12.0) / (24.0 / 360.0);       // mltToSmLon, reference transform.py:388-401
            }
            if (kMagPole) c.bla = c.rla, c.blo = c.rlo;
            if (pole_bin) c.bla = c.sml, c.blo = c.ssl;
            if (kMagBox && !magbox && !pole_bin && !kMagPole) c.bla = c.la, c.blo = c.lo;
        }
        if (BIN) take_pixel(r, even, ch0, ch1, ch2);
        // the last corner row of a chunk is the first of the next one (which owns it) unless it is
        // the image's last; lane 63's column likewise belongs to the next strip unless it is the last
        const bool owner = st_corner && (r < rows || gy == frame_h);
        if (owner) {
            if (!kMagOnly) {
                if (out_lat) at(out_lat, off_corner) = c.la;
                if (out_lon) at(out_lon, off_corner) = c.lo;
            }
            if (MAG && !kPole && A.mlat) {
                at(A.mlat, off_corner) = c.sml;
                at(A.mlt, off_corner) = mt_corner;
            }
        }
        int flag_cur = 0;
        if (r > 0) {
            // ---- centre (gy-1, gx): corners own/next lane x previous/current row -----------------
            vec3 pc, dsum;
            double dscale;               // centre direction = dsum * dscale
            bool corners_ok = true;
            if (FAST) {
                // mean of the 4 corner hits / directions (reference astrometry.py:154-160); the summation
                // order differs from the reference's by rounding only (<= 1e-12 deg)
                const double sx = prev.p.x + c.p.x, sy = prev.p.y + c.p.y, sz = prev.p.z + c.p.z;
                const double tx = prev.d.x + d.x, ty = prev.d.y + d.y, tz = prev.d.z + d.z;
                pc.x = (sx + from_next_lane(sx)) * 0.25;
                pc.y = (sy + from_next_lane(sy)) * 0.25;
                pc.z = (sz + from_next_lane(sz)) * 0.25;
                dsum.x = tx + from_next_lane(tx);
                dsum.y = ty + from_next_lane(ty);
                dsum.z = tz + from_next_lane(tz);
                dscale = 0.25;
            } else {
                // the pixel's own ray (exact centres): the corner's plus half a column minus half a row
                vec3 uc;
                double uuc;
                {
#pragma clang fp contract(off)
                    uc.x = u.x + hcx, uc.y = u.y + hcy, uc.z = u.z + hcz;
                    uuc = fx::dot3(uc.x, uc.y, uc.z, uc.x, uc.y, uc.z);
                    const double rs = fx::rsqrt_n(uuc);
                    dsum.x = uc.x * rs, dsum.y = uc.y * rs, dsum.z = uc.z * rs;
                }
                dscale = 1.0;
                const double tc = shell_t(ry, uc, uuc);
                pc = shell_point(ry, uc, nan_unless(tc >= 0.0, tc));
                if (want_bbox) {
                    // after sanitisation a centre also needs its 4 corners (reference mapping.py:1093-1101)
                    const int h = (prev.p.x == prev.p.x) && (c.p.x == c.p.x);
                    corners_ok = h && from_next_lane(h);
                }
            }
            const bool live = px_ok && pc.x == pc.x;          // this lane's pixel exists and its centre hit the shell
            double lac, loc, el, ml = NAN, mt = NAN, slc = NAN, rlac = NAN, rloc = NAN;
            if (__builtin_amdgcn_ballot_w64(live) != 0) {      // wave-uniform
                double inv_r, cn = NAN, cd = NAN;
                if (kMagOnly || !centre_coords) {
                    // 1 / |P| exactly as the Bowring step computes it (fx::bowring_nd): the elevation keeps its bits
#pragma clang fp contract(off)
                    inv_r = fx::rsqrt_n(__builtin_fma(pc.z, pc.z, __builtin_fma(pc.x, pc.x, pc.y * pc.y)));
                    lac = loc = kMagOnly ? NAN : opaque_nan();
                } else {
                    fx::bowring_nd(bw, pc.x, pc.y, pc.z, cn, cd, inv_r);
                    // relative to this lane's corner of the current row
                    double dla, dlo;
                    bool ok;
                    fx::small_angles(c.bd, c.bn, cd, cn, c.p.x, c.p.y, pc.x, pc.y, small_table().c, dla, dlo, ok);
                    ok = ok && fabs(c.lo) < 178.0;
                    lac = c.la + dla;
                    loc = c.lo + dlo;
                    if (live && !ok) full_angles(cn, cd, pc.x, pc.y, lac, loc);
                }
                // reference astrometry.py:200-212, utils.py:33-46: 90 - angle(-d, P/|P|) = asin(-d.P/|P|)
                // (dot products do not depend on the frame; 1/|P| is a by-product of the Bowring step)
                double cs = -(fx::dot3(dsum.x, dsum.y, dsum.z, pc.x, pc.y, pc.z) * dscale) * inv_r;
                cs = nan_unless(live, fmin(1.0, fmax(-1.0, cs)));          // (v_min / v_max drop a NaN operand)
                el = fx::asin_deg(cs, atan_table().c);
                if (MAG && centre_coords) {
                    // relative to this lane's corner of the current row, like latitude and longitude
                    vec3 sc;
                    double sxyc;
                    if (pole_bin) pole_point(cn, cd, pc.x, pc.y, sc, sxyc); else sm_point(pc, sc, sxyc);
                    sm_angles(c.sv, c.sxy, c.sml, c.ssl, sc, sxyc, live, ml, slc);
                    mt = slc * (24.0 / 360.0) + 12.0;
                    if (kMagPole) {
                        vec3 rc;
                        double rxyc;
                        if (geo_pole) pole_point(cn, cd, pc.x, pc.y, rc, rxyc); else pole_point(sc.z, sxyc, sc.x, sc.y, rc, rxyc);
                        sm_angles(c.rv, c.rxy, c.rla, c.rlo, rc, rxyc, live, rlac, rloc);
                    }
                }
            } else {
                const double n = opaque_nan();
                lac = loc = el = ml = mt = slc = rlac = rloc = n;
            }
            if (st_pixel) {
                if (!kMagOnly && out_lat_c) at(out_lat_c, off_pixel) = lac;
                if (!kMagOnly && out_lon_c) at(out_lon_c, off_pixel) = loc;
                if (out_elev) at(out_elev, off_pixel) = el;
                if (MAG && !kPole && A.mlat_c) {
                    at(A.mlat_c, off_pixel) = ml;
                    at(A.mlt_c, off_pixel) = mt;
                }
            }
            const bool valid = px_ok && (el >= min_elev) && corners_ok;
            int bin_x = 0, bin_y = 0;            // 1-based bin indices of this pixel, 0 = not binned
            long long el_fix = 0;
            if (BIN && valid) {
                // reference resample.py:301-351 on (lon, lat) or, for resampleMLatMLT, on
                // (SM longitude = mltToSmLon(mlt), MLat) (mapping.py:1519-1547, transform.py:388-401)
                const bool bin_mag = kMagOnly || (MAG && !kPole && A.bin_magnetic);
                double bxv = bin_mag ? (mt - 12.0) / (24.0 / 360.0) : loc;
                double byv = bin_mag ? ml : lac;
                if (pole_bin) bxv = slc, byv = ml;
                if (kMagPole) bxv = rloc, byv = rlac;
                if (lon_wrap) bxv = wrap180_shifted(bxv);
                int bx, by;
                bool slow;
                bin_common(bxv, byv, bx, by, slow);
                unsigned int edge_flags = 0;
                if (__ballot(slow)) {          // wave-uniform and rare
                    if (slow) {
                        if (pole_bin || kMagPole) {
                            // next to an edge: the rotated coordinates as the two-pass plan computes them
                            const pole_consts pk = karg_load<pole_consts>(karg_fresh(koff), offsetof(georef_args, pole));
                            if (kMagPole && !geo_pole)
                                rotate_pole_deg(pk.w, pk.rot, pk.e2, ml, (mt - 12.0) / (24.0 / 360.0), pk.alt, byv, bxv);
                            else
                                rotate_pole_deg(pk.w, pk.rot, pk.e2, lac, loc, pk.alt, byv, bxv);
                        }
                        bin_slow(bxv, byv, bx, by, edge_flags);
                    }
                }
                if (bx > 0 && by > 0) {
                    el_fix = to_fix32(el);
                    bin_event* events = nullptr;
                    if (edge_flags) events = karg_load<bin_event*>(karg_fresh(koff), offsetof(georef_args, bin_events));
                    if (events != nullptr) {
                        // on an edge in the sense of the right-most-edge rule: which bin it belongs to depends on
                        // the final grid, so it is recorded instead of binned (see bin_event)
                        karg_ptr KE = karg_fresh(koff);
                        unsigned int* cnt = karg_load<unsigned int*>(KE, offsetof(georef_args, bin_event_count));
                        const long long cap = karg_load<long long>(KE, offsetof(georef_args, bin_event_cap));
                        const unsigned int slot = atomicAdd(cnt, 1u);
                        if ((long long)slot < cap) {
                            bin_event ev;
                            ev.bx = bx, ev.by = by, ev.flags = edge_flags;
                            ev.c0 = ch0, ev.c1 = ch1, ev.c2 = ch2;
                            ev.el = el_fix;
                            events[slot] = ev;
                        }
                    } else {
                        bin_x = bx;
                        bin_y = by;
                    }
                }
            }
            if (BIN) bin_account(bin_x, bin_y, ch0, ch1, ch2, el_fix);
            if (want_bbox) {
                // corner row gy-1 is final now: it keeps a corner when a centre above (prev.flag) or below
                // (this row: own pixel or the left neighbour's) is valid
                const int vi = valid ? 1 : 0;
                flag_cur = vi | from_prev_lane(vi);
                n_valid += vi;
                if (kMagBox) {
                    if ((prev.flag | flag_cur) && prev.bla == prev.bla) box_add(prev.bla, prev.blo);
                } else {
                    if ((prev.flag | flag_cur) && prev.la == prev.la) box_add(prev.la, prev.lo);
                }
            }
        }
        cur.p = c.p;
        cur.d = d;
        cur.la = c.la;
        cur.lo = c.lo;
        cur.bn = c.bn;
        cur.bd = c.bd;
        if (kMagBox) {
            cur.bla = c.bla;
            cur.blo = c.blo;
        }
        if (MAG) {
            cur.s = c.sv;
            cur.sxy = c.sxy;
            cur.ml = c.sml;
            cur.sl = c.ssl;
        }
        if (kMagPole) {
            cur.r = c.rv;
            cur.rxy = c.rxy;
            cur.rla = c.rla;
            cur.rlo = c.rlo;
        }
        cur.flag = flag_cur;
    };

    auto advance = [&](const int r, const bool even, const row_state& prev, row_state& cur) {
        vec3 dj = {dj0, dj1, dj2};
        if (DIRS_IN) {
            // the direction of this corner was loaded one row ahead; fetch the next row's now, so that its
            // latency is covered by this row's arithmetic
            asm volatile("" : "+v"(dj.x), "+v"(dj.y), "+v"(dj.z));
            if (col_ok && r < rows) {
                const double* q = A.dirs_in + 3 * ((int64_t)(y0 + r + 1) * W1 + gx);
                dj0 = q[0], dj1 = q[1], dj2 = q[2];
            }
        }
        step(r, even, dj, prev, cur);
        off_corner += pitch_corner;
        off_pixel += pitch_pixel;
        img_off += img_row;
    };
    if (BIN && px_ok && rows > 0) load_pixel(img_off + img_row, rawB0, rawB1, shiftB);       // pixel row y0: step 1
    for (int r = 0; r <= rows; r += 2) {
        advance(r, true, S0, S1);
        if (r + 1 <= rows) advance(r + 1, false, S1, S0); else S0 = S1;      // the last row's state ends up in S0
    }
    if (BIN && bin_anchor) {
        if (run_key) bin_flush(run_key, run_cnt, run_c0, run_c1, run_c2, run_el);
        // flush this wave's window: one 64-bit integer atomic per touched cell and plane
        __threadfence_block();
        karg_ptr K = karg_fresh(koff);
        const int nby = karg_load<axis_lin>(K, offsetof(georef_args, byl)).nbin;
        const int64_t ncell = (int64_t)karg_load<axis_lin>(K, offsetof(georef_args, bxl)).nbin * nby;
        unsigned long long* acc = karg_load<unsigned long long*>(K, offsetof(georef_args, bin_acc));
#pragma unroll
        for (int i = lane; i < kBinCells; i += 64) {
            const unsigned int cnt = sCnt[wave][i];
            if (cnt == 0) continue;
            const int dx = i / kBinW, dy = i - dx * kBinW;
            const int64_t cell = (int64_t)(bin_ax0 + dx - 1) * nby + (bin_ay0 + dy - 1);
            atomicAdd(&acc[cell], (unsigned long long)cnt);
            atomicAdd(&acc[ncell + cell], (unsigned long long)sCh[wave][0][i]);
            atomicAdd(&acc[2 * ncell + cell], (unsigned long long)sCh[wave][1][i]);
            atomicAdd(&acc[3 * ncell + cell], (unsigned long long)sCh[wave][2][i]);
            atomicAdd(&acc[4 * ncell + cell], sEl[wave][i]);
        }
    }
    if (want_bbox) {
        // the chunk's last corner row only has centres above it inside this chunk
        if (kMagBox) {
            if (S0.flag && S0.bla == S0.bla) box_add(S0.bla, S0.blo);
        } else {
            if (S0.flag && S0.la == S0.la) box_add(S0.la, S0.lo);
        }
        double v[7];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            v[k] = __hip_atomic_load(&sBox[wave][k][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v[6] = (double)n_valid;
        for (int o = 32; o > 0; o >>= 1) {
            v[0] = fmin(v[0], __shfl_xor(v[0], o));
            v[1] = fmax(v[1], __shfl_xor(v[1], o));
            v[2] = fmin(v[2], __shfl_xor(v[2], o));
            v[3] = fmax(v[3], __shfl_xor(v[3], o));
            v[4] = fmin(v[4], __shfl_xor(v[4], o));
            v[5] = fmax(v[5], __shfl_xor(v[5], o));
            v[6] += __shfl_xor(v[6], o);
        }
        if (lane < 8) A.bbox_partials[(int64_t)item * 8 + lane] = lane == 0 ? v[0] : lane == 1 ? v[1] : lane == 2 ? v[2]
                                                                  : lane == 3 ? v[3] : lane == 4 ? v[4]
                                                                  : lane == 5 ? v[5] : lane == 6 ? v[6] : 0.0;
    }
}

// Folds bbox partials ([n][8]) into gridDim.x rows of out ([gridDim.x][8]); launched twice (n -> 64 -> 1).
// extra (optional, final stage only): a device counter whose value goes out in slot 7 (the number of on-edge
// pixels recorded by the fused binning), so that the host gets it with the same 64 bytes as the box
__global__ __launch_bounds__(kThreads) void k_bbox_fold(const double* __restrict__ partials, int n,
                                                         double* __restrict__ out,
                                                         const unsigned int* __restrict__ extra) {
    __shared__ double sRed[8][kThreads / 64];
    double v[8] = {kInf, -kInf, kInf, -kInf, kInf, -kInf, 0, 0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
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
    if (extra != nullptr) v[7] = threadIdx.x == 0 ? (double)*extra : 0.0;     // slot 7 is summed over the block
    block_reduce8<kThreads>(v, out + (int64_t)blockIdx.x * 8, sRed);
}

constexpr int kTW = 64, kTH = 8;
constexpr int kFoldBlocks = 64;

// The folds of up to kMaxBatch frames in one launch per stage (grid: blocks x frames; two launches behind a launch of three
// frames where there were six).  stage 0: every block folds its share of a frame's partials into one row of the frame's
// scratch; stage 1 (one block per frame): the rows into the frame's box.  (Both stages in ONE launch — the block that
// arrives last at a counter folds the rows — was measured and is slower for the whole pipeline by 15 %: every block needs
// a device-scope release before the counter, which on this chip writes back the L2 of its XCD, 768 times per launch of
// three frames, while the next big kernel is filling those L2s with its stores; profiles/r3/x_ab_fold_variants.txt.)
struct fold_batch {
    const double* in[kMaxBatch];
    double* out[kMaxBatch];
    const unsigned int* extra[kMaxBatch];   // see k_bbox_fold
};

__global__ __launch_bounds__(kThreads) void k_bbox_fold_batch(fold_batch Bf, int n) {
    __shared__ double sRed[8][kThreads / 64];
    const int f = blockIdx.y;
    const double* partials = Bf.in[f];
    double v[8] = {kInf, -kInf, kInf, -kInf, kInf, -kInf, 0, 0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
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
    if (Bf.extra[f] != nullptr) v[7] = threadIdx.x == 0 ? (double)*Bf.extra[f] : 0.0;     // slot 7 is summed over the block
    block_reduce8<kThreads>(v, Bf.out[f] + (int64_t)blockIdx.x * 8, sRed);
}

// start / stop: events attached to the dispatch (timing, and the hand-over to the driver's tail stream) or NULL
struct launch_events {
    hipEvent_t start, stop;
};

template <bool FAST, bool DIRS_IN>
void launch_variant(amt_ctx* ctx, const georef_args& A, dim3 grid, bool mag, launch_events ev) {
    const dim3 block(kThreads);
    if (mag)
        hipExtLaunchKernelGGL((k_georef<kTW, kTH, FAST, DIRS_IN, true>), grid, block, 0, ctx->stream, ev.start, ev.stop, 0, A);
    else
        hipExtLaunchKernelGGL((k_georef<kTW, kTH, FAST, DIRS_IN, false>), grid, block, 0, ctx->stream, ev.start, ev.stop, 0, A);
}

// second: 0 ... 4 as the kernel's SECOND
template <bool FAST, bool DIRS_IN, int BIN>
void launch_rows_bin(amt_ctx* ctx, const georef_batch& B, int n_frames, dim3 grid, int second, int rows, int strips_x,
                     int n_items, launch_events ev) {
    const dim3 block(kRowsThreads);
    if (second == 4) {
        if constexpr (!DIRS_IN)
            hipExtLaunchKernelGGL((k_georef_rows<FAST, DIRS_IN, 4, BIN>), grid, block, 0, ctx->stream, ev.start, ev.stop, 0, B,
                                  rows, strips_x, n_items, n_frames);
    } else if (second == 3) {
        if constexpr (BIN != 0 && !DIRS_IN)
            hipExtLaunchKernelGGL((k_georef_rows<FAST, DIRS_IN, 3, BIN>), grid, block, 0, ctx->stream, ev.start, ev.stop, 0, B,
                                  rows, strips_x, n_items, n_frames);
    } else if (second == 2) {
        if constexpr (BIN != 0 && !DIRS_IN)
            hipExtLaunchKernelGGL((k_georef_rows<FAST, DIRS_IN, 2, BIN>), grid, block, 0, ctx->stream, ev.start, ev.stop, 0, B,
                                  rows, strips_x, n_items, n_frames);
    } else if (second == 1) {
        hipExtLaunchKernelGGL((k_georef_rows<FAST, DIRS_IN, 1, BIN>), grid, block, 0, ctx->stream, ev.start, ev.stop, 0, B,
                              rows, strips_x, n_items, n_frames);
    } else {
        hipExtLaunchKernelGGL((k_georef_rows<FAST, DIRS_IN, 0, BIN>), grid, block, 0, ctx->stream, ev.start, ev.stop, 0, B,
                              rows, strips_x, n_items, n_frames);
    }
}

template <bool FAST, bool DIRS_IN>
void launch_rows(amt_ctx* ctx, const georef_batch& B, int n_frames, dim3 grid, int second, int bin, int rows, int strips_x,
                 int n_items, launch_events ev) {
    if (bin == 1)
        launch_rows_bin<FAST, DIRS_IN, 1>(ctx, B, n_frames, grid, second, rows, strips_x, n_items, ev);
    else if (bin == 2)
        launch_rows_bin<FAST, DIRS_IN, 2>(ctx, B, n_frames, grid, second, rows, strips_x, n_items, ev);
    else
        launch_rows_bin<FAST, DIRS_IN, 0>(ctx, B, n_frames, grid, second, rows, strips_x, n_items, ev);
}

// One thread per lattice corner (every `stride`-th pixel corner): bounding box of the corners whose own ray
// has an elevation >= min_elev, in (lat, lon), (MLat, SM longitude) [magnetic = 1] or (lat, lon) rotated by 90 deg
// about x [magnetic = 2, the pole plan; 3: (MLat, SM longitude) rotated likewise].  Partials per workgroup.
// DIRS: the corner's direction is read from A.dirs_in (caller-supplied unit vectors, J2000) instead of the TAN model.
template <bool DIRS>
__global__ __launch_bounds__(kThreads) void k_coarse_bbox(georef_args A, int stride, int magnetic, double min_elev,
                                                           double* __restrict__ partials) {
    __shared__ double sRed[8][kThreads / 64];
    const int nxl = (A.width + stride - 1) / stride + 1, nyl = (A.height + stride - 1) / stride + 1;
    double v[8] = {kInf, -kInf, kInf, -kInf, kInf, -kInf, 0, 0};
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < nxl * nyl) {
        const int iy = i / nxl, ix = i - iy * nxl;
        const int gx = min(ix * stride, A.width), gy = min(iy * stride, A.height);
        vec3 d;
        if (DIRS) {
            const double* q = A.dirs_in + 3 * ((int64_t)gy * (A.width + 1) + gx);
            d.x = q[0], d.y = q[1], d.z = q[2];
        } else {
            d = tan_direction_fast(A.wcs, gx - 0.5, gy - 0.5);
        }
        const double t = ray_param_fast(A.ray, d);
        if (t == t) {
            // where the expensive rays are, relative to the frame centre (scheduling hint for the full kernel)
            v[7] = (double)(((2 * gx > A.width) - (2 * gx < A.width)) * (1 << 20) + ((2 * gy > A.height) - (2 * gy < A.height)));
            const vec3 p = ray_point(A.ray, d, t);
            double c = -(d.x * p.x + d.y * p.y + d.z * p.z) * fm::rsqrt(dot(p, p));
            c = fmin(1.0, fmax(-1.0, c));
            if (fm::asin_deg(c) >= min_elev) {
                double la, lo;
                if (magnetic == 1 || magnetic == 3) {
                    double mt;
                    sm_to_mlat_mlt_fast(mul(A.m_sm, p), la, mt);
                    lo = (mt - 12.0) / (24.0 / 360.0);
                    if (magnetic == 3) rotate_pole_deg(A.pole.w, A.pole.rot, A.pole.e2, la, lo, A.pole.alt, la, lo);
                } else {
                    const vec3 g = mul(A.m_geo, p);
                    ecef_to_geodetic_deg_fast(A.bw, g.x, g.y, g.z, la, lo);
                    if (magnetic == 2) rotate_pole_deg(A.pole.w, A.pole.rot, A.pole.e2, la, lo, A.pole.alt, la, lo);
                }
                v[0] = v[1] = la;
                v[2] = v[3] = lo;
                if (lo > 0) v[4] = lo; else v[5] = lo;
                v[6] = 1;
            }
        }
    }
    block_reduce8<kThreads>(v, partials + (int64_t)blockIdx.x * 8, sRed);
}

// amt_georef_out.item_order = 0: towards which side of the frame the nadir (the direction of the Earth's centre)
// lies.  In-plane direction of the TAN projection of -cam: native vector v = rot^T d, intermediate coordinates
// proportional to (v_y, -v_x) (no division by v_z, so it also works when the nadir is behind the image plane),
// pixel offsets by the inverse CD matrix.
int nadir_side(const amt_frame_params* p) {
    const double d[3] = {-p->cam[0], -p->cam[1], -p->cam[2]};
    const double* r = p->rot;
    const double vx = r[0] * d[0] + r[3] * d[1] + r[6] * d[2], vy = r[1] * d[0] + r[4] * d[1] + r[7] * d[2];
    const double bx = vy, by = -vx;
    const double det = p->cd[0] * p->cd[3] - p->cd[1] * p->cd[2];
    if (!(det != 0)) return 1;
    const double px = (bx * p->cd[3] - p->cd[1] * by) / det, py = (p->cd[0] * by - p->cd[2] * bx) / det;
    // Earth to the left or right: every row of items mixes cheap and expensive ones, any row order does
    return (std::fabs(py) >= std::fabs(px) && py > 0) ? 2 : 1;
}

int forced_order_env() {
    static const int forced_order = [] {
        const char* e = std::getenv("AMT_ITEM_ORDER");
        return e ? std::atoi(e) : 0;
    }();
    return forced_order;
}

// items (waves) of the row-marching launch / tiles of the tile kernel, and the rows per chunk
struct launch_shape {
    bool use_tiles;
    int rows_per_chunk, tiles_x, tiles_y, strips_x, chunks_y;
    int64_t n_items;
};

launch_shape shape_of(const amt_frame_params* p) {
    // kernel selection: row-marching waves (default) or LDS tiles (AMT_GEOREF_KERNEL=tile), for A/B runs
    static const bool use_tiles = [] {
        const char* e = std::getenv("AMT_GEOREF_KERNEL");
        return e != nullptr && std::strcmp(e, "tile") == 0;
    }();
    static const int rows_per_chunk = [] {
        const char* e = std::getenv("AMT_GEOREF_ROWS");
        const int v = e ? std::atoi(e) : 0;
        return v > 0 ? v : 16;
    }();
    launch_shape s;
    s.use_tiles = use_tiles;
    s.rows_per_chunk = rows_per_chunk;
    s.tiles_x = (p->width + kTW - 1) / kTW;
    s.tiles_y = (p->height + kTH - 1) / kTH;
    s.strips_x = (p->width + 1 + 62) / 63;                       // 64 corner columns, 63 owned, per strip
    s.chunks_y = (p->height + rows_per_chunk - 1) / rows_per_chunk;
    s.n_items = use_tiles ? (int64_t)s.tiles_x * s.tiles_y : (int64_t)s.strips_x * s.chunks_y;
    return s;
}

// Stride s of the interleaved chunk order (item_order 3): rows of work items are visited in the order (k * s) mod n,
// s coprime to n and close to n / golden ratio, so that any stretch of the launch samples the whole frame evenly.
int interleave_stride(int n) {
    if (n < 3) return 1;
    auto gcd = [](int a, int b) {
        while (b) {
            const int t = a % b;
            a = b;
            b = t;
        }
        return a;
    };
    int s = (int)(n * 0.6180339887498949);
    if (s < 1) s = 1;
    while (gcd(s, n) != 1) ++s;
    return s % n ? s % n : 1;
}

// Host copy of the kernel's hit test (shell_t >= 0) for the pixel-corner position (column x, row y).
bool ray_hits_host(const affine_cam& c, const shell_ray& e, double x, double y) {
    const double px = x + c.cx, py = y + c.cy;
    const double ux = c.u0[0] + px * c.ux[0] + py * c.uy[0], uy = c.u0[1] + px * c.ux[1] + py * c.uy[1],
                 uz = c.u0[2] + px * c.ux[2] + py * c.uy[2];
    const double uu = ux * ux + uy * uy + uz * uz;
    const double ku = ux * e.kx + uy * e.ky + uz * e.kz, uo = ux * e.ox + uy * e.oy + uz * e.oz;
    const double a2 = e.qd * ku * ku + e.qa * uu, nb = -(e.qa * uo + e.qd_ko * ku);
    const double disc = nb * nb - a2 * e.c0;
    if (!(disc >= 0)) return false;
    return (nb + e.root_sign * std::sqrt(disc)) / a2 >= 0;
}

// Which rows of work items cannot see the shell?  A corner (x, y) hits when disc(x, y) >= 0 and the root is in front of
// the camera, which for a camera outside the shell needs nb(x, y) >= 0 (shell_t: t = (nb - sqrt(disc)) / a2).  With the
// affine camera model nb is LINEAR and disc QUADRATIC in (x, y) — the limb is a conic section in the image and the set of
// hits is convex —, so both are bounded exactly over a line: a horizontal band of the frame is free of hits when its two
// long sides and its two short sides are (a convex set that meets the band without touching its border would have to
// lie inside it: excluded below by the apparent size of the Earth).  disc is sampled at three columns, each a quadratic
// in y fitted through three rows; per line of corners that costs three evaluations.  The bands are taken a pixel wider
// than the rows of an item (exact centre rays lie between the corners) and a line only counts as free of hits when
// max disc < -1e-9 of its scale or max nb < 0: anything nearer to the limb than that goes the ordinary way.
// The machinery: bands of rows_per_chunk rows (n of them) of a W x H frame in which a convex region {quadratic >= 0, linear >= 0}
// of the image plane has no point; eval(x, y, &linear) -> quadratic.  [0, *top_end) and [*bottom_begin, n) are free of it.
template <class Eval>
void conic_bands(double W, double H, int n, int rows_per_chunk, Eval eval, int* top_end, int* bottom_begin) {
    *top_end = 0, *bottom_begin = n;
    // quadratic through (t0, f0), (t1, f1), (t2, f2) with t1 the midpoint: coefficients of f(t0 + s), s in [0, L]
    struct quad {
        double a, b, c, L;
        double at(double s) const { return (a * s + b) * s + c; }
        double max_on(double s0, double s1) const {
            double m = std::max(at(s0), at(s1));
            if (a < 0) {
                const double v = -b / (2 * a);
                if (v > s0 && v < s1) m = std::max(m, at(v));
            }
            return m;
        }
    };
    auto fit = [](double f0, double f1, double f2, double L) {
        quad q;
        q.L = L;
        q.c = f0;
        q.a = 2 * (f2 - 2 * f1 + f0) / (L * L);
        q.b = (f2 - f0) / L - q.a * L;
        return q;
    };
    const double x0 = -1.0, xL = W + 2.0, y0 = -1.0, yL = H + 2.0;        // one pixel beyond the corners on every side
    // disc along three columns (as quadratics in y) and nb as a plane
    double nb00, nb10, nb01, tmp;
    quad col[3];
    double scale = 0;
    for (int k = 0; k < 3; ++k) {
        const double x = x0 + 0.5 * k * xL;
        double f[3];
        for (int j = 0; j < 3; ++j) {
            f[j] = eval(x, y0 + 0.5 * j * yL, &tmp);
            scale = std::max(scale, std::max(std::fabs(f[j]), tmp * tmp));
        }
        col[k] = fit(f[0], f[1], f[2], yL);
    }
    eval(x0, y0, &nb00);
    eval(x0 + xL, y0, &nb10);
    eval(x0, y0 + yL, &nb01);
    const double nbx = (nb10 - nb00) / xL, nby = (nb01 - nb00) / yL;
    const double tol = 1e-9 * scale, nb_tol = 1e-9 * std::sqrt(scale);
    if (!(scale > 0) || !std::isfinite(scale)) return;
    // a segment [s0, s1] of a line on which disc is the quadratic q and nb = n0 + slope * s: free of hits when disc < 0
    // wherever nb >= 0 (nb is linear: that part of the segment is an interval)
    auto segment_free = [&](const quad& q, double n0, double slope, double s0, double s1) {
        double a = s0, b = s1;                            // the part with nb >= -nb_tol
        const double na = n0 + slope * s0 + nb_tol, nbv = n0 + slope * s1 + nb_tol;
        if (na < 0 && nbv < 0) return true;
        if (na < 0) a = s0 + (s1 - s0) * (-na) / (nbv - na);
        if (nbv < 0) b = s0 + (s1 - s0) * na / (na - nbv);
        return q.max_on(a, b) < -tol;
    };
    // a horizontal line at height s (above y0): free of hits?
    auto line_free = [&](double s) {
        const quad q = fit(col[0].at(s), col[1].at(s), col[2].at(s), xL);
        return segment_free(q, nb00 + nby * s, nbx, 0.0, xL);
    };
    // a piece [s0, s1] of the left (k = 0) or right (k = 2) side: free of hits?
    auto side_free = [&](int k, double s0, double s1) {
        return segment_free(col[k], nb00 + (k == 2 ? nbx * xL : 0.0), nby, s0, s1);
    };
    auto band_free = [&](int c) {
        // corner rows c R ... min((c + 1) R, H), a pixel more on both sides; s counts from y0 = -1
        const double s0 = (double)c * rows_per_chunk, s1 = std::min((double)(c + 1) * rows_per_chunk, H) + 2.0;
        return line_free(s0) && line_free(s1) && side_free(0, s0, s1) && side_free(2, s0, s1);
    };
    int t = 0;
    while (t < n && band_free(t)) ++t;
    *top_end = t;
    if (t == n) return;                                   // nothing of the region in the frame: everything is in the first range
    int b = n;
    while (b > t && band_free(b - 1)) --b;
    *bottom_begin = b;
}

// angle one pixel spans (radians, the larger of the two axes) in the affine camera model
double pixel_angle(const affine_cam& c) {
    return std::sqrt(std::max(c.uy[0] * c.uy[0] + c.uy[1] * c.uy[1] + c.uy[2] * c.uy[2],
                              c.ux[0] * c.ux[0] + c.ux[1] * c.ux[1] + c.ux[2] * c.ux[2])) / kRad2Deg;
}

void sky_bands(const georef_args& A, const launch_shape& sh, int* top_end, int* bottom_begin) {
    const int n = sh.chunks_y;
    *top_end = 0, *bottom_begin = n;
    const shell_ray& e = A.sray;
    if (!(e.root_sign < 0)) return;                       // camera inside the shell: every ray hits
    auto eval = [&](double x, double y, double* nb_out) {
        const affine_cam& c = A.cam;
        const double px = x + c.cx, py = y + c.cy;
        const double ux = c.u0[0] + px * c.ux[0] + py * c.uy[0], uy = c.u0[1] + px * c.ux[1] + py * c.uy[1],
                     uz = c.u0[2] + px * c.ux[2] + py * c.uy[2];
        const double uu = ux * ux + uy * uy + uz * uz;
        const double ku = ux * e.kx + uy * e.ky + uz * e.kz, uo = ux * e.ox + uy * e.oy + uz * e.oz;
        const double a2 = e.qd * ku * ku + e.qa * uu, nb = -(e.qa * uo + e.qd_ko * ku);
        *nb_out = nb;
        return nb * nb - a2 * e.c0;
    };
    // the Earth must be far larger in the image than a band is tall (see above): apparent radius of the shell against
    // the angle a band spans
    {
        const double oo = std::sqrt(e.ox * e.ox + e.oy * e.oy + e.oz * e.oz);
        const double sin_rho = std::min(1.0, 1.0 / (std::sqrt(e.qa) * oo));             // ~ a / |camera|
        if (!(std::asin(sin_rho) > 8.0 * (sh.rows_per_chunk + 2) * pixel_angle(A.cam))) return;
    }
    conic_bands((double)A.width, (double)A.height, n, sh.rows_per_chunk, eval, top_end, bottom_begin);
}

// Which rows of work items can hold a pixel that survives maskedByElevation(min_elev)?  (The rows of a host image that have to
// cross the link: the fused binning reads a pixel's colours only to bin it, and only pixels with elevation >= min_elev are
// binned, reference mapping.py:845-864, resample.py:119-120,315-321.)  The elevation is measured against the geocentric radial
// of the hit point P (astrometry.py:200-212): in the triangle (Earth's centre, camera C, P) the law of sines gives
//     sin(nadir angle of the ray) = |P| / |C| * cos(elevation),
// and |P| <= the shell's larger semi-axis, so every pixel with elevation >= e0 > 0 lies inside the cone of half-angle
// asin(a_max / |C| * cos e0) about the nadir — in the image of the affine camera model a conic section again, bounded per band
// of rows exactly like the limb.  Conservative by construction (|P| bounded from above, a band a pixel wider than its rows and one
// more band on either side); outside [*top_end, *bottom_begin) no pixel is binned.
void elevation_bands(const georef_args& A, const launch_shape& sh, double min_elev_deg, int* top_end, int* bottom_begin) {
    const int n = sh.chunks_y;
    *top_end = 0, *bottom_begin = n;
    const shell_ray& e = A.sray;
    if (!(e.root_sign < 0) || !(min_elev_deg > 0) || !(min_elev_deg < 90)) return;
    const double oo = e.ox * e.ox + e.oy * e.oy + e.oz * e.oz;
    const double r_max = 1.0 / std::sqrt(std::min(e.qa, e.qa + e.qd));                // max(a, b) of the shell
    const double k = r_max / std::sqrt(oo) * std::cos(min_elev_deg * kDeg2Rad);      // sin of the cone's half-angle
    if (!(k < 1.0)) return;
    const double cos2 = 1.0 - k * k;
    if (!(std::asin(k) > 8.0 * (sh.rows_per_chunk + 2) * pixel_angle(A.cam))) return;
    auto eval = [&](double x, double y, double* lin_out) {
        const affine_cam& c = A.cam;
        const double px = x + c.cx, py = y + c.cy;
        const double ux = c.u0[0] + px * c.ux[0] + py * c.uy[0], uy = c.u0[1] + px * c.ux[1] + py * c.uy[1],
                     uz = c.u0[2] + px * c.ux[2] + py * c.uy[2];
        const double uu = ux * ux + uy * uy + uz * uz;
        const double uo = ux * e.ox + uy * e.oy + uz * e.oz;
        *lin_out = -uo;                                   // > 0: the ray points to the Earth's side
        return uo * uo - cos2 * oo * uu;                  // >= 0: inside the cone (or its mirror image, which `lin` excludes)
    };
    int t = 0, b = n;
    conic_bands((double)A.width, (double)A.height, n, sh.rows_per_chunk, eval, &t, &b);
    // one more band on either side: a fast centre is the mean of four corner hits, not a ray of its own
    *top_end = std::max(0, t - 1), *bottom_begin = std::min(n, b + 1);
    if (t == n) *top_end = n, *bottom_begin = n;          // nothing above the threshold in the frame
}

// item_order 4: where the limb cuts the frame's rows of work items.  The middle row of every chunk is probed in three
// columns; when the chunks that see the Earth are one run that reaches the first or the last row of chunks and the sky
// takes at least an eighth of the frame, the launch works on two fronts from the limb (see two_front_chunk), the
// Earth's rows and the sky's in the proportion that lets both fronts finish together.  Returns false when the frame is
// not of that kind (all Earth, all sky, Earth to the left or right, a whole disc in view): the caller keeps its order.
bool two_front_plan(const georef_args& A, const launch_shape& sh, int* split, int* n_e, int* n_s, int* flip) {
    const int n = sh.chunks_y;
    if (n < 16) return false;
    int first = n, last = -1;
    for (int c = 0; c < n; ++c) {
        const double y = std::min((double)A.height, (c + 0.5) * sh.rows_per_chunk);
        int hits = 0;
        for (int k = 1; k <= 3; ++k) hits += ray_hits_host(A.cam, A.sray, 0.25 * k * A.width, y) ? 1 : 0;
        if (hits >= 2) {
            first = std::min(first, c);
            last = c;
        }
    }
    if (last < 0) return false;
    // (the conic section that bounds the hits is convex: per column they are one interval)
    const int earth = last - first + 1, sky = n - earth;
    if (sky < n / 8 || earth < n / 8) return false;
    if (last == n - 1) {
        *flip = 0, *split = first;
    } else if (first == 0) {
        *flip = 1, *split = n - 1 - last;       // in the mirrored frame the Earth's chunks are [split, n)
    } else {
        return false;
    }
    // an Earth row of items costs about 1.6 times a sky row (all-Earth frame 150 us, frame of sky 96 us); small whole
    // numbers whose ratio is close to (earth : sky) chunks weighted that way
    const double want = (double)earth / sky;        // Earth rows per sky row so that both fronts end together
    int best_e = 1, best_s = 1;
    double best = 1e9;
    for (int e = 1; e <= 4; ++e)
        for (int s2 = 1; s2 <= 4; ++s2) {
            const double d = std::fabs(std::log(((double)e / s2) / want));
            if (d < best) best = d, best_e = e, best_s = s2;
        }
    *n_e = best_e, *n_s = best_s;
    return true;
}

// One frame of a launch, validated and with its kernel arguments assembled
struct prepared_frame {
    georef_args A;
    launch_shape sh;
    const amt_frame_params* p;
    const double* dirs;
    const amt_georef_out* out;
    const amt_georef_tail* tail;
    double* fold;       // scratch of the first fold stage (behind the partials)
    bool mag;
    int second;         // k_georef_rows' SECOND
    int bin;
};

int prepare_georef(amt_ctx* ctx, const amt_frame_params* p, const double* dirs, const amt_georef_out* out,
                   const amt_georef_tail* tail, prepared_frame* F) {
    AMT_REQUIRE(ctx, p && out, "NULL argument");
    AMT_REQUIRE(ctx, p->width > 0 && p->height > 0, "empty frame");
    AMT_REQUIRE(ctx, p->a > 0 && p->b > 0 && p->a0 > 0 && p->b0 > 0, "ellipsoid axes must be positive");
    AMT_REQUIRE(ctx, (out->mlat == nullptr) == (out->mlt == nullptr), "mlat and mlt must be given together");
    AMT_REQUIRE(ctx, (out->mlat_c == nullptr) == (out->mlt_c == nullptr), "mlat_c and mlt_c must be given together");
    AMT_REQUIRE(ctx, dirs == nullptr || p->fast_center, "caller-supplied directions need fast_center");
    georef_args& A = F->A;
    A.wcs = make_tan_wcs(p);
    A.ray = make_ray(p->a, p->b, p->cam, 1);
    A.m_geo = make_mat3(p->m_geo);
    A.m_sm = make_mat3(p->m_sm);
    {
        // GEO-frame constants of k_georef_rows: camera model and SM rotation composed with M = J2000 -> GEO
        double rot_geo[9], mt[9], geo_sm[9];
        mat_mul3(p->m_geo, p->rot, rot_geo);
        tan_wcs wcs_geo = A.wcs;
        wcs_geo.rot = make_mat3(rot_geo);
        A.cam = make_affine_cam(wcs_geo);
        A.sray = make_shell_ray(p->a, p->b, p->cam, p->m_geo);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) mt[3 * i + j] = p->m_geo[3 * j + i];
        mat_mul3(p->m_sm, mt, geo_sm);               // M_sm M^T: GEO -> SM
        A.m_geo_sm = make_mat3(geo_sm);
    }
    A.bw = make_bowring_fast(p->a0, p->b0);
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
    int bin = 0;
    A.bin_img = nullptr;
    A.bin_acc = nullptr;
    A.bin_lon_wrap = A.bin_magnetic = A.bin_pole = 0;
    A.row_layout = out->row_layout == AMT_ROWS_STRIP_PADDED ? 1 : 0;
    A.pole = make_pole_consts(p->a0, p->b0, 0.0);
    A.bin_events = nullptr;
    A.bin_event_count = nullptr;
    A.bin_event_cap = 0;
    // AMT_ITEM_ORDER = 1, 2, 3, 4 overrides the order for A/B runs
    const int forced_order = forced_order_env();
    A.item_order = out->item_order >= 1 && out->item_order <= 3 ? out->item_order : (dirs ? 1 : nadir_side(p));
    if (forced_order >= 1 && forced_order <= 4) A.item_order = forced_order;
    A.chunk_stride = 1;
    A.front_split = 0, A.front_e = A.front_s = 1, A.front_flip = 0;
    std::memset(&A.bax, 0, sizeof(A.bax));
    std::memset(&A.bay, 0, sizeof(A.bay));
    std::memset(&A.bxl, 0, sizeof(A.bxl));
    std::memset(&A.byl, 0, sizeof(A.byl));
    if (out->bin_acc != nullptr) {
        AMT_REQUIRE(ctx, out->bin_img && (out->bin_img_dtype == 1 || out->bin_img_dtype == 2),
                    "fused binning needs a uint8 (1) or uint16 (2) RGB image");
        AMT_REQUIRE(ctx, (int64_t)p->width * p->height >= 3, "fused binning needs at least 3 pixels");
        AMT_REQUIRE(ctx, (int64_t)p->width * p->height * 6 < (1ll << 32), "fused binning: image larger than 4 GiB");
        AMT_REQUIRE(ctx, axis_ok(out->bin_xaxis) && axis_ok(out->bin_yaxis) && out->bin_xaxis->uniform &&
                             out->bin_yaxis->uniform, "fused binning needs two uniform axes");
        AMT_REQUIRE(ctx, out->bin_xaxis->nbin < 65535 && out->bin_yaxis->nbin < 65535, "at most 65534 bins per axis");
        make_axis(out->bin_xaxis, &A.bax);
        make_axis(out->bin_yaxis, &A.bay);
        A.bxl = make_axis_lin(A.bax);
        A.byl = make_axis_lin(A.bay);
        A.bin_img = out->bin_img;
        A.bin_acc = reinterpret_cast<unsigned long long*>(out->bin_acc);
        A.bin_lon_wrap = out->bin_lon_wrap ? 1 : 0;
        A.bin_magnetic = out->bin_magnetic ? 1 : 0;
        if (out->bin_pole) {
            AMT_REQUIRE(ctx, !A.bin_lon_wrap, "bin_pole excludes bin_lon_wrap");
            AMT_REQUIRE(ctx, dirs == nullptr, "bin_pole needs the camera model");
            AMT_REQUIRE(ctx, std::fabs((p->a - p->a0) - out->altitude) < 1e-6 && std::fabs((p->b - p->b0) - out->altitude) < 1e-6,
                        "bin_pole: amt_georef_out.altitude does not match the shell of the frame");
            A.bin_pole = 1;
            A.pole = make_pole_consts(p->a0, p->b0, out->altitude);
        }
        if (out->bin_events != nullptr) {
            AMT_REQUIRE(ctx, out->bin_event_count != nullptr && out->bin_event_capacity > 0, "bin_events needs a counter and a capacity");
            A.bin_events = static_cast<bin_event*>(out->bin_events);
            A.bin_event_count = out->bin_event_count;
            A.bin_event_cap = out->bin_event_capacity;
        }
        bin = out->bin_img_dtype;
    } else if (out->bin_magnetic) {
        // without fused binning: the bounding box alone in (MLat, SM longitude) — the MLat / MLT-only variant without an image
        // (the box pass of resampleMLatMLT(arcsecPerPx=...), amt_pipe_launch_box)
        AMT_REQUIRE(ctx, dirs == nullptr && !out->lat && !out->lon && !out->lat_c && !out->lon_c,
                    "bin_magnetic without bin_acc (the box in MLat / SM longitude) excludes the geodetic outputs and caller-supplied directions");
        A.bin_magnetic = 1;
    }
    const launch_shape sh = shape_of(p);
    A.chunk_stride = interleave_stride(sh.chunks_y);
    A.sky_top_end = 0, A.sky_bottom_begin = sh.chunks_y;
    static const bool no_sky_path = std::getenv("AMT_NO_SKY_PATH") != nullptr;      // A/B runs
    if (dirs == nullptr && !sh.use_tiles && !no_sky_path) sky_bands(A, sh, &A.sky_top_end, &A.sky_bottom_begin);
    // two fronts from the limb (item_order 4; AMT_ITEM_ORDER=4, AMT_FRONT_RATIO = "e:s" overrides the proportion of the two
    // fronts).  Measured (profiles/r3/x_ab_two_front_order.txt): 135 / 168 us (georef only / fused, kernel alone) against
    // 126 / 153 us for the side-first order — sky rows (stores only) and Earth rows (VALU-bound) in flight TOGETHER are
    // slower than one after the other, at every proportion; an option for A/B runs, not the default.
    if (dirs == nullptr && !sh.use_tiles && (forced_order_env() == 4 || out->item_order == 4)) {
        int split = 0, fe = 1, fs = 1, flip = 0;
        if (two_front_plan(A, sh, &split, &fe, &fs, &flip)) {
            A.item_order = 4;
            A.front_split = split, A.front_e = fe, A.front_s = fs, A.front_flip = flip;
            static const char* ratio = std::getenv("AMT_FRONT_RATIO");
            if (ratio != nullptr) {
                int e2 = 0, s2 = 0;
                if (std::sscanf(ratio, "%d:%d", &e2, &s2) == 2 && e2 >= 1 && s2 >= 1) A.front_e = e2, A.front_s = s2;
            }
        } else if (A.item_order == 4) {
            A.item_order = nadir_side(p);
        }
    }
    const bool use_tiles = sh.use_tiles;
    const int64_t n_items = sh.n_items;
    AMT_REQUIRE(ctx, n_items < (1ll << 31), "frame too large");
    // the row-marching kernel addresses its arrays with 32-bit byte offsets
    AMT_REQUIRE(ctx, use_tiles || ((int64_t)p->width + 1) * ((int64_t)p->height + 1) * 8 < (1ll << 32),
                "frame too large (more than 2^29 pixel corners)");
    A.bbox_partials = nullptr;
    double* fold = nullptr;
    if (out->bbox) {
        const size_t need = (size_t)(n_items + kFoldBlocks) * 8 * sizeof(double);
        if (tail) {
            AMT_REQUIRE(ctx, tail->partials != nullptr && tail->partials_bytes >= need, "partials buffer too small");
            A.bbox_partials = tail->partials;
        } else {
            A.bbox_partials = static_cast<double*>(amt_workspace(ctx, need));
        }
        if (A.bbox_partials == nullptr) {
            ctx->last_error = "amt_georef_frame: workspace allocation failed";
            return AMT_ENOMEM;
        }
        fold = A.bbox_partials + n_items * 8;
    }
    AMT_REQUIRE(ctx, !(bin && use_tiles), "fused binning is implemented by the row-marching kernel only");
    AMT_REQUIRE(ctx, out->row_layout == AMT_ROWS_CONTIGUOUS || out->row_layout == AMT_ROWS_STRIP_PADDED, "unknown row_layout");
    AMT_REQUIRE(ctx, !(A.row_layout && use_tiles), "strip-padded rows are written by the row-marching kernel only");
    AMT_REQUIRE(ctx, !A.row_layout || ((int64_t)p->height + 1) * sh.strips_x * 512 < (1ll << 32),
                "frame too large for strip-padded rows (32-bit byte offsets)");
    F->sh = sh;
    F->p = p;
    F->dirs = dirs;
    F->out = out;
    F->tail = tail;
    F->fold = fold;
    F->mag = out->mlat != nullptr || out->mlat_c != nullptr || A.bin_magnetic;
    AMT_REQUIRE(ctx, !(A.bin_pole && use_tiles), "bin_pole is implemented by the row-marching kernel only");
    // (bin_pole on a geodetic grid with MLat / MLT outputs: the SECOND = 3 variant with the rotated pair taken from (lat, lon))
    F->second = A.bin_pole ? (F->mag ? 3 : 2) : (F->mag ? 1 : 0);
    // MLat / MLT only (SECOND = 4): a (MLat, SM longitude) grid without a pole plan whose caller wants none of the four
    // geodetic arrays — what resampleMLatMLT consumes (reference mapping.py:1519-1547); AMT_NO_MAG_ONLY=1: A/B runs
    static const bool no_mag_only = std::getenv("AMT_NO_MAG_ONLY") != nullptr;
    const bool no_geo_out = !A.lat && !A.lon && !A.lat_c && !A.lon_c;
    if (F->second == 1 && dirs == nullptr && !use_tiles && no_geo_out) {
        // fused binning on the MLat / MLT grid; without binning: the box in (MLat, SM longitude) (bin_magnetic alone), or
        // MLat / MLT arrays and no box at all (what BaseAstrometryMapping.mLatMlt asks for)
        if (bin ? (A.bin_magnetic && !no_mag_only) : (A.bin_magnetic || (!out->bbox && !no_mag_only))) F->second = 4;
    }
    AMT_REQUIRE(ctx, bin || !A.bin_magnetic || F->second == 4, "bin_magnetic without bin_acc needs the row-marching kernel");
    F->bin = bin;
    return AMT_OK;
}

// Launches n prepared frames: one k_georef_rows launch for all of them when they are equally sized and need the
// same kernel variant (their constants sit side by side in the kernel-argument segment), else one after the other.
int launch_prepared(amt_ctx* ctx, int n, prepared_frame* F) {
    AMT_REQUIRE(ctx, n >= 1 && n <= kMaxBatch, "bad batch size");
    bool together = n > 1;
    for (int i = 1; i < n && together; ++i)
        together = !F[0].sh.use_tiles && F[i].p->width == F[0].p->width && F[i].p->height == F[0].p->height &&
                   F[i].p->fast_center == F[0].p->fast_center && F[i].second == F[0].second && F[i].bin == F[0].bin &&
                   (F[i].dirs != nullptr) == (F[0].dirs != nullptr);
    if (n > 1 && !together) {
        for (int i = 0; i < n; ++i)
            if (int rc = launch_prepared(ctx, 1, F + i)) return rc;
        return AMT_OK;
    }
    const georef_args& A = F[0].A;
    const amt_frame_params* p = F[0].p;
    const double* dirs = F[0].dirs;
    const launch_shape& sh = F[0].sh;
    const bool use_tiles = sh.use_tiles, mag = F[0].mag;
    const int bin = F[0].bin, rows_per_chunk = sh.rows_per_chunk, strips_x = sh.strips_x;
    const int64_t n_items = sh.n_items;
    const int64_t nblocks = use_tiles ? n_items : ((int64_t)n * n_items + kRowsThreads / 64 - 1) / (kRowsThreads / 64);
    AMT_REQUIRE(ctx, (int64_t)n * n_items < (1ll << 31), "launch too large");
    const dim3 grid((unsigned)nblocks);
    // events ride on the dispatch packet itself: the timing pair when this launch is sampled, otherwise the
    // driver's hand-over event as the stop event; nothing is recorded between consecutive big kernels
    launch_events ev;
    amt_timing_pair(ctx, AMT_KERNEL_GEOREF, n, &ev.start, &ev.stop);
    for (int i = 0; i < n && ev.stop == nullptr; ++i)
        if (F[i].tail != nullptr && F[i].out->bbox) ev.stop = F[i].tail->kernel_done;
    if (use_tiles) {
        if (dirs) {
            launch_variant<true, true>(ctx, A, grid, mag, ev);
        } else if (p->fast_center) {
            launch_variant<true, false>(ctx, A, grid, mag, ev);
        } else {
            launch_variant<false, false>(ctx, A, grid, mag, ev);
        }
    } else {
        georef_batch B;
        ctx->last_second = F[0].second, ctx->last_bin = bin, ctx->last_frames = n;
        for (int i = 0; i < kMaxBatch; ++i) B.f[i] = F[i < n ? i : 0].A;
        B.math = fx::make_math_table();
        if (dirs) {
            launch_rows<true, true>(ctx, B, n, grid, F[0].second, bin, rows_per_chunk, strips_x, (int)n_items, ev);
        } else if (p->fast_center) {
            launch_rows<true, false>(ctx, B, n, grid, F[0].second, bin, rows_per_chunk, strips_x, (int)n_items, ev);
        } else {
            launch_rows<false, false>(ctx, B, n, grid, F[0].second, bin, rows_per_chunk, strips_x, (int)n_items, ev);
        }
    }
    AMT_LAUNCH_CHECK(ctx);
    // the folds: one launch per stage for all frames whose folds run on the same stream (the frame drivers of a context
    // share theirs)
    bool folded[kMaxBatch] = {false, false, false};
    for (int i = 0; i < n; ++i) {
        if (!F[i].out->bbox || folded[i]) continue;
        hipStream_t fs = F[i].tail ? F[i].tail->stream : ctx->stream;
        if (F[i].tail) AMT_HIP(ctx, hipStreamWaitEvent(fs, ev.stop, 0));
        fold_batch s0, s1;
        int m = 0;
        for (int k = 0; k < kMaxBatch; ++k) s0.in[k] = s1.in[k] = nullptr, s0.out[k] = s1.out[k] = nullptr, s0.extra[k] = s1.extra[k] = nullptr;
        for (int k = i; k < n; ++k) {
            if (!F[k].out->bbox || folded[k] || (F[k].tail ? F[k].tail->stream : ctx->stream) != fs) continue;
            s0.in[m] = F[k].A.bbox_partials, s0.out[m] = F[k].fold;
            s1.in[m] = F[k].fold, s1.out[m] = F[k].out->bbox, s1.extra[m] = F[k].A.bin_event_count;
            folded[k] = true;
            ++m;
        }
        hipLaunchKernelGGL(k_bbox_fold_batch, dim3(kFoldBlocks, m), dim3(kThreads), 0, fs, s0, (int)n_items);
        hipLaunchKernelGGL(k_bbox_fold_batch, dim3(1, m), dim3(kThreads), 0, fs, s1, kFoldBlocks);
        AMT_LAUNCH_CHECK(ctx);
    }
    return AMT_OK;
}

int launch_georef(amt_ctx* ctx, const amt_frame_params* p, const double* dirs, const amt_georef_out* out,
                  const amt_georef_tail* tail = nullptr) {
    prepared_frame F;
    if (int rc = prepare_georef(ctx, p, dirs, out, tail, &F)) return rc;
    return launch_prepared(ctx, 1, &F);
}


}  // namespace

size_t amt_georef_partials_bytes(const amt_frame_params* p) {
    return (size_t)(shape_of(p).n_items + kFoldBlocks) * 8 * sizeof(double);
}

int amt_georef_launch(amt_ctx* ctx, const amt_frame_params* p, const double* dirs, const amt_georef_out* out,
                      const amt_georef_tail* tail) {
    AMT_CHECK_CTX(ctx);
    return launch_georef(ctx, p, dirs, out, tail);
}

int amt_georef_launch_many(amt_ctx* ctx, int n, const amt_frame_params* const* p, const amt_georef_out* const* out,
                           const amt_georef_tail* const* tail) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n >= 1 && n <= kMaxBatch && p && out && tail, "bad batch");
    prepared_frame F[kMaxBatch];
    for (int i = 0; i < n; ++i)
        if (int rc = prepare_georef(ctx, p[i], nullptr, out[i], tail[i], &F[i])) return rc;
    return launch_prepared(ctx, n, F);
}

int amt_georef_launch_many_dirs(amt_ctx* ctx, int n, const amt_frame_params* const* p, const double* const* dirs,
                                const amt_georef_out* const* out, const amt_georef_tail* const* tail) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n >= 1 && n <= kMaxBatch && p && dirs && out && tail, "bad batch");
    prepared_frame F[kMaxBatch];
    for (int i = 0; i < n; ++i) {
        AMT_REQUIRE(ctx, dirs[i] != nullptr, "corner_dirs is NULL");
        if (int rc = prepare_georef(ctx, p[i], dirs[i], out[i], tail[i], &F[i])) return rc;
    }
    return launch_prepared(ctx, n, F);
}

extern "C" {

int amt_georef_frame(amt_ctx* ctx, const amt_frame_params* p, const amt_georef_out* out) {
    AMT_CHECK_CTX(ctx);
    return launch_georef(ctx, p, nullptr, out);
}

namespace {
int coarse_bbox(amt_ctx* ctx, const amt_frame_params* p, const double* dirs, int32_t stride, double min_elevation, int magnetic,
                double* bbox);
}

int amt_georef_coarse_bbox(amt_ctx* ctx, const amt_frame_params* p, int32_t stride, double min_elevation,
                           int magnetic, double* bbox) {
    AMT_CHECK_CTX(ctx);
    return coarse_bbox(ctx, p, nullptr, stride, min_elevation, magnetic, bbox);
}

int amt_georef_coarse_bbox_dirs(amt_ctx* ctx, const amt_frame_params* p, const double* corner_dirs, int32_t stride,
                                double min_elevation, int magnetic, double* bbox) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, corner_dirs != nullptr, "corner_dirs is NULL");
    AMT_REQUIRE(ctx, magnetic == 0 || magnetic == 1, "direction arrays have no pole plan: magnetic must be 0 or 1");
    return coarse_bbox(ctx, p, corner_dirs, stride, min_elevation, magnetic, bbox);
}

}  // extern "C"

namespace {
int coarse_bbox(amt_ctx* ctx, const amt_frame_params* p, const double* dirs, int32_t stride, double min_elevation, int magnetic,
                double* bbox) {
    AMT_REQUIRE(ctx, p && bbox, "NULL argument");
    AMT_REQUIRE(ctx, p->width > 0 && p->height > 0 && stride > 0, "bad frame or stride");
    georef_args A;
    std::memset(&A, 0, sizeof(A));
    A.wcs = make_tan_wcs(p);
    A.ray = make_ray(p->a, p->b, p->cam, 1);
    A.m_geo = make_mat3(p->m_geo);
    A.m_sm = make_mat3(p->m_sm);
    A.bw = make_bowring_fast(p->a0, p->b0);
    A.pole = make_pole_consts(p->a0, p->b0, p->a - p->a0);      // (coarse: the altitude need not be exact to the bit)
    A.width = p->width;
    A.height = p->height;
    const int nxl = (p->width + stride - 1) / stride + 1, nyl = (p->height + stride - 1) / stride + 1;
    const int nblocks = (nxl * nyl + kThreads - 1) / kThreads;
    double* partials = static_cast<double*>(amt_workspace(ctx, (size_t)nblocks * 8 * sizeof(double)));
    if (partials == nullptr) {
        ctx->last_error = "amt_georef_coarse_bbox: workspace allocation failed";
        return AMT_ENOMEM;
    }
    A.dirs_in = dirs;
    const int mode = (magnetic == 2 || magnetic == 3) ? magnetic : (magnetic ? 1 : 0);
    if (dirs)
        hipLaunchKernelGGL(k_coarse_bbox<true>, dim3(nblocks), dim3(kThreads), 0, ctx->stream, A, stride, mode, min_elevation, partials);
    else
        hipLaunchKernelGGL(k_coarse_bbox<false>, dim3(nblocks), dim3(kThreads), 0, ctx->stream, A, stride, mode, min_elevation, partials);
    AMT_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_bbox_fold, dim3(1), dim3(kThreads), 0, ctx->stream, partials, nblocks, bbox,
                       (const unsigned int*)nullptr);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}
}  // namespace

extern "C" {

int amt_georef_sky_rows(const amt_frame_params* p, int32_t* rows_per_item, int32_t* n_item_rows, int32_t* top_end,
                        int32_t* bottom_begin) {
    if (p == nullptr || rows_per_item == nullptr || n_item_rows == nullptr || top_end == nullptr || bottom_begin == nullptr)
        return AMT_EINVAL;
    if (p->width <= 0 || p->height <= 0) return AMT_EINVAL;
    georef_args A;
    std::memset(&A, 0, sizeof(A));
    A.wcs = make_tan_wcs(p);
    double rot_geo[9];
    mat_mul3(p->m_geo, p->rot, rot_geo);
    tan_wcs wcs_geo = A.wcs;
    wcs_geo.rot = make_mat3(rot_geo);
    A.cam = make_affine_cam(wcs_geo);
    A.sray = make_shell_ray(p->a, p->b, p->cam, p->m_geo);
    A.width = p->width;
    A.height = p->height;
    const launch_shape sh = shape_of(p);
    int t = 0, b = sh.chunks_y;
    sky_bands(A, sh, &t, &b);
    *rows_per_item = sh.rows_per_chunk;
    *n_item_rows = sh.chunks_y;
    *top_end = t;
    *bottom_begin = b;
    return AMT_OK;
}

int64_t amt_padded_pitch(int32_t width) {
    if (width <= 0) return 0;
    amt_frame_params p;
    std::memset(&p, 0, sizeof(p));
    p.width = width, p.height = 1;
    return (int64_t)shape_of(&p).strips_x * 64;
}

int amt_georef_image_rows(const amt_frame_params* p, double min_elevation, int32_t* row_begin, int32_t* row_end) {
    if (p == nullptr || row_begin == nullptr || row_end == nullptr) return AMT_EINVAL;
    if (p->width <= 0 || p->height <= 0) return AMT_EINVAL;
    georef_args A;
    std::memset(&A, 0, sizeof(A));
    A.wcs = make_tan_wcs(p);
    double rot_geo[9];
    mat_mul3(p->m_geo, p->rot, rot_geo);
    tan_wcs wcs_geo = A.wcs;
    wcs_geo.rot = make_mat3(rot_geo);
    A.cam = make_affine_cam(wcs_geo);
    A.sray = make_shell_ray(p->a, p->b, p->cam, p->m_geo);
    A.width = p->width;
    A.height = p->height;
    const launch_shape sh = shape_of(p);
    int t = 0, b = sh.chunks_y, te = 0, be = sh.chunks_y;
    sky_bands(A, sh, &t, &b);
    if (min_elevation > 0) elevation_bands(A, sh, min_elevation, &te, &be);
    t = std::max(t, te), b = std::min(b, be);
    if (t >= b) {
        *row_begin = *row_end = 0;                         // no pixel of the frame can be binned
        return AMT_OK;
    }
    *row_begin = std::max(0, t * sh.rows_per_chunk);
    *row_end = std::min(p->height, b * sh.rows_per_chunk);
    return AMT_OK;
}

int amt_georef_frame_dirs(amt_ctx* ctx, const amt_frame_params* p, const double* corner_dirs,
                          const amt_georef_out* out) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, corner_dirs != nullptr, "corner_dirs is NULL");
    return launch_georef(ctx, p, corner_dirs, out);
}

}  // extern "C"
