// resample(method='nearest') and the outside-outline masking of the interpolating methods
// (reference auromat/resample.py:246-259,301-327; utils.py:58-74).
//
// Nearest neighbour of every grid centre among the valid pixel centres, Euclidean in the (lat, lon) plane in
// degrees, as scipy.interpolate.griddata(method='nearest') (a cKDTree query) defines it.  On the device the
// search structure is the output grid itself: a counting sort of the source pixels by the grid cell they fall
// into (count -> exclusive scan -> fill), then one wavefront per grid centre visits the cells ring by ring and
// stops as soon as the best candidate is closer than anything an unvisited cell can hold.  HBM-bound: two passes
// over the centre coordinates plus ~9 cells x (pixels per cell) gathered coordinate pairs per grid centre.
#include "amt_common.h"

#include <cmath>
#include <cstring>

namespace {

using namespace amt;

constexpr int kBlock = 256;
constexpr int kScanThreads = 1024;

inline dim3 grid_for(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return dim3(static_cast<unsigned>(blocks));
}

#define AMT_GRID_STRIDE(i, n) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

struct nn_args {
    const double* lat_c;
    const double* lon_c;
    const double* elev;
    const uint8_t* mask;
    int64_t n;
    double min_elev;
    int use_elev_threshold;
    int lon_wrap;
    axis_dev ax, ay;
    int nx, ny;
};

__device__ __forceinline__ bool source_xy(const nn_args& A, int64_t i, double& x, double& y) {
    y = A.lat_c[i];
    x = A.lon_c[i];
    if (!(y == y) || !(x == x)) return false;
    if (A.mask && A.mask[i]) return false;
    if (A.use_elev_threshold && !(A.elev[i] >= A.min_elev)) return false;
    if (A.lon_wrap) x = wrap180_shifted(x);
    return true;
}

// cell of a source pixel; pixels beyond the outermost edges go to the border cells (they can still be the
// nearest neighbour of a border centre, and a border cell is never farther from any centre than their own)
__device__ __forceinline__ int source_cell(const nn_args& A, double x, double y) {
    int ix = bin_index<true>(A.ax, x) - 1, iy = bin_index<true>(A.ay, y) - 1;
    ix = ix < 0 ? 0 : (ix >= A.nx ? A.nx - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= A.ny ? A.ny - 1 : iy);
    return iy * A.nx + ix;
}

// Neighbouring pixels mostly fall into the same grid cell: lanes of a wave that hold a run of equal cells let the
// first lane of the run issue ONE atomic for all of them (6-10 atomics per wave instead of 64).
struct lane_run {
    int head;       // lane that starts this lane's run
    int length;     // pixels in the run (meaningful on the head lane)
};

__device__ __forceinline__ lane_run run_of(int c, int lane) {
    const int prev = __shfl_up(c, 1);
    const bool is_head = lane == 0 || prev != c;
    const unsigned long long heads = __ballot(is_head);
    const unsigned long long upto = heads & (~0ull >> (63 - lane));      // heads at or below this lane
    lane_run r;
    r.head = 63 - __clzll(upto);
    const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
    r.length = above ? __ffsll((long long)above) : 64 - lane;
    return r;
}

__global__ __launch_bounds__(kBlock) void k_nn_count(nn_args A, int* __restrict__ cell_of,
                                                    unsigned* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // whole waves stay in the loop together (the shuffles need every lane)
    for (int64_t base = blockIdx.x * (int64_t)blockDim.x + (threadIdx.x & ~63); base < A.n; base += stride) {
        const int64_t i = base + lane;
        double x, y;
        int c = -1;
        if (i < A.n && source_xy(A, i, x, y)) c = source_cell(A, x, y);
        if (i < A.n) cell_of[i] = c;
        const lane_run r = run_of(c, lane);
        if (r.head == lane && c >= 0) atomicAdd(&count[c], (unsigned)r.length);
    }
}

// exclusive prefix sum of count[0..n) into offset[0..n], one workgroup (n is the number of grid cells)
__global__ __launch_bounds__(kScanThreads) void k_nn_scan(const unsigned* __restrict__ count, int64_t n,
                                                         unsigned* __restrict__ offset) {
    __shared__ unsigned sWave[kScanThreads / 64];
    __shared__ unsigned sCarry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sCarry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += kScanThreads) {
        const int64_t i = base + threadIdx.x;
        const unsigned v = i < n ? count[i] : 0u;
        unsigned incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        unsigned before = sCarry;
        for (int w = 0; w < wave; ++w) before += sWave[w];
        if (i < n) offset[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == kScanThreads - 1) sCarry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offset[n] = sCarry;
}

__global__ __launch_bounds__(kBlock) void k_nn_fill(const int* __restrict__ cell_of, int64_t n,
                                                   const unsigned* __restrict__ offset,
                                                   unsigned* __restrict__ cursor, int* __restrict__ order) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = blockIdx.x * (int64_t)blockDim.x + (threadIdx.x & ~63); base < n; base += stride) {
        const int64_t i = base + lane;
        const int c = i < n ? cell_of[i] : -1;
        const lane_run r = run_of(c, lane);
        unsigned first = 0;
        if (r.head == lane && c >= 0) first = offset[c] + atomicAdd(&cursor[c], (unsigned)r.length);
        first = __shfl(first, r.head);
        if (c >= 0) order[first + (unsigned)(lane - r.head)] = (int)i;
    }
}

// One wavefront per grid centre: the lanes share the pixels of the cells of one ring, then reduce (distance, index).
__global__ __launch_bounds__(kBlock) void k_nn_search(nn_args A, const unsigned* __restrict__ offset,
                                                     const int* __restrict__ order,
                                                     const double* __restrict__ target_lat,
                                                     const double* __restrict__ target_lon,
                                                     const uint8_t* __restrict__ target_mask, double safe_step,
                                                     long long* __restrict__ out_index) {
    const int64_t total = (int64_t)A.nx * A.ny;
    const unsigned n_sources = offset[total];
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t t = wave; t < total; t += n_waves) {
        const int row = (int)(t / A.nx), col = (int)(t - (int64_t)row * A.nx);
        if ((target_mask && target_mask[t]) || n_sources == 0) {
            if (lane == 0) out_index[t] = -1;
            continue;
        }
        const double ty = target_lat[row], tx = target_lon[col];
        const int cy = A.ny - 1 - row;                 // histogram rows ascend in latitude, output rows descend
        double best = __builtin_huge_val();
        int best_i = 0x7fffffff;
        // pixels of the cells (iy, ix0..ix1), which are contiguous in `order`
        auto visit = [&](int iy, int ix0, int ix1) {
            if (iy < 0 || iy >= A.ny) return;
            ix0 = ix0 < 0 ? 0 : ix0;
            ix1 = ix1 >= A.nx ? A.nx - 1 : ix1;
            if (ix0 > ix1) return;
            const int c = iy * A.nx;
            for (unsigned k = offset[c + ix0] + lane, e = offset[c + ix1 + 1]; k < e; k += 64) {
                const int i = order[k];
                double y = A.lat_c[i], x = A.lon_c[i];
                if (A.lon_wrap) x = wrap180_shifted(x);
                const double dy = y - ty, dx = x - tx;
                const double d = dy * dy + dx * dx;
                if (d < best || (d == best && i < best_i)) {
                    best = d;
                    best_i = i;
                }
            }
        };
        const int r_max = max(max(col, A.nx - 1 - col), max(cy, A.ny - 1 - cy));
        for (int r = 0; r <= r_max; ++r) {
            if (r == 0) {
                visit(cy, col, col);
            } else {
                visit(cy - r, col - r, col + r);
                visit(cy + r, col - r, col + r);
                for (int iy = cy - r + 1; iy <= cy + r - 1; ++iy) {
                    visit(iy, col - r, col - r);
                    visit(iy, col + r, col + r);
                }
            }
            // wave minimum of (distance, index); every lane ends up with it
            for (int o = 32; o > 0; o >>= 1) {
                const double d2 = __shfl_xor(best, o);
                const int i2 = __shfl_xor(best_i, o);
                if (d2 < best || (d2 == best && i2 < best_i)) {
                    best = d2;
                    best_i = i2;
                }
            }
            // every source closer than (r + 1/2) cells in both axes lies in the rings visited so far
            const double reach = ((double)r + 0.5) * safe_step;
            if (best_i != 0x7fffffff && best <= reach * reach) break;
        }
        if (lane == 0) out_index[t] = best_i == 0x7fffffff ? -1 : best_i;
    }
}

template <typename T>
__global__ void k_nn_gather(const long long* __restrict__ index, int64_t total, const T* __restrict__ img, int nchan,
                            const double* __restrict__ elev, double* __restrict__ mean, T* __restrict__ out_img,
                            uint8_t* __restrict__ out_mask) {
    constexpr double kNaN = __builtin_nan("");
    AMT_GRID_STRIDE(t, total) {
        const long long i = index[t];
        for (int c = 0; c < nchan; ++c) {
            const T v = i >= 0 ? img[i * nchan + c] : (T)0;
            if (out_img) out_img[t * nchan + c] = v;
            if (mean) mean[t * (nchan + 1) + c] = i >= 0 ? (double)v : kNaN;
        }
        if (mean) mean[t * (nchan + 1) + nchan] = (i >= 0 && elev) ? elev[i] : kNaN;
        if (out_mask) out_mask[t] = i >= 0 ? 0 : 1;
    }
}

// ---- method='cubic' (reference resample.py:323-326: scipy griddata(method='cubic') = CloughTocher2DInterpolator): the element ----
// (12 boundary + 7 interior Bezier ordinates per triangle; the three free parameters fixed by making the derivative towards the
// neighbouring triangle's centroid linear along each edge, which is what makes scipy's element affine invariant; restated from
// scipy 1.15's interpnd module and checked against it to rounding on scipy's own triangulation: oracle/ref_numpy.py
// clough_tocher_*, tests/test_oracle_golden.py).  The lattice approximations of rounds 3-4 (k_linear_gather, k_cubic_sweep,
// k_cubic_gather: the Gauss-reduced local lattice of the pixel grid instead of the triangulation) were retired in round 6: 'linear'
// and 'cubic' run on the exact triangulation only (amt_delaunay_*, amt_cubic_gradients_csr, amt_cubic_eval below).

// value of the Clough-Tocher element of the triangle (x, y)[0..2] at barycentric coordinates b; f: vertex values, d: vertex
// gradients (dx, dy), n: whether edge k (opposite vertex k) has a neighbouring triangle, (cx, cy)[k] its centroid
__device__ double clough_tocher(const double* x, const double* y, const double* b, const double* f, const double (*d)[2],
                                const bool* n, const double* cx, const double* cy) {
    const double e12x = x[1] - x[0], e12y = y[1] - y[0], e23x = x[2] - x[1], e23y = y[2] - y[1], e31x = x[0] - x[2], e31y = y[0] - y[2];
    const double f1 = f[0], f2 = f[1], f3 = f[2];
    const double df12 = d[0][0] * e12x + d[0][1] * e12y, df21 = -(d[1][0] * e12x + d[1][1] * e12y);
    const double df23 = d[1][0] * e23x + d[1][1] * e23y, df32 = -(d[2][0] * e23x + d[2][1] * e23y);
    const double df31 = d[2][0] * e31x + d[2][1] * e31y, df13 = -(d[0][0] * e31x + d[0][1] * e31y);
    const double c3000 = f1, c2100 = (df12 + 3 * c3000) / 3, c2010 = (df13 + 3 * c3000) / 3;
    const double c0300 = f2, c1200 = (df21 + 3 * c0300) / 3, c0210 = (df23 + 3 * c0300) / 3;
    const double c0030 = f3, c1020 = (df31 + 3 * c0030) / 3, c0120 = (df32 + 3 * c0030) / 3;
    const double c2001 = (c2100 + c2010 + c3000) / 3, c0201 = (c1200 + c0300 + c0210) / 3, c0021 = (c1020 + c0120 + c0030) / 3;
    double g[3];
    const double det = e12x * (y[2] - y[0]) - (x[2] - x[0]) * e12y;
    for (int k = 0; k < 3; ++k) {
        g[k] = -0.5;
        if (!n[k]) continue;
        // barycentric coordinates of the neighbour's centroid in this triangle
        double c[3];
        c[1] = ((cx[k] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (cy[k] - y[0])) / det;
        c[2] = (e12x * (cy[k] - y[0]) - (cx[k] - x[0]) * e12y) / det;
        c[0] = 1.0 - c[1] - c[2];
        if (k == 0) g[k] = (2 * c[2] + c[1] - 1) / (2 - 3 * c[2] - 3 * c[1]);
        else if (k == 1) g[k] = (2 * c[0] + c[2] - 1) / (2 - 3 * c[0] - 3 * c[2]);
        else g[k] = (2 * c[1] + c[0] - 1) / (2 - 3 * c[1] - 3 * c[0]);
    }
    const double c0111 = (g[0] * (-c0300 + 3 * c0210 - 3 * c0120 + c0030) + (-c0300 + 2 * c0210 - c0120 + c0021 + c0201)) / 2;
    const double c1011 = (g[1] * (-c0030 + 3 * c1020 - 3 * c2010 + c3000) + (-c0030 + 2 * c1020 - c2010 + c2001 + c0021)) / 2;
    const double c1101 = (g[2] * (-c3000 + 3 * c2100 - 3 * c1200 + c0300) + (-c3000 + 2 * c2100 - c1200 + c2001 + c0201)) / 2;
    const double c1002 = (c1101 + c1011 + c2001) / 3, c0102 = (c1101 + c0111 + c0201) / 3, c0012 = (c1011 + c0111 + c0021) / 3;
    const double c0003 = (c1002 + c0102 + c0012) / 3;
    const double m = fmin(b[0], fmin(b[1], b[2]));
    const double b1 = b[0] - m, b2 = b[1] - m, b3 = b[2] - m, b4 = 3 * m;
    return b1 * b1 * b1 * c3000 + 3 * b1 * b1 * b2 * c2100 + 3 * b1 * b1 * b3 * c2010 + 3 * b1 * b1 * b4 * c2001 +
           3 * b1 * b2 * b2 * c1200 + 6 * b1 * b2 * b4 * c1101 + 3 * b1 * b3 * b3 * c1020 + 6 * b1 * b3 * b4 * c1011 +
           3 * b1 * b4 * b4 * c1002 + b2 * b2 * b2 * c0300 + 3 * b2 * b2 * b3 * c0210 + 3 * b2 * b2 * b4 * c0201 +
           3 * b2 * b3 * b3 * c0120 + 6 * b2 * b3 * b4 * c0111 + 3 * b2 * b4 * b4 * c0102 + b3 * b3 * b3 * c0030 +
           3 * b3 * b3 * b4 * c0021 + 3 * b3 * b4 * b4 * c0012 + b4 * b4 * b4 * c0003;
}

// ---- method='cubic' on the EXACT triangulation (round 5) ---------------------------------------------------------------------
// scipy's estimator is a Gauss-Seidel relaxation: point after point in the order of the input points, every point's 2 x 2
// system built from its neighbours' gradients AS THEY ARE AT THAT MOMENT — already updated in this sweep for neighbours with a
// smaller index, still the previous sweep's for the others —, repeated until the largest relative change of a sweep is below
// the tolerance.  The result depends on that order (the sweeps stop at 1e-6, far from the fixed point), so it is reproduced:
//   * neighbours from the Delaunay triangulation itself (amt_delaunay_vertex_neighbours: Qhull's hull-closing triangles and
//     its filling of holes included — their long edges pull on the border pixels' gradients, and the pull decays by a factor
//     of about three per ring of pixels, which is what kept the lattice-only estimate of rounds 3-4 from agreeing);
//   * one WAVE per pixel row walks its row's points in order; lanes are channels (each channel is its own relaxation with its
//     own stopping sweep, as scipy runs them one after the other).  A neighbour with a smaller index in ANOTHER row is written
//     by another wave: every gradient component is its own hand-over — the sweep's output array starts out filled with a
//     marker (a NaN no computation produces), components are written and read with relaxed device-scope atomics (8 bytes:
//     whole or not at all), and a reader that still finds the marker reads again.  No fences, no cache write-backs: a
//     component is either the marker or final.  Neighbours with a larger index are read from the previous sweep's array —
//     two arrays, so nobody can overtake.  Rows are handed out by a ticket in increasing order: a wave only ever waits for
//     rows that were taken before its own, whose waves are running or done, so the lowest unfinished row always advances.
//     (First version, with a stamp per point behind release / acquire fences: 1.35 s per sweep of 5.8 M points — every
//     fence wrote back or invalidated a whole L2; this form: see tools/cubic_full_probe.py.)
constexpr unsigned long long kGsMarker = 0x7ff8dead5eed0001ull;

struct gs_args {
    const double* xy;                // (n, 2) points: lat, lon as the reference hands them to griddata
    const long long* indptr;         // (n + 1)
    const int* indices;
    const long long* row_start;      // (n_rows + 1): first point of every pixel row (points are in row-major pixel order)
    int n_rows, nchan;
    const double* values;            // (n, nchan)
    const double* y_old;             // (n, nchan, 2)
    unsigned long long* y_new;       // (n, nchan, 2) bit patterns, kGsMarker = not written yet
    unsigned int* ticket;
    unsigned long long* err;         // (nchan <= 63): largest relative change of the sweep (bits of a non-negative double);
                                     // err[63]: set when a hand-over did not arrive (see gs_read)
    const unsigned char* active;     // (nchan): 0 = this channel has converged: its gradients are carried over
    const unsigned long long* rec;   // (n, 32) the points' geometry records (k_cubic_geometry)
};

__global__ void k_fill_u64(unsigned long long* __restrict__ p, int64_t n, unsigned long long v) {
    AMT_GRID_STRIDE(i, n) p[i] = v;
}

// (bounded: a hand-over that never comes — it cannot, by the ticket order, unless the device is being torn down — ends in a NaN
// and a flag for the host after ~2 s instead of a wave that never finishes)
constexpr int kGsMaxSpins = 1 << 26;
__device__ __forceinline__ double gs_read(const unsigned long long* p, unsigned long long* stalled) {
    unsigned long long b = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spins = 0; b == kGsMarker; ++spins) {
        if (spins >= kGsMaxSpins) {
            atomicMax(stalled, 1ull);
            return NAN;
        }
        __builtin_amdgcn_s_sleep(1);
        b = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __longlong_as_double((long long)b);
}

// a gradient component as it is handed over: a NaN (from NaN data) in its canonical form, never the marker's bit pattern
__device__ __forceinline__ unsigned long long gs_bits(double v) {
    return v == v ? (unsigned long long)__double_as_longlong(v) : 0x7ff8000000000000ull;
}

// The geometry of a point's system does not change from sweep to sweep or from channel to channel: one 256-byte record per point,
// made once per call (k_cubic_geometry) — the number of neighbours, the first eight neighbours' indices, edge vectors and l^3, and
// the 2 x 2 matrix (q0, q1, q3) summed over ALL neighbours in their order, which is what scipy recomputes for every point of every
// sweep of every channel (the same operations on the same numbers: the same bits).  A sweep then only gathers the right-hand side.
constexpr int kGsPre = 8;           // neighbours held in the record (a lattice point has 6; hull points with more take the slow way)
constexpr int kGsRecWords = 32;     // 8-byte words per record: [0] count, [1..4] indices, [5..12] ex, [13..20] ey, [21..28] l^3, [29..31] q0 q1 q3

__global__ __launch_bounds__(kBlock) void k_cubic_geometry(const double* __restrict__ xy, const long long* __restrict__ indptr,
                                                            const int* __restrict__ indices, int64_t n,
                                                            unsigned long long* __restrict__ rec) {
    AMT_GRID_STRIDE(v, n) {
        const long long b = indptr[v], e = indptr[v + 1];
        const double xi = xy[2 * v], yi = xy[2 * v + 1];
        unsigned long long* r = rec + v * kGsRecWords;
        double q0 = 0, q1 = 0, q3 = 0;
        int jj[kGsPre];
        for (int t = 0; t < kGsPre; ++t) jj[t] = -1;
        for (long long k = b; k < e; ++k) {
            const long long j = indices[k];
            const double ex = xy[2 * j] - xi, ey = xy[2 * j + 1] - yi;
            const double l = sqrt(ex * ex + ey * ey), l3 = l * l * l;
            q0 += 4 * ex * ex / l3;
            q1 += 4 * ex * ey / l3;
            q3 += 4 * ey * ey / l3;
            const int t = (int)(k - b);
            if (t < kGsPre) {
                jj[t] = (int)j;
                r[5 + t] = (unsigned long long)__double_as_longlong(ex);
                r[13 + t] = (unsigned long long)__double_as_longlong(ey);
                r[21 + t] = (unsigned long long)__double_as_longlong(l3);
            }
        }
        for (int t = (int)(e - b); t < kGsPre; ++t) r[5 + t] = r[13 + t] = r[21 + t] = 0;
        r[0] = (unsigned long long)(e - b);
        for (int t = 0; t < kGsPre; t += 2) r[1 + t / 2] = (unsigned long long)(unsigned int)jj[t] | ((unsigned long long)(unsigned int)jj[t + 1] << 32);
        r[29] = (unsigned long long)__double_as_longlong(q0);
        r[30] = (unsigned long long)__double_as_longlong(q1);
        r[31] = (unsigned long long)__double_as_longlong(q3);
    }
}

// Lanes of a wave are (neighbour, channel) pairs — eight neighbours x eight channels —: every lane gathers ONE term of a point's
// right-hand side, and the eight terms of a channel are then added in the order of this library's neighbour lists (ascending
// vertex index, build_vertex_lists; lane to lane, ds_bpermute): a fixed order, so the result does not depend on the schedule.  It is
// NOT scipy's rounding to the bit — scipy lists a vertex's neighbours in the order Qhull's simplices first mention them and
// evaluates a term as (..) * ex / l3 where this code multiplies by a stored 1 / l3 —: equal to scipy up to summation order (1e-13
// of a channel's span on every fixture), and, because the stopping rule `error < tol` is a yes / no decision per sweep, a channel
// whose sweep error lands within rounding of the tolerance could stop one sweep apart from scipy (a difference of ~1e-6 relative;
// not observed: amt_cubic_gradients_csr returns the sweep count per channel for exactly that check).  (With lanes as channels only — the first form — four of 64 lanes worked for an RGB image
// and its elevation, and a step of the sweep's critical path was the ~1000 instructions of all eight terms one after the other.)
constexpr int kGsChan = 8;          // channels per wave; more channels: more waves per row (blockIdx.y), each with its own ticket

// what one lane needs for its term of point u: from the record (j, ex, ey, l3; count and matrix: the same in every lane) and,
// addressed through j, the neighbour's value and gradient
struct gs_term {
    int j, m;
    double ex, ey, l3, q0, q1, q3;
    double fi, g0, g1;                                   // the point's own value and previous gradient (this lane's channel)
    double fj;
    unsigned long long y0, y1;                           // the neighbour's gradient: the previous sweep's (j > u) or this sweep's
                                                         // (a hand-over: the marker where not written yet)
};

// Software pipeline of a row: when point v is done, the lane's registers for it take the DATA of point v + 3, addressed through
// the record fields of v + 3 (requested a step earlier), and the RECORD fields of point v + 4 are requested: every load is issued
// two to three steps before its use, with no branch in between (an absent neighbour reads the point itself; which sweep's array
// a gradient comes from is a select on the address).  A row runs a few points behind the row above it, so a hand-over is
// usually final when it is fetched; one that still shows the marker — another row's, or this row's own point of three steps ago
// whose store the load overtook — is read again when its turn comes (gs_read): a component is the marker or final, never
// anything else.  The sweep's critical path is (row length + lag x rows) steps long: with the loads of a step issued at the step
// itself a step took 10 us (three dependent round trips: CSR pointer, indices, data), with everything one point ahead 9 (the
// same chain, merely started earlier), with records and the loads ahead 3.4 (lanes as channels: instruction latency), in this
// form see tools/cubic_full_probe.py.
__global__ __launch_bounds__(64) void k_cubic_gs(gs_args A) {
    const int lane = threadIdx.x;
    const int t = lane >> 3, c = lane & 7;                // this lane's neighbour slot and channel within the group
    const int group = blockIdx.y;
    int row = 0;
    if (lane == 0) row = (int)atomicAdd(A.ticket + group, 1u);
    row = __builtin_amdgcn_readfirstlane(row);
    if (row >= A.n_rows) return;
    const long long v0 = A.row_start[row], v1 = A.row_start[row + 1];
    if (v0 >= v1) return;
    const int ch = group * kGsChan + c;
    const bool chan = ch < A.nchan;
    const int cl = chan ? ch : 0;                         // (lanes beyond the channels work on channel 0 and store nothing)
    const bool live = chan && A.active[cl] != 0;
    const long long last = v1 - 1;
    const unsigned long long* const y_old = reinterpret_cast<const unsigned long long*>(A.y_old);
    struct rec_fields {
        unsigned int j, m;
        unsigned long long ex, ey, l3, q0, q1, q3;
    };
    auto request_record = [&](long long u, rec_fields& R) {
        u = u < last ? u : last;                          // (past the row's end: the last record again, never used)
        const unsigned long long* r = A.rec + u * kGsRecWords;
        R.j = reinterpret_cast<const unsigned int*>(r + 1)[t];
        R.m = reinterpret_cast<const unsigned int*>(r)[0];
        R.ex = r[5 + t], R.ey = r[13 + t], R.l3 = r[21 + t];
        R.q0 = r[29], R.q1 = r[30], R.q3 = r[31];
    };
    auto request_data = [&](const rec_fields& R, long long u, gs_term& D) {
        u = u < last ? u : last;
        D.j = (int)R.j, D.m = (int)R.m;
        D.ex = __longlong_as_double((long long)R.ex), D.ey = __longlong_as_double((long long)R.ey);
        D.l3 = __longlong_as_double((long long)R.l3);
        D.q0 = __longlong_as_double((long long)R.q0), D.q1 = __longlong_as_double((long long)R.q1);
        D.q3 = __longlong_as_double((long long)R.q3);
        const long long o = (u * A.nchan + cl) * 2;
        D.fi = A.values[u * A.nchan + cl];
        D.g0 = A.y_old[o], D.g1 = A.y_old[o + 1];
        const long long j = D.j < 0 ? u : (long long)D.j;
        const long long q = (j * A.nchan + cl) * 2;
        const unsigned long long* src = j > u ? y_old : A.y_new;
        D.fj = A.values[j * A.nchan + cl];
        D.y0 = __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        D.y1 = __hip_atomic_load(src + q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    double worst = 0;
    // this row's latest two points' new gradients of this lane's channel (every lane of a channel computes them)
    double p0 = 0, p1 = 0, pp0 = 0, pp1 = 0;
    rec_fields pending;
    gs_term d0, d1, d2;
    request_record(v0, pending);
    request_data(pending, v0, d0);
    request_record(v0 + 1, pending);
    request_data(pending, v0 + 1, d1);
    request_record(v0 + 2, pending);
    request_data(pending, v0 + 2, d2);
    request_record(v0 + 3, pending);
    auto step = [&](const long long v, gs_term& cur) {
        const long long o = (v * A.nchan + cl) * 2;
        double g0 = cur.g0, g1 = cur.g1;
        {
            // this sweep's gradient of a neighbour with a smaller index: the previous point or the one before it come from
            // registers, anything older — another row's or this row's — is a hand-over
            auto earlier = [&](long long j, unsigned long long b0, unsigned long long b1, double& y0, double& y1) {
                if (j == v - 1 && j >= v0) {
                    y0 = p0, y1 = p1;
                } else if (j == v - 2 && j >= v0) {
                    y0 = pp0, y1 = pp1;
                } else {
                    const unsigned long long* q = A.y_new + (j * A.nchan + cl) * 2;
                    y0 = b0 == kGsMarker ? gs_read(q, A.err + 63) : __longlong_as_double((long long)b0);
                    y1 = b1 == kGsMarker ? gs_read(q + 1, A.err + 63) : __longlong_as_double((long long)b1);
                }
            };
            auto term = [&](double ex, double ey, double l3, double fj, double y0, double y1, double& a, double& b) {
                const double df2 = -ex * y0 - ey * y1;
                const double tt = (6 * (cur.fi - fj) - 2 * df2) / l3;
                a = tt * ex, b = tt * ey;
            };
            double a = 0, b = 0;
            if (live && t < cur.m) {
                const long long j = cur.j;
                double y0 = __longlong_as_double((long long)cur.y0), y1 = __longlong_as_double((long long)cur.y1);          // j > v
                if (j < v) earlier(j, cur.y0, cur.y1, y0, y1);
                term(cur.ex, cur.ey, cur.l3, cur.fj, y0, y1, a, b);
            }
            // the channel's eight terms, added in the order of the neighbours (an absent neighbour adds + 0)
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int k = 0; k < kGsPre; ++k) {
                s0 += __shfl(a, k * kGsChan + c);
                s1 += __shfl(b, k * kGsChan + c);
            }
            if (cur.m > kGsPre) {
                // More neighbours than a record holds: the vertices that close the hull round a concave outline have hundreds
                // (one at a time — the first form — they were most of a sweep: ~1400 dependent round trips, each holding up
                // every row behind).  Eight at a time, a lane per neighbour and channel like the record's eight; the next eight
                // indices are on their way while these are worked on.
                const long long kb = A.indptr[v], ke = kb + cur.m;
                const double xi = A.xy[2 * v], yi = A.xy[2 * v + 1];
                long long k = kb + kGsPre + t;
                long long j_next = k < ke ? (long long)A.indices[k] : v;
                for (long long k0 = kb + kGsPre; k0 < ke; k0 += kGsPre) {
                    const bool present = k < ke;
                    const long long j = j_next;
                    k += kGsPre;
                    j_next = k < ke ? (long long)A.indices[k] : v;
                    double ea = 0, eb = 0;
                    const double ex = A.xy[2 * j] - xi, ey = A.xy[2 * j + 1] - yi;
                    const double fj = A.values[j * A.nchan + cl];
                    const unsigned long long* src = j > v ? y_old : A.y_new;
                    const unsigned long long b0 = __hip_atomic_load(src + (j * A.nchan + cl) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long b1 = __hip_atomic_load(src + (j * A.nchan + cl) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (live && present) {
                        double y0 = __longlong_as_double((long long)b0), y1 = __longlong_as_double((long long)b1);
                        if (j < v) earlier(j, b0, b1, y0, y1);
                        const double l = sqrt(ex * ex + ey * ey), l3 = l * l * l;
                        term(ex, ey, l3, fj, y0, y1, ea, eb);
                    }
#pragma unroll
                    for (int kk = 0; kk < kGsPre; ++kk) {
                        s0 += __shfl(ea, kk * kGsChan + c);
                        s1 += __shfl(eb, kk * kGsChan + c);
                    }
                }
            }
            if (live) {
                const double q0 = cur.q0, q1 = cur.q1, q3 = cur.q3;
                const double det = q0 * q3 - q1 * q1;
                const double r0_ = (q3 * s0 - q1 * s1) / det, r1_ = (-q1 * s0 + q0 * s1) / det;
                double change = fmax(fabs(g0 + r0_), fabs(g1 + r1_));
                change /= fmax(1.0, fmax(fabs(r0_), fabs(r1_)));
                if (change == change) worst = fmax(worst, change);
                g0 = -r0_, g1 = -r1_;
            }
        }
        pp0 = p0, pp1 = p1;
        p0 = g0, p1 = g1;
        if (chan && t == 0) {
            __hip_atomic_store(A.y_new + o, gs_bits(g0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.y_new + o + 1, gs_bits(g1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        request_data(pending, v + 3, cur);                                // this point's registers: the point three further on
        request_record(v + 4, pending);
    };
    for (long long v = v0; v < v1; v += 3) {
        step(v, d0);
        if (v + 1 < v1) step(v + 1, d1);
        if (v + 2 < v1) step(v + 2, d2);
    }
    if (live && t == 0 && worst > 0) atomicMax(&A.err[ch], (unsigned long long)__double_as_longlong(worst));
}

// The element in the triangle amt_delaunay_locate found for every target (vertices -1: outside the hull -> NaN)
__global__ __launch_bounds__(kBlock) void k_cubic_eval(int64_t m, const double* __restrict__ targets, const int* __restrict__ vertices,
                                                        const double* __restrict__ centroids, const unsigned char* __restrict__ has_nb,
                                                        const double* __restrict__ xy, const double* __restrict__ values,
                                                        const double* __restrict__ grad, int nchan, double* __restrict__ out) {
    AMT_GRID_STRIDE(t, m) {
        const int* w = vertices + 3 * t;
        if (w[0] < 0) {
            for (int c = 0; c < nchan; ++c) out[t * nchan + c] = NAN;
            continue;
        }
        double x[3], y[3], cx[3], cy[3], b[3];
        bool n[3];
        for (int k = 0; k < 3; ++k) {
            x[k] = xy[2 * (int64_t)w[k]], y[k] = xy[2 * (int64_t)w[k] + 1];
            n[k] = has_nb[3 * t + k] != 0;
            cx[k] = centroids[6 * t + 2 * k], cy[k] = centroids[6 * t + 2 * k + 1];
        }
        const double px = targets[2 * t], py = targets[2 * t + 1];
        const double det = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
        b[1] = ((px - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (py - y[0])) / det;
        b[2] = ((x[1] - x[0]) * (py - y[0]) - (px - x[0]) * (y[1] - y[0])) / det;
        b[0] = 1.0 - b[1] - b[2];
        for (int c = 0; c < nchan; ++c) {
            double f[3], d[3][2];
            for (int k = 0; k < 3; ++k) {
                f[k] = values[(int64_t)w[k] * nchan + c];
                d[k][0] = grad[((int64_t)w[k] * nchan + c) * 2], d[k][1] = grad[((int64_t)w[k] * nchan + c) * 2 + 1];
            }
            out[t * nchan + c] = clough_tocher(x, y, b, f, d, n, cx, cy);
        }
    }
}

// matplotlib.path.Path(polygon).contains_points(points) (reference utils.py:58-74): crossing test of a ray towards
// +x with the half-open edge rule (vertex y >= point y) of Agg's point_in_path; the path is closed implicitly.
// Edges are staged through LDS in chunks; an edge whose y-range misses the y-range of the block's points cannot
// change any of them and is dropped while staging (callers order the points so that a block is narrow in y).
__global__ __launch_bounds__(kBlock) void k_points_in_polygon(const double* __restrict__ px,
                                                             const double* __restrict__ py, int64_t n,
                                                             const double* __restrict__ poly, int m,
                                                             uint8_t* __restrict__ inside) {
    __shared__ double sx0[kBlock], sy0[kBlock], sx1[kBlock], sy1[kBlock];
    __shared__ double sLo[kBlock / 64], sHi[kBlock / 64];
    __shared__ int sCount;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool live = i < n;
    const double tx = live ? px[i] : 0.0, ty = live ? py[i] : 0.0;
    // y-range of this block's points (NaN coordinates never compare, they stay outside)
    double lo = live && ty == ty ? ty : __builtin_huge_val(), hi = live && ty == ty ? ty : -__builtin_huge_val();
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o));
        hi = fmax(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        sLo[threadIdx.x >> 6] = lo;
        sHi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    for (int w = 0; w < kBlock / 64; ++w) {
        lo = fmin(lo, sLo[w]);
        hi = fmax(hi, sHi[w]);
    }
    bool in = false;
    for (int base = 0; base < m; base += kBlock) {
        __syncthreads();
        if (threadIdx.x == 0) sCount = 0;
        __syncthreads();
        const int k = base + (int)threadIdx.x;
        if (k < m) {
            const int k1 = k + 1 == m ? 0 : k + 1;
            const double x0 = poly[2 * (int64_t)k], y0 = poly[2 * (int64_t)k + 1];
            const double x1 = poly[2 * (int64_t)k1], y1 = poly[2 * (int64_t)k1 + 1];
            // the edge flips a point only if (y0 >= ty) != (y1 >= ty) for some ty in [lo, hi]
            if (fmax(y0, y1) >= lo && fmin(y0, y1) <= hi) {
                const int slot = atomicAdd(&sCount, 1);
                sx0[slot] = x0;
                sy0[slot] = y0;
                sx1[slot] = x1;
                sy1[slot] = y1;
            }
        }
        __syncthreads();
        const int cnt = sCount;
        for (int e = 0; e < cnt; ++e) {
            const double x0 = sx0[e], y0 = sy0[e], x1 = sx1[e], y1 = sy1[e];
            const bool f0 = y0 >= ty, f1 = y1 >= ty;
            if (f0 != f1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1)) in = !in;
        }
    }
    if (live) inside[i] = in ? 1 : 0;
}

bool uniform_axis_ok(const amt_axis* a) {
    return axis_ok(a) && a->uniform == 1 && a->step > 0;
}

}  // namespace

extern "C" {

int amt_nearest_frame(amt_ctx* ctx, const double* lat_c, const double* lon_c, const double* elev,
                      const uint8_t* center_mask, int32_t height, int32_t width, double min_elevation,
                      const amt_axis* xaxis, const amt_axis* yaxis, int lon_wrap, const double* target_lat,
                      const double* target_lon, const uint8_t* target_mask, int64_t* out_index) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, lat_c && lon_c && xaxis && yaxis && target_lat && target_lon && out_index, "NULL argument");
    AMT_REQUIRE(ctx, height > 0 && width > 0, "empty frame");
    AMT_REQUIRE(ctx, (int64_t)height * width < 2147483647LL, "frame too large for 32-bit pixel indices");
    AMT_REQUIRE(ctx, uniform_axis_ok(xaxis) && uniform_axis_ok(yaxis), "axes must be uniform (amt_grid_layout)");
    AMT_REQUIRE(ctx, (int64_t)xaxis->nbin * yaxis->nbin < 2147483647LL, "grid too large");
    nn_args A;
    A.lat_c = lat_c;
    A.lon_c = lon_c;
    A.elev = elev;
    A.mask = center_mask;
    A.n = (int64_t)height * width;
    A.min_elev = min_elevation;
    A.use_elev_threshold = (elev != nullptr) && !(std::isinf(min_elevation) && min_elevation < 0);
    A.lon_wrap = lon_wrap ? 1 : 0;
    make_axis(xaxis, &A.ax);
    make_axis(yaxis, &A.ay);
    A.nx = xaxis->nbin;
    A.ny = yaxis->nbin;
    const int64_t cells = (int64_t)A.nx * A.ny;
    // workspace: count[cells], cursor[cells], offset[cells + 1] (u32), cell_of[n], order[n] (i32)
    const size_t bytes = (size_t)(3 * cells + 1) * sizeof(unsigned) + (size_t)2 * A.n * sizeof(int) + 64;
    char* ws = static_cast<char*>(amt_workspace(ctx, bytes));
    if (ws == nullptr) {
        ctx->last_error = "amt_nearest_frame: workspace allocation failed";
        return AMT_ENOMEM;
    }
    unsigned* count = reinterpret_cast<unsigned*>(ws);
    unsigned* cursor = count + cells;
    unsigned* offset = cursor + cells;
    int* cell_of = reinterpret_cast<int*>(offset + cells + 1);
    int* order = cell_of + A.n;
    if (hipMemsetAsync(count, 0, (size_t)2 * cells * sizeof(unsigned), ctx->stream) != hipSuccess) {
        ctx->last_error = "amt_nearest_frame: memset failed";
        return AMT_EHIP;
    }
    hipLaunchKernelGGL(k_nn_count, grid_for(A.n), dim3(kBlock), 0, ctx->stream, A, cell_of, count);
    hipLaunchKernelGGL(k_nn_scan, dim3(1), dim3(kScanThreads), 0, ctx->stream, count, cells, offset);
    hipLaunchKernelGGL(k_nn_fill, grid_for(A.n), dim3(kBlock), 0, ctx->stream, cell_of, A.n, offset, cursor, order);
    // cell sizes as the centres see them, with a margin for the rounding of edges and centres
    const double safe_step = std::fmin(xaxis->step, yaxis->step) * (1.0 - 1e-9);
    hipLaunchKernelGGL(k_nn_search, grid_for(cells * 64), dim3(kBlock), 0, ctx->stream, A, offset, order, target_lat,
                       target_lon, target_mask, safe_step, reinterpret_cast<long long*>(out_index));
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_nearest_gather(amt_ctx* ctx, const int64_t* index, int64_t n_targets, const void* img, int32_t img_dtype,
                       int32_t nchan, const double* elev, double* mean, void* out_img, uint8_t* out_mask) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, index != nullptr && n_targets >= 0, "NULL argument");
    AMT_REQUIRE(ctx, nchan >= 0 && nchan <= 4, "nchan must be 0..4");
    AMT_REQUIRE(ctx, nchan == 0 || (img && (img_dtype == 1 || img_dtype == 2)), "img must be uint8 (1) or uint16 (2)");
    if (n_targets == 0) return AMT_OK;
    const long long* idx = reinterpret_cast<const long long*>(index);
    if (img_dtype == 2) {
        hipLaunchKernelGGL(k_nn_gather<uint16_t>, grid_for(n_targets), dim3(kBlock), 0, ctx->stream, idx, n_targets,
                           static_cast<const uint16_t*>(img), nchan, elev, mean, static_cast<uint16_t*>(out_img),
                           out_mask);
    } else {
        hipLaunchKernelGGL(k_nn_gather<uint8_t>, grid_for(n_targets), dim3(kBlock), 0, ctx->stream, idx, n_targets,
                           static_cast<const uint8_t*>(img), nchan, elev, mean, static_cast<uint8_t*>(out_img),
                           out_mask);
    }
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_cubic_gradients_csr(amt_ctx* ctx, const double* xy, int64_t n, const int64_t* indptr, const int32_t* indices,
                            const int64_t* row_start, int32_t n_rows, const double* values, int32_t nchan, double tolerance,
                            int32_t max_iterations, double* gradients, int32_t* iterations) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, xy && indptr && indices && row_start && values && gradients && iterations, "NULL argument");
    AMT_REQUIRE(ctx, n >= 3 && n < 2147483647LL && n_rows >= 1, "bad size");
    AMT_REQUIRE(ctx, nchan >= 1 && nchan <= 63, "1..63 channels");
    AMT_REQUIRE(ctx, tolerance > 0 && max_iterations >= 1, "tolerance and max_iterations must be positive");
    const size_t grad_bytes = ((size_t)n * nchan * 2 * sizeof(double) + 255) & ~(size_t)255;
    const size_t rec_bytes = (size_t)n * kGsRecWords * sizeof(unsigned long long);
    char* ws = static_cast<char*>(amt_workspace(ctx, grad_bytes + 1024 + rec_bytes));
    if (ws == nullptr) {
        ctx->last_error = "amt_cubic_gradients_csr: workspace allocation failed";
        return AMT_ENOMEM;
    }
    double* other = reinterpret_cast<double*>(ws);
    unsigned long long* err = reinterpret_cast<unsigned long long*>(ws + grad_bytes);      // [64]
    unsigned int* ticket = reinterpret_cast<unsigned int*>(ws + grad_bytes + 512);
    unsigned char* active = reinterpret_cast<unsigned char*>(ws + grad_bytes + 576);      // [64]
    unsigned long long* rec = reinterpret_cast<unsigned long long*>(ws + grad_bytes + 1024);
    AMT_HIP(ctx, hipMemsetAsync(gradients, 0, (size_t)n * nchan * 2 * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_cubic_geometry, grid_for(n), dim3(kBlock), 0, ctx->stream, xy, reinterpret_cast<const long long*>(indptr), indices,
                       n, rec);
    AMT_LAUNCH_CHECK(ctx);
    unsigned char host_active[64];
    for (int c = 0; c < 64; ++c) host_active[c] = c < nchan ? 1 : 0;
    for (int c = 0; c < nchan; ++c) iterations[c] = 0;
    AMT_HIP(ctx, hipMemcpyAsync(active, host_active, 64, hipMemcpyHostToDevice, ctx->stream));
    gs_args A;
    A.xy = xy, A.indptr = reinterpret_cast<const long long*>(indptr), A.indices = indices;
    A.row_start = reinterpret_cast<const long long*>(row_start), A.n_rows = n_rows, A.nchan = nchan;
    A.values = values, A.ticket = ticket, A.err = err, A.active = active, A.rec = rec;
    double* bufs[2] = {gradients, other};
    int cur = 0;                                   // bufs[cur] holds the latest sweep
    const int64_t n_comp = n * nchan * 2;
    for (int it = 1; it <= max_iterations; ++it) {
        AMT_HIP(ctx, hipMemsetAsync(err, 0, 512 + 64, ctx->stream));              // err[64] and the ticket
        A.y_old = bufs[cur], A.y_new = reinterpret_cast<unsigned long long*>(bufs[1 - cur]);
        hipLaunchKernelGGL(k_fill_u64, grid_for(n_comp), dim3(kBlock), 0, ctx->stream, A.y_new, n_comp, kGsMarker);
        hipLaunchKernelGGL(k_cubic_gs, dim3((unsigned)n_rows, (unsigned)((nchan + kGsChan - 1) / kGsChan)), dim3(64), 0, ctx->stream, A);
        AMT_LAUNCH_CHECK(ctx);
        unsigned long long bits[64];
        AMT_HIP(ctx, hipMemcpyAsync(bits, err, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
        AMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (bits[63] != 0) {
            ctx->last_error = "amt_cubic_gradients_csr: a sweep stalled (a point's gradient was never handed over)";
            return AMT_EHIP;
        }
        bool any = false, changed = false;
        for (int c = 0; c < nchan; ++c) {
            if (!host_active[c]) continue;
            iterations[c] = it;
            double worst;
            std::memcpy(&worst, &bits[c], sizeof(worst));
            if (worst < tolerance) {
                host_active[c] = 0;                  // scipy returns after the sweep whose largest change is below the tolerance
                changed = true;
            } else {
                any = true;
            }
        }
        cur = 1 - cur;
        if (!any) break;
        if (changed) AMT_HIP(ctx, hipMemcpyAsync(active, host_active, 64, hipMemcpyHostToDevice, ctx->stream));
    }
    if (cur != 0)
        AMT_HIP(ctx, hipMemcpyAsync(gradients, other, (size_t)n * nchan * 2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return AMT_OK;
}

int amt_cubic_eval(amt_ctx* ctx, int64_t m, const double* targets, const int32_t* vertices, const double* centroids,
                   const uint8_t* has_neighbour, const double* xy, const double* values, const double* gradients, int32_t nchan,
                   double* out) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, m >= 0 && nchan >= 1, "bad size");
    if (m == 0) return AMT_OK;
    AMT_REQUIRE(ctx, targets && vertices && centroids && has_neighbour && xy && values && gradients && out, "NULL argument");
    hipLaunchKernelGGL(k_cubic_eval, grid_for(m), dim3(kBlock), 0, ctx->stream, m, targets, vertices, centroids, has_neighbour, xy,
                       values, gradients, nchan, out);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_points_in_polygon(amt_ctx* ctx, const double* px, const double* py, int64_t n, const double* polygon,
                          int32_t n_vertices, uint8_t* out_inside) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n == 0 || (px && py && out_inside), "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && n_vertices >= 0 && (n_vertices == 0 || polygon), "bad size or NULL polygon");
    if (n == 0) return AMT_OK;
    if (n_vertices < 3) {                               // no area: nothing is inside
        if (hipMemsetAsync(out_inside, 0, (size_t)n, ctx->stream) != hipSuccess) {
            ctx->last_error = "amt_points_in_polygon: memset failed";
            return AMT_EHIP;
        }
        return AMT_OK;
    }
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_points_in_polygon, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, px, py, n, polygon,
                       n_vertices, out_inside);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

}  // extern "C"
