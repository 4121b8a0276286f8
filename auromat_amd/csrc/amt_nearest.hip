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

// ---- method='linear' (reference resample.py:323-326: scipy.interpolate.griddata(method='linear'), i.e. barycentric
// interpolation on a Delaunay triangulation of the valid pixel centres in the (lat, lon) plane) ----------------------
// The pixel centres are a smoothly mapped regular grid, so their Delaunay triangulation is, locally, that of a lattice:
// the cells of the REDUCED basis (Gauss reduction of the two grid steps at the pixel centre nearest to the grid centre,
// amt_nearest_frame) cut along the diagonal the empty-circle criterion picks; a cell with three valid centres is that
// triangle.  Where a cell is close to cocircular Qhull's triangulation takes either diagonal: values then differ
// from scipy's within the spread of the two diagonal interpolants of the cell, which the kernel reports beside the
// value (`alt`); tests/test_gpu_nearest.py pins the rule.
struct lin_args {
    const double* lat_c;
    const double* lon_c;
    const double* elev;
    const uint8_t* center_mask;
    int height, width;
    double min_elev;
    int lon_wrap, nchan;
};

__device__ __forceinline__ bool lin_valid(const lin_args& A, int i, int j, double* x, double* y) {
    if (i < 0 || j < 0 || i >= A.height || j >= A.width) return false;
    const int64_t p = (int64_t)i * A.width + j;
    const double la = A.lat_c[p];
    double lo = A.lon_c[p];
    if (!(la == la) || !(lo == lo)) return false;
    if (A.center_mask != nullptr && A.center_mask[p]) return false;
    if (A.elev != nullptr && !(A.elev[p] >= A.min_elev)) return false;
    if (A.lon_wrap) lo = wrap180_shifted(lo);
    *x = la, *y = lo;
    return true;
}

// barycentric coordinates of (px, py) in triangle (x0,y0),(x1,y1),(x2,y2); false for a degenerate triangle
__device__ __forceinline__ bool lin_bary(double x0, double y0, double x1, double y1, double x2, double y2, double px, double py,
                                         double* w) {
    const double d = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    if (!(fabs(d) > 0)) return false;
    w[1] = ((px - x0) * (y2 - y0) - (x2 - x0) * (py - y0)) / d;
    w[2] = ((x1 - x0) * (py - y0) - (px - x0) * (y1 - y0)) / d;
    w[0] = 1.0 - w[1] - w[2];
    return true;
}

__device__ __forceinline__ bool lin_inside(const double* w) {
    constexpr double eps = 1e-12;
    return w[0] >= -eps && w[1] >= -eps && w[2] >= -eps;
}

// is D strictly inside the circumcircle of A, B, C (any orientation)?
__device__ __forceinline__ bool lin_incircle(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
    const double adx = ax - dx, ady = ay - dy, bdx = bx - dx, bdy = by - dy, cdx = cx - dx, cdy = cy - dy;
    const double ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    const double det = adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
    const double orient = (bx - ax) * (cy - ay) - (cx - ax) * (by - ay);
    return orient > 0 ? det > 0 : det < 0;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_linear_gather(lin_args A, const long long* __restrict__ index, int ny, int nx,
                                                          const double* __restrict__ tlat, const double* __restrict__ tlon,
                                                          const T* __restrict__ img, double* __restrict__ mean,
                                                          T* __restrict__ out_img, uint8_t* __restrict__ out_mask,
                                                          double* __restrict__ alt, long long* __restrict__ out_tri) {
    constexpr double kNaN = __builtin_nan("");
    const int nch = A.nchan;
    AMT_GRID_STRIDE(t, (int64_t)ny * nx) {
        const long long near = index[t];
        const int ty = (int)(t / nx), tx = (int)(t - (int64_t)ty * nx);
        const double px = tlat[ty], py = tlon[tx];
        long long tri[3] = {-1, -1, -1}, tri2[3] = {-1, -1, -1};
        double w[3] = {0, 0, 0}, w2[3] = {0, 0, 0};
        bool found = false, found2 = false;
        if (near >= 0) {
            const int pi = (int)(near / A.width), pj = (int)(near - (long long)pi * A.width);
            // The Delaunay triangulation of a smoothly mapped pixel grid joins neighbours of the REDUCED lattice basis, not
            // of the index grid: seen obliquely, a pixel's footprint is several times longer than wide in the (lat, lon)
            // plane, and the compact triangles join pixels like (i, j) and (i + 1, j + 3).  So: the local basis (one step
            // in j, one step in i) at the nearest centre, Gauss-reduced (integer combinations with the shortest vectors),
            // and the cells of THAT lattice around the grid centre, each cut along the diagonal the empty-circle test picks.
            double x0, y0, xa, ya, xb, yb;
            int uj = 1, ui = 0, vj = 0, vi = 1;                 // index steps of the two basis vectors
            bool have = lin_valid(A, pi, pj, &x0, &y0);
            double ux = 0, uy = 0, vx = 0, vy = 0;
            if (have) {
                if (lin_valid(A, pi, pj + 1, &xa, &ya)) ux = xa - x0, uy = ya - y0;
                else if (lin_valid(A, pi, pj - 1, &xa, &ya)) ux = x0 - xa, uy = y0 - ya;
                else have = false;
                if (lin_valid(A, pi + 1, pj, &xb, &yb)) vx = xb - x0, vy = yb - y0;
                else if (lin_valid(A, pi - 1, pj, &xb, &yb)) vx = x0 - xb, vy = y0 - yb;
                else have = false;
            }
            if (have) {
                for (int it = 0; it < 16; ++it) {
                    double uu = ux * ux + uy * uy, vv = vx * vx + vy * vy;
                    if (uu > vv) {
                        double tx_ = ux; ux = vx; vx = tx_;
                        double ty_ = uy; uy = vy; vy = ty_;
                        int tj = uj; uj = vj; vj = tj;
                        int ti = ui; ui = vi; vi = ti;
                        uu = vv;
                    }
                    if (!(uu > 0)) break;
                    const double m = rint((ux * vx + uy * vy) / uu);
                    if (m == 0) break;
                    vx -= m * ux, vy -= m * uy;
                    vj -= (int)m * uj, vi -= (int)m * ui;
                }
                const double det = ux * vy - uy * vx;
                have = fabs(det) > 0;
                if (have) {
                    // lattice coordinates of the grid centre relative to the nearest pixel centre
                    const double al = ((px - x0) * vy - (py - y0) * vx) / det, be = (ux * (py - y0) - uy * (px - x0)) / det;
                    const int a0 = (int)floor(al), b0 = (int)floor(be);
                    for (int ring = 0; ring < 2 && !found; ++ring)
                        for (int da = -ring; da <= ring && !found; ++da)
                            for (int db = -ring; db <= ring && !found; ++db) {
                                if (ring == 1 && da == 0 && db == 0) continue;
                                const int a = a0 + da, b = b0 + db;
                                double x[4], y[4];
                                // corners in cyclic order: (a, b), (a+1, b), (a+1, b+1), (a, b+1) of the reduced lattice
                                const int ca[4] = {a, a + 1, a + 1, a}, cb[4] = {b, b, b + 1, b + 1};
                                int ci[4], cj[4];
                                bool v[4];
                                int nv = 0;
                                for (int k = 0; k < 4; ++k) {
                                    ci[k] = pi + ca[k] * ui + cb[k] * vi;
                                    cj[k] = pj + ca[k] * uj + cb[k] * vj;
                                    v[k] = lin_valid(A, ci[k], cj[k], &x[k], &y[k]);
                                    nv += v[k] ? 1 : 0;
                                }
                                if (nv < 3) continue;
                                auto id = [&](int k) { return (long long)ci[k] * A.width + cj[k]; };
                                auto try_tri = [&](int k0, int k1, int k2, double* ww, long long* tt) {
                                    if (!lin_bary(x[k0], y[k0], x[k1], y[k1], x[k2], y[k2], px, py, ww) || !lin_inside(ww)) return false;
                                    tt[0] = id(k0), tt[1] = id(k1), tt[2] = id(k2);
                                    return true;
                                };
                                if (nv == 3) {
                                    int k[3], m = 0;
                                    for (int q = 0; q < 4; ++q)
                                        if (v[q]) k[m++] = q;
                                    found = try_tri(k[0], k[1], k[2], w, tri);
                                    continue;
                                }
                                // diagonal 0-2 unless corner 3 lies inside the circle through 0, 1, 2 (then 1-3)
                                const bool flip = lin_incircle(x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3]);
                                if (!flip) {
                                    found = try_tri(0, 1, 2, w, tri) || try_tri(0, 2, 3, w, tri);
                                    if (found) found2 = try_tri(0, 1, 3, w2, tri2) || try_tri(1, 2, 3, w2, tri2);
                                } else {
                                    found = try_tri(0, 1, 3, w, tri) || try_tri(1, 2, 3, w, tri);
                                    if (found) found2 = try_tri(0, 1, 2, w2, tri2) || try_tri(0, 2, 3, w2, tri2);
                                }
                            }
                }
            }
        }
        for (int c = 0; c <= nch; ++c) {
            double val = kNaN, val2 = kNaN;
            if (found) {
                val = 0;
                for (int k = 0; k < 3; ++k)
                    val += w[k] * (c < nch ? (double)img[tri[k] * nch + c] : (A.elev ? A.elev[tri[k]] : kNaN));
                val2 = val;
                if (found2) {
                    val2 = 0;
                    for (int k = 0; k < 3; ++k)
                        val2 += w2[k] * (c < nch ? (double)img[tri2[k] * nch + c] : (A.elev ? A.elev[tri2[k]] : kNaN));
                }
            }
            if (mean) mean[t * (nch + 1) + c] = val;
            if (alt) alt[t * (nch + 1) + c] = val2;
            if (c < nch && out_img) out_img[t * nch + c] = found ? (T)rint(val) : (T)0;      // np.round: half to even
        }
        if (out_mask) out_mask[t] = found ? 0 : 1;
        if (out_tri)
            for (int k = 0; k < 3; ++k) out_tri[t * 3 + k] = tri[k];
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

int amt_linear_gather(amt_ctx* ctx, const int64_t* index, int32_t ny, int32_t nx, const double* lat_c, const double* lon_c,
                      const double* elev, const uint8_t* center_mask, int32_t height, int32_t width, double min_elevation,
                      int lon_wrap, const double* target_lat, const double* target_lon, const void* img, int32_t img_dtype,
                      int32_t nchan, double* mean, void* out_img, uint8_t* out_mask, double* alt_mean, int64_t* out_triangles) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, index && lat_c && lon_c && target_lat && target_lon, "NULL argument");
    AMT_REQUIRE(ctx, ny >= 0 && nx >= 0 && height > 0 && width > 0, "bad size");
    AMT_REQUIRE(ctx, nchan >= 0 && nchan <= 4, "nchan must be 0..4");
    AMT_REQUIRE(ctx, nchan == 0 || (img && (img_dtype == 1 || img_dtype == 2)), "img must be uint8 (1) or uint16 (2)");
    const int64_t total = (int64_t)ny * nx;
    if (total == 0) return AMT_OK;
    lin_args A;
    A.lat_c = lat_c, A.lon_c = lon_c, A.elev = elev, A.center_mask = center_mask;
    A.height = height, A.width = width;
    A.min_elev = min_elevation;
    A.lon_wrap = lon_wrap ? 1 : 0;
    A.nchan = nchan;
    const long long* idx = reinterpret_cast<const long long*>(index);
    long long* tri = reinterpret_cast<long long*>(out_triangles);
    if (img_dtype == 2) {
        hipLaunchKernelGGL(k_linear_gather<uint16_t>, grid_for(total), dim3(kBlock), 0, ctx->stream, A, idx, ny, nx, target_lat,
                           target_lon, static_cast<const uint16_t*>(img), mean, static_cast<uint16_t*>(out_img), out_mask,
                           alt_mean, tri);
    } else {
        hipLaunchKernelGGL(k_linear_gather<uint8_t>, grid_for(total), dim3(kBlock), 0, ctx->stream, A, idx, ny, nx, target_lat,
                           target_lon, static_cast<const uint8_t*>(img), mean, static_cast<uint8_t*>(out_img), out_mask,
                           alt_mean, tri);
    }
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
