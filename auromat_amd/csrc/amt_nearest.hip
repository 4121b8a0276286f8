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

// ---- method='cubic' (reference resample.py:323-326: scipy griddata(method='cubic') = CloughTocher2DInterpolator) --------
// The piecewise cubic, C1 Clough-Tocher interpolant on the same triangulation as 'linear', with the vertex gradients of
// scipy's global estimator: the gradient at every data point minimises the curvature energy of the cubic Hermite curves
// along its triangulation edges (Nielson 1983; Renka & Cline 1984), a 2 x 2 system per point coupled to the neighbours'
// gradients, which scipy relaxes point after point until the largest relative change drops below 1e-6.  Here: one Jacobi
// sweep per launch over all valid pixels (the 2 x 2 blocks dominate the coupling 2 : 1, so Jacobi converges too), the
// neighbours of a pixel taken from the reduced lattice around it — the four axis neighbours and, for each of the four
// cells that meet at the pixel, the partner on the cell's Delaunay diagonal when that diagonal passes through the pixel.
// The element itself (12 boundary + 7 interior Bezier ordinates per triangle; the three free parameters fixed by making the
// derivative towards the neighbouring triangle's centroid linear along each edge, which is what makes scipy's element
// affine invariant) is restated from scipy 1.15's interpnd module and checked against it to rounding on scipy's own
// triangulation (oracle/ref_numpy.py: clough_tocher_*, tests/test_oracle_golden.py).

struct lat_basis {
    double x0, y0, ux, uy, vx, vy;
    int uj, ui, vj, vi;          // index steps of the two reduced basis vectors
};

// the local lattice basis at pixel (pi, pj) — one step in j, one step in i —, Gauss-reduced (see k_linear_gather)
__device__ bool lattice_basis(const lin_args& A, int pi, int pj, lat_basis* B) {
    double xa, ya, xb, yb;
    if (!lin_valid(A, pi, pj, &B->x0, &B->y0)) return false;
    double ux, uy, vx, vy;
    if (lin_valid(A, pi, pj + 1, &xa, &ya)) ux = xa - B->x0, uy = ya - B->y0;
    else if (lin_valid(A, pi, pj - 1, &xa, &ya)) ux = B->x0 - xa, uy = B->y0 - ya;
    else return false;
    if (lin_valid(A, pi + 1, pj, &xb, &yb)) vx = xb - B->x0, vy = yb - B->y0;
    else if (lin_valid(A, pi - 1, pj, &xb, &yb)) vx = B->x0 - xb, vy = B->y0 - yb;
    else return false;
    int uj = 1, ui = 0, vj = 0, vi = 1;
    for (int it = 0; it < 16; ++it) {
        double uu = ux * ux + uy * uy, vv = vx * vx + vy * vy;
        if (uu > vv) {
            double t = ux; ux = vx; vx = t;
            t = uy; uy = vy; vy = t;
            int k = uj; uj = vj; vj = k;
            k = ui; ui = vi; vi = k;
            uu = vv;
        }
        if (!(uu > 0)) break;
        const double m = rint((ux * vx + uy * vy) / uu);
        if (m == 0) break;
        vx -= m * ux, vy -= m * uy;
        vj -= (int)m * uj, vi -= (int)m * ui;
    }
    if (!(fabs(ux * vy - uy * vx) > 0)) return false;
    B->ux = ux, B->uy = uy, B->vx = vx, B->vy = vy;
    B->uj = uj, B->ui = ui, B->vj = vj, B->vi = vi;
    return true;
}

// cell (a, b) of the reduced lattice around pixel (pi, pj): corners in cyclic order (a, b), (a+1, b), (a+1, b+1), (a, b+1);
// flip: cut along 1-3 instead of 0-2 (corner 3 inside the circle through 0, 1, 2), defined for 4 valid corners
struct lat_cell {
    int ci[4], cj[4];
    double x[4], y[4];
    bool v[4];
    int nv;
    bool flip;
};

__device__ void lattice_cell(const lin_args& A, int pi, int pj, const lat_basis& B, int a, int b, lat_cell* c) {
    const int ca[4] = {a, a + 1, a + 1, a}, cb[4] = {b, b, b + 1, b + 1};
    c->nv = 0;
    for (int k = 0; k < 4; ++k) {
        c->ci[k] = pi + ca[k] * B.ui + cb[k] * B.vi;
        c->cj[k] = pj + ca[k] * B.uj + cb[k] * B.vj;
        c->x[k] = c->y[k] = 0;
        c->v[k] = lin_valid(A, c->ci[k], c->cj[k], &c->x[k], &c->y[k]);
        c->nv += c->v[k] ? 1 : 0;
    }
    c->flip = c->nv == 4 && lin_incircle(c->x[0], c->y[0], c->x[1], c->y[1], c->x[2], c->y[2], c->x[3], c->y[3]);
}

// the corner of cell c that completes the triangle on the cell's side s (corners s, s+1); -1: the cell has no such triangle
__device__ int lattice_third(const lat_cell& c, int s) {
    if (c.nv == 4) {
        const int plain[4] = {2, 0, 0, 2}, flipped[4] = {3, 3, 1, 1};
        return c.flip ? flipped[s] : plain[s];
    }
    if (c.nv == 3 && c.v[s] && c.v[(s + 1) & 3]) return c.v[(s + 2) & 3] ? (s + 2) & 3 : (s + 3) & 3;
    return -1;
}

constexpr int kCubicNb = 8;

// neighbours of every pixel in the triangulation: flat pixel indices, -1 = none.  Slots: +u, -u, +v, -v, +u+v, -u-v, -u+v, +u-v
__global__ __launch_bounds__(kBlock) void k_cubic_neighbours(lin_args A, int* __restrict__ nb) {
    AMT_GRID_STRIDE(p, (int64_t)A.height * A.width) {
        int out[kCubicNb];
        for (int k = 0; k < kCubicNb; ++k) out[k] = -1;
        const int i = (int)(p / A.width), j = (int)(p - (int64_t)i * A.width);
        lat_basis B;
        if (lattice_basis(A, i, j, &B)) {
            lat_cell c00, cm0, cmm, c0m;
            lattice_cell(A, i, j, B, 0, 0, &c00);        // the pixel is its corner 0
            lattice_cell(A, i, j, B, -1, 0, &cm0);       // corner 1
            lattice_cell(A, i, j, B, -1, -1, &cmm);      // corner 2
            lattice_cell(A, i, j, B, 0, -1, &c0m);       // corner 3
            auto id = [&](const lat_cell& c, int k) { return c.ci[k] * A.width + c.cj[k]; };
            if (c00.v[1] && (c00.nv >= 3 || c0m.nv >= 3)) out[0] = id(c00, 1);
            if (cm0.v[0] && (cm0.nv >= 3 || cmm.nv >= 3)) out[1] = id(cm0, 0);
            if (c00.v[3] && (c00.nv >= 3 || cm0.nv >= 3)) out[2] = id(c00, 3);
            if (c0m.v[0] && (c0m.nv >= 3 || cmm.nv >= 3)) out[3] = id(c0m, 0);
            if (c00.v[2] && ((c00.nv == 4 && !c00.flip) || c00.nv == 3)) out[4] = id(c00, 2);
            if (cmm.v[0] && ((cmm.nv == 4 && !cmm.flip) || cmm.nv == 3)) out[5] = id(cmm, 0);
            if (cm0.v[3] && ((cm0.nv == 4 && cm0.flip) || cm0.nv == 3)) out[6] = id(cm0, 3);
            if (c0m.v[1] && ((c0m.nv == 4 && c0m.flip) || c0m.nv == 3)) out[7] = id(c0m, 1);
        }
        for (int k = 0; k < kCubicNb; ++k) nb[p * kCubicNb + k] = out[k];
    }
}

constexpr int kCubicMaxChan = 5;      // image channels + elevation

template <typename T>
__device__ __forceinline__ double cubic_value(const lin_args& A, const T* img, int64_t p, int c) {
    return c < A.nchan ? (double)img[p * A.nchan + c] : A.elev[p];
}

// one Jacobi sweep of the gradient estimator (scipy interpnd: _estimate_gradients_2d_global): y_out from y_in; the largest
// change, relative as scipy measures it, goes to *err (bit pattern of a non-negative double: ordered like an integer)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_cubic_sweep(lin_args A, const int* __restrict__ nb, const T* __restrict__ img,
                                                        const double* __restrict__ y_in, double* __restrict__ y_out,
                                                        unsigned long long* __restrict__ err) {
    const int nc = A.nchan + (A.elev != nullptr ? 1 : 0);
    double worst = 0;
    AMT_GRID_STRIDE(p, (int64_t)A.height * A.width) {
        const int i = (int)(p / A.width), j = (int)(p - (int64_t)i * A.width);
        double x1, y1;
        double s0[kCubicMaxChan], s1[kCubicMaxChan], f1[kCubicMaxChan];
        double q0 = 0, q1 = 0, q3 = 0;
        for (int c = 0; c < nc; ++c) s0[c] = s1[c] = 0;
        int used = 0;
        if (lin_valid(A, i, j, &x1, &y1)) {
            for (int c = 0; c < nc; ++c) f1[c] = cubic_value(A, img, p, c);
            for (int k = 0; k < kCubicNb; ++k) {
                const int q = nb[p * kCubicNb + k];
                if (q < 0) continue;
                double x2, y2;
                const int qi = q / A.width, qj = q - qi * A.width;
                if (!lin_valid(A, qi, qj, &x2, &y2)) continue;
                ++used;
                const double ex = x2 - x1, ey = y2 - y1;
                const double l2 = ex * ex + ey * ey, l3 = l2 * sqrt(l2);
                q0 += 4 * ex * ex / l3;
                q1 += 4 * ex * ey / l3;
                q3 += 4 * ey * ey / l3;
                for (int c = 0; c < nc; ++c) {
                    const double f2 = cubic_value(A, img, (int64_t)q, c);
                    const double* g2 = y_in + ((int64_t)q * nc + c) * 2;
                    const double df2 = -ex * g2[0] - ey * g2[1];
                    const double t = (6 * (f1[c] - f2) - 2 * df2) / l3;
                    s0[c] += t * ex;
                    s1[c] += t * ey;
                }
            }
        }
        const double det = q0 * q3 - q1 * q1;
        for (int c = 0; c < nc; ++c) {
            double* g = y_out + (p * nc + c) * 2;
            if (used < 2 || !(fabs(det) > 0)) {
                g[0] = g[1] = 0;
                continue;
            }
            const double r0 = (q3 * s0[c] - q1 * s1[c]) / det, r1 = (-q1 * s0[c] + q0 * s1[c]) / det;
            const double* old = y_in + (p * nc + c) * 2;
            double change = fmax(fabs(old[0] + r0), fabs(old[1] + r1));
            change /= fmax(1.0, fmax(fabs(r0), fabs(r1)));
            if (change == change) worst = fmax(worst, change);
            g[0] = -r0, g[1] = -r1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o));
    if ((threadIdx.x & 63) == 0 && worst > 0) atomicMax(err, (unsigned long long)__double_as_longlong(worst));
}

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

// The triangle (corners k[0..2] of the cell `c` at lattice position (a, b) around pixel (pi, pj)) with everything the element
// needs: for each edge the neighbouring triangle's centroid — across the cell's diagonal the cell's other triangle, across
// a side the triangle of the adjacent cell on that side
template <typename T>
__device__ void cubic_in_triangle(const lin_args& A, int pi, int pj, const lat_basis& B, int a, int b, const lat_cell& c,
                                  const int* k, const double* w, const T* img, const double* grad, double* out) {
    double x[3], y[3], cx[3] = {0, 0, 0}, cy[3] = {0, 0, 0};
    bool has[3];
    int64_t pix[3];
    for (int m = 0; m < 3; ++m) {
        x[m] = c.x[k[m]], y[m] = c.y[k[m]];
        pix[m] = (int64_t)c.ci[k[m]] * A.width + c.cj[k[m]];
    }
    for (int m = 0; m < 3; ++m) {
        // edge opposite vertex m: corners p, q
        const int p = k[(m + 1) % 3], q = k[(m + 2) % 3];
        has[m] = false;
        double tx = 0, ty = 0;
        if (((p - q) & 3) == 2) {
            // the cell's diagonal: the other triangle is the remaining corner's
            if (c.nv == 4) {
                const int r = 6 - k[0] - k[1] - k[2];
                tx = c.x[r], ty = c.y[r];
                has[m] = true;
            }
        } else {
            const int s = ((p + 1) & 3) == q ? p : q;                 // side s: corners s, s+1
            const int da[4] = {0, 1, 0, -1}, db[4] = {-1, 0, 1, 0};
            lat_cell nbc;
            lattice_cell(A, pi, pj, B, a + da[s], b + db[s], &nbc);
            const int r = lattice_third(nbc, (s + 2) & 3);
            if (r >= 0) {
                tx = nbc.x[r], ty = nbc.y[r];
                has[m] = true;
            }
        }
        if (has[m]) {
            cx[m] = (c.x[p] + c.x[q] + tx) / 3;
            cy[m] = (c.y[p] + c.y[q] + ty) / 3;
        }
    }
    const int nc = A.nchan + (A.elev != nullptr ? 1 : 0);
    for (int ch = 0; ch < nc; ++ch) {
        double f[3], d[3][2];
        for (int m = 0; m < 3; ++m) {
            f[m] = cubic_value(A, img, pix[m], ch);
            d[m][0] = grad[(pix[m] * nc + ch) * 2];
            d[m][1] = grad[(pix[m] * nc + ch) * 2 + 1];
        }
        out[ch] = clough_tocher(x, y, w, f, d, has, cx, cy);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_cubic_gather(lin_args A, const long long* __restrict__ index, int ny, int nx,
                                                         const double* __restrict__ tlat, const double* __restrict__ tlon,
                                                         const T* __restrict__ img, const double* __restrict__ grad,
                                                         double* __restrict__ mean, T* __restrict__ out_img,
                                                         uint8_t* __restrict__ out_mask, double* __restrict__ alt,
                                                         long long* __restrict__ out_tri) {
    constexpr double kNaN = __builtin_nan("");
    const int nch = A.nchan, nc = nch + (A.elev != nullptr ? 1 : 0);
    AMT_GRID_STRIDE(t, (int64_t)ny * nx) {
        const long long near = index[t];
        const int ty = (int)(t / nx), tx = (int)(t - (int64_t)ty * nx);
        const double px = tlat[ty], py = tlon[tx];
        double val[kCubicMaxChan], val2[kCubicMaxChan];
        long long tri[3] = {-1, -1, -1};
        bool found = false, found2 = false;
        if (near >= 0) {
            const int pi = (int)(near / A.width), pj = (int)(near - (long long)pi * A.width);
            lat_basis B;
            if (lattice_basis(A, pi, pj, &B)) {
                const double det = B.ux * B.vy - B.uy * B.vx;
                const double al = ((px - B.x0) * B.vy - (py - B.y0) * B.vx) / det, be = (B.ux * (py - B.y0) - B.uy * (px - B.x0)) / det;
                const int a0 = (int)floor(al), b0 = (int)floor(be);
                for (int ring = 0; ring < 2 && !found; ++ring)
                    for (int da = -ring; da <= ring && !found; ++da)
                        for (int db = -ring; db <= ring && !found; ++db) {
                            if (ring == 1 && da == 0 && db == 0) continue;
                            const int a = a0 + da, b = b0 + db;
                            lat_cell c;
                            lattice_cell(A, pi, pj, B, a, b, &c);
                            if (c.nv < 3) continue;
                            int k[3], k2[3];
                            double w[3], w2[3];
                            auto try_tri = [&](int k0, int k1, int k2_, int* kk, double* ww) {
                                if (!lin_bary(c.x[k0], c.y[k0], c.x[k1], c.y[k1], c.x[k2_], c.y[k2_], px, py, ww) || !lin_inside(ww)) return false;
                                kk[0] = k0, kk[1] = k1, kk[2] = k2_;
                                return true;
                            };
                            if (c.nv == 3) {
                                int q[3], m = 0;
                                for (int r = 0; r < 4; ++r)
                                    if (c.v[r]) q[m++] = r;
                                found = try_tri(q[0], q[1], q[2], k, w);
                            } else if (!c.flip) {
                                found = try_tri(0, 1, 2, k, w) || try_tri(0, 2, 3, k, w);
                                if (found) found2 = try_tri(0, 1, 3, k2, w2) || try_tri(1, 2, 3, k2, w2);
                            } else {
                                found = try_tri(0, 1, 3, k, w) || try_tri(1, 2, 3, k, w);
                                if (found) found2 = try_tri(0, 1, 2, k2, w2) || try_tri(0, 2, 3, k2, w2);
                            }
                            if (!found) continue;
                            for (int m = 0; m < 3; ++m) tri[m] = (long long)c.ci[k[m]] * A.width + c.cj[k[m]];
                            cubic_in_triangle(A, pi, pj, B, a, b, c, k, w, img, grad, val);
                            if (found2 && alt != nullptr) {
                                // the cell cut along its other diagonal (what Qhull may have chosen for a near-cocircular quad)
                                lat_cell c2 = c;
                                c2.flip = !c.flip;
                                cubic_in_triangle(A, pi, pj, B, a, b, c2, k2, w2, img, grad, val2);
                            }
                        }
            }
        }
        for (int ch = 0; ch < nc; ++ch) {
            const double v = found ? val[ch] : kNaN;
            if (mean) mean[t * nc + ch] = v;
            if (alt) alt[t * nc + ch] = found ? (found2 ? val2[ch] : val[ch]) : kNaN;
            if (sizeof(T) < 8 && ch < nch && out_img) {
                // np.round (half to even), then the cast of the reference (resample.py:129-132); a cubic overshoots, and
                // numpy's cast of an out-of-range float wraps modulo the type's range on x86-64: the same here
                const double r = found ? rint(v) : 0.0;
                out_img[t * nch + ch] = (T)(unsigned long long)(long long)r;
            }
        }
        if (out_mask) out_mask[t] = found ? 0 : 1;
        if (out_tri)
            for (int m = 0; m < 3; ++m) out_tri[t * 3 + m] = tri[m];
    }
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
// right-hand side, and the eight terms of a channel are then added in the order of the neighbours (lane to lane, ds_bpermute), so
// that the sum has scipy's rounding.  (With lanes as channels only — the first form — four of 64 lanes worked for an RGB image
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

int amt_cubic_gradients(amt_ctx* ctx, const double* lat_c, const double* lon_c, const double* elev, const uint8_t* center_mask,
                        int32_t height, int32_t width, double min_elevation, int lon_wrap, const void* img, int32_t img_dtype,
                        int32_t nchan, double tolerance, int32_t max_iterations, double* gradients, int32_t* iterations) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, lat_c && lon_c && gradients, "NULL argument");
    AMT_REQUIRE(ctx, height > 0 && width > 0 && (int64_t)height * width < 2147483647LL, "bad size");
    AMT_REQUIRE(ctx, nchan >= 0 && nchan + (elev ? 1 : 0) <= kCubicMaxChan && nchan + (elev ? 1 : 0) >= 1,
                "1..5 channels (image channels + elevation)");
    AMT_REQUIRE(ctx, nchan == 0 || (img && img_dtype >= 1 && img_dtype <= 3), "img must be uint8 (1), uint16 (2) or float64 (3)");
    AMT_REQUIRE(ctx, tolerance > 0 && max_iterations >= 1, "tolerance and max_iterations must be positive");
    lin_args A;
    A.lat_c = lat_c, A.lon_c = lon_c, A.elev = elev, A.center_mask = center_mask;
    A.height = height, A.width = width;
    A.min_elev = min_elevation;
    A.lon_wrap = lon_wrap ? 1 : 0;
    A.nchan = nchan;
    const int64_t n = (int64_t)height * width;
    const size_t grad_bytes = (size_t)n * (nchan + (elev ? 1 : 0)) * 2 * sizeof(double);
    const size_t nb_bytes = (size_t)n * kCubicNb * sizeof(int);
    char* ws = static_cast<char*>(amt_workspace(ctx, grad_bytes + nb_bytes + 64));
    if (ws == nullptr) {
        ctx->last_error = "amt_cubic_gradients: workspace allocation failed";
        return AMT_ENOMEM;
    }
    double* other = reinterpret_cast<double*>(ws);
    int* nb = reinterpret_cast<int*>(ws + grad_bytes);
    unsigned long long* err = reinterpret_cast<unsigned long long*>(ws + grad_bytes + nb_bytes);
    if (hipMemsetAsync(gradients, 0, grad_bytes, ctx->stream) != hipSuccess) {
        ctx->last_error = "amt_cubic_gradients: memset failed";
        return AMT_EHIP;
    }
    hipLaunchKernelGGL(k_cubic_neighbours, grid_for(n), dim3(kBlock), 0, ctx->stream, A, nb);
    auto sweep = [&](const double* from, double* to) {
        if (img_dtype == 3)
            hipLaunchKernelGGL(k_cubic_sweep<double>, grid_for(n), dim3(kBlock), 0, ctx->stream, A, nb,
                               static_cast<const double*>(img), from, to, err);
        else if (img_dtype == 2)
            hipLaunchKernelGGL(k_cubic_sweep<uint16_t>, grid_for(n), dim3(kBlock), 0, ctx->stream, A, nb,
                               static_cast<const uint16_t*>(img), from, to, err);
        else
            hipLaunchKernelGGL(k_cubic_sweep<uint8_t>, grid_for(n), dim3(kBlock), 0, ctx->stream, A, nb,
                               static_cast<const uint8_t*>(img), from, to, err);
    };
    // sweeps in pairs (the result of a pair is in `gradients` again); the second one's largest change decides
    int done = 0;
    double worst = 0;
    while (done < max_iterations) {
        sweep(gradients, other);
        if (hipMemsetAsync(err, 0, sizeof(*err), ctx->stream) != hipSuccess) {
            ctx->last_error = "amt_cubic_gradients: memset failed";
            return AMT_EHIP;
        }
        sweep(other, gradients);
        done += 2;
        unsigned long long bits = 0;
        if (hipMemcpyAsync(&bits, err, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            ctx->last_error = "amt_cubic_gradients: reading the convergence flag failed";
            return AMT_EHIP;
        }
        std::memcpy(&worst, &bits, sizeof(worst));
        if (worst < tolerance) break;
    }
    if (iterations) *iterations = done;
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_cubic_gather(amt_ctx* ctx, const int64_t* index, int32_t ny, int32_t nx, const double* lat_c, const double* lon_c,
                     const double* elev, const uint8_t* center_mask, int32_t height, int32_t width, double min_elevation,
                     int lon_wrap, const double* target_lat, const double* target_lon, const void* img, int32_t img_dtype,
                     int32_t nchan, const double* gradients, double* mean, void* out_img, uint8_t* out_mask, double* alt_mean,
                     int64_t* out_triangles) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, index && lat_c && lon_c && target_lat && target_lon && gradients, "NULL argument");
    AMT_REQUIRE(ctx, ny >= 0 && nx >= 0 && height > 0 && width > 0, "bad size");
    AMT_REQUIRE(ctx, nchan >= 0 && nchan + (elev ? 1 : 0) <= kCubicMaxChan && nchan + (elev ? 1 : 0) >= 1,
                "1..5 channels (image channels + elevation)");
    AMT_REQUIRE(ctx, nchan == 0 || (img && img_dtype >= 1 && img_dtype <= 3), "img must be uint8 (1), uint16 (2) or float64 (3)");
    AMT_REQUIRE(ctx, img_dtype != 3 || out_img == nullptr, "out_img needs an integer image");
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
    if (img_dtype == 3) {
        hipLaunchKernelGGL(k_cubic_gather<double>, grid_for(total), dim3(kBlock), 0, ctx->stream, A, idx, ny, nx, target_lat,
                           target_lon, static_cast<const double*>(img), gradients, mean, static_cast<double*>(nullptr), out_mask,
                           alt_mean, tri);
    } else if (img_dtype == 2) {
        hipLaunchKernelGGL(k_cubic_gather<uint16_t>, grid_for(total), dim3(kBlock), 0, ctx->stream, A, idx, ny, nx, target_lat,
                           target_lon, static_cast<const uint16_t*>(img), gradients, mean, static_cast<uint16_t*>(out_img),
                           out_mask, alt_mean, tri);
    } else {
        hipLaunchKernelGGL(k_cubic_gather<uint8_t>, grid_for(total), dim3(kBlock), 0, ctx->stream, A, idx, ny, nx, target_lat,
                           target_lon, static_cast<const uint8_t*>(img), gradients, mean, static_cast<uint8_t*>(out_img),
                           out_mask, alt_mean, tri);
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
