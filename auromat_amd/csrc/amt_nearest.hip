// resample(method='nearest') and the outside-outline masking of the interpolating methods
// (reference auromat/resample.py:246-259,301-327; utils.py:58-74).
//
// Nearest neighbour of every grid centre among the valid pixel centres, Euclidean in the (lat, lon) plane in
// degrees, as scipy.interpolate.griddata(method='nearest') (a cKDTree query) defines it.  On the device the
// search structure is the output grid itself: a counting sort of the source pixels by the grid cell they fall
// into (count -> exclusive scan -> fill), then one thread per grid centre visits the cells ring by ring and
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

__global__ void k_nn_count(nn_args A, int* __restrict__ cell_of, unsigned* __restrict__ count) {
    AMT_GRID_STRIDE(i, A.n) {
        double x, y;
        int c = -1;
        if (source_xy(A, i, x, y)) {
            c = source_cell(A, x, y);
            atomicAdd(&count[c], 1u);
        }
        cell_of[i] = c;
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

__global__ void k_nn_fill(const int* __restrict__ cell_of, int64_t n, const unsigned* __restrict__ offset,
                          unsigned* __restrict__ cursor, int* __restrict__ order) {
    AMT_GRID_STRIDE(i, n) {
        const int c = cell_of[i];
        if (c < 0) continue;
        const unsigned slot = offset[c] + atomicAdd(&cursor[c], 1u);
        order[slot] = (int)i;
    }
}

__global__ void k_nn_search(nn_args A, const unsigned* __restrict__ offset, const int* __restrict__ order,
                            const double* __restrict__ target_lat, const double* __restrict__ target_lon,
                            const uint8_t* __restrict__ target_mask, double safe_step,
                            long long* __restrict__ out_index) {
    const int64_t total = (int64_t)A.nx * A.ny;
    const unsigned n_sources = offset[total];
    AMT_GRID_STRIDE(t, total) {
        const int row = (int)(t / A.nx), col = (int)(t - (int64_t)row * A.nx);
        if ((target_mask && target_mask[t]) || n_sources == 0) {
            out_index[t] = -1;
            continue;
        }
        const double ty = target_lat[row], tx = target_lon[col];
        const int cy = A.ny - 1 - row;                 // histogram rows ascend in latitude, output rows descend
        double best = __builtin_huge_val();
        int best_i = -1;
        auto visit = [&](int iy, int ix) {
            if (iy < 0 || iy >= A.ny || ix < 0 || ix >= A.nx) return;
            const int c = iy * A.nx + ix;
            for (unsigned k = offset[c], e = offset[c + 1]; k < e; ++k) {
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
                visit(cy, col);
            } else {
                for (int ix = col - r; ix <= col + r; ++ix) {
                    visit(cy - r, ix);
                    visit(cy + r, ix);
                }
                for (int iy = cy - r + 1; iy <= cy + r - 1; ++iy) {
                    visit(iy, col - r);
                    visit(iy, col + r);
                }
            }
            // every source closer than (r + 1/2) cells in both axes lies in the rings visited so far
            const double reach = ((double)r + 0.5) * safe_step;
            if (best_i >= 0 && best <= reach * reach) break;
        }
        out_index[t] = best_i;
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

// matplotlib.path.Path(polygon).contains_points(points) (reference utils.py:58-74): crossing test of a ray towards
// +x with the half-open edge rule (vertex y >= point y) of Agg's point_in_path; the path is closed implicitly.
__global__ void k_points_in_polygon(const double* __restrict__ px, const double* __restrict__ py, int64_t n,
                                    const double* __restrict__ poly, int m, uint8_t* __restrict__ inside) {
    __shared__ double sx[kBlock + 1], sy[kBlock + 1];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool live = i < n;
    const double tx = live ? px[i] : 0.0, ty = live ? py[i] : 0.0;
    bool in = false;
    for (int base = 0; base < m; base += kBlock) {
        const int cnt = min(kBlock, m - base);
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            sx[threadIdx.x] = poly[2 * (int64_t)(base + threadIdx.x)];
            sy[threadIdx.x] = poly[2 * (int64_t)(base + threadIdx.x) + 1];
        }
        if (threadIdx.x == 0) {                       // the vertex that closes this chunk's last edge
            const int nxt = (base + cnt) % m;
            sx[cnt] = poly[2 * (int64_t)nxt];
            sy[cnt] = poly[2 * (int64_t)nxt + 1];
        }
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const double x0 = sx[k], y0 = sy[k], x1 = sx[k + 1], y1 = sy[k + 1];
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
    hipLaunchKernelGGL(k_nn_search, grid_for(cells), dim3(kBlock), 0, ctx->stream, A, offset, order, target_lat,
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
