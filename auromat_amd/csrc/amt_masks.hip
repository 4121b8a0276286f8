// Mask rules of BaseMapping (reference auromat/mapping/mapping.py:299-316, 845-864, 1063-1125)
// as 2x2-neighbourhood stencils on uint8 masks (1 = masked), plus the corner bounding-box reduction.
#include "amt_common.h"

namespace {

using namespace amt;

constexpr int kBlock = 256;
constexpr double kInf = __builtin_huge_val();

inline dim3 grid_for(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return dim3(static_cast<unsigned>(blocks));
}

#define AMT_GRID_STRIDE(i, n) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void k_center_mask_by_elevation(const double* __restrict__ elev, int64_t n, double min_elev,
                                           uint8_t* __restrict__ center_mask, unsigned long long* n_valid) {
    unsigned long long local = 0;
    AMT_GRID_STRIDE(i, n) {
        const bool valid = elev[i] >= min_elev;   // NaN -> masked  ((elev < min).filled(True))
        center_mask[i] = valid ? 0 : 1;
        local += valid;
    }
    if (n_valid) {
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
        if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_valid, local);
    }
}

// corner masked |= all (existing) adjacent centres masked      (mapping.py:1080-1091)
template <bool FROM_NAN>
__global__ void k_corner_mask_neighbours(const uint8_t* __restrict__ center_mask, const double* __restrict__ corner_lat,
                                         int height, int width, uint8_t* __restrict__ corner_mask) {
    const int W1 = width + 1;
    const int64_t n = (int64_t)(height + 1) * W1;
    AMT_GRID_STRIDE(i, n) {
        const int r = (int)(i / W1), c = (int)(i - (int64_t)r * W1);
        bool any_valid = false;
        if (r > 0 && c > 0) any_valid |= center_mask[(int64_t)(r - 1) * width + (c - 1)] == 0;
        if (r > 0 && c < width) any_valid |= center_mask[(int64_t)(r - 1) * width + c] == 0;
        if (r < height && c > 0) any_valid |= center_mask[(int64_t)r * width + (c - 1)] == 0;
        if (r < height && c < width) any_valid |= center_mask[(int64_t)r * width + c] == 0;
        bool masked = !any_valid;
        if (FROM_NAN) {
            const double v = corner_lat[i];
            masked |= !(v == v);
        } else {
            masked |= corner_mask[i] != 0;
        }
        corner_mask[i] = masked ? 1 : 0;
    }
}

// centre masked |= img mask                                     (mapping.py:1075-1078)
__global__ void k_or_mask(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t n) {
    AMT_GRID_STRIDE(i, n) dst[i] = (dst[i] | src[i]) ? 1 : 0;
}

// centre masked |= any of its 4 corners masked                  (mapping.py:1093-1101)
__global__ void k_center_mask_corners(const uint8_t* __restrict__ corner_mask, int height, int width,
                                      uint8_t* __restrict__ center_mask) {
    const int W1 = width + 1;
    const int64_t n = (int64_t)height * width;
    AMT_GRID_STRIDE(i, n) {
        const int r = (int)(i / width), c = (int)(i - (int64_t)r * width);
        const int64_t k = (int64_t)r * W1 + c;
        const bool any = corner_mask[k] | corner_mask[k + 1] | corner_mask[k + W1] | corner_mask[k + W1 + 1];
        center_mask[i] = (center_mask[i] | any) ? 1 : 0;
    }
}

// v[0..5]: min/max slots (even = min, odd = max), v[6..7]: sums.  Result of the whole block in out[0..7].
__device__ __forceinline__ void block_reduce8(double (&v)[8], double* __restrict__ out) {
    __shared__ double sRed[8][kBlock / 64];
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
        for (int w = 1; w < kBlock / 64; ++w) {
            const double o = sRed[k][w];
            r = (k >= 6) ? r + o : ((k & 1) ? fmax(r, o) : fmin(r, o));
        }
        out[k] = r;
    }
}

__global__ __launch_bounds__(kBlock) void k_bbox_corners(const double* __restrict__ lat, const double* __restrict__ lon,
                                                          const uint8_t* __restrict__ corner_mask,
                                                          const uint8_t* __restrict__ center_mask, int height,
                                                          int width, double* __restrict__ partials) {
    const int W1 = width + 1;
    const int64_t nc = (int64_t)(height + 1) * W1, np = (int64_t)height * width;
    double v[8] = {kInf, -kInf, kInf, -kInf, kInf, -kInf, 0, 0};
    AMT_GRID_STRIDE(i, nc) {
        const double a = lat[i], o = lon[i];
        const bool ok = (a == a) && (corner_mask == nullptr || corner_mask[i] == 0);
        if (ok) {
            v[0] = fmin(v[0], a);
            v[1] = fmax(v[1], a);
            v[2] = fmin(v[2], o);
            v[3] = fmax(v[3], o);
            if (o > 0) v[4] = fmin(v[4], o); else v[5] = fmax(v[5], o);
            v[6] += 1;
        }
    }
    AMT_GRID_STRIDE(i, np) {
        const int r = (int)(i / width), c = (int)(i - (int64_t)r * width);
        const int64_t k = (int64_t)r * W1 + c;
        bool ok;
        if (center_mask) {
            ok = center_mask[i] == 0;
        } else if (corner_mask) {
            ok = !(corner_mask[k] | corner_mask[k + 1] | corner_mask[k + W1] | corner_mask[k + W1 + 1]);
        } else {
            ok = true;
        }
        const double o00 = lon[k], o01 = lon[k + 1], o10 = lon[k + W1], o11 = lon[k + W1 + 1];
        ok = ok && (o00 == o00) && (o01 == o01) && (o10 == o10) && (o11 == o11);
        if (ok && quad_winds_pole(o00, o01, o11, o10)) v[7] += 1;
    }
    block_reduce8(v, partials + (int64_t)blockIdx.x * 8);
}

__global__ __launch_bounds__(kBlock) void k_bbox_fold_small(const double* __restrict__ partials, int n,
                                                             double* __restrict__ bbox) {
    double v[8] = {kInf, -kInf, kInf, -kInf, kInf, -kInf, 0, 0};
    for (int i = threadIdx.x; i < n; i += kBlock) {
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
    block_reduce8(v, bbox);
}

// Contour links of a mask (reference utils.py:97-151: skimage find_contours at level 0.99 on the padded mask, rounded
// to pixel indices).  A "crossing" is a valid pixel p together with a direction d (0 up, 1 right, 2 down, 3 left) whose
// 4-neighbour is masked or outside; the contour visits every crossing exactly once, with the valid region on its
// right-hand side (clockwise in image coordinates, valid pixels joined through edges only, as marching squares joins the
// values above the level).  One thread per pixel writes (key, next key) for each of its crossings, key = 4*index + d.
__global__ void k_mask_outline_links(const uint8_t* __restrict__ mask, int height, int width,
                                     long long* __restrict__ links, long long capacity,
                                     unsigned long long* __restrict__ count) {
    const int64_t n = (int64_t)height * width;
    auto valid = [&](int r, int c) -> bool {
        return r >= 0 && c >= 0 && r < height && c < width && mask[(int64_t)r * width + c] == 0;
    };
    AMT_GRID_STRIDE(i, n) {
        if (mask[i] != 0) continue;
        const int r = (int)(i / width), c = (int)(i - (int64_t)r * width);
        const int dr[4] = {-1, 0, 1, 0}, dc[4] = {0, 1, 0, -1};
        for (int d = 0; d < 4; ++d) {
            if (valid(r + dr[d], c + dc[d])) continue;
            const int f = (d + 1) & 3;
            const int fr = r + dr[f], fc = c + dc[f];          // the pixel ahead along the contour
            long long next;
            if (!valid(fr, fc)) {
                next = 4 * i + f;                                // around this pixel's corner
            } else if (!valid(fr + dr[d], fc + dc[d])) {
                next = 4 * ((int64_t)fr * width + fc) + d;       // straight on
            } else {
                next = 4 * ((int64_t)(fr + dr[d]) * width + (fc + dc[d])) + ((d + 3) & 3);   // inner corner
            }
            const unsigned long long slot = atomicAdd(count, 1ull);
            if ((long long)slot < capacity) {
                links[2 * slot] = 4 * i + d;
                links[2 * slot + 1] = next;
            }
        }
    }
}

// Pixel polygons for drawing (reference draw_helpers.py:34-94): the four corners (lat, lon) of every listed pixel in the
// order (r,c), (r,c+1), (r+1,c+1), (r+1,c) and its colour (DefaultRGBMixin: uint16 * (255/65535) truncated to uint8).
template <typename T>
__global__ void k_pixel_polygons(const double* __restrict__ lat, const double* __restrict__ lon, const T* __restrict__ img,
                                 int nchan, int width, const long long* __restrict__ index, int64_t n,
                                 double* __restrict__ verts, uint8_t* __restrict__ colors_u8,
                                 double* __restrict__ colors_f64) {
    const int W1 = width + 1;
    AMT_GRID_STRIDE(k, n) {
        const long long i = index[k];
        const int r = (int)(i / width), c = (int)(i - (long long)r * width);
        const int64_t q = (int64_t)r * W1 + c;
        double* v = verts + 8 * k;
        v[0] = lat[q];
        v[1] = lon[q];
        v[2] = lat[q + 1];
        v[3] = lon[q + 1];
        v[4] = lat[q + W1 + 1];
        v[5] = lon[q + W1 + 1];
        v[6] = lat[q + W1];
        v[7] = lon[q + W1];
        for (int ch = 0; ch < 3; ++ch) {
            const T raw = img[i * nchan + (nchan == 1 ? 0 : ch)];
            const uint8_t u = sizeof(T) == 1 ? (uint8_t)raw : (uint8_t)((double)raw * (255.0 / 65535.0));
            if (colors_u8) colors_u8[3 * k + ch] = u;
            if (colors_f64) colors_f64[3 * k + ch] = (double)u / 255.0;
        }
    }
}

}  // namespace

extern "C" {

int amt_mask_by_elevation(amt_ctx* ctx, const double* elev, const double* corner_lat, int32_t height,
                          int32_t width, double min_elevation, uint8_t* center_mask, uint8_t* corner_mask,
                          int64_t* n_valid) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, elev && center_mask, "NULL argument");
    AMT_REQUIRE(ctx, (corner_mask == nullptr) == (corner_lat == nullptr), "corner_lat and corner_mask go together");
    AMT_REQUIRE(ctx, height > 0 && width > 0, "empty frame");
    const int64_t n = (int64_t)height * width;
    if (n_valid) AMT_HIP(ctx, hipMemsetAsync(n_valid, 0, sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL(k_center_mask_by_elevation, grid_for(n), dim3(kBlock), 0, ctx->stream, elev, n, min_elevation,
                       center_mask, reinterpret_cast<unsigned long long*>(n_valid));
    AMT_LAUNCH_CHECK(ctx);
    if (corner_mask) {
        const int64_t nc = (int64_t)(height + 1) * (width + 1);
        hipLaunchKernelGGL((k_corner_mask_neighbours<true>), grid_for(nc), dim3(kBlock), 0, ctx->stream, center_mask,
                           corner_lat, height, width, corner_mask);
        AMT_LAUNCH_CHECK(ctx);
    }
    return AMT_OK;
}

int amt_sanitize_masks(amt_ctx* ctx, uint8_t* corner_mask, uint8_t* center_mask, const uint8_t* img_mask,
                       int32_t height, int32_t width, int after_masking) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, corner_mask && center_mask, "NULL argument");
    AMT_REQUIRE(ctx, height > 0 && width > 0, "empty frame");
    const int64_t n = (int64_t)height * width, nc = (int64_t)(height + 1) * (width + 1);
    if (img_mask) {
        hipLaunchKernelGGL(k_or_mask, grid_for(n), dim3(kBlock), 0, ctx->stream, center_mask, img_mask, n);
        AMT_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL((k_corner_mask_neighbours<false>), grid_for(nc), dim3(kBlock), 0, ctx->stream, center_mask,
                       (const double*)nullptr, height, width, corner_mask);
    AMT_LAUNCH_CHECK(ctx);
    if (!after_masking) {
        hipLaunchKernelGGL(k_center_mask_corners, grid_for(n), dim3(kBlock), 0, ctx->stream, corner_mask, height, width,
                           center_mask);
        AMT_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((k_corner_mask_neighbours<false>), grid_for(nc), dim3(kBlock), 0, ctx->stream, center_mask,
                           (const double*)nullptr, height, width, corner_mask);
        AMT_LAUNCH_CHECK(ctx);
    }
    return AMT_OK;
}

int amt_bbox_corners(amt_ctx* ctx, const double* lat, const double* lon, const uint8_t* corner_mask,
                     const uint8_t* center_mask, int32_t height, int32_t width, double* bbox) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, lat && lon && bbox, "NULL argument");
    AMT_REQUIRE(ctx, height > 0 && width > 0, "empty grid");
    const dim3 grid = grid_for((int64_t)(height + 1) * (width + 1));
    double* partials = static_cast<double*>(amt_workspace(ctx, (size_t)grid.x * 8 * sizeof(double)));
    if (partials == nullptr) {
        ctx->last_error = "amt_bbox_corners: workspace allocation failed";
        return AMT_ENOMEM;
    }
    hipLaunchKernelGGL(k_bbox_corners, grid, dim3(kBlock), 0, ctx->stream, lat, lon, corner_mask, center_mask, height,
                       width, partials);
    AMT_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_bbox_fold_small, dim3(1), dim3(kBlock), 0, ctx->stream, partials, (int)grid.x, bbox);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_mask_outline_links(amt_ctx* ctx, const uint8_t* mask, int32_t height, int32_t width, int64_t* links,
                           int64_t capacity, uint64_t* count) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, mask && count && (capacity == 0 || links), "NULL argument");
    AMT_REQUIRE(ctx, height > 0 && width > 0 && capacity >= 0, "bad size");
    if (hipMemsetAsync(count, 0, sizeof(uint64_t), ctx->stream) != hipSuccess) {
        ctx->last_error = "amt_mask_outline_links: memset failed";
        return AMT_EHIP;
    }
    hipLaunchKernelGGL(k_mask_outline_links, grid_for((int64_t)height * width), dim3(kBlock), 0, ctx->stream, mask,
                       height, width, reinterpret_cast<long long*>(links), (long long)capacity,
                       reinterpret_cast<unsigned long long*>(count));
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_pixel_polygons(amt_ctx* ctx, const double* lat, const double* lon, const void* img, int32_t img_dtype,
                       int32_t nchan, int32_t height, int32_t width, const int64_t* index, int64_t n, double* verts,
                       uint8_t* colors_u8, double* colors_f64) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, lat && lon && img && (n == 0 || (index && verts)), "NULL argument");
    AMT_REQUIRE(ctx, height > 0 && width > 0 && n >= 0, "bad size");
    AMT_REQUIRE(ctx, (nchan == 1 || nchan == 3) && (img_dtype == 1 || img_dtype == 2), "image must be 1 or 3 channels of uint8 (1) / uint16 (2)");
    if (n == 0) return AMT_OK;
    const long long* idx = reinterpret_cast<const long long*>(index);
    if (img_dtype == 2) {
        hipLaunchKernelGGL(k_pixel_polygons<uint16_t>, grid_for(n), dim3(kBlock), 0, ctx->stream, lat, lon,
                           static_cast<const uint16_t*>(img), nchan, width, idx, n, verts, colors_u8, colors_f64);
    } else {
        hipLaunchKernelGGL(k_pixel_polygons<uint8_t>, grid_for(n), dim3(kBlock), 0, ctx->stream, lat, lon,
                           static_cast<const uint8_t*>(img), nchan, width, idx, n, verts, colors_u8, colors_f64);
    }
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

}  // extern "C"
