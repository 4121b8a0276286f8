// Single-pass frame driver (see include/auromat_hip.h, "grid layout and the single-pass frame driver").
// Host orchestration only: every kernel it launches lives in amt_georef.hip / amt_binning.hip.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <new>

#include "amt_common.h"
#include "amt_grid.h"

struct amt_pipe {
    amt_ctx* ctx;
    // The context's stream carries only the big kernel of each frame.  Everything small runs beside it:
    hipStream_t pre_stream;        // coarse pre-pass of a later frame (must not queue behind the running kernel)
    hipStream_t tail_stream;       // bounding-box folds of this frame, waiting for its big kernel
    hipStream_t fin_stream;        // crop/finalise kernel: waits for nothing on the GPU (the host has seen the box),
                                   // so it must not queue behind another frame's folds, which do wait
                                   // (all three owned by the context and shared with its other drivers)
    hipEvent_t coarse_done, kernel_done, bbox_done, tail_done;
    double* host_small;            // pinned host memory: [0..7] coarse bbox, [8..15] exact bbox
    double* host_small_dev;        // the same 16 doubles as the kernels address them (the folds write them directly)
    void* events;                  // on-edge pixels of the frame in flight (bin_event records, device)
    uint32_t* event_count;         // device counter; zero between frames
    long long n_events;            // what the frame reported (host copy)
    uint64_t* acc;                 // superset accumulators (device), 5 planes
    size_t acc_cells;              // capacity per plane
    bool acc_zero;                 // the accumulators are known to be all zero
    double* partials;              // per-wave partial boxes of the big kernel (read by the folds on tail_stream)
    size_t partials_bytes;
    // state of the frame in flight
    bool coarse_pending, launched, fused, ready, tail_pending;
    bool coarse_hinted;            // the pending coarse box came from the caller (amt_pipe_coarse_hint), no kernel ran
    int coarse_magnetic;           // coordinates of the pending coarse box (0 geodetic, 1 MLat / SM longitude,
                                   // 2 / 3 the same rotated by 90 deg about x: the pole plans)
    int lon_wrap;                  // the frame straddles the 180 deg discontinuity: longitudes are binned shifted by 180
    amt_grid super, exact;
    int32_t off_x, off_y;          // window of the exact grid inside the superset
    double lat_ppd, lon_ppd, min_elev;
    int pole;
    bool pole_unknown;             // direction arrays without a decision of the caller: see amt_pipe_launch_dirs
    bool pole_plan;                // this frame is binned in the coordinates rotated by 90 deg about x (bin_pole)
    int img_dtype;
    // the two-pass plan on the driver's streams (amt_pipe_general_layout / _finalize): the frame's coordinate arrays, image
    // and size as the big kernel was given them
    bool two_pass;                 // amt_pipe_set_plan: never fuse the binning into the big kernel
    bool general_ready;            // amt_pipe_general_layout has laid out the exact grid of the frame in flight
    const double* g_lat_c;
    int g_row_layout;              // amt_georef_out.row_layout of the frame in flight (strip-padded arrays: no general path here)
    const double* g_lon_c;
    const double* g_elev;
    const void* g_img;
    int g_width, g_height, g_fast, g_mode;      // g_mode: 0 geodetic grid without a pole in view (see pipe_prepare's mode)
};

namespace {

constexpr int kCoarseStride = 16;      // every 16th pixel corner at most; >= 128 lattice points on the short side
constexpr long long kEventCapacity = 16384;   // on-edge pixels per frame (a few dozen in practice)
constexpr double kMarginDeg = 1.0;     // safety margin around the coarse box (> 3 lattice steps on the ground)
constexpr double kPoleGuardDeg = 85.0; // direction arrays, pole unknown: boxes that reach beyond this latitude are not fused

int ensure_acc(amt_pipe* pipe, size_t cells) {
    if (cells <= pipe->acc_cells) return AMT_OK;
    amt_ctx* ctx = pipe->ctx;
    if (pipe->acc) {
        AMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        AMT_HIP(ctx, hipStreamSynchronize(pipe->tail_stream));
        AMT_HIP(ctx, hipFree(pipe->acc));
        pipe->acc = nullptr;
        pipe->acc_cells = 0;
    }
    const size_t cap = cells < (1u << 16) ? (1u << 16) : cells + cells / 4;
    AMT_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&pipe->acc), cap * 5 * sizeof(uint64_t)));
    pipe->acc_cells = cap;
    pipe->acc_zero = false;
    return AMT_OK;
}

int ensure_partials(amt_pipe* pipe, size_t bytes) {
    if (bytes <= pipe->partials_bytes) return AMT_OK;
    amt_ctx* ctx = pipe->ctx;
    if (pipe->partials) {
        AMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        AMT_HIP(ctx, hipStreamSynchronize(pipe->tail_stream));
        AMT_HIP(ctx, hipFree(pipe->partials));
        pipe->partials = nullptr;
        pipe->partials_bytes = 0;
    }
    AMT_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&pipe->partials), bytes));
    pipe->partials_bytes = bytes;
    return AMT_OK;
}

// astropy Angle.wrap_at(180 deg), as auromat_amd/mapping/mapping.py wrap_at_180 computes it (reference
// resample.py:212-218): into [-180, 180)
double wrap_at_180(double v) {
    const double wraps = std::floor((v + 180.0) / 360.0);
    double a = v - wraps * 360.0;
    if (a >= 180.0) a -= 360.0;
    if (a < -180.0) a += 360.0;
    return a;
}

// Is the north or south pole of the mapping shell imaged by a valid pixel?  The pole point is projected
// through the inverse TAN model; it counts when it falls inside the frame, is the first hit of its ray and
// lies above the elevation threshold (host mirror: auromat_amd/mapping/astrometry.py pole_in_view; replaces
// the outline-based test of the reference, geodesic.py:183 / mapping.py:705-721, for known camera models).
bool pole_visible(const amt_frame_params* p, double min_elevation, int magnetic) {
    const double* m = magnetic ? p->m_sm : p->m_geo;
    const double* r = p->rot;
    const double sc[3] = {1 / p->a, 1 / p->a, 1 / p->b};
    for (int sign = 1; sign >= -1; sign -= 2) {
        double u[3], pole[3], los[3], d[3], n2 = 0;
        for (int i = 0; i < 3; ++i) u[i] = m[6 + i] * sign;              // m^T (0,0,sign): pole axis in J2000
        for (int i = 0; i < 3; ++i) n2 += u[i] * sc[i] * u[i] * sc[i];
        double dist2 = 0;
        for (int i = 0; i < 3; ++i) {
            pole[i] = u[i] / std::sqrt(n2);
            los[i] = pole[i] - p->cam[i];
            dist2 += los[i] * los[i];
        }
        const double dist = std::sqrt(dist2);
        double d_o = 0, d_d = 0, o_o = 0;
        for (int i = 0; i < 3; ++i) {
            d[i] = los[i] / dist;
            const double ds = d[i] * sc[i], os = -p->cam[i] * sc[i];
            d_o += ds * os;
            d_d += ds * ds;
            o_o += os * os;
        }
        const double disc = d_o * d_o - o_o * d_d + d_d;
        if (disc < 0) continue;
        const double t = (o_o < 1 ? d_o + std::sqrt(disc) : d_o - std::sqrt(disc)) / d_d;
        if (std::fabs(t - dist) > 1e-6 * dist) continue;                  // the pole is on the far side
        double v[3];
        for (int i = 0; i < 3; ++i) v[i] = r[i] * d[0] + r[3 + i] * d[1] + r[6 + i] * d[2];   // rot^T d
        if (v[2] <= 0) continue;
        const double k = 180.0 / M_PI, bx = k * v[1] / v[2], by = -k * v[0] / v[2];
        const double det = p->cd[0] * p->cd[3] - p->cd[1] * p->cd[2];
        const double px = (bx * p->cd[3] - p->cd[1] * by) / det, py = (p->cd[0] * by - p->cd[2] * bx) / det;
        const double x = px + p->crpix[0] - 1, y = py + p->crpix[1] - 1;
        if (!(x >= -0.5 && x <= p->width - 0.5 && y >= -0.5 && y <= p->height - 0.5)) continue;
        if (!std::isinf(min_elevation)) {
            double dp = 0, pp = 0;
            for (int i = 0; i < 3; ++i) {
                dp += d[i] * pole[i];
                pp += pole[i] * pole[i];
            }
            double sn = -dp / std::sqrt(pp);
            sn = sn < -1 ? -1 : (sn > 1 ? 1 : sn);
            if (!(std::asin(sn) * k >= min_elevation)) continue;
        }
        return true;
    }
    return false;
}

}  // namespace

extern "C" {

int amt_grid_layout(double lat_px_per_deg, double lon_px_per_deg, double lat_min, double lat_max, double lon_min,
                    double lon_max, amt_grid* out) {
    if (out == nullptr) return AMT_EINVAL;
    std::memset(out, 0, sizeof(*out));
    return amt_gl::layout(lat_px_per_deg, lon_px_per_deg, lat_min, lat_max, lon_min, lon_max, out) ? AMT_OK
                                                                                                       : AMT_EINVAL;
}

int amt_plate_carree_resolution(double lat_south, double lon_west, double lat_north, double lon_east, double arcsec_per_px,
                                double* lat_px_per_deg, double* lon_px_per_deg) {
    if (lat_px_per_deg == nullptr || lon_px_per_deg == nullptr) return AMT_EINVAL;
    if (!amt_gl::plate_carree_resolution(lat_south, lon_west, lat_north, lon_east, arcsec_per_px, lat_px_per_deg, lon_px_per_deg))
        return AMT_EINVAL;
    return *lon_px_per_deg > 0 ? AMT_OK : AMT_EDOMAIN;
}

int amt_pipe_create(amt_ctx* ctx, amt_pipe** out_pipe) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, out_pipe != nullptr, "out_pipe is NULL");
    *out_pipe = nullptr;
    if (amt_set_device(ctx)) return AMT_EHIP;
    amt_pipe* pipe = new (std::nothrow) amt_pipe();
    if (pipe == nullptr) return AMT_ENOMEM;
    std::memset(pipe, 0, sizeof(*pipe));
    pipe->ctx = ctx;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (ctx->aux_pre == nullptr) (void)hipStreamCreateWithPriority(&ctx->aux_pre, hipStreamNonBlocking, hi);
    if (ctx->aux_tail == nullptr) (void)hipStreamCreateWithPriority(&ctx->aux_tail, hipStreamNonBlocking, hi);
    if (ctx->aux_fin == nullptr) (void)hipStreamCreateWithPriority(&ctx->aux_fin, hipStreamNonBlocking, hi);
    pipe->pre_stream = ctx->aux_pre;
    pipe->tail_stream = ctx->aux_tail;
    pipe->fin_stream = ctx->aux_fin;
    bool ok = pipe->pre_stream != nullptr && pipe->tail_stream != nullptr && pipe->fin_stream != nullptr &&
              hipEventCreateWithFlags(&pipe->coarse_done, hipEventDisableTiming) == hipSuccess &&
              hipEventCreate(&pipe->kernel_done) == hipSuccess &&
              hipEventCreateWithFlags(&pipe->bbox_done, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&pipe->tail_done, hipEventDisableTiming) == hipSuccess &&
              hipMalloc(&pipe->events, kEventCapacity * 32) == hipSuccess &&
              // (zeroed on the context's stream before the first fused launch, see pipe_prepare: a memset on the null stream
              // here is not ordered against the context's non-blocking stream and may run late when processes share the GPU)
              hipMalloc(reinterpret_cast<void**>(&pipe->event_count), sizeof(uint32_t)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&pipe->host_small), 16 * sizeof(double), hipHostMallocMapped) ==
                  hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void**>(&pipe->host_small_dev), pipe->host_small, 0) == hipSuccess;
    if (!ok) {
        ctx->last_error = "amt_pipe_create: resource allocation failed";
        amt_pipe_destroy(pipe);
        return AMT_EHIP;
    }
    *out_pipe = pipe;
    return AMT_OK;
}

int amt_pipe_destroy(amt_pipe* pipe) {
    if (pipe == nullptr) return AMT_EINVAL;
    if (pipe->pre_stream) (void)hipStreamSynchronize(pipe->pre_stream);
    if (pipe->tail_stream) (void)hipStreamSynchronize(pipe->tail_stream);
    if (pipe->fin_stream) (void)hipStreamSynchronize(pipe->fin_stream);
    if (pipe->coarse_done) (void)hipEventDestroy(pipe->coarse_done);
    if (pipe->kernel_done) (void)hipEventDestroy(pipe->kernel_done);
    if (pipe->bbox_done) (void)hipEventDestroy(pipe->bbox_done);
    if (pipe->tail_done) (void)hipEventDestroy(pipe->tail_done);
    if (pipe->partials) (void)hipFree(pipe->partials);
    if (pipe->events) (void)hipFree(pipe->events);
    if (pipe->event_count) (void)hipFree(pipe->event_count);
    if (pipe->host_small) (void)hipHostFree(pipe->host_small);
    if (pipe->acc) (void)hipFree(pipe->acc);
    delete pipe;
    return AMT_OK;
}

}  // extern "C"

namespace {

// coarse box in the coordinates of `mode` (see amt_pipe.coarse_magnetic)
int pipe_coarse(amt_pipe* pipe, const amt_frame_params* p, double min_elevation, int mode, const double* dirs = nullptr) {
    amt_ctx* ctx = pipe->ctx;
    // the pre-pass runs on the driver's own stream (and with that stream's workspace, see amt_workspace): the
    // context's stream is busy with the previous frames' kernels
    hipStream_t saved = ctx->stream;
    ctx->stream = pipe->pre_stream;
    const double thr = std::isinf(min_elevation) ? min_elevation : min_elevation - 0.5;
    const int shorter = p->width < p->height ? p->width : p->height;
    const int stride = std::max(1, std::min(kCoarseStride, shorter / 128));
    int rc = dirs ? amt_georef_coarse_bbox_dirs(ctx, p, dirs, stride, thr, mode, pipe->host_small_dev)
                  : amt_georef_coarse_bbox(ctx, p, stride, thr, mode, pipe->host_small_dev);
    ctx->stream = saved;
    if (rc != AMT_OK) return rc;
    AMT_HIP(ctx, hipEventRecord(pipe->coarse_done, pipe->pre_stream));
    pipe->coarse_pending = true;
    pipe->coarse_hinted = false;
    pipe->coarse_magnetic = mode;
    return AMT_OK;
}

}  // namespace

extern "C" {

int amt_pipe_coarse_dirs(amt_pipe* pipe, const amt_frame_params* p, const double* corner_dirs, double min_elevation,
                         int magnetic) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, p != nullptr && corner_dirs != nullptr, "NULL argument");
    return pipe_coarse(pipe, p, min_elevation, magnetic ? 1 : 0, corner_dirs);
}

int amt_pipe_coarse(amt_pipe* pipe, const amt_frame_params* p, double min_elevation, int magnetic) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, p != nullptr, "NULL argument");
    // a frame with a pole of its grid in view is binned in rotated coordinates (the pole plans): its box is needed in those
    const int mode = (magnetic ? 1 : 0) + (pole_visible(p, min_elevation, magnetic ? 1 : 0) ? 2 : 0);
    return pipe_coarse(pipe, p, min_elevation, mode);
}

}  // extern "C"

namespace {

int pipe_wait_tail(amt_pipe* pipe);
int pipe_prepare_rest(amt_pipe* pipe, const amt_frame_params* p, const amt_georef_out* out, const void* img,
                      int32_t img_dtype, double min_elevation, double lat_px_per_deg, double lon_px_per_deg,
                      int magnetic, int mode, amt_georef_out* o_out, amt_georef_tail* tail_out);

// Everything amt_pipe_launch does before the big kernel: wait for the coarse box, superset grid, accumulators,
// kernel outputs (`o`) and the tail description for this frame.
int pipe_prepare(amt_pipe* pipe, const amt_frame_params* p, const amt_georef_out* out, const void* img,
                 int32_t img_dtype, double min_elevation, double lat_px_per_deg, double lon_px_per_deg,
                 int pole_in_view, int magnetic, amt_georef_out* o_out, amt_georef_tail* tail_out,
                 const double* dirs = nullptr) {
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, p && out && img, "NULL argument");
    AMT_REQUIRE(ctx, img_dtype == 1 || img_dtype == 2, "img must be uint8 (1) or uint16 (2)");
    magnetic = magnetic ? 1 : 0;
    // (direction arrays: there is no camera model to project the pole through — the caller decides, or nobody does)
    pipe->pole_unknown = dirs != nullptr && pole_in_view < 0;
    pipe->pole = pole_in_view < 0 ? (dirs == nullptr && pole_visible(p, min_elevation, magnetic) ? 1 : 0) : (pole_in_view ? 1 : 0);
    // frames with a pole of their grid in view take a pole plan (binned in rotated coordinates; a geodetic pole frame of
    // a caller that wants the MLat / MLT arrays as well runs the kernel variant of the magnetic pole plan with the
    // rotated pair taken from (lat, lon), see prepare_georef); direction arrays have none: their pole frames are not fused
    const int mode = dirs ? magnetic : (magnetic ? (pipe->pole ? 3 : 1) : (pipe->pole ? 2 : 0));
    pipe->pole_plan = mode >= 2;
    if (!pipe->coarse_pending || pipe->coarse_magnetic != mode) {
        if (pipe->coarse_pending && !pipe->coarse_hinted) AMT_HIP(ctx, hipEventSynchronize(pipe->coarse_done));
        if (int rc = pipe_coarse(pipe, p, min_elevation, mode, dirs)) return rc;
    }
    if (!pipe->coarse_hinted) AMT_HIP(ctx, hipEventSynchronize(pipe->coarse_done));
    pipe->coarse_pending = false;
    pipe->coarse_hinted = false;
    pipe->lat_ppd = lat_px_per_deg;
    pipe->lon_ppd = lon_px_per_deg;
    pipe->min_elev = min_elevation;
    pipe->img_dtype = img_dtype;
    pipe->fused = false;

    if (int rc = pipe_wait_tail(pipe)) return rc;
    if (int rc = ensure_partials(pipe, amt_georef_partials_bytes(p))) return rc;
    return pipe_prepare_rest(pipe, p, out, img, img_dtype, min_elevation, lat_px_per_deg, lon_px_per_deg, magnetic, mode, o_out,
                             tail_out);
}

// the folds / finalise of the previous frame on this driver still use the partials and the accumulators
int pipe_wait_tail(amt_pipe* pipe) {
    amt_ctx* ctx = pipe->ctx;
    if (pipe->tail_pending) {
        // usually long finished (the host saw the frame's bounding box): then no packet goes on the stream.  A host
        // that is quick (the native sequence runner launches the next batch microseconds after it enqueued this driver's
        // finalise kernel) waits a moment for it instead — the big kernel of the batch before is running anyway —: every
        // wait packet between two big kernels lengthens their boundary by 3-4 us (kernel trace: 27 us per boundary
        // with three of them, 15-16 us with none)
        bool done = hipEventQuery(pipe->tail_done) == hipSuccess;
        // (two-pass plan: what the buffer waits for is its previous frame's binning pass, a kernel as long as the big one —
        // the wait goes on the stream at once)
        if (!done && !pipe->two_pass) {
            const auto t0 = std::chrono::steady_clock::now();
            while (!done && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(150))
                done = hipEventQuery(pipe->tail_done) == hipSuccess;
        }
        if (!done) AMT_HIP(ctx, hipStreamWaitEvent(ctx->stream, pipe->tail_done, 0));
        pipe->tail_pending = false;
    }
    return AMT_OK;
}

int pipe_prepare_rest(amt_pipe* pipe, const amt_frame_params* p, const amt_georef_out* out, const void* img,
                      int32_t img_dtype, double min_elevation, double lat_px_per_deg, double lon_px_per_deg,
                      int magnetic, int mode, amt_georef_out* o_out, amt_georef_tail* tail_out) {
    amt_ctx* ctx = pipe->ctx;
    amt_georef_out& o = *o_out;
    o = *out;
    o.bbox = pipe->host_small_dev + 8;          // the last fold writes straight into pinned host memory
    o.bbox_min_elevation = min_elevation;
    o.bin_acc = nullptr;
    o.bin_xaxis = o.bin_yaxis = nullptr;
    o.bin_img = nullptr;
    o.bin_img_dtype = o.bin_lon_wrap = o.bin_magnetic = o.bin_pole = 0;
    o.bin_events = nullptr;
    o.bin_event_count = nullptr;
    o.bin_event_capacity = 0;
    {
        // start with the side of the frame where the coarse pass found the hits (see amt_georef_out.item_order)
        const long long packed = (long long)pipe->host_small[7];
        long long sy = ((packed % (1 << 20)) + (1 << 20)) % (1 << 20);
        if (sy >= (1 << 19)) sy -= (1 << 20);
        const long long sx = (packed - sy) / (1 << 20);
        const long long ax = sx < 0 ? -sx : sx, ay = sy < 0 ? -sy : sy;
        o.item_order = (ax | ay) == 0 ? 0 : ((ay >= ax && sy > 0) ? 2 : 1);
    }

    pipe->g_lat_c = out->lat_c, pipe->g_lon_c = out->lon_c, pipe->g_elev = out->elev, pipe->g_img = img;
    pipe->g_row_layout = out->row_layout;
    pipe->g_width = p->width, pipe->g_height = p->height, pipe->g_fast = p->fast_center, pipe->g_mode = mode;
    pipe->general_ready = false;
    // coarse [lat_min, lat_max, lon_min, lon_max, lon_min_positive, lon_max_nonpositive, n, hint]
    const double* c = pipe->host_small;
    bool fuse = !pipe->two_pass && c[6] > 0 && (!pipe->pole || pipe->pole_plan);
    if (pipe->pole_unknown && !(c[0] > -kPoleGuardDeg && c[1] < kPoleGuardDeg)) fuse = false;
    pipe->lon_wrap = 0;
    double box_lo = c[2], box_hi = c[3];
    if (fuse && pipe->pole_plan && c[3] - c[2] > 180) fuse = false;      // (cannot happen: the rotated frame sits at the equator)
    if (fuse && c[3] - c[2] > 180) {
        // the box straddles the 180 deg discontinuity (mappings are narrower than 180 deg, mapping.py:722-737):
        // west = smallest positive, east = largest non-positive longitude; bin longitudes shifted by 180 deg
        // (reference resample.py:203-218)
        fuse = std::isfinite(c[4]) && std::isfinite(c[5]);
        box_lo = wrap_at_180(c[4] + 180.0);
        box_hi = wrap_at_180(c[5] + 180.0);
        pipe->lon_wrap = 1;
        fuse = fuse && box_hi > box_lo;
    }
    if (fuse) {
        const double lat_abs = std::fmax(std::fabs(c[0]), std::fabs(c[1]));
        const double lon_margin = std::fmin(10.0, kMarginDeg / std::fmax(0.1, std::cos(lat_abs * amt::kDeg2Rad)));
        const double lat_lo = std::fmax(-89.0, c[0] - kMarginDeg), lat_hi = std::fmin(89.0, c[1] + kMarginDeg);
        // (a box whose MARGIN reaches past +-180 deg — a footprint that ends just short of the date line — is cut there: when
        // the exact box turns out to straddle after all, amt_pipe_wait sees that the coarse pass judged it differently)
        const double lon_lo = std::fmax(-180.0, box_lo - lon_margin), lon_hi = std::fmin(180.0, box_hi + lon_margin);
        fuse = box_lo > -179.9 && box_hi < 179.9 &&
               amt_gl::layout(lat_px_per_deg, lon_px_per_deg, lat_lo, lat_hi, lon_lo, lon_hi, &pipe->super);
    }
    if (fuse) {
        const size_t cells = (size_t)pipe->super.nx * pipe->super.ny;
        if (int rc = ensure_acc(pipe, cells)) return rc;
        if (!pipe->acc_zero) {
            // first use / after a frame that was not finalised; otherwise the finalise kernel leaves zeros behind
            AMT_HIP(ctx, hipMemsetAsync(pipe->acc, 0, pipe->acc_cells * 5 * sizeof(uint64_t), ctx->stream));
            AMT_HIP(ctx, hipMemsetAsync(pipe->event_count, 0, sizeof(uint32_t), ctx->stream));
            pipe->acc_zero = true;
        }
        o.bin_xaxis = &pipe->super.xaxis;
        o.bin_yaxis = &pipe->super.yaxis;
        o.bin_img = img;
        o.bin_img_dtype = img_dtype;
        o.bin_acc = pipe->acc;
        o.bin_magnetic = magnetic;
        o.bin_pole = pipe->pole_plan ? 1 : 0;
        o.bin_lon_wrap = pipe->lon_wrap;
        o.bin_events = pipe->events;
        o.bin_event_count = pipe->event_count;
        o.bin_event_capacity = kEventCapacity;
        pipe->fused = true;
        pipe->acc_zero = false;
    }
    tail_out->stream = pipe->tail_stream;
    tail_out->kernel_done = pipe->kernel_done;
    tail_out->partials = pipe->partials;
    tail_out->partials_bytes = pipe->partials_bytes;
    return AMT_OK;
}

int pipe_after_launch(amt_pipe* pipe) {
    amt_ctx* ctx = pipe->ctx;
    AMT_HIP(ctx, hipEventRecord(pipe->bbox_done, pipe->tail_stream));
    AMT_HIP(ctx, hipEventRecord(pipe->tail_done, pipe->tail_stream));
    pipe->tail_pending = true;
    pipe->launched = true;
    return AMT_OK;
}

}  // namespace

extern "C" {

int amt_pipe_coarse_hint(amt_pipe* pipe, const double* bbox, int magnetic) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, bbox != nullptr, "NULL argument");
    if (pipe->coarse_pending) {
        // a pre-pass kernel may still be writing the same 8 doubles
        AMT_HIP(ctx, hipEventSynchronize(pipe->coarse_done));
    }
    for (int i = 0; i < 7; ++i) pipe->host_small[i] = bbox[i];
    pipe->host_small[7] = 0;                       // no hit statistics: item order from the camera model
    pipe->coarse_pending = true;
    // the exact box of a pole-plan frame (bbox[7] = 1 in amt_pipe_result of a fused frame) is in rotated coordinates
    pipe->coarse_magnetic = (magnetic ? 1 : 0) + (bbox[7] != 0 ? 2 : 0);
    pipe->coarse_hinted = true;
    return AMT_OK;
}

int amt_pipe_launch_many(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p,
                         const amt_georef_out* const* out, const void* const* img, int32_t img_dtype,
                         double min_elevation, double lat_px_per_deg, double lon_px_per_deg, int pole_in_view,
                         int magnetic) {
    double la[AMT_MAX_BATCH], lo[AMT_MAX_BATCH];
    for (int i = 0; i < AMT_MAX_BATCH; ++i) la[i] = lat_px_per_deg, lo[i] = lon_px_per_deg;
    return amt_pipe_launch_many_res(pipes, n, p, out, img, img_dtype, min_elevation, la, lo, pole_in_view, magnetic);
}

int amt_pipe_launch_many_res(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p,
                             const amt_georef_out* const* out, const void* const* img, int32_t img_dtype,
                             double min_elevation, const double* lat_px_per_deg, const double* lon_px_per_deg,
                             int pole_in_view, int magnetic) {
    if (pipes == nullptr || n < 1 || pipes[0] == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipes[0]->ctx;
    AMT_REQUIRE(ctx, n <= AMT_MAX_BATCH, "at most AMT_PIPE_MAX_BATCH frames per launch");
    AMT_REQUIRE(ctx, p && out && img && lat_px_per_deg && lon_px_per_deg, "NULL argument");
    amt_georef_out o[AMT_MAX_BATCH];
    amt_georef_tail tails[AMT_MAX_BATCH];
    const amt_georef_out* op[AMT_MAX_BATCH];
    const amt_georef_tail* tp[AMT_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        AMT_REQUIRE(ctx, pipes[i] != nullptr && pipes[i]->ctx == ctx, "drivers of one launch must share the context");
        for (int k = 0; k < i; ++k) AMT_REQUIRE(ctx, pipes[k] != pipes[i], "a driver can hold one frame of a launch");
        if (int rc = pipe_prepare(pipes[i], p[i], out[i], img[i], img_dtype, min_elevation, lat_px_per_deg[i],
                                  lon_px_per_deg[i], pole_in_view, magnetic, &o[i], &tails[i]))
            return rc;
        op[i] = &o[i];
        tp[i] = &tails[i];
    }
    if (int rc = amt_georef_launch_many(ctx, n, p, op, tp)) return rc;
    for (int i = 0; i < n; ++i)
        if (int rc = pipe_after_launch(pipes[i])) return rc;
    return AMT_OK;
}

int amt_pipe_launch_dirs(amt_pipe* pipe, const amt_frame_params* p, const double* corner_dirs, const amt_georef_out* out,
                         const void* img, int32_t img_dtype, double min_elevation, double lat_px_per_deg,
                         double lon_px_per_deg, int pole_in_view, int magnetic) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, p && corner_dirs && out && img, "NULL argument");
    AMT_REQUIRE(ctx, p->fast_center, "caller-supplied directions need fast_center");
    amt_georef_out o;
    amt_georef_tail tail;
    if (int rc = pipe_prepare(pipe, p, out, img, img_dtype, min_elevation, lat_px_per_deg, lon_px_per_deg, pole_in_view, magnetic,
                              &o, &tail, corner_dirs))
        return rc;
    pipe->g_mode = -1;                    // (amt_pipe_general_layout: the pole of a direction-array frame is the caller's business)
    if (int rc = amt_georef_launch(ctx, p, corner_dirs, &o, &tail)) return rc;
    return pipe_after_launch(pipe);
}

int amt_pipe_launch_dirs_many(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p, const double* const* corner_dirs,
                              const amt_georef_out* const* out, const void* const* img, int32_t img_dtype, double min_elevation,
                              double lat_px_per_deg, double lon_px_per_deg, int pole_in_view, int magnetic) {
    if (pipes == nullptr || n < 1 || pipes[0] == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipes[0]->ctx;
    AMT_REQUIRE(ctx, n <= AMT_MAX_BATCH, "at most AMT_PIPE_MAX_BATCH frames per launch");
    AMT_REQUIRE(ctx, p && corner_dirs && out && img, "NULL argument");
    amt_georef_out o[AMT_MAX_BATCH];
    amt_georef_tail tails[AMT_MAX_BATCH];
    const amt_georef_out* op[AMT_MAX_BATCH];
    const amt_georef_tail* tp[AMT_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        AMT_REQUIRE(ctx, pipes[i] != nullptr && pipes[i]->ctx == ctx, "drivers of one launch must share the context");
        for (int k = 0; k < i; ++k) AMT_REQUIRE(ctx, pipes[k] != pipes[i], "a driver can hold one frame of a launch");
        AMT_REQUIRE(ctx, p[i] && corner_dirs[i] && out[i] && img[i], "NULL argument");
        AMT_REQUIRE(ctx, p[i]->fast_center, "caller-supplied directions need fast_center");
        if (int rc = pipe_prepare(pipes[i], p[i], out[i], img[i], img_dtype, min_elevation, lat_px_per_deg, lon_px_per_deg,
                                  pole_in_view, magnetic, &o[i], &tails[i], corner_dirs[i]))
            return rc;
        pipes[i]->g_mode = -1;
        op[i] = &o[i];
        tp[i] = &tails[i];
    }
    if (int rc = amt_georef_launch_many_dirs(ctx, n, p, corner_dirs, op, tp)) return rc;
    for (int i = 0; i < n; ++i)
        if (int rc = pipe_after_launch(pipes[i])) return rc;
    return AMT_OK;
}

int amt_pipe_launch_box_many(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p, double min_elevation,
                             int magnetic) {
    if (pipes == nullptr || n < 1 || pipes[0] == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipes[0]->ctx;
    AMT_REQUIRE(ctx, n <= AMT_MAX_BATCH, "at most AMT_PIPE_MAX_BATCH frames per launch");
    AMT_REQUIRE(ctx, p != nullptr, "NULL argument");
    magnetic = magnetic ? 1 : 0;
    amt_georef_out o[AMT_MAX_BATCH];
    amt_georef_tail tails[AMT_MAX_BATCH];
    const amt_georef_out* op[AMT_MAX_BATCH];
    const amt_georef_tail* tp[AMT_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        amt_pipe* pipe = pipes[i];
        AMT_REQUIRE(ctx, pipe != nullptr && pipe->ctx == ctx && p[i] != nullptr, "drivers of one launch must share the context");
        for (int k = 0; k < i; ++k) AMT_REQUIRE(ctx, pipes[k] != pipe, "a driver can hold one frame of a launch");
        // (an estimate that is still pending belongs to another plan of this frame: dropped)
        if (pipe->coarse_pending && !pipe->coarse_hinted) AMT_HIP(ctx, hipEventSynchronize(pipe->coarse_done));
        pipe->coarse_pending = pipe->coarse_hinted = false;
        pipe->pole = pole_visible(p[i], min_elevation, magnetic) ? 1 : 0;
        pipe->pole_unknown = false;
        pipe->pole_plan = false;
        pipe->lat_ppd = pipe->lon_ppd = 0;
        pipe->min_elev = min_elevation;
        pipe->fused = false;
        pipe->general_ready = false;
        pipe->lon_wrap = 0;
        pipe->g_lat_c = pipe->g_lon_c = pipe->g_elev = nullptr;
        pipe->g_img = nullptr;
        if (int rc = pipe_wait_tail(pipe)) return rc;
        if (int rc = ensure_partials(pipe, amt_georef_partials_bytes(p[i]))) return rc;
        std::memset(&o[i], 0, sizeof(o[i]));
        o[i].bbox = pipe->host_small_dev + 8;
        o[i].bbox_min_elevation = min_elevation;
        o[i].bin_magnetic = magnetic;          // without bin_acc: the box in (MLat, SM longitude)
        tails[i].stream = pipe->tail_stream;
        tails[i].kernel_done = pipe->kernel_done;
        tails[i].partials = pipe->partials;
        tails[i].partials_bytes = pipe->partials_bytes;
        op[i] = &o[i];
        tp[i] = &tails[i];
    }
    if (int rc = amt_georef_launch_many(ctx, n, p, op, tp)) return rc;
    for (int i = 0; i < n; ++i)
        if (int rc = pipe_after_launch(pipes[i])) return rc;
    return AMT_OK;
}

int amt_pipe_launch_box(amt_pipe* pipe, const amt_frame_params* p, double min_elevation, int magnetic) {
    if (pipe == nullptr) return AMT_EINVAL;
    return amt_pipe_launch_box_many(&pipe, 1, &p, min_elevation, magnetic);
}

int amt_pipe_launch(amt_pipe* pipe, const amt_frame_params* p, const amt_georef_out* out, const void* img,
                    int32_t img_dtype, double min_elevation, double lat_px_per_deg, double lon_px_per_deg,
                    int pole_in_view, int magnetic) {
    if (pipe == nullptr) return AMT_EINVAL;
    return amt_pipe_launch_many(&pipe, 1, &p, &out, &img, img_dtype, min_elevation, lat_px_per_deg, lon_px_per_deg,
                                pole_in_view, magnetic);
}


int amt_pipe_wait(amt_pipe* pipe, amt_pipe_result* result) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, result != nullptr, "result is NULL");
    AMT_REQUIRE(ctx, pipe->launched, "amt_pipe_launch has not been called for this frame");
    pipe->launched = false;
    pipe->ready = false;
    AMT_HIP(ctx, hipEventSynchronize(pipe->bbox_done));
    std::memset(result, 0, sizeof(*result));
    const double* b = pipe->host_small + 8;
    for (int i = 0; i < 8; ++i) result->bbox[i] = b[i];
    result->bbox[7] = pipe->pole ? 1.0 : 0.0;      // pole containment comes from the camera model
    result->fused = pipe->fused ? 1 : 0;
    if (b[6] == 0) {
        result->status = 2;
        return AMT_OK;
    }
    result->status = 1;
    pipe->n_events = pipe->fused ? (long long)b[7] : 0;        // the last fold put the counter into slot 7
    if (!pipe->fused || (pipe->pole && !pipe->pole_plan)) return AMT_OK;
    result->edge_pixels = (int32_t)(pipe->n_events > 2000000000ll ? 2000000000ll : pipe->n_events);
    if (pipe->n_events > kEventCapacity) return AMT_OK;         // more on-edge pixels than records: general path
    // direction arrays whose caller left the pole open: a box that comes near a pole goes back to the caller (status 1)
    if (pipe->pole_unknown && !(b[0] > -kPoleGuardDeg && b[1] < kPoleGuardDeg)) return AMT_OK;
    const bool straddles = b[3] - b[2] > 180;
    if (straddles != (pipe->lon_wrap != 0)) return AMT_OK;      // the coarse pass judged the discontinuity differently
    if (pipe->pole_plan) {
        // the kernel knows the rotated corners to ~1e-11 deg, the two-pass plan to the bit: the grids are the same
        // unless an extreme of the box sits that close to a grid node (k / px-per-deg); then the general path decides
        if (straddles) return AMT_OK;
        const double v[4] = {b[0] * pipe->lat_ppd, b[1] * pipe->lat_ppd, b[2] * pipe->lon_ppd, b[3] * pipe->lon_ppd};
        for (int i = 0; i < 4; ++i)
            if (!(std::fabs(v[i] - std::nearbyint(v[i])) > 1e-6)) return AMT_OK;
    }
    double lon_lo = b[2], lon_hi = b[3];
    if (straddles) {
        if (!(std::isfinite(b[4]) && std::isfinite(b[5]))) return AMT_OK;
        lon_lo = wrap_at_180(b[4] + 180.0);
        lon_hi = wrap_at_180(b[5] + 180.0);
    }
    result->lon_wrapped = straddles ? 1 : 0;
    amt_grid& g = result->grid;
    if (!amt_gl::layout(pipe->lat_ppd, pipe->lon_ppd, b[0], b[1], lon_lo, lon_hi, &g)) return AMT_OK;
    const amt_grid& s = pipe->super;
    // window of the exact grid inside the superset (same global nodes => integer offsets)
    const long off_x = std::lround((g.lon_center_first - s.lon_center_first) / s.lon_step);
    const long off_y = std::lround((g.lat_center_last - s.lat_center_last) / std::fabs(s.lat_step));
    if (off_x < 0 || off_y < 0 || off_x + g.nx > s.nx || off_y + g.ny > s.ny) return AMT_OK;
    const double lon_at = s.lon_center_first + off_x * s.lon_step;
    const double lat_at = s.lat_center_last + off_y * std::fabs(s.lat_step);
    if (std::fabs(lon_at - g.lon_center_first) > 1e-9 || std::fabs(lat_at - g.lat_center_last) > 1e-9) return AMT_OK;
    pipe->exact = g;
    pipe->off_x = (int32_t)off_x;
    pipe->off_y = (int32_t)off_y;
    pipe->ready = true;
    result->status = 0;
    return AMT_OK;
}

int amt_pipe_finalize(amt_pipe* pipe, double* mean, void* out_img, uint8_t* out_mask, double* out_count) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, pipe->ready, "amt_pipe_wait has not returned status 0 for this frame");
    pipe->ready = false;
    const amt_grid& s = pipe->super;
    const amt_grid& g = pipe->exact;
    // on-edge pixels first (their bin depends on where the final grid ends), then the crop: on the finalise stream
    // (this frame's big kernel and folds are complete: the host has read the box), zeroing the accumulators for
    // the next frame on the way
    if (pipe->n_events > 0) {
        if (int rc = amt_bin_apply_events_on(ctx, pipe->fin_stream, pipe->events, pipe->event_count, pipe->acc, s.nx, s.ny,
                                             pipe->off_x, pipe->off_y, g.nx, g.ny))
            return rc;
    }
    if (int rc = amt_bin_finalize_on(ctx, pipe->fin_stream, pipe->acc, s.nx, s.ny, pipe->off_x, pipe->off_y, g.nx, g.ny,
                                     3, pipe->img_dtype, mean, out_img, out_mask, out_count, 1))
        return rc;
    pipe->acc_zero = true;
    AMT_HIP(ctx, hipEventRecord(pipe->tail_done, pipe->fin_stream));
    pipe->tail_pending = true;
    return AMT_OK;
}

int amt_pipe_finalize_many(amt_pipe* const* pipes, int32_t n, double* const* mean, void* const* out_img,
                           uint8_t* const* out_mask, double* const* out_count) {
    if (pipes == nullptr || n < 1 || pipes[0] == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipes[0]->ctx;
    AMT_REQUIRE(ctx, n <= AMT_MAX_BATCH, "at most AMT_PIPE_MAX_BATCH frames per call");
    AMT_REQUIRE(ctx, mean && out_img && out_mask && out_count, "NULL argument");
    finish_batch B;
    std::memset(&B, 0, sizeof(B));
    B.n = n;
    for (int i = 0; i < n; ++i) {
        amt_pipe* pipe = pipes[i];
        AMT_REQUIRE(ctx, pipe != nullptr && pipe->ctx == ctx && pipe->fin_stream == pipes[0]->fin_stream,
                    "drivers of one call must share the context");
        for (int k = 0; k < i; ++k) AMT_REQUIRE(ctx, pipes[k] != pipe, "a driver can hold one frame of a call");
        AMT_REQUIRE(ctx, pipe->ready, "amt_pipe_wait has not returned status 0 for this frame");
        AMT_REQUIRE(ctx, pipe->img_dtype == pipes[0]->img_dtype, "frames of one call must share the image type");
        AMT_REQUIRE(ctx, pipe->n_events <= kEventCapacity, "more on-edge pixels than records");
        finish_frame& F = B.f[i];
        F.events = pipe->events;
        F.count = pipe->event_count;
        F.acc = reinterpret_cast<unsigned long long*>(pipe->acc);
        F.acc_nx = pipe->super.nx, F.acc_ny = pipe->super.ny;
        F.off_x = pipe->off_x, F.off_y = pipe->off_y;
        F.nx = pipe->exact.nx, F.ny = pipe->exact.ny;
        F.n_events = (unsigned int)pipe->n_events;
        F.mean = mean[i], F.img = out_img[i], F.mask = out_mask[i], F.out_count = out_count[i];
    }
    // one launch for all of them, on the finalise stream (see amt_pipe_finalize)
    if (int rc = amt_pipe_finish_on(ctx, pipes[0]->fin_stream, B, pipes[0]->img_dtype)) return rc;
    for (int i = 0; i < n; ++i) {
        amt_pipe* pipe = pipes[i];
        pipe->ready = false;
        pipe->acc_zero = true;
        AMT_HIP(ctx, hipEventRecord(pipe->tail_done, pipe->fin_stream));
        pipe->tail_pending = true;
    }
    return AMT_OK;
}

int amt_pipe_set_plan(amt_pipe* pipe, int two_pass) {
    if (pipe == nullptr) return AMT_EINVAL;
    pipe->two_pass = two_pass != 0;
    return AMT_OK;
}

int amt_pipe_general_layout(amt_pipe* pipe, amt_pipe_result* result) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, result != nullptr, "result is NULL");
    pipe->general_ready = false;
    if (result->status != 1) return AMT_OK;
    // what the kernels of this path do not cover stays with the caller: a pole in view (binned in rotated coordinates),
    // exact centres (their masks are reconciled first), MLat / MLT grids, frames whose coordinate arrays were not written or
    // lie in strip-padded rows (amt_bin_frame reads contiguous rows: the caller compacts them first, amt_unpad_rows)
    if (pipe->pole || !pipe->g_fast || pipe->g_mode != 0 || !pipe->g_lat_c || !pipe->g_lon_c || !pipe->g_elev || pipe->g_row_layout != 0) return AMT_OK;
    const double* b = result->bbox;
    if (!(b[6] > 0)) return AMT_OK;
    // BaseMapping.boundingBox + the date-line branch of _resample (reference mapping.py:711-741, resample.py:203-218)
    const bool straddles = b[3] - b[2] > 180;
    double lon_lo = b[2], lon_hi = b[3];
    if (straddles) {
        if (!(std::isfinite(b[4]) && std::isfinite(b[5]))) return AMT_OK;
        lon_lo = wrap_at_180(b[4] + 180.0);
        lon_hi = wrap_at_180(b[5] + 180.0);
    }
    amt_grid g;
    if (!amt_gl::layout(pipe->lat_ppd, pipe->lon_ppd, b[0], b[1], lon_lo, lon_hi, &g)) return AMT_OK;
    if (g.nx >= 65535 || g.ny >= 65535) return AMT_OK;
    result->grid = g;
    result->lon_wrapped = straddles ? 1 : 0;
    result->status = 0;
    pipe->exact = g;
    pipe->lon_wrap = straddles ? 1 : 0;
    pipe->general_ready = true;
    return AMT_OK;
}

int amt_pipe_general_finalize(amt_pipe* pipe, double* mean, void* out_img, uint8_t* out_mask, double* out_count) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    AMT_REQUIRE(ctx, pipe->general_ready, "amt_pipe_general_layout has not laid out this frame");
    AMT_REQUIRE(ctx, mean && out_img && out_mask && out_count, "NULL argument");
    pipe->general_ready = false;
    const amt_grid& g = pipe->exact;
    const size_t cells = (size_t)g.nx * g.ny;
    if (int rc = ensure_acc(pipe, cells)) return rc;
    // The binning pass and the finalise kernel go on the context's stream, IN LINE with the big kernels (behind this frame's,
    // which has finished: the host has its box; in a sequence behind the next frame's, which is running or queued).  Beside
    // a big kernel — on the finalise stream — the two slow each other down by more than they overlap (kernel trace of a
    // two-pass sequence: binning 82 -> 190-240 us, big kernel 118 -> 140-245 us; profiles/r3/x_two_pass_timeline.txt).
    hipStream_t fs = ctx->stream;
    AMT_HIP(ctx, hipMemsetAsync(pipe->acc, 0, cells * 5 * sizeof(uint64_t), fs));
    int rc = amt_bin_frame(ctx, pipe->g_lat_c, pipe->g_lon_c, pipe->g_elev, pipe->g_img, pipe->img_dtype, 3, nullptr, pipe->g_height,
                           pipe->g_width, pipe->min_elev, &g.xaxis, &g.yaxis, pipe->lon_wrap, pipe->acc);
    if (rc != AMT_OK) return rc;
    rc = amt_bin_finalize_on(ctx, fs, pipe->acc, g.nx, g.ny, 0, 0, g.nx, g.ny, 3, pipe->img_dtype, mean, out_img, out_mask,
                             out_count, 0);
    if (rc != AMT_OK) return rc;
    pipe->acc_zero = false;
    AMT_HIP(ctx, hipEventRecord(pipe->tail_done, fs));
    pipe->tail_pending = true;
    return AMT_OK;
}

int amt_pipe_finalize_stream(amt_pipe* pipe, void** stream) {
    if (pipe == nullptr || stream == nullptr) return AMT_EINVAL;
    *stream = reinterpret_cast<void*>(pipe->fin_stream);
    return AMT_OK;
}

int amt_pipe_join(amt_pipe* pipe) {
    if (pipe == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = pipe->ctx;
    if (pipe->tail_pending && hipEventQuery(pipe->tail_done) != hipSuccess)
        AMT_HIP(ctx, hipStreamWaitEvent(ctx->stream, pipe->tail_done, 0));
    return AMT_OK;
}

}  // extern "C"
