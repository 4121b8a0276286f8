// Native sequence runner (include/auromat_hip.h, "native sequence runner"): the per-frame host loop of a sequence —
// host scalars, box hints, launches, waits, grid layouts, finalise kernels — in one C call.  Host orchestration only;
// it drives the single-pass frame drivers of amt_pipe.hip through their public entry points.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "amt_common.h"
#include "amt_grid.h"
#include "amt_params.h"

struct amt_run {
    amt_ctx* ctx;
    amt_run_config cfg;
    std::vector<amt_pipe*> pipes;
    std::vector<amt_georef_out> outs;       // per slot: the caller's arrays (+ the altitude of the frame in the slot)
    std::vector<amt_frame_params> prm;      // per slot: the frame in flight
    std::vector<double> alt;                // per slot
    std::vector<int> hinted;                // per slot
    hipEvent_t entry;                       // the context's stream at the start of a call (the finalise stream waits for it)
    // box hints: the two latest frames finished by the single-pass plan (exact reduction, params, running index)
    struct hint {
        bool valid;
        double box[8];
        amt_frame_params p;
        long long index;
    } last, prev;
    long long frames_done;
    // the call in progress (amt_run_begin ... amt_run_end)
    bool active, full;
    double* grids;
    char* images;
    int64_t grids_capacity, images_capacity, grid_used, image_used;
    amt_run_result* results;
    int max_frames, n_pushed;
    int batch_k0, batch_count;              // prepared frames of the batch being collected
    int launched[2][2], n_launched;         // (first frame, count) of the batches in flight, older first
    std::vector<const void*> img;           // per slot: the image of the frame in the slot
    // box-first plan (cfg.arcsec_per_px > 0): batches whose box pass is in flight, older first; per slot the resolution the
    // frame's box gave, and what became of a frame that was not launched (2 no valid pixel, 4 a pole in view, 0 launched)
    int boxed[2][2], n_boxed;
    std::vector<double> ppd_lat, ppd_lon;
    std::vector<int> pre_status;
    // host images (amt_run_frame.img_host): per slot a device image buffer of the runner's own (allocated when the slot first
    // takes such a frame), the event behind the frame's upload on the runner's copy stream (the launch that reads the buffer
    // waits for it), and the event behind a separate binning pass that still reads the buffer (the next upload waits for it)
    std::vector<void*> own_img;
    std::vector<hipEvent_t> uploaded, img_free;
    std::vector<char> upload_pending, img_busy;
    std::vector<int64_t> uploaded_bytes;
    hipStream_t copy_stream;
};

namespace {

bool all_within(const double* a, const double* b, int n, double tol) {
    for (int i = 0; i < n; ++i)
        if (std::fabs(a[i] - b[i]) > tol) return false;
    return true;
}

// neighbours in a sequence (auromat_amd/pipeline.py _close): same frame size, camera model within 1 % in scale, camera
// within 100 km, boresight and Earth rotation within about half a degree, shell within 30 km
bool close_frames(const amt_frame_params& a, const amt_frame_params& b) {
    if (a.width != b.width || a.height != b.height || a.fast_center != b.fast_center) return false;
    if (std::fabs(a.a - b.a) > 30.0 || std::fabs(a.b - b.b) > 30.0) return false;
    double cd_max = 0;
    for (int i = 0; i < 4; ++i) cd_max = std::max(cd_max, std::fabs(a.cd[i]));
    return all_within(a.cam, b.cam, 3, 100.0) && all_within(a.rot, b.rot, 9, 0.01) && all_within(a.m_geo, b.m_geo, 9, 0.01) &&
           all_within(a.m_sm, b.m_sm, 9, 0.01) && all_within(a.cd, b.cd, 4, 0.01 * cd_max) && all_within(a.crpix, b.crpix, 2, 5.0);
}

// frames a, b (n_ab apart) and c (n_bc after b): a steady sequence (auromat_amd/pipeline.py _steady)
bool steady_frames(const amt_frame_params& a, const amt_frame_params& b, const amt_frame_params& c, long long n_ab, long long n_bc) {
    if (n_ab <= 0 || n_bc <= 0 || n_bc > 16) return false;
    if (b.width != c.width || b.height != c.height || b.fast_center != c.fast_center) return false;
    if (std::fabs(b.a - c.a) > 30.0 || std::fabs(b.b - c.b) > 30.0) return false;
    if (!(all_within(b.cam, c.cam, 3, 400.0) && all_within(b.rot, c.rot, 9, 0.05) && all_within(b.m_geo, c.m_geo, 9, 0.05) &&
          all_within(b.m_sm, c.m_sm, 9, 0.05) && all_within(b.crpix, c.crpix, 2, 5.0)))
        return false;
    const double scale_b = std::sqrt(std::fabs(b.cd[0] * b.cd[3] - b.cd[1] * b.cd[2]));
    const double scale_c = std::sqrt(std::fabs(c.cd[0] * c.cd[3] - c.cd[1] * c.cd[2]));
    if (!(scale_b > 0 && std::fabs(scale_c - scale_b) <= 0.01 * scale_b)) return false;
    for (int i = 0; i < 4; ++i) {
        const double step = (b.cd[i] - a.cd[i]) / n_ab;
        if (std::fabs((c.cd[i] - b.cd[i]) - step * n_bc) > 0.3 * std::fabs(step * n_bc) + 0.01 * scale_b) return false;
    }
    for (int i = 0; i < 3; ++i) {
        const double step = (b.cam[i] - a.cam[i]) / n_ab;
        if (std::fabs((c.cam[i] - b.cam[i]) - step * n_bc) > 0.2 * std::fabs(step * n_bc) + 5.0) return false;
    }
    for (int i = 0; i < 9; ++i) {
        const double step = (b.rot[i] - a.rot[i]) / n_ab;
        if (std::fabs((c.rot[i] - b.rot[i]) - step * n_bc) > 0.3 * std::fabs(step * n_bc) + 2e-3) return false;
    }
    return true;
}

// estimate of frame k's box reduction from finished frames (auromat_amd/pipeline.py SequencePipeline._box_hint)
bool box_hint(const amt_run* run, long long k, const amt_frame_params& p, double* est) {
    const amt_run::hint &last = run->last, &prev = run->prev;
    if (!last.valid) return false;
    if (close_frames(last.p, p)) {
        std::memcpy(est, last.box, sizeof(last.box));
        return true;
    }
    if (!prev.valid || !close_frames(prev.p, last.p) || !steady_frames(prev.p, last.p, p, last.index - prev.index, k - last.index))
        return false;
    if ((prev.box[7] != 0) != (last.box[7] != 0) || (last.box[3] - last.box[2] > 180) != (prev.box[3] - prev.box[2] > 180))
        return false;                               // a pole or the date line came into view between the two
    const double f = (double)(k - last.index) / (double)(last.index - prev.index);
    for (int i = 0; i < 6; ++i) est[i] = last.box[i] + f * (last.box[i] - prev.box[i]);
    est[6] = last.box[6], est[7] = last.box[7];
    est[0] = std::max(est[0], -90.0), est[1] = std::min(est[1], 90.0);
    for (int i = 2; i < 6; ++i) est[i] = std::min(std::max(est[i], -180.0), 180.0);
    return true;
}

int64_t image_bytes(const amt_run* run, int64_t cells) {
    return (cells * 3 * (run->cfg.img_dtype == 2 ? 2 : 1) + cells + 255) & ~(int64_t)255;
}

}  // namespace

extern "C" {

int amt_frame_params_from_wcs(const amt_run_frame* frame, int32_t width, int32_t height, int32_t fast_center,
                              double altitude, int32_t want_sm, amt_frame_params* out) {
    if (frame == nullptr || out == nullptr || width <= 0 || height <= 0) return AMT_EINVAL;
    return amt_prm::frame_params(frame, width, height, fast_center, altitude, want_sm, out);
}

int amt_run_create(amt_ctx* ctx, const amt_run_config* config, amt_run** out_run) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, config != nullptr && out_run != nullptr, "NULL argument");
    *out_run = nullptr;
    AMT_REQUIRE(ctx, config->width > 0 && config->height > 0, "empty frame");
    AMT_REQUIRE(ctx, config->img_dtype == 1 || config->img_dtype == 2, "img_dtype must be 1 (uint8) or 2 (uint16)");
    AMT_REQUIRE(ctx, config->batch >= 1 && config->batch <= AMT_PIPE_MAX_BATCH, "batch out of range");
    AMT_REQUIRE(ctx, config->n_slots >= 2 * config->batch && config->slots != nullptr, "n_slots must be at least 2 * batch");
    AMT_REQUIRE(ctx, config->arcsec_per_px > 0 || (config->lat_px_per_deg > 0 && config->lon_px_per_deg > 0), "px per degree must be positive");
    AMT_REQUIRE(ctx, !(config->arcsec_per_px > 0) || (!config->two_pass && config->n_slots >= 3 * config->batch),
                "arcsec_per_px: the box-first plan is a single-pass plan and needs n_slots >= 3 * batch");
    amt_run* run = new (std::nothrow) amt_run();
    if (run == nullptr) return AMT_ENOMEM;
    run->ctx = ctx;
    run->cfg = *config;
    run->entry = nullptr;
    run->last.valid = run->prev.valid = false;
    run->frames_done = 0;
    const int ns = config->n_slots;
    run->outs.assign(config->slots, config->slots + ns);
    run->cfg.slots = nullptr;
    run->prm.resize(ns);
    run->alt.assign(ns, config->altitude);
    run->hinted.assign(ns, 0);
    run->img.assign(ns, nullptr);
    run->ppd_lat.assign(ns, config->lat_px_per_deg);
    run->ppd_lon.assign(ns, config->lon_px_per_deg);
    run->pre_status.assign(ns, 0);
    run->own_img.assign(ns, nullptr);
    run->uploaded.assign(ns, nullptr);
    run->img_free.assign(ns, nullptr);
    run->upload_pending.assign(ns, 0);
    run->img_busy.assign(ns, 0);
    run->uploaded_bytes.assign(ns, 0);
    run->copy_stream = nullptr;
    run->active = run->full = false;
    run->n_launched = 0, run->batch_count = 0, run->n_boxed = 0;
    run->pipes.assign(ns, nullptr);
    if (amt_set_device(ctx)) {
        delete run;
        return AMT_EHIP;
    }
    for (int i = 0; i < ns; ++i) {
        int rc = amt_pipe_create(ctx, &run->pipes[i]);
        if (rc == AMT_OK) rc = amt_pipe_set_plan(run->pipes[i], config->two_pass);
        if (rc != AMT_OK) {
            amt_run_destroy(run);
            return rc;
        }
    }
    if (hipEventCreateWithFlags(&run->entry, hipEventDisableTiming) != hipSuccess) {
        ctx->last_error = "amt_run_create: event creation failed";
        amt_run_destroy(run);
        return AMT_EHIP;
    }
    *out_run = run;
    return AMT_OK;
}

int amt_run_destroy(amt_run* run) {
    if (run == nullptr) return AMT_EINVAL;
    for (amt_pipe* p : run->pipes)
        if (p != nullptr) amt_pipe_destroy(p);
    if (run->entry) (void)hipEventDestroy(run->entry);
    if (run->copy_stream) (void)hipStreamSynchronize(run->copy_stream);
    for (void* b : run->own_img)
        if (b != nullptr) (void)hipFree(b);
    for (hipEvent_t e : run->uploaded)
        if (e != nullptr) (void)hipEventDestroy(e);
    for (hipEvent_t e : run->img_free)
        if (e != nullptr) (void)hipEventDestroy(e);
    if (run->copy_stream) (void)hipStreamDestroy(run->copy_stream);
    delete run;
    return AMT_OK;
}

int amt_run_reset_hints(amt_run* run) {
    if (run == nullptr) return AMT_EINVAL;
    run->last.valid = run->prev.valid = false;
    return AMT_OK;
}

// ---- the loop as a state machine: begin / push (one frame at a time) / end ------------------------------------------
// The batches in flight form a queue of at most two: when a batch of prepared frames is complete, the older launched
// batch is finished first if two are in flight (wait for the boxes, lay out the grids, one finalise kernel), then the new
// batch is launched; and a frame whose slot still belongs to a frame in flight finishes that frame's batch first
// (amt_run_push).  The frames arrive one by one, so the first launch happens after the first frame's preparation whatever
// the length of the sequence.

namespace {

int run_finish(amt_run* run, int k0, int count) {
    amt_ctx* ctx = run->ctx;
    const amt_run_config& cfg = run->cfg;
    const int ns = cfg.n_slots;
    amt_pipe* pp[AMT_PIPE_MAX_BATCH];
    double* mean[AMT_PIPE_MAX_BATCH];
    void* oimg[AMT_PIPE_MAX_BATCH];
    uint8_t* omask[AMT_PIPE_MAX_BATCH];
    double* ocount[AMT_PIPE_MAX_BATCH];
    int m = 0;
    for (int i = 0; i < count; ++i) {
        const int k = k0 + i, slot = k % ns;
        amt_run_result& r = run->results[k];
        std::memset(&r, 0, sizeof(r));
        r.slot = slot;
        r.hinted = run->hinted[slot];
        r.altitude = run->alt[slot];
        r.params = run->prm[slot];
        r.lat_px_per_deg = run->ppd_lat[slot], r.lon_px_per_deg = run->ppd_lon[slot];
        r.uploaded_bytes = run->uploaded_bytes[slot];
        if (run->pre_status[slot] != 0) {
            // box-first plan: the box pass found no valid pixel (2) or a pole in view (4); nothing was launched
            r.status = run->pre_status[slot];
            run->last.valid = run->prev.valid = false;
            continue;
        }
        amt_pipe_result pr;
        if (int rc = amt_pipe_wait(run->pipes[slot], &pr)) return rc;
        if (pr.status == 1 && pr.fused && !cfg.two_pass && pr.bbox[6] > 0 && pr.edge_pixels <= 16384) {
            // handed back although the launch was fused — the exact box does not fit the superset grid of a poor estimate,
            // the date line judged differently —: once more, with the exact box (in the coordinates of the plan: bbox[7])
            // as the estimate.  On the context's stream, behind the batch launched after this one; rare.
            const int mag = cfg.magnetic ? 1 : 0;
            run->outs[slot].altitude = run->alt[slot];
            if (int rc = amt_pipe_coarse_hint(run->pipes[slot], pr.bbox, mag)) return rc;
            if (int rc = amt_pipe_launch(run->pipes[slot], &run->prm[slot], &run->outs[slot], run->img[slot], cfg.img_dtype,
                                         cfg.min_elevation, run->ppd_lat[slot], run->ppd_lon[slot], -1, mag))
                return rc;
            if (int rc = amt_pipe_wait(run->pipes[slot], &pr)) return rc;
            r.retried = 1;
        }
        bool general = false;
        if (pr.status == 1) {
            // the two-pass plan, natively, when the frame's coordinate arrays exist and nothing else is needed
            if (int rc = amt_pipe_general_layout(run->pipes[slot], &pr)) return rc;
            general = pr.status == 0;
        }
        r.status = pr.status;
        r.two_pass = general ? 1 : 0;
        std::memcpy(r.bbox, pr.bbox, sizeof(r.bbox));
        r.edge_pixels = pr.edge_pixels;
        if (pr.status != 0) {
            run->last.valid = run->prev.valid = false;      // the next frame gets a real pre-pass
            continue;
        }
        const int64_t cells = (int64_t)pr.grid.ny * pr.grid.nx;
        const int64_t ib = image_bytes(run, cells);
        if (run->full || run->grid_used + 5 * cells > run->grids_capacity || run->image_used + ib > run->images_capacity) {
            // no room: this frame's accumulators are dropped with the frame (the driver zeroes them before its next
            // launch); the caller runs the rest again with larger arenas
            run->full = true;
            r.status = 3;
            continue;
        }
        r.ny = pr.grid.ny, r.nx = pr.grid.nx;
        r.contains_pole = pr.bbox[7] != 0 ? 1 : 0;
        r.lon_wrapped = pr.lon_wrapped;
        r.grid = pr.grid;
        r.grid_offset = run->grid_used;
        r.image_offset = run->image_used;
        double* f_mean = run->grids + run->grid_used;
        void* f_img = run->images + run->image_used;
        uint8_t* f_mask = reinterpret_cast<uint8_t*>(run->images + run->image_used + cells * 3 * (cfg.img_dtype == 2 ? 2 : 1));
        if (general) {
            if (int rc = amt_pipe_general_finalize(run->pipes[slot], f_mean, f_img, f_mask, f_mean + 4 * cells)) return rc;
            if (run->img[slot] == run->own_img[slot] && run->own_img[slot] != nullptr) {
                // the binning pass reads the slot's image buffer on the context's stream: the slot's next upload waits for it
                AMT_HIP(ctx, hipEventRecord(run->img_free[slot], ctx->stream));
                run->img_busy[slot] = 1;
            }
        } else {
            pp[m] = run->pipes[slot];
            mean[m] = f_mean, ocount[m] = f_mean + 4 * cells, oimg[m] = f_img, omask[m] = f_mask;
            ++m;
        }
        run->grid_used += 5 * cells;
        run->image_used += ib;
        run->prev = run->last;
        run->last.valid = true;
        std::memcpy(run->last.box, pr.bbox, sizeof(pr.bbox));
        run->last.p = run->prm[slot];
        run->last.index = run->frames_done + k;
    }
    if (m > 0)
        if (int rc = amt_pipe_finalize_many(pp, m, mean, oimg, omask, ocount)) return rc;
    return AMT_OK;
}

int run_launch(amt_run* run, int k0, int count) {
    amt_ctx* ctx = run->ctx;
    const amt_run_config& cfg = run->cfg;
    const int ns = cfg.n_slots;
    amt_pipe* pp[AMT_PIPE_MAX_BATCH];
    const amt_frame_params* prm[AMT_PIPE_MAX_BATCH];
    const amt_georef_out* oo[AMT_PIPE_MAX_BATCH];
    const void* ii[AMT_PIPE_MAX_BATCH];
    double la[AMT_PIPE_MAX_BATCH], lo[AMT_PIPE_MAX_BATCH];
    // the frames of the batch that are launched (box-first plan: not those without a valid pixel or with a pole in view),
    // as runs of consecutive frames
    int i = 0;
    while (i < count) {
        if (run->pre_status[(k0 + i) % ns] != 0) {
            ++i;
            continue;
        }
        int m = 0;
        while (i < count && run->pre_status[(k0 + i) % ns] == 0) {
            const int slot = (k0 + i) % ns;
            run->outs[slot].altitude = run->alt[slot];
            pp[m] = run->pipes[slot], prm[m] = &run->prm[slot], oo[m] = &run->outs[slot], ii[m] = run->img[slot];
            la[m] = run->ppd_lat[slot], lo[m] = run->ppd_lon[slot];
            if (ii[m] == nullptr) {
                ctx->last_error = "amt_run: a frame has no image";
                return AMT_EINVAL;
            }
            if (run->upload_pending[slot]) {
                // the frame's image rows are crossing the link on the copy stream (since its push, one batch ago)
                AMT_HIP(ctx, hipStreamWaitEvent(ctx->stream, run->uploaded[slot], 0));
                run->upload_pending[slot] = 0;
            }
            ++m, ++i;
        }
        if (int rc = amt_pipe_launch_many_res(pp, m, prm, oo, ii, cfg.img_dtype, cfg.min_elevation, la, lo, -1, cfg.magnetic ? 1 : 0))
            return rc;
    }
    return AMT_OK;
}

// ---- box-first plan (cfg.arcsec_per_px > 0) --------------------------------------------------------------------------
// box(b): ONE launch of the frame kernel without outputs for the batch's frames (amt_pipe_launch_box_many)
int run_box(amt_run* run, int k0, int count) {
    const amt_run_config& cfg = run->cfg;
    const int ns = cfg.n_slots;
    amt_pipe* pp[AMT_PIPE_MAX_BATCH];
    const amt_frame_params* prm[AMT_PIPE_MAX_BATCH];
    for (int i = 0; i < count; ++i) {
        const int slot = (k0 + i) % ns;
        pp[i] = run->pipes[slot], prm[i] = &run->prm[slot];
        run->pre_status[slot] = 0;
    }
    return amt_pipe_launch_box_many(pp, count, prm, cfg.min_elevation, cfg.magnetic ? 1 : 0);
}

// the boxes of a boxed batch -> px/deg per frame (BaseMapping.boundingBox of the reduction, reference mapping.py:711-741, then
// plateCarreeResolution, resample.py:36-61), the exact box as the estimate of the single-pass launch
int run_resolve(amt_run* run, int k0, int count) {
    const amt_run_config& cfg = run->cfg;
    const int ns = cfg.n_slots, mag = cfg.magnetic ? 1 : 0;
    for (int i = 0; i < count; ++i) {
        const int slot = (k0 + i) % ns;
        amt_pipe_result pr;
        if (int rc = amt_pipe_wait(run->pipes[slot], &pr)) return rc;
        const double* b = pr.bbox;
        if (pr.status == 2 || !(b[6] > 0)) {
            run->pre_status[slot] = 2;
            continue;
        }
        const bool straddles = b[3] - b[2] > 180;
        const double west = straddles ? b[4] : b[2], east = straddles ? b[5] : b[3];
        double la = 0, lo = 0;
        if (b[7] != 0 || !amt_gl::plate_carree_resolution(b[0], west, b[1], east, cfg.arcsec_per_px, &la, &lo) || !(lo > 0)) {
            run->pre_status[slot] = 4;
            continue;
        }
        run->ppd_lat[slot] = la, run->ppd_lon[slot] = lo;
        if (int rc = amt_pipe_coarse_hint(run->pipes[slot], b, mag)) return rc;
        run->hinted[slot] = 1;
    }
    return AMT_OK;
}

void pop_front(int (*q)[2], int* n) {
    q[0][0] = q[1][0], q[0][1] = q[1][1];
    --*n;
}

// box-first plan: the oldest boxed batch -> resolutions, its single-pass launch; then at most one batch stays in flight
int run_launch_boxed(amt_run* run) {
    const int k0 = run->boxed[0][0], count = run->boxed[0][1];
    pop_front(run->boxed, &run->n_boxed);
    if (int rc = run_resolve(run, k0, count)) return rc;
    if (!run->full)
        if (int rc = run_launch(run, k0, count)) return rc;
    // (arenas full: the frames are reported as not processed by run_finish's capacity check... they were not launched:
    // mark them here)
    if (run->full)
        for (int i = 0; i < count; ++i)
            if (run->pre_status[(k0 + i) % run->cfg.n_slots] == 0) run->pre_status[(k0 + i) % run->cfg.n_slots] = 3;
    run->launched[run->n_launched][0] = k0, run->launched[run->n_launched][1] = count;
    ++run->n_launched;
    while (run->n_launched > 1) {
        if (int rc = run_finish(run, run->launched[0][0], run->launched[0][1])) return rc;
        pop_front(run->launched, &run->n_launched);
    }
    return AMT_OK;
}

// box-first plan, a complete batch b of prepared frames: launch(b - 2), finish(b - 3), box(b)
int run_batch_ready_box(amt_run* run, int k0, int count) {
    if (run->n_boxed == 2)
        if (int rc = run_launch_boxed(run)) return rc;
    if (run->full) {
        for (int i = 0; i < count; ++i) {
            std::memset(&run->results[k0 + i], 0, sizeof(amt_run_result));
            run->results[k0 + i].status = 3;
        }
        return AMT_OK;
    }
    if (int rc = run_box(run, k0, count)) return rc;
    run->boxed[run->n_boxed][0] = k0, run->boxed[run->n_boxed][1] = count;
    ++run->n_boxed;
    return AMT_OK;
}

// a complete batch of prepared frames (k0, count): finish the older of two batches in flight, then launch it
int run_batch_ready(amt_run* run, int k0, int count) {
    if (run->cfg.arcsec_per_px > 0) return run_batch_ready_box(run, k0, count);
    if (run->n_launched == 2) {
        if (int rc = run_finish(run, run->launched[0][0], run->launched[0][1])) return rc;
        run->launched[0][0] = run->launched[1][0], run->launched[0][1] = run->launched[1][1];
        run->n_launched = 1;
    }
    if (run->full) {
        // the arenas ran full: frames that were only prepared are reported as not processed
        for (int i = 0; i < count; ++i) {
            std::memset(&run->results[k0 + i], 0, sizeof(amt_run_result));
            run->results[k0 + i].status = 3;
        }
        return AMT_OK;
    }
    if (int rc = run_launch(run, k0, count)) return rc;
    run->launched[run->n_launched][0] = k0, run->launched[run->n_launched][1] = count;
    ++run->n_launched;
    return AMT_OK;
}

}  // namespace

int amt_run_begin(amt_run* run, double* grids, int64_t grids_capacity, void* images, int64_t images_capacity,
                  amt_run_result* results, int32_t max_frames) {
    if (run == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = run->ctx;
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, !run->active, "amt_run_begin: a call is already in progress (amt_run_end it first)");
    AMT_REQUIRE(ctx, max_frames >= 0 && (max_frames == 0 || results != nullptr), "NULL argument");
    AMT_REQUIRE(ctx, max_frames == 0 || (grids != nullptr && images != nullptr), "the arenas are NULL");
    if (amt_set_device(ctx)) return AMT_EHIP;
    // the finalise kernels run on a stream of the drivers' own and write into the arenas: whatever the context's stream
    // has queued on that memory before this call must be done first
    void* fin = nullptr;
    if (int rc = amt_pipe_finalize_stream(run->pipes[0], &fin)) return rc;
    AMT_HIP(ctx, hipEventRecord(run->entry, ctx->stream));
    AMT_HIP(ctx, hipStreamWaitEvent(static_cast<hipStream_t>(fin), run->entry, 0));
    run->grids = grids, run->grids_capacity = grids_capacity;
    run->images = static_cast<char*>(images), run->images_capacity = images_capacity;
    run->results = results, run->max_frames = max_frames;
    run->grid_used = run->image_used = 0;
    run->full = false;
    run->n_pushed = 0, run->batch_k0 = 0, run->batch_count = 0, run->n_launched = 0, run->n_boxed = 0;
    run->active = true;
    return AMT_OK;
}

// A frame whose image lies in page-locked HOST memory: the rows of it that can be binned (amt_georef_image_rows: inside the limb
// and inside the cone of elevations >= min_elevation) go to the slot's own device buffer on the runner's copy stream, now — the
// frame is launched one batch later, behind the batch that is running —, and the launch waits for the event behind them.
static int run_upload(amt_run* run, int slot, const amt_frame_params& p, const void* host) {
    amt_ctx* ctx = run->ctx;
    const amt_run_config& cfg = run->cfg;
    const size_t row_bytes = (size_t)cfg.width * 3 * (cfg.img_dtype == 2 ? 2 : 1);
    if (run->copy_stream == nullptr) AMT_HIP(ctx, hipStreamCreateWithFlags(&run->copy_stream, hipStreamNonBlocking));
    if (run->own_img[slot] == nullptr) {
        if (hipMalloc(&run->own_img[slot], row_bytes * cfg.height) != hipSuccess) {
            (void)hipGetLastError();
            run->own_img[slot] = nullptr;
            ctx->last_error = "amt_run_push: no device memory for a slot's image buffer";
            return AMT_ENOMEM;
        }
        AMT_HIP(ctx, hipEventCreateWithFlags(&run->uploaded[slot], hipEventDisableTiming));
        AMT_HIP(ctx, hipEventCreateWithFlags(&run->img_free[slot], hipEventDisableTiming));
    }
    int32_t r0 = 0, r1 = cfg.height;
    if (int rc = amt_georef_image_rows(&p, cfg.min_elevation, &r0, &r1)) return rc;
    if (run->img_busy[slot]) {
        AMT_HIP(ctx, hipStreamWaitEvent(run->copy_stream, run->img_free[slot], 0));
        run->img_busy[slot] = 0;
    }
    if (r1 > r0) {
        // (one copy: the band in two or three pieces on as many streams is slower, 13.7 / 12.9 k against 15.7 k Mpixel/s —
        // profiles/REJECTED.md)
        AMT_HIP(ctx, hipMemcpyAsync(static_cast<char*>(run->own_img[slot]) + (size_t)r0 * row_bytes,
                                    static_cast<const char*>(host) + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes,
                                    hipMemcpyHostToDevice, run->copy_stream));
        AMT_HIP(ctx, hipEventRecord(run->uploaded[slot], run->copy_stream));
        run->upload_pending[slot] = 1;
    }
    run->uploaded_bytes[slot] = (int64_t)(r1 - r0) * (int64_t)row_bytes;
    run->img[slot] = run->own_img[slot];
    return AMT_OK;
}

static int run_push_impl(amt_run* run, const amt_run_frame* f);

int amt_run_push(amt_run* run, const amt_run_frame* f) {
    if (run == nullptr) return AMT_EINVAL;
    const int rc = run_push_impl(run, f);
    //  a failed push leaves no hints behind whose indices refer to frames that were never counted
    if (rc != AMT_OK) run->last.valid = run->prev.valid = false;
    return rc;
}

static int run_push_impl(amt_run* run, const amt_run_frame* f) {
    amt_ctx* ctx = run->ctx;
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, run->active, "amt_run_push without amt_run_begin");
    AMT_REQUIRE(ctx, f != nullptr, "NULL frame");
    AMT_REQUIRE(ctx, run->n_pushed < run->max_frames, "more frames than amt_run_begin announced");
    const amt_run_config& cfg = run->cfg;
    const int k = run->n_pushed++, slot = k % cfg.n_slots;
    std::memset(&run->results[k], 0, sizeof(amt_run_result));
    run->results[k].status = 3;                     // "not processed" until the frame is finished (also if this call fails)
    if (run->full) return AMT_OK;
    // the slot's previous frame (k - n_slots) may still be in flight: it is finished before its slot — parameters, shell,
    // driver — takes the new frame.  (With n_slots = 2 * batch that is the batch launched before the one now running, so
    // nothing waits that the loop `launch(B); finish(A); prepare(C)` would not wait for.)
    for (;;) {
        if (run->n_launched > 0 && run->launched[0][0] <= k - cfg.n_slots) {
            if (int rc = run_finish(run, run->launched[0][0], run->launched[0][1])) return rc;
            pop_front(run->launched, &run->n_launched);
            if (run->full) return AMT_OK;
            continue;
        }
        if (run->n_boxed > 0 && run->boxed[0][0] <= k - cfg.n_slots) {
            if (int rc = run_launch_boxed(run)) return rc;      // (cannot happen with n_slots >= 3 * batch)
            continue;
        }
        break;
    }
    const int mag = cfg.magnetic ? 1 : 0;
    const double altitude = f->altitude > 0 ? f->altitude : cfg.altitude;
    amt_frame_params& p = run->prm[slot];
    // (the J2000 -> SM matrix only where somebody reads it: MLat / MLT grids or arrays)
    const int want_sm = mag || run->outs[slot].mlat != nullptr || run->outs[slot].mlat_c != nullptr;
    int rc = amt_prm::frame_params(f, cfg.width, cfg.height, cfg.fast_center, altitude, want_sm, &p);
    if (rc != AMT_OK) {
        ctx->last_error = "amt_run_push: the date of a frame is outside the IGRF table";
        return rc;
    }
    run->alt[slot] = altitude;
    run->img[slot] = f->img;
    run->uploaded_bytes[slot] = 0;
    run->upload_pending[slot] = 0;
    if (f->img == nullptr && f->img_host != nullptr) {
        if (int rc2 = run_upload(run, slot, p, f->img_host)) return rc2;
    }
    run->pre_status[slot] = 0;
    if (!(cfg.arcsec_per_px > 0)) {
        double est[8];
        const bool hint = cfg.use_hints && box_hint(run, run->frames_done + k, p, est);
        rc = hint ? amt_pipe_coarse_hint(run->pipes[slot], est, mag) : amt_pipe_coarse(run->pipes[slot], &p, cfg.min_elevation, mag);
        if (rc != AMT_OK) return rc;
        run->hinted[slot] = hint ? 1 : 0;
    }
    if (run->batch_count == 0) run->batch_k0 = k;
    ++run->batch_count;
    // the first launch carries one frame only: the GPU starts after one frame's preparation instead of `batch`
    // (sizing the first launch so that the LAST one of the announced frames is full — 20 frames as 2 + 6 x 3 instead of
    // 1 + 6 x 3 + 1 — was measured in round 4: 287 against 251 us of non-kernel time per 20-frame call; not kept)
    if (run->batch_count == (k == 0 && !(cfg.arcsec_per_px > 0) ? 1 : cfg.batch)) {
        rc = run_batch_ready(run, run->batch_k0, run->batch_count);
        run->batch_count = 0;
    }
    return rc;
}

int amt_run_end(amt_run* run, int32_t* frames_done) {
    if (run == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = run->ctx;
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, run->active, "amt_run_end without amt_run_begin");
    run->active = false;
    int rc_all = AMT_OK;
    if (run->batch_count > 0) {
        rc_all = run_batch_ready(run, run->batch_k0, run->batch_count);
        run->batch_count = 0;
    }
    while (run->n_boxed > 0 && rc_all == AMT_OK) rc_all = run_launch_boxed(run);
    for (int i = 0; i < run->n_launched; ++i) {
        const int rc = run_finish(run, run->launched[i][0], run->launched[i][1]);
        if (rc_all == AMT_OK) rc_all = rc;
    }
    run->n_launched = run->n_boxed = 0;
    // host images are read "until amt_run_end returns": the upload of a frame that was pushed but never launched (arenas full, a
    // failed call) may still be in flight
    if (run->copy_stream != nullptr && hipStreamSynchronize(run->copy_stream) != hipSuccess && rc_all == AMT_OK) {
        (void)hipGetLastError();
        ctx->last_error = "amt_run_end: the copy stream failed";
        rc_all = AMT_EHIP;
    }
    std::fill(run->upload_pending.begin(), run->upload_pending.end(), 0);
    // order the context's stream behind every finalise kernel of this call
    for (amt_pipe* p : run->pipes) {
        const int rc = amt_pipe_join(p);
        if (rc_all == AMT_OK) rc_all = rc;
    }
    if (rc_all != AMT_OK) {
        //  a failed call leaves no half-valid state behind: no hints, nothing counted
        run->last.valid = run->prev.valid = false;
        if (frames_done) *frames_done = 0;
        return rc_all;
    }
    const int n = run->n_pushed;
    int done = n;
    for (int i = 0; i < n; ++i)
        if (run->results[i].status == 3) {
            done = i;
            break;
        }
    for (int i = done; i < n; ++i) run->results[i].status = 3;
    if (frames_done) *frames_done = done;
    run->frames_done += done;
    if (done < n) run->last.valid = run->prev.valid = false;
    return AMT_OK;
}

int amt_run_process(amt_run* run, const amt_run_frame* frames, int32_t n, double* grids, int64_t grids_capacity,
                    void* images, int64_t images_capacity, amt_run_result* results, int32_t* frames_done) {
    if (run == nullptr) return AMT_EINVAL;
    amt_ctx* ctx = run->ctx;
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, n >= 0 && (n == 0 || frames != nullptr), "NULL argument");
    if (frames_done) *frames_done = 0;
    if (n == 0) return AMT_OK;
    if (int rc = amt_run_begin(run, grids, grids_capacity, images, images_capacity, results, n)) return rc;
    int rc_all = AMT_OK;
    for (int i = 0; i < n && rc_all == AMT_OK; ++i) rc_all = amt_run_push(run, frames + i);
    const int rc = amt_run_end(run, frames_done);
    return rc_all != AMT_OK ? rc_all : rc;
}

}  // extern "C"
