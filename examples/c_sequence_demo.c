/* Plain C99 host of a SEQUENCE: no Python, no torch, no HIP headers.  Reads n frame descriptors (amt_run_frame records:
 * the WCS cards, the camera position, the photo time as a Julian date) and their uint16 RGB images, runs the whole
 * sequence through the native runner — host scalars, box hints, launches, waits, grid layouts and finalise kernels are
 * the library's (amt_run_*) — and writes every frame's result record followed by the arena of the grids
 * (mean | count of all frames back to back: the payload of the gather's wire format as it stands).  What the reference
 * does with `map(getMapping, ...)` + `resample` per frame (mapping/spacecraft.py:326-332, cli/convert.py:178-185).
 * tests/test_gpu_sequence.py builds and runs it and compares the grids with the Python host's.
 *
 *   cc -std=c99 -Iinclude examples/c_sequence_demo.c -Lauromat_amd/lib -lauromat_hip -Wl,-rpath,$PWD/auromat_amd/lib -lm
 *   ./a.out frames.bin images_u16.bin n width height px_per_deg out.bin
 * A NEGATIVE px_per_deg is a resolution in arcsec per pixel — `auromat-convert --resolution R`, the reference's own call form
 * `resample(mapping, arcsecPerPx=R)` (cli/convert.py:176-185): every frame's px/deg then follows from its own bounding box
 * (amt_run_config.arcsec_per_px: the box-first plan; the pair a frame was binned at is in its result record).
 * An eighth argument `host`: the images stay in page-locked HOST memory (amt_malloc_host) and every frame names its image there
 * (amt_run_frame.img_host) — the runner sends the rows of each that can be binned, on a copy stream of its own, one batch ahead
 * of the frame's launch: what a real sequence does with the image it has just read (reference cli/convert.py:178-216).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "auromat_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != AMT_OK) {                                                               \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? amt_last_error(ctx) : ""); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static void* read_file(const char* path, size_t bytes) {
    FILE* fp = fopen(path, "rb");
    void* buf = malloc(bytes);
    if (!fp || !buf || fread(buf, 1, bytes, fp) != bytes) {
        fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, path);
        exit(2);
    }
    fclose(fp);
    return buf;
}

int main(int argc, char** argv) {
    amt_ctx* ctx = NULL;
    if (argc != 8 && !(argc == 9 && strcmp(argv[8], "host") == 0)) {
        fprintf(stderr, "usage: %s frames.bin images_u16.bin n width height px_per_deg out.bin [host]\n", argv[0]);
        return 2;
    }
    const int host_images = argc == 9;
    const int n = atoi(argv[3]), width = atoi(argv[4]), height = atoi(argv[5]);
    const double ppd = atof(argv[6]);
    const size_t img_bytes = (size_t)width * height * 6;
    amt_run_frame* frames = (amt_run_frame*)read_file(argv[1], (size_t)n * sizeof(amt_run_frame));
    unsigned char* images = (unsigned char*)read_file(argv[2], (size_t)n * img_bytes);

    CHECK(amt_ctx_create(0, NULL, 1, &ctx));                       /* device 0, a stream owned by the library */
    void* d_images = NULL;
    void* h_images = NULL;
    if (host_images) {
        /* (a reader would fill the page-locked buffer directly) */
        CHECK(amt_malloc_host(ctx, (size_t)n * img_bytes, &h_images));
        memcpy(h_images, images, (size_t)n * img_bytes);
        for (int k = 0; k < n; ++k) frames[k].img = NULL, frames[k].img_host = (const char*)h_images + (size_t)k * img_bytes;
    } else {
        CHECK(amt_malloc(ctx, (size_t)n * img_bytes, &d_images));
        CHECK(amt_upload_staged(ctx, d_images, images, (size_t)n * img_bytes));      /* pageable memory at the link's rate */
        for (int k = 0; k < n; ++k) frames[k].img = (const char*)d_images + (size_t)k * img_bytes, frames[k].img_host = NULL;
    }

    /* grids only: no per-pixel array is written (NULL pointers in every slot) */
    enum { BATCH = 3, SLOTS = 3 * BATCH };      /* 2 x BATCH for a fixed px/deg, 3 x BATCH for the box-first plan */
    amt_georef_out slots[SLOTS];
    memset(slots, 0, sizeof slots);
    amt_run_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.width = width, cfg.height = height, cfg.img_dtype = 2, cfg.fast_center = 1, cfg.magnetic = 0;
    cfg.batch = BATCH, cfg.use_hints = 1, cfg.n_slots = SLOTS;
    cfg.altitude = 110.0, cfg.min_elevation = 10.0;
    if (ppd < 0) cfg.arcsec_per_px = -ppd; else cfg.lat_px_per_deg = cfg.lon_px_per_deg = ppd;
    cfg.slots = slots;
    amt_run* run = NULL;
    CHECK(amt_run_create(ctx, &cfg, &run));

    const int64_t cells_per_frame = 1 << 16;
    const int64_t gcap = 5 * cells_per_frame * n, icap = (7 * cells_per_frame + 256) * n;
    double* grids;
    void* imgs;
    CHECK(amt_malloc(ctx, (size_t)gcap * 8, (void**)&grids));
    CHECK(amt_malloc(ctx, (size_t)icap, &imgs));
    amt_run_result* res = (amt_run_result*)calloc((size_t)n, sizeof(amt_run_result));
    int32_t done = 0;
    CHECK(amt_run_process(run, frames, n, grids, gcap, imgs, icap, res, &done));
    CHECK(amt_ctx_synchronize(ctx));
    if (done != n) {
        fprintf(stderr, "the arenas were too small: %d of %d frames\n", (int)done, n);
        return 3;
    }
    int64_t used = 0, uploaded = 0;
    int hinted = 0;
    for (int k = 0; k < n; ++k) {
        if (res[k].status == 0) used = res[k].grid_offset + 5 * (int64_t)res[k].ny * res[k].nx;
        hinted += res[k].hinted;
        uploaded += res[k].uploaded_bytes;
    }
    double* host = (double*)malloc((size_t)(used > 0 ? used : 1) * 8);
    CHECK(amt_memcpy_d2h(ctx, host, grids, (size_t)used * 8));
    FILE* fp = fopen(argv[7], "wb");
    if (!fp) return 2;
    fwrite(res, sizeof(amt_run_result), (size_t)n, fp);
    fwrite(host, 8, (size_t)used, fp);
    fclose(fp);
    CHECK(amt_run_destroy(run));
    CHECK(amt_free(ctx, grids));
    CHECK(amt_free(ctx, imgs));
    CHECK(amt_free(ctx, d_images));
    CHECK(amt_free_host(ctx, h_images));
    CHECK(amt_ctx_destroy(ctx));
    printf("ok %d frames of %d x %d, %d without a pre-pass, %lld doubles of grids, %lld of %lld image bytes sent by the runner\n", n, width,
           height, hinted, (long long)used, (long long)uploaded, (long long)(host_images ? (int64_t)n * (int64_t)img_bytes : 0));
    free(host), free(res), free(images), free(frames);
    return 0;
}
