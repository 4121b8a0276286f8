/* Plain C99 client of include/auromat_hip.h: no Python, no torch, no HIP headers.  Georeferences one frame whose
 * amt_frame_params block is read from a binary file (written by the caller with the struct's layout) and writes
 * lat, lon, lat_c, lon_c, elev, the 8-number bounding-box reduction and the 0.1-degree-style binned counts of a
 * two-pass resample to a binary output file.  tests/test_gpu_frames.py builds and runs it.
 *
 *   cc -std=c99 -Iinclude examples/c_abi_demo.c -Lauromat_amd/lib -lauromat_hip -Wl,-rpath,$PWD/auromat_amd/lib -lm
 *   ./a.out params.bin image_u16.bin px_per_deg out.bin
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

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
    if (argc != 5) {
        fprintf(stderr, "usage: %s params.bin image_u16.bin px_per_deg out.bin\n", argv[0]);
        return 2;
    }
    amt_frame_params* p = (amt_frame_params*)read_file(argv[1], sizeof(amt_frame_params));
    const double ppd = atof(argv[3]);
    const size_t np = (size_t)p->width * p->height, nc = (size_t)(p->width + 1) * (p->height + 1);
    unsigned short* img = (unsigned short*)read_file(argv[2], np * 3 * sizeof(unsigned short));

    CHECK(amt_ctx_create(0, NULL, 1, &ctx));                       /* device 0, a stream owned by the library */
    double *lat, *lon, *lat_c, *lon_c, *elev, *bbox;
    void* d_img;
    CHECK(amt_malloc(ctx, nc * 8, (void**)&lat));
    CHECK(amt_malloc(ctx, nc * 8, (void**)&lon));
    CHECK(amt_malloc(ctx, np * 8, (void**)&lat_c));
    CHECK(amt_malloc(ctx, np * 8, (void**)&lon_c));
    CHECK(amt_malloc(ctx, np * 8, (void**)&elev));
    CHECK(amt_malloc(ctx, 8 * 8, (void**)&bbox));
    CHECK(amt_malloc(ctx, np * 6, &d_img));
    CHECK(amt_memcpy_h2d(ctx, d_img, img, np * 6));

    amt_georef_out out = {0};
    out.lat = lat, out.lon = lon, out.lat_c = lat_c, out.lon_c = lon_c, out.elev = elev, out.bbox = bbox;
    out.bbox_min_elevation = 10.0;                                  /* maskedByElevation(10) */
    CHECK(amt_georef_frame(ctx, p, &out));

    double h_bbox[8];
    CHECK(amt_memcpy_d2h(ctx, h_bbox, bbox, sizeof h_bbox));        /* synchronises */
    if (h_bbox[6] == 0) {
        fprintf(stderr, "minElevation=10 would mask all pixels!\n");
        return 3;
    }
    /* grid for the box (no pole, no date line in this demo), then the separate binning pass */
    amt_grid grid;
    CHECK(amt_grid_layout(ppd, ppd, h_bbox[0], h_bbox[1], h_bbox[2], h_bbox[3], &grid));
    const size_t cells = (size_t)grid.nx * grid.ny;
    uint64_t* acc;
    double *mean, *count;
    CHECK(amt_malloc(ctx, cells * 5 * 8, (void**)&acc));
    CHECK(amt_malloc(ctx, cells * 4 * 8, (void**)&mean));
    CHECK(amt_malloc(ctx, cells * 8, (void**)&count));
    CHECK(amt_memset(ctx, acc, 0, cells * 5 * 8));
    CHECK(amt_bin_frame(ctx, lat_c, lon_c, elev, d_img, 2, 3, NULL, p->height, p->width, 10.0, &grid.xaxis, &grid.yaxis,
                        0, acc));
    CHECK(amt_bin_frame_finalize(ctx, acc, grid.nx, grid.ny, 3, 2, mean, NULL, NULL, count));

    FILE* fp = fopen(argv[4], "wb");
    if (!fp) return 2;
    double* host = (double*)malloc((nc > cells * 4 ? nc : cells * 4) * 8);
    double* arrays[5] = {lat, lon, lat_c, lon_c, elev};
    for (int k = 0; k < 5; ++k) {
        const size_t n = k < 2 ? nc : np;
        CHECK(amt_memcpy_d2h(ctx, host, arrays[k], n * 8));
        fwrite(host, 8, n, fp);
    }
    fwrite(h_bbox, 8, 8, fp);
    const double dims[2] = {(double)grid.ny, (double)grid.nx};
    fwrite(dims, 8, 2, fp);
    CHECK(amt_memcpy_d2h(ctx, host, count, cells * 8));
    fwrite(host, 8, cells, fp);
    CHECK(amt_memcpy_d2h(ctx, host, mean, cells * 4 * 8));
    fwrite(host, 8, cells * 4, fp);
    fclose(fp);

    void* all[] = {lat, lon, lat_c, lon_c, elev, bbox, d_img, acc, mean, count};
    for (size_t k = 0; k < sizeof all / sizeof all[0]; ++k) CHECK(amt_free(ctx, all[k]));
    CHECK(amt_ctx_destroy(ctx));
    printf("ok %d x %d frame, grid %d x %d\n", (int)p->width, (int)p->height, (int)grid.ny, (int)grid.nx);
    free(host), free(img), free(p);
    return 0;
}
