// Store-pattern microbenchmark for the row kernel's outputs (round 5): the real launch shape (177 rows of work items x 68 strips of
// 63 pixel columns, 16 pixel rows per item, 4 strips per workgroup, five arrays of doubles), a tunable amount of dependent FP64
// work per row in place of the ray cast, and two ways of writing a row:
//   A  every wave writes its own 63 x 8-byte run into each of the five arrays (what k_georef_rows does)
//   B  the four waves of a workgroup stage their values in LDS; after a barrier wave w writes the workgroup's whole
//      252-pixel run (2016 bytes) of array w (the fifth array is split over the four waves)
// beside the work alone and the stores spread over a row's work, at full occupancy and at the real kernel's four waves per SIMD.
// usage (GPU box): hipcc -O3 --offload-arch=gfx950 -o /tmp/store_pattern tools/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

constexpr int W = 4240, H = 2832, ROWS = 16, STRIPS = 68, CHUNKS = 177, NARR = 5;

__device__ __forceinline__ double work(double x, int spin) {
    // eight independent chains of FP64 multiply-adds: issue-bound at four waves per SIMD, as a row of the real kernel is
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = x + k;
    for (int i = 0; i < spin; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], 0.9999999, 1e-7);
    }
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// MODE 0: A, stores at the end of a row   1: B, staged   2: no stores (the work alone)   3: A with the five stores spread over the row's work
template <int MODE>
__global__ __launch_bounds__(256) void k_pattern(double* __restrict__ a0, double* __restrict__ a1, double* __restrict__ a2,
                                                 double* __restrict__ a3, double* __restrict__ a4, int spin, int first_earth_chunk, int stride, int sw, int corner_pitch = W) {
    __shared__ double stage[2][NARR][4 * 63 + 4];
    extern __shared__ double occupancy_limiter[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    const int strips = sw == 63 ? STRIPS : (W + sw - 1) / sw;
    if (item >= CHUNKS * strips) return;
    const int chunk_in_order = item / strips, strip = item - chunk_in_order * strips;
    const int chunk = chunk_in_order * stride % CHUNKS;     // stride 1: sky first, then Earth; 76: the two kinds alternate
    const int x0 = strip * sw, y0 = chunk * ROWS;
    const int gx = x0 + lane;
    const bool ok = lane < sw && gx < W;
    double* arr[NARR] = {a0, a1, a2, a3, a4};
    if (spin < 0) occupancy_limiter[threadIdx.x] = 0.0;          // (keeps the dynamic allocation alive; never taken)
    if (chunk < first_earth_chunk) {
        // "sky": contiguous fill of the chunk's rows, 16 bytes per lane (as the real kernel's sky path)
        const long long first = (long long)y0 * W, count = (long long)ROWS * W;
        for (int k = 0; k < NARR; ++k) {
            long long a = first + count * strip / strips, b = first + count * (strip + 1) / strips;
            a &= ~1ll, b &= ~1ll;
            const double2 two = {1.0, 2.0};
            for (long long i = a + 2 * lane; i + 1 < b; i += 128) *reinterpret_cast<double2*>(arr[k] + i) = two;
        }
        return;
    }
    double v = (double)gx;
    for (int r = 0; r < ROWS && y0 + r < H; ++r) {
        const long long row = (long long)(y0 + r) * W;
        if (MODE == 3) {
            for (int k = 0; k < NARR; ++k) {
                v = work(v, spin / NARR);
                if (ok) arr[k][row + gx] = v + k;
            }
            continue;
        }
        v = work(v, spin);
        if (MODE == 0) {
            if (ok) {
                const long long crow = (long long)(y0 + r) * corner_pitch;
                arr[0][crow + gx] = v, arr[1][crow + gx] = v + 1;
#pragma unroll
                for (int k = 2; k < NARR; ++k) arr[k][row + gx] = v + k;
            }
        } else if (MODE == 2) {
            if (ok && v == 12345.678) arr[0][row + gx] = v;      // never true: the work stays live, nothing is written
        } else {
            const int buf = r & 1;
            if (lane < 63) {
#pragma unroll
                for (int k = 0; k < NARR; ++k) stage[buf][k][wave * 63 + lane] = v + k;
            }
            __syncthreads();
            // the workgroup's run: pixels [gx0, gx0 + 252) of this row
            const int gx0 = (blockIdx.x * 4 - chunk_in_order * STRIPS) * 63;
            const int n = min(252, W - gx0);
            auto put = [&](int k, int from, int to) {            // elements [from, to) of array k's run, 2 per lane per pass
                for (int i = from + 2 * lane; i < to; i += 128) {
                    arr[k][row + gx0 + i] = stage[buf][k][i];
                    if (i + 1 < to) arr[k][row + gx0 + i + 1] = stage[buf][k][i + 1];
                }
            };
            put(wave, 0, n);
            put(4, wave * 63, min(n, wave * 63 + 63));
        }
    }
}

template <int MODE>
static float run(double** a, int spin, int sky, unsigned lds, int stride, int sw = 63, int corner_pitch = W) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = (CHUNKS * (sw == 63 ? STRIPS : (W + sw - 1) / sw) + 3) / 4, reps = 12;
    float sum = 0;
    for (int rep = 0; rep < reps + 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_pattern<MODE>, dim3(blocks), dim3(256), lds, 0, a[0], a[1], a[2], a[3], a[4], spin, sky, stride, sw, corner_pitch);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) sum += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return sum / reps * 1e3f;
}


// Round 6: output layouts.  The same launch (items of 63 or 64 pixel columns x 16 rows, sky / Earth rows alternating, four waves per
// SIMD), every array with its own pitch, a strip's columns at `slot` doubles per strip in memory:
//   L0  the kernel of rounds 1-5: 63-column strips, contiguous rows (pitch W for the pixel arrays, W + 1 for the corner arrays)
//   L1  VERDICT r5 item 1: 64-column strips, all five arrays at a pitch of 4288 doubles (every run 512 bytes on a 512-byte boundary)
//   L2  strip-padded rows: 63-column strips (the kernel's own work items), strip s of a row at doubles [64 s, 64 s + 64) of a row of
//       68 x 64 = 4352 doubles, all 64 lanes store (a corner array's 64th value is the next strip's first corner, a pixel array's
//       a pad): every run 512 bytes on a 512-byte boundary with no change to the arithmetic
//   L3  as L2, lane 63 does not store (504-byte runs on 512-byte boundaries)
__global__ __launch_bounds__(256) void k_layout(double* __restrict__ a0, double* __restrict__ a1, double* __restrict__ a2,
                                                double* __restrict__ a3, double* __restrict__ a4, int spin, int first_earth_chunk, int stride,
                                                int sw, int slot, int pitch_corner, int pitch_pixel, int store_lanes, int with_stores,
                                                int dist_fill = 0) {
    extern __shared__ double occupancy_limiter[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    const int strips = (W + sw - 1) / sw;
    if (item >= CHUNKS * strips) return;
    const int chunk_in_order = item / strips, strip = item - chunk_in_order * strips;
    const int chunk = chunk_in_order * stride % CHUNKS;
    const int y0 = chunk * ROWS;
    const int col = strip * slot + lane;                           // column in memory
    double* arr[NARR] = {a0, a1, a2, a3, a4};
    const int pitch[NARR] = {pitch_corner, pitch_corner, pitch_pixel, pitch_pixel, pitch_pixel};
    if (spin < 0) occupancy_limiter[threadIdx.x] = 0.0;
    if (chunk < first_earth_chunk) {
        if (dist_fill) return;            // (the Earth items write the sky rows, a slice after each of their own rows)
        for (int k = 0; k < NARR; ++k) {
            const long long first = (long long)y0 * pitch[k], count = (long long)ROWS * pitch[k];
            long long a = first + count * strip / strips, b = first + count * (strip + 1) / strips;
            a &= ~1ll, b &= ~1ll;
            const double2 two = {1.0, 2.0};
            for (long long i = a + 2 * lane; i + 1 < b; i += 128) *reinterpret_cast<double2*>(arr[k] + i) = two;
        }
        return;
    }
    const bool ok = lane < store_lanes && col < pitch_pixel && strip * sw + lane < W + (store_lanes == 64 && slot == 64 && sw == 63 ? 1 : 0);
    double v = (double)col;
    const long long n_e = (long long)(CHUNKS - first_earth_chunk) * strips, e = (long long)(chunk - first_earth_chunk) * strips + strip;
    // (this wave's share of the sky rows of every array, in units of 128-byte lines; computed once: 64-bit divisions are slow)
    unsigned int l0[NARR], ln[NARR];
    if (dist_fill) {
        for (int k = 0; k < NARR; ++k) {
            const long long lines = (long long)first_earth_chunk * ROWS * pitch[k] / 16;
            const long long a = lines * e / n_e, b = lines * (e + 1) / n_e;
            l0[k] = (unsigned int)a, ln[k] = (unsigned int)(b - a);
        }
    }
    for (int r = 0; r < ROWS && y0 + r < H; ++r) {
        v = work(v, spin);
        if (with_stores) {
            if (ok) {
#pragma unroll
                for (int k = 0; k < NARR; ++k) arr[k][(long long)(y0 + r) * pitch[k] + col] = v + k;
            }
            if (dist_fill) {
                // this wave's share of the sky rows of every array, one sixteenth of it per row step, whole lines
                for (int k = 0; k < NARR; ++k) {
                    const unsigned int a = l0[k] + ((ln[k] * (unsigned int)r) >> 4), b = l0[k] + ((ln[k] * (unsigned int)(r + 1)) >> 4);
                    const double2 two = {1.0, 2.0};
                    // a line is 8 lanes x 16 bytes: the wave writes 8 lines per pass
                    for (unsigned int line = a + (lane >> 3); line < b; line += 8)
                        *reinterpret_cast<double2*>(arr[k] + (size_t)line * 16 + 2 * (lane & 7)) = two;
                }
            }
        } else if (ok && v == 12345.678) arr[0][col] = v;
    }
}

struct layout { const char* name; int sw, slot, pitch_corner, pitch_pixel, store_lanes; };

static float run_layout(double** a, const layout& L, int spin, int sky, int with_stores, int reps, int dist_fill = 0, int stride = 76) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = (CHUNKS * ((W + L.sw - 1) / L.sw) + 3) / 4;
    std::vector<float> t;
    for (int rep = 0; rep < reps + 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_layout, dim3(blocks), dim3(256), 28000u, 0, a[0], a[1], a[2], a[3], a[4], spin, sky, stride, L.sw, L.slot,
                           L.pitch_corner, L.pitch_pixel, L.store_lanes, with_stores, dist_fill);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) t.push_back(ms * 1e3f);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

// Round 6, second question: WHEN the sky rows are written.  Padded layout L2 throughout; (a) sky rows of items first, then the Earth's
// (the kernel's order), (b) the two kinds alternating, (c) no sky items at all: every Earth wave writes a sixteenth of its share of the
// sky after each of its own rows (the fill's traffic spread evenly over the launch).
static void fill_schedules(double** a, int sky) {
    const layout L2 = {"L2", 63, 64, 4352, 4352, 64};
    std::printf("when the sky is written (padded rows; median of 5 passes x 9 launches; us)\n");
    std::printf("  spin | work alone | sky first | alternating | spread over the Earth waves\n");
    for (int spin : {0, 20, 25, 30, 35, 40}) {
        std::vector<float> m[4];
        for (int pass = 0; pass < 5; ++pass) {
            m[0].push_back(run_layout(a, L2, spin, sky, 0, 9));
            m[1].push_back(run_layout(a, L2, spin, sky, 1, 9, 0, 1));
            m[2].push_back(run_layout(a, L2, spin, sky, 1, 9, 0, 76));
            m[3].push_back(run_layout(a, L2, spin, sky, 1, 9, 1, 1));
        }
        auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        std::printf("  %4d | %10.1f | %9.1f | %11.1f | %9.1f\n", spin, med(m[0]), med(m[1]), med(m[2]), med(m[3]));
    }
}

static void layouts(double** a, int sky) {
    const layout Ls[] = {
        {"L0 63-col strips, contiguous (W+1 / W)", 63, 63, W + 1, W, 63},
        {"L1 64-col strips, pitch 4288", 64, 64, 4288, 4288, 64},
        {"L2 63-col strips padded to 64, pitch 4352, 64 lanes store", 63, 64, 4352, 4352, 64},
        {"L3 63-col strips padded to 64, pitch 4352, 63 lanes store", 63, 64, 4352, 4352, 63},
        {"L4 63-col strips, contiguous, all at pitch W", 63, 63, W, W, 63},
    };
    const int nL = sizeof(Ls) / sizeof(Ls[0]);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_layout), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::printf("layouts (median of 5 passes x 9 launches, passes interleaved over the layouts; us)\n");
    for (int k = 0; k < nL; ++k) std::printf("  %s\n", Ls[k].name);
    std::printf("  spin | work alone L0 / L1 |   L0   |   L1   |   L2   |   L3   |   L4\n");
    for (int spin : {0, 20, 25, 30, 35, 40}) {
        std::vector<float> m[nL], w[2];
        for (int pass = 0; pass < 5; ++pass) {
            for (int k = 0; k < nL; ++k) m[k].push_back(run_layout(a, Ls[k], spin, sky, 1, 9));
            for (int k = 0; k < 2; ++k) w[k].push_back(run_layout(a, Ls[k], spin, sky, 0, 9));
        }
        auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        std::printf("  %4d | %8.1f / %8.1f |", spin, med(w[0]), med(w[1]));
        for (int k = 0; k < nL; ++k) std::printf(" %6.1f |", med(m[k]));
        std::printf("\n");
    }
}

int main() {
    const size_t n = (size_t)W * H;
    double* a[NARR];
    for (int k = 0; k < NARR; ++k) (void)hipMalloc(&a[k], ((size_t)4352 * (H + 2) + 64) * sizeof(double));
    const int sky = (int)(0.43 * CHUNKS);
    if (std::getenv("FILL_ONLY")) { fill_schedules(a, sky); return 0; }
    if (std::getenv("LAYOUTS_ONLY")) { layouts(a, sky); return 0; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pattern<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pattern<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pattern<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pattern<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::printf("five arrays of %d x %d doubles = %.1f MB; %d of %d rows of work items are sky\n", W, H, NARR * n * 8 / 1e6, sky, CHUNKS);
    std::printf("all sky (contiguous fill): %.1f us\n", run<0>(a, 0, CHUNKS + 1, 0, 1));
    for (int stride : {1, 76})
        for (unsigned lds : {0u, 28000u}) {                      // 28000 + 10 KB static: four workgroups = four waves per SIMD
            std::printf("%s, dynamic LDS %u bytes per workgroup\n", stride == 1 ? "sky rows first" : "sky and Earth rows alternate", lds);
            std::printf("  spin | work alone | A at row end | A spread | B staged  (us)\n");
            for (int spin : {0, 10, 20, 25, 30, 35, 40, 50, 60}) {
                const float w = run<2>(a, spin, sky, lds, stride), s0 = run<0>(a, spin, sky, lds, stride),
                            s3 = run<3>(a, spin, sky, lds, stride), s1 = run<1>(a, spin, sky, lds, stride);
                std::printf("  %4d | %10.1f | %12.1f | %8.1f | %8.1f\n", spin, w, s0, s3, s1);
            }
        }
    // strips of 64 pixel columns: every run is 512 bytes and starts on a 512-byte boundary of a pixel array's row (4240 x 8 bytes
    // per row = 66.25 runs: the rows themselves start on 64-byte boundaries only)
    std::printf("sky and Earth rows alternate, four waves per SIMD: own runs of 63 columns (504 bytes) against 64 columns (512 bytes, aligned)\n");
    std::printf("  spin | work alone 63 / 64 | A at row end 63 / 64  (us)\n");
    for (int rep = 0; rep < 2; ++rep)
        for (int spin : {0, 20, 30, 40}) {
            const float w63 = run<2>(a, spin, sky, 28000u, 76, 63), w64 = run<2>(a, spin, sky, 28000u, 76, 64);
            const float s63 = run<0>(a, spin, sky, 28000u, 76, 63), s64 = run<0>(a, spin, sky, 28000u, 76, 64);
            const float c63 = run<0>(a, spin, sky, 28000u, 76, 63, W + 1), c64 = run<0>(a, spin, sky, 28000u, 76, 64, W + 1);
            std::printf("  %4d | %8.1f / %8.1f | %8.1f / %8.1f | two of the five arrays at a pitch of W + 1 (corner arrays): %8.1f / %8.1f\n", spin,
                        w63, w64, s63, s64, c63, c64);
        }
    layouts(a, sky);
    return 0;
}
