// Write bandwidth of the row kernel's store pattern on its own, and of variations of it (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern.hip -o build/store_pattern && build/store_pattern
// Frame 4240 x 2832: five f64 arrays (two of (H+1) x (W+1) corners, three of H x W pixels) = 480 MB per pass.
// A wave owns a strip of STRIP pixel columns and marches down ROWS rows, storing one 8-byte value per lane, row and
// array — what k_georef_rows does.  Variants: strip width 63 (the kernel's: 504-B runs that never start on a 128-B
// line) or 64 (aligned), rows per work item, 16-byte stores (two columns per lane and half the lanes ... as a bound),
// arrays padded to a 512-B pitch, and the dispatch order of the work items.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e));                   \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

struct arrays {
    double* a[5];
    int pitch[5];       // elements per row
    int rows[5];
};

// MODE 0: 8 B per lane (lane = column); MODE 1: 16 B per lane, lanes 0..31 write two columns each (1 row per instruction)
// MODE 2: 16 B per lane, all 64 lanes, two ROWS per instruction (lane pairs own (row, row+1) x two columns)
template <int STRIP, int MODE>
__global__ __launch_bounds__(256) void k_store(arrays A, int width, int height, int rows_per_item, int strips_x, int n_items,
                                                int narr, int spin) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int chunk = item / strips_x, strip = item - chunk * strips_x;
    const int x0 = strip * STRIP, y0 = chunk * rows_per_item;
    const int rows = min(rows_per_item, height - y0);
    double v = (double)item;
    if (MODE == 0) {
        const int gx = x0 + lane;
        const bool ok = lane < STRIP && gx < width;
        for (int r = 0; r < rows; ++r) {
            int nspin = spin;
            if (spin < 0) {      // desynchronised waves: a different amount of arithmetic per wave and row
                unsigned h = (unsigned)item * 2654435761u + (unsigned)r * 40503u;
                h ^= h >> 15;
                nspin = (int)(h % (unsigned)(-spin));
                nspin = __builtin_amdgcn_readfirstlane(nspin);
            }
            for (int k = 0; k < nspin; ++k) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v));
            if (ok) {
#pragma unroll
                for (int a = 0; a < 5; ++a)
                    if (a < narr) A.a[a][(size_t)(y0 + r) * A.pitch[a] + gx] = v;
            }
        }
    } else if (MODE == 1) {
        const int gx = x0 + 2 * lane;
        const bool ok = 2 * lane + 1 < STRIP && gx + 1 < width;
        for (int r = 0; r < rows; ++r) {
            for (int k = 0; k < spin; ++k) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v));
            if (ok) {
#pragma unroll
                for (int a = 0; a < 5; ++a)
                    if (a < narr) *reinterpret_cast<double2*>(&A.a[a][(size_t)(y0 + r) * A.pitch[a] + gx]) = make_double2(v, v);
            }
        }
    } else {
        const int gx = x0 + (lane & ~1);
        const bool ok = (lane | 1) < STRIP && gx + 1 < width;
        for (int r = 0; r + 1 < rows + 1; r += 2) {
            for (int k = 0; k < 2 * spin; ++k) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v));
            const int row = y0 + r + (lane & 1);
            if (ok && row < y0 + rows) {
#pragma unroll
                for (int a = 0; a < 5; ++a)
                    if (a < narr) *reinterpret_cast<double2*>(&A.a[a][(size_t)row * A.pitch[a] + gx]) = make_double2(v, v);
            }
        }
    }
}

template <int STRIP, int MODE>
double run(const arrays& A, int width, int height, int rows_per_item, int narr, int spin, const char* what) {
    const int strips_x = (width + STRIP - 1) / STRIP, chunks_y = (height + rows_per_item - 1) / rows_per_item;
    const int n_items = strips_x * chunks_y;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_store<STRIP, MODE>), dim3((n_items + 3) / 4), dim3(256), 0, 0, A, width, height, rows_per_item, strips_x,
                           n_items, narr, spin);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) {
            sum += ms;
            best = ms < best ? ms : best;
        }
    }
    double bytes = 0;
    for (int a = 0; a < narr; ++a) bytes += (double)width * height * 8;
    std::printf("%-64s strip %2d rows %3d arrays %d spin %3d: mean %.4f ms (best %.4f) = %.2f TB/s\n", what, STRIP, rows_per_item, narr,
                spin, sum / 6, best, bytes / (sum / 6 * 1e-3) / 1e12);
    return sum / 6;
}

int main() {
    const int W = 4240, H = 2832;
    arrays A, P;
    for (int a = 0; a < 5; ++a) {
        CHECK(hipMalloc(&A.a[a], (size_t)(W + 1) * (H + 1) * 8 + 4096));
        A.pitch[a] = W;
        A.rows[a] = H;
        const int pp = (W + 63) / 64 * 64;      // 512-byte pitch
        CHECK(hipMalloc(&P.a[a], (size_t)pp * (H + 1) * 8 + 4096));
        P.pitch[a] = pp;
        P.rows[a] = H;
    }
    for (int spin : {0, 60, 120}) {
        run<63, 0>(A, W, H, 16, 5, spin, "kernel's pattern: 63 columns, 8 B per lane");
        run<64, 0>(A, W, H, 16, 5, spin, "64 columns (512-B runs; rows start at multiples of 33920 B)");
        run<64, 0>(P, W, H, 16, 5, spin, "64 columns, pitch padded to 512 B (every run line-aligned)");
        run<63, 0>(A, W, H, 64, 5, spin, "63 columns, 64 rows per item");
        run<64, 1>(P, W, H, 16, 5, spin, "64 columns, 16 B per lane x 32 lanes, padded pitch");
        run<64, 2>(P, W, H, 16, 5, spin, "64 columns, 16 B per lane, two rows per instruction, padded");
        run<63, 2>(A, W, H, 16, 5, spin, "62 of 63 columns, 16 B per lane, two rows per instruction");
    }
    for (int spin : {-240, -480, -960}) {
        run<63, 0>(A, W, H, 16, 5, spin, "63 columns, 8 B per lane, DESYNCHRONISED waves (random spin)");
        run<64, 0>(P, W, H, 16, 5, spin, "64 columns padded pitch, desynchronised");
        run<63, 0>(A, W, H, 4, 5, spin, "63 columns, 4 rows per item, desynchronised");
        run<63, 0>(A, W, H, 64, 5, spin, "63 columns, 64 rows per item, desynchronised");
    }
    run<63, 0>(A, W, H, 16, 1, 0, "one array only");
    run<63, 0>(A, W, H, 16, 3, 0, "three arrays");
    run<63, 0>(A, W, H, 8, 5, 0, "63 columns, 8 rows per item");
    run<63, 0>(A, W, H, 32, 5, 0, "63 columns, 32 rows per item");
    run<63, 0>(A, W, H, 2832, 5, 0, "63 columns, whole column per item (68 waves)");
    return 0;
}
