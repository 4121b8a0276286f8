// Histogram binning for resample(method='mean')
// (reference auromat/util/histogram.py:57-282, auromat/resample.py:301-351).
//
// k_bin_frame: one workgroup bins a BW x BH tile of the image.  Neighbouring pixels fall into
// neighbouring cells, so the tile's cells form a small window of the output grid: the window is
// privatised in LDS (u32 count / channel sums, i64 fixed-point elevation), filled with LDS atomics
// and flushed with one 64-bit global integer atomic per touched cell and plane.  Integer
// accumulation is exact and order independent, so the result is bit-reproducible.
// Algorithmic HBM bytes: 24 B (lat, lon, elev) + nchan * sizeof(pixel) per pixel read; the grid is negligible.
#include "amt_common.h"

namespace {

using namespace amt;

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------
// generic float64 histogram (operator-level API)
// ------------------------------------------------------------------------------------------
constexpr int kMaxWeights = 8;

struct hist_args {
    const double* x;
    const double* y;
    int64_t n;
    const double* w[kMaxWeights];
    double* s[kMaxWeights];
    double* count;
    int nweights;
    axis_dev ax, ay;
    int lon_wrap;
};

__global__ __launch_bounds__(kBlock) void k_hist2d(hist_args A) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
        double xv = A.x[i];
        if (A.lon_wrap) xv = wrap180_shifted(xv);
        const int ix = bin_index(A.ax, xv);
        const int iy = bin_index(A.ay, A.y[i]);
        if (ix < 1 || ix > A.ax.nbin || iy < 1 || iy > A.ay.nbin) continue;
        const int64_t cell = (int64_t)(ix - 1) * A.ay.nbin + (iy - 1);
        unsafeAtomicAdd(&A.count[cell], 1.0);
        for (int k = 0; k < A.nweights; ++k) unsafeAtomicAdd(&A.s[k][cell], A.w[k][i]);
    }
}

__global__ void k_hist2d_finalize(const double* __restrict__ count, hist_args A, int nx, int ny,
                                  double* __restrict__ mean) {
    const int64_t n = (int64_t)nx * ny;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / nx), c = (int)(i - (int64_t)r * nx);
        const int64_t cell = (int64_t)c * ny + (ny - 1 - r);
        const double cnt = count[cell];
        for (int k = 0; k < A.nweights; ++k)
            mean[i * A.nweights + k] = cnt == 0.0 ? NAN : A.s[k][cell] / cnt;
    }
}

// ------------------------------------------------------------------------------------------
// fused frame binning
// ------------------------------------------------------------------------------------------
struct bin_args {
    const double* lat_c;
    const double* lon_c;
    const double* elev;
    const void* img;
    const uint8_t* mask;
    int height, width;
    double min_elev;
    int use_elev_threshold;
    axis_dev ax, ay;
    int lon_wrap;
    unsigned long long* acc;
};

constexpr int kPPT = 4;                      // consecutive pixels (along x) per thread and row, loaded as two pairs
constexpr int kBW = 64 * kPPT, kBH = kBlock / 64, kWCap = 1024;   // tile: 256 x 4 pixels, one image row per wave
// Column of pixel j of lane l inside the tile row.  (A layout with the two pairs 128 pixels apart, which makes
// every 16-byte wave load one contiguous 1 KiB segment, measured 25 % slower: 146 vs 118 us per frame.)
__device__ __forceinline__ int tile_col(int lane, int j) { return kPPT * lane + j; }

// Loads the thread's two pixel pairs of one row; `row` points at the tile's first pixel of that row, `n_row`
// is the number of pixels of the tile row inside the image.  VEC promises 16-byte alignment of `row`.
template <bool VEC>
__device__ __forceinline__ void load_run(const double* __restrict__ row, int lane, int n_row, double (&v)[kPPT]) {
#pragma unroll
    for (int k = 0; k < kPPT / 2; ++k) {
        const int c = tile_col(lane, 2 * k);
        if (VEC && c + 1 < n_row) {
            const double2 a = *reinterpret_cast<const double2*>(row + c);
            v[2 * k] = a.x;
            v[2 * k + 1] = a.y;
        } else {
            v[2 * k] = c < n_row ? row[c] : NAN;
            v[2 * k + 1] = c + 1 < n_row ? row[c + 1] : NAN;
        }
    }
}

constexpr int kWX = 32, kWY = 32;            // LDS window of kWX x kWY cells centred on the tile's anchor cell
constexpr int kRowIters = 4;                 // a workgroup walks kRowIters x kBH image rows (16) with one window
static_assert(kWX * kWY == kWCap, "window size");

template <typename IMG_T, int NCH, bool VEC>
__global__ __launch_bounds__(kBlock) void k_bin_frame(bin_args A) {
    __shared__ unsigned int sCnt[kWCap];
    __shared__ unsigned int sCh[NCH > 0 ? NCH : 1][kWCap];
    __shared__ unsigned long long sEl[kWCap];
    __shared__ int sCand[kBlock / 64];

    const int tiles_x = (A.width + kBW - 1) / kBW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx0 = tile_x * kBW;
    const int64_t ncell = (int64_t)A.ax.nbin * A.ay.nbin;
    const IMG_T* img = static_cast<const IMG_T*>(A.img);

    for (int i = threadIdx.x; i < kWCap; i += kBlock) {
        sCnt[i] = 0;
        sEl[i] = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) sCh[c][i] = 0;
    }
    // anchor cell (block-uniform): the window covers cells [ax0, ax0 + kWX) x [ay0, ay0 + kWY)
    int ax0 = 0, ay0 = 0;
    bool have_anchor = false;
    __syncthreads();

    for (int it = 0; it < kRowIters; ++it) {
        const int gy = (tile_y * kRowIters + it) * kBH + wave;
        const int n_row = (gy < A.height) ? min(kBW, A.width - gx0) : 0;   // pixels of this tile row inside the image

        // ---- all loads first (one memory latency per row group), then arithmetic ---------------------
        double la[kPPT], lo[kPPT], ev[kPPT];
        unsigned int ch[kPPT][NCH > 0 ? NCH : 1];
        unsigned char mk[kPPT];
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            la[j] = NAN; lo[j] = NAN; ev[j] = 0.0; mk[j] = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) ch[j][c] = 0;
        }
        if (n_row > 0) {
            const int64_t g0 = (int64_t)gy * A.width + gx0;
            load_run<VEC>(A.lat_c + g0, lane, n_row, la);
        }
        // a pixel without a latitude is not binned (resample.py:315-321): where a wave's whole tile row has none — the sky
        // above the limb, 30-40 % of an ISS frame — its longitudes, elevations and image bytes (22 of the 30 B per pixel)
        // are not read at all
        bool any_lat = false;
#pragma unroll
        for (int j = 0; j < kPPT; ++j) any_lat = any_lat || la[j] == la[j];
        if (n_row > 0 && __any(any_lat)) {
            const int64_t g0 = (int64_t)gy * A.width + gx0;
            load_run<VEC>(A.lon_c + g0, lane, n_row, lo);
            if (A.elev) load_run<VEC>(A.elev + g0, lane, n_row, ev);
            if (A.mask) {
#pragma unroll
                for (int j = 0; j < kPPT; ++j) mk[j] = tile_col(lane, j) < n_row ? A.mask[g0 + tile_col(lane, j)] : 1;
            }
            if (NCH > 0) {
                constexpr int kPairBytes = 2 * NCH * (int)sizeof(IMG_T);
#pragma unroll
                for (int k = 0; k < kPPT / 2; ++k) {
                    const int c0 = tile_col(lane, 2 * k);
                    const IMG_T* q = img + (g0 + c0) * NCH;
                    if (VEC && kPairBytes % 4 == 0 && c0 + 1 < n_row) {
                        // a pixel pair is kPairBytes contiguous, 4-byte aligned bytes (even column, even width)
                        constexpr int kWords = kPairBytes / 4;
                        const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
                        uint32_t buf[kWords > 0 ? kWords : 1];
#pragma unroll
                        for (int i = 0; i < kWords; ++i) buf[i] = w[i];
#pragma unroll
                        for (int e = 0; e < 2 * NCH; ++e) {
                            const unsigned int val = sizeof(IMG_T) == 2 ? (buf[e >> 1] >> ((e & 1) * 16)) & 0xffffu
                                                                        : (buf[e >> 2] >> ((e & 3) * 8)) & 0xffu;
                            ch[2 * k + e / NCH][e % NCH] = val;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int c = 0; c < NCH; ++c) ch[2 * k + e][c] = c0 + e < n_row ? q[e * NCH + c] : 0;
                    }
                }
            }
        }

        int cellx[kPPT], celly[kPPT];
        int first = 0;                                  // first valid cell of this thread, packed (x << 16 | y) + 1
#pragma unroll
        for (int j = kPPT - 1; j >= 0; --j) {
            bool ok = la[j] == la[j];                                          // resample.py:315-321
            if (A.use_elev_threshold) ok = ok && (ev[j] >= A.min_elev);        // mapping.py:856
            ok = ok && mk[j] == 0;
            cellx[j] = 0;
            celly[j] = 0;
            if (ok) {
                const double xv = A.lon_wrap ? wrap180_shifted(lo[j]) : lo[j];
                const int bx = bin_index(A.ax, xv), by = bin_index(A.ay, la[j]);
                if (bx >= 1 && bx <= A.ax.nbin && by >= 1 && by <= A.ay.nbin) {
                    cellx[j] = bx;
                    celly[j] = by;
                    first = -1 - j;   // the loop runs downwards: the smallest valid j wins
                }
            }
        }
        if (!have_anchor) {
            // elect the anchor: cell of the first valid pixel of the lowest wave that has one
            const unsigned long long m = __ballot(first < 0);
            int cand = 0;
            if (m) {
                const int src = __builtin_ctzll(m);
                const int j = -1 - __shfl(first, src);
                int cx = 0, cy = 0;
#pragma unroll
                for (int k = 0; k < kPPT; ++k) {
                    cx = (k == j) ? cellx[k] : cx;
                    cy = (k == j) ? celly[k] : cy;
                }
                cx = __shfl(cx, src);
                cy = __shfl(cy, src);
                cand = ((cx & 0xffff) << 16 | (cy & 0xffff)) + 1;     // grids have < 65535 bins per axis here
            }
            if (lane == 0) sCand[wave] = cand;
            __syncthreads();
            int chosen = 0;
#pragma unroll
            for (int w = kBlock / 64 - 1; w >= 0; --w) chosen = sCand[w] ? sCand[w] : chosen;
            __syncthreads();
            if (chosen) {
                have_anchor = true;
                ax0 = (((chosen - 1) >> 16) & 0xffff) - kWX / 2;
                ay0 = ((chosen - 1) & 0xffff) - kWY / 2;
            }
        }

        // consecutive pixels of a thread mostly share a cell: sum runs in registers, one atomic set per run.
        // (a NaN elevation of a kept pixel contributes 0; the reference would poison the cell — the mask
        //  invariants of mapping.py:299-316 make that unreachable)
        int run_x = 0, run_y = 0;
        unsigned int rcnt = 0, rch[NCH > 0 ? NCH : 1];
        long long rel = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) rch[c] = 0;
        auto flush = [&]() {
            const int dx = run_x - ax0, dy = run_y - ay0;
            if (dx >= 0 && dx < kWX && dy >= 0 && dy < kWY) {
                const int wi = dx * kWY + dy;
                atomicAdd(&sCnt[wi], rcnt);
#pragma unroll
                for (int c = 0; c < NCH; ++c) atomicAdd(&sCh[c][wi], rch[c]);
                atomicAdd(&sEl[wi], (unsigned long long)rel);
            } else {
                // outside the LDS window (very fine grids or strongly stretched tiles): global atomics
                const int64_t cell = (int64_t)(run_x - 1) * A.ay.nbin + (run_y - 1);
                atomicAdd(&A.acc[cell], (unsigned long long)rcnt);
#pragma unroll
                for (int c = 0; c < NCH; ++c) atomicAdd(&A.acc[(int64_t)(1 + c) * ncell + cell], (unsigned long long)rch[c]);
                atomicAdd(&A.acc[(int64_t)(1 + NCH) * ncell + cell], (unsigned long long)rel);
            }
        };
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            if (cellx[j] == 0) continue;
            if (cellx[j] != run_x || celly[j] != run_y) {
                if (run_x > 0) flush();
                run_x = cellx[j];
                run_y = celly[j];
                rcnt = 0;
                rel = 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) rch[c] = 0;
            }
            rcnt += 1;
#pragma unroll
            for (int c = 0; c < NCH; ++c) rch[c] += ch[j][c];
            rel += (ev[j] == ev[j]) ? __double2ll_rn(ev[j] * kFix) : 0;
        }
        if (run_x > 0) flush();
    }

    if (!have_anchor) return;      // nothing of this tile landed on the grid (block-uniform)
    __syncthreads();
    for (int i = threadIdx.x; i < kWCap; i += kBlock) {
        const unsigned int cnt = sCnt[i];
        if (cnt == 0) continue;
        const int dx = i / kWY, dy = i - dx * kWY;
        const int64_t cell = (int64_t)(ax0 + dx - 1) * A.ay.nbin + (ay0 + dy - 1);
        atomicAdd(&A.acc[cell], (unsigned long long)cnt);
#pragma unroll
        for (int c = 0; c < NCH; ++c) atomicAdd(&A.acc[(int64_t)(1 + c) * ncell + cell], (unsigned long long)sCh[c][i]);
        atomicAdd(&A.acc[(int64_t)(1 + NCH) * ncell + cell], sEl[i]);
    }
}

template <typename IMG_T>
__global__ void k_bin_finalize(unsigned long long* __restrict__ acc, int acc_nx, int acc_ny, int off_x,
                               int off_y, int nx, int ny, int nch, double* __restrict__ mean,
                               IMG_T* __restrict__ out_img, uint8_t* __restrict__ out_mask,
                               double* __restrict__ out_count, int clear) {
    const int64_t n = (int64_t)nx * ny, plane = (int64_t)acc_nx * acc_ny;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / nx), c = (int)(i - (int64_t)r * nx);
        // transpose + flipud (resample.py:339-349), inside the [off_x, off_x+nx) x [off_y, off_y+ny) window
        const int64_t cell = (int64_t)(off_x + c) * acc_ny + (off_y + ny - 1 - r);
        const unsigned long long cnt = acc[cell];
        const double dc = (double)cnt;
        for (int k = 0; k < nch; ++k) {
            const double m = cnt ? (double)acc[(int64_t)(1 + k) * plane + cell] / dc : NAN;
            if (mean) mean[i * (nch + 1) + k] = m;
            if (out_img) out_img[i * nch + k] = cnt ? (IMG_T)rint(m) : (IMG_T)0;   // np.round: half to even
        }
        if (mean) {
            const long long fx = (long long)acc[(int64_t)(1 + nch) * plane + cell];
            mean[i * (nch + 1) + nch] = cnt ? ((double)fx / kFix) / dc : NAN;
        }
        if (out_mask) out_mask[i] = cnt ? 0 : 1;
        if (out_count) out_count[i] = dc;
    }
    if (clear) {
        // leave the whole accumulator grid zeroed for the next frame (every thread clears only cells whose
        // values it read itself above, or cells outside the window that nobody reads)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const int r = (int)(i / nx), c = (int)(i - (int64_t)r * nx);
            const int64_t cell = (int64_t)(off_x + c) * acc_ny + (off_y + ny - 1 - r);
            for (int k = 0; k < nch + 2; ++k) acc[(int64_t)k * plane + cell] = 0;
        }
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
            const int cx = (int)(i / acc_ny), cy = (int)(i - (int64_t)cx * acc_ny);
            if (cx >= off_x && cx < off_x + nx && cy >= off_y && cy < off_y + ny) continue;
            for (int k = 0; k < nch + 2; ++k) acc[(int64_t)k * plane + i] = 0;
        }
    }
}

// On-edge pixels recorded by the fused binning (bin_event): now that the final grid is known — the window
// [off_x, off_x+nx) x [off_y, off_y+ny) of the accumulator grid — each goes into the bin the reference's
// right-most-edge rule gives it: a pixel on the lower edge of the first cell beyond the window sits on the LAST
// edge of the final grid and belongs to the last bin; everywhere else the edge is an interior one and the pixel
// stays in the cell above it.  One block; resets the counter for the next frame.
__global__ void k_apply_bin_events(const bin_event* __restrict__ events, unsigned int* __restrict__ count,
                                   unsigned long long* __restrict__ acc, int acc_nx, int acc_ny, int off_x, int off_y,
                                   int nx, int ny) {
    const unsigned int n = *count;
    const int64_t plane = (int64_t)acc_nx * acc_ny;
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) {
        const bin_event ev = events[i];
        int cx = ev.bx - 1, cy = ev.by - 1;
        if ((ev.flags & 1u) && cx == off_x + nx) cx -= 1;
        if ((ev.flags & 2u) && cy == off_y + ny) cy -= 1;
        if (cx < 0 || cx >= acc_nx || cy < 0 || cy >= acc_ny) continue;
        const int64_t cell = (int64_t)cx * acc_ny + cy;
        atomicAdd(&acc[cell], 1ull);
        atomicAdd(&acc[plane + cell], (unsigned long long)ev.c0);
        atomicAdd(&acc[2 * plane + cell], (unsigned long long)ev.c1);
        atomicAdd(&acc[3 * plane + cell], (unsigned long long)ev.c2);
        atomicAdd(&acc[4 * plane + cell], (unsigned long long)ev.el);
    }
    __syncthreads();
    if (threadIdx.x == 0) *count = 0;
}

// The finalise step of up to three frames of the single-pass driver in ONE launch (blockIdx.y = frame): crop + mean /
// image / mask / count as k_bin_finalize with clear = 1, and the on-edge pixels (k_apply_bin_events) folded in —
// instead of adding them to the accumulators first, every output cell adds the recorded pixels that the
// right-most-edge rule puts into it (there are a few dozen per frame; integer sums, so the result is the same).
// n_events is the host's copy of the counter (amt_pipe_wait has read it); the counter is reset for the next frame.
template <typename IMG_T>
__global__ void k_pipe_finish(finish_batch B) {
    const finish_frame& F = B.f[blockIdx.y];
    const int nch = 3;
    const int64_t n = (int64_t)F.nx * F.ny, plane = (int64_t)F.acc_nx * F.acc_ny;
    unsigned long long* __restrict__ acc = F.acc;
    IMG_T* __restrict__ out_img = static_cast<IMG_T*>(F.img);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / F.nx), c = (int)(i - (int64_t)r * F.nx);
        const int ax = F.off_x + c, ay = F.off_y + F.ny - 1 - r;
        const int64_t cell = (int64_t)ax * F.acc_ny + ay;
        unsigned long long cnt = acc[cell], s0 = acc[plane + cell], s1 = acc[2 * plane + cell], s2 = acc[3 * plane + cell];
        long long fx = (long long)acc[4 * plane + cell];
        for (unsigned int e = 0; e < F.n_events; ++e) {
            const bin_event ev = static_cast<const bin_event*>(F.events)[e];
            int cx = ev.bx - 1, cy = ev.by - 1;
            if ((ev.flags & 1u) && cx == F.off_x + F.nx) cx -= 1;
            if ((ev.flags & 2u) && cy == F.off_y + F.ny) cy -= 1;
            if (cx == ax && cy == ay) {
                cnt += 1;
                s0 += ev.c0, s1 += ev.c1, s2 += ev.c2;
                fx += ev.el;
            }
        }
        const double dc = (double)cnt;
        const unsigned long long sums[3] = {s0, s1, s2};
        for (int k = 0; k < nch; ++k) {
            const double m = cnt ? (double)sums[k] / dc : NAN;
            if (F.mean) F.mean[i * (nch + 1) + k] = m;
            if (out_img) out_img[i * nch + k] = cnt ? (IMG_T)rint(m) : (IMG_T)0;
        }
        if (F.mean) F.mean[i * (nch + 1) + nch] = cnt ? ((double)fx / kFix) / dc : NAN;
        if (F.mask) F.mask[i] = cnt ? 0 : 1;
        if (F.out_count) F.out_count[i] = dc;
    }
    // leave the whole accumulator grid zeroed (as k_bin_finalize: a thread clears the window cells it read itself, and
    // cells outside the window, which nobody reads)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / F.nx), c = (int)(i - (int64_t)r * F.nx);
        const int64_t cell = (int64_t)(F.off_x + c) * F.acc_ny + (F.off_y + F.ny - 1 - r);
        for (int k = 0; k < nch + 2; ++k) acc[(int64_t)k * plane + cell] = 0;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const int cx = (int)(i / F.acc_ny), cy = (int)(i - (int64_t)cx * F.acc_ny);
        if (cx >= F.off_x && cx < F.off_x + F.nx && cy >= F.off_y && cy < F.off_y + F.ny) continue;
        for (int k = 0; k < nch + 2; ++k) acc[(int64_t)k * plane + i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && F.count != nullptr) *F.count = 0;
}

inline dim3 grid_for(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return dim3(static_cast<unsigned>(blocks));
}

}  // namespace

int amt_bin_apply_events_on(amt_ctx* ctx, hipStream_t stream, const void* events, uint32_t* count, uint64_t* acc,
                            int32_t acc_nx, int32_t acc_ny, int32_t off_x, int32_t off_y, int32_t nx, int32_t ny) {
    AMT_REQUIRE(ctx, events && count && acc, "NULL argument");
    hipLaunchKernelGGL(k_apply_bin_events, dim3(1), dim3(kBlock), 0, stream, static_cast<const bin_event*>(events), count,
                       reinterpret_cast<unsigned long long*>(acc), acc_nx, acc_ny, off_x, off_y, nx, ny);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_bin_finalize_on(amt_ctx* ctx, hipStream_t stream, uint64_t* acc, int32_t acc_nx, int32_t acc_ny, int32_t off_x,
                        int32_t off_y, int32_t nx, int32_t ny, int32_t nchan, int32_t img_dtype, double* mean,
                        void* out_img, uint8_t* out_mask, double* out_count, int clear) {
    AMT_REQUIRE(ctx, acc != nullptr, "NULL argument");
    AMT_REQUIRE(ctx, nx > 0 && ny > 0 && nchan >= 0 && nchan <= 4, "bad shape");
    AMT_REQUIRE(ctx, off_x >= 0 && off_y >= 0 && off_x + nx <= acc_nx && off_y + ny <= acc_ny,
                "window outside the accumulator grid");
    AMT_REQUIRE(ctx, out_img == nullptr || img_dtype == 1 || img_dtype == 2, "img must be uint8 (1) or uint16 (2)");
    const dim3 grid = grid_for(clear ? (int64_t)acc_nx * acc_ny : (int64_t)nx * ny), block(kBlock);
    unsigned long long* a = reinterpret_cast<unsigned long long*>(acc);
    if (img_dtype == 1)
        hipLaunchKernelGGL((k_bin_finalize<uint8_t>), grid, block, 0, stream, a, acc_nx, acc_ny, off_x, off_y, nx, ny,
                           nchan, mean, static_cast<uint8_t*>(out_img), out_mask, out_count, clear);
    else
        hipLaunchKernelGGL((k_bin_finalize<uint16_t>), grid, block, 0, stream, a, acc_nx, acc_ny, off_x, off_y, nx, ny,
                           nchan, mean, static_cast<uint16_t*>(out_img), out_mask, out_count, clear);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_pipe_finish_on(amt_ctx* ctx, hipStream_t stream, const finish_batch& B, int32_t img_dtype) {
    AMT_REQUIRE(ctx, B.n >= 1 && B.n <= 3, "bad batch");
    AMT_REQUIRE(ctx, img_dtype == 1 || img_dtype == 2, "img must be uint8 (1) or uint16 (2)");
    int64_t most = 1;
    for (int i = 0; i < B.n; ++i) {
        const finish_frame& F = B.f[i];
        AMT_REQUIRE(ctx, F.acc != nullptr && F.nx > 0 && F.ny > 0, "bad frame");
        AMT_REQUIRE(ctx, F.off_x >= 0 && F.off_y >= 0 && F.off_x + F.nx <= F.acc_nx && F.off_y + F.ny <= F.acc_ny,
                    "window outside the accumulator grid");
        AMT_REQUIRE(ctx, F.n_events == 0 || F.events != nullptr, "events missing");
        most = std::max<int64_t>(most, (int64_t)F.acc_nx * F.acc_ny);
    }
    dim3 grid = grid_for(most);
    grid.y = (unsigned)B.n;
    if (img_dtype == 1)
        hipLaunchKernelGGL((k_pipe_finish<uint8_t>), grid, dim3(kBlock), 0, stream, B);
    else
        hipLaunchKernelGGL((k_pipe_finish<uint16_t>), grid, dim3(kBlock), 0, stream, B);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

extern "C" {

int amt_hist2d_accumulate(amt_ctx* ctx, const double* x, const double* y, int64_t n, const double* const* weights,
                          int32_t nweights, const amt_axis* xaxis, const amt_axis* yaxis, int lon_wrap,
                          double* count, double* const* sums) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, xaxis && yaxis && count, "NULL argument");
    AMT_REQUIRE(ctx, n >= 0 && nweights >= 0 && nweights <= kMaxWeights, "bad n or too many weight arrays (max 8)");
    AMT_REQUIRE(ctx, nweights == 0 || (weights && sums), "weights/sums missing");
    AMT_REQUIRE(ctx, axis_ok(xaxis) && axis_ok(yaxis), "bad axis");
    if (n == 0) return AMT_OK;
    AMT_REQUIRE(ctx, x && y, "NULL coordinates");
    hist_args A;
    A.x = x;
    A.y = y;
    A.n = n;
    A.count = count;
    A.nweights = nweights;
    for (int k = 0; k < kMaxWeights; ++k) {
        A.w[k] = k < nweights ? weights[k] : nullptr;
        A.s[k] = k < nweights ? sums[k] : nullptr;
        AMT_REQUIRE(ctx, k >= nweights || (A.w[k] && A.s[k]), "NULL weight or sum array");
    }
    make_axis(xaxis, &A.ax);
    make_axis(yaxis, &A.ay);
    A.lon_wrap = lon_wrap ? 1 : 0;
    hipLaunchKernelGGL(k_hist2d, grid_for(n), dim3(kBlock), 0, ctx->stream, A);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_hist2d_finalize_mean(amt_ctx* ctx, const double* count, const double* const* sums, int32_t nweights,
                             int32_t nx, int32_t ny, double* mean) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, count && mean && (nweights == 0 || sums), "NULL argument");
    AMT_REQUIRE(ctx, nx > 0 && ny > 0 && nweights >= 0 && nweights <= kMaxWeights, "bad shape");
    hist_args A = {};
    A.nweights = nweights;
    for (int k = 0; k < nweights; ++k) A.s[k] = const_cast<double*>(sums[k]);
    hipLaunchKernelGGL(k_hist2d_finalize, grid_for((int64_t)nx * ny), dim3(kBlock), 0, ctx->stream, count, A, nx, ny,
                       mean);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_bin_frame(amt_ctx* ctx, const double* lat_c, const double* lon_c, const double* elev, const void* img,
                  int32_t img_dtype, int32_t nchan, const uint8_t* center_mask, int32_t height, int32_t width,
                  double min_elevation, const amt_axis* xaxis, const amt_axis* yaxis, int lon_wrap, uint64_t* acc) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, lat_c && lon_c && xaxis && yaxis && acc, "NULL argument");
    AMT_REQUIRE(ctx, height > 0 && width > 0, "empty frame");
    AMT_REQUIRE(ctx, nchan >= 0 && nchan <= 4, "nchan must be 0..4");
    AMT_REQUIRE(ctx, nchan == 0 || (img && (img_dtype == 1 || img_dtype == 2)), "img must be uint8 (1) or uint16 (2)");
    AMT_REQUIRE(ctx, axis_ok(xaxis) && axis_ok(yaxis), "bad axis");
    AMT_REQUIRE(ctx, xaxis->nbin < 65535 && yaxis->nbin < 65535, "at most 65534 bins per axis");
    bin_args A;
    A.lat_c = lat_c;
    A.lon_c = lon_c;
    A.elev = elev;
    A.img = img;
    A.mask = center_mask;
    A.height = height;
    A.width = width;
    A.min_elev = min_elevation;
    A.use_elev_threshold = (elev != nullptr) && !(std::isinf(min_elevation) && min_elevation < 0);
    make_axis(xaxis, &A.ax);
    make_axis(yaxis, &A.ay);
    A.lon_wrap = lon_wrap ? 1 : 0;
    A.acc = reinterpret_cast<unsigned long long*>(acc);
    const int tiles_x = (width + kBW - 1) / kBW, tiles_y = (height + kBH * kRowIters - 1) / (kBH * kRowIters);
    const dim3 grid((unsigned)((int64_t)tiles_x * tiles_y)), block(kBlock);
    auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    // vector path: rows start 16-byte aligned (even width) so that every pixel pair is one aligned 16-byte load
    const bool vec = (width % 2 == 0) && aligned16(lat_c) && aligned16(lon_c) && (elev == nullptr || aligned16(elev));
#define AMT_BIN_CASE(T, N)                                                                  \
    do {                                                                                    \
        if (vec) hipExtLaunchKernelGGL((k_bin_frame<T, N, true>), grid, block, 0, ctx->stream, t0, t1, 0, A);  \
        else hipExtLaunchKernelGGL((k_bin_frame<T, N, false>), grid, block, 0, ctx->stream, t0, t1, 0, A);     \
    } while (0)
    const bool u8 = img_dtype == 1;
    hipEvent_t t0, t1;
    amt_timing_pair(ctx, AMT_KERNEL_BIN, 1, &t0, &t1);
    switch (nchan) {
        case 0: AMT_BIN_CASE(uint8_t, 0); break;
        case 1: if (u8) AMT_BIN_CASE(uint8_t, 1); else AMT_BIN_CASE(uint16_t, 1); break;
        case 2: if (u8) AMT_BIN_CASE(uint8_t, 2); else AMT_BIN_CASE(uint16_t, 2); break;
        case 3: if (u8) AMT_BIN_CASE(uint8_t, 3); else AMT_BIN_CASE(uint16_t, 3); break;
        default: if (u8) AMT_BIN_CASE(uint8_t, 4); else AMT_BIN_CASE(uint16_t, 4); break;
    }
#undef AMT_BIN_CASE
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

int amt_bin_frame_finalize_window(amt_ctx* ctx, const uint64_t* acc, int32_t acc_nx, int32_t acc_ny, int32_t off_x,
                                  int32_t off_y, int32_t nx, int32_t ny, int32_t nchan, int32_t img_dtype,
                                  double* mean, void* out_img, uint8_t* out_mask, double* out_count) {
    AMT_CHECK_CTX(ctx);
    // clear == 0: the accumulators are only read
    return amt_bin_finalize_on(ctx, ctx->stream, const_cast<uint64_t*>(acc), acc_nx, acc_ny, off_x, off_y, nx, ny, nchan,
                               img_dtype, mean, out_img, out_mask, out_count, 0);
}

int amt_bin_frame_finalize(amt_ctx* ctx, const uint64_t* acc, int32_t nx, int32_t ny, int32_t nchan,
                           int32_t img_dtype, double* mean, void* out_img, uint8_t* out_mask, double* out_count) {
    return amt_bin_frame_finalize_window(ctx, acc, nx, ny, 0, 0, nx, ny, nchan, img_dtype, mean, out_img, out_mask,
                                         out_count);
}

}  // extern "C"
