// Host <-> device copies of whole arrays at the rate of the link (include/auromat_hip.h: amt_upload_staged,
// amt_download_staged): the host boundary of the class API, where the reference's users hand NumPy arrays in and take NumPy
// arrays out (reference mapping/mapping.py:318-337, resample.py:73-157).
//
// A copy between PAGEABLE host memory and the device goes through a staging buffer either way; the runtime's own stages at
// 5-8 GB/s, one host thread's memcpy into page-locked memory at 12-28 GB/s, the link moves 55 GB/s.  Here a few worker threads
// each take every T-th piece of the array: copy it into a page-locked piece of their own (two per worker, so the next memcpy
// runs while the DMA engine drains the previous one) and hand it to the DMA engine on the context's stream — no
// synchronisation between the workers at all, the stream orders the transfers against the kernels before and after.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>

#include "amt_common.h"

#ifndef AMT_COPY_PIECE_MB
#define AMT_COPY_PIECE_MB 4u
#endif

struct amt_copier {
    static constexpr int kMaxWorkers = 16, kBuffers = 3;
    static constexpr size_t kPiece = AMT_COPY_PIECE_MB << 20;
    int n_workers;
    void* stage[kMaxWorkers][kBuffers];
    hipEvent_t done[kMaxWorkers][kBuffers];
    bool used[kMaxWorkers][kBuffers];
    // the transfers go on two streams of the copier's own, alternating by worker, so that the gap between two pieces on one
    // stream (~15 us per 4 MiB piece: 43 GB/s) is covered by the other's transfer; they start behind what the context's
    // stream holds at the call (entry) and the context's stream continues behind them (leave)
    static constexpr int kStreams = 2;
    hipStream_t streams[kStreams];
    hipEvent_t entry, leave[kStreams];
    // the pool: workers sleep between jobs; a job is one function run once per worker
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable wake, finished;
    std::function<void(int)> job;
    unsigned long long generation;
    int remaining;
    bool quit;
    std::atomic<int> failed;
};

namespace {

// Pieces: every worker's FIRST piece is a quarter of the others, so that the link starts after a quarter of a piece's memcpy
// (a 4 MiB memcpy takes 0.3 ms of one thread: a third of the whole transfer of a 36 MB image)
struct piece_layout {
    size_t bytes, small, big, n_small, n_pieces;
    piece_layout(size_t total, int workers) : bytes(total), small(amt_copier::kPiece / 4), big(amt_copier::kPiece) {
        n_small = (size_t)workers;
        const size_t head = n_small * small;
        n_pieces = total <= head ? (total + small - 1) / small : n_small + (total - head + big - 1) / big;
    }
    size_t offset(size_t k) const { return k < n_small ? k * small : n_small * small + (k - n_small) * big; }
    size_t size(size_t k) const {
        const size_t off = offset(k), full = k < n_small ? small : big;
        return bytes - off < full ? bytes - off : full;
    }
};

void worker_loop(amt_copier* c, int index, int device) {
    (void)hipSetDevice(device);
    unsigned long long seen = 0;
    for (;;) {
        std::function<void(int)> job;
        {
            std::unique_lock<std::mutex> lock(c->m);
            c->wake.wait(lock, [&] { return c->quit || c->generation != seen; });
            if (c->quit) return;
            seen = c->generation;
            job = c->job;
        }
        job(index);
        {
            std::lock_guard<std::mutex> lock(c->m);
            if (--c->remaining == 0) c->finished.notify_one();
        }
    }
}

// run fn(worker) on every worker (worker 0 is the calling thread) and wait
void run_on_workers(amt_copier* c, const std::function<void(int)>& fn) {
    {
        std::lock_guard<std::mutex> lock(c->m);
        c->job = fn;
        c->remaining = c->n_workers - 1;
        ++c->generation;
    }
    c->wake.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lock(c->m);
    c->finished.wait(lock, [&] { return c->remaining == 0; });
}

amt_copier* copier_of(amt_ctx* ctx) {
    if (ctx->copier != nullptr) return ctx->copier;
    amt_copier* c = new (std::nothrow) amt_copier();
    if (c == nullptr) return nullptr;
    // AMT_COPY_THREADS: host threads that copy (default 8, at most 16 and half of this rank's share of the host's cores —
    // amt_host_threads: the cores this process may run on divided by the ranks on the node —: 43 GB/s with 4, 50 with 8)
    const char* e = std::getenv("AMT_COPY_THREADS");
    int n = e ? std::atoi(e) : 8;
    const int share = amt_host_threads(1 << 20, nullptr, nullptr);
    if (n > (share + 1) / 2) n = (share + 1) / 2;
    c->n_workers = n < 1 ? 1 : (n > amt_copier::kMaxWorkers ? amt_copier::kMaxWorkers : n);
    c->generation = 0, c->remaining = 0, c->quit = false;
    c->failed = 0;
    bool ok = hipEventCreateWithFlags(&c->entry, hipEventDisableTiming) == hipSuccess;
    for (int s = 0; s < amt_copier::kStreams; ++s)
        ok = ok && hipStreamCreateWithFlags(&c->streams[s], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&c->leave[s], hipEventDisableTiming) == hipSuccess;
    for (int w = 0; w < c->n_workers; ++w)
        for (int b = 0; b < amt_copier::kBuffers; ++b) {
            c->stage[w][b] = nullptr, c->done[w][b] = nullptr, c->used[w][b] = false;
            ok = ok && hipHostMalloc(&c->stage[w][b], amt_copier::kPiece, hipHostMallocDefault) == hipSuccess &&
                 hipEventCreateWithFlags(&c->done[w][b], hipEventDisableTiming) == hipSuccess;
        }
    if (!ok) {
        for (int w = 0; w < c->n_workers; ++w)
            for (int b = 0; b < amt_copier::kBuffers; ++b) {
                if (c->stage[w][b]) (void)hipHostFree(c->stage[w][b]);
                if (c->done[w][b]) (void)hipEventDestroy(c->done[w][b]);
            }
        delete c;
        return nullptr;
    }
    for (int w = 1; w < c->n_workers; ++w) c->threads.emplace_back(worker_loop, c, w, ctx->device);
    ctx->copier = c;
    return c;
}

}  // namespace

void amt_copier_destroy(amt_copier* c) {
    if (c == nullptr) return;
    {
        std::lock_guard<std::mutex> lock(c->m);
        c->quit = true;
    }
    c->wake.notify_all();
    for (std::thread& t : c->threads) t.join();
    for (int w = 0; w < c->n_workers; ++w)
        for (int b = 0; b < amt_copier::kBuffers; ++b) {
            if (c->used[w][b]) (void)hipEventSynchronize(c->done[w][b]);
            (void)hipHostFree(c->stage[w][b]);
            (void)hipEventDestroy(c->done[w][b]);
        }
    for (int s = 0; s < amt_copier::kStreams; ++s) {
        (void)hipStreamSynchronize(c->streams[s]);
        (void)hipStreamDestroy(c->streams[s]);
        (void)hipEventDestroy(c->leave[s]);
    }
    (void)hipEventDestroy(c->entry);
    delete c;
}

namespace {

// the copier's streams start behind the context's stream ...
int copier_enter(amt_ctx* ctx, amt_copier* c) {
    AMT_HIP(ctx, hipEventRecord(c->entry, ctx->stream));
    for (int s = 0; s < amt_copier::kStreams; ++s) AMT_HIP(ctx, hipStreamWaitEvent(c->streams[s], c->entry, 0));
    return AMT_OK;
}

// ... and the context's stream continues behind them
int copier_leave(amt_ctx* ctx, amt_copier* c) {
    for (int s = 0; s < amt_copier::kStreams; ++s) {
        AMT_HIP(ctx, hipEventRecord(c->leave[s], c->streams[s]));
        AMT_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->leave[s], 0));
    }
    return AMT_OK;
}

}  // namespace


// Strip-padded rows -> contiguous rows (amt_georef_out.row_layout): element x of a row comes from 64 (x / 63) + x % 63.  A thread
// moves two consecutive elements of the destination row (rows of width + 1 doubles start at odd multiples of 8 bytes: the pair is
// two 8-byte stores; the reads of a wave are two runs of 63 doubles).
__global__ __launch_bounds__(256) void k_unpad_rows(const double* __restrict__ src, int64_t pitch, int cols, double* __restrict__ dst) {
    const int x = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (x >= cols) return;
    const double* s = src + (int64_t)blockIdx.y * pitch;
    double* d = dst + (int64_t)blockIdx.y * cols;
    const int q0 = x / 63;
    d[x] = s[q0 * 64 + (x - q0 * 63)];
    if (x + 1 < cols) {
        const int q1 = (x + 1) / 63;
        d[x + 1] = s[q1 * 64 + (x + 1 - q1 * 63)];
    }
}

extern "C" {

int amt_upload_staged(amt_ctx* ctx, void* dst_device, const void* src_host, size_t bytes) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, bytes == 0 || (dst_device != nullptr && src_host != nullptr), "NULL argument");
    if (bytes == 0) return AMT_OK;
    if (amt_set_device(ctx)) return AMT_EHIP;
    amt_copier* c = copier_of(ctx);
    if (c == nullptr) {
        ctx->last_error = "amt_upload_staged: staging buffers could not be allocated";
        return AMT_ENOMEM;
    }
    const piece_layout L(bytes, c->n_workers);
    const size_t n_pieces = L.n_pieces;
    c->failed = 0;
    if (int rc = copier_enter(ctx, c)) return rc;
    char* dst = static_cast<char*>(dst_device);
    const char* src = static_cast<const char*>(src_host);
    run_on_workers(c, [&](int w) {
        hipStream_t stream = c->streams[w % amt_copier::kStreams];
        int turn = 0;
        for (size_t k = (size_t)w; k < n_pieces; k += (size_t)c->n_workers, ++turn) {
            const int b = turn % amt_copier::kBuffers;
            if (c->used[w][b] && hipEventSynchronize(c->done[w][b]) != hipSuccess) c->failed = 1;      // its last transfer is out
            const size_t off = L.offset(k), n = L.size(k);
            std::memcpy(c->stage[w][b], src + off, n);
            if (hipMemcpyAsync(dst + off, c->stage[w][b], n, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipEventRecord(c->done[w][b], stream) != hipSuccess)
                c->failed = 1;
            c->used[w][b] = true;
        }
    });
    if (int rc = copier_leave(ctx, c)) return rc;
    if (c->failed) {
        ctx->last_error = "amt_upload_staged: a transfer failed";
        return AMT_EHIP;
    }
    return AMT_OK;
}

int amt_download_staged(amt_ctx* ctx, void* dst_host, const void* src_device, size_t bytes) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, bytes == 0 || (dst_host != nullptr && src_device != nullptr), "NULL argument");
    if (bytes == 0) return AMT_OK;
    if (amt_set_device(ctx)) return AMT_EHIP;
    amt_copier* c = copier_of(ctx);
    if (c == nullptr) {
        ctx->last_error = "amt_download_staged: staging buffers could not be allocated";
        return AMT_ENOMEM;
    }
    const piece_layout L(bytes, c->n_workers);
    const size_t n_pieces = L.n_pieces;
    c->failed = 0;
    if (int rc = copier_enter(ctx, c)) return rc;
    char* dst = static_cast<char*>(dst_host);
    const char* src = static_cast<const char*>(src_device);
    run_on_workers(c, [&](int w) {
        hipStream_t stream = c->streams[w % amt_copier::kStreams];
        // a worker keeps one transfer in flight while it copies the piece before it out of its other buffer
        auto issue = [&](size_t k, int b) {
            if (c->used[w][b] && hipEventSynchronize(c->done[w][b]) != hipSuccess) c->failed = 1;
            const size_t off = L.offset(k), n = L.size(k);
            if (hipMemcpyAsync(c->stage[w][b], src + off, n, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipEventRecord(c->done[w][b], stream) != hipSuccess)
                c->failed = 1;
            c->used[w][b] = true;
        };
        int turn = 0;
        size_t k = (size_t)w;
        if (k < n_pieces) issue(k, 0);
        for (; k < n_pieces; k += (size_t)c->n_workers, ++turn) {
            const int b = turn % amt_copier::kBuffers;
            const size_t next = k + (size_t)c->n_workers;
            if (next < n_pieces) issue(next, (turn + 1) % amt_copier::kBuffers);
            if (hipEventSynchronize(c->done[w][b]) != hipSuccess) c->failed = 1;
            c->used[w][b] = false;
            const size_t off = L.offset(k), n = L.size(k);
            std::memcpy(dst + off, c->stage[w][b], n);
        }
    });
    if (c->failed) {
        ctx->last_error = "amt_download_staged: a transfer failed";
        return AMT_EHIP;
    }
    return AMT_OK;
}


int amt_unpad_rows(amt_ctx* ctx, const double* src_padded, int32_t rows, int32_t cols, int32_t width, double* dst) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, src_padded && dst, "NULL argument");
    AMT_REQUIRE(ctx, rows >= 0 && width > 0 && (cols == width || cols == width + 1), "cols must be width or width + 1");
    if (rows == 0) return AMT_OK;
    const int64_t pitch = amt_padded_pitch(width);
    const int pairs = (cols + 1) / 2;
    const dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(k_unpad_rows, grid, dim3(256), 0, ctx->stream, src_padded, pitch, cols, dst);
    AMT_LAUNCH_CHECK(ctx);
    return AMT_OK;
}

}  // extern "C"
