// Host-side helper of the file writers (no GPU code): the HDF5 filter pipeline of the reference's netCDF-4 files — byte
// shuffle, then deflate — over the chunks of one variable, on a few threads.  netCDF4-python's createVariable(zlib=True)
// gives every chunk (one row of the array: reference auromat/export/netcdf.py:128-326, chunksizes=(1, w)) to HDF5's shuffle
// and deflate filters; a resampled grid has ~1 400 such chunks of ~2 KB per file, an unresampled frame 34 000 of 34 KB, and
// the per-chunk work in Python (a NumPy transpose, a zlib call, a bytes object) held the interpreter lock for most of a
// file's 18 ms.  Same zlib, same level, same bytes as zlib.compress() of the shuffled chunk.
//
// Built by auromat_amd/export/_io.py with g++ -lz into auromat_amd/lib/libauromat_io.so; the writers fall back to their
// Python path when it is absent (it is a speed-up of file writing, not part of the GPU product path).
#include <zlib.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

// bytes deflate may need for a chunk of `chunk_bytes`
int64_t amt_io_deflate_bound(int64_t chunk_bytes) { return (int64_t)compressBound((uLong)chunk_bytes); }

// n_chunks chunks of chunk_bytes each, back to back in src (elements of `itemsize` bytes) -> chunk i as a zlib stream at
// out + i * out_stride (out_stride >= amt_io_deflate_bound), its length in out_sizes[i].  shuffle != 0: HDF5's shuffle
// filter first (byte k of every element gathered into the k-th block of the chunk).  Returns 0, or the zlib error of the
// first chunk that failed.
int amt_io_deflate_chunks(const uint8_t* src, int64_t n_chunks, int64_t chunk_bytes, int32_t itemsize, int32_t level,
                          int32_t shuffle, uint8_t* out, int64_t out_stride, int64_t* out_sizes, int32_t n_threads) {
    if (n_chunks <= 0) return 0;
    if (itemsize < 1 || chunk_bytes % itemsize != 0 || chunk_bytes >= (1ll << 31) || out_stride >= (1ll << 31)) return Z_DATA_ERROR;
    std::atomic<int64_t> next{0};
    std::atomic<int> status{0};
    auto work = [&]() {
        std::vector<uint8_t> buf(shuffle && itemsize > 1 ? (size_t)chunk_bytes : 0);
        const int64_t n_elem = chunk_bytes / itemsize;
        // ONE deflate state per thread, reset per chunk: compress2() allocates and frees its 256 KB of state for every chunk,
        // and allocations of that size go through mmap / munmap, which serialises the threads in the kernel
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        int rc = deflateInit(&z, level);
        if (rc != Z_OK) {
            int expected = 0;
            status.compare_exchange_strong(expected, rc);
            return;
        }
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n_chunks || status.load() != 0) break;
            const uint8_t* in = src + i * chunk_bytes;
            if (!buf.empty()) {
                for (int32_t k = 0; k < itemsize; ++k) {
                    uint8_t* dst = buf.data() + (int64_t)k * n_elem;
                    const uint8_t* s = in + k;
                    for (int64_t e = 0; e < n_elem; ++e) dst[e] = s[e * itemsize];
                }
                in = buf.data();
            }
            deflateReset(&z);
            z.next_in = const_cast<Bytef*>(in);
            z.avail_in = (uInt)chunk_bytes;
            z.next_out = out + i * out_stride;
            z.avail_out = (uInt)out_stride;
            rc = deflate(&z, Z_FINISH);
            if (rc != Z_STREAM_END) {
                int expected = 0;
                status.compare_exchange_strong(expected, rc == Z_OK ? Z_BUF_ERROR : rc);
                break;
            }
            out_sizes[i] = (int64_t)z.total_out;
        }
        deflateEnd(&z);
    };
    int64_t nt = n_threads < 1 ? 1 : n_threads;
    if (nt > n_chunks) nt = n_chunks;
    // (small jobs: a thread costs more than it deflates)
    if (n_chunks * chunk_bytes < (1 << 16)) nt = 1;
    std::vector<std::thread> pool;
    for (int64_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return status.load();
}

// ONE gzip member out of blocks deflated side by side (what pigz does): the input is cut into pieces of `block_bytes`, every
// piece becomes a raw deflate stream of its own that ends on a byte boundary with an empty stored block (Z_SYNC_FLUSH) — the
// last one with the final block (Z_FINISH) —, and the pieces are laid end to end behind a 10-byte gzip header and in front
// of CRC-32 and length of the whole input.  Any inflate reads it as one stream; it is a few tenths of a percent larger than
// a single deflate run (no history across the cuts).  A compressed CDF variable is one such gzip member per record (the
// format leaves nothing else to split): 96 MB per array of a full frame.
// out: capacity out_cap >= amt_io_gzip_bound(n, block_bytes); *out_len = bytes written.  Returns 0 or a zlib error.
int64_t amt_io_gzip_bound(int64_t n, int64_t block_bytes) {
    const int64_t blocks = n > 0 ? (n + block_bytes - 1) / block_bytes : 1;
    return 18 + blocks * ((int64_t)deflateBound(nullptr, (uLong)block_bytes) + 16);
}

int amt_io_gzip_parallel(const uint8_t* src, int64_t n, int32_t level, int64_t block_bytes, uint8_t* out, int64_t out_cap,
                         int64_t* out_len, int32_t n_threads) {
    if (n < 0 || block_bytes < 1024 || block_bytes >= (1ll << 30)) return Z_DATA_ERROR;
    const int64_t blocks = n > 0 ? (n + block_bytes - 1) / block_bytes : 1;
    const int64_t slot = (int64_t)deflateBound(nullptr, (uLong)block_bytes) + 16;
    if (out_cap < 18 + blocks * slot) return Z_BUF_ERROR;
    std::vector<int64_t> sizes((size_t)blocks, 0);
    // every block into its own slot behind the header, then the slots are moved together
    std::atomic<int64_t> next{0};
    std::atomic<int> status{0};
    auto work = [&]() {
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        int rc = deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        if (rc != Z_OK) {
            int expected = 0;
            status.compare_exchange_strong(expected, rc);
            return;
        }
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= blocks || status.load() != 0) break;
            const int64_t a = i * block_bytes, len = (i + 1 == blocks) ? n - a : block_bytes;
            deflateReset(&z);
            z.next_in = const_cast<Bytef*>(src + a);
            z.avail_in = (uInt)len;
            z.next_out = out + 10 + i * slot;
            z.avail_out = (uInt)slot;
            rc = deflate(&z, i + 1 == blocks ? Z_FINISH : Z_SYNC_FLUSH);
            if ((i + 1 == blocks) ? rc != Z_STREAM_END : (rc != Z_OK || z.avail_in != 0 || z.avail_out == 0)) {
                int expected = 0;
                status.compare_exchange_strong(expected, rc < 0 ? rc : Z_BUF_ERROR);
                break;
            }
            sizes[(size_t)i] = (int64_t)z.total_out;
        }
        deflateEnd(&z);
    };
    int64_t nt = n_threads < 1 ? 1 : n_threads;
    if (nt > blocks) nt = blocks;
    std::vector<std::thread> pool;
    for (int64_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (status.load() != 0) return status.load();
    static const uint8_t header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};      // deflate, no flags, no time, Unix
    std::memcpy(out, header, 10);
    int64_t at = 10;
    for (int64_t i = 0; i < blocks; ++i) {
        if (at != 10 + i * slot) std::memmove(out + at, out + 10 + i * slot, (size_t)sizes[(size_t)i]);
        at += sizes[(size_t)i];
    }
    // CRC-32 of the whole input: per block on the threads would need crc32_combine; one pass at ~1 GB/s is 0.1 s per 96 MB —
    // done in pieces on the same threads and combined
    std::vector<uLong> crcs((size_t)blocks, 0);
    std::atomic<int64_t> next_crc{0};
    auto crc_work = [&]() {
        for (;;) {
            const int64_t i = next_crc.fetch_add(1);
            if (i >= blocks) return;
            const int64_t a = i * block_bytes, len = (i + 1 == blocks) ? n - a : block_bytes;
            crcs[(size_t)i] = crc32(crc32(0L, Z_NULL, 0), src + a, (uInt)len);
        }
    };
    pool.clear();
    for (int64_t t = 1; t < nt; ++t) pool.emplace_back(crc_work);
    crc_work();
    for (auto& th : pool) th.join();
    uLong crc = crc32(0L, Z_NULL, 0);
    for (int64_t i = 0; i < blocks; ++i) {
        const int64_t a = i * block_bytes, len = (i + 1 == blocks) ? n - a : block_bytes;
        crc = crc32_combine(crc, crcs[(size_t)i], (z_off_t)len);
    }
    const uint32_t c = (uint32_t)crc, isize = (uint32_t)((uint64_t)n & 0xffffffffu);
    for (int k = 0; k < 4; ++k) out[at + k] = (uint8_t)(c >> (8 * k));
    for (int k = 0; k < 4; ++k) out[at + 4 + k] = (uint8_t)(isize >> (8 * k));
    *out_len = at + 8;
    return 0;
}

}  // extern "C"
