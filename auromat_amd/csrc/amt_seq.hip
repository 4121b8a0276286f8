// Packing of per-frame grids for the multi-GPU gather (include/auromat_hip.h, "sequences over several GPUs").
// Host code only: descriptors are assembled on the host, grids move device to device with hipMemcpyAsync.
#include "amt_common.h"

namespace {

bool frame_ok(const amt_seq_frame& f) {
    if (f.ny == 0) return true;            // a frame without any valid pixel travels as an empty descriptor
    return f.ny > 0 && f.nx > 0 && f.nc > 0 && f.mean != nullptr && f.count != nullptr;
}

void describe(const amt_seq_frame& f, double* d) {
    d[0] = f.ny, d[1] = f.ny ? f.nx : 0, d[2] = f.ny ? f.nc : 0;
    d[3] = f.lat0, d[4] = f.lon0, d[5] = f.dlat, d[6] = f.dlon;
    d[7] = f.index;
    d[8] = f.contains_pole ? 1.0 : 0.0, d[9] = f.contains_discontinuity ? 1.0 : 0.0;
    d[10] = f.altitude;
    d[11] = f.magnetic ? 1.0 : 0.0;
}

}  // namespace

extern "C" {

int amt_seq_payload_size(const amt_seq_frame* frames, int32_t n, int64_t* n_doubles) {
    if ((frames == nullptr && n > 0) || n < 0 || n_doubles == nullptr) return AMT_EINVAL;
    int64_t total = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (!frame_ok(frames[i])) return AMT_EINVAL;
        if (frames[i].ny == 0) continue;
        total += (int64_t)frames[i].ny * frames[i].nx * (frames[i].nc + 1);
    }
    *n_doubles = total;
    return AMT_OK;
}

int amt_seq_pack(amt_ctx* ctx, const amt_seq_frame* frames, int32_t n, int32_t max_frames, double* buffer,
                 int64_t capacity_doubles) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, (frames != nullptr || n == 0) && n >= 0 && max_frames >= n && buffer != nullptr, "bad argument");
    int64_t payload = 0;
    AMT_REQUIRE(ctx, amt_seq_payload_size(frames, n, &payload) == AMT_OK, "bad frame description");
    const int64_t head = (int64_t)max_frames * AMT_SEQ_DESC_LEN;
    AMT_REQUIRE(ctx, capacity_doubles >= head + payload, "buffer too small");
    if (amt_set_device(ctx)) return AMT_EHIP;
    std::vector<double> descs((size_t)head, 0.0);
    for (int32_t i = 0; i < n; ++i) describe(frames[i], descs.data() + (size_t)i * AMT_SEQ_DESC_LEN);
    if (head > 0) {
        // (pageable source: the copy has consumed `descs` when the call returns)
        AMT_HIP(ctx, hipMemcpyAsync(buffer, descs.data(), (size_t)head * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        AMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    double* at = buffer + head;
    for (int32_t i = 0; i < n; ++i) {
        const amt_seq_frame& f = frames[i];
        if (f.ny == 0) continue;
        const size_t n_mean = (size_t)f.ny * f.nx * f.nc, n_cnt = (size_t)f.ny * f.nx;
        AMT_HIP(ctx, hipMemcpyAsync(at, f.mean, n_mean * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        AMT_HIP(ctx, hipMemcpyAsync(at + n_mean, f.count, n_cnt * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        at += n_mean + n_cnt;
    }
    return AMT_OK;
}

int amt_seq_unpack(const double* host_buffer, int64_t n_doubles, int32_t n_frames, int32_t max_frames,
                   amt_seq_frame* out, int32_t capacity, int32_t* n_out) {
    if (host_buffer == nullptr || n_frames < 0 || max_frames < n_frames || (out == nullptr && capacity > 0) || capacity < 0 ||
        n_out == nullptr)
        return AMT_EINVAL;
    const int64_t head = (int64_t)max_frames * AMT_SEQ_DESC_LEN;
    if (n_doubles < head) return AMT_EINVAL;
    int64_t off = head;
    int32_t k = 0;
    for (int32_t i = 0; i < n_frames; ++i) {
        const double* d = host_buffer + (size_t)i * AMT_SEQ_DESC_LEN;
        const int64_t ny = (int64_t)d[0], nx = (int64_t)d[1], nc = (int64_t)d[2];
        if (ny == 0) continue;
        if (ny < 0 || nx <= 0 || nc <= 0) return AMT_EINVAL;
        const int64_t n_mean = ny * nx * nc, n_cnt = ny * nx;
        if (off + n_mean + n_cnt > n_doubles) return AMT_EINVAL;
        if (k < capacity) {
            amt_seq_frame& f = out[k];
            f.ny = (int32_t)ny, f.nx = (int32_t)nx, f.nc = (int32_t)nc;
            f.lat0 = d[3], f.lon0 = d[4], f.dlat = d[5], f.dlon = d[6];
            f.index = (int32_t)d[7];
            f.contains_pole = d[8] != 0, f.contains_discontinuity = d[9] != 0;
            f.altitude = d[10];
            f.magnetic = d[11] != 0;
            f.reserved = 0;
            f.mean = host_buffer + off;
            f.count = host_buffer + off + n_mean;
        }
        ++k;
        off += n_mean + n_cnt;
    }
    *n_out = k < capacity ? k : capacity;
    return k <= capacity ? AMT_OK : AMT_EINVAL;
}

}  // extern "C"
