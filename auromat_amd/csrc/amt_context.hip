// Context, stream, memory and event helpers of the C ABI (include/auromat_hip.h).
#include <sched.h>

#include <cstdlib>
#include <thread>

#include "amt_common.h"

extern "C" {

int amt_abi_version(void) { return AMT_ABI_VERSION; }

int amt_ctx_create(int device_id, void* stream, int own_stream, amt_ctx** out_ctx) {
    if (out_ctx == nullptr) return AMT_EINVAL;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) return AMT_EHIP;
    amt_ctx* ctx = new (std::nothrow) amt_ctx();
    if (ctx == nullptr) return AMT_ENOMEM;
    ctx->device = device_id;
    ctx->stream = nullptr;
    ctx->owns_stream = false;
    ctx->scratch = nullptr;
    ctx->timing = 0;
    ctx->copier = nullptr;
    ctx->last_second = ctx->last_bin = ctx->last_frames = -1;
    ctx->aux_pre = ctx->aux_tail = ctx->aux_fin = nullptr;
    ctx->tlaunch[0] = ctx->tlaunch[1] = 0;
    ctx->tframes[0] = ctx->tframes[1] = 0;
    ctx->tused[0] = ctx->tused[1] = 0;
    if (hipSetDevice(device_id) != hipSuccess) {
        delete ctx;
        return AMT_EHIP;
    }
    if (!own_stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);   // NULL = default stream
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return AMT_EHIP;
        }
        ctx->owns_stream = true;
    }
    if (hipMalloc(reinterpret_cast<void**>(&ctx->scratch), 256) != hipSuccess) {
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return AMT_ENOMEM;
    }
    *out_ctx = ctx;
    return AMT_OK;
}

int amt_ctx_destroy(amt_ctx* ctx) {
    AMT_CHECK_CTX(ctx);
    (void)hipSetDevice(ctx->device);
    amt_copier_destroy(ctx->copier);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (auto& w : ctx->workspaces)
        if (w.ptr) (void)hipFree(w.ptr);
    for (int k = 0; k < 2; ++k)
        for (hipEvent_t e : ctx->tev[k]) (void)hipEventDestroy(e);
    for (hipStream_t st : {ctx->aux_pre, ctx->aux_tail, ctx->aux_fin})
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    if (ctx->owns_stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return AMT_OK;
}

int amt_ctx_set_stream(amt_ctx* ctx, void* stream) {
    AMT_CHECK_CTX(ctx);
    if (ctx->owns_stream) {
        AMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        AMT_HIP(ctx, hipStreamDestroy(ctx->stream));
        ctx->owns_stream = false;
    }
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    return AMT_OK;
}

void* amt_ctx_get_stream(amt_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

int amt_ctx_synchronize(amt_ctx* ctx) {
    AMT_CHECK_CTX(ctx);
    AMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AMT_OK;
}

const char* amt_last_error(amt_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int amt_device_info(amt_ctx* ctx, char* name, size_t name_len, int* compute_units, int* clock_khz,
                    size_t* total_mem) {
    AMT_CHECK_CTX(ctx);
    hipDeviceProp_t prop;
    AMT_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) {
        std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        std::strncpy(name, s.c_str(), name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return AMT_OK;
}

int amt_malloc(amt_ctx* ctx, size_t bytes, void** out_dptr) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, out_dptr != nullptr, "out_dptr is NULL");
    if (amt_set_device(ctx)) return AMT_EHIP;
    hipError_t e = hipMalloc(out_dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        ctx->last_error = std::string("amt_malloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? AMT_ENOMEM : AMT_EHIP;
    }
    return AMT_OK;
}

int amt_free(amt_ctx* ctx, void* dptr) {
    AMT_CHECK_CTX(ctx);
    if (dptr == nullptr) return AMT_OK;
    if (amt_set_device(ctx)) return AMT_EHIP;
    AMT_HIP(ctx, hipFree(dptr));
    return AMT_OK;
}

int amt_host_threads(int wanted, int* cores, int* local_ranks) {
    int n_cores = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n_cores = CPU_COUNT(&set);
    if (n_cores <= 0) n_cores = (int)std::thread::hardware_concurrency();
    if (n_cores <= 0) n_cores = 1;
    int ranks = 1;
    for (const char* name : {"AMT_LOCAL_RANKS", "LOCAL_WORLD_SIZE"}) {
        const char* e = std::getenv(name);
        const int v = e ? std::atoi(e) : 0;
        if (v > 0) {
            ranks = v;
            break;
        }
    }
    if (cores) *cores = n_cores;
    if (local_ranks) *local_ranks = ranks;
    int share = n_cores / ranks;
    if (share < 1) share = 1;
    if (wanted < 1) wanted = 1;
    return wanted < share ? wanted : share;
}

int amt_malloc_host(amt_ctx* ctx, size_t bytes, void** out_hptr) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, out_hptr != nullptr, "out_hptr is NULL");
    if (amt_set_device(ctx)) return AMT_EHIP;
    hipError_t e = hipHostMalloc(out_hptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out_hptr = nullptr;
        ctx->last_error = std::string("amt_malloc_host: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? AMT_ENOMEM : AMT_EHIP;
    }
    return AMT_OK;
}

int amt_free_host(amt_ctx* ctx, void* hptr) {
    AMT_CHECK_CTX(ctx);
    if (hptr == nullptr) return AMT_OK;
    if (amt_set_device(ctx)) return AMT_EHIP;
    AMT_HIP(ctx, hipHostFree(hptr));
    return AMT_OK;
}

int amt_memcpy_h2d(amt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, dst && src, "NULL pointer");
    AMT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return AMT_OK;
}

int amt_memcpy_d2h(amt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, dst && src, "NULL pointer");
    AMT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AMT_OK;
}

int amt_memset(amt_ctx* ctx, void* dst, int value, size_t bytes) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, dst != nullptr, "NULL pointer");
    AMT_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return AMT_OK;
}

int amt_event_create(amt_ctx* ctx, void** out_event) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, out_event != nullptr, "out_event is NULL");
    hipEvent_t ev;
    AMT_HIP(ctx, hipEventCreate(&ev));
    *out_event = reinterpret_cast<void*>(ev);
    return AMT_OK;
}

int amt_event_destroy(amt_ctx* ctx, void* event) {
    AMT_CHECK_CTX(ctx);
    if (event) AMT_HIP(ctx, hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return AMT_OK;
}

int amt_event_record(amt_ctx* ctx, void* event) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, event != nullptr, "event is NULL");
    AMT_HIP(ctx, hipEventRecord(reinterpret_cast<hipEvent_t>(event), ctx->stream));
    return AMT_OK;
}

int amt_event_elapsed_ms(amt_ctx* ctx, void* start, void* stop, float* out_ms) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, start && stop && out_ms, "NULL argument");
    AMT_HIP(ctx, hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    AMT_HIP(ctx, hipEventElapsedTime(out_ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return AMT_OK;
}

int amt_timing_enable(amt_ctx* ctx, int enable) {
    AMT_CHECK_CTX(ctx);
    ctx->timing = enable > 0 ? enable : 0;
    ctx->tused[0] = ctx->tused[1] = 0;
    ctx->tlaunch[0] = ctx->tlaunch[1] = 0;
    ctx->tframes[0] = ctx->tframes[1] = 0;
    return AMT_OK;
}

int amt_timing_read(amt_ctx* ctx, int kernel, double* total_ms, int* launches) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, (kernel == AMT_KERNEL_GEOREF || kernel == AMT_KERNEL_BIN) && total_ms && launches, "bad argument");
    const size_t n = ctx->tused[kernel] / 2;
    double sum = 0;
    for (size_t i = 0; i < n; ++i) {
        float ms = 0;
        AMT_HIP(ctx, hipEventSynchronize(ctx->tev[kernel][2 * i + 1]));
        AMT_HIP(ctx, hipEventElapsedTime(&ms, ctx->tev[kernel][2 * i], ctx->tev[kernel][2 * i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = ctx->tframes[kernel];     // frames: a launch of the frame driver can cover more than one
    return AMT_OK;
}

int amt_georef_last_variant(amt_ctx* ctx, int32_t* second, int32_t* bin, int32_t* frames) {
    AMT_CHECK_CTX(ctx);
    AMT_REQUIRE(ctx, second && bin && frames, "NULL argument");
    *second = ctx->last_second, *bin = ctx->last_bin, *frames = ctx->last_frames;
    return AMT_OK;
}

}  // extern "C"
