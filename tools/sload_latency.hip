// Latency of scalar loads (s_load + s_waitcnt lgkmcnt(0)) from the kernel-argument segment, from a __constant__ array and
// from plain device memory, with every SIMD busy (W waves each), and of an LDS broadcast read, on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/sload_latency.hip -o build/sload_latency && build/sload_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct big_args {
    unsigned long long w[448];      // 3.5 KB, like the row kernel's georef_batch
};
__constant__ unsigned long long kConst[448];

typedef const __attribute__((address_space(4))) unsigned long long* cptr;

template <int SRC>
__global__ __launch_bounds__(1024) void k_lat(big_args A, const unsigned long long* dev, int iters, unsigned long long* out) {
    __shared__ unsigned long long lds[448];
    if (threadIdx.x < 448) lds[threadIdx.x] = threadIdx.x & 7;
    __syncthreads();
    cptr base = SRC == 0 ? (cptr)__builtin_amdgcn_kernarg_segment_ptr() : SRC == 1 ? (cptr)kConst : (cptr)dev;
    unsigned long long idx = 0, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (SRC < 3) {
                // dependent chain: the loaded value (0..7) picks the next word
                asm volatile("" : "+s"(idx));
                const unsigned long long v = base[(idx * 37 + k * 5) % 448];
                idx = v & 7;
                acc += v;
            } else {
                const unsigned long long v = lds[(idx * 37 + k * 5) % 448];     // uniform address: broadcast
                idx = __builtin_amdgcn_readfirstlane((unsigned)v) & 7;
                acc += v;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0 + (acc == 12345 ? 1 : 0);
}

template <int SRC>
void run(const char* name, const big_args& A, const unsigned long long* dev, unsigned long long* out, std::vector<unsigned long long>& host) {
    for (int w : {1, 4}) {
        const int threads = 256 * w, blocks = 256, iters = 2000;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_lat<SRC>), dim3(blocks), dim3(threads), 0, 0, A, dev, iters, out);
        (void)hipDeviceSynchronize();
        const size_t nw = (size_t)blocks * threads / 64;
        (void)hipMemcpy(host.data(), out, nw * 8, hipMemcpyDeviceToHost);
        double c = 0;
        for (size_t i = 0; i < nw; ++i) c += (double)host[i];
        std::printf("%-28s W=%d  %.0f cycles per dependent load\n", name, w, c / nw / iters / 8);
    }
}

int main() {
    big_args A;
    for (int i = 0; i < 448; ++i) A.w[i] = i & 7;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(kConst), A.w, sizeof(A.w));
    unsigned long long *dev, *out;
    (void)hipMalloc(&dev, sizeof(A.w));
    (void)hipMemcpy(dev, A.w, sizeof(A.w), hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 1 << 20);
    std::vector<unsigned long long> host(1 << 17);
    run<0>("s_load kernarg segment", A, dev, out, host);
    run<1>("s_load __constant__ array", A, dev, out, host);
    run<2>("s_load device memory", A, dev, out, host);
    run<3>("ds_read broadcast + readlane", A, dev, out, host);
    return 0;
}
