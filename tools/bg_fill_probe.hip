// Probe (round 6): a BACKGROUND fill — few, long-lived, small waves (one per CU, a dozen VGPRs: they fit beside the row kernel's four
// waves per SIMD) that write the sky rows' NaN at a throttled rate on a stream of their own while the row kernel (built with
// -DAMT_PROBE_SKIP_SKY: its sky items return at once) computes the Earth rows.  Linked into a variant build of the library:
//   bash tools/build_variant.sh nosky -DAMT_PROBE_SKIP_SKY tools/bg_fill_probe.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void k_background_fill(double* __restrict__ base, long long n_lines, int sleep) {
    // wave w of W writes lines w, w + W, ... eight at a time (a line: eight lanes x 16 bytes)
    const int lane = threadIdx.x;
    const long long W8 = (long long)gridDim.x * 8;
    const double2 two = {__builtin_nan(""), __builtin_nan("")};
    for (long long l = (long long)blockIdx.x * 8 + (lane >> 3); l < n_lines; l += W8) {
        *reinterpret_cast<double2*>(reinterpret_cast<char*>(base) + l * 128 + (lane & 7) * 16) = two;
        if (sleep > 0) {
            for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(2);          // 128 cycles each
        }
    }
}

extern "C" int amt_probe_background_fill(void* stream, double* base, long long n_lines, int waves, int sleep) {
    hipLaunchKernelGGL(k_background_fill, dim3((unsigned)waves), dim3(64), 0, static_cast<hipStream_t>(stream), base, n_lines, sleep);
    return (int)hipGetLastError();
}
