// Probe: raw accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and of 1- and 2-step Newton refinements.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probe_f64_approx.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, int n, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double e = fma(-v, r0, 1.0);
    double r1 = fma(r0, e, r0);
    e = fma(-v, r1, 1.0);
    double r2 = fma(r1, e, r1);
    double y0 = __builtin_amdgcn_rsq(v);
    double g = v * y0, h = 0.5 * y0;
    double rr = fma(-h, g, 0.5);
    double g1 = fma(g, rr, g), h1 = fma(h, rr, h);
    rr = fma(-h1, g1, 0.5);
    double g2 = fma(g1, rr, g1), h2 = fma(h1, rr, h1);
    out[8 * i + 0] = r0; out[8 * i + 1] = r1; out[8 * i + 2] = r2;
    out[8 * i + 3] = y0; out[8 * i + 4] = 2 * h1; out[8 * i + 5] = 2 * h2;
    out[8 * i + 6] = g1; out[8 * i + 7] = g2;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), out(8 * (size_t)n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::exp((u - 0.5) * 60.0);   // 1e-13 .. 1e13
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 8 * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, n, dout);
    hipMemcpy(out.data(), dout, 8 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double m[8] = {0};
    for (int i = 0; i < n; ++i) {
        long double v = x[i];
        long double ref[8] = {1 / v, 1 / v, 1 / v, 1 / sqrtl(v), 1 / sqrtl(v), 1 / sqrtl(v), sqrtl(v), sqrtl(v)};
        for (int j = 0; j < 8; ++j) {
            double rel = (double)fabsl((out[8 * (size_t)i + j] - ref[j]) / ref[j]);
            if (rel > m[j]) m[j] = rel;
        }
    }
    printf("max rel err: rcp raw %.3e, 1NR %.3e, 2NR %.3e | rsq raw %.3e, 1NR %.3e, 2NR %.3e | sqrt 1NR %.3e 2NR %.3e\n",
           m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
    return 0;
}
