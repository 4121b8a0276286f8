// Issue cost (cycles per wave-instruction) of the vector instructions the frame kernel is made of, on gfx950:
// independent streams (4 accumulators) and dependent chains, W waves per SIMD, every SIMD of the chip busy.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Clock: s_memtime (shader clock) against s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum Op { FMA64, MUL64, ADD64, MOV32, MOV64, CND32, RCP64, RSQ64, SQRT64, DPP32, CVT_F64_I32, FMA32, RCP32, RSQ32, FLOOR64,
          CMP64, MAX64, LDEXP64, FMAC64_LIT, RCP64_DEP_FMA, CVT_F32_F64, CVT_F64_F32, PKFMA32,
          CMP_CND_VCC, CMP_CND_SGPR, CND_E64_SGPR, BFI32, ADDU32, AND32, LSHLADD, CVT_I32_F64, MULLO, READLANE, CND_LIT, CMP32_CND,
          DS_BPERM, DS_ADD, DS_MIN64, CMPX_MOV, MAD64, NOPS };
static const char* kNames[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_mov_b32", "v_mov_b64", "v_cndmask_b32", "v_rcp_f64",
                               "v_rsq_f64", "v_sqrt_f64", "v_mov_b32_dpp", "v_cvt_f64_i32", "v_fma_f32", "v_rcp_f32",
                               "v_rsq_f32", "v_floor_f64", "v_cmp_lt_f64", "v_max_f64", "v_ldexp_f64", "v_mov_b64+v_fmac_f64",
                               "v_rcp_f64+2fma(newton)", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_pk_fma_f32",
                               "v_cmp_f64 vcc+v_cndmask", "v_cmp_f64 s[]+v_cndmask_e64", "v_cndmask_e64 s[] (fixed mask)", "v_bfi_b32",
                               "v_add_u32", "v_and_b32", "v_lshl_add_u32", "v_cvt_i32_f64", "v_mul_lo_u32", "v_readlane_b32",
                               "v_cndmask vcc (vcc set by s_mov)", "v_cmp_u32 vcc+v_cndmask", "ds_bpermute_b32", "ds_add_u32",
                               "ds_min_f64", "saveexec+v_mov+restore", "v_mad_u64_u32"};

template <int OP, bool DEP>
__global__ __launch_bounds__(1024) void k_rate(int iters, unsigned long long* out, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const double b = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const float fb = 1.0000001f, fc = 1e-9f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    __shared__ double lds[1024];
    lds[threadIdx.x] = 0;
    const unsigned lds_addr = (unsigned)(size_t)(&lds[threadIdx.x]);
    asm volatile("s_mov_b32 s22, 0x55555555\n s_mov_b32 s23, 0x55555555" ::: "s22", "s23");
    if (OP == CND_LIT) asm volatile("s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define ONE(x, y, z, w)                                                                                     \
    if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));                  \
    if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));                              \
    if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));                              \
    if (OP == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(y));                                  \
    if (OP == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(x));                                  \
    if (OP == CND32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y) : "v"(i0) : );                 \
    if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));                                           \
    if (OP == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));                                           \
    if (OP == SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));                                         \
    if (OP == DPP32) asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(y)); \
    if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(y));                        \
    if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(z) : "v"(fb), "v"(fc));                \
    if (OP == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(z));                                           \
    if (OP == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(z));                                           \
    if (OP == FLOOR64) asm volatile("v_floor_f64 %0, %0" : "+v"(x));                                       \
    if (OP == CMP64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");                  \
    if (OP == MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));                              \
    if (OP == LDEXP64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(i0));                         \
    if (OP == FMAC64_LIT) asm volatile("v_mov_b64 %1, %2\n v_fmac_f64 %1, %0, %3" : "+v"(x), "=&v"(w) : "v"(b), "v"(c)); \
    if (OP == RCP64_DEP_FMA) asm volatile("v_rcp_f64 %1, %0\n v_fma_f64 %0, -%0, %1, 1.0\n v_fma_f64 %0, %1, %0, %1" : "+v"(x), "=&v"(w)); \
    if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(z) : "v"(x));                        \
    if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(z));                        \
    if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));                         \
    if (OP == CMP_CND_VCC) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(y) : "v"(x), "v"(b), "v"(i0) : "vcc"); \
    if (OP == CMP_CND_SGPR) asm volatile("v_cmp_lt_f64 s[20:21], %1, %2\n v_cndmask_b32 %0, %0, %3, s[20:21]" : "+v"(y) : "v"(x), "v"(b), "v"(i0) : "s20", "s21"); \
    if (OP == CND_E64_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(y) : "v"(i0));        \
    if (OP == BFI32) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(y) : "v"(i0), "v"(i1));                 \
    if (OP == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(i0));                              \
    if (OP == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(y) : "v"(i0));                               \
    if (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(y) : "v"(i0));                     \
    if (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(y) : "v"(x));                          \
    if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y) : "v"(i0));                            \
    if (OP == READLANE) asm volatile("v_readlane_b32 s24, %0, 3" : : "v"(y) : "s24");                        \
    if (OP == CND_LIT) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y) : "v"(i0));                    \
    if (OP == CMP32_CND) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(y) : "v"(i1), "v"(i0) : "vcc"); \
    if (OP == DS_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(y) : "v"(i0)); \
    if (OP == DS_ADD) asm volatile("ds_add_u32 %0, %1" : : "v"(lds_addr), "v"(y));                           \
    if (OP == DS_MIN64) asm volatile("ds_min_f64 %0, %1" : : "v"(lds_addr), "v"(x));                         \
    if (OP == CMPX_MOV) asm volatile("s_and_saveexec_b64 s[26:27], s[22:23]\n v_mov_b32 %0, %1\n s_or_b64 exec, exec, s[26:27]" : "+v"(y) : "v"(i0) : "s26", "s27"); \
    if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, s[28:29], %1, %1, %0" : "+v"(x) : "v"(i0) : "s28", "s29");
        double w0, w1, w2, w3;
        if (DEP) {
            REP16(ONE(a0, i0, f0, w0))
        } else {
            REP4(ONE(a0, i0, f0, w0) ONE(a1, i1, f1, w1) ONE(a2, i2, f2, w2) ONE(a3, i3, f3, w3))
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
    if (a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + i0 + i1 + i2 + i3 == 12345.678) out[0] = 0;
}

template <int OP, bool DEP>
void run(int waves_per_simd, unsigned long long* dev, std::vector<unsigned long long>& host) {
    const int cus = 256, iters = 4000;
    // one workgroup of 256 * min(W, 4) threads per CU (x2 workgroups for W = 8): waves spread evenly over the 4 SIMDs
    const int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd);
    const int blocks = cus * (waves_per_simd > 4 ? waves_per_simd / 4 : 1);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rate<OP, DEP>), dim3(blocks), dim3(threads), 0, 0, iters, dev, 1.5);
    hipDeviceSynchronize();
    const size_t nw = (size_t)blocks * threads / 64;
    hipMemcpy(host.data(), dev, nw * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (size_t i = 0; i < nw; ++i) {
        cyc += (double)host[2 * i];
        real += (double)host[2 * i + 1];
    }
    const int per_iter = 16 * ((OP == FMAC64_LIT || OP == CMP_CND_VCC || OP == CMP_CND_SGPR || OP == CMP32_CND) ? 2 : (OP == RCP64_DEP_FMA) ? 3 : 1);
    // cycles one SIMD spends per wave-instruction = wave's cycles / instructions / waves sharing the SIMD
    std::printf("%-24s %-4s W=%d  %6.2f cyc/instr/SIMD  (wave sees %7.2f)  clock %.0f MHz\n", kNames[OP], DEP ? "dep" : "ind",
                waves_per_simd, cyc / nw / iters / per_iter / waves_per_simd, cyc / nw / iters / per_iter, cyc / real * 100.0);
}

template <int OP>
void both(unsigned long long* dev, std::vector<unsigned long long>& host) {
    for (int w : {1, 2, 4}) run<OP, false>(w, dev, host);
    for (int w : {1, 4}) run<OP, true>(w, dev, host);
}

int main() {
    unsigned long long* dev;
    hipMalloc(&dev, 1 << 22);
    std::vector<unsigned long long> host(1 << 19);
    both<FMA64>(dev, host);
    both<MUL64>(dev, host);
    both<ADD64>(dev, host);
    both<MOV32>(dev, host);
    both<MOV64>(dev, host);
    both<CND32>(dev, host);
    both<RCP64>(dev, host);
    both<RSQ64>(dev, host);
    both<SQRT64>(dev, host);
    both<DPP32>(dev, host);
    both<CVT_F64_I32>(dev, host);
    both<FMA32>(dev, host);
    both<RCP32>(dev, host);
    both<RSQ32>(dev, host);
    both<FLOOR64>(dev, host);
    both<CMP64>(dev, host);
    both<MAX64>(dev, host);
    both<LDEXP64>(dev, host);
    both<FMAC64_LIT>(dev, host);
    both<RCP64_DEP_FMA>(dev, host);
    both<CVT_F32_F64>(dev, host);
    both<CVT_F64_F32>(dev, host);
    both<PKFMA32>(dev, host);
    both<CMP_CND_VCC>(dev, host);
    both<CMP_CND_SGPR>(dev, host);
    both<CND_E64_SGPR>(dev, host);
    both<CND_LIT>(dev, host);
    both<CMP32_CND>(dev, host);
    both<BFI32>(dev, host);
    both<ADDU32>(dev, host);
    both<AND32>(dev, host);
    both<LSHLADD>(dev, host);
    both<CVT_I32_F64>(dev, host);
    both<MULLO>(dev, host);
    both<MAD64>(dev, host);
    both<READLANE>(dev, host);
    both<DS_BPERM>(dev, host);
    both<DS_ADD>(dev, host);
    both<DS_MIN64>(dev, host);
    both<CMPX_MOV>(dev, host);
    hipFree(dev);
    return 0;
}
