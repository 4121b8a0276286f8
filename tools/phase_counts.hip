// Static VALU instruction counts of the phases of k_georef_rows' row step: every phase as a kernel of its own (inputs from
// memory, outputs to memory so that nothing folds away), compiled for gfx950; tools/phase_counts.py counts the ISA.
#include "/root/repo/auromat_amd/csrc/amt_georef.hip"
namespace {
using namespace amt;
#define LOADS(n) double v[n]; for (int i = 0; i < n; ++i) v[i] = in[threadIdx.x + 64 * i];
#define K(name) extern "C" __global__ void name(const double* __restrict__ in, double* __restrict__ out, shell_ray e, bowring_fast bw, fx::math_table mt)
}
K(ph_empty) { LOADS(2) out[threadIdx.x] = v[0]; out[threadIdx.x + 64] = v[1]; }
K(ph_ray_uu) { LOADS(7) vec3 u = affine_ray(v[0], v[1], v[2], v[3], v[4], v[5], v[6]); out[threadIdx.x] = fx::dot3(u.x, u.y, u.z, u.x, u.y, u.z); out[threadIdx.x+64] = u.x; out[threadIdx.x+128] = u.y; out[threadIdx.x+192] = u.z; }
K(ph_shell_t) { LOADS(4) vec3 u = {v[0], v[1], v[2]}; out[threadIdx.x] = shell_t(e, u, v[3]); }
K(ph_unit_dir) { LOADS(4) const double rs = fx::rsqrt_n(v[3]); out[threadIdx.x] = v[0] * rs; out[threadIdx.x+64] = v[1] * rs; out[threadIdx.x+128] = v[2] * rs; }
K(ph_point) { LOADS(4) vec3 u = {v[0], v[1], v[2]}; vec3 p = shell_point(e, u, v[3]); out[threadIdx.x] = p.x; out[threadIdx.x+64] = p.y; out[threadIdx.x+128] = p.z; }
K(ph_bowring) { LOADS(3) double n, d, ir; fx::bowring_nd(bw, v[0], v[1], v[2], n, d, ir); out[threadIdx.x] = n; out[threadIdx.x+64] = d; out[threadIdx.x+128] = ir; }
K(ph_small_angles) { LOADS(8) double a, b; bool ok; fx::small_angles(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], mt.small4, a, b, ok); out[threadIdx.x] = ok ? a : 0.0; out[threadIdx.x+64] = b; }
K(ph_full_angles) { LOADS(4) out[threadIdx.x] = fx::atan_pos_deg(v[0], v[1], mt.atan9); out[threadIdx.x+64] = fx::atan2_deg(v[2], v[3], mt.atan9); }
K(ph_elevation_low) { LOADS(1) out[threadIdx.x] = fx::asin_deg_low(v[0], mt.atan9); }
K(ph_elevation_dot) { LOADS(7) double c = -(fx::dot3(v[0], v[1], v[2], v[3], v[4], v[5]) * 0.25) * v[6]; c = fmin(1.0, fmax(-1.0, c)); out[threadIdx.x] = c; }
K(ph_bin_fast2) { LOADS(2) bool s1, s2; int bx = bin_fast(e.qa, e.qd, e.kx, 100, v[0], s1); int by = bin_fast(e.ky, e.kz, e.c0, 60, v[1], s2); out[threadIdx.x] = (double)(bx + 1000 * by + (s1 || s2 ? 1000000 : 0)); }
K(ph_centre_sums) { LOADS(12) double s[6]; for (int i = 0; i < 6; ++i) { const double t = v[i] + v[6 + i]; s[i] = t + from_next_lane(t); } for (int i = 0; i < 3; ++i) out[threadIdx.x + 64 * i] = s[i] * 0.25; for (int i = 3; i < 6; ++i) out[threadIdx.x + 64 * i] = s[i]; }
