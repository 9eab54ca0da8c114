// Wide observations (obs_dim > 32, e.g. Robot Warehouse's 71 + A features): the observation-side first layers run on the
// MFMA dense kernels over rows padded to 128 columns instead of the small-input row kernels (rowops.hip: F <= 32).
//   k_obsnorm_fwd   on[r][f] = obs[r][f] * rsqrt(mean_{f < F} obs[r]^2 + eps) * s_obs[f]   (nn.RMSNorm over the F features,
//                   sable_network.py:93-95), columns F..127 written as zeros
//   k_obsnorm_bwd   slab_s[g][f] = sum_r don[r][f] * obs[r][f] * rstd_r   (gradient of s_obs; the observation needs none)
//   k_add_pe        out[r] = x[r] + pe[clamp(pos[r])]   (E-wide rows, E = 64 or 128)
#include "common.hpp"

namespace magpo {

constexpr int WP = 128;   // padded observation width

// one row per 32 lanes (float4 per lane), 8 rows per 256-thread block and grid-stride iteration
__global__ __launch_bounds__(256) void k_obsnorm_fwd(const float* __restrict__ obs, long ldo, int F, const float* __restrict__ s_obs,
                                                     float* __restrict__ on, long R) {
  const int sub = threadIdx.x >> 5, l = threadIdx.x & 31, c4 = 4 * l;
  for (long row = (long)blockIdx.x * 8 + sub; row < R; row += (long)gridDim.x * 8) {
    float4 v = *reinterpret_cast<const float4*>(obs + row * ldo + c4);
    if (c4 + 0 >= F) v.x = 0.f;
    if (c4 + 1 >= F) v.y = 0.f;
    if (c4 + 2 >= F) v.z = 0.f;
    if (c4 + 3 >= F) v.w = 0.f;
    float ms = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ms += __shfl_xor(ms, o, 64);
    const float rstd = rsqrtf(ms / (float)F + 1e-6f);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 + 0 < F) s.x = s_obs[c4];
    if (c4 + 1 < F) s.y = s_obs[c4 + 1];
    if (c4 + 2 < F) s.z = s_obs[c4 + 2];
    if (c4 + 3 < F) s.w = s_obs[c4 + 3];
    *reinterpret_cast<float4*>(on + row * WP + c4) = make_float4(v.x * rstd * s.x, v.y * rstd * s.y, v.z * rstd * s.z, v.w * rstd * s.w);
  }
}

__global__ __launch_bounds__(256) void k_obsnorm_bwd(const float* __restrict__ obs, long ldo, int F, const float* __restrict__ don,
                                                     float* __restrict__ slab_s /*[grid][128]*/, long R) {
  __shared__ float4 sh[8][32];
  const int sub = threadIdx.x >> 5, l = threadIdx.x & 31, c4 = 4 * l;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long row = (long)blockIdx.x * 8 + sub; row < R; row += (long)gridDim.x * 8) {
    float4 v = *reinterpret_cast<const float4*>(obs + row * ldo + c4);
    if (c4 + 0 >= F) v.x = 0.f;
    if (c4 + 1 >= F) v.y = 0.f;
    if (c4 + 2 >= F) v.z = 0.f;
    if (c4 + 3 >= F) v.w = 0.f;
    float ms = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ms += __shfl_xor(ms, o, 64);
    const float rstd = rsqrtf(ms / (float)F + 1e-6f);
    const float4 d = *reinterpret_cast<const float4*>(don + row * WP + c4);
    acc.x += d.x * v.x * rstd; acc.y += d.y * v.y * rstd; acc.z += d.z * v.z * rstd; acc.w += d.w * v.w * rstd;
  }
  sh[sub][l] = acc;
  __syncthreads();
  if (sub == 0) {
    float4 t = sh[0][l];
    for (int j = 1; j < 8; ++j) { const float4 u = sh[j][l]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(slab_s + (long)blockIdx.x * WP + c4) = t;
  }
}

__global__ __launch_bounds__(256) void k_add_pe(const float* __restrict__ x, long ldx, const float* __restrict__ pe, const int* __restrict__ pos,
                                                long pos_stride, int npos, float* __restrict__ out, long ldout, long R, int E) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int w4 = E >> 2;
  const long row = i / w4;
  if (row >= R) return;
  const int c4 = 4 * (int)(i - row * w4);
  int p = pos[row * pos_stride];
  p = p < 0 ? 0 : (p >= npos ? npos - 1 : p);
  const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c4), b = *reinterpret_cast<const float4*>(pe + (long)p * E + c4);
  *reinterpret_cast<float4*>(out + row * ldout + c4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

}  // namespace magpo

using namespace magpo;

static unsigned wide_grid(long R) {
  long b = (R + 7) / 8;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int magpo_obsnorm_grid(long R) { return (int)wide_grid(R); }

extern "C" int magpo_obsnorm_fwd(const float* obs, long ldo, int F, const float* s_obs, float* on, long R, hipStream_t st) {
  if (F < 1 || F > WP || ldo < WP || (ldo & 3)) { set_error("obsnorm: 1 <= F <= 128 and rows padded to >= 128 floats (stride a multiple of 4)"); return MAGPO_EINVAL; }
  if (R <= 0) return MAGPO_OK;
  hipLaunchKernelGGL(k_obsnorm_fwd, dim3(wide_grid(R)), dim3(256), 0, st, obs, ldo, F, s_obs, on, R);
  return check_launch("magpo_obsnorm_fwd");
}

// slab_s: [magpo_obsnorm_grid(R)][128]
extern "C" int magpo_obsnorm_bwd(const float* obs, long ldo, int F, const float* don, float* slab_s, long R, hipStream_t st) {
  if (F < 1 || F > WP || ldo < WP || (ldo & 3)) { set_error("obsnorm: 1 <= F <= 128 and rows padded to >= 128 floats (stride a multiple of 4)"); return MAGPO_EINVAL; }
  if (R <= 0) return MAGPO_OK;
  hipLaunchKernelGGL(k_obsnorm_bwd, dim3(wide_grid(R)), dim3(256), 0, st, obs, ldo, F, don, slab_s, R);
  return check_launch("magpo_obsnorm_bwd");
}

extern "C" int magpo_add_pe(const float* x, long ldx, const float* pe, const int* pos, long pos_stride, int npos, float* out, long ldout,
                            long R, int E, hipStream_t st) {
  if (E != 64 && E != 128) { set_error("magpo_add_pe: the row width must be 64 or 128"); return MAGPO_EINVAL; }
  if (R <= 0) return MAGPO_OK;
  hipLaunchKernelGGL(k_add_pe, dim3((unsigned)((R * (E / 4) + 255) / 256)), dim3(256), 0, st, x, ldx, pe, pos, pos_stride, npos, out, ldout, R, E);
  return check_launch("magpo_add_pe");
}
