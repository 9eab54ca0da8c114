// optax.chain(clip_by_global_norm(max_norm), adam(lr, eps=1e-5)) + apply_updates on a flat fp32
// parameter buffer (rec_magpo.py:581-589, :412-420).  Two launches: a deterministic two-level
// sum of squares, then one fused elementwise pass (clip scale, moments, bias correction, update).
#include "common.hpp"

namespace magpo {

__global__ void k_sumsq_partial(const float* __restrict__ g, long n, float gscale, double* __restrict__ part) {
  __shared__ double sh[256];
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = (double)(g[i] * gscale);
    s += v * v;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void k_sumsq_final(const double* __restrict__ part, int nb, float* __restrict__ gnorm) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  const double s = wave_sum_strided(part, nb, 1, 0);
  if (threadIdx.x != 0) return;
  gnorm[0] = (float)sqrt(s);
}
// g <- g * gscale (mean over groups); clip; adam.  bc1 = 1 - b1^count, bc2 = 1 - b2^count (fp32, from host).
__global__ void k_clip_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu, float* __restrict__ nu,
                            long n, const float* __restrict__ gnorm, float gscale, float max_norm, float lr, float b1, float b2,
                            float eps, float bc1, float bc2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gn = gnorm[0];
  float gi = g[i] * gscale;
  if (!(gn < max_norm)) gi = (gi / gn) * max_norm;   // optax: select(g_norm < max_norm, t, (t / g_norm) * max_norm)
  const float m = b1 * mu[i] + (1.f - b1) * gi;
  const float v = b2 * nu[i] + (1.f - b2) * gi * gi;
  mu[i] = m;
  nu[i] = v;
  const float mh = m / bc1, vh = v / bc2;
  p[i] = p[i] + (-lr) * (mh / (sqrtf(vh) + eps));
}

}  // namespace magpo

using namespace magpo;

// workspace: >= 1024 doubles; gnorm: 1 float (device)
extern "C" int magpo_clip_adam(float* params, const float* grads, float* mu, float* nu, long n, float grad_scale,
                               float max_norm, float lr, float b1, float b2, float eps, float bc1, float bc2,
                               double* workspace, float* gnorm, hipStream_t st) {
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, st, grads, n, grad_scale, workspace);
  hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(64), 0, st, workspace, nb, gnorm);
  hipLaunchKernelGGL(k_clip_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, grads, mu, nu, n, gnorm, grad_scale,
                     max_norm, lr, b1, b2, eps, bc1, bc2);
  return check_launch("magpo_clip_adam");
}
