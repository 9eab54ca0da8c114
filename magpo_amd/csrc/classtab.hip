// Input-class tables: the first layers of both networks see inputs from a small finite alphabet (CoordSum: an actor
// observation is (agent id, target), an encoder token (agent id, target, step count), a decoder token (previous action,
// step count)), so a layer applied to R rows is that layer applied to the C << R distinct rows followed by a row gather, and
// its parameter gradient is the layer's backward on the C per-class SUMS of the output gradient (the layer is linear in
// its parameters per row; rows of a class share the input, so sum_r f'(x_c) dy_r = f'(x_c) sum_r dy_r).
//   k_gather_rows      out[r] = table[cls[r]]                       (HBM write-bound; the table stays in L2 / MALL)
//   k_class_sum        partial[s][c] = sum of X rows of class c     (HBM read-bound, bit-stable: fixed ranges, fixed order)
// Rows of a class are visited through `order` (row ids sorted stably by class) and `offsets` (class boundaries): the
// summation order depends on the data only, never on scheduling, so reruns and graph replays agree bit for bit.
#include "common.hpp"

namespace magpo {

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ table, long ldt, const int* __restrict__ cls,
                                                     float* __restrict__ out, long ldo, long R, int W4) {
  const long n = R * W4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / W4;
    const int c4 = (int)(i - r * W4);
    const float4 v = *reinterpret_cast<const float4*>(table + (long)cls[r] * ldt + 4 * c4);
    *reinterpret_cast<float4*>(out + r * ldo + 4 * c4) = v;
  }
}

// grid (S, C): workgroup (s, c) sums slot s of class c's rows, i.e. sorted positions [lo + s*len/S, lo + (s+1)*len/S).
// 256 threads = G row groups x TPR threads per row (TPR = W/4 float4 columns); group g takes every G-th row of the slot,
// four rows in flight per thread; the G partial rows are folded in fixed order through LDS.
__global__ __launch_bounds__(256) void k_class_sum(const float* __restrict__ X, long ldx, const long* __restrict__ order,
                                                   const long* __restrict__ offsets, float* __restrict__ partial, int C, int W4,
                                                   int G) {
  extern __shared__ __align__(16) float4 sh4[];   // [G][W4]
  const int s = blockIdx.x, S = gridDim.x, c = blockIdx.y;
  const int tid = threadIdx.x;
  const int g = tid / W4, c4 = tid - g * W4;
  const long lo0 = offsets[c], len = offsets[c + 1] - lo0;
  const long lo = lo0 + len * s / S, hi = lo0 + len * (s + 1) / S;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (g < G) {
    long i = lo + g;
    const long st = G;
    for (; i + 3 * st < hi; i += 4 * st) {
      const long r0 = order[i], r1 = order[i + st], r2 = order[i + 2 * st], r3 = order[i + 3 * st];
      const float4 v0 = *reinterpret_cast<const float4*>(X + r0 * ldx + 4 * c4);
      const float4 v1 = *reinterpret_cast<const float4*>(X + r1 * ldx + 4 * c4);
      const float4 v2 = *reinterpret_cast<const float4*>(X + r2 * ldx + 4 * c4);
      const float4 v3 = *reinterpret_cast<const float4*>(X + r3 * ldx + 4 * c4);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; i < hi; i += st) {
      const float4 v0 = *reinterpret_cast<const float4*>(X + order[i] * ldx + 4 * c4);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    a0.x = (a0.x + a1.x) + (a2.x + a3.x); a0.y = (a0.y + a1.y) + (a2.y + a3.y);
    a0.z = (a0.z + a1.z) + (a2.z + a3.z); a0.w = (a0.w + a1.w) + (a2.w + a3.w);
    sh4[g * W4 + c4] = a0;
  }
  __syncthreads();
  if (g == 0) {
    float4 t = sh4[c4];
    for (int j = 1; j < G; ++j) {
      const float4 u = sh4[j * W4 + c4];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(partial + ((long)s * C + c) * (4L * W4) + 4 * c4) = t;
  }
}

}  // namespace magpo

using namespace magpo;

extern "C" int magpo_reduce_slabs(const float* slab, float* out, int G, long P, long stride, float scale, int accumulate, hipStream_t stream);

extern "C" int magpo_gather_rows(const float* table, long ldt, const int* cls, float* out, long ldo, long R, int W, hipStream_t st) {
  if (W < 4 || (W & 3) || (ldt & 3) || (ldo & 3)) { set_error("gather_rows: W and the strides must be multiples of 4"); return MAGPO_EINVAL; }
  if (R <= 0) return MAGPO_OK;
  const long n = R * (W / 4);
  const unsigned grid = resident_grid(k_gather_rows, 256, (n + 255) / 256);
  hipLaunchKernelGGL(k_gather_rows, dim3(grid), dim3(256), 0, st, table, ldt, cls, out, ldo, R, W / 4);
  return check_launch("magpo_gather_rows");
}

// Slots per class for C classes: enough workgroups to fill the chip, few enough that the second-stage reduce stays small.
extern "C" int magpo_class_sum_slots(int C) {
  int s = 8192 / (C < 1 ? 1 : C);
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

// X [R][W] (stride ldx), order [R] row ids sorted stably by class, offsets [C+1] class boundaries in `order`;
// partial [slots][C][W] (slots = magpo_class_sum_slots(C)); out [C][W] = per-class sums (fixed-order fold of the slots).
extern "C" int magpo_class_sum(const float* X, long ldx, const long* order, const long* offsets, int C, int W, float* partial,
                               float* out, hipStream_t st) {
  if (W < 4 || (W & 3) || W > 1024 || (ldx & 3) || C < 1) { set_error("class_sum: W must be a multiple of 4 in [4, 1024], ldx a multiple of 4"); return MAGPO_EINVAL; }
  const int W4 = W / 4, G = 256 / W4, S = magpo_class_sum_slots(C);
  hipLaunchKernelGGL(k_class_sum, dim3(S, C), dim3(256), (size_t)G * W4 * sizeof(float4), st, X, ldx, order, offsets, partial, C, W4, G);
  if (check_launch("magpo_class_sum") != MAGPO_OK) return MAGPO_ELAUNCH;
  return magpo_reduce_slabs(partial, out, S, (long)C * W, (long)C * W, 1.0f, 0, st);
}
