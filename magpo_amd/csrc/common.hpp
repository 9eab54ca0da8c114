// Shared device helpers for the MAGPO gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MAGPO_OK 0
#define MAGPO_EINVAL (-1)
#define MAGPO_ELAUNCH (-2)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


namespace magpo {

void set_error(const char* msg);
int check_launch(const char* what);

// Grid of a grid-stride kernel: at most `want` blocks, and never more than are co-resident on the device (a partial
// second round of blocks would leave most CUs idle while the stragglers finish).
template <typename K>
inline unsigned resident_grid(K kernel, int threads, long want) {
  int per_cu = 0, dev = 0, ncu = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  const long cap = (long)per_cu * ncu;
  return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}

// Row pitch (floats) of a 64-column LDS tile: +4 keeps 16-B alignment and makes the
// row-per-lane ds_read_b128 pattern conflict-free (bank = 4*row + c mod 64).
constexpr int LDP = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum of part[q + stride * i], i < nb, over one wave (lane-strided partial sums, then a fixed butterfly): the
// deterministic final stage of the two-level double-precision reductions.
__device__ __forceinline__ double wave_sum_strided(const double* __restrict__ part, int nb, int stride, int q) {
  double s = 0.0;
  for (int i = threadIdx.x & 63; i < nb; i += 64) s += part[(long)stride * i + q];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// all-reduce sum over aligned groups of 16 consecutive lanes (= one DPP row): row mirrors + quad permutes on the VALU,
// no LDS round trip (a __shfl_xor is a ds_bpermute)
template <int CTRL> __device__ __forceinline__ float dpp_mov_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum16(float v) {
  v += dpp_mov_<0x140>(v);   // row_mirror
  v += dpp_mov_<0x141>(v);   // row_half_mirror
  v += dpp_mov_<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_<0xB1>(v);    // quad_perm [1,0,3,2]
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// hardware exp2 / rcp based variants for the GRU gates (abs. error ~1e-7, far inside the fp32 parity tolerance)
// v_rcp_f32 (1 ulp) directly: `__frcp_rn` / `1.0f / x` compile to the ~10-instruction IEEE division sequence
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * fast_rcp(1.0f + __expf(-2.0f * x)) - 1.0f; }
__device__ __forceinline__ float gelu_tanh(float x) {
  // jax.nn.gelu(approximate=True): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float c = 0.7978845608028654f;
  float u = c * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + fast_tanh(u));   // hardware exp / rcp (abs. error ~1e-7): the row kernels that use it are VALU-bound with libm's tanhf
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float c = 0.7978845608028654f;
  float x2 = x * x;
  float u = c * (x + 0.044715f * x * x2);
  float t = fast_tanh(u);
  float du = c * (1.0f + 3.0f * 0.044715f * x2);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float swish_grad(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// ---- Threefry-2x32-20 (Random123), the JAX default PRNG block function ----------------------
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                                      uint32_t& o0, uint32_t& o1) {
  const uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
#define TF_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)  x0 += ks1; x1 += ks2 + 1u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24) x0 += ks2; x1 += ks0 + 2u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)  x0 += ks0; x1 += ks1 + 3u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24) x0 += ks1; x1 += ks2 + 4u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)  x0 += ks2; x1 += ks0 + 5u;
#undef TF_R
  o0 = x0; o1 = x1;
}
// 32 random bits for flat element index idx (< 2^32) under key (k0,k1): x0 ^ x1.
__host__ __device__ __forceinline__ uint32_t random_bits32(uint32_t k0, uint32_t k1, uint32_t idx) {
  uint32_t a, b;
  threefry2x32(k0, k1, 0u, idx, a, b);
  return a ^ b;
}
// jax.random.gumbel (mode "low") from raw bits: -log(-log(max(tiny, u))), u = bitcast((b>>9)|0x3f800000)-1
__device__ __forceinline__ float gumbel_from_bits(uint32_t bits) {
  float f = __uint_as_float((bits >> 9) | 0x3f800000u) - 1.0f;
  const float tiny = 1.17549435e-38f;
  // floats * (1 - tiny) + tiny with (1 - tiny) == 1 in fp32, then max(tiny, .)
  float u = fmaxf(tiny, f + tiny);
  return -logf(-logf(u));
}

// jax.random.uniform(key, shape) element from raw bits: [0, 1) on the 2^-23 grid
__device__ __forceinline__ float uniform01_from_bits(uint32_t bits) { return __uint_as_float((bits >> 9) | 0x3f800000u) - 1.0f; }
// the gumbel of the sampling kernels: both logs in double, rounded once to fp32 (bit-identical to oracle/prng.py:bits_to_gumbel)
__device__ __forceinline__ float gumbel_exact_from_bits(uint32_t bits) {
  const float f = __uint_as_float((bits >> 9) | 0x3f800000u) - 1.0f;
  const float u = fmaxf(1.17549435e-38f, f + 1.17549435e-38f);
  return (float)(-log(-log((double)u)));
}
// jax.random.choice(key, n, shape=(), p=mask) (replace=True) over a bit mask of up to 256 cells (oracle/prng.py:choice):
// p_cuml = cumsum(mask); r = p_cuml[-1] * (1 - uniform(key, ())); searchsorted(p_cuml, r, side='left') = the first cell whose
// cumulative count reaches r.  The counts are integers, so that cell is the ceil(r)-th set bit; an all-zero mask gives cell 0.
__device__ __forceinline__ int choice_mask_cumsum(const unsigned long long (&m)[4], uint32_t k0, uint32_t k1) {
  const int total = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
  const float r = __fmul_rn((float)total, 1.0f - uniform01_from_bits(random_bits32(k0, k1, 0u)));
  int cum = 0;
  for (int w = 0; w < 4; ++w) {
    unsigned long long bits = m[w];
    while (bits) {
      const int b = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      if ((float)(++cum) >= r) return 64 * w + b;
    }
  }
  return 0;
}

// An fp32 operand as NP bf16 pieces for v_mfma_f32_32x32x16_bf16: 2 (x = hi + lo, 16 mantissa bits, products hh + hl + lh) or 3
// (x = hi + mid + lo, 24 mantissa bits -- what an fp32 operand holds -- products hh + hm + mh + hl + lh + mm on the same instruction:
// 6/16 of the fp32 MFMA time at fp32 accuracy; the dropped products ml, lm, ll are below 2^-24 of the result).
template <int NP> __device__ __forceinline__ void split_pieces(float x, __bf16 (&p)[NP]) {
  p[0] = (__bf16)x;
  float r = x - (float)p[0];
  p[1] = (__bf16)r;
  if (NP == 3) { r -= (float)p[1]; p[2] = (__bf16)r; }
}

}  // namespace magpo
