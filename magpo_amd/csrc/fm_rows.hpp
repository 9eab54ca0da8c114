// Feature-major register rows and transposed dense layers on v_mfma_f32_16x16x4_f32 (shared by the fused acting kernel and the
// fused training segments).  Lane (row = l & 15, kq = l >> 4) holds the 16 features n = 16 g + 4 kq + r (g, r in 0..3) of one
// 64-wide row; a dense layer is computed transposed (weights = A operand, activation registers = B operand) and its
// accumulator comes out in the same layout, so GEMM -> norm -> activation -> GEMM chains stay in registers.
#pragma once
#include "common.hpp"

namespace magpo {

constexpr int AE = 64;           // embed dim
constexpr float EPSN = 1e-6f;

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming accesses for the retention states (each byte is touched once per pass): keep them from displacing the
// weights and the scratch rows in L2
__device__ __forceinline__ float4 ld4nt(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4nt(float* p, float4 v) {
  const f32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}
// order this wave's LDS / global writes before its later reads by other lanes (single-wave workgroup)
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// LDS-only ordering inside one wave (DS ops of a wave execute in order; this only pins the compiler's schedule)
__device__ __forceinline__ void lsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// swish gate on the hardware exp / rcp units (abs. error ~1e-7, inside the fp32 parity tolerance; cf. gru.hip)
__device__ __forceinline__ float fswish(float x) { return x * fast_sigmoid(x); }

// ---- cross-lane sums on the VALU (no LDS round trip): v_permlane{16,32}_swap for lane ^ 16 / lane ^ 32, DPP inside a 16-lane row
__device__ __forceinline__ float xsum16(float v) {   // v[l] + v[l ^ 16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float xsum32(float v) {   // v[l] + v[l ^ 32]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float xget32(float v, int lane) {   // v[l ^ 32]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(lane < 32 ? r[1] : r[0]);
}
__device__ __forceinline__ float xget16(float v, int lane) {   // v[l ^ 16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float((lane & 16) ? r[0] : r[1]);
}
template <int CTRL> __device__ __forceinline__ float dpp_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// all-reduce sum over aligned groups of GL lanes (GL = 1, 2, 4, 8, 16) of a 16-lane row
__device__ __forceinline__ float gsum(float v, int gl) {
  if (gl >= 16) v += dpp_<0x140>(v);   // row_mirror
  if (gl >= 8) v += dpp_<0x141>(v);    // row_half_mirror
  if (gl >= 4) v += dpp_<0x4E>(v);     // quad_perm [2,3,0,1]
  if (gl >= 2) v += dpp_<0xB1>(v);     // quad_perm [1,0,3,2]
  return v;
}

// ---- feature-major rows: reg j <-> feature 16 (j >> 2) + 4 kq + (j & 3) -------------------------------------------
struct Row { float v[16]; };
__device__ __forceinline__ Row row_load(const float* p /* row base */, int kq) {
  Row r;
#pragma unroll
  for (int g = 0; g < 4; ++g) { const float4 t = ld4g(p + 16 * g + 4 * kq); r.v[4 * g] = t.x; r.v[4 * g + 1] = t.y; r.v[4 * g + 2] = t.z; r.v[4 * g + 3] = t.w; }
  return r;
}
__device__ __forceinline__ void row_store(float* p, int kq, const Row& r) {
#pragma unroll
  for (int g = 0; g < 4; ++g) st4g(p + 16 * g + 4 * kq, make_float4(r.v[4 * g], r.v[4 * g + 1], r.v[4 * g + 2], r.v[4 * g + 3]));
}
__device__ __forceinline__ float row_sum(const Row& r) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += r.v[j];
  return xsum32(xsum16(s));
}
__device__ __forceinline__ Row row_add(const Row& a, const Row& b) {
  Row r;
#pragma unroll
  for (int j = 0; j < 16; ++j) r.v[j] = a.v[j] + b.v[j];
  return r;
}
// Timing experiments only (results are WRONG with them; scripts/debug/act_ab2.sh): -DMAGPO_X_NOROWMATH makes RMSNorm / GELU pass-through,
// -DMAGPO_X_NOW replaces the weight-fragment loads of the dense layers by constants, -DMAGPO_X_NOMFMA drops their MFMAs.
__device__ __forceinline__ Row row_rms(const Row& x, const float* scale, int kq) {
#ifdef MAGPO_X_NOROWMATH
  return x;
#endif
  Row q;
#pragma unroll
  for (int j = 0; j < 16; ++j) q.v[j] = x.v[j] * x.v[j];
  const float rstd = rsqrtf(row_sum(q) * (1.0f / 64.0f) + EPSN);
  const Row s = row_load(scale, kq);
  Row r;
#pragma unroll
  for (int j = 0; j < 16; ++j) r.v[j] = x.v[j] * rstd * s.v[j];
  return r;
}
// GELU (tanh form) on the hardware exp / rcp units: the fused kernels run one or two waves per SIMD and are VALU-issue bound,
// libm's tanhf is ~40 instructions per element (abs. error of the fast form ~1e-7, far inside the fp32 parity tolerance)
__device__ __forceinline__ Row row_gelu(const Row& x) {
#ifdef MAGPO_X_NOROWMATH
  return x;
#endif
  Row r;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float v = x.v[j];
    r.v[j] = 0.5f * v * (1.0f + fast_tanh(0.7978845608028654f * (v + 0.044715f * v * v * v)));
  }
  return r;
}

// ---- dense layer, transposed on 16x16x4 fp32 MFMA: out(g, acc) receives features 16 g + 4 kq + (0..3) of every env ------
#ifdef MAGPO_X_NOW
#define WLD(p) make_float4(1e-3f, -2e-3f, 3e-3f, -1e-3f)
#else
#define WLD(p) ld4g(p)
#endif
#ifdef MAGPO_X_NOMFMA
#define WMFMA(a, b, c) f32x4{c[0] + (a) * (b), c[1], c[2], c[3]}
#else
#define WMFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#endif
// Weight operand layouts: the transposed copy Wt [n][64] of magpo_transpose_pad (FRAG = false: lane (m, kq) reads 16 bytes of row 16 g + m,
// i.e. one instruction touches 16 rows x 64 B), or the FRAGMENT-MAJOR copy of the acting kernel (FRAG = true, SableGuider.build_act_weights):
//   Wf[g][gk][lane = m + 16 kq][4] = Wt[16 g + m][16 gk + 4 kq .. + 3]
// so that one instruction reads 1 KB contiguous (8 full cache lines instead of 16 half lines) from base + 16 * lane.  A wave that streams
// its weights from L2 for every token (the acting kernel: 1.15 MB per wave and launch) is bound by those requests, not by the MFMAs.
template <bool FRAG>
__device__ __forceinline__ float4 wfrag(const float* __restrict__ Wt, int g, int gk, int m, int kq) {
  if (FRAG) return WLD(Wt + g * 1024 + gk * 256 + 4 * (m + 16 * kq));
  return WLD(Wt + (long)(16 * g + m) * AE + 16 * gk + 4 * kq);
}
#ifndef MAGPO_WGEMM_PD
#define MAGPO_WGEMM_PD 4
#endif
// The first fragments of a layer can be requested AHEAD of the code that produces the layer's input (wload<NG>() before the row math of the
// previous layer, wgemm_pre<NG>() after it): a wave that runs alone on its SIMD otherwise sits out one L2 round trip at the top of every
// dense layer -- ~45 of them per acting step.
struct WPre { float4 w[MAGPO_WGEMM_PD][4]; };
template <int NG, bool FRAG>
__device__ __forceinline__ WPre wload(const float* __restrict__ Wt, int m, int kq) {
  constexpr int PD = NG < MAGPO_WGEMM_PD ? NG : MAGPO_WGEMM_PD;
  WPre r;
#pragma unroll
  for (int p = 0; p < PD; ++p)
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) r.w[p][gk] = wfrag<FRAG>(Wt, p, gk, m, kq);
  return r;
}
template <int NG, bool FRAG = false, class OUT = void>
__device__ __forceinline__ void wgemm_pre(const Row& x, const float* __restrict__ Wt, const WPre& pre, int m, int kq, OUT&& out) {
  constexpr int PD = NG < MAGPO_WGEMM_PD ? NG : MAGPO_WGEMM_PD;   // weight fragments in flight ahead of the MFMAs (column groups)
  float4 w[PD][4];
#pragma unroll
  for (int p = 0; p < PD; ++p)
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) w[p][gk] = pre.w[p][gk];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      acc = WMFMA(w[g % PD][gk].x, x.v[4 * gk], acc);
      acc = WMFMA(w[g % PD][gk].y, x.v[4 * gk + 1], acc);
      acc = WMFMA(w[g % PD][gk].z, x.v[4 * gk + 2], acc);
      acc = WMFMA(w[g % PD][gk].w, x.v[4 * gk + 3], acc);
    }
    if (g + PD < NG) {
#pragma unroll
      for (int gk = 0; gk < 4; ++gk) w[g % PD][gk] = wfrag<FRAG>(Wt, g + PD, gk, m, kq);
    }
    out(g, acc);
  }
}
template <int NG, bool FRAG = false, class OUT = void>
__device__ __forceinline__ void wgemm(const Row& x, const float* __restrict__ Wt, int m, int kq, OUT&& out) {
  const WPre pre = wload<NG, FRAG>(Wt, m, kq);
  wgemm_pre<NG, FRAG>(x, Wt, pre, m, kq, out);
}
// The same layer for NT rows at once (the A tokens of an env step, all known up front in the encoder): every weight fragment is fetched
// ONCE and multiplied into NT accumulators -- a quarter of the weight traffic of NT separate calls.  nt (uniform) <= NT rows are live.
template <int NG, int NT, bool FRAG = false, class OUT = void>
__device__ __forceinline__ void wgemm_multi(const Row (&x)[NT], int nt, const float* __restrict__ Wt, int m, int kq, OUT&& out) {
  constexpr int PD = NG < 2 ? NG : 2;
  float4 w[PD][4];
#pragma unroll
  for (int p = 0; p < PD; ++p)
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) w[p][gk] = wfrag<FRAG>(Wt, p, gk, m, kq);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < nt) {
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) {
          acc[t] = WMFMA(w[g % PD][gk].x, x[t].v[4 * gk], acc[t]);
          acc[t] = WMFMA(w[g % PD][gk].y, x[t].v[4 * gk + 1], acc[t]);
          acc[t] = WMFMA(w[g % PD][gk].z, x[t].v[4 * gk + 2], acc[t]);
          acc[t] = WMFMA(w[g % PD][gk].w, x[t].v[4 * gk + 3], acc[t]);
        }
      }
    }
    if (g + PD < NG) {
#pragma unroll
      for (int gk = 0; gk < 4; ++gk) w[g % PD][gk] = wfrag<FRAG>(Wt, g + PD, gk, m, kq);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) out(t, g, acc[t]);
  }
}
// 64 -> 64 layer into registers (+ optional bias)
template <bool FRAG = false>
__device__ __forceinline__ Row dense64(const Row& x, const float* __restrict__ Wt, const float* __restrict__ bias, int m, int kq) {
  Row y;
  wgemm<4, FRAG>(x, Wt, m, kq, [&](int g, f32x4 acc) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = ld4g(bias + 16 * g + 4 * kq);
    y.v[4 * g] = acc[0] + b.x; y.v[4 * g + 1] = acc[1] + b.y; y.v[4 * g + 2] = acc[2] + b.z; y.v[4 * g + 3] = acc[3] + b.w;
  });
  return y;
}
template <bool FRAG = false>
__device__ __forceinline__ Row dense64_pre(const Row& x, const float* __restrict__ Wt, const WPre& pre, const float* __restrict__ bias, int m, int kq) {
  Row y;
  wgemm_pre<4, FRAG>(x, Wt, pre, m, kq, [&](int g, f32x4 acc) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = ld4g(bias + 16 * g + 4 * kq);
    y.v[4 * g] = acc[0] + b.x; y.v[4 * g + 1] = acc[1] + b.y; y.v[4 * g + 2] = acc[2] + b.z; y.v[4 * g + 3] = acc[3] + b.w;
  });
  return y;
}
template <int NT, bool FRAG = false>
__device__ __forceinline__ void dense64_multi(const Row (&x)[NT], int nt, Row (&y)[NT], const float* __restrict__ Wt, const float* __restrict__ bias, int m, int kq) {
  wgemm_multi<4, NT, FRAG>(x, nt, Wt, m, kq, [&](int t, int g, f32x4 acc) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = ld4g(bias + 16 * g + 4 * kq);
    y[t].v[4 * g] = acc[0] + b.x; y[t].v[4 * g + 1] = acc[1] + b.y; y[t].v[4 * g + 2] = acc[2] + b.z; y[t].v[4 * g + 3] = acc[3] + b.w;
  });
}

}  // namespace magpo
