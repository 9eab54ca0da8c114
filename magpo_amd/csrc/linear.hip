// Row-batched dense layers on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
//   k_linear<KIN> : Y[R,NOUT] = act(X[R,KIN] @ W + b), W given transposed ("Wt[NOUT_pad][KIN]").
//                   The same kernel computes dX = dY @ W^T by passing W in its natural [in][out]
//                   layout as "Wt" (replaces XLA's dot_general for flax nn.Dense and the Sable
//                   projections: sable_network.py:93-109,258-284, retention.py:73-75,294, base.py:175-181).
//   k_wgrad<NB>   : dW[KIN,NOUT] partial sums = X^T dY per workgroup slab (+ column sums of dY for
//                   the bias), reduced in fixed order by k_reduce_slabs (bit-stable reruns).
//
// Tile: 64 rows per workgroup (4 waves as 2x2 quadrants of 32x32 accumulators).  X rows are staged
// once in LDS ([64][KIN+4] floats, ds_read_b128 per lane conflict-free); B fragments come
// straight from L2 into VGPRs (each lane owns one output column and 32 contiguous k's).
#include "common.hpp"
#include <stdlib.h>

namespace magpo {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SWISH = 3, ACT_MASKPOS = 4 };   // MASKPOS: y = aux > 0 ? y : 0 (ReLU backward fused into dX = dY W^T)

template <int KIN>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ X, int ldx,
                                                const float* __restrict__ Wt, const float* __restrict__ bias,
                                                float* __restrict__ Y, int ldy, float* __restrict__ Ypre,
                                                int R, int NOUT, int act) {
  constexpr int LD = KIN + LDP;
  extern __shared__ __align__(16) float xs[];  // [64][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, h = lane >> 5;
  const long row0 = (long)blockIdx.x * 64;

  // stage X tile
  constexpr int F4 = KIN / 4;
  for (int i = tid; i < 64 * F4; i += 256) {
    int r = i / F4, c4 = i - r * F4;
    long gr = row0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < R) v = *reinterpret_cast<const float4*>(X + gr * (long)ldx + 4 * c4);
    *reinterpret_cast<float4*>(&xs[r * LD + 4 * c4]) = v;
  }
  __syncthreads();

  const float* arow = &xs[(32 * wr + lr) * LD + 32 * h];
  const int nblk = (NOUT + 63) >> 6;
  for (int nb = 0; nb < nblk; ++nb) {
    const int n = nb * 64 + 32 * wc + lr;
    if (nb * 64 + 32 * wc >= NOUT) continue;  // wave-uniform
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* wrow = Wt + (long)n * KIN + 32 * h;  // Wt is padded to a multiple of 32 rows
#pragma unroll 1
    for (int kc = 0; kc < KIN / 64; ++kc) {
      float4 b[8], a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(wrow + kc * 64 + 4 * u);
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(arow + kc * 64 + 4 * u);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
      }
    }
    if (n < NOUT) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        long gr = row0 + 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (gr < R) {
          float v = acc[i] + bv;
          if (Ypre) Ypre[gr * (long)ldy + n] = v;
          if (act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (act == ACT_GELU) v = gelu_tanh(v);
          else if (act == ACT_SWISH) v = swishf_(v);
          Y[gr * (long)ldy + n] = v;
        }
      }
    }
  }
}

// Wave-autonomous variant for KIN in {64, 128}: every wave owns 32*NCB output columns, keeps their weight
// fragments in VGPRs for the whole persistent loop and streams 32-row activation tiles straight from HBM
// (each lane reads one contiguous half-row, a wave one contiguous 32 x KIN block).  No LDS, no barriers:
// latency is hidden by several independent waves per SIMD.
template <int KIN, int NCB>
__global__ __launch_bounds__(256) void k_linear_w(const float* __restrict__ X, int ldx, const float* __restrict__ Wt,
                                                  const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                  float* __restrict__ Ypre, int R, int NOUT, int act) {
  constexpr int KH = KIN / 2;  // k values per lane half
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, h = lane >> 5;
  const int c0 = (blockIdx.y * (blockDim.x >> 6) + wave) * 32 * NCB;
  if (c0 >= NOUT) return;
  float4 wf[NCB][KH / 4];
  float bv[NCB];
#pragma unroll
  for (int b = 0; b < NCB; ++b) {
    const int n = c0 + 32 * b + lr;
    const bool on = c0 + 32 * b < NOUT;  // Wt rows are padded to a multiple of 32
    bv[b] = (bias && n < NOUT) ? bias[n] : 0.f;
#pragma unroll
    for (int u = 0; u < KH / 4; ++u)
      wf[b][u] = on ? *reinterpret_cast<const float4*>(Wt + (long)n * KIN + h * KH + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int ntiles = (R + 31) >> 5;
  float4 af[KH / 4], an[KH / 4];
  auto fetch = [&](int tile, float4* dst) {
    // unconditional loads from a clamped row (rows past the end are never stored): no exec-masked block in the pipeline
    const long row = min((long)tile * 32 + lr, (long)R - 1);
    const float* xp = X + row * (long)ldx + h * KH;
#pragma unroll
    for (int u = 0; u < KH / 4; ++u) dst[u] = *reinterpret_cast<const float4*>(xp + 4 * u);
  };
  fetch(blockIdx.x, af);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    fetch(tile + gridDim.x, an);  // next tile's activations are in flight while this tile's MFMAs run
    f32x16 acc[NCB];
#pragma unroll
    for (int b = 0; b < NCB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
    for (int u = 0; u < KH / 4; ++u) {
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].x, wf[b][u].x, acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].y, wf[b][u].y, acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].z, wf[b][u].z, acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].w, wf[b][u].w, acc[b], 0, 0, 0);
    }
    if (!Ypre && act <= ACT_RELU && (long)tile * 32 + 32 <= R && c0 + 32 * NCB <= NOUT) {
      // the common case (whole tile, plain / relu output): straight-line stores, no per-element exec masking
#pragma unroll
      for (int b = 0; b < NCB; ++b) {
        float* yp = Y + ((long)tile * 32 + 4 * h) * (long)ldy + c0 + 32 * b + lr;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = acc[b][i] + bv[b];
          if (act == ACT_RELU) v = fmaxf(v, 0.f);
          yp[(long)((i & 3) + 8 * (i >> 2)) * ldy] = v;
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < NCB; ++b) {
        const int n = c0 + 32 * b + lr;
        if (n < NOUT) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const long gr = (long)tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (gr < R) {
              float v = acc[b][i] + bv[b];
              if (Ypre) Ypre[gr * (long)ldy + n] = v;
              if (act == ACT_RELU) v = fmaxf(v, 0.f);
              else if (act == ACT_GELU) v = gelu_tanh(v);
              else if (act == ACT_SWISH) v = swishf_(v);
              Y[gr * (long)ldy + n] = v;
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < KH / 4; ++u) af[u] = an[u];
  }
}

// ---------------------------------------------------------------------------------------------------
// Prologue-fused skinny GEMM for acting (64 -> NOUT): the row-wise op that precedes a dense layer in the
// Sable blocks is evaluated on the activation fragment itself.  In this kernel a lane holds one half
// (32 consecutive features) of one row, so a row statistic is a 32-term local sum plus one shuffle
// with the partner lane (lane ^ 32).  Side outputs (normalised rows needed later) are stored from here.
//   PRO_EMBED_ACT : z = W_act[idx] -> gelu -> rmsnorm*s1 (=xa, stored) -> +pe          (sable_network.py:306-307)
//   PRO_EMBED_OBS : z = (rmsnorm_F(obs)*s_obs) @ W_obs -> gelu -> rmsnorm*s1 (=xn, stored) -> +pe   (:93-101,132)
//   PRO_RESNORM   : x = rmsnorm(a + y)*s1 [-> rmsnorm*s2] (stored) [-> +pe (stored)]     (:69-70, 203, 214-215)
//   PRO_HEADMID   : h = rmsnorm(gelu(hpre))*s1                                            (:102-109, 277-284)
enum { PRO_EMBED_ACT = 1, PRO_EMBED_OBS = 2, PRO_RESNORM = 3, PRO_HEADMID = 4 };
struct ProArgs {
  const float* a; long lda;          // primary input rows (RESNORM: a, HEADMID: hpre, EMBED_OBS: obs [ld = lda, F features])
  const float* y; long ldy_;         // RESNORM: second addend
  const float* s1; const float* s2;  // norm scales (s2 optional)
  const float* pe; const int* pos; long pos_stride; int npos; int use_pe;   // A operand gets + pe[pos[row]] when use_pe
  const float* W; const int* idx; long idx_stride;   // EMBED_ACT: W_act [K+1][64]; EMBED_OBS: W_obs [F][64]
  const float* s_obs; int F;
  float* out; long ldout;            // normalised row (before pe), nullable
  float* outpe; long ldoutpe;        // row + pe, nullable
};

__device__ __forceinline__ float half_row_mean_sq(const float4* v) {
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
  s += __shfl_xor(s, 32, 64);
  return s * (1.0f / 64.0f);
}
__device__ __forceinline__ void rms_inplace(float4* v, const float* __restrict__ scale, int h) {
  const float rstd = rsqrtf(half_row_mean_sq(v) + 1e-6f);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float4 sc = *reinterpret_cast<const float4*>(scale + 32 * h + 4 * u);
    v[u].x *= rstd * sc.x; v[u].y *= rstd * sc.y; v[u].z *= rstd * sc.z; v[u].w *= rstd * sc.w;
  }
}
__device__ __forceinline__ void gelu_inplace(float4* v) {
#pragma unroll
  for (int u = 0; u < 8; ++u) { v[u].x = gelu_tanh(v[u].x); v[u].y = gelu_tanh(v[u].y); v[u].z = gelu_tanh(v[u].z); v[u].w = gelu_tanh(v[u].w); }
}

template <int PRO, int NCB>
__global__ __launch_bounds__(256) void k_linear_pro(ProArgs p, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                    float* __restrict__ Y, long ldy, int R, int NOUT) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, h = lane >> 5;
  const int c0 = (blockIdx.y * 4 + wave) * 32 * NCB;
  if (c0 >= NOUT) return;
  const bool writer = c0 == 0;  // exactly one wave per row tile stores the side outputs
  float4 wf[NCB][8];
  float bv[NCB];
#pragma unroll
  for (int b = 0; b < NCB; ++b) {
    const int n = c0 + 32 * b + lr;
    const bool on = c0 + 32 * b < NOUT;
    bv[b] = (bias && n < NOUT) ? bias[n] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      wf[b][u] = on ? *reinterpret_cast<const float4*>(Wt + (long)n * 64 + 32 * h + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int ntiles = (R + 31) >> 5;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long row = (long)tile * 32 + lr;
    const bool ok = row < R;
    const long rr = ok ? row : 0;
    float4 af[8];
    if (PRO == PRO_EMBED_ACT) {
      const float* wp = p.W + (long)p.idx[rr * p.idx_stride] * 64 + 32 * h;
#pragma unroll
      for (int u = 0; u < 8; ++u) af[u] = *reinterpret_cast<const float4*>(wp + 4 * u);
      gelu_inplace(af);
      rms_inplace(af, p.s1, h);
    } else if (PRO == PRO_EMBED_OBS) {
      const float* o = p.a + rr * p.lda;
      float ms = 0.f;
      for (int f = 0; f < p.F; ++f) ms += o[f] * o[f];
      const float rstd = rsqrtf(ms / (float)p.F + 1e-6f);
#pragma unroll
      for (int u = 0; u < 8; ++u) af[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int f = 0; f < p.F; ++f) {
        const float of = o[f] * rstd * p.s_obs[f];
        const float* wp = p.W + (long)f * 64 + 32 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 w = *reinterpret_cast<const float4*>(wp + 4 * u);
          af[u].x += of * w.x; af[u].y += of * w.y; af[u].z += of * w.z; af[u].w += of * w.w;
        }
      }
      gelu_inplace(af);
      rms_inplace(af, p.s1, h);
    } else if (PRO == PRO_RESNORM) {
      const float* ap = p.a + rr * p.lda + 32 * h;
#pragma unroll
      for (int u = 0; u < 8; ++u) af[u] = *reinterpret_cast<const float4*>(ap + 4 * u);
      if (p.y) {  // null addend = plain norm (input of encoder blocks > 0)
        const float* yp = p.y + rr * p.ldy_ + 32 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 y = *reinterpret_cast<const float4*>(yp + 4 * u);
          af[u].x += y.x; af[u].y += y.y; af[u].z += y.z; af[u].w += y.w;
        }
      }
      rms_inplace(af, p.s1, h);
      if (p.s2) rms_inplace(af, p.s2, h);
    } else {  // PRO_HEADMID
      const float* ap = p.a + rr * p.lda + 32 * h;
#pragma unroll
      for (int u = 0; u < 8; ++u) af[u] = *reinterpret_cast<const float4*>(ap + 4 * u);
      gelu_inplace(af);
      rms_inplace(af, p.s1, h);
    }
    if (writer && ok && p.out) {
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(p.out + row * p.ldout + 32 * h + 4 * u) = af[u];
    }
    if (p.use_pe || p.outpe) {
      int ps = p.pos[rr * p.pos_stride];
      ps = ps < 0 ? 0 : (ps >= p.npos ? p.npos - 1 : ps);
      const float* pp = p.pe + (long)ps * 64 + 32 * h;
      float4 ape[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 e = *reinterpret_cast<const float4*>(pp + 4 * u);
        ape[u] = make_float4(af[u].x + e.x, af[u].y + e.y, af[u].z + e.z, af[u].w + e.w);
      }
      if (writer && ok && p.outpe) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(p.outpe + row * p.ldoutpe + 32 * h + 4 * u) = ape[u];
      }
      if (p.use_pe) {
#pragma unroll
        for (int u = 0; u < 8; ++u) af[u] = ape[u];
      }
    }
    f32x16 acc[NCB];
#pragma unroll
    for (int b = 0; b < NCB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].x, wf[b][u].x, acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].y, wf[b][u].y, acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].z, wf[b][u].z, acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u].w, wf[b][u].w, acc[b], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < NCB; ++b) {
      const int n = c0 + 32 * b + lr;
      if (n < NOUT) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const long gr = (long)tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (gr < R) Y[gr * ldy + n] = acc[b][i] + bv[b];
        }
      }
    }
  }
}

// Wide-input variant (KIN = 192 / 256 / 384, the dX GEMMs of the fused projections and of the GRU input
// layer): a wave owns 32 output columns and keeps ALL their weight fragments in VGPRs (KIN/2 registers);
// activations stream in 64-wide k-chunks (one contiguous 128-B piece per lane and chunk).
template <int KIN>
__global__ __launch_bounds__(256) void k_linear_wk(const float* __restrict__ X, int ldx, const float* __restrict__ Wt,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                   int R, int NOUT, int act) {
  constexpr int NKC = KIN / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, h = lane >> 5;
  const int c0 = (blockIdx.y * (blockDim.x >> 6) + wave) * 32;
  if (c0 >= NOUT) return;
  const int n = c0 + lr;
  float4 wf[NKC][8];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
    for (int u = 0; u < 8; ++u) wf[kc][u] = *reinterpret_cast<const float4*>(Wt + (long)n * KIN + kc * 64 + 32 * h + 4 * u);
  const float bv = (bias && n < NOUT) ? bias[n] : 0.f;
  const int ntiles = (R + 31) >> 5;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long row = (long)tile * 32 + lr;
    const bool ok = row < R;
    const float* xp = X + (ok ? row : 0) * (long)ldx + 32 * h;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 af[2][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) af[0][u] = ok ? *reinterpret_cast<const float4*>(xp + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (kc + 1 < NKC) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          af[(kc + 1) & 1][u] = ok ? *reinterpret_cast<const float4*>(xp + (kc + 1) * 64 + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][u].x, wf[kc][u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][u].y, wf[kc][u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][u].z, wf[kc][u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][u].w, wf[kc][u].w, acc, 0, 0, 0);
      }
    }
    if (act <= ACT_RELU && (long)tile * 32 + 32 <= R && c0 + 32 <= NOUT) {
      // the common case (whole tile, plain / relu output): straight-line stores, no per-element exec masking
      float* yp = Y + ((long)tile * 32 + 4 * h) * (long)ldy + n;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = acc[i] + bv;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        yp[(long)((i & 3) + 8 * (i >> 2)) * ldy] = v;
      }
    } else if (n < NOUT) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const long gr = (long)tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (gr < R) {
          float v = acc[i] + bv;
          if (act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (act == ACT_GELU) v = gelu_tanh(v);
          else if (act == ACT_SWISH) v = swishf_(v);
          Y[gr * (long)ldy + n] = v;
        }
      }
    }
  }
}

// Wide-input dense layer, second form: weights register-resident as in k_linear_wk, but the 32-row activation tile is
// fetched ONCE per workgroup with fully coalesced loads and shared by the 4 waves through a double-buffered LDS tile.
// k_linear_wk lets every wave fetch its own copy of the tile as per-lane 128-B row pieces: 64 cache lines touched by each
// load instruction, 4096 line requests per tile and workgroup -- the L1 tag rate, not MFMA or HBM, bounds it (~50 % of the
// MFMA peak).  Here a tile costs KIN/4 coalesced line requests and one barrier.
template <int KIN, int NW, bool BF3 = false>
__global__ __launch_bounds__(64 * NW) void k_linear_lds(const float* __restrict__ X, int ldx, const float* __restrict__ Wt,
                                                    const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                    int R, int NOUT, int act, const float* __restrict__ aux) {
  constexpr int NKC = KIN / 64, PT = KIN + LDP, NT = 64 * NW, NLD = 32 * (KIN / 4) / NT;   // NLD float4 per thread and tile
  // BF3 (per-call variant bit 2): the GEMM on v_mfma_f32_32x32x16_bf16 with BOTH operands split into three bf16 pieces (24 mantissa bits =
  // what an fp32 operand holds; six products hh hm mh hl lh mm, fp32 accumulate): 6/16 of the fp32 MFMA time at fp32 accuracy
  // (tests/test_kernels_gpu.py::test_linear_bf16_triples_keep_fp32_accuracy).  The weight fragments are split once per kernel, a row tile once
  // per workgroup when it is stashed (three bf16 planes [32][KIN + 8] instead of one fp32 tile); k of MFMA step s = (KIN / 2) h + 8 s + j.
  constexpr int PB = KIN + 8, NS = KIN / 16;
  extern __shared__ __align__(16) float ll_smem[];                // 2 x [32][PT]  (BF3: 2 x 3 x [32][PB] bf16)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int c0 = (blockIdx.y * NW + wave) * 32;
  const bool colon = c0 < NOUT;                                   // waves past the last column group only help with the loads
  const int n = c0 + lr;
  float4 wf[BF3 ? 1 : NKC][8];
  bf16x8 wp[BF3 ? 3 : 1][BF3 ? NS : 1];
  if constexpr (!BF3) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        wf[kc][u] = colon ? *reinterpret_cast<const float4*>(Wt + (long)n * KIN + kc * 64 + 32 * h + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
#pragma unroll
    for (int s8 = 0; s8 < NS; ++s8) {
      const float* w = Wt + (long)n * KIN + (KIN / 2) * h + 8 * s8;
      const float4 w0 = colon ? *reinterpret_cast<const float4*>(w) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 w1 = colon ? *reinterpret_cast<const float4*>(w + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 pc[3];
        split_pieces<3>(wv[j], pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) wp[q][s8][j] = pc[q];
      }
    }
  }
  const float bv = (bias && n < NOUT) ? bias[n] : 0.f;
  const int ntiles = (R + 31) >> 5;
  float4 pre[NLD];
  // UB (the one-wave-per-SIMD widths): uniform tile base in scalar registers + per-thread 32-bit offsets computed once.  The
  // generic form spends ~10 VALU instructions per load / store on 64-bit addresses and clamps, and f32 MFMA and VALU time
  // add up; at KIN <= 128 the extra registers would cost a wave of occupancy, which matters more there.
  constexpr bool UB = KIN >= 192;
  // LITE (KIN == 64, four waves per SIMD): the same idea without per-thread tables -- the row part of a store address is a
  // scalar, loads are min(local row, last valid row) * ldx + column.  (At KIN == 128 either form costs a wave of occupancy
  // and was slower: 128->384 3.92 -> 4.10 ms.)
  constexpr bool LITE = KIN == 64;
  const unsigned lt_r = (unsigned)(tid / (KIN / 4)), lt_c = 4u * (unsigned)(tid % (KIN / 4)), lt_o = (unsigned)(4 * h) * (unsigned)ldy + (unsigned)n;
  unsigned xr[UB ? NLD : 1], xc[UB ? NLD : 1], so[UB ? 16 : 1];
  if constexpr (UB) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) { xr[j] = (unsigned)((tid + NT * j) / (KIN / 4)); xc[j] = 4u * (unsigned)((tid + NT * j) % (KIN / 4)); }
#pragma unroll
    for (int i = 0; i < 16; ++i) so[i] = (unsigned)((i & 3) + 8 * (i >> 2) + 4 * h) * (unsigned)ldy + (unsigned)n;
  }
#define LL_FETCH(TILE)                                                                                   \
  if constexpr (LITE) {                                                                                  \
    const long r0_ = (long)(TILE) * 32;                                                                  \
    const float* xb_ = X + r0_ * ldx;                                                                    \
    const unsigned rmax_ = (unsigned)min(31L, (long)R - 1 - r0_);                                        \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j)                                                      \
      pre[j] = *reinterpret_cast<const float4*>(xb_ + (min(lt_r + (unsigned)(j * (NT / (KIN / 4))), rmax_) * (unsigned)ldx + lt_c)); \
  } else if constexpr (UB) {                                                                             \
    const long r0_ = (long)(TILE) * 32;                                                                  \
    const float* xb_ = X + r0_ * ldx;                                                                    \
    const unsigned rmax_ = (unsigned)min(31L, (long)R - 1 - r0_);   /* rows past the end re-read the last valid one */ \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j)                                                      \
      pre[j] = *reinterpret_cast<const float4*>(xb_ + (min(xr[j], rmax_) * (unsigned)ldx + xc[j]));      \
  } else {                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                    \
      const int idx = tid + NT * j;                                                                      \
      const int row = min((int)(TILE) * 32 + idx / (KIN / 4), R - 1);   /* 32-bit clamp, one 64-bit multiply-add */ \
      pre[j] = *reinterpret_cast<const float4*>(X + (long)row * ldx + 4 * (idx % (KIN / 4)));            \
    }                                                                                                    \
  }
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LL_STASH(BUF)                                                                                    \
  _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                      \
    const int idx = tid + NT * j;                                                                        \
    if constexpr (!BF3) {                                                                                \
      *reinterpret_cast<float4*>(&(BUF)[(idx / (KIN / 4)) * PT + 4 * (idx % (KIN / 4))]) = pre[j];       \
    } else {                                                                                             \
      const float v4_[4] = {pre[j].x, pre[j].y, pre[j].z, pre[j].w};                                     \
      bf16x4 pl_[3];                                                                                     \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
        __bf16 pc_[3];                                                                                   \
        split_pieces<3>(v4_[e], pc_);                                                                    \
        pl_[0][e] = pc_[0]; pl_[1][e] = pc_[1]; pl_[2][e] = pc_[2];                                      \
      }                                                                                                  \
      __bf16* bb_ = reinterpret_cast<__bf16*>(BUF) + (idx / (KIN / 4)) * PB + 4 * (idx % (KIN / 4));     \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(bb_ + q * 32 * PB) = pl_[q]; \
    }                                                                                                    \
  }
  LL_FETCH(blockIdx.x)
  LL_STASH(ll_smem)
  __syncthreads();
  int par = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const float* buf = ll_smem + par * (BF3 ? 3 * 32 * PB / 2 : 32 * PT);
    LL_FETCH(min(tile + (int)gridDim.x, ntiles - 1))   // in flight under this tile's MFMAs
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if constexpr (!BF3) {
      const float* ap = buf + lr * PT + 32 * h;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 av = *reinterpret_cast<const float4*>(ap + kc * 64 + 4 * u);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wf[kc][u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wf[kc][u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wf[kc][u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wf[kc][u].w, acc, 0, 0, 0);
        }
      }
    } else {
      const __bf16* ab = reinterpret_cast<const __bf16*>(buf) + lr * PB + (KIN / 2) * h;
#pragma unroll
      for (int s8 = 0; s8 < NS; ++s8) {
        bf16x8 xp[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) xp[q] = *reinterpret_cast<const bf16x8*>(ab + q * 32 * PB + 8 * s8);
        // products in decreasing order of magnitude: (0,0) (0,1) (1,0) (0,2) (2,0) (1,1)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[0], wp[0][s8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[0], wp[1][s8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[1], wp[0][s8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[0], wp[2][s8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[2], wp[0][s8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[1], wp[1][s8], acc, 0, 0, 0);
      }
    }
    float* nbuf = ll_smem + (par ^ 1) * (BF3 ? 3 * 32 * PB / 2 : 32 * PT);     // the other buffer: nobody reads it during this tile
    LL_STASH(nbuf)
    if (colon) {
      if ((act <= ACT_RELU || act == ACT_MASKPOS) && (long)tile * 32 + 32 <= R && c0 + 32 <= NOUT) {
        const long o0 = ((long)tile * 32 + 4 * h) * (long)ldy + n;
        if constexpr (LITE) {
#define LL_ROWP(P, I) ((P) + ((long)tile * 32 + ((I) & 3) + 8 * ((I) >> 2)) * ldy)
          if (act == ACT_MASKPOS) {
            float mk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) mk[i] = LL_ROWP(aux, i)[lt_o];
#pragma unroll
            for (int i = 0; i < 16; ++i) LL_ROWP(Y, i)[lt_o] = mk[i] > 0.f ? acc[i] + bv : 0.f;
          } else if (act == ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) LL_ROWP(Y, i)[lt_o] = fmaxf(acc[i] + bv, 0.f);
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) LL_ROWP(Y, i)[lt_o] = acc[i] + bv;
          }
#undef LL_ROWP
        } else if constexpr (UB) {
          float* yt = Y + (long)tile * 32 * ldy;
          if (act == ACT_MASKPOS) {
            const float* mt = aux + (long)tile * 32 * ldy;
            float mk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) mk[i] = mt[so[i]];
#pragma unroll
            for (int i = 0; i < 16; ++i) yt[so[i]] = mk[i] > 0.f ? acc[i] + bv : 0.f;
          } else if (act == ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) yt[so[i]] = fmaxf(acc[i] + bv, 0.f);
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) yt[so[i]] = acc[i] + bv;
          }
        } else if (act == ACT_MASKPOS) {   // mask values first (16 independent loads), then the stores
          float mk[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) mk[i] = aux[o0 + (long)((i & 3) + 8 * (i >> 2)) * ldy];
#pragma unroll
          for (int i = 0; i < 16; ++i) Y[o0 + (long)((i & 3) + 8 * (i >> 2)) * ldy] = mk[i] > 0.f ? acc[i] + bv : 0.f;
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float v = acc[i] + bv;
            if (act == ACT_RELU) v = fmaxf(v, 0.f);
            Y[o0 + (long)((i & 3) + 8 * (i >> 2)) * ldy] = v;
          }
        }
      } else if (n < NOUT) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const long gr = (long)tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (gr < R) {
            float v = acc[i] + bv;
            if (act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == ACT_GELU) v = gelu_tanh(v);
            else if (act == ACT_SWISH) v = swishf_(v);
            else if (act == ACT_MASKPOS) v = aux[gr * (long)ldy + n] > 0.f ? v : 0.f;
            Y[gr * (long)ldy + n] = v;
          }
        }
      }
    }
    __syncthreads();   // next tile stashed by everyone, this tile's LDS reads done
    par ^= 1;
  }
#undef LL_FETCH
#undef LL_STASH
}

// dW slab: grid (G, ceil(NOUT / (64 NB)), KIN / 64).  Software pipeline: the next row tile is fetched
// from HBM into registers while the MFMAs of the current tile run out of LDS.
#ifdef MAGPO_WG_PROF
__device__ unsigned long long g_wg_prof[8];
#define WP_DECL() unsigned long long wp_acc[4] = {0, 0, 0, 0}; unsigned long long wp_last = clock64();
#define WP(k) do { unsigned long long t_ = clock64(); wp_acc[k] += t_ - wp_last; wp_last = t_; } while (0)
#define WP_FLUSH() do { if (threadIdx.x == 0 && (blockIdx.x & 31) == 0 && blockIdx.y == 0 && blockIdx.z == 0) { for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&g_wg_prof[k_], wp_acc[k_]); } } while (0)
#else
#define WP_DECL()
#define WP(k)
#define WP_FLUSH()
#endif

template <int NB>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ X, int ldx, const float* __restrict__ dY, int ldy,
                                               int R, int KIN, int NOUT, float* __restrict__ slab,
                                               float* __restrict__ bias_slab) {
  constexpr int LDX = 64 + LDP, LDY = 64 * NB + LDP;
  __shared__ __align__(16) float xs[64 * LDX];
  __shared__ __align__(16) float ys[64 * LDY];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, h = lane >> 5;
  const int g = blockIdx.x, G = gridDim.x;
  const int c0 = blockIdx.y * 64 * NB, kb = blockIdx.z;
  const int ntiles = (R + 63) >> 6;
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  float bsum = 0.f;
  const bool do_bias = bias_slab != nullptr && kb == 0;
  const bool full_cols = (c0 + 64 * NB <= NOUT) && ((ldy & 3) == 0);

  WP_DECL();
  if (full_cols) {
    float4 nx[4 + 4 * NB];   // next tile: 4 float4 of X, then 4 NB of dY
    // Fast path (whole column block, aligned rows): the next tile's 4 + 4 NB loads are issued ONE AT A TIME between the
    // MFMA steps of the current tile.  Issued as a burst they sit in front of the MFMA loop in the wave's in-order
    // instruction stream while the memory pipeline back-pressures (in-kernel timing: the burst took as long as the
    // MFMA loop itself).  Loads are unconditional from clamped rows; rows past the end are zeroed when stashed.
    constexpr int YR = 16 * NB;                      // float4 per dY row of this column block
    const int rx = tid >> 4, cx = 4 * (tid & 15);    // X: float4 j of this thread is row rx + 16 j
    const int ry = tid / YR, cy = 4 * (tid % YR);    // dY: float4 j is row ry + (256 / YR) j
    const float* xb = X + kb * 64 + cx;
    const float* yb = dY + c0 + cy;
#define WG_LOADX(J, ROW0) nx[J] = *reinterpret_cast<const float4*>(xb + (long)min((int)(ROW0) + rx + 16 * (J), R - 1) * ldx)
#define WG_LOADY(J, ROW0) nx[4 + (J)] = *reinterpret_cast<const float4*>(yb + (long)min((int)(ROW0) + ry + (256 / YR) * (J), R - 1) * ldy)
    {
      const long row0 = (long)min(g, ntiles - 1) * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j) WG_LOADX(j, row0);
#pragma unroll
      for (int j = 0; j < 4 * NB; ++j) WG_LOADY(j, row0);
    }
    for (int tile = g; tile < ntiles; tile += G) {
      const long row0 = (long)tile * 64;
      const long row0n = (long)min(tile + G, ntiles - 1) * 64;
      __syncthreads();          // MFMAs of the previous tile have finished reading LDS
      WP(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // (component-wise selects: a select between &nx[j] and a zero slot would pin nx[] in scratch)
        float4 v = nx[j];
        const bool ok = row0 + rx + 16 * j < R;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        *reinterpret_cast<float4*>(&xs[(rx + 16 * j) * LDX + cx]) = v;
      }
#pragma unroll
      for (int j = 0; j < 4 * NB; ++j) {
        float4 v = nx[4 + j];
        const bool ok = row0 + ry + (256 / YR) * j < R;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        *reinterpret_cast<float4*>(&ys[(ry + (256 / YR) * j) * LDY + cy]) = v;
      }
      __syncthreads();
      WP(1);
      WP(2);
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int tok = 32 * h + s;
        const float a = xs[tok * LDX + 32 * wr + lr];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float bb = ys[tok * LDY + 64 * b + 32 * wc + lr];
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[b], 0, 0, 0);
        }
        // one load of the next tile every second MFMA step (after unrolling, s and the register index are constants)
        if ((s & 1) && (s >> 1) < 4 + 4 * NB) {
          const int li = s >> 1;
          const float* src = li < 4 ? xb + min(row0n + rx + 16 * li, (long)R - 1) * (long)ldx
                                    : yb + min(row0n + ry + (256 / YR) * (li - 4), (long)R - 1) * (long)ldy;
          nx[li] = *reinterpret_cast<const float4*>(src);
        }
      }
      WP(3);
      if (do_bias && tid < 64 * NB) {
        float sb = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) sb += ys[r * LDY + tid];
        bsum += sb;
      }
    }
#undef WG_LOADX
#undef WG_LOADY
  } else {
    float4 xq[4], yq[4 * NB];
    auto fetch = [&](int tile) {
      const long row0 = (long)tile * 64;
  #pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        const int r = i >> 4, c4 = i & 15;
        const long gr = row0 + r;
        xq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < R) xq[j] = *reinterpret_cast<const float4*>(X + gr * (long)ldx + kb * 64 + 4 * c4);
      }
  #pragma unroll
      for (int j = 0; j < 4 * NB; ++j) {
        const int i = tid + 256 * j;
        const int r = i / (16 * NB), c4 = i - r * (16 * NB);
        const long gr = row0 + r;
        const int col = c0 + 4 * c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < R) {
          const float* p = dY + gr * (long)ldy + col;
          if (full_cols || col + 3 < NOUT) v = *reinterpret_cast<const float4*>(p);
          else {
            if (col + 0 < NOUT) v.x = p[0];
            if (col + 1 < NOUT) v.y = p[1];
            if (col + 2 < NOUT) v.z = p[2];
          }
        }
        yq[j] = v;
      }
    };
    auto stash = [&]() {
  #pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        *reinterpret_cast<float4*>(&xs[(i >> 4) * LDX + 4 * (i & 15)]) = xq[j];
      }
  #pragma unroll
      for (int j = 0; j < 4 * NB; ++j) {
        const int i = tid + 256 * j;
        const int r = i / (16 * NB), c4 = i - r * (16 * NB);
        *reinterpret_cast<float4*>(&ys[r * LDY + 4 * c4]) = yq[j];
      }
    };

    if (g < ntiles) fetch(g);
    for (int tile = g; tile < ntiles; tile += G) {
      __syncthreads();          // MFMAs of the previous tile have finished reading LDS
      stash();
      __syncthreads();
      if (tile + G < ntiles) fetch(tile + G);   // in flight during the MFMA loop below
#pragma unroll 4
      for (int s = 0; s < 32; ++s) {
        const int tok = 32 * h + s;
        const float a = xs[tok * LDX + 32 * wr + lr];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float bb = ys[tok * LDY + 64 * b + 32 * wc + lr];
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[b], 0, 0, 0);
        }
      }
      if (do_bias && tid < 64 * NB) {
        float sb = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) sb += ys[r * LDY + tid];
        bsum += sb;
      }
    }
  }
  WP_FLUSH();
  float* out = slab + (long)g * KIN * NOUT;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = c0 + 64 * b + 32 * wc + lr;
    if (n < NOUT) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int kr = kb * 64 + 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        out[(long)kr * NOUT + n] = acc[b][i];
      }
    }
  }
  if (do_bias && tid < 64 * NB && c0 + tid < NOUT) bias_slab[(long)g * NOUT + c0 + tid] = bsum;
}

// out[p] = scale * sum_g slab[g*stride + p]  (fixed summation tree => bit-stable).
// Block = 64 columns x 16 row-lanes; each row-lane sums every 16th slab, then a fixed LDS tree.
__global__ __launch_bounds__(1024) void k_reduce_slabs(const float* __restrict__ slab, float* __restrict__ out, int G, long P, long stride,
                                                       float scale, int accumulate) {
  __shared__ float sh[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long p = (long)blockIdx.x * 64 + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (p < P) {
    int g = ty;
    for (; g + 48 < G; g += 64) {
      s0 += slab[(long)g * stride + p];
      s1 += slab[(long)(g + 16) * stride + p];
      s2 += slab[(long)(g + 32) * stride + p];
      s3 += slab[(long)(g + 48) * stride + p];
    }
    for (; g < G; g += 16) s0 += slab[(long)g * stride + p];
  }
  sh[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && p < P) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += sh[i][tx];
    s *= scale;
    out[p] = accumulate ? out[p] + s : s;
  }
}

// Wt[Npad][K] = W[K][N]^T, rows N..Npad-1 zero.
__global__ void k_transpose_pad(const float* __restrict__ W, float* __restrict__ Wt, int K, int N, int Npad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad * K) return;
  int n = i / K, k = i - n * K;
  Wt[i] = n < N ? W[(long)k * N + n] : 0.f;
}

// dW slab, whole-matrix form: ONE workgroup accumulates the full [KIN x NOUT] block for its row slab (wave w owns NOUT/4
// columns, all KIN rows: KT x NT accumulator tiles of 32x32, up to 192 AGPRs), so every row of X and dY is read from HBM
// exactly once.  k_wgrad splits a slab over (NOUT/128) x (KIN/64) workgroups that re-read the same rows and only partly meet
// in L2 (PMC: 1.6x the algorithmic bytes), and it needs 1.5 LDS operand reads per MFMA where this form needs
// (KT + NT) / (KT NT).  One workgroup per CU (LDS: a 64-row tile of X and of dY); the next tile's loads are issued one at
// a time between the MFMA steps.
template <int KT, int NT>
__global__ __launch_bounds__(256, 1) void k_wgrad_full(const float* __restrict__ X, int ldx, const float* __restrict__ dY, int ldy,
                                                       int R, float* __restrict__ slab, float* __restrict__ bias_slab) {
  constexpr int KIN = 32 * KT, NOUT = 128 * NT, LDX = KIN + LDP, LDY = NOUT + LDP;
  constexpr int NX = 64 * KIN / 4 / 256, NY = 64 * NOUT / 4 / 256;   // float4 per thread and tile
  extern __shared__ __align__(16) float wf_smem[];
  float* xs = wf_smem;              // [64][LDX]
  float* ys = xs + 64 * LDX;        // [64][LDY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int g = blockIdx.x, G = gridDim.x;
  const int ntiles = (R + 63) >> 6;
  f32x16 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[kt][nt][i] = 0.f;
  float bsum[(NOUT + 255) / 256];
#pragma unroll
  for (int j = 0; j < (NOUT + 255) / 256; ++j) bsum[j] = 0.f;
  float4 nx[NX + NY];
#define WF_SRC(LI, ROW0)                                                                                                    \
  ((LI) < NX ? X + min((ROW0) + (tid + 256 * (LI)) / (KIN / 4), (long)R - 1) * (long)ldx + 4 * ((tid + 256 * (LI)) % (KIN / 4)) \
             : dY + min((ROW0) + (tid + 256 * ((LI) - NX)) / (NOUT / 4), (long)R - 1) * (long)ldy + 4 * ((tid + 256 * ((LI) - NX)) % (NOUT / 4)))
  {
    const long row0 = (long)min(g, ntiles - 1) * 64;
#pragma unroll
    for (int li = 0; li < NX + NY; ++li) nx[li] = *reinterpret_cast<const float4*>(WF_SRC(li, row0));
  }
  for (int tile = g; tile < ntiles; tile += G) {
    const long row0 = (long)tile * 64;
    const long row0n = (long)min(tile + G, ntiles - 1) * 64;
    __syncthreads();          // MFMAs of the previous tile have finished reading LDS
#pragma unroll
    for (int li = 0; li < NX + NY; ++li) {   // stash (rows past the end are zeroed; component-wise selects keep nx[] in registers)
      float4 v = nx[li];
      const int idx = tid + 256 * (li < NX ? li : li - NX);
      const int r = li < NX ? idx / (KIN / 4) : idx / (NOUT / 4);
      const int c = li < NX ? 4 * (idx % (KIN / 4)) : 4 * (idx % (NOUT / 4));
      const bool ok = row0 + r < R;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      if (li < NX) *reinterpret_cast<float4*>(&xs[r * LDX + c]) = v;
      else *reinterpret_cast<float4*>(&ys[r * LDY + c]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int tok = 32 * h + s;
      float av[KT], bv[NT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) av[kt] = xs[tok * LDX + 32 * kt + lr];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[nt] = ys[tok * LDY + 32 * NT * wave + 32 * nt + lr];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kt], bv[nt], acc[kt][nt], 0, 0, 0);
      // the next tile's loads, spread over the 32 MFMA steps
#pragma unroll
      for (int li = 0; li < NX + NY; ++li)
        if (li * 32 / (NX + NY) == s) nx[li] = *reinterpret_cast<const float4*>(WF_SRC(li, row0n));
    }
    if (bias_slab) {
#pragma unroll
      for (int j = 0; j < (NOUT + 255) / 256; ++j) {
        const int col = tid + 256 * j;
        if (col < NOUT) {
          float sb = 0.f;
#pragma unroll 8
          for (int r = 0; r < 64; ++r) sb += ys[r * LDY + col];
          bsum[j] += sb;
        }
      }
    }
  }
#undef WF_SRC
  float* out = slab + (long)g * KIN * NOUT;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = 32 * NT * wave + 32 * nt + lr;
#pragma unroll
      for (int i = 0; i < 16; ++i) out[(long)(32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h) * NOUT + n] = acc[kt][nt][i];
    }
  if (bias_slab) {
#pragma unroll
    for (int j = 0; j < (NOUT + 255) / 256; ++j) {
      const int col = tid + 256 * j;
      if (col < NOUT) bias_slab[(long)g * NOUT + col] = bsum[j];
    }
  }
}

// k_wgrad_full for R % 64 == 0 (every tile full: no clamps, no zero-fill selects).  Differences that matter per tile:
//  * tile loads are `uniform base (SGPR) + per-thread 32-bit offset`: the generic form spent ~10 VALU instructions (64-bit
//    multiplies, selects) per load on addresses, and f32 MFMA time and VALU time add up;
//  * the LDS operand reads of k-step s+1 are issued before the MFMAs of step s (two register sets): the compiler's own order
//    was read -> wait -> MFMA, exposing one LDS latency per step;
//  * the bias column sums are accumulated from the registers at stash time (each thread owns fixed columns of dY), not by
//    re-reading the dY tile from LDS.
// BF3 (per-call variant bit 6, 128 x 384): the outer products on v_mfma_f32_32x32x16_bf16 with both operands split into three bf16 pieces
// (24 mantissa bits, six products, fp32 accumulate).  OPT-IN only: 3.6 -> 2.9 ms per launch, but over 65 536 rows its error against fp64 is
// 19 % LARGER than the fp32-MFMA kernel's (accumulation error, six partial products per k-block: tests/test_kernels_gpu.py::
// test_wgrad_bf16_triples), so it does not meet the bar the GRU scan and the dense layers meet.  The contraction
// runs over ROWS, so a lane's eight consecutive k are eight rows of one column: it reads them as eight scalars from the fp32 tiles (the
// same LDS reads as the fp32 path: one element per lane, row and operand) and splits / packs them in registers, once per operand and
// 16-row step; 72 bf16 MFMAs (2 304 cycles) replace the 96 fp32 ones (6 144) of those 16 rows.
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&p)[3]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 pc[3];
    split_pieces<3>(v[j], pc);
    p[0][j] = pc[0]; p[1][j] = pc[1]; p[2][j] = pc[2];
  }
}
template <int KT, int NT, int PAD = LDP, bool BF3 = false>   // PAD 0: the operand reads here are column-consecutive (no row-per-lane reads), so the
                                           // tiles need no pad; 64x256 then takes exactly 80 KB and two workgroups share a CU
__global__ __launch_bounds__(256, 1) void k_wgrad_full_x(const float* __restrict__ X, int ldx, const float* __restrict__ dY, int ldy,
                                                         int R, float* __restrict__ slab, float* __restrict__ bias_slab) {
  constexpr int KIN = 32 * KT, NOUT = 128 * NT, LDX = KIN + PAD, LDY = NOUT + PAD;
  constexpr int TX = KIN / 4, RPX = 256 / TX, NX = 64 / RPX;   // X: TX threads per row, RPX rows per pass, NX passes
  constexpr int NY = 8 * NT;                                   // dY: pass (rr, j) = rows 8 rr + (tid >> 5), float4 column 32 j + (tid & 31)
  extern __shared__ __align__(16) float wf_smem[];
  float* xs = wf_smem;              // [64][LDX]
  float* ys = xs + 64 * LDX;        // [64][LDY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int g = blockIdx.x, G = gridDim.x;
  const int ntiles = R >> 6;
  f32x16 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[kt][nt][i] = 0.f;
  float4 bsum[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned xoff = (unsigned)(tid / TX) * (unsigned)ldx + 4u * (unsigned)(tid % TX);
  const unsigned yoff = (unsigned)(tid >> 5) * (unsigned)ldy + 4u * (unsigned)(tid & 31);
  float* xsw = xs + (tid / TX) * LDX + 4 * (tid % TX);
  float* ysw = ys + (tid >> 5) * LDY + 4 * (tid & 31);
  float4 nxx[NX], nxy[NY];
#define WFX_LOADX(LI, ROW0) nxx[LI] = *reinterpret_cast<const float4*>(X + ((ROW0) + (LI) * RPX) * (long)ldx + xoff)
#define WFX_LOADY(LI, ROW0) nxy[LI] = *reinterpret_cast<const float4*>(dY + ((ROW0) + 8 * ((LI) / NT)) * (long)ldy + 128 * ((LI) % NT) + yoff)
  {
    const long row0 = (long)min(g, ntiles - 1) * 64;
#pragma unroll
    for (int li = 0; li < NX; ++li) { WFX_LOADX(li, row0); }
#pragma unroll
    for (int li = 0; li < NY; ++li) { WFX_LOADY(li, row0); }
  }
  const float* xr = xs + (32 * h) * LDX + lr;
  const float* yr = ys + (32 * h) * LDY + 32 * NT * wave + lr;
  for (int tile = g; tile < ntiles; tile += G) {
    const long row0n = (long)min(tile + G, ntiles - 1) * 64;
    __syncthreads();          // MFMAs of the previous tile have finished reading LDS
#pragma unroll
    for (int li = 0; li < NX; ++li) {   // component-wise: a whole-float4 copy out of the array leaves it in scratch
      const float4 v = nxx[li];
      *reinterpret_cast<float4*>(xsw + li * RPX * LDX) = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int li = 0; li < NY; ++li) {
      const float4 v = nxy[li];
      *reinterpret_cast<float4*>(ysw + 8 * (li / NT) * LDY + 128 * (li % NT)) = v;
      bsum[li % NT].x += v.x; bsum[li % NT].y += v.y; bsum[li % NT].z += v.z; bsum[li % NT].w += v.w;
    }
    __syncthreads();
    if constexpr (BF3) {
      // this lane's k of MFMA step s4: rows 16 s4 + 8 h + j, j = 0..7 (any bijection of the tile's 64 rows that A and B share)
      const float* xb = xs + (8 * h) * LDX + lr;
      const float* yb = ys + (8 * h) * LDY + 32 * NT * wave + lr;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        bf16x8 xp[KT][3], yp[NT][3];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = xb[(16 * s4 + j) * LDX + 32 * kt];
          split8(v, xp[kt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = yb[(16 * s4 + j) * LDY + 32 * nt];
          split8(v, yp[nt]);
        }
        // products in decreasing order of magnitude: (0,0) (0,1) (1,0) (0,2) (2,0) (1,1); consecutive MFMAs go to different accumulators
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          const int qa = pr == 0 ? 0 : (pr == 1 ? 0 : (pr == 2 ? 1 : (pr == 3 ? 0 : (pr == 4 ? 2 : 1))));
          const int qb = pr == 0 ? 0 : (pr == 1 ? 1 : (pr == 2 ? 0 : (pr == 3 ? 2 : (pr == 4 ? 0 : 1))));
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[kt][qa], yp[nt][qb], acc[kt][nt], 0, 0, 0);
        }
        // the next tile's loads, spread over the four steps
#pragma unroll
        for (int li = 0; li < NY; ++li)
          if (li * 4 / (NX + NY) == s4) { WFX_LOADY(li, row0n); }
#pragma unroll
        for (int li = 0; li < NX; ++li)
          if ((NY + li) * 4 / (NX + NY) == s4) { WFX_LOADX(li, row0n); }
      }
      continue;
    }
    float av[2][KT], bv[2][NT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) av[0][kt] = xr[32 * kt];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[0][nt] = yr[32 * nt];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      if (s + 1 < 32) {       // operands of the next k-step: their LDS latency passes under this step's MFMAs
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) av[(s + 1) & 1][kt] = xr[(s + 1) * LDX + 32 * kt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[(s + 1) & 1][nt] = yr[(s + 1) * LDY + 32 * nt];
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][kt], bv[s & 1][nt], acc[kt][nt], 0, 0, 0);
      // the next tile's loads, spread over the 32 MFMA steps
#pragma unroll
      for (int li = 0; li < NY; ++li)
        if (li * 32 / (NX + NY) == s) { WFX_LOADY(li, row0n); }
#pragma unroll
      for (int li = 0; li < NX; ++li)
        if ((NY + li) * 32 / (NX + NY) == s) { WFX_LOADX(li, row0n); }
    }
  }
#undef WFX_LOADX
#undef WFX_LOADY
  float* out = slab + (long)g * KIN * NOUT;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = 32 * NT * wave + 32 * nt + lr;
#pragma unroll
      for (int i = 0; i < 16; ++i) out[(long)(32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h) * NOUT + n] = acc[kt][nt][i];
    }
  if (bias_slab) {   // fold the 8 row groups' partial column sums through LDS (the tiles are dead now)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<float4*>(&wf_smem[(tid >> 5) * NOUT + 128 * j + 4 * (tid & 31)]) = bsum[j];
    __syncthreads();
    for (int col = tid; col < NOUT; col += 256) {
      float sb = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sb += wf_smem[q * NOUT + col];
      bias_slab[(long)g * NOUT + col] = sb;
    }
  }
}

// Whole-matrix weight gradient on a 2 x 2 wave grid, every tile full: a workgroup accumulates (64 KTW) weight rows x (64 NTW)
// columns (wave = (k half, column half): 32 KTW rows x 32 NTW columns, KTW x NTW accumulator tiles); blockIdx.y selects the column block.
//   KTW = 1: 64 x 192 (the retention K/V/G projection of the cross site) -- the split kernel reads X twice for this shape;
//   KTW = 2: 128 x 384 as two column halves: X is read twice (+25 % bytes), but the tiles take 80 KB and the accumulators
//            96 registers, so TWO workgroups share a CU and one's barrier / stash phases run under the other's MFMAs.
// Same pipeline as k_wgrad_full_x (uniform-base tile loads, operands one k-step ahead, bias sums at stash time), unpadded tiles.
template <int KTW, int NTW>
__global__ __launch_bounds__(256, 2) void k_wgrad_full_g(const float* __restrict__ X, int ldx, const float* __restrict__ dY, int ldy,
                                                         int R, int NOUT, float* __restrict__ slab, float* __restrict__ bias_slab) {
  constexpr int KIN = 64 * KTW, NB = 64 * NTW, LDX = KIN, LDY = NB, TX = KIN / 4, RPX = 256 / TX, NX = 64 / RPX, NY = 4 * NTW;
  extern __shared__ __align__(16) float wf_smem[];
  float* xs = wf_smem;              // [64][LDX]
  float* ys = xs + 64 * LDX;        // [64][LDY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int wk = wave >> 1, wn = wave & 1;
  const int g = blockIdx.x, G = gridDim.x, cb = NB * blockIdx.y;
  const int ntiles = R >> 6;
  dY += cb;
  f32x16 acc[KTW][NTW];
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[kt][nt][i] = 0.f;
  float4 bsum[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  // X: TX threads per row, RPX rows per pass; dY: pass (rr, j) = rows 16 rr + (tid >> 4), float4 column 16 j + (tid & 15)
  const unsigned xoff = (unsigned)(tid / TX) * (unsigned)ldx + 4u * (unsigned)(tid % TX);
  const unsigned yoff = (unsigned)(tid >> 4) * (unsigned)ldy + 4u * (unsigned)(tid & 15);
  float* xsw = xs + (tid / TX) * LDX + 4 * (tid % TX);
  float* ysw = ys + (tid >> 4) * LDY + 4 * (tid & 15);
  float4 nxx[NX], nxy[NY];
#define WG_LOADXG(LI, ROW0) nxx[LI] = *reinterpret_cast<const float4*>(X + ((ROW0) + RPX * (LI)) * (long)ldx + xoff)
#define WG_LOADYG(LI, ROW0) nxy[LI] = *reinterpret_cast<const float4*>(dY + ((ROW0) + 16 * ((LI) / NTW)) * (long)ldy + 64 * ((LI) % NTW) + yoff)
  {
    const long row0 = (long)min(g, ntiles - 1) * 64;
#pragma unroll
    for (int li = 0; li < NX; ++li) { WG_LOADXG(li, row0); }
#pragma unroll
    for (int li = 0; li < NY; ++li) { WG_LOADYG(li, row0); }
  }
  const float* xr = xs + (32 * h) * LDX + 32 * KTW * wk + lr;
  const float* yr = ys + (32 * h) * LDY + 32 * NTW * wn + lr;
  for (int tile = g; tile < ntiles; tile += G) {
    const long row0n = (long)min(tile + G, ntiles - 1) * 64;
    __syncthreads();          // MFMAs of the previous tile have finished reading LDS
#pragma unroll
    for (int li = 0; li < NX; ++li) {   // component-wise: a whole-float4 copy out of the array leaves it in scratch
      const float4 v = nxx[li];
      *reinterpret_cast<float4*>(xsw + RPX * li * LDX) = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int li = 0; li < NY; ++li) {
      const float4 v = nxy[li];
      *reinterpret_cast<float4*>(ysw + 16 * (li / NTW) * LDY + 64 * (li % NTW)) = v;
      bsum[li % NTW].x += v.x; bsum[li % NTW].y += v.y; bsum[li % NTW].z += v.z; bsum[li % NTW].w += v.w;
    }
    __syncthreads();
    float av[2][KTW], bv[2][NTW];
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) av[0][kt] = xr[32 * kt];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) bv[0][nt] = yr[32 * nt];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      if (s + 1 < 32) {
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) av[(s + 1) & 1][kt] = xr[(s + 1) * LDX + 32 * kt];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bv[(s + 1) & 1][nt] = yr[(s + 1) * LDY + 32 * nt];
      }
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][kt], bv[s & 1][nt], acc[kt][nt], 0, 0, 0);
      // the next tile's loads, one per MFMA step
#pragma unroll
      for (int li = 0; li < NY; ++li)
        if (li == s) { WG_LOADYG(li, row0n); }
#pragma unroll
      for (int li = 0; li < NX; ++li)
        if (NY + li == s) { WG_LOADXG(li, row0n); }
    }
  }
#undef WG_LOADXG
#undef WG_LOADYG
  float* out = slab + (long)g * KIN * NOUT + cb;
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = 32 * NTW * wn + 32 * nt + lr;
#pragma unroll
      for (int i = 0; i < 16; ++i) out[(long)(32 * KTW * wk + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h) * NOUT + n] = acc[kt][nt][i];
    }
  if (bias_slab) {   // fold the 16 row groups' partial column sums through LDS (the tiles are dead now)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NTW; ++j) *reinterpret_cast<float4*>(&wf_smem[(tid >> 4) * NB + 64 * j + 4 * (tid & 15)]) = bsum[j];
    __syncthreads();
    for (int col = tid; col < NB; col += 256) {
      float sb = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sb += wf_smem[q * NB + col];
      bias_slab[(long)g * NOUT + cb + col] = sb;
    }
  }
}

}  // namespace magpo

using namespace magpo;

extern "C" int magpo_linear(const float* X, int ldx, const float* Wt, const float* bias, float* Y, int ldy, float* Ypre,
                            long R, int KIN, int NOUT, int act, int variant, hipStream_t stream) {
  if (R <= 0) return MAGPO_OK;
  if ((ldx & 3) || KIN % 64 || NOUT <= 0) { set_error("magpo_linear: KIN must be a multiple of 64, ldx of 4"); return MAGPO_EINVAL; }
  // variant (A/B reference paths, same results up to fp32 summation order; 0 = the fast path): bit 0 = wave-autonomous k_linear_wk instead of
  // the shared-tile k_linear_lds, bit 1 = the same for KIN = 64 only
  // bit 2 = bf16 triples (k_linear_lds<.., BF3>: KIN 128 / 192 on the shared-tile path; ignored elsewhere)
  if (variant < 0 || variant > 7) { set_error("magpo_linear: variant must be in [0, 7]"); return MAGPO_EINVAL; }
  const bool lds64 = !(variant & 2);
  const float* aux = nullptr;
  if (act == ACT_MASKPOS) {   // the Ypre argument carries the mask INPUT (same shape / stride as Y), nothing else is written
    if (!Ypre) { set_error("magpo_linear: act 4 (mask) needs the mask tensor in the Ypre argument"); return MAGPO_EINVAL; }
    aux = Ypre;
    Ypre = nullptr;
  }
  if ((KIN == 128 || KIN == 192 || KIN == 256 || KIN == 384 || (KIN == 64 && lds64)) && !Ypre) {
    // persistent waves: one wave per (walker, 32-column group); blocks of 4 / 2 / 1 waves so that every wave slot of
    // a CU can be filled (a 3-wave block leaves a quarter of the slots idle), about one resident wave set in total
    const int ncg = (NOUT + 31) / 32;
    const int wpb = (ncg % 4 == 0) ? 4 : ((ncg % 2 == 0) ? 2 : 1);
    const long ntiles = (R + 31) / 32;
    long walkers = 2048 / wpb;
    if (walkers > ntiles) walkers = ntiles;
    dim3 grid((unsigned)walkers, (unsigned)(ncg / wpb)), block(64 * wpb);
    const bool use_lds = !(variant & 1);
    if (use_lds) {
      // shared-tile form: 4 waves per block, 2 when the column groups do not fill blocks of 4 (every wave then has MFMA work);
      // column groups past NOUT idle in the MFMA part but help loading
      const int nw = (ncg % 4 == 0) ? 4 : 2;
      const int gy = (ncg + nw - 1) / nw;
      long wk2 = 2048 / nw;   // about 2 waves per SIMD; LDS 2 x 32 x (KIN + 4) floats per block
      // several column blocks per row walker: all of them co-resident (walkers x column blocks <= resident workgroups), so the
      // column blocks of a walker -- same XCD, since the walker count is a multiple of 8 -- read a tile at about the same
      // time and the re-reads hit that XCD's L2 instead of HBM (in a second round they would come from HBM again)
      if (gy >= 3) wk2 /= 2;
      if (wk2 > ntiles) wk2 = ntiles;
      dim3 g2((unsigned)wk2, (unsigned)gy), b2(64 * nw);
      if ((variant & 4) && (KIN == 128 || KIN == 192) && nw == 4) {   // (two-wave blocks -- narrow outputs -- are slower on it: 128 -> 20 0.86 vs 1.35 ms)
        const size_t ldb = (size_t)2 * 3 * 32 * (KIN + 8) * sizeof(__bf16);
#define LAUNCH_BF3(K_)                                                                                                  \
        {                                                                                                               \
          static bool attr = false;                                                                                     \
          if (!attr && ldb > 65536) {                                                                                   \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_lds<K_, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldb); \
            attr = true;                                                                                                \
          }                                                                                                             \
          hipLaunchKernelGGL((k_linear_lds<K_, 4, true>), g2, b2, ldb, stream, X, ldx, Wt, bias, Y, ldy, (int)R, NOUT, act, aux); \
        }
        if (KIN == 128) LAUNCH_BF3(128) else LAUNCH_BF3(192)
#undef LAUNCH_BF3
        return check_launch("magpo_linear");
      }
      const size_t lds = (size_t)2 * 32 * (KIN + LDP) * sizeof(float);
#define LAUNCH_LDS(K_)                                                                                                   \
      {                                                                                                                 \
        static bool attr = false;                                                                                       \
        if (!attr && lds > 65536) {                                                                                     \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_lds<K_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_lds<K_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
          attr = true;                                                                                                  \
        }                                                                                                               \
        if (nw == 4) hipLaunchKernelGGL((k_linear_lds<K_, 4>), g2, b2, lds, stream, X, ldx, Wt, bias, Y, ldy, (int)R, NOUT, act, aux); \
        else hipLaunchKernelGGL((k_linear_lds<K_, 2>), g2, b2, lds, stream, X, ldx, Wt, bias, Y, ldy, (int)R, NOUT, act, aux); \
      }
      if (KIN == 64) LAUNCH_LDS(64) else if (KIN == 128) LAUNCH_LDS(128) else if (KIN == 192) LAUNCH_LDS(192) else if (KIN == 256) LAUNCH_LDS(256) else LAUNCH_LDS(384)
#undef LAUNCH_LDS
      return check_launch("magpo_linear");
    }
    if (KIN == 128) hipLaunchKernelGGL((k_linear_wk<128>), grid, block, 0, stream, X, ldx, Wt, bias, Y, ldy, (int)R, NOUT, act);
    else if (KIN == 192) hipLaunchKernelGGL((k_linear_wk<192>), grid, block, 0, stream, X, ldx, Wt, bias, Y, ldy, (int)R, NOUT, act);
    else if (KIN == 256) hipLaunchKernelGGL((k_linear_wk<256>), grid, block, 0, stream, X, ldx, Wt, bias, Y, ldy, (int)R, NOUT, act);
    else hipLaunchKernelGGL((k_linear_wk<384>), grid, block, 0, stream, X, ldx, Wt, bias, Y, ldy, (int)R, NOUT, act);
    return check_launch("magpo_linear");
  }
  if (KIN == 64 || KIN == 128) {
    // wave-autonomous path: 64 columns per wave, up to 4 waves (256 columns) per workgroup
    const int cpw = KIN == 64 ? 64 : 32;  // columns per wave (128 per wave was measured slower: 256 VGPRs + spills)
    const int ncg = (NOUT + cpw - 1) / cpw;
    const int wpb = (ncg % 4 == 0) ? 4 : ((ncg % 2 == 0) ? 2 : 1);
    const long ntiles = (R + 31) / 32;
    long walkers = 2048 / ncg;  // about one resident wave set (2 waves per SIMD on 256 CUs, register-limited)
    if (walkers < 1) walkers = 1;
    if (walkers > ntiles) walkers = ntiles;
    dim3 grid((unsigned)walkers, (unsigned)(ncg / wpb)), block(64 * wpb);
    if (KIN == 64) hipLaunchKernelGGL((k_linear_w<64, 2>), grid, block, 0, stream, X, ldx, Wt, bias, Y, ldy, Ypre, (int)R, NOUT, act);
    else hipLaunchKernelGGL((k_linear_w<128, 1>), grid, block, 0, stream, X, ldx, Wt, bias, Y, ldy, Ypre, (int)R, NOUT, act);
    return check_launch("magpo_linear");
  }
  dim3 grid((unsigned)((R + 63) / 64)), block(256);
  size_t lds = (size_t)64 * (KIN + LDP) * sizeof(float);
#define LAUNCH(K_)                                                                                              \
  if (lds > 65536) {                                                                                            \
    static bool attr_set = false;                                                                               \
    if (!attr_set) {                                                                                            \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear<K_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_set = true;                                                                                          \
    }                                                                                                           \
  }                                                                                                             \
  hipLaunchKernelGGL(k_linear<K_>, grid, block, lds, stream, X, ldx, Wt, bias, Y, ldy, Ypre, (int)R, NOUT, act)
  switch (KIN) {
    case 64: LAUNCH(64); break;
    case 128: LAUNCH(128); break;
    case 192: LAUNCH(192); break;
    case 256: LAUNCH(256); break;
    case 384: LAUNCH(384); break;
    default: set_error("magpo_linear: unsupported KIN"); return MAGPO_EINVAL;
  }
#undef LAUNCH
  return check_launch("magpo_linear");
}

// Prologue-fused 64 -> NOUT dense layer (see k_linear_pro).  pro: 1 embed-action, 2 embed-observation, 3 residual+norm, 4 gelu+norm.
extern "C" int magpo_linear_pro(int pro, const float* a, long lda, const float* y, long ldy_in, const float* s1, const float* s2,
                                const float* pe, const int* pos, long pos_stride, int npos, int use_pe, const float* W,
                                const int* idx, long idx_stride, const float* s_obs, int F, float* out, long ldout,
                                float* outpe, long ldoutpe, const float* Wt, const float* bias, float* Y, long ldy, long R,
                                int NOUT, hipStream_t stream) {
  if (R <= 0) return MAGPO_OK;
  if (pro < 1 || pro > 4 || NOUT <= 0) { set_error("magpo_linear_pro: bad prologue id / NOUT"); return MAGPO_EINVAL; }
  ProArgs p{a, lda, y, ldy_in, s1, s2, pe, pos, pos_stride, npos, use_pe, W, idx, idx_stride, s_obs, F, out, ldout, outpe, ldoutpe};
  const int ncg = (NOUT + 63) / 64;
  const int wpb = ncg < 4 ? ncg : 4;
  const long ntiles = (R + 31) / 32;
  long walkers = 2048 / wpb;
  if (walkers > ntiles) walkers = ntiles;
  dim3 grid((unsigned)walkers, (unsigned)((ncg + 3) / 4)), block(64 * wpb);
#define LP(P_) hipLaunchKernelGGL((k_linear_pro<P_, 2>), grid, block, 0, stream, p, Wt, bias, Y, ldy, (int)R, NOUT)
  switch (pro) {
    case PRO_EMBED_ACT: LP(PRO_EMBED_ACT); break;
    case PRO_EMBED_OBS: LP(PRO_EMBED_OBS); break;
    case PRO_RESNORM: LP(PRO_RESNORM); break;
    default: LP(PRO_HEADMID); break;
  }
#undef LP
  return check_launch("magpo_linear_pro");
}

extern "C" long magpo_wgrad_workspace_floats(int KIN, int NOUT, int G) { return (long)G * ((long)KIN * NOUT + NOUT); }

// dW[KIN][NOUT] (+ optional db[NOUT]) = scale * X^T dY ; workspace >= magpo_wgrad_workspace_floats floats.
// Only the first `krows` rows of dW are written (krows < KIN for zero-padded small operands).
extern "C" int magpo_wgrad(const float* X, int ldx, const float* dY, int ldy, long R, int KIN, int krows, int NOUT, float* dW,
                           float* db, float* workspace, int G, float scale, int accumulate, int variant, hipStream_t stream) {
  if ((ldx & 3) || (ldy & 3) || KIN % 64) { set_error("magpo_wgrad: bad strides / KIN"); return MAGPO_EINVAL; }
  // variant (A/B reference paths, same results up to fp32 summation order; 0 = the fast path), bit mask: 1 = split kernel k_wgrad for every
  // shape, 2 = generic k_wgrad_full also on full tiles, 4 = no unpadded-tile 64 x 256 kernel, 8 = 128 x 384 as two column halves on the
  // wave-grid kernel, 16 = 64 x 64 on the wave-grid kernel, 32 = 64 x 256 on the wave-grid kernel
  if (variant < 0 || variant > 127) { set_error("magpo_wgrad: variant must be in [0, 127]"); return MAGPO_EINVAL; }
  float* slab = workspace;
  float* bslab = db ? workspace + (long)G * KIN * NOUT : nullptr;
  const bool use_full = !(variant & 1);
  const bool use_x = !(variant & 2);
  // whole-matrix form: one workgroup per slab accumulates the full KIN x NOUT block, every row read once.  128x384 always
  // (k_wgrad_full, or k_wgrad_full_x when every tile is full); with full tiles also 128x128 (two workgroups per CU: 1.53 ->
  // 1.30 ms) and 64x256 (1.84 -> 1.76 ms).  Other shapes stay on the split kernel k_wgrad.
  const bool exact = R % 64 == 0 && use_x;
  const bool use_pad0 = !(variant & 4);
  const int g_alt = (variant >> 4) & 3;   // experiments (bit mask): 1 = 64x64 on the wave-grid kernel (0.43 vs 0.40 ms), 2 = 64x256 (1.63 vs 1.64 ms)
  if (use_full && exact && use_pad0 && !(g_alt & 2) && KIN == 64 && NOUT == 256 && R >= 64 * 256) {
    if (G > 512) G = 512;
    float* bsl = db ? workspace + (long)G * KIN * NOUT : nullptr;
    const size_t lds = (size_t)64 * (KIN + NOUT) * sizeof(float);
    static bool attrp = false;
    if (!attrp) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_full_x<2, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attrp = true; }
    hipLaunchKernelGGL((k_wgrad_full_x<2, 2, 0>), dim3(G), dim3(256), lds, stream, X, ldx, dY, ldy, (int)R, slab, bsl);
    long P = (long)krows * NOUT;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((P + 63) / 64)), dim3(1024), 0, stream, slab, dW, G, P, (long)KIN * NOUT, scale, accumulate);
    if (db) hipLaunchKernelGGL(k_reduce_slabs, dim3((NOUT + 63) / 64), dim3(1024), 0, stream, bsl, db, G, (long)NOUT, (long)NOUT, scale, accumulate);
    return check_launch("magpo_wgrad");
  }
  // 128 x 384 on this kernel (two column halves, two workgroups per CU) measured 3.57 vs 3.63 ms for k_wgrad_full_x<4,3> while
  // reading X twice: within noise, so the one-pass kernel stays the default (variant bit 3 selects this one)
  const int use_g2 = (variant >> 3) & 1;
  int gk = 0, gn = 0;   // (KTW, NTW) of the 2 x 2 wave-grid kernel, 0 = not this kernel
  if (KIN == 64 && NOUT == 192) { gk = 1; gn = 3; }
  else if (use_g2 && KIN == 128 && NOUT == 384) { gk = 2; gn = 3; }
  else if ((g_alt & 1) && KIN == 64 && NOUT == 64) { gk = 1; gn = 1; }
  else if ((g_alt & 2) && KIN == 64 && NOUT == 256) { gk = 1; gn = 4; }
  else if (KIN == 128 && NOUT == 128) { gk = 2; gn = 2; }        // 1.23 vs 1.32 ms for k_wgrad_full_x<4,1>
  if (use_full && exact && R >= 64 * 256 && gk) {
    const int nbk = 64 * gn, gy = NOUT / nbk;
    const size_t lds = (size_t)64 * (KIN + nbk) * sizeof(float);
    const int per_cu = lds <= 32 * 1024 ? 4 : (lds <= 40 * 1024 ? 3 : 2);   // resident workgroups per CU (LDS; launch bound 2 waves / SIMD for the wide ones)
    const int gcap = 256 * per_cu / gy;
    if (G > gcap) G = gcap;
    float* bsl = db ? workspace + (long)G * KIN * NOUT : nullptr;
#define LAUNCH_G(K_, N_)                                                                                                      \
    {                                                                                                                          \
      static bool attr = false;                                                                                                \
      if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_full_g<K_, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (64 * K_ + 64 * N_) * 4); attr = true; } \
      hipLaunchKernelGGL((k_wgrad_full_g<K_, N_>), dim3(G, gy), dim3(256), lds, stream, X, ldx, dY, ldy, (int)R, NOUT, slab, bsl); \
    }
    if (gk == 1 && gn == 3) LAUNCH_G(1, 3) else if (gk == 2 && gn == 3) LAUNCH_G(2, 3) else if (gk == 1 && gn == 1) LAUNCH_G(1, 1)
    else if (gk == 1 && gn == 4) LAUNCH_G(1, 4) else LAUNCH_G(2, 2)
#undef LAUNCH_G
    long P = (long)krows * NOUT;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((P + 63) / 64)), dim3(1024), 0, stream, slab, dW, G, P, (long)KIN * NOUT, scale, accumulate);
    if (db) hipLaunchKernelGGL(k_reduce_slabs, dim3((NOUT + 63) / 64), dim3(1024), 0, stream, bsl, db, G, (long)NOUT, (long)NOUT, scale, accumulate);
    return check_launch("magpo_wgrad");
  }
  const bool full_shape = (KIN == 128 && NOUT == 384) || (exact && ((KIN == 128 && NOUT == 128) || (KIN == 64 && NOUT == 256)));
  if (use_full && full_shape && R >= 64 * 256) {
    const int gcap = (KIN == 128 && NOUT == 128) ? 512 : 256;
    if (G > gcap) G = gcap;
    float* bsl = db ? workspace + (long)G * KIN * NOUT : nullptr;
    const size_t lds = (size_t)64 * ((KIN + LDP) + (NOUT + LDP)) * sizeof(float);
#define LAUNCH_FULL(KT_, NT_)                                                                                                 \
    {                                                                                                                          \
      static bool attr = false;                                                                                                \
      if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_full<KT_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; } \
      static bool attrx = false;                                                                                               \
      if (!attrx) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_full_x<KT_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attrx = true; } \
      if (R % 64 == 0 && use_x) hipLaunchKernelGGL((k_wgrad_full_x<KT_, NT_>), dim3(G), dim3(256), lds, stream, X, ldx, dY, ldy, (int)R, slab, bsl); \
      else hipLaunchKernelGGL((k_wgrad_full<KT_, NT_>), dim3(G), dim3(256), lds, stream, X, ldx, dY, ldy, (int)R, slab, bsl);  \
    }
    if ((variant & 64) && KIN == 128 && NOUT == 384 && R % 64 == 0 && use_x) {   // bf16 triples (k_wgrad_full_x<4, 3, LDP, true>)
      static bool attrb = false;
      if (!attrb) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_full_x<4, 3, LDP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attrb = true; }
      hipLaunchKernelGGL((k_wgrad_full_x<4, 3, LDP, true>), dim3(G), dim3(256), lds, stream, X, ldx, dY, ldy, (int)R, slab, bsl);
    } else
    if (KIN == 64 && NOUT == 128) LAUNCH_FULL(2, 1) else if (KIN == 64 && NOUT == 256) LAUNCH_FULL(2, 2) else if (KIN == 64) LAUNCH_FULL(2, 3)
    else if (NOUT == 128) LAUNCH_FULL(4, 1) else if (NOUT == 256) LAUNCH_FULL(4, 2) else LAUNCH_FULL(4, 3)
#undef LAUNCH_FULL
    long P = (long)krows * NOUT;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((P + 63) / 64)), dim3(1024), 0, stream, slab, dW, G, P, (long)KIN * NOUT, scale, accumulate);
    if (db) hipLaunchKernelGGL(k_reduce_slabs, dim3((NOUT + 63) / 64), dim3(1024), 0, stream, bsl, db, G, (long)NOUT, (long)NOUT, scale, accumulate);
    return check_launch("magpo_wgrad");
  }
  int nb = NOUT >= 128 ? 2 : 1;
  // fill the chip in whole waves of workgroups: 3 (NB=2) / 4 (NB=1) resident workgroups per CU x 256 CUs
  {
    const int per_g = ((NOUT + 64 * nb - 1) / (64 * nb)) * (KIN / 64);
    const int resident = 256 * (nb == 2 ? 3 : 4);
    int g_fit = resident / per_g;
    if (g_fit < 1) g_fit = 1;
    if (G > g_fit) G = g_fit;
  }
  dim3 grid(G, (NOUT + 64 * nb - 1) / (64 * nb), KIN / 64), block(256);
  if (nb == 2) hipLaunchKernelGGL(k_wgrad<2>, grid, block, 0, stream, X, ldx, dY, ldy, (int)R, KIN, NOUT, slab, bslab);
  else hipLaunchKernelGGL(k_wgrad<1>, grid, block, 0, stream, X, ldx, dY, ldy, (int)R, KIN, NOUT, slab, bslab);
  long P = (long)krows * NOUT;
  hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((P + 63) / 64)), dim3(1024), 0, stream, slab, dW, G, P, (long)KIN * NOUT, scale, accumulate);
  if (db) hipLaunchKernelGGL(k_reduce_slabs, dim3((NOUT + 63) / 64), dim3(1024), 0, stream, bslab, db, G, (long)NOUT, (long)NOUT, scale, accumulate);
  return check_launch("magpo_wgrad");
}

extern "C" int magpo_reduce_slabs(const float* slab, float* out, int G, long P, long stride, float scale, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((P + 63) / 64)), dim3(1024), 0, stream, slab, out, G, P, stride, scale, accumulate);
  return check_launch("magpo_reduce_slabs");
}

extern "C" int magpo_transpose_pad(const float* W, float* Wt, int K, int N, int Npad, hipStream_t stream) {
  int total = Npad * K;
  hipLaunchKernelGGL(k_transpose_pad, dim3((total + 255) / 256), dim3(256), 0, stream, W, Wt, K, N, Npad);
  return check_launch("magpo_transpose_pad");
}

#ifdef MAGPO_WG_PROF
extern "C" int magpo_debug_wg_prof(unsigned long long* out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(magpo::g_wg_prof), sizeof(unsigned long long) * 8) != hipSuccess) return MAGPO_ELAUNCH;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(magpo::g_wg_prof), z, sizeof(z)) != hipSuccess) return MAGPO_ELAUNCH; }
  return MAGPO_OK;
}
#endif
