// Row-wise (token-local) pieces of the Sable guider between the dense layers and the retention
// kernels: embeddings, RMSNorm / GroupNorm, GELU / swish gates, residual adds, positional encoding,
// forward and hand-derived backward.  HBM-bound: 16 lanes x float4 per 64-wide row (4 rows per
// wave-instruction, 1 KiB coalesced), row statistics by 16-lane xor-shuffles, parameter gradients
// accumulated per lane and written as per-workgroup slabs (reduced in fixed order afterwards).
//
// Reference maths: sable_network.py:62-71,93-109,121-137,188-217,255-284,296-319 (blocks, heads),
// retention.py:289-295 (GroupNorm + swish gate), positional_encoding.py:24-60, torsos.py:79-99
// (SwiGLU is identically zero at its zero init and is not evaluated; the host verifies that).
#include "common.hpp"

namespace magpo {

constexpr float NORM_EPS = 1e-6f;
constexpr int ROWS_PER_BLOCK = 16;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4sum(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Row geometry: a W-wide row (W = embed_dim = 64 or 128) is held as float4 by LPR = W / 4 consecutive lanes, RPW = 64 / LPR rows per
// wave, RPB = 4 RPW rows per 256-thread block.  Grids come from row_grid (sized for 16 rows per block; the kernels are grid-stride).
template <int W> struct RowGeo { static constexpr int LPR = W / 4, RPW = 64 / LPR, RPB = 4 * RPW; };

// mean over aligned groups of gs channels (gs = 4 ... W, a power of two) of a row
__device__ __forceinline__ float groupmean(float4 v, int gs) {
  float s = f4sum(v);
  for (int o = gs >> 3; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s / (float)gs;
}
// all-reduce over the lanes of one row
template <int W> __device__ __forceinline__ float rowsum(float v) {
  v = sum16(v);
  if (W == 128) v += __shfl_xor(v, 16, 64);
  return v;
}
template <int W> __device__ __forceinline__ float rowmean(float4 v) { return rowsum<W>(f4sum(v)) * (1.0f / (float)W); }

struct RmsFwd { float4 y; float4 xhat; float rstd; };
template <int W> __device__ __forceinline__ RmsFwd rms_fwd(float4 x, float4 s) {
  RmsFwd o;
  o.rstd = rsqrtf(rowmean<W>(f4mul(x, x)) + NORM_EPS);
  o.xhat = f4scale(x, o.rstd);
  o.y = f4mul(o.xhat, s);
  return o;
}
// returns dx; accumulates ds
template <int W> __device__ __forceinline__ float4 rms_bwd(float4 dy, const RmsFwd& f, float4 s, float4& ds) {
  float4 g = f4mul(dy, s);
  float dot = rowmean<W>(f4mul(g, f.xhat));
  ds = f4add(ds, f4mul(dy, f.xhat));
  return f4scale(make_float4(g.x - f.xhat.x * dot, g.y - f.xhat.y * dot, g.z - f.xhat.z * dot, g.w - f.xhat.w * dot), f.rstd);
}

// Sum a per-lane float4 accumulator over all rows handled by the workgroup and store W floats.
template <int W> __device__ __forceinline__ void block_store_colsum(float4 acc, float* __restrict__ outw, float* lds /*[4][W]*/) {
  if (W == 64) {
    acc.x += __shfl_xor(acc.x, 16, 64); acc.y += __shfl_xor(acc.y, 16, 64);
    acc.z += __shfl_xor(acc.z, 16, 64); acc.w += __shfl_xor(acc.w, 16, 64);
  }
  acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
  acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane < W / 4) st4(&lds[wave * W + 4 * lane], acc);
  __syncthreads();
  if (threadIdx.x < W) outw[threadIdx.x] = (lds[threadIdx.x] + lds[W + threadIdx.x]) + (lds[2 * W + threadIdx.x] + lds[3 * W + threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// pe_table[pos][2i] = sin(pos * div_i), [2i+1] = cos(pos * div_i), div_i = exp(2i * -ln(1e4)/E)
__global__ void k_pe_table(float* __restrict__ pe, int npos, int E) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npos * E) return;
  int pos = i / E, c = i - pos * E;
  // fp32 arithmetic as the reference, each transcendental correctly rounded (evaluated in fp64)
  const float a = (float)(c & ~1) * (-(float)log(10000.0) / (float)E);
  const float div = (float)exp((double)a);
  const float x = (float)pos * div;
  pe[i] = (c & 1) ? (float)cos((double)x) : (float)sin((double)x);
}

// ------------------------------------------------------------------------------------------------
// Embedding rows: z -> x0 = gelu(z) -> xn = rmsnorm(x0) * s_ln -> kin = xn + pe[pos]
//   mode 0 (observation encoder): z = (rmsnorm_F(obs) * s_obs) @ W_obs          (sable_network.py:93-101,126,132)
//   mode 1 (action encoder):      z = W_act[idx]  (one-hot input, no bias)      (sable_network.py:258-267,306-307)
struct EmbedArgs {
  const float* obs; int ldo; int F; const float* s_obs; const float* W;  // W: [F][64] or [K+1][64]
  const int* idx; int idx_stride;
  const float* s_ln; const float* pe; const int* pos; int pos_stride; int npos;
  float* z; float* xn; float* kin; int ldz, ldxn, ldkin;
  long R;
};

// Small inputs of one embedding row, fetched one grid-stride iteration ahead of their use.
struct EmbedIn { float of[8]; int p; int idx; };

__device__ __forceinline__ EmbedIn embed_fetch(const EmbedArgs& a, int mode, long row) {
  EmbedIn in;
  row = row < a.R ? row : a.R - 1;
  in.p = a.pos[row * a.pos_stride];
  in.idx = 0;
#pragma unroll
  for (int f = 0; f < 8; ++f) in.of[f] = 0.f;
  if (mode == 0) {
    if (a.F <= 8) {
      const float* o = a.obs + row * a.ldo;
#pragma unroll
      for (int f = 0; f < 8; ++f) in.of[f] = o[f < a.F ? f : 0];
    }
  } else {
    in.idx = a.idx[row * a.idx_stride];
  }
  return in;
}

template <int W> __global__ __launch_bounds__(256) void k_embed_fwd(EmbedArgs a, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 sln = ld4(a.s_ln + c4);
  const long stride = (long)gridDim.x * RowGeo<W>::RPB;
  const long lrow = wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
  long base = (long)blockIdx.x * RowGeo<W>::RPB;
  if (base >= a.R) return;
  EmbedIn nx = embed_fetch(a, mode, base + lrow);
  for (; base < a.R; base += stride) {
    const long row = base + lrow;
    const bool ok = row < a.R;
    const EmbedIn in = nx;
    if (base + stride < a.R) nx = embed_fetch(a, mode, base + stride + lrow);
    float4 z = f4zero();
    if (mode == 0 && a.F <= 8) {   // small observation: every load of the row issued together (no dependent run-time loop)
      float ms = 0.f;
#pragma unroll
      for (int f = 0; f < 8; ++f) ms += f < a.F ? in.of[f] * in.of[f] : 0.f;
      const float rstd = rsqrtf(ms / (float)a.F + NORM_EPS);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        if (f < a.F) {
          const float on = in.of[f] * rstd * a.s_obs[f];
          const float4 w = ld4(a.W + f * W + c4);
          z.x += on * w.x; z.y += on * w.y; z.z += on * w.z; z.w += on * w.w;
        }
      }
    } else if (mode == 0) {
      if (ok) {
        const float* o = a.obs + row * a.ldo;
        float ms = 0.f;
        for (int f = 0; f < a.F; ++f) ms += o[f] * o[f];
        const float rstd = rsqrtf(ms / (float)a.F + NORM_EPS);
        for (int f = 0; f < a.F; ++f) {
          const float of = o[f] * rstd * a.s_obs[f];
          const float4 w = ld4(a.W + f * W + c4);
          z.x += of * w.x; z.y += of * w.y; z.z += of * w.z; z.w += of * w.w;
        }
      }
    } else {
      z = ld4(a.W + (long)in.idx * W + c4);
    }
    float4 x0 = make_float4(gelu_tanh(z.x), gelu_tanh(z.y), gelu_tanh(z.z), gelu_tanh(z.w));
    RmsFwd n = rms_fwd<W>(x0, sln);
    int p = in.p;
    p = p < 0 ? 0 : (p >= a.npos ? a.npos - 1 : p);
    const float4 pe = ld4(a.pe + (long)p * W + c4);
    if (ok) {
      if (a.z) st4(a.z + row * a.ldz + c4, z);
      st4(a.xn + row * a.ldxn + c4, n.y);
      st4(a.kin + row * a.ldkin + c4, f4add(n.y, pe));
    }
  }
}

// F <= 8 (mode 0) / action rows (mode 1): 64 rows per block iteration.  The per-row inputs (observation, position,
// action index) arrive as one coalesced tile through double-buffered LDS, fetched one iteration ahead, and the small
// weight stays in registers, so an output row costs one table load and two stores instead of ~19 memory requests.
constexpr int EMB_ROWS = 64;
template <int MODE>
__global__ __launch_bounds__(256) void k_embed_fwd_tile(EmbedArgs a) {
  __shared__ float xs[2][EMB_ROWS * 8];
  __shared__ int ps[2][EMB_ROWS];
  __shared__ int is[2][EMB_ROWS];
  const int t = threadIdx.x, c4 = 4 * (t & 15), slot = t >> 4;
  const float4 sln = ld4(a.s_ln + c4);
  float4 w[8];
  float so[8];
  if (MODE == 0) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int fc = f < a.F ? f : 0;
      w[f] = ld4(a.W + fc * 64 + c4);
      so[f] = a.s_obs[fc];
    }
  }
  const long stride = (long)gridDim.x * EMB_ROWS;
  long base = (long)blockIdx.x * EMB_ROWS;
  if (base >= a.R) return;
  float x0 = 0.f, x1 = 0.f;
  int pn = 0;
  auto fetch = [&](long b) {
    if (MODE == 0) {
      long r0 = b + (t >> 3), r1 = r0 + 32;
      r0 = r0 < a.R ? r0 : a.R - 1;
      r1 = r1 < a.R ? r1 : a.R - 1;
      const int f = (t & 7) < a.F ? (t & 7) : 0;
      x0 = a.obs[r0 * a.ldo + f];
      x1 = a.obs[r1 * a.ldo + f];
    }
    if (t < 2 * EMB_ROWS) {
      long r = b + (t & (EMB_ROWS - 1));
      r = r < a.R ? r : a.R - 1;
      if (t < EMB_ROWS) pn = a.pos[r * a.pos_stride];
      else if (MODE == 1) pn = a.idx[r * a.idx_stride];
    }
  };
  fetch(base);
  for (int it = 0; base < a.R; base += stride, it ^= 1) {
    if (MODE == 0) { xs[it][t] = x0; xs[it][t + 256] = x1; }
    if (t < EMB_ROWS) ps[it][t] = pn;
    else if (MODE == 1 && t < 2 * EMB_ROWS) is[it][t - EMB_ROWS] = pn;
    __syncthreads();
    if (base + stride < a.R) fetch(base + stride);
#pragma unroll
    for (int j = 0; j < EMB_ROWS / 16; ++j) {
      const int lrow = slot + 16 * j;
      const long row = base + lrow;
      float4 z = f4zero();
      if (MODE == 0) {
        const float4 xa = *reinterpret_cast<const float4*>(&xs[it][lrow * 8]);
        const float4 xb = *reinterpret_cast<const float4*>(&xs[it][lrow * 8 + 4]);
        const float of[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        float ms = 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f) ms += f < a.F ? of[f] * of[f] : 0.f;
        const float rstd = rsqrtf(ms / (float)a.F + NORM_EPS);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          if (f < a.F) {
            const float on = of[f] * rstd * so[f];
            z.x += on * w[f].x; z.y += on * w[f].y; z.z += on * w[f].z; z.w += on * w[f].w;
          }
        }
      } else {
        z = ld4(a.W + (long)is[it][lrow] * 64 + c4);
      }
      const float4 g0 = make_float4(gelu_tanh(z.x), gelu_tanh(z.y), gelu_tanh(z.z), gelu_tanh(z.w));
      const RmsFwd n = rms_fwd<64>(g0, sln);
      int p = ps[it][lrow];
      p = p < 0 ? 0 : (p >= a.npos ? a.npos - 1 : p);
      const float4 pe = ld4(a.pe + (long)p * 64 + c4);
      if (row < a.R) {
        if (a.z) st4(a.z + row * a.ldz + c4, z);
        st4(a.xn + row * a.ldxn + c4, n.y);
        st4(a.kin + row * a.ldkin + c4, f4add(n.y, pe));
      }
    }
  }
}

// Backward of the embedding rows: d(xn) = d0 + d1 (+ d2) -> dz; slabs: ds_ln[64], the small weight gradient
// dW[rows <= 32][64] (mode 0: rows = obs features, dW_obs = o^T dz; mode 1: rows = action slots, dW_act[idx] += dz)
// and, for mode 0, ds_obs[F].  The weight gradient is accumulated per lane (compile-time indexed registers).
struct EmbedBwdArgs {
  const float* z; int ldz;
  const float* d0; const float* d1; const float* d2; int ldd0, ldd1, ldd2;
  const float* s_ln;
  float* dz; int lddz;
  float* slab_sln;   // [grid][64]
  float* slab_w;     // [grid][32][64]
  int nrows;         // rows of the small weight (F or K+1), <= 32
  // mode 0 extras
  const float* obs; int ldo; int F; const float* s_obs; const float* W; float* slab_sobs;  // [grid][32]
  // mode 1 extras
  const int* idx; int idx_stride;
  long R;
};

template <int MODE, int NR, int W>  // NR = compile-time bound on the rows of the small weight (register accumulators)
__global__ __launch_bounds__(256) void k_embed_bwd(EmbedBwdArgs a) {
  __shared__ float lds[4 * W];
  __shared__ float sobs_acc[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 sln = ld4(a.s_ln + c4);
  float4 dsln = f4zero();
  float4 wacc[NR];
#pragma unroll
  for (int f = 0; f < NR; ++f) wacc[f] = f4zero();
  float dsobs[MODE == 0 ? NR : 1];
  // MODE 0: the small encoder's parameters live in registers (a load inside a run-time loop over F is a dependent L1 round trip per
  // feature: the kernel spent ~7 us per row group in such loops)
  float sob[MODE == 0 ? NR : 1];
  float4 wob[MODE == 0 ? NR : 1];
  if (MODE == 0) {
#pragma unroll
    for (int f = 0; f < NR; ++f) {
      dsobs[f] = 0.f;
      sob[f] = f < a.F ? a.s_obs[f] : 0.f;
      wob[f] = f < a.F ? ld4(a.W + f * W + c4) : f4zero();
    }
  }
  for (long base = (long)blockIdx.x * RowGeo<W>::RPB; base < a.R; base += (long)gridDim.x * RowGeo<W>::RPB) {
    const long row = base + wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
    const bool ok = row < a.R;
    float4 z = f4zero(), d = f4zero();
    float of[MODE == 0 ? NR : 1];   // MODE 0: normalised observation features of this row (all loads issued together)
    if (MODE == 0) {
      const float* o = a.obs + (ok ? row : 0) * a.ldo;
      float ms = 0.f;
#pragma unroll
      for (int f = 0; f < NR; ++f) { of[f] = f < a.F ? o[f] : 0.f; ms += of[f] * of[f]; }
      const float rstd = rsqrtf(ms / (float)a.F + NORM_EPS);
#pragma unroll
      for (int f = 0; f < NR; ++f) of[f] *= rstd;
    }
    if (ok) {
      if (a.z) z = ld4(a.z + row * a.ldz + c4);
      else if (MODE == 0) {   // pre-activation not saved by the forward: recompute it from the observation row
#pragma unroll
        for (int f = 0; f < NR; ++f) {
          const float on = of[f] * sob[f];
          z.x += on * wob[f].x; z.y += on * wob[f].y; z.z += on * wob[f].z; z.w += on * wob[f].w;
        }
      } else z = ld4(a.W + (long)a.idx[row * a.idx_stride] * W + c4);   // ... or gather it from the embedding table
      d = ld4(a.d0 + row * a.ldd0 + c4);
      if (a.d1) d = f4add(d, ld4(a.d1 + row * a.ldd1 + c4));
      if (a.d2) d = f4add(d, ld4(a.d2 + row * a.ldd2 + c4));
    }
    float4 x0 = make_float4(gelu_tanh(z.x), gelu_tanh(z.y), gelu_tanh(z.z), gelu_tanh(z.w));
    RmsFwd n = rms_fwd<W>(x0, sln);
    float4 dx0 = rms_bwd<W>(d, n, sln, dsln);
    float4 dz = make_float4(dx0.x * gelu_tanh_grad(z.x), dx0.y * gelu_tanh_grad(z.y), dx0.z * gelu_tanh_grad(z.z),
                            dx0.w * gelu_tanh_grad(z.w));
    if (!ok) dz = f4zero();
    if (ok && a.dz) st4(a.dz + row * a.lddz + c4, dz);
    if (MODE == 0) {
      // dW_obs[f] += o[f] * dz ; d(s_obs)[f] += (dz . W[f,:]) * obs[f] * rstd   (the obs need no gradient)
#pragma unroll
      for (int f = 0; f < NR; ++f) {
        const float on = of[f] * sob[f];
        wacc[f].x += on * dz.x; wacc[f].y += on * dz.y; wacc[f].z += on * dz.z; wacc[f].w += on * dz.w;
        const float dof = rowsum<W>(f4sum(f4mul(dz, wob[f])));
        dsobs[f] += dof * of[f];     // every lane of the row keeps the same partial; lane 0 of the row group publishes it
      }
    } else {
      const int id = ok ? a.idx[row * a.idx_stride] : -1;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if (k < a.nrows && id == k) { wacc[k].x += dz.x; wacc[k].y += dz.y; wacc[k].z += dz.z; wacc[k].w += dz.w; }
      }
    }
  }
  block_store_colsum<W>(dsln, a.slab_sln + (long)blockIdx.x * W, lds);
#pragma unroll
  for (int f = 0; f < NR; ++f)
    if (f < a.nrows) block_store_colsum<W>(wacc[f], a.slab_w + ((long)blockIdx.x * 32 + f) * W, lds);
  if (MODE == 0) {
    if (threadIdx.x < 32) sobs_acc[0][threadIdx.x] = sobs_acc[1][threadIdx.x] = sobs_acc[2][threadIdx.x] = sobs_acc[3][threadIdx.x] = 0.f;
    __syncthreads();
    // the first lane of every row of the wave holds the row's partial sum (lanes 0,16,32,48 for W = 64; 0,32 for W = 128)
#pragma unroll
    for (int f = 0; f < NR; ++f) {
      float v = dsobs[f];
      if (W == 64) v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane == 0) sobs_acc[wave][f] = v;
    }
    __syncthreads();
    if (threadIdx.x < 32)
      a.slab_sobs[(long)blockIdx.x * 32 + threadIdx.x] =
          (sobs_acc[0][threadIdx.x] + sobs_acc[1][threadIdx.x]) + (sobs_acc[2][threadIdx.x] + sobs_acc[3][threadIdx.x]);
  }
}

// Backward of a small-input ReLU layer Y = relu(X[R,F] @ W + b), F <= 32, N = 128 (actor pre-torso, torsos.py:36-47):
// dW[F][128] = X^T (dY * [Y > 0]), db = colsum(dY * [Y > 0]); 32 lanes x float4 per row.
template <int NF>
__global__ __launch_bounds__(256) void k_small_relu_wgrad(const float* __restrict__ X, int ldx, int F, const float* __restrict__ Yact,
                                                          const float* __restrict__ dY, float* __restrict__ slab_w /*[grid][33][128]*/, long R) {
  __shared__ float lds[4 * 128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & 31);
  float4 wacc[NF + 1];  // [NF] = bias
#pragma unroll
  for (int f = 0; f <= NF; ++f) wacc[f] = f4zero();
  for (long base = (long)blockIdx.x * 8; base < R; base += (long)gridDim.x * 8) {
    const long row = base + wave * 2 + (lane >> 5);
    if (row < R) {
      const float4 y = ld4(Yact + row * 128 + c4), d = ld4(dY + row * 128 + c4);
      const float4 g = make_float4(y.x > 0.f ? d.x : 0.f, y.y > 0.f ? d.y : 0.f, y.z > 0.f ? d.z : 0.f, y.w > 0.f ? d.w : 0.f);
      const float* x = X + row * ldx;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f < F) { const float xv = x[f]; wacc[f].x += xv * g.x; wacc[f].y += xv * g.y; wacc[f].z += xv * g.z; wacc[f].w += xv * g.w; }
      }
      wacc[NF] = f4add(wacc[NF], g);
    }
  }
#pragma unroll
  for (int ff = 0; ff <= NF; ++ff) {
    const int f = ff == NF ? 32 : ff;   // slab row 32 holds the bias gradient
    if (ff < F || ff == NF) {
      float4 v = wacc[ff];
      v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
      __syncthreads();
      if (lane < 32) st4(&lds[wave * 128 + c4], v);
      __syncthreads();
      if (threadIdx.x < 128)
        slab_w[((long)blockIdx.x * 33 + f) * 128 + threadIdx.x] =
            (lds[threadIdx.x] + lds[128 + threadIdx.x]) + (lds[256 + threadIdx.x] + lds[384 + threadIdx.x]);
    }
  }
}

// Materialise the (normalised) observation features padded to 64 columns, or the one-hot of an index,
// as the [R][64] left operand of the small weight-gradient GEMMs (dW_obs, dW_act, actor pre-torso).
__global__ void k_small_operand(const float* __restrict__ obs, int ldo, int F, const float* __restrict__ s_obs,
                                const int* __restrict__ idx, int idx_stride, float* __restrict__ out, long R, int mode) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * 64) return;
  long row = i >> 6;
  int f = (int)(i & 63);
  float v = 0.f;
  if (mode == 0) {        // rmsnorm_F(obs) * s_obs
    const float* o = obs + row * ldo;
    float ms = 0.f;
    for (int k = 0; k < F; ++k) ms += o[k] * o[k];
    if (f < F) v = o[f] * rsqrtf(ms / (float)F + NORM_EPS) * s_obs[f];
  } else if (mode == 1) { // one-hot(idx)
    v = (idx[row * idx_stride] == f) ? 1.f : 0.f;
  } else {                // raw obs
    if (f < F) v = obs[row * ldo + f];
  }
  out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// Retention epilogue: rn = GroupNorm(r) (per-row over the 64 head channels, fast variance),
// u = swish(gpre) * rn                                                  (retention.py:289-294)
template <int W> __global__ __launch_bounds__(256) void k_retpost_fwd(const float* __restrict__ r, int ldr, const float* __restrict__ gp, int ldg,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ u, int ldu, long R, int hs, int gs) {
  // flax GroupNorm(num_groups = n_head) on (token*head, hs) rows: groups of gs = hs / n_head channels; scale/bias [hs]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 ga = ld4(gamma + (c4 % hs)), be = ld4(beta + (c4 % hs));
  for (long base = (long)blockIdx.x * RowGeo<W>::RPB; base < R; base += (long)gridDim.x * RowGeo<W>::RPB) {
    const long row = base + wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
    const bool ok = row < R;
    float4 x = f4zero(), g = f4zero();
    if (ok) { x = ld4(r + row * ldr + c4); g = ld4(gp + row * ldg + c4); }
    const float mu = groupmean(x, gs);
    const float m2 = groupmean(f4mul(x, x), gs);
    const float rstd = rsqrtf(fmaxf(m2 - mu * mu, 0.f) + NORM_EPS);
    float4 rn = make_float4((x.x - mu) * rstd * ga.x + be.x, (x.y - mu) * rstd * ga.y + be.y,
                            (x.z - mu) * rstd * ga.z + be.z, (x.w - mu) * rstd * ga.w + be.w);
    float4 o = make_float4(swishf_(g.x) * rn.x, swishf_(g.y) * rn.y, swishf_(g.z) * rn.z, swishf_(g.w) * rn.w);
    if (ok) st4(u + row * ldu + c4, o);
  }
}

template <int W> __global__ __launch_bounds__(256) void k_retpost_bwd(const float* __restrict__ r, int ldr, const float* __restrict__ gp, int ldg,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ du, int lddu,
                                                     float* __restrict__ dr, int lddr, float* __restrict__ dgp, int lddg,
                                                     float* __restrict__ slab_gamma, float* __restrict__ slab_beta, long R, int hs, int gs) {
  __shared__ float lds[4 * W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 ga = ld4(gamma + (c4 % hs)), be = ld4(beta + (c4 % hs));
  float4 dga = f4zero(), dbe = f4zero();
  for (long base = (long)blockIdx.x * RowGeo<W>::RPB; base < R; base += (long)gridDim.x * RowGeo<W>::RPB) {
    const long row = base + wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
    const bool ok = row < R;
    float4 x = f4zero(), g = f4zero(), d = f4zero();
    if (ok) { x = ld4(r + row * ldr + c4); g = ld4(gp + row * ldg + c4); d = ld4(du + row * lddu + c4); }
    const float mu = groupmean(x, gs);
    const float m2 = groupmean(f4mul(x, x), gs);
    const float rstd = rsqrtf(fmaxf(m2 - mu * mu, 0.f) + NORM_EPS);
    float4 xh = make_float4((x.x - mu) * rstd, (x.y - mu) * rstd, (x.z - mu) * rstd, (x.w - mu) * rstd);
    float4 rn = f4add(f4mul(xh, ga), be);
    float4 sw = make_float4(swishf_(g.x), swishf_(g.y), swishf_(g.z), swishf_(g.w));
    float4 drn = f4mul(d, sw);
    float4 dg = make_float4(d.x * rn.x * swish_grad(g.x), d.y * rn.y * swish_grad(g.y), d.z * rn.z * swish_grad(g.z),
                            d.w * rn.w * swish_grad(g.w));
    dga = f4add(dga, f4mul(drn, xh));
    dbe = f4add(dbe, drn);
    float4 gg = f4mul(drn, ga);
    const float mg = groupmean(gg, gs);
    const float mgx = groupmean(f4mul(gg, xh), gs);
    float4 dx = make_float4((gg.x - mg - xh.x * mgx) * rstd, (gg.y - mg - xh.y * mgx) * rstd,
                            (gg.z - mg - xh.z * mgx) * rstd, (gg.w - mg - xh.w * mgx) * rstd);
    if (ok) { st4(dr + row * lddr + c4, dx); st4(dgp + row * lddg + c4, dg); }
  }
  block_store_colsum<W>(dga, slab_gamma + (long)blockIdx.x * W, lds);
  block_store_colsum<W>(dbe, slab_beta + (long)blockIdx.x * W, lds);
}

// ------------------------------------------------------------------------------------------------
// Residual + RMSNorm (+ second RMSNorm where the zero SwiGLU sits between two norms) (+ pe):
//   x1 = rmsnorm(a + y) * s1 ; x2 = s2 ? rmsnorm(x1) * s2 : x1 ; out = x2 ; outpe = x2 + pe[pos]
// (sable_network.py:69-70, 78-79, 203, 214-215, 233, 239-240)
struct ResNormArgs {
  const float* a; const float* y; int lda, ldy;
  const float* s1; const float* s2;
  const float* pe; const int* pos; int pos_stride; int npos;
  float* out; float* outpe; int ldout, ldoutpe;
  long R;
};
template <int W> __global__ __launch_bounds__(256) void k_resnorm_fwd(ResNormArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 s1 = ld4(p.s1 + c4);
  const float4 s2 = p.s2 ? ld4(p.s2 + c4) : f4zero();
  for (long base = (long)blockIdx.x * RowGeo<W>::RPB; base < p.R; base += (long)gridDim.x * RowGeo<W>::RPB) {
    const long row = base + wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
    const bool ok = row < p.R;
    float4 x = f4zero();
    if (ok) {
      x = ld4(p.a + row * p.lda + c4);
      if (p.y) x = f4add(x, ld4(p.y + row * p.ldy + c4));
    }
    float4 o = rms_fwd<W>(x, s1).y;
    if (p.s2) o = rms_fwd<W>(o, s2).y;
    if (ok) {
      if (p.out) st4(p.out + row * p.ldout + c4, o);
      if (p.outpe) {
        int ps = p.pos[row * p.pos_stride];
        ps = ps < 0 ? 0 : (ps >= p.npos ? p.npos - 1 : ps);
        st4(p.outpe + row * p.ldoutpe + c4, f4add(o, ld4(p.pe + (long)ps * W + c4)));
      }
    }
  }
}

struct ResNormBwdArgs {
  const float* a; const float* y; int lda, ldy;
  const float* s1; const float* s2;
  const float* d0; const float* d1; const float* d2; int ldd0, ldd1, ldd2;
  float* dsum; int lddsum;          // d(a + y)
  float* slab_s1; float* slab_s2;   // [grid][64]
  long R;
};
template <int W> __global__ __launch_bounds__(256) void k_resnorm_bwd(ResNormBwdArgs p) {
  __shared__ float lds[4 * W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 s1 = ld4(p.s1 + c4);
  const float4 s2 = p.s2 ? ld4(p.s2 + c4) : f4zero();
  float4 ds1 = f4zero(), ds2 = f4zero();
  for (long base = (long)blockIdx.x * RowGeo<W>::RPB; base < p.R; base += (long)gridDim.x * RowGeo<W>::RPB) {
    const long row = base + wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
    const bool ok = row < p.R;
    float4 x = f4zero(), d = f4zero();
    if (ok) {
      x = ld4(p.a + row * p.lda + c4);
      if (p.y) x = f4add(x, ld4(p.y + row * p.ldy + c4));
      d = ld4(p.d0 + row * p.ldd0 + c4);
      if (p.d1) d = f4add(d, ld4(p.d1 + row * p.ldd1 + c4));
      if (p.d2) d = f4add(d, ld4(p.d2 + row * p.ldd2 + c4));
    }
    RmsFwd n1 = rms_fwd<W>(x, s1);
    if (p.s2) {
      RmsFwd n2 = rms_fwd<W>(n1.y, s2);
      d = rms_bwd<W>(d, n2, s2, ds2);
    }
    float4 dx = rms_bwd<W>(d, n1, s1, ds1);
    if (ok) st4(p.dsum + row * p.lddsum + c4, dx);
  }
  block_store_colsum<W>(ds1, p.slab_s1 + (long)blockIdx.x * W, lds);
  if (p.s2) block_store_colsum<W>(ds2, p.slab_s2 + (long)blockIdx.x * W, lds);
}

// ------------------------------------------------------------------------------------------------
// Head middle: h = gelu(hpre) ; hn = rmsnorm(h) * s ; mode 0 -> out hn ; mode 1 -> value = hn . w + b
// (sable_network.py:102-109 value head, :277-284 logit head)
template <int W> __global__ __launch_bounds__(256) void k_headmid_fwd(const float* __restrict__ hpre, int ldh, const float* __restrict__ s,
                                                     float* __restrict__ hn, int ldhn,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ value, int value_stride, long R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 sc = ld4(s + c4);
  const float4 wv = w ? ld4(w + c4) : f4zero();
  for (long base = (long)blockIdx.x * RowGeo<W>::RPB; base < R; base += (long)gridDim.x * RowGeo<W>::RPB) {
    const long row = base + wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
    const bool ok = row < R;
    float4 z = ok ? ld4(hpre + row * ldh + c4) : f4zero();
    float4 h = make_float4(gelu_tanh(z.x), gelu_tanh(z.y), gelu_tanh(z.z), gelu_tanh(z.w));
    float4 o = rms_fwd<W>(h, sc).y;
    if (hn && ok) st4(hn + row * ldhn + c4, o);
    if (w) {
      float v = rowsum<W>(f4sum(f4mul(o, wv))) + b[0];
      if (ok && (lane & (RowGeo<W>::LPR - 1)) == 0) value[row * value_stride] = v;
    }
  }
}

// Backward: incoming either dhn [R,64] (mode 0) or dvalue [R] (mode 1, dhn = dvalue * w).
// Outputs dhpre; slabs ds[64]; mode 1 also dw[64] (slab) and db (slab_b[grid]).
template <int W> __global__ __launch_bounds__(256) void k_headmid_bwd(const float* __restrict__ hpre, int ldh, const float* __restrict__ s,
                                                     const float* __restrict__ dhn, int lddhn,
                                                     const float* __restrict__ w, const float* __restrict__ dvalue, int dvalue_stride,
                                                     float* __restrict__ dhpre, int lddh,
                                                     float* __restrict__ slab_s, float* __restrict__ slab_w, float* __restrict__ slab_b,
                                                     long R) {
  __shared__ float lds[4 * W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = 4 * (lane & (RowGeo<W>::LPR - 1));
  const float4 sc = ld4(s + c4);
  const float4 wv = w ? ld4(w + c4) : f4zero();
  float4 ds = f4zero(), dw = f4zero();
  float db = 0.f;
  for (long base = (long)blockIdx.x * RowGeo<W>::RPB; base < R; base += (long)gridDim.x * RowGeo<W>::RPB) {
    const long row = base + wave * RowGeo<W>::RPW + lane / RowGeo<W>::LPR;
    const bool ok = row < R;
    float4 z = ok ? ld4(hpre + row * ldh + c4) : f4zero();
    float4 h = make_float4(gelu_tanh(z.x), gelu_tanh(z.y), gelu_tanh(z.z), gelu_tanh(z.w));
    RmsFwd n = rms_fwd<W>(h, sc);
    float4 d;
    if (w) {
      const float dv = ok ? dvalue[row * dvalue_stride] : 0.f;
      d = f4scale(wv, dv);
      dw = f4add(dw, f4scale(n.y, dv));
      if ((lane & (RowGeo<W>::LPR - 1)) == 0) db += dv;
    } else {
      d = ok ? ld4(dhn + row * lddhn + c4) : f4zero();
    }
    float4 dh = rms_bwd<W>(d, n, sc, ds);
    float4 dz = make_float4(dh.x * gelu_tanh_grad(z.x), dh.y * gelu_tanh_grad(z.y), dh.z * gelu_tanh_grad(z.z),
                            dh.w * gelu_tanh_grad(z.w));
    if (ok) st4(dhpre + row * lddh + c4, dz);
  }
  block_store_colsum<W>(ds, slab_s + (long)blockIdx.x * W, lds);
  if (w) {
    block_store_colsum<W>(dw, slab_w + (long)blockIdx.x * W, lds);
    float v = wave_sum(db);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) slab_b[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  }
}

// generic elementwise helpers -------------------------------------------------------------------
// dst[i] += src[i]
__global__ void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = ld4(dst + i), b = ld4(src + i);
  st4(dst + i, f4add(a, b));
}
// dst[r][0..W) += src[r][0..W) for R rows with row strides ldd / lds (W, strides multiples of 4)
__global__ void k_add_rows(float* __restrict__ dst, long ldd, const float* __restrict__ src, long lds, long R, int W4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * W4) return;
  const long r = i / W4;
  const int c = 4 * (int)(i - r * W4);
  st4(dst + r * ldd + c, f4add(ld4(dst + r * ldd + c), ld4(src + r * lds + c)));
}
// y = relu'(act) * dy  (act is the post-relu activation), in place allowed
__global__ void k_relu_bwd(const float* __restrict__ act, const float* __restrict__ dy, float* __restrict__ dx, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = ld4(act + i), d = ld4(dy + i);
  st4(dx + i, make_float4(a.x > 0.f ? d.x : 0.f, a.y > 0.f ? d.y : 0.f, a.z > 0.f ? d.z : 0.f, a.w > 0.f ? d.w : 0.f));
}

}  // namespace magpo

using namespace magpo;

static int check_width(int E) {
  if (E != 64 && E != 128) { set_error("row kernels: the row width (embed_dim of the device network) must be 64 or 128"); return MAGPO_EINVAL; }
  return MAGPO_OK;
}

static inline int row_grid(long R) {
  long b = (R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int magpo_row_grid(long R) { return row_grid(R); }

extern "C" int magpo_pe_table(float* pe, int npos, int E, hipStream_t st) {
  int n = npos * E;
  hipLaunchKernelGGL(k_pe_table, dim3((n + 255) / 256), dim3(256), 0, st, pe, npos, E);
  return check_launch("magpo_pe_table");
}

extern "C" int magpo_embed_fwd(int mode, const float* obs, int ldo, int F, const float* s_obs, const float* W,
                               const int* idx, int idx_stride, const float* s_ln, const float* pe, const int* pos,
                               int pos_stride, int npos, float* z, int ldz, float* xn, int ldxn, float* kin, int ldkin,
                               long R, int E, hipStream_t st) {
  if (mode == 0 && F > 1024) { set_error("magpo_embed_fwd: F too large"); return MAGPO_EINVAL; }
  if (int e = check_width(E)) return e;
  EmbedArgs a{obs, ldo, F, s_obs, W, idx, idx_stride, s_ln, pe, pos, pos_stride, npos, z, xn, kin, ldz, ldxn, ldkin, R};
  if (E == 128) {
    hipLaunchKernelGGL(k_embed_fwd<128>, dim3(row_grid(R)), dim3(256), 0, st, a, mode);
    return check_launch("magpo_embed_fwd");
  }
  if (mode == 1 || F <= 8) {
    const long nb = (R + EMB_ROWS - 1) / EMB_ROWS;
    static const unsigned cap0 = resident_grid(k_embed_fwd_tile<0>, 256, 1L << 30), cap1 = resident_grid(k_embed_fwd_tile<1>, 256, 1L << 30);
    const unsigned cap = mode == 0 ? cap0 : cap1;
    const dim3 grid(nb < (long)cap ? (unsigned)nb : cap);
    if (mode == 0) hipLaunchKernelGGL(k_embed_fwd_tile<0>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_embed_fwd_tile<1>, grid, dim3(256), 0, st, a);
    return check_launch("magpo_embed_fwd");
  }
  hipLaunchKernelGGL(k_embed_fwd<64>, dim3(row_grid(R)), dim3(256), 0, st, a, mode);
  return check_launch("magpo_embed_fwd");
}

// slab_sln: [grid][64], slab_w: [grid][32][64], slab_sobs: [grid][32]; grid = magpo_row_grid(R).
// nrows = F (mode 0) or K+1 (mode 1), <= 32: rows of the small weight whose gradient slab is produced.
extern "C" int magpo_embed_bwd(int mode, const float* z, int ldz, const float* d0, int ldd0, const float* d1, int ldd1,
                               const float* d2, int ldd2, const float* s_ln, float* dz, int lddz, float* slab_sln,
                               float* slab_w, int nrows, const float* obs, int ldo, int F, const float* s_obs, const float* W,
                               float* slab_sobs, const int* idx, int idx_stride, long R, int E, hipStream_t st) {
  if (nrows < 1 || nrows > 32 || (mode == 0 && F != nrows)) { set_error("magpo_embed_bwd: 1 <= nrows <= 32 (F or K+1)"); return MAGPO_EINVAL; }
  if (int e = check_width(E)) return e;
  EmbedBwdArgs a{z, ldz, d0, d1, d2, ldd0, ldd1, ldd2, s_ln, dz, lddz, slab_sln, slab_w, nrows, obs, ldo, F, s_obs, W, slab_sobs, idx, idx_stride, R};
#define EB(M_, N_)                                                                                              \
  {                                                                                                             \
    if (E == 64) hipLaunchKernelGGL((k_embed_bwd<M_, N_, 64>), dim3(row_grid(R)), dim3(256), 0, st, a);          \
    else hipLaunchKernelGGL((k_embed_bwd<M_, N_, 128>), dim3(row_grid(R)), dim3(256), 0, st, a);                 \
  }
  if (mode == 0) { if (nrows <= 8) EB(0, 8) else if (nrows <= 16) EB(0, 16) else EB(0, 32) }
  else { if (nrows <= 8) EB(1, 8) else if (nrows <= 16) EB(1, 16) else if (nrows <= 24) EB(1, 24) else EB(1, 32) }
#undef EB
  return check_launch("magpo_embed_bwd");
}

// slab_w: [grid][33][128] (rows 0..F-1 = dW, row 32 = db); grid = magpo_row_grid(R)
extern "C" int magpo_small_relu_wgrad(const float* X, int ldx, int F, const float* Yact, const float* dY, float* slab_w, long R,
                                      hipStream_t st) {
  if (F < 1 || F > 32) { set_error("magpo_small_relu_wgrad: 1 <= F <= 32"); return MAGPO_EINVAL; }
  if (F <= 8) hipLaunchKernelGGL(k_small_relu_wgrad<8>, dim3(row_grid(R)), dim3(256), 0, st, X, ldx, F, Yact, dY, slab_w, R);
  else if (F <= 16) hipLaunchKernelGGL(k_small_relu_wgrad<16>, dim3(row_grid(R)), dim3(256), 0, st, X, ldx, F, Yact, dY, slab_w, R);
  else hipLaunchKernelGGL(k_small_relu_wgrad<32>, dim3(row_grid(R)), dim3(256), 0, st, X, ldx, F, Yact, dY, slab_w, R);
  return check_launch("magpo_small_relu_wgrad");
}

extern "C" int magpo_small_operand(int mode, const float* obs, int ldo, int F, const float* s_obs, const int* idx,
                                   int idx_stride, float* out, long R, hipStream_t st) {
  if (F > 64) { set_error("magpo_small_operand: F > 64 not supported"); return MAGPO_EINVAL; }
  long n = R * 64;
  hipLaunchKernelGGL(k_small_operand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, obs, ldo, F, s_obs, idx, idx_stride, out, R, mode);
  return check_launch("magpo_small_operand");
}

static int check_groups(int hs, int gs, int E) {
  if (int e = check_width(E)) return e;
  if (hs < 4 || hs > E || (E % hs) || gs < 4 || gs > hs || (gs & (gs - 1))) {
    set_error("retpost: head width must divide the row width and the group size must be a power of two in [4, hs]");
    return MAGPO_EINVAL;
  }
  return MAGPO_OK;
}
#define ROW_LAUNCH(K_, ...)                                                                           \
  {                                                                                                   \
    if (E == 64) hipLaunchKernelGGL(K_<64>, dim3(row_grid(R)), dim3(256), 0, st, __VA_ARGS__);         \
    else hipLaunchKernelGGL(K_<128>, dim3(row_grid(R)), dim3(256), 0, st, __VA_ARGS__);                \
  }

extern "C" int magpo_retpost_fwd(const float* r, int ldr, const float* gp, int ldg, const float* gamma, const float* beta,
                                 float* u, int ldu, long R, int hs, int gs, int E, hipStream_t st) {
  if (int e = check_groups(hs, gs, E)) return e;
  ROW_LAUNCH(k_retpost_fwd, r, ldr, gp, ldg, gamma, beta, u, ldu, R, hs, gs)
  return check_launch("magpo_retpost_fwd");
}

extern "C" int magpo_retpost_bwd(const float* r, int ldr, const float* gp, int ldg, const float* gamma, const float* beta,
                                 const float* du, int lddu, float* dr, int lddr, float* dgp, int lddg, float* slab_gamma,
                                 float* slab_beta, long R, int hs, int gs, int E, hipStream_t st) {
  if (int e = check_groups(hs, gs, E)) return e;
  ROW_LAUNCH(k_retpost_bwd, r, ldr, gp, ldg, gamma, beta, du, lddu, dr, lddr, dgp, lddg, slab_gamma, slab_beta, R, hs, gs)
  return check_launch("magpo_retpost_bwd");
}

extern "C" int magpo_resnorm_fwd(const float* a, int lda, const float* y, int ldy, const float* s1, const float* s2,
                                 const float* pe, const int* pos, int pos_stride, int npos, float* out, int ldout,
                                 float* outpe, int ldoutpe, long R, int E, hipStream_t st) {
  if (int e = check_width(E)) return e;
  ResNormArgs p{a, y, lda, ldy, s1, s2, pe, pos, pos_stride, npos, out, outpe, ldout, ldoutpe, R};
  ROW_LAUNCH(k_resnorm_fwd, p)
  return check_launch("magpo_resnorm_fwd");
}

extern "C" int magpo_resnorm_bwd(const float* a, int lda, const float* y, int ldy, const float* s1, const float* s2,
                                 const float* d0, int ldd0, const float* d1, int ldd1, const float* d2, int ldd2,
                                 float* dsum, int lddsum, float* slab_s1, float* slab_s2, long R, int E, hipStream_t st) {
  if (int e = check_width(E)) return e;
  ResNormBwdArgs p{a, y, lda, ldy, s1, s2, d0, d1, d2, ldd0, ldd1, ldd2, dsum, lddsum, slab_s1, slab_s2, R};
  ROW_LAUNCH(k_resnorm_bwd, p)
  return check_launch("magpo_resnorm_bwd");
}

extern "C" int magpo_headmid_fwd(const float* hpre, int ldh, const float* s, float* hn, int ldhn, const float* w,
                                 const float* b, float* value, int value_stride, long R, int E, hipStream_t st) {
  if (int e = check_width(E)) return e;
  ROW_LAUNCH(k_headmid_fwd, hpre, ldh, s, hn, ldhn, w, b, value, value_stride, R)
  return check_launch("magpo_headmid_fwd");
}

extern "C" int magpo_headmid_bwd(const float* hpre, int ldh, const float* s, const float* dhn, int lddhn, const float* w,
                                 const float* dvalue, int dvalue_stride, float* dhpre, int lddh, float* slab_s,
                                 float* slab_w, float* slab_b, long R, int E, hipStream_t st) {
  if (int e = check_width(E)) return e;
  ROW_LAUNCH(k_headmid_bwd, hpre, ldh, s, dhn, lddhn, w, dvalue, dvalue_stride, dhpre, lddh, slab_s, slab_w, slab_b, R)
  return check_launch("magpo_headmid_bwd");
}

extern "C" int magpo_add_inplace(float* dst, const float* src, long n, hipStream_t st) {
  if (n & 3) { set_error("magpo_add_inplace: n must be a multiple of 4"); return MAGPO_EINVAL; }
  hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, dst, src, n);
  return check_launch("magpo_add_inplace");
}

extern "C" int magpo_add_rows(float* dst, long ldd, const float* src, long lds, long R, int W, hipStream_t st) {
  if ((W & 3) || (ldd & 3) || (lds & 3) || W < 4) { set_error("magpo_add_rows: W and the row strides must be multiples of 4"); return MAGPO_EINVAL; }
  if (R <= 0) return MAGPO_OK;
  const long n = R * (W / 4);
  hipLaunchKernelGGL(k_add_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, ldd, src, lds, R, W / 4);
  return check_launch("magpo_add_rows");
}

extern "C" int magpo_relu_bwd(const float* act, const float* dy, float* dx, long n, hipStream_t st) {
  if (n & 3) { set_error("magpo_relu_bwd: n must be a multiple of 4"); return MAGPO_EINVAL; }
  hipLaunchKernelGGL(k_relu_bwd, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, act, dy, dx, n);
  return check_launch("magpo_relu_bwd");
}
