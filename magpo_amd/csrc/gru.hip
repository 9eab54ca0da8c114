// GRU actor (ScannedRNN, mava/networks/base.py:121-149; flax.linen.GRUCell) on fp32 MFMA for gfx950.
//
//   r = sigmoid(xi_r + h W_hr) ; z = sigmoid(xi_z + h W_hz) ; n = tanh(xi_n + r * (h W_hn + b_hn))
//   h' = (1 - z) n + z h ,   h <- 0 before the step wherever the reset flag is set  (base.py:136-141)
// xi = emb @ [W_ir|W_iz|W_in] + [b_ir|b_iz|b_in] is computed for all timesteps up front by magpo_linear;
// only the recurrent part h @ W_h is sequential.  One workgroup owns 64 recurrent rows (sequence x agent)
// for the whole scan: h lives in LDS (double-buffered [64][132]), W_h^T fragments stream from L2.
// Each wave computes the r, z and n accumulators of the same 32x32 (row, column) block, so the gate
// math runs on the accumulators with no LDS round trip.  The backward scan carries dL/dh in LDS and
// needs one GEMM per step, dh_prev = dhh @ W_h^T, because the forward saves its gates.
#include "common.hpp"

namespace magpo {

constexpr int H = 128;
constexpr int HP = H + LDP;        // h tile pitch
constexpr int G3 = 3 * H;
constexpr int G3P = G3 + LDP;

struct GruArgs {
  const float* xi;        // [R][3H]
  const float* Wht;       // [3H][H]  (W_h transposed: row n = output column n of [W_hr|W_hz|W_hn])
  const float* b_hn;      // [H]
  const float* h0;        // [*][H]
  const int* h0_idx;      // [NR] row of h0 per recurrent row (nullable: identity)
  const unsigned char* reset;  // [nseq][T] reset-before-step flags
  float* hs;              // [R][H] h after each step
  float* gates;           // [R][4H] r | z | n | (h W_hn + b_hn)   (nullable: acting)
  float* hprev;           // [R][H] reset-applied state each step started from (nullable: acting)
  int T, A, NR;           // NR = nseq * A recurrent rows
};

__device__ __forceinline__ long tok_row(int rho, int t, int T, int A) {
  int seq = rho / A, ag = rho - seq * A;
  return ((long)seq * T + t) * A + ag;
}

__global__ __launch_bounds__(256) void k_gru_scan_fwd(GruArgs a) {
  __shared__ __align__(16) float hbuf[2][64 * HP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int rho0 = blockIdx.x * 64;
  // initial carry (with the reset of step 0 applied)
  for (int i = tid; i < 64 * (H / 4); i += 256) {
    int r = i / (H / 4), c4 = i - r * (H / 4);
    int rho = rho0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rho < a.NR) {
      int seq = rho / a.A;
      if (!a.reset[(long)seq * a.T]) {
        long src = a.h0_idx ? a.h0_idx[rho] : rho;
        v = *reinterpret_cast<const float4*>(a.h0 + src * H + 4 * c4);
      }
    }
    *reinterpret_cast<float4*>(&hbuf[0][r * HP + 4 * c4]) = v;
  }
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    const float* hold = hbuf[t & 1];
    float* hnew = hbuf[(t + 1) & 1];
#pragma unroll 1
    for (int job = 0; job < 2; ++job) {
      const int jb = wave * 2 + job, wr = jb & 1, cb = jb >> 1;
      const int col = 32 * cb + lr;
      f32x16 ar, az, an;
#pragma unroll
      for (int i = 0; i < 16; ++i) { ar[i] = 0.f; az[i] = 0.f; an[i] = 0.f; }
#pragma unroll 1
      for (int kc = 0; kc < 2; ++kc) {
        const float* ap = hold + (32 * wr + lr) * HP + kc * 64 + 32 * h;
        const float* br = a.Wht + (long)col * H + kc * 64 + 32 * h;
        const float* bz = br + (long)H * H;
        const float* bn = bz + (long)H * H;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 av = *reinterpret_cast<const float4*>(ap + 4 * u);
          const float4 r4 = *reinterpret_cast<const float4*>(br + 4 * u);
          const float4 z4 = *reinterpret_cast<const float4*>(bz + 4 * u);
          const float4 n4 = *reinterpret_cast<const float4*>(bn + 4 * u);
          ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, r4.x, ar, 0, 0, 0);
          az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, z4.x, az, 0, 0, 0);
          an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, n4.x, an, 0, 0, 0);
          ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, r4.y, ar, 0, 0, 0);
          az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, z4.y, az, 0, 0, 0);
          an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, n4.y, an, 0, 0, 0);
          ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, r4.z, ar, 0, 0, 0);
          az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, z4.z, az, 0, 0, 0);
          an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, n4.z, an, 0, 0, 0);
          ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, r4.w, ar, 0, 0, 0);
          az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, z4.w, az, 0, 0, 0);
          an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, n4.w, an, 0, 0, 0);
        }
      }
      const float bhn = a.b_hn[col];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int rho = rho0 + rl;
        float hn_new = 0.f;
        if (rho < a.NR) {
          const long row = tok_row(rho, t, a.T, a.A);
          const float* x = a.xi + row * G3;
          const float hb = an[i] + bhn;
          const float r = sigmoidf_(x[col] + ar[i]);
          const float z = sigmoidf_(x[H + col] + az[i]);
          const float n = tanhf(x[2 * H + col] + r * hb);
          const float hp = hold[rl * HP + col];
          hn_new = (1.0f - z) * n + z * hp;
          a.hs[row * H + col] = hn_new;
          if (a.gates) {
            float* g = a.gates + row * (4 * H);
            g[col] = r; g[H + col] = z; g[2 * H + col] = n; g[3 * H + col] = hb;
          }
          if (a.hprev) a.hprev[row * H + col] = hp;
          if (t + 1 < a.T && a.reset[(long)(rho / a.A) * a.T + t + 1]) hn_new = 0.f;
        }
        hnew[rl * HP + col] = hn_new;
      }
    }
    __syncthreads();
  }
}

struct GruBwdArgs {
  const float* gates;     // [R][4H]
  const float* hprev;     // [R][H] from the forward
  const unsigned char* reset;
  const float* dhs;       // [R][H] dL/dh_t from the post-torso path
  const float* Wh;        // [H][3H] natural layout (used as "Wt" of dh_prev = dhh @ W_h^T)
  float* dxi;             // [R][3H]
  float* dhh;             // [R][3H]
  float* slab_bhn;        // [grid][H]
  int T, A, NR;
};

__global__ __launch_bounds__(256) void k_gru_scan_bwd(GruBwdArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* dht = smem;                 // [64][HP]   dL/dh carried from step t+1 (already includes the direct z path)
  float* dhht = dht + 64 * HP;       // [64][G3P]  dhh of the current step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int rho0 = blockIdx.x * 64;
  for (int i = tid; i < 64 * HP; i += 256) dht[i] = 0.f;
  float bacc = 0.f;  // thread owns column (tid & 127) for rows (tid >> 7) + 2k
  __syncthreads();
  for (int t = a.T - 1; t >= 0; --t) {
    // ---- elementwise phase: thread handles column c = tid & 127, rows rl = (tid >> 7) + 2 k
    const int c = tid & 127;
    for (int rl = tid >> 7; rl < 64; rl += 2) {
      const int rho = rho0 + rl;
      float d_r = 0.f, d_z = 0.f, d_n = 0.f, d_hb = 0.f, carry = 0.f;
      if (rho < a.NR) {
        const int seq = rho / a.A;
        const long row = tok_row(rho, t, a.T, a.A);
        const bool rst = a.reset[(long)seq * a.T + t] != 0;
        const float hp = a.hprev[row * H + c];
        const float* g = a.gates + row * (4 * H);
        const float r = g[c], z = g[H + c], n = g[2 * H + c], hb = g[3 * H + c];
        const float dh = a.dhs[row * H + c] + dht[rl * HP + c];
        const float dn = dh * (1.0f - z);
        const float dz = dh * (hp - n);
        const float dan = dn * (1.0f - n * n);
        d_n = dan;
        d_hb = dan * r;
        d_r = dan * hb * r * (1.0f - r);
        d_z = dz * z * (1.0f - z);
        carry = rst ? 0.f : dh * z;
        float* dx = a.dxi + row * G3;
        dx[c] = d_r; dx[H + c] = d_z; dx[2 * H + c] = d_n;
        float* dq = a.dhh + row * G3;
        dq[c] = d_r; dq[H + c] = d_z; dq[2 * H + c] = d_hb;
        bacc += d_hb;
        if (rst) { d_r = 0.f; d_z = 0.f; d_hb = 0.f; }  // no gradient into the (zeroed) previous state
      }
      dhht[rl * G3P + c] = d_r;
      dhht[rl * G3P + H + c] = d_z;
      dhht[rl * G3P + 2 * H + c] = d_hb;
      dht[rl * HP + c] = carry;  // direct path; the GEMM below adds dhh @ W_h^T
    }
    __syncthreads();
    // ---- dh_prev += dhh[64][3H] @ W_h^T  -> [64][H]; 8 quadrant jobs, 2 per wave
#pragma unroll 1
    for (int job = 0; job < 2; ++job) {
      const int jb = wave * 2 + job, wr = jb & 1, cb = jb >> 1;
      const int col = 32 * cb + lr;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 1
      for (int kc = 0; kc < G3 / 64; ++kc) {
        const float* ap = dhht + (32 * wr + lr) * G3P + kc * 64 + 32 * h;
        const float* bp = a.Wh + (long)col * G3 + kc * 64 + 32 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 av = *reinterpret_cast<const float4*>(ap + 4 * u);
          const float4 bv = *reinterpret_cast<const float4*>(bp + 4 * u);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        dht[rl * HP + col] += acc[i];
      }
    }
    __syncthreads();
  }
  // b_hn gradient: sum over the two row-parities handled by threads c and c+128
  __shared__ float bsh[256];
  bsh[tid] = bacc;
  __syncthreads();
  if (tid < 128) a.slab_bhn[(long)blockIdx.x * H + tid] = bsh[tid] + bsh[tid + 128];
}

// Y[R][N] = act(X[R][F] @ W[F][N] + b) for small F (actor pre-torso, torsos.py:36-47)
__global__ void k_small_linear(const float* __restrict__ X, int ldx, int F, const float* __restrict__ W, const float* __restrict__ b,
                               float* __restrict__ Y, int ldy, int N, long R, int relu) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = N / 4;
  if (i >= R * n4) return;
  long row = i / n4;
  int c4 = 4 * (int)(i - row * n4);
  float4 acc = *reinterpret_cast<const float4*>(b + c4);
  const float* x = X + row * ldx;
  for (int f = 0; f < F; ++f) {
    const float xv = x[f];
    const float4 w = *reinterpret_cast<const float4*>(W + (long)f * N + c4);
    acc.x += xv * w.x; acc.y += xv * w.y; acc.z += xv * w.z; acc.w += xv * w.w;
  }
  if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
  *reinterpret_cast<float4*>(Y + row * ldy + c4) = acc;
}

}  // namespace magpo

using namespace magpo;

extern "C" int magpo_gru_scan_fwd(const float* xi, const float* Wht, const float* b_hn, const float* h0, const int* h0_idx,
                                  const unsigned char* reset, float* hs, float* gates, float* hprev, int nseq, int T, int A,
                                  hipStream_t st) {
  GruArgs a{xi, Wht, b_hn, h0, h0_idx, reset, hs, gates, hprev, T, A, nseq * A};
  hipLaunchKernelGGL(k_gru_scan_fwd, dim3((a.NR + 63) / 64), dim3(256), 0, st, a);
  return check_launch("magpo_gru_scan_fwd");
}

// slab_bhn: [ceil(NR/64)][128]
extern "C" int magpo_gru_scan_bwd(const float* gates, const float* hprev, const unsigned char* reset, const float* dhs,
                                  const float* Wh, float* dxi, float* dhh, float* slab_bhn, int nseq, int T, int A,
                                  hipStream_t st) {
  GruBwdArgs a{gates, hprev, reset, dhs, Wh, dxi, dhh, slab_bhn, T, A, nseq * A};
  size_t lds = (size_t)(64 * HP + 64 * G3P) * sizeof(float);
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(k_gru_scan_bwd, dim3((a.NR + 63) / 64), dim3(256), lds, st, a);
  return check_launch("magpo_gru_scan_bwd");
}

extern "C" int magpo_small_linear(const float* X, int ldx, int F, const float* W, const float* b, float* Y, int ldy, int N,
                                  long R, int relu, hipStream_t st) {
  if (N & 3) { set_error("magpo_small_linear: N must be a multiple of 4"); return MAGPO_EINVAL; }
  long n = R * (N / 4);
  hipLaunchKernelGGL(k_small_linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, ldx, F, W, b, Y, ldy, N, R, relu);
  return check_launch("magpo_small_linear");
}
