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
#include <stdlib.h>

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
  float* gates;           // [R][4H] (r, z, n, h W_hn + b_hn) interleaved per hidden column: opaque to the host (nullable: acting)
  float* hprev;           // [R][H] reset-applied state each step started from (nullable: acting)
  int T, A, NR;           // NR = nseq * A recurrent rows
  int time_major;         // 0: rows (seq, t, agent), reset [nseq][T];  1: rows (t, seq, agent), reset [T][nseq] (rollout trajectory)
  float* h_last;          // [NR][H] state after the last step (nullable)
  const int* xi_cls;      // [R] nullable: row r takes xi[xi_cls[r]] (xi = table over the distinct input rows, csrc/classtab.hip)
};

#ifdef MAGPO_GRU_PROF
__device__ unsigned long long g_gru_prof[8];
#define GP_DECL() unsigned long long gp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long gp_last = clock64();
#define GP(k) do { unsigned long long t_ = clock64(); gp_acc[k] += t_ - gp_last; gp_last = t_; } while (0)
#define GP_FLUSH() do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_gru_prof[k_], gp_acc[k_]); } } while (0)
#else
#define GP_DECL()
#define GP(k)
#define GP_FLUSH()
#endif

__device__ __forceinline__ long tok_row(int rho, int t, int T, int A) {
  int seq = rho / A, ag = rho - seq * A;
  return ((long)seq * T + t) * A + ag;
}

// W_h^T fragments (3 gates x 128 k x 32 columns per wave = 192 VGPRs) stay in registers for the whole scan;
// wave w owns hidden columns 32w..32w+31 for both 32-row halves.  One wave per SIMD (launch bound 1).
// ROWS = 64 recurrent rows per workgroup, or 32 when 64-row blocks would leave compute units idle (NR / 64 < 256: the scan is a
// latency chain of T steps, so half-size blocks on twice the CUs halve its time).
// FULL = every one of the block's 64 rows is valid (all but the last block): loads and stores are then unconditional
// straight-line code; a predicated access is an exec-masked block with its own wait and a per-element global flag load
// sits on the step's critical path, so the reset flags of the block's rows are staged in LDS once.
// MODE (compile-time, so that the training scan is branch-free with a constant row stride): 0 = training scan, h / gates /
// h_prev of every step saved; 1 = time-major trajectory (rollout carry: rows (t, env, agent), only the last state is
// written); 2 = any of the save buffers may be NULL (acting step)
template <bool FULL, int MODE, int ROWS>
__global__ __launch_bounds__(256, 1) void k_gru_scan_fwd(GruArgs a, int block0) {
  constexpr bool TM = MODE == 1;
  __shared__ __align__(16) float hbuf[2][ROWS * HP];
  extern __shared__ unsigned char rflag[];   // [ROWS][T] reset-before-step flags of the block's rows, then (xi_cls) [ROWS][T] int xi rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int rho0 = (block0 + blockIdx.x) * ROWS;
  const int col = 32 * wave + lr;
  const int T = a.T;
  float4 wf[3][16];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int u = 0; u < 16; ++u) wf[g][u] = *reinterpret_cast<const float4*>(a.Wht + ((long)g * H + col) * H + 64 * h + 4 * u);
  const float bhn = a.b_hn[col];
  for (int i = tid; i < ROWS * T; i += 256) {
    const int rl = i / T, t = i - rl * T;
    const int rho = min(rho0 + rl, a.NR - 1);
    rflag[i] = TM ? a.reset[(long)t * (a.NR / a.A) + rho / a.A] : a.reset[(long)(rho / a.A) * T + t];
  }
  // xi rows through the class table: the block's 64 x T class indices are staged once, like the flags (a dependent global
  // load per step would sit in front of every xi load)
  int* ctab = reinterpret_cast<int*>(rflag + ROWS * T);
  const bool by_cls = a.xi_cls != nullptr;
  if (by_cls) {
    for (int i = tid; i < ROWS * T; i += 256) {
      const int rl = i / T, t = i - rl * T;
      const int rho = min(rho0 + rl, a.NR - 1);
      ctab[i] = a.xi_cls[TM ? (long)t * a.NR + rho : tok_row(rho, t, T, a.A)];
    }
  }
  // initial carry (with the reset of step 0 applied)
  for (int i = tid; i < ROWS * (H / 4); i += 256) {
    int r = i / (H / 4), c4 = i - r * (H / 4);
    int rho = rho0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rho < a.NR) {
      int seq = rho / a.A;
      if (!(TM ? a.reset[seq] : a.reset[(long)seq * T])) {
        long src = a.h0_idx ? a.h0_idx[rho] : rho;
        v = *reinterpret_cast<const float4*>(a.h0 + src * H + 4 * c4);
      }
    }
    *reinterpret_cast<float4*>(&hbuf[0][r * HP + 4 * c4]) = v;
  }
  // per-row bookkeeping (no integer division inside the scan): token row of step 0 (invalid rows shadow the last valid one)
  __shared__ long rbase[ROWS];
  if (tid < ROWS) rbase[tid] = TM ? (long)min(rho0 + tid, a.NR - 1) : tok_row(min(rho0 + tid, a.NR - 1), 0, T, a.A);
  const long t_stride = TM ? a.NR : a.A;   // rows between consecutive steps of one recurrent row
  __syncthreads();
  GP_DECL();
  for (int t = 0; t < T; ++t) {
    const float* hold = hbuf[t & 1];
    float* hnew = hbuf[(t + 1) & 1];
#pragma unroll
    for (int wr = 0; wr < ROWS / 32; ++wr) {
      GP(6);
      // issue this job's xi loads first: they are in flight under the 192 MFMAs below
      float xr[16], xz[16], xn[16];
      long rowi[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        const long row = rbase[rl] + (long)t * t_stride;
        rowi[i] = row;
        const float* x = a.xi + (by_cls ? (long)ctab[rl * T + t] : row) * G3;
        xr[i] = x[col]; xz[i] = x[H + col]; xn[i] = x[2 * H + col];
      }
      // carry values and next-step reset flags of the job's rows: LDS round trips hidden under the MFMAs as well
      float hpv[16];
      unsigned char rf[16];
      const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        hpv[i] = hold[rl * HP + col];
        rf[i] = rflag[rl * T + tn];
      }
      GP(0 + 3 * wr);
      f32x16 ar, az, an;
#pragma unroll
      for (int i = 0; i < 16; ++i) { ar[i] = 0.f; az[i] = 0.f; an[i] = 0.f; }
      const float* ap = hold + (32 * wr + lr) * HP + 64 * h;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float4 av = *reinterpret_cast<const float4*>(ap + 4 * u);
        ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wf[0][u].x, ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wf[1][u].x, az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wf[2][u].x, an, 0, 0, 0);
        ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wf[0][u].y, ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wf[1][u].y, az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wf[2][u].y, an, 0, 0, 0);
        ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wf[0][u].z, ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wf[1][u].z, az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wf[2][u].z, an, 0, 0, 0);
        ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wf[0][u].w, ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wf[1][u].w, az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wf[2][u].w, an, 0, 0, 0);
      }
      GP(1 + 3 * wr);
      const bool more = t + 1 < T;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        const long row = rowi[i];
        const float hb = an[i] + bhn;
        const float r = fast_sigmoid(xr[i] + ar[i]);
        const float z = fast_sigmoid(xz[i] + az[i]);
        const float n = fast_tanh(xn[i] + r * hb);
        const float hp = hpv[i];
        float hn_new = (1.0f - z) * n + z * hp;
        if (FULL || rho0 + rl < a.NR) {
          if (MODE == 0) {
            a.hs[row * H + col] = hn_new;
            *reinterpret_cast<float4*>(a.gates + row * (4 * H) + 4 * col) = make_float4(r, z, n, hb);   // one 16-byte store per element
            a.hprev[row * H + col] = hp;
          } else if (MODE == 1) {
            if (!more) a.h_last[(long)(rho0 + rl) * H + col] = hn_new;
          } else {
            if (a.hs) a.hs[row * H + col] = hn_new;
            if (a.gates) {
              *reinterpret_cast<float4*>(a.gates + row * (4 * H) + 4 * col) = make_float4(r, z, n, hb);
            }
            if (a.hprev) a.hprev[row * H + col] = hp;
          }
        }
        hnew[rl * HP + col] = (more & (rf[i] != 0)) ? 0.f : hn_new;
      }
      GP(2 + 3 * wr);
    }
    __syncthreads();
    GP(7);
  }
  GP_FLUSH();
}

// ---- split-bf16 (x3) variant of the training scan ---------------------------------------------------------------------------
// fp32 MFMA runs at the fp32 VALU rate on gfx950 (1/16 of bf16): the recurrent GEMM h W_h of one step is 10 us of the 19 us a
// step takes.  Here both operands are split x = hi + lo (hi = bf16(x), lo = bf16(x - hi); 16 mantissa bits together) and the
// product is taken as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: relative error of a product
// ~2^-16 (SURVEY 7 names split-bf16 x3 as the alternative to fp32 MFMA), 3 x 1/16 of the MFMA time.  W_h is split once per
// kernel (same 192 VGPRs as the fp32 fragments); h is written to LDS as two bf16 tiles when the gate phase produces it (each
// element is produced once and read by all four waves).  The carry itself (z * h_prev) stays exact fp32.
constexpr int HB = H + 8;   // bf16 tile pitch (elements): 272-byte rows, 16-byte aligned

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

template <bool FULL, int NP>
__global__ __launch_bounds__(256, 1) void k_gru_scan_fwd_bf3(GruArgs a, int block0) {
  constexpr int ROWS = NP == 3 ? 32 : 64;   // recurrent rows per workgroup: three pieces double-buffered for 64 rows would not fit the 160 KB of LDS
  __shared__ __align__(16) float hf[ROWS * HP];            // fp32 state (carry term), updated in place by the lane that owns the element
  __shared__ __align__(16) __bf16 hsp[2][NP][ROWS * HB];   // [buffer][piece][row][k] MFMA A operand
  extern __shared__ unsigned char rflag[];                 // [ROWS][T] reset flags, then (xi_cls) [ROWS][T] int xi rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int rho0 = (block0 + blockIdx.x) * ROWS;
  const int col = 32 * wave + lr;
  const int T = a.T;
  // B fragments: gate g, k-step s: k = 64 h + 8 s + j (j = 0..7), column col
  bf16x8 wp[NP][3][8];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const float* w = a.Wht + ((long)g * H + col) * H + 64 * h + 8 * s8;
      const float4 w0 = *reinterpret_cast<const float4*>(w), w1 = *reinterpret_cast<const float4*>(w + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 pc[NP];
        split_pieces<NP>(wv[j], pc);
#pragma unroll
        for (int q = 0; q < NP; ++q) wp[q][g][s8][j] = pc[q];
      }
    }
  const float bhn = a.b_hn[col];
  for (int i = tid; i < ROWS * T; i += 256) {
    const int rl = i / T, t = i - rl * T;
    const int rho = min(rho0 + rl, a.NR - 1);
    rflag[i] = a.reset[(long)(rho / a.A) * T + t];
  }
  int* ctab = reinterpret_cast<int*>(rflag + ROWS * T);
  const bool by_cls = a.xi_cls != nullptr;
  if (by_cls) {
    for (int i = tid; i < ROWS * T; i += 256) {
      const int rl = i / T, t = i - rl * T;
      ctab[i] = a.xi_cls[tok_row(min(rho0 + rl, a.NR - 1), t, T, a.A)];
    }
  }
  for (int i = tid; i < ROWS * (H / 4); i += 256) {   // initial carry (with the reset of step 0 applied)
    const int r = i / (H / 4), c4 = i - r * (H / 4);
    const int rho = rho0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rho < a.NR && !a.reset[(long)(rho / a.A) * T]) {
      const long src = a.h0_idx ? a.h0_idx[rho] : rho;
      v = *reinterpret_cast<const float4*>(a.h0 + src * H + 4 * c4);
    }
    *reinterpret_cast<float4*>(&hf[r * HP + 4 * c4]) = v;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __bf16 pc[NP];
      split_pieces<NP>(vv[j], pc);
#pragma unroll
      for (int q = 0; q < NP; ++q) hsp[0][q][r * HB + 4 * c4 + j] = pc[q];
    }
  }
  __shared__ long rbase[ROWS];
  if (tid < ROWS) rbase[tid] = tok_row(min(rho0 + tid, a.NR - 1), 0, T, a.A);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1, nxt = (t + 1) & 1;
#pragma unroll
    for (int wr = 0; wr < ROWS / 32; ++wr) {
      float xr[16], xz[16], xn[16];
      long rowi[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        const long row = rbase[rl] + (long)t * a.A;
        rowi[i] = row;
        const float* x = a.xi + (by_cls ? (long)ctab[rl * T + t] : row) * G3;
        xr[i] = x[col]; xz[i] = x[H + col]; xn[i] = x[2 * H + col];
      }
      float hpv[16];
      unsigned char rf[16];
      const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        hpv[i] = hf[rl * HP + col];
        rf[i] = rflag[rl * T + tn];
      }
      f32x16 ar, az, an;
#pragma unroll
      for (int i = 0; i < 16; ++i) { ar[i] = 0.f; az[i] = 0.f; an[i] = 0.f; }
      const int aoff = (32 * wr + lr) * HB + 64 * h;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        bf16x8 xp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) xp[q] = *reinterpret_cast<const bf16x8*>(&hsp[cur][q][aoff + 8 * s8]);
        // products in decreasing order of magnitude: (0,0) (0,1) (1,0) [(0,2) (2,0) (1,1)]
#pragma unroll
        for (int pr = 0; pr < (NP == 3 ? 6 : 3); ++pr) {
          const int qa = pr == 0 ? 0 : (pr == 1 ? 0 : (pr == 2 ? 1 : (pr == 3 ? 0 : (pr == 4 ? 2 : 1))));
          const int qb = pr == 0 ? 0 : (pr == 1 ? 1 : (pr == 2 ? 0 : (pr == 3 ? 2 : (pr == 4 ? 0 : 1))));
          ar = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[qa], wp[qb][0][s8], ar, 0, 0, 0);
          az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[qa], wp[qb][1][s8], az, 0, 0, 0);
          an = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[qa], wp[qb][2][s8], an, 0, 0, 0);
        }
      }
      const bool more = t + 1 < T;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        const long row = rowi[i];
        const float hb = an[i] + bhn;
        const float r = fast_sigmoid(xr[i] + ar[i]);
        const float z = fast_sigmoid(xz[i] + az[i]);
        const float n = fast_tanh(xn[i] + r * hb);
        const float hp = hpv[i];
        const float hn_new = (1.0f - z) * n + z * hp;
        if (FULL || rho0 + rl < a.NR) {
          a.hs[row * H + col] = hn_new;
          *reinterpret_cast<float4*>(a.gates + row * (4 * H) + 4 * col) = make_float4(r, z, n, hb);
          a.hprev[row * H + col] = hp;
        }
        const float nx = (more & (rf[i] != 0)) ? 0.f : hn_new;
        hf[rl * HP + col] = nx;
        __bf16 pc[NP];
        split_pieces<NP>(nx, pc);
#pragma unroll
        for (int q = 0; q < NP; ++q) hsp[nxt][q][rl * HB + col] = pc[q];
      }
    }
    __syncthreads();
  }
}

struct GruBwdArgs {
  const float* gates;     // [R][4H] as written by the forward scan: (r, z, n, hb) per hidden column
  const float* hprev;     // [R][H] from the forward
  const unsigned char* reset;
  const float* dhs;       // [R][H] dL/dh_t from the post-torso path
  const float* Wh;        // [H][3H] natural layout (used as "Wt" of dh_prev = dhh @ W_h^T)
  float* dg;              // [R][4H] gate pre-activation gradients (dn_in | dr | dz | dn_hid), see magpo.h
  float* slab_bhn;        // [grid][H]
  int T, A, NR;
};

template <bool FULL, int ROWS>
__global__ __launch_bounds__(256, 1) void k_gru_scan_bwd(GruBwdArgs a, int block0) {
  extern __shared__ __align__(16) float smem[];
  float* dht = smem;                 // [ROWS][HP]   dL/dh carried from step t+1 (already includes the direct z path)
  float* dhht = dht + ROWS * HP;     // [ROWS][G3P]  dhh of the current step
  unsigned char* rflag = reinterpret_cast<unsigned char*>(dhht + ROWS * G3P);   // [ROWS][T] reset flags of the block's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int rho0 = (block0 + blockIdx.x) * ROWS;
  const int col = 32 * wave + lr;
  // W_h (natural [H][3H]) as the B operand of dh_prev = dhh @ W_h^T: lane (col, h) holds k in [192 h, 192 h + 192)
  float4 wf[48];
#pragma unroll
  for (int u = 0; u < 48; ++u) wf[u] = *reinterpret_cast<const float4*>(a.Wh + (long)col * G3 + 192 * h + 4 * u);
  for (int i = tid; i < ROWS * HP; i += 256) dht[i] = 0.f;
  __shared__ long rbase[ROWS];
  if (tid < ROWS) rbase[tid] = tok_row(min(rho0 + tid, a.NR - 1), 0, a.T, a.A);   // invalid rows shadow the last valid one
  for (int i = tid; i < ROWS * a.T; i += 256) {
    const int rl = i / a.T, t = i - rl * a.T;
    const int rho = min(rho0 + rl, a.NR - 1);
    rflag[i] = a.reset[(long)(rho / a.A) * a.T + t];
  }
  __syncthreads();
  // ---- elementwise map: thread owns 4 consecutive columns c4..c4+3 of rows rl = (tid >> 5) + 8 k (k = 0..7): every global
  // access is a float4 (6 loads + 6 stores per row instead of 24 + 24 scalars; the scalar version kept >63 stores in flight,
  // so the in-order vmcnt throttled the whole phase)
  const int c4 = 4 * (tid & 31), rg = tid >> 5;
  float4 bacc4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = a.T - 1; t >= 0; --t) {
    {
      constexpr int NK = ROWS / 8;   // rows per thread
      float4 gr[NK], gz[NK], gn[NK], gh[NK], hp[NK], dh[NK];
      long rowv[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int rl = rg + 8 * k;
        const long row = rbase[rl] + (long)t * a.A;
        rowv[k] = row;
        const float* g = a.gates + row * (4 * H) + 4 * c4;   // (r, z, n, hb) of columns c4 .. c4 + 3: 64 contiguous bytes
        const float4 q0 = *reinterpret_cast<const float4*>(g), q1 = *reinterpret_cast<const float4*>(g + 4);
        const float4 q2 = *reinterpret_cast<const float4*>(g + 8), q3 = *reinterpret_cast<const float4*>(g + 12);
        gr[k] = make_float4(q0.x, q1.x, q2.x, q3.x); gz[k] = make_float4(q0.y, q1.y, q2.y, q3.y);
        gn[k] = make_float4(q0.z, q1.z, q2.z, q3.z); gh[k] = make_float4(q0.w, q1.w, q2.w, q3.w);
        hp[k] = *reinterpret_cast<const float4*>(a.hprev + row * H + c4);
        dh[k] = *reinterpret_cast<const float4*>(a.dhs + row * H + c4);
      }
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int rl = rg + 8 * k;
        const bool ok = FULL || rho0 + rl < a.NR;
        const bool rst = rflag[rl * a.T + t] != 0;
        const float4 dc = *reinterpret_cast<const float4*>(&dht[rl * HP + c4]);
        float4 o_r, o_z, o_an, o_hb, carry;
#define GRU_BWD_ELEM(X)                                                   \
        {                                                                 \
          const float r = gr[k].X, z = gz[k].X, n = gn[k].X, hb = gh[k].X; \
          const float dht_ = dh[k].X + dc.X;                              \
          const float dn = dht_ * (1.0f - z);                             \
          const float dz = dht_ * (hp[k].X - n);                          \
          const float dan = dn * (1.0f - n * n);                          \
          o_an.X = dan;                                                   \
          o_hb.X = dan * r;                                               \
          o_r.X = dan * hb * r * (1.0f - r);                              \
          o_z.X = dz * z * (1.0f - z);                                    \
          carry.X = rst ? 0.f : dht_ * z;                                 \
        }
        GRU_BWD_ELEM(x) GRU_BWD_ELEM(y) GRU_BWD_ELEM(z) GRU_BWD_ELEM(w)
#undef GRU_BWD_ELEM
        if (ok) {
          float* dx = a.dg + rowv[k] * (4 * H) + c4;   // (dr, dz are shared by the input and the hidden side: written once)
          *reinterpret_cast<float4*>(dx) = o_an; *reinterpret_cast<float4*>(dx + H) = o_r; *reinterpret_cast<float4*>(dx + 2 * H) = o_z;
          *reinterpret_cast<float4*>(dx + 3 * H) = o_hb;
          bacc4.x += o_hb.x; bacc4.y += o_hb.y; bacc4.z += o_hb.z; bacc4.w += o_hb.w;
        }
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rst || !ok) { o_r = z4; o_z = z4; o_hb = z4; }  // no gradient into the (zeroed) previous state
        if (!ok) carry = z4;
        *reinterpret_cast<float4*>(&dhht[rl * G3P + c4]) = o_r;
        *reinterpret_cast<float4*>(&dhht[rl * G3P + H + c4]) = o_z;
        *reinterpret_cast<float4*>(&dhht[rl * G3P + 2 * H + c4]) = o_hb;
        *reinterpret_cast<float4*>(&dht[rl * HP + c4]) = carry;  // direct path; the GEMM below adds dhh @ W_h^T
      }
    }
    __syncthreads();
    // ---- dh_prev += dhh[64][3H] @ W_h^T  -> [64][H]; wave w owns columns 32w.. for both row halves
#pragma unroll 1
    for (int wr = 0; wr < ROWS / 32; ++wr) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = dhht + (32 * wr + lr) * G3P + 192 * h;
#pragma unroll
      for (int u = 0; u < 48; ++u) {
        const float4 av = *reinterpret_cast<const float4*>(ap + 4 * u);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wf[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wf[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wf[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wf[u].w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        dht[rl * HP + col] += acc[i];
      }
    }
    __syncthreads();
  }
  // b_hn gradient: sum over the 8 row groups (threads with the same tid & 31)
  __shared__ float bsh[8][H];
  *reinterpret_cast<float4*>(&bsh[rg][c4]) = bacc4;
  __syncthreads();
  if (tid < H) {
    float sb = 0.f;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) sb += bsh[r8][tid];
    if (ROWS == 64) a.slab_bhn[(long)(block0 + blockIdx.x) * H + tid] = sb;
    else atomicAdd(&a.slab_bhn[(long)((block0 + blockIdx.x) >> 1) * H + tid], sb);   // zeroed by the launcher; exactly two addends per element
  }
}

// split-bf16 (x3) variant of the backward scan (see k_gru_scan_fwd_bf3): the per-step GEMM dh_prev += dhh W_h^T on
// v_mfma_f32_32x32x16_bf16; the element-wise phase writes dhh of the step to LDS as two bf16 tiles.
constexpr int G3B = G3 + 8;   // bf16 pitch of the dhh tile

template <bool FULL>
__global__ __launch_bounds__(256, 1) void k_gru_scan_bwd_bf3(GruBwdArgs a, int block0) {
  extern __shared__ __align__(16) float smem[];
  float* dht = smem;                                                 // [64][HP] fp32 dL/dh carried from step t+1
  __bf16* dsp = reinterpret_cast<__bf16*>(dht + 64 * HP);            // [2 (hi | lo)][64][G3B]
  unsigned char* rflag = reinterpret_cast<unsigned char*>(dsp + 2 * 64 * G3B);   // [64][T]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int rho0 = (block0 + blockIdx.x) * 64;
  const int col = 32 * wave + lr;
  bf16x8 whi[24], wlo[24];   // B[k][col] = W_h[col][k], k = 192 h + 8 s + j
#pragma unroll
  for (int s8 = 0; s8 < 24; ++s8) {
    const float* w = a.Wh + (long)col * G3 + 192 * h + 8 * s8;
    const float4 w0 = *reinterpret_cast<const float4*>(w), w1 = *reinterpret_cast<const float4*>(w + 4);
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) { __bf16 hi, lo; split_bf16(wv[j], hi, lo); whi[s8][j] = hi; wlo[s8][j] = lo; }
  }
  for (int i = tid; i < 64 * HP; i += 256) dht[i] = 0.f;
  __shared__ long rbase[64];
  if (tid < 64) rbase[tid] = tok_row(min(rho0 + tid, a.NR - 1), 0, a.T, a.A);
  for (int i = tid; i < 64 * a.T; i += 256) {
    const int rl = i / a.T, t = i - rl * a.T;
    const int rho = min(rho0 + rl, a.NR - 1);
    rflag[i] = a.reset[(long)(rho / a.A) * a.T + t];
  }
  __syncthreads();
  const int c4 = 4 * (tid & 31), rg = tid >> 5;
  float4 bacc4 = make_float4(0.f, 0.f, 0.f, 0.f);
  __bf16* dhi = dsp;
  __bf16* dlo = dsp + 64 * G3B;
  for (int t = a.T - 1; t >= 0; --t) {
    {
      float4 gr[8], gz[8], gn[8], gh[8], hp[8], dh[8];
      long rowv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int rl = rg + 8 * k;
        const long row = rbase[rl] + (long)t * a.A;
        rowv[k] = row;
        const float* g = a.gates + row * (4 * H) + 4 * c4;
        const float4 q0 = *reinterpret_cast<const float4*>(g), q1 = *reinterpret_cast<const float4*>(g + 4);
        const float4 q2 = *reinterpret_cast<const float4*>(g + 8), q3 = *reinterpret_cast<const float4*>(g + 12);
        gr[k] = make_float4(q0.x, q1.x, q2.x, q3.x); gz[k] = make_float4(q0.y, q1.y, q2.y, q3.y);
        gn[k] = make_float4(q0.z, q1.z, q2.z, q3.z); gh[k] = make_float4(q0.w, q1.w, q2.w, q3.w);
        hp[k] = *reinterpret_cast<const float4*>(a.hprev + row * H + c4);
        dh[k] = *reinterpret_cast<const float4*>(a.dhs + row * H + c4);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int rl = rg + 8 * k;
        const bool ok = FULL || rho0 + rl < a.NR;
        const bool rst = rflag[rl * a.T + t] != 0;
        const float4 dc = *reinterpret_cast<const float4*>(&dht[rl * HP + c4]);
        float4 o_r, o_z, o_an, o_hb, carry;
#define GRU_BWD_ELEM(X)                                                   \
        {                                                                 \
          const float r = gr[k].X, z = gz[k].X, n = gn[k].X, hb = gh[k].X; \
          const float dht_ = dh[k].X + dc.X;                              \
          const float dn = dht_ * (1.0f - z);                             \
          const float dz = dht_ * (hp[k].X - n);                          \
          const float dan = dn * (1.0f - n * n);                          \
          o_an.X = dan;                                                   \
          o_hb.X = dan * r;                                               \
          o_r.X = dan * hb * r * (1.0f - r);                              \
          o_z.X = dz * z * (1.0f - z);                                    \
          carry.X = rst ? 0.f : dht_ * z;                                 \
        }
        GRU_BWD_ELEM(x) GRU_BWD_ELEM(y) GRU_BWD_ELEM(z) GRU_BWD_ELEM(w)
#undef GRU_BWD_ELEM
        if (ok) {
          float* dx = a.dg + rowv[k] * (4 * H) + c4;   // (dr, dz are shared by the input and the hidden side: written once)
          *reinterpret_cast<float4*>(dx) = o_an; *reinterpret_cast<float4*>(dx + H) = o_r; *reinterpret_cast<float4*>(dx + 2 * H) = o_z;
          *reinterpret_cast<float4*>(dx + 3 * H) = o_hb;
          bacc4.x += o_hb.x; bacc4.y += o_hb.y; bacc4.z += o_hb.z; bacc4.w += o_hb.w;
        }
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rst || !ok) { o_r = z4; o_z = z4; o_hb = z4; }
        if (!ok) carry = z4;
        const float v12[12] = {o_r.x, o_r.y, o_r.z, o_r.w, o_z.x, o_z.y, o_z.z, o_z.w, o_hb.x, o_hb.y, o_hb.z, o_hb.w};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            __bf16 hi, lo;
            split_bf16(v12[4 * q + j], hi, lo);
            dhi[rl * G3B + q * H + c4 + j] = hi;
            dlo[rl * G3B + q * H + c4 + j] = lo;
          }
        *reinterpret_cast<float4*>(&dht[rl * HP + c4]) = carry;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int wr = 0; wr < 2; ++wr) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const int aoff = (32 * wr + lr) * G3B + 192 * h;
#pragma unroll
      for (int s8 = 0; s8 < 24; ++s8) {
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(dhi + aoff + 8 * s8);
        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(dlo + aoff + 8 * s8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, whi[s8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wlo[s8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, whi[s8], acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
        dht[rl * HP + col] += acc[i];
      }
    }
    __syncthreads();
  }
  __shared__ float bsh[8][H];
  *reinterpret_cast<float4*>(&bsh[rg][c4]) = bacc4;
  __syncthreads();
  if (tid < H) {
    float sb = 0.f;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) sb += bsh[r8][tid];
    a.slab_bhn[(long)(block0 + blockIdx.x) * H + tid] = sb;
  }
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
  if (F <= 8) {   // all loads of the row issued together (a run-time loop makes them dependent round trips)
    float xv[8];
    float4 w[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      xv[f] = f < F ? x[f] : 0.f;
      w[f] = f < F ? *reinterpret_cast<const float4*>(W + (long)f * N + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) { acc.x += xv[f] * w[f].x; acc.y += xv[f] * w[f].y; acc.z += xv[f] * w[f].z; acc.w += xv[f] * w[f].w; }
  } else {
    for (int f = 0; f < F; ++f) {
      const float xv = x[f];
      const float4 w = *reinterpret_cast<const float4*>(W + (long)f * N + c4);
      acc.x += xv * w.x; acc.y += xv * w.y; acc.z += xv * w.z; acc.w += xv * w.w;
    }
  }
  if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
  *reinterpret_cast<float4*>(Y + row * ldy + c4) = acc;
}

// N = 128, F <= 8: 32 rows per block iteration.  The weight rows stay in registers, the observation tile arrives as one
// coalesced load per thread through double-buffered LDS (fetched one iteration ahead), so a 16-byte store costs a
// quarter of a memory request instead of the ~16 of the one-row kernel.
__global__ __launch_bounds__(256) void k_small_linear128(const float* __restrict__ X, int ldx, int F, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ Y, int ldy, long R, int relu) {
  __shared__ float xs[2][256];
  const int t = threadIdx.x, c4 = 4 * (t & 31), slot = t >> 5;
  float4 w[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const float4 v = *reinterpret_cast<const float4*>(W + (f < F ? f : 0) * 128 + c4);
    const float m = f < F ? 1.f : 0.f;
    w[f] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
  }
  const float4 bias = *reinterpret_cast<const float4*>(b + c4);
  const long stride = (long)gridDim.x * 32;
  long base = (long)blockIdx.x * 32;
  if (base >= R) return;
  const int fx = (t & 7) < F ? (t & 7) : 0;
  auto fetch = [&](long bs) {
    long r = bs + (t >> 3);
    r = r < R ? r : R - 1;
    return X[r * ldx + fx];
  };
  float xn = fetch(base);
  for (int it = 0; base < R; base += stride, it ^= 1) {
    xs[it][t] = xn;
    __syncthreads();
    if (base + stride < R) xn = fetch(base + stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lrow = slot + 8 * j;
      const float4 xa = *reinterpret_cast<const float4*>(&xs[it][lrow * 8]);
      const float4 xb = *reinterpret_cast<const float4*>(&xs[it][lrow * 8 + 4]);
      const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
      float4 acc = bias;
#pragma unroll
      for (int f = 0; f < 8; ++f) { acc.x += xv[f] * w[f].x; acc.y += xv[f] * w[f].y; acc.z += xv[f] * w[f].z; acc.w += xv[f] * w[f].w; }
      if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
      const long row = base + lrow;
      if (row < R) *reinterpret_cast<float4*>(Y + row * ldy + c4) = acc;
    }
  }
}

}  // namespace magpo

using namespace magpo;

// Per-call tuning arguments (no library state):
//   split_bf16  the TRAINING scans (T > 1 with all save buffers) on 0 = fp32 MFMA, 1 = bf16 pairs (16 mantissa bits, 3 products, forward and
//               backward), 2 = bf16 triples (24 mantissa bits = fp32 operands, 6 products; forward scan only, the backward stays on fp32 MFMA:
//               it is bound by its gate / gradient traffic);
//   block_rows  recurrent rows per workgroup of the fp32 scans: 0 = by size (32 when 64-row blocks would occupy at most half of the compute
//               units: the scan is a latency chain of T steps whose step time follows the block's rows), or 32 / 64 forced.
static int check_gru_tuning(int split_bf16, int block_rows) {
  if (split_bf16 < 0 || split_bf16 > 2 || (block_rows != 0 && block_rows != 32 && block_rows != 64)) {
    set_error("gru: split_bf16 must be 0 / 1 / 2, block_rows 0 / 32 / 64");
    return MAGPO_EINVAL;
  }
  return MAGPO_OK;
}
static bool gru_half_blocks(int NR, int block_rows) {
  if (block_rows) return block_rows == 32;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  return (NR + 63) / 64 <= ncu / 2;
}

extern "C" int magpo_gru_scan_fwd(const float* xi, const float* Wht, const float* b_hn, const float* h0, const int* h0_idx,
                                  const unsigned char* reset, float* hs, float* gates, float* hprev, int nseq, int T, int A,
                                  const int* xi_cls, int split_bf16, int block_rows, hipStream_t st) {
  GruArgs a{xi, Wht, b_hn, h0, h0_idx, reset, hs, gates, hprev, T, A, nseq * A, 0, nullptr, xi_cls};
  if (int e = check_gru_tuning(split_bf16, block_rows)) return e;
  if (a.NR <= 0) return MAGPO_OK;
  const size_t lds = (size_t)64 * T * (xi_cls ? 5 : 1);   // reset flags (+ xi class rows) of the block's rows
  if (lds > (xi_cls ? 80 : 24) * 1024) { set_error("magpo_gru_scan_fwd: T too large for the LDS tables"); return MAGPO_EINVAL; }
  const int nfull = a.NR / 64;
  const bool split = split_bf16 != 0;
  if (hs && gates && hprev && split && T > 1) {   // training scan on bf16 MFMA with split operands (see k_gru_scan_fwd_bf3)
    static size_t lf_set = 0;   // (memoised device attribute: idempotent)
    if (lds > lf_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_fwd_bf3<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_fwd_bf3<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_fwd_bf3<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_fwd_bf3<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      lf_set = lds;
    }
    if (split_bf16 == 2) {   // 32-row blocks
      const int nf = a.NR / 32;
      if (nf) hipLaunchKernelGGL((k_gru_scan_fwd_bf3<true, 3>), dim3(nf), dim3(256), lds / 2, st, a, 0);
      if (a.NR % 32) hipLaunchKernelGGL((k_gru_scan_fwd_bf3<false, 3>), dim3(1), dim3(256), lds / 2, st, a, nf);
    } else {
      if (nfull) hipLaunchKernelGGL((k_gru_scan_fwd_bf3<true, 2>), dim3(nfull), dim3(256), lds, st, a, 0);
      if (a.NR % 64) hipLaunchKernelGGL((k_gru_scan_fwd_bf3<false, 2>), dim3(1), dim3(256), lds, st, a, nfull);
    }
  } else if (hs && gates && hprev) {
    if (gru_half_blocks(a.NR, block_rows)) {
      const int nf = a.NR / 32;
      if (nf) hipLaunchKernelGGL((k_gru_scan_fwd<true, 0, 32>), dim3(nf), dim3(256), lds, st, a, 0);
      if (a.NR % 32) hipLaunchKernelGGL((k_gru_scan_fwd<false, 0, 32>), dim3(1), dim3(256), lds, st, a, nf);
    } else {
      if (nfull) hipLaunchKernelGGL((k_gru_scan_fwd<true, 0, 64>), dim3(nfull), dim3(256), lds, st, a, 0);
      if (a.NR % 64) hipLaunchKernelGGL((k_gru_scan_fwd<false, 0, 64>), dim3(1), dim3(256), lds, st, a, nfull);
    }
  } else {
    if (nfull) hipLaunchKernelGGL((k_gru_scan_fwd<true, 2, 64>), dim3(nfull), dim3(256), lds, st, a, 0);
    if (a.NR % 64) hipLaunchKernelGGL((k_gru_scan_fwd<false, 2, 64>), dim3(1), dim3(256), lds, st, a, nfull);
  }
  return check_launch("magpo_gru_scan_fwd");
}

// Hidden-state carry over a time-major trajectory (the rollout's obs / done buffers): rows (t, env, agent), reset [T][nenv];
// only the state after the last step is written.  The carry is a pure function of (obs, done), so the rollout computes it
// once for all T steps instead of once per env step (ScannedRNN semantics, base.py:121-149).
extern "C" int magpo_gru_carry(const float* xi, const float* Wht, const float* b_hn, const float* h0, const unsigned char* reset_tm,
                               float* h_last, int nenv, int T, int A, const int* xi_cls, int block_rows, hipStream_t st) {
  if (int e = check_gru_tuning(0, block_rows)) return e;
  GruArgs a{xi, Wht, b_hn, h0, nullptr, reset_tm, nullptr, nullptr, nullptr, T, A, nenv * A, 1, h_last, xi_cls};
  if (a.NR <= 0 || T <= 0) return MAGPO_OK;
  const size_t lds = (size_t)64 * T * (xi_cls ? 5 : 1);
  if (lds > (xi_cls ? 80 : 24) * 1024) { set_error("magpo_gru_carry: T too large for the LDS tables"); return MAGPO_EINVAL; }
  if (gru_half_blocks(a.NR, block_rows)) {
    const int nf = a.NR / 32;
    if (nf) hipLaunchKernelGGL((k_gru_scan_fwd<true, 1, 32>), dim3(nf), dim3(256), lds, st, a, 0);
    if (a.NR % 32) hipLaunchKernelGGL((k_gru_scan_fwd<false, 1, 32>), dim3(1), dim3(256), lds, st, a, nf);
  } else {
    const int nfull = a.NR / 64;
    if (nfull) hipLaunchKernelGGL((k_gru_scan_fwd<true, 1, 64>), dim3(nfull), dim3(256), lds, st, a, 0);
    if (a.NR % 64) hipLaunchKernelGGL((k_gru_scan_fwd<false, 1, 64>), dim3(1), dim3(256), lds, st, a, nfull);
  }
  return check_launch("magpo_gru_carry");
}

// slab_bhn: [ceil(NR/64)][128]
extern "C" int magpo_gru_scan_bwd(const float* gates, const float* hprev, const unsigned char* reset, const float* dhs,
                                  const float* Wh, float* dg, float* slab_bhn, int nseq, int T, int A,
                                  int split_bf16, int block_rows, hipStream_t st) {
  if (int e = check_gru_tuning(split_bf16, block_rows)) return e;
  GruBwdArgs a{gates, hprev, reset, dhs, Wh, dg, slab_bhn, T, A, nseq * A};
  if (a.NR <= 0) return MAGPO_OK;
  const size_t lds = (size_t)(64 * HP + 64 * G3P) * sizeof(float) + (size_t)64 * T;
  if (lds > 150 * 1024) { set_error("magpo_gru_scan_bwd: T too large for the LDS flag table"); return MAGPO_EINVAL; }
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_bwd<true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_bwd<false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_bwd<true, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_bwd<false, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_set = lds;
  }
  const int nfull = a.NR / 64;
  if (split_bf16 == 1) {   // bf16 pairs (k_gru_scan_bwd_bf3): fp32 dht + two bf16 dhh tiles + flags
    const size_t ldb = (size_t)64 * HP * sizeof(float) + (size_t)2 * 64 * G3B * sizeof(__bf16) + (size_t)64 * T;
    static size_t ldb_set = 0;
    if (ldb > ldb_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_bwd_bf3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldb);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_scan_bwd_bf3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldb);
      ldb_set = ldb;
    }
    if (nfull) hipLaunchKernelGGL(k_gru_scan_bwd_bf3<true>, dim3(nfull), dim3(256), ldb, st, a, 0);
    if (a.NR % 64) hipLaunchKernelGGL(k_gru_scan_bwd_bf3<false>, dim3(1), dim3(256), ldb, st, a, nfull);
    return check_launch("magpo_gru_scan_bwd");
  }
  if (gru_half_blocks(a.NR, block_rows)) {   // two 32-row blocks add into one slab row (two addends: the sum does not depend on their order)
    if (hipMemsetAsync(slab_bhn, 0, sizeof(float) * H * (size_t)((a.NR + 63) / 64), st) != hipSuccess) { set_error("magpo_gru_scan_bwd: memset failed"); return MAGPO_ELAUNCH; }
    const int nf = a.NR / 32;
    if (nf) hipLaunchKernelGGL((k_gru_scan_bwd<true, 32>), dim3(nf), dim3(256), lds, st, a, 0);
    if (a.NR % 32) hipLaunchKernelGGL((k_gru_scan_bwd<false, 32>), dim3(1), dim3(256), lds, st, a, nf);
    return check_launch("magpo_gru_scan_bwd");
  }
  if (nfull) hipLaunchKernelGGL((k_gru_scan_bwd<true, 64>), dim3(nfull), dim3(256), lds, st, a, 0);
  if (a.NR % 64) hipLaunchKernelGGL((k_gru_scan_bwd<false, 64>), dim3(1), dim3(256), lds, st, a, nfull);
  return check_launch("magpo_gru_scan_bwd");
}

extern "C" int magpo_small_linear(const float* X, int ldx, int F, const float* W, const float* b, float* Y, int ldy, int N,
                                  long R, int relu, hipStream_t st) {
  if (N & 3) { set_error("magpo_small_linear: N must be a multiple of 4"); return MAGPO_EINVAL; }
  if (N == 128 && F <= 8 && R >= 4096) {
    const long nb = (R + 31) / 32;
    static const unsigned cap = resident_grid(k_small_linear128, 256, 1L << 30);
    hipLaunchKernelGGL(k_small_linear128, dim3(nb < (long)cap ? (unsigned)nb : cap), dim3(256), 0, st, X, ldx, F, W, b, Y, ldy, R, relu);
    return check_launch("magpo_small_linear");
  }
  long n = R * (N / 4);
  hipLaunchKernelGGL(k_small_linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, ldx, F, W, b, Y, ldy, N, R, relu);
  return check_launch("magpo_small_linear");
}

#ifdef MAGPO_GRU_PROF
extern "C" int magpo_debug_gru_prof(unsigned long long* out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(magpo::g_gru_prof), sizeof(unsigned long long) * 8) != hipSuccess) return MAGPO_ELAUNCH;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(magpo::g_gru_prof), z, sizeof(z)) != hipSuccess) return MAGPO_ELAUNCH; }
  return MAGPO_OK;
}
#endif
