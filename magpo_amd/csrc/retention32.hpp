// Chunkwise retention on 32-token chunks (included by retention.hip; same math and arguments as k_ret_chunk_fwd / _bwd).
//
// Why a second tiling: on gfx950 MFMA time and VALU time add up per SIMD, so the only thing a second resident
// workgroup can hide is what the 64-token kernels spend in LDS round trips, barriers and vmcnt waits -- ~14 K of the backward's
// 36 K cycles per chunk -- and the 64-token backward cannot have one: its eight 64x64 tiles take 139 KB of LDS.  With 32-token
// chunks the token tiles are 32x64 (8.7 KB), the chunk-entry state of the backward is read as MFMA fragments straight from
// global memory, and a workgroup needs 67 KB at 16 chunks per sequence: two fit a CU (forward: 54 KB, three).  The causally /
// episode-masked half of the intra-chunk products shrinks with the chunk as well: 0.72 x the MFMA work of the 64-token kernels per
// token (forward 0.75 x).  Measured at the bench minibatch: backward 3.56 -> 2.5 ms per launch, forward 1.75 -> 1.45-1.5 ms.
//
// Tiles are 16x16 (v_mfma_f32_16x16x4_f32, the same flop rate as 32x32x2): lane l = (idx = l & 15, kq = l >> 4) supplies
// A[m0 + idx][k] and B[k][n0 + idx] for the k-slot k = kb + 4 kq + c of step c (c = 0..3 of a float4), and holds
// D[m0 + 4 kq + i][n0 + idx], i = 0..3.  Four waves: 32x32 outputs = one tile per wave; 32x64 outputs = column tile `wave`,
// both row tiles; 64x64 outputs (states) = column tile `wave`, four row tiles (the B fragment is shared by the row tiles).
#pragma once

namespace magpo {

constexpr int TP = 64 + LDP;     // pitch of the 64-column tiles (= TL)
constexpr int PP = 32 + LDP;     // pitch of the 32x32 score tiles
constexpr int MAXC32 = 32;       // chunks per sequence whose bookkeeping is built up front

struct ChunkMeta32 {
  signed char cnt[32];   // per token: # dones among chunk timesteps [0..lt]   (invalid tokens: -1)
  signed char lt[32];    // per token: chunk-local timestep (invalid tokens: -1)
  float beta[32];        // incoming-state weight per token
  float eta[32];         // outgoing-state weight per token
  float gamma;           // state carry factor
  float pad_[3];
};
struct SeqMeta32 {
  float kpow[36];        // kappa^p, p = 0..33
  ChunkMeta32 ch[MAXC32];
};

__device__ __forceinline__ void build_meta32(const float* __restrict__ kpow, ChunkMeta32* __restrict__ ch, const unsigned char* __restrict__ dones,
                                             int T, int Lt, int A, int c0, int count) {
  const int tid = threadIdx.x;
  for (int x = tid; x < count * 32; x += 256) {
    const int cc = x >> 5, tok = x & 31;
    const int t0 = (c0 + cc) * Lt, ltc = min(Lt, T - t0);
    const int lt = tok / A;
    int c = -1, l = -1;
    if (lt < ltc) {
      c = 0;
      l = lt;
      for (int s2 = 0; s2 <= lt; ++s2) c += dones[t0 + s2] ? 1 : 0;
    }
    ch[cc].lt[tok] = (signed char)l;
    ch[cc].cnt[tok] = (signed char)c;
  }
  __syncthreads();
  for (int x = tid; x < count * 32; x += 256) {
    const int cc = x >> 5, tok = x & 31;
    const int t0 = (c0 + cc) * Lt, ltc = min(Lt, T - t0);
    ChunkMeta32& m = ch[cc];
    const int lt = m.lt[tok];
    const int ctot = m.cnt[(ltc - 1) * A];
    float b = 0.f, e = 0.f;
    if (lt >= 0) {
      b = (m.cnt[tok] == 0) ? kpow[lt + 1] : 0.f;
      e = (m.cnt[tok] == ctot) ? kpow[ltc - 1 - lt] : 0.f;
    }
    m.beta[tok] = b;
    m.eta[tok] = e;
    if (tok == 0) m.gamma = (ctot == 0) ? kpow[ltc] : 0.f;
  }
  __syncthreads();
}

// decay weight of the (query token i, key token j) pair of a chunk (retention.py:117-187)
__device__ __forceinline__ float w32(const float* __restrict__ kpow, const ChunkMeta32& m, int i, int j, int masked) {
  const int li = m.lt[i], lj = m.lt[j];
  const bool on = li >= 0 && lj >= 0 && li >= lj && m.cnt[i] == m.cnt[j] && !(masked && j > i);
  const float kp = kpow[on ? li - lj : 0];
  return on ? kp : 0.f;
}

// ---- 32 x 64 token tiles: global -> registers (one chunk ahead) -> LDS -------------------------------------------------------
struct Tile32 { float4 v[2]; };
struct RowIdx32 { int r[2]; };
__device__ __forceinline__ void fetch32(Tile32& t, const float* __restrict__ src, long ld, int nvalid, int w4) {
  const int c4 = min((int)(threadIdx.x & 15), w4 - 1);
#pragma unroll
  for (int j = 0; j < 2; ++j) t.v[j] = *reinterpret_cast<const float4*>(src + (long)min((int)(threadIdx.x >> 4) + 16 * j, nvalid - 1) * ld + 4 * c4);
}
__device__ __forceinline__ void fetch_idx32(RowIdx32& x, const int* __restrict__ rows, long row0, int nvalid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) x.r[j] = rows[row0 + min((int)(threadIdx.x >> 4) + 16 * j, nvalid - 1)];
}
__device__ __forceinline__ void fetch32_rows(Tile32& t, const float* __restrict__ tab, long ld, const RowIdx32& x, int w4) {
  const int c4 = min((int)(threadIdx.x & 15), w4 - 1);
#pragma unroll
  for (int j = 0; j < 2; ++j) t.v[j] = *reinterpret_cast<const float4*>(tab + (long)x.r[j] * ld + 4 * c4);
}
__device__ __forceinline__ void stash32(float* __restrict__ dst, const Tile32& t, int nvalid, int w4) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (threadIdx.x >> 4) + 16 * j, c4 = threadIdx.x & 15;
    const bool ok = r < nvalid && c4 < w4;
    *reinterpret_cast<float4*>(&dst[r * TP + 4 * c4]) = ok ? t.v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---- acc[r] (+)= A B on 16x16 tiles: row tile r covers A rows am0 + 16 r .. + 15, the column tile B columns bn0 .. bn0 + 15 ------
// AROW: A[m][k] = At[m * apitch + k] (k contiguous, float4 reads), else A[m][k] = At[k * apitch + m] * (ascale ? ascale[k] : 1).
// BROW: B[k][n] = Bt[n * bpitch + k], else B[k][n] = Bt[k * bpitch + n].
template <bool AROW, bool BROW, int KK, int NR>
__device__ __forceinline__ void mma16(f32x4 (&acc)[NR], const float* __restrict__ At, int apitch, int am0, const float* __restrict__ Bt,
                                      int bpitch, int bn0, int idx, int kq, const float* __restrict__ ascale = nullptr) {
#pragma unroll
  for (int kb = 0; kb < KK; kb += 16) {
    const int k0 = kb + 4 * kq;
    float b[4];
    if (BROW) {
      const float4 b4 = *reinterpret_cast<const float4*>(Bt + (bn0 + idx) * bpitch + k0);
      b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = Bt[(k0 + c) * bpitch + bn0 + idx];
    }
    float sc[4] = {1.f, 1.f, 1.f, 1.f};
    if (!AROW && ascale) {
      const float4 s4 = *reinterpret_cast<const float4*>(ascale + k0);
      sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
    }
    float av[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (AROW) {
        const float4 a4 = *reinterpret_cast<const float4*>(At + (am0 + 16 * r + idx) * apitch + k0);
        av[r][0] = a4.x; av[r][1] = a4.y; av[r][2] = a4.z; av[r][3] = a4.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) av[r][c] = At[(k0 + c) * apitch + am0 + 16 * r + idx] * sc[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)   // consecutive MFMAs go to different accumulators (a dependent one waits for its predecessor's passes)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][c], b[c], acc[r], 0, 0, 0);
  }
}
// same with the B fragment already in registers (row form: breg[kb / 16] = B^T[bn0 + idx][kb + 4 kq .. + 3])
template <int KK, int NR>
__device__ __forceinline__ void mma16_breg(f32x4 (&acc)[NR], const float* __restrict__ At, int apitch, int am0, const float4 (&breg)[KK / 16], int idx,
                                           int kq) {
#pragma unroll
  for (int kb = 0; kb < KK; kb += 16) {
    const float4 b4 = breg[kb / 16];
    float4 a4[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) a4[r] = *reinterpret_cast<const float4*>(At + (am0 + 16 * r + idx) * apitch + kb + 4 * kq);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r].x, b4.x, acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r].y, b4.y, acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r].z, b4.z, acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r].w, b4.w, acc[r], 0, 0, 0);
  }
}

// rows m0 + 4 kq + i (i = 0..3) of column n of a [32 x 64] result to out[(r0 + m) * ld + n]; rows >= nvalid / columns >= hs are dropped
__device__ __forceinline__ void store16(float* __restrict__ out, long r0, long ld, const f32x4& v, int m0, int n, int kq, int nvalid, int hs) {
  if (n < hs) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 4 * kq + i;
      if (m < nvalid) out[(r0 + m) * ld + n] = v[i];
    }
  }
}

// LDS per workgroup: the tiles + the bookkeeping of the chunks that are built up front (all of them when the sequence has at most MAXC32,
// else one at a time): 53.6 KB for the forward at 16 chunks (three workgroups per CU), 67 KB for the backward (two).
__host__ __device__ inline size_t ret32_meta_bytes(int nch) { return sizeof(float) * 36 + sizeof(ChunkMeta32) * (size_t)(nch <= MAXC32 ? nch : 1); }
__host__ __device__ inline size_t ret32_fwd_lds(int nch) { return (size_t)(3 * 32 * TP + 64 * TP + 32 * PP) * sizeof(float) + ret32_meta_bytes(nch); }
__host__ __device__ inline size_t ret32_bwd_lds(int nch) { return (size_t)(4 * 32 * TP + 64 * TP + 2 * 32 * PP) * sizeof(float) + ret32_meta_bytes(nch); }

__global__ __launch_bounds__(256, 3) void k_ret32_fwd(RetArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;
  float* Ks = Qs + 32 * TP;
  float* Vs = Ks + 32 * TP;
  float* Ss = Vs + 32 * TP;      // [64][TP] carried state
  float* Ps = Ss + 64 * TP;      // [32][PP] masked scores
  SeqMeta32& sm = *reinterpret_cast<SeqMeta32*>(Ps + 32 * PP);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, kq = lane >> 4;
  const int seq = blockIdx.x;
  const int Lt = 32 / a.A, L = Lt * a.A;
  const int nch = (a.T + Lt - 1) / Lt;
  const long row_base = (long)seq * a.T * a.A;
  const float* s0 = a.s0 ? a.s0 + (long)(a.seq_env ? a.seq_env[seq] : seq) * 4096 : nullptr;
  const int w4 = a.hs >> 2;
  load_state(Ss, s0);
  const bool pre = nch <= MAXC32;
  if (tid < 36) sm.kpow[tid] = powf(a.kappa, (float)tid);
  __syncthreads();
  if (pre) build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, 0, nch);
  Tile32 pq, pk, pv;
  RowIdx32 ri;
  const bool by_rows = a.rows != nullptr;
#define R32_FETCH(T_, P_, LD_, ROW0_, NV_) \
  do { if (by_rows) fetch32_rows(T_, a.P_, a.LD_, ri, w4); else fetch32(T_, a.P_ + (ROW0_) * a.LD_, a.LD_, NV_, w4); } while (0)
  {
    const int nv0 = min(Lt, a.T) * a.A;
    if (by_rows) fetch_idx32(ri, a.rows, row_base, nv0);
    R32_FETCH(pq, q, ldq, row_base, nv0);
    R32_FETCH(pk, k, ldk, row_base, nv0);
    R32_FETCH(pv, v, ldv, row_base, nv0);
    if (by_rows) {
      const int c1 = min(1, nch - 1);
      fetch_idx32(ri, a.rows, row_base + (long)c1 * L, min(Lt, a.T - c1 * Lt) * a.A);
    }
  }
  const int tr = wave >> 1, tc = wave & 1;   // this wave's tile of a 32x32 result
  const int n64 = 16 * wave + idx;           // this lane's column of a 64-column result
  for (int c = 0; c < nch; ++c) {
    const int t0 = c * Lt;
    const int nvalid = min(Lt, a.T - t0) * a.A;
    const long r0 = row_base + (long)c * L;
    __syncthreads();   // previous chunk finished with Qs / Ks / Vs / Ps, Ss updated
    stash32(Qs, pq, nvalid, w4);
    stash32(Ks, pk, nvalid, w4);
    stash32(Vs, pv, nvalid, w4);
    if (!pre) build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, c, 1);   // contains barriers
    else __syncthreads();
    const ChunkMeta32& meta = sm.ch[pre ? c : 0];
    const bool more = c + 1 < nch;
    const int nvn = more ? min(Lt, a.T - (t0 + Lt)) * a.A : nvalid;
    const long rn = more ? r0 + L : r0;
    R32_FETCH(pq, q, ldq, rn, nvn);
    if (a.states) store_state(a.states + ((long)seq * nch + c) * 4096, Ss);
    // scores (one 16x16 tile per wave) and Q S (column tile `wave`, both row tiles)
    f32x4 sc[1] = {{0.f, 0.f, 0.f, 0.f}};
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma16<true, true, 64, 1>(sc, Qs, TP, 16 * tr, Ks, TP, 16 * tc, idx, kq);
    mma16<true, false, 64, 2>(o, Qs, TP, 0, Ss, TP, 16 * wave, idx, kq);
    {
      const int j = 16 * tc + idx;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * tr + 4 * kq + i;
        Ps[m * PP + j] = sc[0][i] * w32(sm.kpow, meta, m, j, a.masked);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[r][i] *= meta.beta[16 * r + 4 * kq + i];
    }
    __syncthreads();
    R32_FETCH(pk, k, ldk, rn, nvn);
    mma16<true, false, 32, 2>(o, Ps, PP, 0, Vs, TP, 16 * wave, idx, kq);      // P V
#pragma unroll
    for (int r = 0; r < 2; ++r) store16(a.r, r0, a.ldr, o[r], 16 * r, n64, kq, nvalid, a.hs);
    R32_FETCH(pv, v, ldv, rn, nvn);
    if (by_rows) {
      const int c2 = min(c + 2, nch - 1);
      fetch_idx32(ri, a.rows, row_base + (long)c2 * L, min(Lt, a.T - c2 * Lt) * a.A);
    }
    // state update  S <- gamma S + (eta K)^T V   (column tile `wave`, four row tiles; each lane rewrites the elements it read)
    f32x4 sn[4];
    const float gm = meta.gamma;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) sn[r][i] = gm * Ss[(16 * r + 4 * kq + i) * TP + n64];
    mma16<false, false, 32, 4>(sn, Ks, TP, 0, Vs, TP, 16 * wave, idx, kq, meta.eta);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) Ss[(16 * r + 4 * kq + i) * TP + n64] = sn[r][i];
  }
  if (a.s_final) {
    __syncthreads();
    store_state(a.s_final + (long)seq * 4096, Ss);
  }
}

__global__ __launch_bounds__(256, 2) void k_ret32_bwd(RetBwdArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;
  float* Ks = Qs + 32 * TP;
  float* Vs = Ks + 32 * TP;
  float* Ds = Vs + 32 * TP;      // dO
  float* Gs = Ds + 32 * TP;      // [64][TP] dL/dS_{c+1}
  float* Ps = Gs + 64 * TP;      // [32][PP]
  float* dPs = Ps + 32 * PP;
  SeqMeta32& sm = *reinterpret_cast<SeqMeta32*>(dPs + 32 * PP);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, kq = lane >> 4;
  const int seq = blockIdx.x;
  const int Lt = 32 / a.A, L = Lt * a.A;
  const int nch = (a.T + Lt - 1) / Lt;
  const long row_base = (long)seq * a.T * a.A;
  const int w4 = a.hs >> 2;
  load_state(Gs, nullptr);
  const bool pre = nch <= MAXC32;
  if (tid < 36) sm.kpow[tid] = powf(a.kappa, (float)tid);
  __syncthreads();
  if (pre) build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, 0, nch);
  Tile32 pq, pk, pv, pd;
  RowIdx32 ri;
  const bool by_rows = a.rows != nullptr;
  {
    const int cl = nch - 1;
    const int nvl = min(Lt, a.T - cl * Lt) * a.A;
    const long rl = row_base + (long)cl * L;
    if (by_rows) fetch_idx32(ri, a.rows, rl, nvl);
    R32_FETCH(pq, q, ldq, rl, nvl);
    R32_FETCH(pk, k, ldk, rl, nvl);
    R32_FETCH(pv, v, ldv, rl, nvl);
    fetch32(pd, a.dr + rl * a.lddr, a.lddr, nvl, w4);
    if (by_rows) {
      const int cp = max(cl - 1, 0);
      fetch_idx32(ri, a.rows, row_base + (long)cp * L, min(Lt, a.T - cp * Lt) * a.A);
    }
  }
  const int tr = wave >> 1, tc = wave & 1;
  const int n64 = 16 * wave + idx;
  // tiles of the last chunk; inside the loop the next chunk's tiles are stashed behind the barrier that ends the G update, so a chunk
  // costs three barriers (after the stash, after P / dP, before the writes to G and the tiles)
  {
    const int nvl = min(Lt, a.T - (nch - 1) * Lt) * a.A;
    stash32(Qs, pq, nvl, w4); stash32(Ks, pk, nvl, w4); stash32(Vs, pv, nvl, w4); stash32(Ds, pd, nvl, w4);
  }
  RP_DECL();
  for (int c = nch - 1; c >= 0; --c) {
    const int t0 = c * Lt;
    const int nvalid = min(Lt, a.T - t0) * a.A;
    const long r0 = row_base + (long)c * L;
    if (!pre) { __syncthreads(); build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, c, 1); }
    else __syncthreads();
    const ChunkMeta32& meta = sm.ch[pre ? c : 0];
    // chunk-entry state S_c as the row-form B fragment of dO S_c^T: rows n64 of the saved state, straight from global memory
    float4 sreg[4];
    {
      const float* Sc = a.states + ((long)seq * nch + c) * 4096 + (long)n64 * 64 + 4 * kq;
#pragma unroll
      for (int j = 0; j < 4; ++j) sreg[j] = *reinterpret_cast<const float4*>(Sc + 16 * j);
    }
    const long rn = c > 0 ? r0 - L : r0;
    const int nvn = c > 0 ? L : nvalid;
    R32_FETCH(pq, q, ldq, rn, nvn);
    RP(0);
    // P = (Q K^T) * w ; dP = (dO V^T) * w
    {
      f32x4 p[1] = {{0.f, 0.f, 0.f, 0.f}}, dp[1] = {{0.f, 0.f, 0.f, 0.f}};
      mma16<true, true, 64, 1>(p, Qs, TP, 16 * tr, Ks, TP, 16 * tc, idx, kq);
      mma16<true, true, 64, 1>(dp, Ds, TP, 16 * tr, Vs, TP, 16 * tc, idx, kq);
      const int j = 16 * tc + idx;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * tr + 4 * kq + i;
        const float w = w32(sm.kpow, meta, m, j, a.masked);
        Ps[m * PP + j] = p[0][i] * w;
        dPs[m * PP + j] = dp[0][i] * w;
      }
    }
    __syncthreads();
    R32_FETCH(pk, k, ldk, rn, nvn);
    RP(1);
    // dQ = dP K + beta * (dO S_c^T)
    {
      f32x4 a1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16<true, false, 32, 2>(a1, dPs, PP, 0, Ks, TP, 16 * wave, idx, kq);
      mma16_breg<64, 2>(a2, Ds, TP, 0, sreg, idx, kq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[r][i] += meta.beta[16 * r + 4 * kq + i] * a2[r][i];
        store16(a.dq, r0, a.lddq, a1[r], 16 * r, n64, kq, nvalid, a.hs);
      }
    }
    R32_FETCH(pv, v, ldv, rn, nvn);
    if (by_rows) {
      const int cp = max(c - 2, 0);
      fetch_idx32(ri, a.rows, row_base + (long)cp * L, min(Lt, a.T - cp * Lt) * a.A);
    }
    RP(2);
    // dK = dP^T Q + eta * (V G^T)
    {
      f32x4 a1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16<false, false, 32, 2>(a1, dPs, PP, 0, Qs, TP, 16 * wave, idx, kq);
      mma16<true, true, 64, 2>(a2, Vs, TP, 0, Gs, TP, 16 * wave, idx, kq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[r][i] += meta.eta[16 * r + 4 * kq + i] * a2[r][i];
        store16(a.dk, r0, a.lddk, a1[r], 16 * r, n64, kq, nvalid, a.hs);
      }
    }
    fetch32(pd, a.dr + rn * a.lddr, a.lddr, nvn, w4);
    RP(3);
    // dV = P^T dO + eta * (K G)
    {
      f32x4 a1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16<false, false, 32, 2>(a1, Ps, PP, 0, Ds, TP, 16 * wave, idx, kq);
      mma16<true, false, 64, 2>(a2, Ks, TP, 0, Gs, TP, 16 * wave, idx, kq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[r][i] += meta.eta[16 * r + 4 * kq + i] * a2[r][i];
        store16(a.dv, r0, a.lddv, a1[r], 16 * r, n64, kq, nvalid, a.hs);
      }
    }
    RP(4);
    // G <- gamma G + (beta Q)^T dO
    {
      f32x4 gn[4];
      const float gm = meta.gamma;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) gn[r][i] = gm * Gs[(16 * r + 4 * kq + i) * TP + n64];
      mma16<false, false, 32, 4>(gn, Qs, TP, 0, Ds, TP, 16 * wave, idx, kq, meta.beta);
      __syncthreads();   // every wave is done reading Gs (dK, dV) and this chunk's tiles
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) Gs[(16 * r + 4 * kq + i) * TP + n64] = gn[r][i];
    }
    if (c > 0) { stash32(Qs, pq, nvn, w4); stash32(Ks, pk, nvn, w4); stash32(Vs, pv, nvn, w4); stash32(Ds, pd, nvn, w4); }
    RP(5);
  }
  RP_FLUSH();
}
#undef R32_FETCH

}  // namespace magpo
