// Chunkwise retention on 32-token chunks (included by retention.hip; same math and arguments as k_ret_chunk_fwd / _bwd).
//
// Why a second tiling: on gfx950 MFMA time and VALU time add up per SIMD, so the only thing a second resident
// workgroup can hide is what the 64-token kernels spend in LDS round trips, barriers and vmcnt waits -- ~14 K of the backward's
// 36 K cycles per chunk -- and the 64-token backward cannot have one: its eight 64x64 tiles take 139 KB of LDS.  With 32-token
// chunks the token tiles are 32x64 (8.7 KB), the chunk-entry state of the backward is read as MFMA fragments straight from
// global memory, and a workgroup needs 69 KB at 16 chunks per sequence: two fit a CU (forward: 54 KB, three).  The causally /
// episode-masked half of the intra-chunk products shrinks with the chunk as well: 0.72 x the MFMA work of the 64-token kernels per
// token (forward 0.75 x).  Measured at the bench minibatch: backward 3.56 -> 2.5 ms per launch, forward 1.75 -> 1.45-1.5 ms (round 2);
// 2.36-2.45 / 1.26-1.60 ms with the round-3 load / store schedule (FAST instances, transposed result tiles: see below and DESIGN 4i).
//
// Tiles are 16x16 (v_mfma_f32_16x16x4_f32, the same flop rate as 32x32x2): lane l = (idx = l & 15, kq = l >> 4) supplies
// A[m0 + idx][k] and B[k][n0 + idx] for the k-slot k = kb + 4 kq + c of step c (c = 0..3 of a float4), and holds
// D[m0 + 4 kq + i][n0 + idx], i = 0..3.  Four waves: 32x32 outputs = one tile per wave; 32x64 outputs = column tile `wave`,
// both row tiles; 64x64 outputs (states) = column tile `wave`, four row tiles (the B fragment is shared by the row tiles).
//
// LDS layouts.  A tile is read two ways: by rows (ds_read_b128, lane (idx, kq) takes the float4 at k = kb + 4 kq of row idx) and
// by columns (ds_read_b32, lane takes element (kb + 4 kq + c, n0 + idx)).  DEFAULT: the +4 pitch (L64 = 68, L32 = L36 = 36 floats) -- the
// column reads are conflict-free, the row reads are not: the hardware serves a ds_read_b128 in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, (+32), a group mixes rows idx of two DIFFERENT kq, i.e. 16-byte slots (row offset + kq) and (row offset + kq + 1),
// and no pitch keeps those sixteen slots distinct (an odd slot pitch p collides wherever p (i - j) = 1 mod 16, and i - j takes every
// residue; the column reads need pitch = 4 mod 8 floats): one 2-way slot per group, SQ_LDS_BANK_CONFLICT 1.0e8 per backward launch.
// -DMAGPO_RET32_SWIZZLE (measured experiment): UNPADDED tiles with the 16-byte slot of a row XORed with the row number -- L64: slot ^= row
// & 15; L32 (two rows per 256-byte bank row): slot ^= g(row), g = row bits (2, 3, 1) -- are conflict-free for both reads and all writes
// (counter: 0), and not faster (2.46 vs 2.45 ms; ~70 more registers for the addresses): the conflicts were never the limiter.
// The P tile of the backward is only ever read by columns and keeps the +4 pitch either way (L36).
#pragma once

namespace magpo {

constexpr int MAXC32 = 32;       // chunks per sequence whose bookkeeping is built up front

#ifndef MAGPO_RET32_SWIZZLE
struct L64 { static constexpr int P = 68; static __device__ __forceinline__ int at(int row, int col) { return row * 68 + col; } };
struct L32 { static constexpr int P = 36; static __device__ __forceinline__ int at(int row, int col) { return row * 36 + col; } };
#else
struct L64 {
  static constexpr int P = 64;
  static __device__ __forceinline__ int at(int row, int col) { return (row << 6) + ((((col >> 2) ^ row) & 15) << 2) + (col & 3); }
};
struct L32 {
  static constexpr int P = 32;
  static __device__ __forceinline__ int g(int row) { return (row & 4) | ((row >> 1) & 1) | ((row >> 2) & 2); }
  static __device__ __forceinline__ int at(int row, int col) { return (row << 5) + ((((col >> 2) ^ g(row)) & 7) << 2) + (col & 3); }
};
#endif
struct L36 {
  static __device__ __forceinline__ int at(int row, int col) { return row * 36 + col; }
};

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

// decay weights of the pairs (query token m0 + i, key token j), i = 0..3, m0 a multiple of 4 (retention.py:117-187): the bookkeeping
// bytes of the four query tokens come in as two 32-bit words and every test is evaluated unconditionally -- a short-circuit chain puts
// each of its LDS reads into its own exec-masked block, four dependent round trips per weight in front of the barrier that ends the
// score phase (and a branch inside the chunk loop, see FAST below)
__device__ __forceinline__ void w32x4(float (&w)[4], const float* __restrict__ kpow, const ChunkMeta32& m, int m0, int j, int masked) {
  const int lj = m.lt[j], cj = m.cnt[j];
  const unsigned lt4 = *reinterpret_cast<const unsigned*>(&m.lt[m0]), cn4 = *reinterpret_cast<const unsigned*>(&m.cnt[m0]);
  bool on[4];
  int d[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int li = (int)(signed char)(lt4 >> (8 * i)), ci = (int)(signed char)(cn4 >> (8 * i));
    on[i] = (li >= 0) & (lj >= 0) & (li >= lj) & (ci == cj) & !((masked != 0) & (j > m0 + i));
    d[i] = on[i] ? li - lj : 0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float kp = kpow[d[i]];
    w[i] = on[i] ? kp : 0.f;
  }
}

// ---- 32 x 64 token tiles: global -> registers (one chunk ahead) -> LDS -------------------------------------------------------
struct Tile32 { float4 v[2]; };
struct RowIdx32 { int r[2]; };
__device__ __forceinline__ void fetch32(Tile32& t, const float* __restrict__ src, long ld, int nvalid, int w4) {
  if (nvalid == 32 && w4 == 16) {   // uniform: full tile = scalar row-block base + ONE per-thread 32-bit offset (no 64-bit address VALU)
    const unsigned toff = (unsigned)(threadIdx.x >> 4) * (unsigned)ld + 4u * (threadIdx.x & 15);
#pragma unroll
    for (int j = 0; j < 2; ++j) t.v[j] = *reinterpret_cast<const float4*>(src + 16 * j * ld + toff);
    return;
  }
  const int c4 = min((int)(threadIdx.x & 15), w4 - 1);   // unconditional loads from clamped addresses; stash32 zero-fills
#pragma unroll
  for (int j = 0; j < 2; ++j) t.v[j] = *reinterpret_cast<const float4*>(src + (long)min((int)(threadIdx.x >> 4) + 16 * j, nvalid - 1) * ld + 4 * c4);
}
__device__ __forceinline__ void fetch_idx32(RowIdx32& x, const int* __restrict__ rows, long row0, int nvalid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) x.r[j] = rows[row0 + min((int)(threadIdx.x >> 4) + 16 * j, nvalid - 1)];
}
// (FAST kernels: the row table of the whole sequence sits in LDS, built once per workgroup -- an index load carried across the back edge
// of the chunk loop would be waited for with vmcnt(0), behind the previous chunk's stores)
__device__ __forceinline__ void idx_from_lds(RowIdx32& x, const int* __restrict__ rowtab, int tok0) {
#pragma unroll
  for (int j = 0; j < 2; ++j) x.r[j] = rowtab[tok0 + (int)(threadIdx.x >> 4) + 16 * j];
}
__device__ __forceinline__ void fetch32_rows(Tile32& t, const float* __restrict__ tab, long ld, const RowIdx32& x, int w4) {
  const int c4 = min((int)(threadIdx.x & 15), w4 - 1);
#pragma unroll
  for (int j = 0; j < 2; ++j) t.v[j] = *reinterpret_cast<const float4*>(tab + (long)x.r[j] * ld + 4 * c4);
}
// (FULL: no zero-fill select -- the scheduler would hoist it to right behind the prefetch loads and wait for them there)
template <bool FULL>
__device__ __forceinline__ void stash32(float* __restrict__ dst, const Tile32& t, int nvalid, int w4) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (threadIdx.x >> 4) + 16 * j, c4 = threadIdx.x & 15;
    const bool ok = FULL || (r < nvalid && c4 < w4);
    *reinterpret_cast<float4*>(&dst[L64::at(r, 4 * c4)]) = ok ? t.v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// 64 x 64 state <-> L64 tile
__device__ __forceinline__ void load_state32(float* __restrict__ dst, const float* __restrict__ src) {
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {
    const int r = i >> 4, c4 = i & 15;
    const float4 v = src ? *reinterpret_cast<const float4*>(src + r * 64 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&dst[L64::at(r, 4 * c4)]) = v;
  }
}
__device__ __forceinline__ void store_state32(float* __restrict__ dst, const float* __restrict__ src) {
  const unsigned toff = threadIdx.x * 4u;   // thread i -> row i >> 4, slot i & 15: 16 bytes at float offset 4 i; + 1024 floats per round
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (threadIdx.x >> 4) + 16 * j, c4 = threadIdx.x & 15;
    *reinterpret_cast<float4*>(dst + 1024 * j + toff) = *reinterpret_cast<const float4*>(&src[L64::at(r, 4 * c4)]);
  }
}

// ---- acc[r] (+)= A B on 16x16 tiles: row tile r covers A rows am0 + 16 r .. + 15, the column tile B columns bn0 .. bn0 + 15 ------
// AROW: A[m][k] = At(m, k) (k contiguous, float4 reads), else A[m][k] = At(k, m) * (ascale ? ascale[k] : 1).
// BROW: B[k][n] = Bt(n, k), else B[k][n] = Bt(k, n).  LA / LB: the LDS layouts of the two tiles.
// TR: the two MFMA operands trade places, which computes the TRANSPOSED tile with the same products in the same order (bit-identical
// sums): the lane then holds out[am0 + 16 r + idx][bn0 + 4 kq + i], i = 0..3 -- four consecutive columns of one row, i.e. one 16-byte
// store per row tile instead of four 4-byte ones (the 32 x 64 results go to global memory; see store32x64).
template <bool AROW, bool BROW, int KK, int NR, class LA, class LB, bool TR = false>
__device__ __forceinline__ void mma16(f32x4 (&acc)[NR], const float* __restrict__ At, int am0, const float* __restrict__ Bt, int bn0, int idx, int kq,
                                      const float* __restrict__ ascale = nullptr) {
#pragma unroll
  for (int kb = 0; kb < KK; kb += 16) {
    const int k0 = kb + 4 * kq;
    float b[4];
    if (BROW) {
      const float4 b4 = *reinterpret_cast<const float4*>(Bt + LB::at(bn0 + idx, k0));
      b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = Bt[LB::at(k0 + c, bn0 + idx)];
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
        const float4 a4 = *reinterpret_cast<const float4*>(At + LA::at(am0 + 16 * r + idx, k0));
        av[r][0] = a4.x; av[r][1] = a4.y; av[r][2] = a4.z; av[r][3] = a4.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) av[r][c] = At[LA::at(k0 + c, am0 + 16 * r + idx)] * sc[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)   // consecutive MFMAs go to different accumulators (a dependent one waits for its predecessor's passes)
#pragma unroll
      for (int r = 0; r < NR; ++r)
        acc[r] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(b[c], av[r][c], acc[r], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][c], b[c], acc[r], 0, 0, 0);
  }
}
// same with the B fragment already in registers (row form: breg[kb / 16] = B^T[bn0 + idx][kb + 4 kq .. + 3]); transposed output (TR above)
template <int KK, int NR, class LA>
__device__ __forceinline__ void mma16_breg_tr(f32x4 (&acc)[NR], const float* __restrict__ At, int am0, const float4 (&breg)[KK / 16], int idx, int kq) {
#pragma unroll
  for (int kb = 0; kb < KK; kb += 16) {
    const float4 b4 = breg[kb / 16];
    float4 a4[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) a4[r] = *reinterpret_cast<const float4*>(At + LA::at(am0 + 16 * r + idx, kb + 4 * kq));
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.x, a4[r].x, acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.y, a4[r].y, acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.z, a4[r].z, acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.w, a4[r].w, acc[r], 0, 0, 0);
  }
}

// A [32 x 64] result held transposed (mma16 TR): this lane's columns n0 + 4 kq .. + 3 of rows 16 r + idx (r = 0, 1) to
// out[(r0 + row) * ld + column], one 16-byte store per row tile (a wave instruction writes 64 contiguous bytes of 16 rows).  Every lane
// ALWAYS issues its two stores -- full tiles (uniform): scalar row-tile base + one per-lane 32-bit offset; ragged chunks / narrow heads:
// the dropped rows / column quads go to the trash tile -- so the number of stores between a chunk's prefetch loads and their first use is
// a compile-time constant and the wait in front of the stash is vmcnt(#stores), not a drain of the store queue.  (Round 2 stored the
// untransposed accumulators: 24 four-byte stores per lane and chunk in the backward, each in its own exec-masked block; with the stores
// removed the kernel ran 15 % faster, scripts/debug/ret32_hooks.sh.)
__device__ __forceinline__ void store32x64(float* __restrict__ out, long r0, long ld, const f32x4 (&v)[2], int n0, int idx, int kq, bool full, int nvalid,
                                           int hs, float* __restrict__ trash) {
  const int n = n0 + 4 * kq;
  if (full) {
    const unsigned loff = (unsigned)idx * (unsigned)ld + (unsigned)n;
#pragma unroll
    for (int r = 0; r < 2; ++r) *reinterpret_cast<f32x4*>((out + (r0 + 16 * r) * ld) + loff) = v[r];
  } else {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int m = 16 * r + idx;
      float* dst = (m < nvalid && n < hs) ? out + (r0 + m) * ld + n : trash + m * 64 + n;
      *reinterpret_cast<f32x4*>(dst) = v[r];
    }
  }
}

// LDS per workgroup: the tiles + the bookkeeping of the chunks that are built up front (all of them when the sequence has at most MAXC32,
// else one at a time): 50 KB for the forward at 16 chunks (three workgroups per CU), 63 KB for the backward (two).
__host__ __device__ inline size_t ret32_rowtab_bytes(int nch, bool fast_by_rows) { return fast_by_rows ? (size_t)nch * 32 * sizeof(int) : 0; }
__host__ __device__ inline size_t ret32_meta_bytes(int nch) { return sizeof(float) * 36 + sizeof(ChunkMeta32) * (size_t)(nch <= MAXC32 ? nch : 1); }
__host__ __device__ inline size_t ret32_fwd_lds(int nch) { return (size_t)(3 * 32 * L64::P + 64 * L64::P + 32 * L32::P) * sizeof(float) + ret32_meta_bytes(nch); }
__host__ __device__ inline size_t ret32_bwd_lds(int nch) { return (size_t)(4 * 32 * L64::P + 64 * L64::P + 32 * 36 + 32 * L32::P) * sizeof(float) + ret32_meta_bytes(nch); }

// FAST: every chunk is full (32 % A == 0, T a multiple of the chunk's timesteps, 64-wide head) and the bookkeeping of all chunks is built
// up front -- the chunk loop then has NO branch, which is what lets the compiler count the stores between a prefetch and its use (a
// uniform branch around loads or stores inside the loop makes its s_waitcnt placement fall back to vmcnt(0)).  BYROWS: q | k | v through
// the row table (csrc/classtab.hip).
template <bool FAST, bool BYROWS>
__global__ __launch_bounds__(256, 3) void k_ret32_fwd(RetArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;              // [32][64] L64
  float* Ks = Qs + 32 * L64::P;
  float* Vs = Ks + 32 * L64::P;
  float* Ss = Vs + 32 * L64::P;  // [64][64] L64 carried state
  float* Ps = Ss + 64 * L64::P;  // [32][32] L32 masked scores (read by rows)
  SeqMeta32& sm = *reinterpret_cast<SeqMeta32*>(Ps + 32 * L32::P);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, kq = lane >> 4;
  const int seq = blockIdx.x;
  const int Lt = 32 / a.A, L = Lt * a.A;
  const int nch = (a.T + Lt - 1) / Lt;
  const long row_base = (long)seq * a.T * a.A;
  const float* s0 = a.s0 ? a.s0 + (long)(a.seq_env ? a.seq_env[seq] : seq) * 4096 : nullptr;
  const int w4 = FAST ? 16 : a.hs >> 2;
  load_state32(Ss, s0);
  const bool pre = FAST || nch <= MAXC32;
  if (tid < 36) sm.kpow[tid] = powf(a.kappa, (float)tid);
  __syncthreads();
  if (pre) build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, 0, nch);
  Tile32 pq, pk, pv;
  RowIdx32 ri;
  // (no LDS row table here, unlike the backward: its 2 KB would cost the third workgroup per CU, measured 1.28 vs 1.22 ms per launch;
  // the index registers carried across the back edge are waited for with vmcnt(0), which in this kernel only covers the eight
  // stores of the previous chunk's retention rows, issued a state update earlier)
  constexpr bool by_rows = BYROWS;
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
    stash32<FAST>(Qs, pq, nv0, w4);
    stash32<FAST>(Ks, pk, nv0, w4);
    stash32<FAST>(Vs, pv, nv0, w4);
  }
  const int tr = wave >> 1, tc = wave & 1;   // this wave's tile of a 32x32 result
  const int n64 = 16 * wave + idx;           // this lane's column of a 64-column result
  for (int c = 0; c < nch; ++c) {
    const int t0 = c * Lt;
    const int nvalid = FAST ? 32 : min(Lt, a.T - t0) * a.A;
    const bool full = FAST || (nvalid == 32 && a.hs == 64);
    const long r0 = row_base + (long)c * L;
    __syncthreads();   // this chunk's tiles are in place (stashed at the end of the previous iteration), Ss updated
    if (!pre) build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, c, 1);   // contains barriers
    const ChunkMeta32& meta = sm.ch[pre ? c : 0];
    // the next chunk's tiles (the last chunk re-reads itself, branch-free) are requested here, AHEAD of this chunk's stores: the stash at
    // the end of the iteration then waits with vmcnt(#stores) instead of draining the store queue
    const bool more = c + 1 < nch;
    const int nvn = FAST ? 32 : (more ? min(Lt, a.T - (t0 + Lt)) * a.A : nvalid);
    const long rn = more ? r0 + L : r0;
    R32_FETCH(pq, q, ldq, rn, nvn);
    R32_FETCH(pk, k, ldk, rn, nvn);
    R32_FETCH(pv, v, ldv, rn, nvn);
    if (by_rows) {
      const int c2 = min(c + 2, nch - 1);
      fetch_idx32(ri, a.rows, row_base + (long)c2 * L, min(Lt, a.T - c2 * Lt) * a.A);
    }
    if (FAST || a.states) store_state32(a.states + ((long)seq * nch + c) * 4096, Ss);   // (FAST: the host passes a states buffer)
    // scores (one 16x16 tile per wave) and Q S (column tile `wave`, both row tiles)
    f32x4 sc[1] = {{0.f, 0.f, 0.f, 0.f}};
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma16<true, true, 64, 1, L64, L64>(sc, Qs, 16 * tr, Ks, 16 * tc, idx, kq);
    mma16<true, false, 64, 2, L64, L64, true>(o, Qs, 0, Ss, 16 * wave, idx, kq);
    {
      const int j = 16 * tc + idx;
      float w[4];
      w32x4(w, sm.kpow, meta, 16 * tr + 4 * kq, j, a.masked);
#pragma unroll
      for (int i = 0; i < 4; ++i) Ps[L32::at(16 * tr + 4 * kq + i, j)] = sc[0][i] * w[i];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float be = meta.beta[16 * r + idx];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[r][i] *= be;
      }
    }
    __syncthreads();
    mma16<true, false, 32, 2, L32, L64, true>(o, Ps, 0, Vs, 16 * wave, idx, kq);      // P V
    store32x64(a.r, r0, a.ldr, o, 16 * wave, idx, kq, full, nvalid, a.hs, g_ret_trash);
    // state update  S <- gamma S + (eta K)^T V   (column tile `wave`, four row tiles; each lane rewrites the elements it read)
    f32x4 sn[4];
    const float gm = meta.gamma;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) sn[r][i] = gm * Ss[L64::at(16 * r + 4 * kq + i, n64)];
    mma16<false, false, 32, 4, L64, L64>(sn, Ks, 0, Vs, 16 * wave, idx, kq, meta.eta);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) Ss[L64::at(16 * r + 4 * kq + i, n64)] = sn[r][i];
    __syncthreads();   // every wave is done with this chunk's tiles
    stash32<FAST>(Qs, pq, nvn, w4);   // (after the last chunk: its own tiles again, unread)
    stash32<FAST>(Ks, pk, nvn, w4);
    stash32<FAST>(Vs, pv, nvn, w4);
  }
  if (a.s_final) {
    __syncthreads();
    store_state32(a.s_final + (long)seq * 4096, Ss);
  }
}

// timing hooks of the backward (debug builds only; results are wrong under them): -DMAGPO_RET32_NOSTORE / _NOLOAD / _NOBAR
#ifdef MAGPO_RET32_NOSTORE
#define R32_ST(...) do { if (a.T < 0) store32x64(__VA_ARGS__); } while (0)
#else
#define R32_ST(...) store32x64(__VA_ARGS__)
#endif
#ifdef MAGPO_RET32_NOBAR
#define R32_BAR() do { } while (0)
#else
#define R32_BAR() __syncthreads()
#endif
#ifdef MAGPO_RET32_NOLOAD
#define R32_LD(X) do { if (a.T < 0) { X; } } while (0)
#else
#define R32_LD(X) do { X; } while (0)
#endif
template <bool FAST, bool BYROWS>
__global__ __launch_bounds__(256, 2) void k_ret32_bwd(RetBwdArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;              // [32][64] L64
  float* Ks = Qs + 32 * L64::P;
  float* Vs = Ks + 32 * L64::P;
  float* Ds = Vs + 32 * L64::P;  // dO
  float* Gs = Ds + 32 * L64::P;  // [64][64] L64 dL/dS_{c+1}
  float* Ps = Gs + 64 * L64::P;  // [32][36] L36 (read by columns only)
  float* dPs = Ps + 32 * 36;     // [32][32] L32
  SeqMeta32& sm = *reinterpret_cast<SeqMeta32*>(dPs + 32 * L32::P);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, kq = lane >> 4;
  const int seq = blockIdx.x;
  const int Lt = 32 / a.A, L = Lt * a.A;
  const int nch = (a.T + Lt - 1) / Lt;
  const long row_base = (long)seq * a.T * a.A;
  const int w4 = FAST ? 16 : a.hs >> 2;
  load_state32(Gs, nullptr);
  const bool pre = FAST || nch <= MAXC32;
  if (tid < 36) sm.kpow[tid] = powf(a.kappa, (float)tid);
  __syncthreads();
  if (pre) build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, 0, nch);
  const int tr = wave >> 1, tc = wave & 1;
  const int n64 = 16 * wave + idx;
  Tile32 pq, pk, pv, pd;
  RowIdx32 ri;
  // chunk-entry state S_c as the row-form B fragment of dO S_c^T: rows n64 of the saved state, straight from global memory (no LDS copy);
  // sreg = the chunk being worked on, snext = the one after it in the reverse sweep
  float4 sreg[4], snext[4];
  const float* Sq = a.states + (long)seq * nch * 4096 + (long)n64 * 64 + 4 * kq;
  constexpr bool by_rows = BYROWS, TAB = FAST && BYROWS;
  int* rowtab = reinterpret_cast<int*>(reinterpret_cast<char*>(&sm) + ret32_meta_bytes(nch));   // [nch * 32] (TAB)
  if (TAB) {
    for (int i = tid; i < nch * 32; i += 256) rowtab[i] = a.rows[row_base + i];
    __syncthreads();
  }
  {
    const int cl = nch - 1;
    const int nvl = min(Lt, a.T - cl * Lt) * a.A;
    const long rl = row_base + (long)cl * L;
    if (TAB) idx_from_lds(ri, rowtab, cl * 32);
    else if (by_rows) fetch_idx32(ri, a.rows, rl, nvl);
#pragma unroll
    for (int j = 0; j < 4; ++j) sreg[j] = *reinterpret_cast<const float4*>(Sq + (long)cl * 4096 + 16 * j);
    R32_FETCH(pq, q, ldq, rl, nvl);
    R32_FETCH(pk, k, ldk, rl, nvl);
    R32_FETCH(pv, v, ldv, rl, nvl);
    fetch32(pd, a.dr + rl * a.lddr, a.lddr, nvl, w4);
    if (by_rows && !TAB) {
      const int cp = max(cl - 1, 0);
      fetch_idx32(ri, a.rows, row_base + (long)cp * L, min(Lt, a.T - cp * Lt) * a.A);
    }
    stash32<FAST>(Qs, pq, nvl, w4); stash32<FAST>(Ks, pk, nvl, w4); stash32<FAST>(Vs, pv, nvl, w4); stash32<FAST>(Ds, pd, nvl, w4);
  }
  RP_DECL();
  // Per chunk: three barriers (tiles + G in place; P / dP written; every wave done with G and the tiles), and ALL global loads of the
  // next chunk of the sweep (its four token tiles and its S fragments) are requested at the top, ahead of this chunk's six result stores
  // per lane, so that no wait further down has to drain the store queue.
  for (int c = nch - 1; c >= 0; --c) {
    const int t0 = c * Lt;
    const int nvalid = FAST ? 32 : min(Lt, a.T - t0) * a.A;
    const bool full = FAST || (nvalid == 32 && a.hs == 64);
    const long r0 = row_base + (long)c * L;
    if (!pre) { __syncthreads(); build_meta32(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, c, 1); }
    else __syncthreads();
    const ChunkMeta32& meta = sm.ch[pre ? c : 0];
    const long rn = c > 0 ? r0 - L : r0;      // chunk 0 re-reads itself (branch-free)
    const int nvn = FAST ? 32 : (c > 0 ? L : nvalid);
    {
      const int cn = max(c - 1, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) R32_LD(snext[j] = *reinterpret_cast<const float4*>(Sq + (long)cn * 4096 + 16 * j));
    }
    if (TAB) idx_from_lds(ri, rowtab, max(c - 1, 0) * 32);
    R32_LD(R32_FETCH(pq, q, ldq, rn, nvn));
    R32_LD(R32_FETCH(pk, k, ldk, rn, nvn));
    R32_LD(R32_FETCH(pv, v, ldv, rn, nvn));
    R32_LD(fetch32(pd, a.dr + rn * a.lddr, a.lddr, nvn, w4));
    if (by_rows && !TAB) {
      const int cp = max(c - 2, 0);
      fetch_idx32(ri, a.rows, row_base + (long)cp * L, min(Lt, a.T - cp * Lt) * a.A);
    }
    RP(0);
    // P = (Q K^T) * w ; dP = (dO V^T) * w
    {
      f32x4 p[1] = {{0.f, 0.f, 0.f, 0.f}}, dp[1] = {{0.f, 0.f, 0.f, 0.f}};
      mma16<true, true, 64, 1, L64, L64>(p, Qs, 16 * tr, Ks, 16 * tc, idx, kq);
      mma16<true, true, 64, 1, L64, L64>(dp, Ds, 16 * tr, Vs, 16 * tc, idx, kq);
      const int j = 16 * tc + idx;
      float w[4];
      w32x4(w, sm.kpow, meta, 16 * tr + 4 * kq, j, a.masked);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * tr + 4 * kq + i;
        Ps[L36::at(m, j)] = p[0][i] * w[i];
        dPs[L32::at(m, j)] = dp[0][i] * w[i];
      }
    }
    R32_BAR();
    RP(1);
    // dQ = dP K + beta * (dO S_c^T)
    {
      f32x4 a1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16<true, false, 32, 2, L32, L64, true>(a1, dPs, 0, Ks, 16 * wave, idx, kq);
      mma16_breg_tr<64, 2, L64>(a2, Ds, 0, sreg, idx, kq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float be = meta.beta[16 * r + idx];
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[r][i] += be * a2[r][i];
      }
      R32_ST(a.dq, r0, a.lddq, a1, 16 * wave, idx, kq, full, nvalid, a.hs, g_ret_trash);
    }
    RP(2);
    // dK = dP^T Q + eta * (V G^T)
    {
      f32x4 a1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16<false, false, 32, 2, L32, L64, true>(a1, dPs, 0, Qs, 16 * wave, idx, kq);
      mma16<true, true, 64, 2, L64, L64, true>(a2, Vs, 0, Gs, 16 * wave, idx, kq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float et = meta.eta[16 * r + idx];
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[r][i] += et * a2[r][i];
      }
      R32_ST(a.dk, r0, a.lddk, a1, 16 * wave, idx, kq, full, nvalid, a.hs, g_ret_trash);
    }
    RP(3);
    // dV = P^T dO + eta * (K G)
    {
      f32x4 a1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16<false, false, 32, 2, L36, L64, true>(a1, Ps, 0, Ds, 16 * wave, idx, kq);
      mma16<true, false, 64, 2, L64, L64, true>(a2, Ks, 0, Gs, 16 * wave, idx, kq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float et = meta.eta[16 * r + idx];
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[r][i] += et * a2[r][i];
      }
      R32_ST(a.dv, r0, a.lddv, a1, 16 * wave, idx, kq, full, nvalid, a.hs, g_ret_trash);
    }
    RP(4);
    // G <- gamma G + (beta Q)^T dO
    {
      f32x4 gn[4];
      const float gm = meta.gamma;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) gn[r][i] = gm * Gs[L64::at(16 * r + 4 * kq + i, n64)];
      mma16<false, false, 32, 4, L64, L64>(gn, Qs, 0, Ds, 16 * wave, idx, kq, meta.beta);
      R32_BAR();         // every wave is done reading Gs (dK, dV) and this chunk's tiles
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) Gs[L64::at(16 * r + 4 * kq + i, n64)] = gn[r][i];
    }
    stash32<FAST>(Qs, pq, nvn, w4); stash32<FAST>(Ks, pk, nvn, w4); stash32<FAST>(Vs, pv, nvn, w4); stash32<FAST>(Ds, pd, nvn, w4);   // (after chunk 0: its own tiles again, unread)
#pragma unroll
    for (int j = 0; j < 4; ++j) sreg[j] = snext[j];
    RP(5);
  }
  RP_FLUSH();
}
#undef R32_ST
#undef R32_BAR
#undef R32_LD
#undef R32_FETCH

}  // namespace magpo
