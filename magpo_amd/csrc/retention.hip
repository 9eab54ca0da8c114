// Sable retention on fp32 MFMA for gfx950.
//
// Reference semantics: SimpleRetention (mava/networks/retention.py:66-115): chunkwise
//   ret = ((Q K^T) * D) V + (Q S0) * xi     with the done-aware decay matrix D (:117-187) and xi (:189-213),
// recurrent  S <- S + k^T v ; ret = q S  with  S <- kappa S  applied once per timestep by the caller
// (sable_network.py:457) and S <- 0 at episode ends (rec_magpo.py:164-169).
//
// Both forms are the same linear recurrence  S_t = d_t S_{t-1} + sum_a k_{t,a}^T v_{t,a},
// d_t = kappa * (1 - done_t).  The reference evaluates one C x C chunk (C = T*A tokens); this kernel
// evaluates the identical sum chunk by chunk (64-token tiles) carrying the 64x64 state on chip, which
// needs 2.5x fewer FLOPs than the causal half of the C x C form and never materialises D:
//   intra-chunk  P = (Q K^T) * w ,  w(i,j) = kappa^(t_i - t_j) [same episode segment] [causal]
//   inter-chunk  O += beta_i (q_i S_c) ,  S_{c+1} = gamma_c S_c + sum_j eta_j k_j^T v_j
// One workgroup (4 waves, 2x2 quadrants of 32x32 accumulators) per sequence.  LDS tiles are
// [64][68] floats: row-per-lane operands are read with ds_read_b128, column-per-lane with ds_read_b32.
#include "common.hpp"
#include <stdlib.h>

namespace magpo {

constexpr int TL = 64 + LDP;  // tile pitch
// Sink for the stores of invalid tile elements (ragged last chunk, narrow heads): every lane always issues the same
// number of stores, so the compiler can wait for the prefetched tiles with vmcnt(#stores) instead of draining the
// store queue (vmcnt(0)) at the top of every chunk.
__device__ float g_ret_trash[64 * 64];
constexpr int FWD_MAXC = 8, BWD_MAXC = 16;   // chunks per sequence whose bookkeeping is built up front (LDS budget: 2 fwd workgroups / CU)

struct ChunkMeta {      // per-chunk decay bookkeeping in LDS
  signed char cnt[64];  // per token: # dones among chunk timesteps [0..lt]   (invalid tokens: -1)
  signed char lt[64];   // per token: chunk-local timestep (invalid tokens: -1)
  float beta[64];       // incoming-state weight per token
  float eta[64];        // outgoing-state weight per token
  float gamma;          // state carry factor
  float pad_[3];
};
// Bookkeeping of a whole sequence: built ONCE per workgroup (all chunks in parallel) when the sequence has at most
// MAXC chunks -- building it chunk by chunk costs ~6 us per chunk (dependent flag loads, powf, two barriers), a quarter
// of the backward kernel's time (in-kernel phase timing, scripts/ret_prof.py).
template <int MAXC> struct SeqMeta {
  float kpow[68];       // kappa^p, p = 0..65
  ChunkMeta ch[MAXC];
};

__device__ __forceinline__ void load_tile(float* __restrict__ dst, const float* __restrict__ src, long ld, int nvalid, int w4 = 16) {
  // 64 rows x 64 floats, rows >= nvalid and columns >= 4*w4 (head width) zero-filled
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {
    int r = i >> 4, c4 = i & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nvalid && c4 < w4) v = *reinterpret_cast<const float4*>(src + (long)r * ld + 4 * c4);
    *reinterpret_cast<float4*>(&dst[r * TL + 4 * c4]) = v;
  }
}
__device__ __forceinline__ void load_state(float* __restrict__ dst, const float* __restrict__ src) {
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {
    int r = i >> 4, c4 = i & 15;
    float4 v = src ? *reinterpret_cast<const float4*>(src + r * 64 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&dst[r * TL + 4 * c4]) = v;
  }
}
__device__ __forceinline__ void store_state(float* __restrict__ dst, const float* __restrict__ src) {
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {
    int r = i >> 4, c4 = i & 15;
    *reinterpret_cast<float4*>(dst + r * 64 + 4 * c4) = *reinterpret_cast<const float4*>(&src[r * TL + 4 * c4]);
  }
}

// register staging of a 64x64 tile (4 float4 per thread): fetch from HBM now, stash into LDS one chunk later
struct TileRegs { float4 v[4]; };
__device__ __forceinline__ void fetch_tile(TileRegs& t, const float* __restrict__ src, long ld, int nvalid, int w4) {
  // unconditional loads from clamped addresses (a predicated load is an exec-masked block); stash_tile zero-fills
  if (nvalid == 64 && w4 == 16) {   // uniform: full tile = scalar row-block base + ONE per-thread 32-bit offset (no 64-bit address VALU)
    const unsigned toff = (unsigned)(threadIdx.x >> 4) * (unsigned)ld + 4u * (threadIdx.x & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) t.v[j] = *reinterpret_cast<const float4*>(src + 16 * j * ld + toff);
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int r = min(i >> 4, nvalid - 1), c4 = min(i & 15, w4 - 1);
    t.v[j] = *reinterpret_cast<const float4*>(src + (long)r * ld + 4 * c4);
  }
}
// q | k | v read through a row table (csrc/classtab.hip: block-0 projections exist once per distinct input row; token row r
// reads table row rows[r]): the four table rows a thread fetches for a tile are looked up ONE CHUNK AHEAD (RowIdx), so the index
// load is never in front of the tile loads it feeds.
struct RowIdx { int r[4]; };
__device__ __forceinline__ void fetch_idx(RowIdx& x, const int* __restrict__ rows, long row0, int nvalid) {
#pragma unroll
  for (int j = 0; j < 4; ++j) x.r[j] = rows[row0 + min((int)(threadIdx.x >> 4) + 16 * j, nvalid - 1)];
}
__device__ __forceinline__ void fetch_tile_rows(TileRegs& t, const float* __restrict__ tab, long ld, const RowIdx& x, int w4) {
  const int c4 = min((int)(threadIdx.x & 15), w4 - 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) t.v[j] = *reinterpret_cast<const float4*>(tab + (long)x.r[j] * ld + 4 * c4);
}
__device__ __forceinline__ void fetch_state(TileRegs& t, const float* __restrict__ src) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j;
    t.v[j] = *reinterpret_cast<const float4*>(src + (i >> 4) * 64 + 4 * (i & 15));
  }
}
__device__ __forceinline__ void stash_tile(float* __restrict__ dst, const TileRegs& t, int nvalid = 64, int w4 = 16) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j;
    const bool ok = (i >> 4) < nvalid && (i & 15) < w4;
    *reinterpret_cast<float4*>(&dst[(i >> 4) * TL + 4 * (i & 15)]) = ok ? t.v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// fragment of 32 k-values for a row-per-lane operand: T[row][32h .. 32h+31]
struct Frag { float4 v[8]; };
__device__ __forceinline__ Frag load_rowfrag(const float* __restrict__ tile, int row, int h) {
  Frag f;
  const float* p = tile + row * TL + 32 * h;
#pragma unroll
  for (int u = 0; u < 8; ++u) f.v[u] = *reinterpret_cast<const float4*>(p + 4 * u);
  return f;
}
__device__ __forceinline__ float frag_at(const Frag& f, int s) {
  const float4& q = f.v[s >> 2];
  return (s & 3) == 0 ? q.x : ((s & 3) == 1 ? q.y : ((s & 3) == 2 ? q.z : q.w));
}

// acc += A B with A row-per-lane frag, B row-per-lane frag (B[k][n] = Tb[n][k])
__device__ __forceinline__ void mma_rr(f32x16& acc, const Frag& a, const Frag& b) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].x, b.v[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].y, b.v[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].z, b.v[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].w, b.v[u].w, acc, 0, 0, 0);
  }
}
// acc += A B with A row-per-lane frag, B column-per-lane from tile Tb[k][col]
__device__ __forceinline__ void mma_rc(f32x16& acc, const Frag& a, const float* __restrict__ tb, int col, int h) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float* p = tb + (32 * h + 4 * u) * TL + col;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].x, p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].y, p[TL], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].z, p[2 * TL], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u].w, p[3 * TL], acc, 0, 0, 0);
  }
}
// acc += A^T B: A[i][kk] = Ta[kk][arow_col] * scale[kk] (scale optional), B[kk][n] = Tb[kk][col]
__device__ __forceinline__ void mma_cc(f32x16& acc, const float* __restrict__ ta, int acol, const float* __restrict__ tb,
                                       int bcol, int h, const float* __restrict__ scale) {
#pragma unroll 8
  for (int s = 0; s < 32; ++s) {
    const int kk = 32 * h + s;
    float a = ta[kk * TL + acol];
    if (scale) a *= scale[kk];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, tb[kk * TL + bcol], acc, 0, 0, 0);
  }
}
// The 16 accumulator rows of a lane to out[(r0 + row)][32 wc + lr].  Full tiles (uniform): scalar row base per element + one
// per-lane 32-bit offset; otherwise always 16 stores per lane, invalid elements to the trash tile (no exec-masked blocks).
// f32 MFMA time and VALU time add up, and the generic 64-bit address of a store costs ~8 VALU instructions.
__device__ __forceinline__ void store_acc_rows(float* __restrict__ out, long r0, long ld, const float (&v)[16], bool full, int nvalid, int hs,
                                               int wr, int wc, int lr, int h, float* __restrict__ trash) {
  if (full) {
    const unsigned loff = (unsigned)(32 * wr + 4 * h) * (unsigned)ld + (unsigned)(32 * wc + lr);
#pragma unroll
    for (int i = 0; i < 16; ++i) (out + (r0 + (i & 3) + 8 * (i >> 2)) * ld)[loff] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ri = 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
      float* dst = (ri < nvalid && 32 * wc + lr < hs) ? out + (r0 + ri) * ld + 32 * wc + lr : trash + ri * 64 + 32 * wc + lr;
      *dst = v[i];
    }
  }
}
__device__ __forceinline__ void acc_zero(f32x16& a) {
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.f;
}
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// Build the metadata of chunks c0 .. c0+count-1 into ch[0 .. count-1].  dones: this sequence's per-timestep flags.
__device__ __forceinline__ void build_meta(float* __restrict__ kpow, ChunkMeta* __restrict__ ch, const unsigned char* __restrict__ dones,
                                           int T, int Lt, int A, float kappa, int c0, int count) {
  const int tid = threadIdx.x;
  for (int idx = tid; idx < count * 64; idx += 256) {
    const int cc = idx >> 6, tok = idx & 63;
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
  for (int idx = tid; idx < count * 64; idx += 256) {
    const int cc = idx >> 6, tok = idx & 63;
    const int t0 = (c0 + cc) * Lt, ltc = min(Lt, T - t0);
    ChunkMeta& m = ch[cc];
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

__device__ __forceinline__ float decay_w(const float* __restrict__ kpow, const ChunkMeta& m, int i, int j, int masked) {
  const int li = m.lt[i], lj = m.lt[j];
  if (li < 0 || lj < 0 || li < lj || m.cnt[i] != m.cnt[j]) return 0.f;
  if (masked && j > i) return 0.f;
  return kpow[li - lj];
}

// Per-lane copy of the chunk bookkeeping for the 16 accumulator rows (and the one column) a lane owns: loaded once per
// chunk with batched LDS reads, so that the GEMM epilogues do not pay one dependent LDS round trip per element.
struct LaneMeta {
  float beta[16], eta[16], w[16];   // w: decay weights of this lane's 16 (row, column) pairs
};
__device__ __forceinline__ void load_lane_meta(LaneMeta& lm, const float* __restrict__ kpow, const ChunkMeta& m, int wr, int col, int h,
                                               int masked) {
  const int lj = m.lt[col], cj = m.cnt[col];
  int li[16], ci[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ri = 32 * wr + acc_row(i, h);
    li[i] = m.lt[ri]; ci[i] = m.cnt[ri];
    lm.beta[i] = m.beta[ri]; lm.eta[i] = m.eta[ri];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ri = 32 * wr + acc_row(i, h);
    const bool on = li[i] >= 0 && lj >= 0 && li[i] >= lj && ci[i] == cj && !(masked && col > ri);
    const float kp = kpow[on ? li[i] - lj : 0];   // unconditional (clamped) read, selected afterwards
    lm.w[i] = on ? kp : 0.f;
  }
}

struct RetArgs {
  const float* q; const float* k; const float* v; long ldq, ldk, ldv;   // row strides (floats)
  float* r; long ldr;
  const float* s0;            // [*][64][64] initial states, indexed by seq_env (or seq)
  const int* seq_env;         // nullable
  const unsigned char* dones; // [nseq][T]
  float* states;              // [nseq][nch][64][64] chunk-entry states (nullable in fwd)
  float* s_final;             // [nseq][64][64] state after the last chunk (nullable)
  int T, A, masked; float kappa; int hs;   // hs = head width (<= 64): tiles are zero-padded to 64 columns on chip
  const int* rows;            // nullable: q | k | v are row tables, token row r reads table row rows[r]
};

__global__ __launch_bounds__(256, 2) void k_ret_chunk_fwd(RetArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;             // later P
  float* Ks = Qs + 64 * TL;
  float* Vs = Ks + 64 * TL;
  float* Ss = Vs + 64 * TL;
  SeqMeta<FWD_MAXC>& sm = *reinterpret_cast<SeqMeta<FWD_MAXC>*>(Ss + 64 * TL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, h = lane >> 5;
  const int seq = blockIdx.x;
  const int Lt = 64 / a.A, L = Lt * a.A;
  const int nch = (a.T + Lt - 1) / Lt;
  const long row_base = (long)seq * a.T * a.A;
  const float* s0 = a.s0 ? a.s0 + (long)(a.seq_env ? a.seq_env[seq] : seq) * 4096 : nullptr;
  const int w4 = a.hs >> 2;
  load_state(Ss, s0);
  const bool pre = nch <= FWD_MAXC;   // whole-sequence bookkeeping up front
  if (tid < 66) sm.kpow[tid] = powf(a.kappa, (float)tid);   // (visible after the first barrier inside build_meta)
  if (pre) build_meta(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, a.kappa, 0, nch);
  TileRegs pq, pk, pv;
  RowIdx ri;                          // table rows of the tile fetched next
  const bool by_rows = a.rows != nullptr;
#define RET_FETCH(T_, P_, LD_, ROW0_, NV_) \
  do { if (by_rows) fetch_tile_rows(T_, a.P_, a.LD_, ri, w4); else fetch_tile(T_, a.P_ + (ROW0_) * a.LD_, a.LD_, NV_, w4); } while (0)
  {
    const int nv0 = min(Lt, a.T) * a.A;
    if (by_rows) fetch_idx(ri, a.rows, row_base, nv0);
    RET_FETCH(pq, q, ldq, row_base, nv0);
    RET_FETCH(pk, k, ldk, row_base, nv0);
    RET_FETCH(pv, v, ldv, row_base, nv0);
    if (by_rows) {   // chunk 1 (or chunk 0 again when it is the only one)
      const int c1 = min(1, nch - 1);
      fetch_idx(ri, a.rows, row_base + (long)c1 * L, min(Lt, a.T - c1 * Lt) * a.A);
    }
  }
  for (int c = 0; c < nch; ++c) {
    const int t0 = c * Lt;
    const int ltc = min(Lt, a.T - t0);
    const int nvalid = ltc * a.A;
    const bool full = nvalid == 64 && a.hs == 64;
    const long r0 = row_base + (long)c * L;
    __syncthreads();  // previous chunk finished with Qs/Ks/Vs, Ss updated
    stash_tile(Qs, pq, nvalid, w4);
    stash_tile(Ks, pk, nvalid, w4);
    stash_tile(Vs, pv, nvalid, w4);
    if (!pre) build_meta(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, a.kappa, c, 1);  // contains __syncthreads
    else __syncthreads();
    const ChunkMeta& meta = sm.ch[pre ? c : 0];
    // next chunk's tiles (last chunk: re-reads itself, branch-free) are fetched one at a time between the GEMM phases
    const bool more = c + 1 < nch;
    const int nvn = more ? min(Lt, a.T - (t0 + Lt)) * a.A : nvalid;
    const long rn = more ? r0 + L : r0;
    RET_FETCH(pq, q, ldq, rn, nvn);
    if (a.states) store_state(a.states + ((long)seq * nch + c) * 4096, Ss);

    f32x16 sc, o;
    acc_zero(sc);
    acc_zero(o);
    LaneMeta lm;
    load_lane_meta(lm, sm.kpow, meta, wr, 32 * wc + lr, h, a.masked);
    {
      Frag qa = load_rowfrag(Qs, 32 * wr + lr, h);
      Frag kb = load_rowfrag(Ks, 32 * wc + lr, h);
      mma_rr(sc, qa, kb);                      // Q K^T
      mma_rc(o, qa, Ss, 32 * wc + lr, h);      // Q S
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ri = 32 * wr + acc_row(i, h);
      o[i] *= lm.beta[i];
      sc[i] *= lm.w[i];
    }
    __syncthreads();  // everyone done reading Qs
    RET_FETCH(pk, k, ldk, rn, nvn);
#pragma unroll
    for (int i = 0; i < 16; ++i) Qs[(32 * wr + acc_row(i, h)) * TL + 32 * wc + lr] = sc[i];
    __syncthreads();
    {
      Frag pa = load_rowfrag(Qs, 32 * wr + lr, h);
      mma_rc(o, pa, Vs, 32 * wc + lr, h);      // P V
    }
    {
      float ov[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ov[i] = o[i];
      store_acc_rows(a.r, r0, a.ldr, ov, full, nvalid, a.hs, wr, wc, lr, h, g_ret_trash);
    }
    RET_FETCH(pv, v, ldv, rn, nvn);
    if (by_rows) {   // rows of the chunk after the next one (the last chunk re-reads itself)
      const int c2 = min(c + 2, nch - 1);
      fetch_idx(ri, a.rows, row_base + (long)c2 * L, min(Lt, a.T - c2 * Lt) * a.A);
    }
    // state update  S <- gamma S + (eta K)^T V
    f32x16 sn;
    const float gm = meta.gamma;
#pragma unroll
    for (int i = 0; i < 16; ++i) sn[i] = gm * Ss[(32 * wr + acc_row(i, h)) * TL + 32 * wc + lr];
    mma_cc(sn, Ks, 32 * wr + lr, Vs, 32 * wc + lr, h, meta.eta);
    __syncthreads();  // all reads of Ss done
#pragma unroll
    for (int i = 0; i < 16; ++i) Ss[(32 * wr + acc_row(i, h)) * TL + 32 * wc + lr] = sn[i];
  }
  if (a.s_final) {
    __syncthreads();
    store_state(a.s_final + (long)seq * 4096, Ss);
  }
}

struct RetBwdArgs {
  const float* q; const float* k; const float* v; long ldq, ldk, ldv;
  const float* dr; long lddr;
  float* dq; float* dk; float* dv; long lddq, lddk, lddv;
  const unsigned char* dones;
  const float* states;  // [nseq][nch][64][64] from the forward
  int T, A, masked; float kappa; int hs;
  const int* rows;      // nullable: as in RetArgs (q | k | v only; dr / dq / dk / dv are per token row)
};

#ifdef MAGPO_RET_PROF
__device__ unsigned long long g_ret_prof[8];
#define RP_DECL() unsigned long long rp_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long rp_last = clock64();
#define RP(k) do { unsigned long long t_ = clock64(); rp_acc[k] += t_ - rp_last; rp_last = t_; } while (0)
#define RP_FLUSH() do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_ret_prof[k_], rp_acc[k_]); } } while (0)
#else
#define RP_DECL()
#define RP(k)
#define RP_FLUSH()
#endif

__global__ __launch_bounds__(256, 1) void k_ret_chunk_bwd(RetBwdArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;
  float* Ks = Qs + 64 * TL;
  float* Vs = Ks + 64 * TL;
  float* Ds = Vs + 64 * TL;   // dO
  float* Ss = Ds + 64 * TL;   // chunk-entry state S_c
  float* Gs = Ss + 64 * TL;   // dL/dS_{c+1}
  float* Ps = Gs + 64 * TL;
  float* dPs = Ps + 64 * TL;
  SeqMeta<BWD_MAXC>& sm = *reinterpret_cast<SeqMeta<BWD_MAXC>*>(dPs + 64 * TL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, h = lane >> 5;
  const int seq = blockIdx.x;
  const int Lt = 64 / a.A, L = Lt * a.A;
  const int nch = (a.T + Lt - 1) / Lt;
  const long row_base = (long)seq * a.T * a.A;
  const int w4 = a.hs >> 2;
  load_state(Gs, nullptr);
  const bool pre = nch <= BWD_MAXC;
  if (tid < 66) sm.kpow[tid] = powf(a.kappa, (float)tid);   // (visible after the first barrier inside build_meta)
  if (pre) build_meta(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, a.kappa, 0, nch);
  RP_DECL();
  TileRegs pq, pk, pv, pd, ps;
  RowIdx ri;                          // table rows of the q | k | v tiles fetched next
  const bool by_rows = a.rows != nullptr;
  {
    const int cl = nch - 1;
    const int nvl = min(Lt, a.T - cl * Lt) * a.A;
    const long rl = row_base + (long)cl * L;
    if (by_rows) fetch_idx(ri, a.rows, rl, nvl);
    RET_FETCH(pq, q, ldq, rl, nvl);
    RET_FETCH(pk, k, ldk, rl, nvl);
    RET_FETCH(pv, v, ldv, rl, nvl);
    if (by_rows) {   // the chunk before (chunk 0 re-reads itself)
      const int cp = max(cl - 1, 0);
      fetch_idx(ri, a.rows, row_base + (long)cp * L, min(Lt, a.T - cp * Lt) * a.A);
    }
    fetch_tile(pd, a.dr + rl * a.lddr, a.lddr, nvl, w4);
    fetch_state(ps, a.states + ((long)seq * nch + cl) * 4096);
  }
  for (int c = nch - 1; c >= 0; --c) {
    const int t0 = c * Lt;
    const int ltc = min(Lt, a.T - t0);
    const int nvalid = ltc * a.A;
    const bool full = nvalid == 64 && a.hs == 64;
    const long r0 = row_base + (long)c * L;
    __syncthreads();
    stash_tile(Qs, pq, nvalid, w4);
    stash_tile(Ks, pk, nvalid, w4);
    stash_tile(Vs, pv, nvalid, w4);
    stash_tile(Ds, pd, nvalid, w4);
    stash_tile(Ss, ps);
    if (!pre) {
      build_meta(sm.kpow, sm.ch, a.dones + (long)seq * a.T, a.T, Lt, a.A, a.kappa, c, 1);
    } else __syncthreads();
    const ChunkMeta& meta = sm.ch[pre ? c : 0];
    // previous chunk (next in the reverse sweep; chunk 0 re-reads itself): its five tiles are fetched one at a time between
    // the GEMM phases below -- issued as one burst of 20 loads they would sit in front of the first MFMAs in the wave's in-order
    // instruction stream while the memory pipeline back-pressures
    const long rn = c > 0 ? r0 - L : r0;
    const int nvn = c > 0 ? L : nvalid;
    RET_FETCH(pq, q, ldq, rn, nvn);

    RP(0);
    LaneMeta lm;
    load_lane_meta(lm, sm.kpow, meta, wr, 32 * wc + lr, h, a.masked);
    // P = (Q K^T) * w ; dP = (dO V^T) * w
    Frag doa = load_rowfrag(Ds, 32 * wr + lr, h);
    {
      f32x16 p, dp;
      acc_zero(p);
      acc_zero(dp);
      Frag qa = load_rowfrag(Qs, 32 * wr + lr, h);
      Frag kb = load_rowfrag(Ks, 32 * wc + lr, h);
      mma_rr(p, qa, kb);
      Frag vb = load_rowfrag(Vs, 32 * wc + lr, h);
      mma_rr(dp, doa, vb);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ri = 32 * wr + acc_row(i, h);
        Ps[ri * TL + 32 * wc + lr] = p[i] * lm.w[i];
        dPs[ri * TL + 32 * wc + lr] = dp[i] * lm.w[i];
      }
    }
    __syncthreads();
    RET_FETCH(pk, k, ldk, rn, nvn);
    RP(1);
    // dQ = dP K + beta * (dO S_c^T)
    {
      f32x16 acc1, acc2;
      acc_zero(acc1);
      acc_zero(acc2);
      Frag dpa = load_rowfrag(dPs, 32 * wr + lr, h);
      mma_rc(acc1, dpa, Ks, 32 * wc + lr, h);
      Frag sb = load_rowfrag(Ss, 32 * wc + lr, h);
      mma_rr(acc2, doa, sb);
      float ov[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ov[i] = acc1[i] + lm.beta[i] * acc2[i];
      store_acc_rows(a.dq, r0, a.lddq, ov, full, nvalid, a.hs, wr, wc, lr, h, g_ret_trash);
    }
    RET_FETCH(pv, v, ldv, rn, nvn);
    if (by_rows) {
      const int cp = max(c - 2, 0);
      fetch_idx(ri, a.rows, row_base + (long)cp * L, min(Lt, a.T - cp * Lt) * a.A);
    }
    RP(2);
    // dK = dP^T Q + eta * (V G^T)
    {
      f32x16 acc1, acc2;
      acc_zero(acc1);
      acc_zero(acc2);
      mma_cc(acc1, dPs, 32 * wr + lr, Qs, 32 * wc + lr, h, nullptr);
      Frag va = load_rowfrag(Vs, 32 * wr + lr, h);
      Frag gb = load_rowfrag(Gs, 32 * wc + lr, h);
      mma_rr(acc2, va, gb);
      float ov[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ov[i] = acc1[i] + lm.eta[i] * acc2[i];
      store_acc_rows(a.dk, r0, a.lddk, ov, full, nvalid, a.hs, wr, wc, lr, h, g_ret_trash);
    }
    fetch_tile(pd, a.dr + rn * a.lddr, a.lddr, nvn, w4);
    RP(3);
    // dV = P^T dO + eta * (K G)
    {
      f32x16 acc1, acc2;
      acc_zero(acc1);
      acc_zero(acc2);
      mma_cc(acc1, Ps, 32 * wr + lr, Ds, 32 * wc + lr, h, nullptr);
      Frag ka = load_rowfrag(Ks, 32 * wr + lr, h);
      mma_rc(acc2, ka, Gs, 32 * wc + lr, h);
      float ov[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ov[i] = acc1[i] + lm.eta[i] * acc2[i];
      store_acc_rows(a.dv, r0, a.lddv, ov, full, nvalid, a.hs, wr, wc, lr, h, g_ret_trash);
    }
    fetch_state(ps, a.states + ((long)seq * nch + max(c - 1, 0)) * 4096);
    RP(4);
    // G <- gamma G + (beta Q)^T dO
    {
      f32x16 gn;
      const float gm = meta.gamma;
#pragma unroll
      for (int i = 0; i < 16; ++i) gn[i] = gm * Gs[(32 * wr + acc_row(i, h)) * TL + 32 * wc + lr];
      mma_cc(gn, Qs, 32 * wr + lr, Ds, 32 * wc + lr, h, meta.beta);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) Gs[(32 * wr + acc_row(i, h)) * TL + 32 * wc + lr] = gn[i];
    }
    RP(5);
  }
  RP_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// Recurrent form for acting (retention.py:102-115): one workgroup per env.
//   S_eff = decay * S + sum_{a<ntok} k_a^T v_a ;  ret_a = q_a S_eff   for a in [ret_from, ntok)
// The state is written back only when write_state != 0: the autoregressive decoder calls this once per
// agent with the tokens decoded so far (ntok = i + 1, ret_from = i) and stores the state after the last
// agent only, so a decoder state costs A reads + 1 write per env step instead of A reads + A writes.
__global__ __launch_bounds__(256) void k_ret_recurrent(float* __restrict__ S, const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, long ldq, long ldk, long ldv, long env_stride_rows,
                                                       float* __restrict__ r, long ldr, int ntok, int ret_from, float decay,
                                                       int write_state, const float* __restrict__ gp, long ldg,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int hs, int gs) {
  // thread -> 4 state columns (c4..c4+3) x 4 state rows (rw, rw+16, rw+32, rw+48): float4 accesses, a wave
  // touches 4 consecutive 256-B rows (1 KiB contiguous) per instruction.
  __shared__ __align__(16) float qs[16][64], ks[16][64], vs[16][64];
  __shared__ __align__(16) float part[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4 = 4 * (tid & 15), rw = tid >> 4;
  const long env = blockIdx.x;
  const long row0 = env * env_stride_rows;
  float* Se = S + env * 4096;
  float4 s[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = *reinterpret_cast<const float4*>(Se + (rw + 16 * i) * 64 + c4);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < ntok * 16; i += 256) {
    const int a = i >> 4, cc = 4 * (i & 15);
    const bool in = cc < hs;  // head width: columns beyond it are zero (state padded to 64 x 64)
    if (a >= ret_from) *reinterpret_cast<float4*>(&qs[a][cc]) = in ? *reinterpret_cast<const float4*>(q + (row0 + a) * ldq + cc) : z4;
    *reinterpret_cast<float4*>(&ks[a][cc]) = in ? *reinterpret_cast<const float4*>(k + (row0 + a) * ldk + cc) : z4;
    *reinterpret_cast<float4*>(&vs[a][cc]) = in ? *reinterpret_cast<const float4*>(v + (row0 + a) * ldv + cc) : z4;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s[i].x *= decay; s[i].y *= decay; s[i].z *= decay; s[i].w *= decay;
  }
  for (int a = 0; a < ntok; ++a) {
    const float4 vv = *reinterpret_cast<const float4*>(&vs[a][c4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float kk = ks[a][rw + 16 * i];
      s[i].x += kk * vv.x; s[i].y += kk * vv.y; s[i].z += kk * vv.z; s[i].w += kk * vv.w;
    }
  }
  if (write_state) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(Se + (rw + 16 * i) * 64 + c4) = s[i];
  }
  for (int a = ret_from; a < ntok; ++a) {
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float qq = qs[a][rw + 16 * i];
      p.x += qq * s[i].x; p.y += qq * s[i].y; p.z += qq * s[i].z; p.w += qq * s[i].w;
    }
    // sum over the 4 row-lanes of the wave (lanes l, l^16, l^32, l^48), then over the 4 waves through LDS
    p.x += __shfl_xor(p.x, 16, 64); p.y += __shfl_xor(p.y, 16, 64); p.z += __shfl_xor(p.z, 16, 64); p.w += __shfl_xor(p.w, 16, 64);
    p.x += __shfl_xor(p.x, 32, 64); p.y += __shfl_xor(p.y, 32, 64); p.z += __shfl_xor(p.z, 32, 64); p.w += __shfl_xor(p.w, 32, 64);
    if (lane < 16) *reinterpret_cast<float4*>(&part[wave][a][c4]) = p;
  }
  __syncthreads();
  if (!gp) {
    for (int i = tid; i < (ntok - ret_from) * 64; i += 256) {
      int a = ret_from + (i >> 6), c = i & 63;
      if (c < hs) r[(row0 + a) * ldr + c] = (part[0][a][c] + part[1][a][c]) + (part[2][a][c] + part[3][a][c]);
    }
  } else {
    // fused retention epilogue (retention.py:289-294): u = swish(gpre) * GroupNorm(ret); one wave per token.
    // flax GroupNorm(num_groups = n_head) on the (token*head, hs) rows: statistics over groups of gs = hs / n_head channels.
    for (int a = ret_from + wave; a < ntok; a += 4) {
      const float x = (part[0][a][lane] + part[1][a][lane]) + (part[2][a][lane] + part[3][a][lane]);
      float s1 = x, s2 = x * x;
      for (int o = gs >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      const float mu = s1 / (float)gs, m2 = s2 / (float)gs;
      const float rstd = rsqrtf(fmaxf(m2 - mu * mu, 0.f) + 1e-6f);
      if (lane < hs) {
        const float rn = (x - mu) * rstd * gamma[lane] + beta[lane];
        r[(row0 + a) * ldr + lane] = swishf_(gp[(row0 + a) * ldg + lane]) * rn;
      }
    }
  }
}

// zero the three retention states of envs whose episode just ended (rec_magpo.py:164-169)
__global__ void k_zero_states(float* __restrict__ s0, float* __restrict__ s1, float* __restrict__ s2,
                              const unsigned char* __restrict__ done) {
  if (!done[blockIdx.x]) return;
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
    reinterpret_cast<float4*>(s0 + (long)blockIdx.x * 4096)[i] = z;
    reinterpret_cast<float4*>(s1 + (long)blockIdx.x * 4096)[i] = z;
    reinterpret_cast<float4*>(s2 + (long)blockIdx.x * 4096)[i] = z;
  }
}

}  // namespace magpo

#include "retention32.hpp"

using namespace magpo;

// Tokens per chunk of the chunkwise kernels, a per-call argument (no library state): 32 (k_ret32_*, retention32.hpp; also 0 = default;
// teams of at most 32 agents) or 64 (k_ret_chunk_*).  The forward's saved chunk-entry states and the backward must be given the same value
// (magpo_retention_num_chunks takes it too).
static bool ret32(int A, int chunk_tokens) { return chunk_tokens != 64 && A <= 32; }
// branch-free chunk loop of the 32-token kernels: every chunk full, 64-wide head, all bookkeeping built up front
static bool ret32_fast(int T, int A, int hs, int nch) { return hs == 64 && 32 % A == 0 && T % (32 / A) == 0 && nch <= MAXC32; }
static int check_chunk_tokens(int chunk_tokens) {
  if (chunk_tokens != 0 && chunk_tokens != 32 && chunk_tokens != 64) { set_error("retention: chunk_tokens must be 0 (default), 32 or 64"); return MAGPO_EINVAL; }
  return MAGPO_OK;
}

static int check_ret_shape(int T, int A, long ldq, long ldk, long ldv) {
  if (A < 1 || A > 64 || T < 1) { set_error("retention: need 1 <= A <= 64, T >= 1"); return MAGPO_EINVAL; }
  if ((ldq & 3) || (ldk & 3) || (ldv & 3)) { set_error("retention: row strides must be multiples of 4 floats"); return MAGPO_EINVAL; }
  return MAGPO_OK;
}

extern "C" int magpo_retention_num_chunks(int T, int A, int chunk_tokens) { int Lt = (ret32(A, chunk_tokens) ? 32 : 64) / A; return (T + Lt - 1) / Lt; }

extern "C" int magpo_retention_chunk_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                         float* r, long ldr, const float* s0, const int* seq_env,
                                         const unsigned char* dones, float* states, float* s_final, int nseq, int T, int A,
                                         int masked, float kappa, int hs, const int* qkv_rows, int chunk_tokens, hipStream_t st) {
  if (int e = check_ret_shape(T, A, ldq, ldk, ldv)) return e;
  if (int e = check_chunk_tokens(chunk_tokens)) return e;
  if (hs < 4 || hs > 64 || (hs & 3)) { set_error("retention: head width must be a multiple of 4 in [4, 64]"); return MAGPO_EINVAL; }
  RetArgs a{q, k, v, ldq, ldk, ldv, r, ldr, s0, seq_env, dones, states, s_final, T, A, masked, kappa, hs, qkv_rows};
  if (ret32(A, chunk_tokens)) {
    const int nch = magpo_retention_num_chunks(T, A, chunk_tokens);
    const bool fast = ret32_fast(T, A, hs, nch) && states != nullptr;
    const size_t lds32 = ret32_fwd_lds(nch);
    const int v = (fast ? 2 : 0) | (qkv_rows ? 1 : 0);
    static size_t attr32[4] = {0, 0, 0, 0};
    void (*const kern[4])(RetArgs) = {k_ret32_fwd<false, false>, k_ret32_fwd<false, true>, k_ret32_fwd<true, false>, k_ret32_fwd<true, true>};
    if (lds32 > attr32[v]) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern[v]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32); attr32[v] = lds32; }
    hipLaunchKernelGGL(kern[v], dim3(nseq), dim3(256), lds32, st, a);
    return check_launch("magpo_retention_chunk_fwd");
  }
  size_t lds = 4 * 64 * TL * sizeof(float) + sizeof(SeqMeta<FWD_MAXC>);
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ret_chunk_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(k_ret_chunk_fwd, dim3(nseq), dim3(256), lds, st, a);
  return check_launch("magpo_retention_chunk_fwd");
}

extern "C" int magpo_retention_chunk_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                         const float* dr, long lddr, float* dq, long lddq, float* dk, long lddk, float* dv,
                                         long lddv, const unsigned char* dones, const float* states, int nseq, int T, int A,
                                         int masked, float kappa, int hs, const int* qkv_rows, int chunk_tokens, hipStream_t st) {
  if (int e = check_ret_shape(T, A, ldq, ldk, ldv)) return e;
  if (int e = check_chunk_tokens(chunk_tokens)) return e;
  if (hs < 4 || hs > 64 || (hs & 3)) { set_error("retention: head width must be a multiple of 4 in [4, 64]"); return MAGPO_EINVAL; }
  RetBwdArgs a{q, k, v, ldq, ldk, ldv, dr, lddr, dq, dk, dv, lddq, lddk, lddv, dones, states, T, A, masked, kappa, hs, qkv_rows};
  if (ret32(A, chunk_tokens)) {
    const int nch = magpo_retention_num_chunks(T, A, chunk_tokens);
    const bool fast = ret32_fast(T, A, hs, nch);
    const size_t lds32 = ret32_bwd_lds(nch) + ret32_rowtab_bytes(nch, fast && qkv_rows);
    const int v = (fast ? 2 : 0) | (qkv_rows ? 1 : 0);
    static size_t attr32[4] = {0, 0, 0, 0};
    void (*const kern[4])(RetBwdArgs) = {k_ret32_bwd<false, false>, k_ret32_bwd<false, true>, k_ret32_bwd<true, false>, k_ret32_bwd<true, true>};
    if (lds32 > attr32[v]) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern[v]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32); attr32[v] = lds32; }
    hipLaunchKernelGGL(kern[v], dim3(nseq), dim3(256), lds32, st, a);
    return check_launch("magpo_retention_chunk_bwd");
  }
  size_t lds = 8 * 64 * TL * sizeof(float) + sizeof(SeqMeta<BWD_MAXC>);
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ret_chunk_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(k_ret_chunk_bwd, dim3(nseq), dim3(256), lds, st, a);
  return check_launch("magpo_retention_chunk_bwd");
}

extern "C" int magpo_retention_recurrent(float* S, const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                         long env_stride_rows, float* r, long ldr, int nenv, int ntok, int ret_from, float decay,
                                         int write_state, const float* gp, long ldg, const float* gamma, const float* beta,
                                         int hs, int gs, hipStream_t st) {
  if (ntok < 1 || ntok > 16 || ret_from < 0 || ret_from >= ntok) { set_error("retention_recurrent: 1 <= ntok <= 16, 0 <= ret_from < ntok"); return MAGPO_EINVAL; }
  if (hs < 4 || hs > 64 || (hs & 3) || gs < 1 || gs > hs || (gs & (gs - 1))) { set_error("retention_recurrent: bad head width / group size"); return MAGPO_EINVAL; }
  hipLaunchKernelGGL(k_ret_recurrent, dim3(nenv), dim3(256), 0, st, S, q, k, v, ldq, ldk, ldv, env_stride_rows, r, ldr, ntok, ret_from,
                     decay, write_state, gp, ldg, gamma, beta, hs, gs);
  return check_launch("magpo_retention_recurrent");
}

extern "C" int magpo_zero_states_where_done(float* s0, float* s1, float* s2, const unsigned char* done, int nenv,
                                            hipStream_t st) {
  hipLaunchKernelGGL(k_zero_states, dim3(nenv), dim3(256), 0, st, s0, s1, s2, done);
  return check_launch("magpo_zero_states_where_done");
}

#ifdef MAGPO_RET_PROF
extern "C" int magpo_debug_ret_prof(unsigned long long* out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(magpo::g_ret_prof), sizeof(unsigned long long) * 8) != hipSuccess) return MAGPO_ELAUNCH;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(magpo::g_ret_prof), z, sizeof(z)) != hipSuccess) return MAGPO_ELAUNCH; }
  return MAGPO_OK;
}
#endif
