// Fused "post-retention" training-forward segments of the Sable blocks (sable_network.py:62-71, 188-217, 277-319;
// retention.py:289-295) for gfx950.  Between two retention ops the network is token-local; the kernel-by-kernel path runs
// 4-6 launches there (GroupNorm + gate, W_o, residual + RMSNorm, next projection / head) and re-reads every intermediate
// from HBM.  Here one persistent wave carries 16 token rows through the whole segment in registers (feature-major rows
// and transposed 16x16x4 fp32 MFMA layers, fm_rows.hpp); it reads the three input rows once and writes only what the
// hand-written backward (or the next retention) consumes.  The weights of the segment stay in VGPRs across tiles.
//   front (all segments):  u = swish(g) * GroupNorm(r) ; y = u W_o ; o = rms(res + y) s1 [-> rms(.) s2] ; ope = o + pe[pos]
//   tail ENC  : hv = o W_v0 + b ; value = rms(gelu(hv)) s . w + b1 ; q2_k = ope W_q2[k]   (every decoder block k)
//   tail DEC1 : kvg2 = ope W_kvg2                                                      (192 columns)
//   tail DEC2 : (last block) hp = o W_h0 + b ; hn = rms(gelu(hp)) s ; logits = hn W_h1 + b   (K columns)
#include "fm_rows.hpp"

namespace magpo {

struct SegArgs {
  int tail;                               // 0 none, 1 ENC, 2 DEC1, 3 DEC2
  long R; int K;
  // front
  const float* r; const float* gp; long ldg; const float* gamma; const float* beta;
  const float* wo_t; const float* res; const float* s1; const float* s2;
  const float* pe; const int* pos; int npos;
  float* u; float* y; float* o; float* ope;            // o / ope nullable
  // tails
  const float* w0_t; const float* b0; float* out0; long ld0;          // ENC: vh0 -> hv ; DEC1: kvg2 (192) ; DEC2: h0 -> hp
  const float* hs; const float* hw; const float* hb1; float* value;   // ENC: head norm scale, value weights / bias
  const float* q2_t[4]; float* q2[4]; int nq2;                        // ENC: cross-retention queries of every decoder block
  float* hn; const float* w1_t; const float* b1; float* logits;       // DEC2: head
  const int* rows;   // nullable: gp and res are row tables (csrc/classtab.hip), token row r reads table row rows[r]
};

__device__ __forceinline__ Row row_rms_reg(const Row& x, const Row& s) {
  Row q;
#pragma unroll
  for (int j = 0; j < 16; ++j) q.v[j] = x.v[j] * x.v[j];
  const float rstd = rsqrtf(row_sum(q) * (1.0f / 64.0f) + EPSN);
  Row r;
#pragma unroll
  for (int j = 0; j < 16; ++j) r.v[j] = x.v[j] * rstd * s.v[j];
  return r;
}

// 64 -> 64 layer with the weight fragments already in registers (w[g][gk]: rows 16 g + m, columns 16 gk + 4 kq ..)
__device__ __forceinline__ Row dense64_reg(const Row& x, const float4 (&w)[4][4], const float* __restrict__ bias, int kq) {
  Row y;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][gk].x, x.v[4 * gk], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][gk].y, x.v[4 * gk + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][gk].z, x.v[4 * gk + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][gk].w, x.v[4 * gk + 3], acc, 0, 0, 0);
    }
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = ld4g(bias + 16 * g + 4 * kq);
    y.v[4 * g] = acc[0] + b.x; y.v[4 * g + 1] = acc[1] + b.y; y.v[4 * g + 2] = acc[2] + b.z; y.v[4 * g + 3] = acc[3] + b.w;
  }
  return y;
}
// same layer with the weight fragments read from an LDS copy of W^T ([64][WSP] floats: same products in the same order as dense64_reg)
constexpr int WSP = 68;
__device__ __forceinline__ Row dense64_lds(const Row& x, const float* __restrict__ ws, int m, int kq) {
  Row y;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      const float4 w = *reinterpret_cast<const float4*>(ws + (16 * g + m) * WSP + 16 * gk + 4 * kq);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.v[4 * gk], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.v[4 * gk + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.v[4 * gk + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.v[4 * gk + 3], acc, 0, 0, 0);
    }
    y.v[4 * g] = acc[0]; y.v[4 * g + 1] = acc[1]; y.v[4 * g + 2] = acc[2]; y.v[4 * g + 3] = acc[3];
  }
  return y;
}
__device__ __forceinline__ void load_w64(float4 (&w)[4][4], const float* __restrict__ Wt, int m, int kq) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) w[g][gk] = ld4g(Wt + (long)(16 * g + m) * AE + 16 * gk + 4 * kq);
}

// Rows past the end shadow the last valid row (same inputs, same results, same addresses): every store is unconditional, so the
// wait for the prefetched rows at the top of a tile is vmcnt(#stores of the tile) and not a drain of the store queue.
template <int TAIL>
__global__ __launch_bounds__(64, 1) void k_seg_post(SegArgs a) {
  const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
  const long ntiles = (a.R + 15) >> 4;
  float4 wo[4][4], w0[4][4], w0b[TAIL != 0 ? 4 : 1][4], w0c[TAIL == 2 ? 4 : 1][4];
  load_w64(wo, a.wo_t, m, kq);
  if (TAIL == 1 || TAIL == 3) load_w64(w0, a.w0_t, m, kq);
  if constexpr (TAIL == 1) { if (a.nq2 > 0) load_w64(w0b, a.q2_t[0], m, kq); }
  Row b1r;
  if constexpr (TAIL == 3) {   // logit head weights and bias resident as well (columns >= K of W1^T are zero rows, the bias is masked)
    load_w64(w0b, a.w1_t, m, kq);
#pragma unroll
    for (int j = 0; j < 16; ++j) { const int n = 16 * (j >> 2) + 4 * kq + (j & 3); b1r.v[j] = n < a.K ? a.b1[n] : 0.f; }
  }   // first decoder block's query projection stays resident too
  if (TAIL == 2) {   // the 64 -> 192 projection as three register-resident 64 -> 64 blocks (k | v | g)
    load_w64(w0, a.w0_t, m, kq);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int gk = 0; gk < 4; ++gk) {
        w0b[g][gk] = ld4g(a.w0_t + (long)(64 + 16 * g + m) * AE + 16 * gk + 4 * kq);
        w0c[g][gk] = ld4g(a.w0_t + (long)(128 + 16 * g + m) * AE + 16 * gk + 4 * kq);
      }
  }
  const Row gam = row_load(a.gamma, kq), bet = row_load(a.beta, kq);
  const Row s1 = row_load(a.s1, kq);
  // every per-feature parameter row of the segment is loaded once (a load inside the tile loop is a dependent L1/L2 round trip)
  Row s2r, hsr, hwr, b0r;
  if (a.s2) s2r = row_load(a.s2, kq);
  if (TAIL == 1 || TAIL == 3) { hsr = row_load(a.hs, kq); b0r = row_load(a.b0, kq); }
  if (TAIL == 1) hwr = row_load(a.hw, kq);
  const float hb1 = TAIL == 1 ? a.hb1[0] : 0.f;
  // prefetch of the next tile's three input rows (rows past the end shadow the last row and are never stored)
  auto rowc = [&](long tile) { return min(tile * 16 + m, a.R - 1); };
  long tile = blockIdx.x;
  Row nr, ng, nres, npe;
  const bool want_pe = a.ope || TAIL == 1 || TAIL == 2;
  auto pe_row = [&](long row) {
    int p = a.pos[row];
    p = p < 0 ? 0 : (p >= a.npos ? a.npos - 1 : p);
    return row_load(a.pe + (long)p * AE, kq);
  };
  auto src_row = [&](long row) { return a.rows ? (long)a.rows[row] : row; };   // an index load a whole tile ahead, like pe_row's
  {
    const long row = rowc(tile), src = src_row(row);
    nr = row_load(a.r + row * AE, kq); ng = row_load(a.gp + src * a.ldg, kq); nres = row_load(a.res + src * AE, kq);
    if (want_pe) npe = pe_row(row);
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const long row = tile * 16 + m;
    const long rw = min(row, a.R - 1);
    const Row r = nr, g = ng, res = nres, per = npe;
    {
      const long nrow = rowc(min(tile + (long)gridDim.x, ntiles - 1)), nsrc = src_row(nrow);
      nr = row_load(a.r + nrow * AE, kq); ng = row_load(a.gp + nsrc * a.ldg, kq); nres = row_load(a.res + nsrc * AE, kq);
      if (want_pe) npe = pe_row(nrow);   // position -> table row: two dependent loads, issued a whole tile ahead
    }
    // GroupNorm over the whole 64-wide row (n_head = 1) and the swish gate
    Row sq;
#pragma unroll
    for (int j = 0; j < 16; ++j) sq.v[j] = r.v[j] * r.v[j];
    const float mu = row_sum(r) * (1.0f / 64.0f), m2 = row_sum(sq) * (1.0f / 64.0f);
    const float rstd = rsqrtf(fmaxf(m2 - mu * mu, 0.f) + EPSN);
    Row u;
#pragma unroll
    for (int j = 0; j < 16; ++j) u.v[j] = fswish(g.v[j]) * ((r.v[j] - mu) * rstd * gam.v[j] + bet.v[j]);
    row_store(a.u + rw * AE, kq, u);
    const Row y = dense64_reg(u, wo, nullptr, kq);
    if (a.y) row_store(a.y + rw * AE, kq, y);   // (not kept for the fused backward, which recomputes it from r and gp: k_seg_bwd)
    // o = rms(res + y) s1 [-> rms s2]
    Row o = row_add(res, y);
    {
      Row q;
#pragma unroll
      for (int j = 0; j < 16; ++j) q.v[j] = o.v[j] * o.v[j];
      const float rs = rsqrtf(row_sum(q) * (1.0f / 64.0f) + EPSN);
#pragma unroll
      for (int j = 0; j < 16; ++j) o.v[j] = o.v[j] * rs * s1.v[j];
    }
    if (a.s2) o = row_rms_reg(o, s2r);
    if (a.o) row_store(a.o + rw * AE, kq, o);
    Row ope = o;
    if (want_pe) {
      ope = row_add(o, per);
      if (a.ope) row_store(a.ope + rw * AE, kq, ope);
    }
    if (TAIL == 1) {          // encoder: value head + cross-retention queries
      const Row hv = row_add(dense64_reg(o, w0, nullptr, kq), b0r);
      row_store(a.out0 + rw * a.ld0, kq, hv);
      const Row hn = row_rms_reg(row_gelu(hv), hsr);
      Row hw;
#pragma unroll
      for (int j = 0; j < 16; ++j) hw.v[j] = hn.v[j] * hwr.v[j];
      const float val = row_sum(hw) + hb1;
      a.value[rw] = val;
      if constexpr (TAIL == 1) {
        if (a.nq2 > 0) row_store(a.q2[0] + rw * AE, kq, dense64_reg(ope, w0b, nullptr, kq));
      }
      for (int k = 1; k < a.nq2; ++k) {
        const Row q2 = dense64(ope, a.q2_t[k], nullptr, m, kq);
        row_store(a.q2[k] + rw * AE, kq, q2);
      }
    } else if (TAIL == 2) {   // decoder, after the self-retention: k | v | g of the cross-retention
      float* orow = a.out0 + rw * a.ld0;
      if constexpr (TAIL == 2) {
        row_store(orow, kq, dense64_reg(ope, w0, nullptr, kq));
        row_store(orow + 64, kq, dense64_reg(ope, w0b, nullptr, kq));
        row_store(orow + 128, kq, dense64_reg(ope, w0c, nullptr, kq));
      }
    } else if (TAIL == 3) {   // last decoder block: logit head
      const Row hp = row_add(dense64_reg(o, w0, nullptr, kq), b0r);
      row_store(a.out0 + rw * a.ld0, kq, hp);
      const Row hn = row_rms_reg(row_gelu(hp), hsr);
      row_store(a.hn + rw * AE, kq, hn);
      if constexpr (TAIL == 3) {
        Row lg = row_add(dense64_reg(hn, w0b, nullptr, kq), b1r);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int n = 16 * (j >> 2) + 4 * kq + (j & 3); lg.v[j] = n < a.K ? lg.v[j] : 0.f; }
        row_store(a.logits + rw * AE, kq, lg);
      }
    }
  }
}


// ---- backward of the front of a post-retention segment (all three sites of a block share it) -------------------------
//   dsum = d(res + y)  from  o = rms(rms(res + y) s1) s2   with incoming d0 (+ d1 + d2)          (k_resnorm_bwd)
//   du   = dsum W_o^T                                                                            (dense layer)
//   dr, dg from u = swish(g) * GroupNorm(r)                                                      (k_retpost_bwd)
// plus the four parameter-gradient rows (s1, s2, gamma, beta) as per-wave slabs.  Reads 5-7 rows, writes 3 (was 12-13 streams).
struct SegBwdArgs {
  long R;
  const float* a; const float* y; const float* s1; const float* s2;
  const float* d0; const float* d1; const float* d2;
  const float* wo_nat;                 // W_o [64 in][64 out] natural layout = "W^T" of du = dsum W_o^T
  const float* wo_t;                   // nullable: W_o^T as the forward holds it -- then y is not read but recomputed (y = u W_o, u from r and gp)
  const float* r; const float* gp; long ldg; const float* gamma; const float* beta;
  float* dsum; float* dr; float* dgp; long lddg;
  float* slab_s1; float* slab_s2; float* slab_ga; float* slab_be;   // [grid][64]
  const int* rows;   // nullable: a and gp are row tables, token row r reads table row rows[r]
};

template <bool RECOMP>
__global__ __launch_bounds__(64, 1) void k_seg_bwd(SegBwdArgs a) {
  const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
  const long ntiles = (a.R + 15) >> 4;
  float4 wo[4][4];
  __shared__ __align__(16) float wot[RECOMP ? 64 * WSP : 4];   // W_o^T for the recomputed y: in LDS, the register file is full (466 of 512)
  load_w64(wo, a.wo_nat, m, kq);
  // The forward's y = u W_o is recomputed here instead of being written there and read here: the kernel streams at the HBM rate with the
  // matrix cores at 0.16, a second 64 x 64 GEMM per row is free, 256 bytes per row on either side are not.  Same instructions on the same
  // operands as k_seg_post (fswish, the fused multiply-adds of the GroupNorm, dense64_reg on W_o^T): the same bits.
  constexpr bool recompute = RECOMP;
  if (recompute) {
    for (int i = lane; i < 64 * 16; i += 64) *reinterpret_cast<float4*>(&wot[(i >> 4) * WSP + 4 * (i & 15)]) = ld4g(a.wo_t + (i >> 4) * AE + 4 * (i & 15));
    __syncthreads();
  }
  // the four per-feature parameter rows: in registers, or (RECOMP: no registers left) in LDS and re-read where a tile uses them -- the
  // opaque offset keeps the compiler from hoisting those reads back out of the tile loop into 64 live registers
  __shared__ __align__(16) float prm[RECOMP ? 4 * 64 : 4];
  Row s1_, s2_, gam_, bet_;
  if (!RECOMP) {
    s1_ = row_load(a.s1, kq); gam_ = row_load(a.gamma, kq); bet_ = row_load(a.beta, kq);
    if (a.s2) s2_ = row_load(a.s2, kq);
  } else {
    prm[lane] = a.s1[lane]; prm[64 + lane] = a.s2 ? a.s2[lane] : 0.f; prm[128 + lane] = a.gamma[lane]; prm[192 + lane] = a.beta[lane];
    __syncthreads();
  }
  auto prow = [&](int which, const Row& reg) -> Row {
    if (!RECOMP) return reg;
    int off = which * 64 + 4 * kq;
    asm volatile("" : "+v"(off));
    Row r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t4 = *reinterpret_cast<const float4*>(&prm[off + 16 * g]);
      r.v[4 * g] = t4.x; r.v[4 * g + 1] = t4.y; r.v[4 * g + 2] = t4.z; r.v[4 * g + 3] = t4.w;
    }
    return r;
  };
  Row ds1, ds2, dga, dbe;
#pragma unroll
  for (int j = 0; j < 16; ++j) { ds1.v[j] = 0.f; ds2.v[j] = 0.f; dga.v[j] = 0.f; dbe.v[j] = 0.f; }
  auto rowc = [&](long tile) { return min(tile * 16 + m, a.R - 1); };
  Row na, ny, nd, nd1, nd2, nr, ng;   // next tile's rows: loads only here (the sums are formed when the tile is used)
  auto fetch = [&](long row) {
    const long src = a.rows ? (long)a.rows[row] : row;
    na = row_load(a.a + src * AE, kq);
    if (a.y && !recompute) ny = row_load(a.y + row * AE, kq);
    nd = row_load(a.d0 + row * AE, kq);
    if (a.d1) nd1 = row_load(a.d1 + row * AE, kq);
    if (a.d2) nd2 = row_load(a.d2 + row * AE, kq);
    nr = row_load(a.r + row * AE, kq);
    ng = row_load(a.gp + src * a.ldg, kq);
  };
  long tile = blockIdx.x;
  fetch(rowc(tile));
  for (; tile < ntiles; tile += gridDim.x) {
    const long row = tile * 16 + m;
    const long rw = min(row, a.R - 1);
    const float live = row < a.R ? 1.f : 0.f;   // shadow rows are stored (same values as the last row) but not accumulated
    Row x = (a.y && !recompute) ? row_add(na, ny) : na;
    Row d = nd;
    if (a.d1) d = row_add(d, nd1);
    if (a.d2) d = row_add(d, nd2);
    const Row r = nr, g = ng;
    fetch(rowc(min(tile + (long)gridDim.x, ntiles - 1)));
    // ---- GroupNorm + swish gate forward (their backward further down uses the same values)
    Row t;
#pragma unroll
    for (int j = 0; j < 16; ++j) t.v[j] = r.v[j] * r.v[j];
    const float mu = row_sum(r) * (1.0f / 64.0f), m2 = row_sum(t) * (1.0f / 64.0f);
    const float rstd = rsqrtf(fmaxf(m2 - mu * mu, 0.f) + EPSN);
    if (recompute) {
      const Row gam = prow(2, gam_), bet = prow(3, bet_);
      Row u;
#pragma unroll
      for (int j = 0; j < 16; ++j) u.v[j] = fswish(g.v[j]) * ((r.v[j] - mu) * rstd * gam.v[j] + bet.v[j]);
      x = row_add(x, dense64_lds(u, wot, m, kq));
    }
    // ---- residual + RMSNorm(s) backward
#pragma unroll
    for (int j = 0; j < 16; ++j) t.v[j] = x.v[j] * x.v[j];
    const float rstd1 = rsqrtf(row_sum(t) * (1.0f / 64.0f) + EPSN);
    Row xh1;
#pragma unroll
    for (int j = 0; j < 16; ++j) xh1.v[j] = x.v[j] * rstd1;
    const Row s1 = prow(0, s1_);
    if (a.s2) {
      const Row s2 = prow(1, s2_);
      Row y1;
#pragma unroll
      for (int j = 0; j < 16; ++j) { y1.v[j] = xh1.v[j] * s1.v[j]; t.v[j] = y1.v[j] * y1.v[j]; }
      const float rstd2 = rsqrtf(row_sum(t) * (1.0f / 64.0f) + EPSN);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float xh2 = y1.v[j] * rstd2;
        ds2.v[j] += live * d.v[j] * xh2;
        y1.v[j] = xh2;                       // y1 <- xhat2
        d.v[j] = d.v[j] * s2.v[j];           // d <- g2
        t.v[j] = d.v[j] * xh2;
      }
      const float dot2 = row_sum(t) * (1.0f / 64.0f);
#pragma unroll
      for (int j = 0; j < 16; ++j) d.v[j] = (d.v[j] - y1.v[j] * dot2) * rstd2;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      ds1.v[j] += live * d.v[j] * xh1.v[j];
      d.v[j] = d.v[j] * s1.v[j];             // d <- g1
      t.v[j] = d.v[j] * xh1.v[j];
    }
    const float dot1 = row_sum(t) * (1.0f / 64.0f);
    Row dsum;
#pragma unroll
    for (int j = 0; j < 16; ++j) dsum.v[j] = (d.v[j] - xh1.v[j] * dot1) * rstd1;
    row_store(a.dsum + rw * AE, kq, dsum);
    // ---- du = dsum W_o^T, then GroupNorm + swish gate backward
    const Row du = dense64_reg(dsum, wo, nullptr, kq);
    const Row gam = prow(2, gam_), bet = prow(3, bet_);
    Row xh, gg, dg;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      xh.v[j] = (r.v[j] - mu) * rstd;
      const float rn = xh.v[j] * gam.v[j] + bet.v[j];
      const float sg = fast_sigmoid(g.v[j]);
      const float sw = g.v[j] * sg;                                  // swish
      const float swg = sg * (1.0f + g.v[j] * (1.0f - sg));          // d swish / dg
      const float drn = du.v[j] * sw;
      dg.v[j] = du.v[j] * rn * swg;
      dga.v[j] += live * drn * xh.v[j];
      dbe.v[j] += live * drn;
      gg.v[j] = drn * gam.v[j];
      t.v[j] = gg.v[j] * xh.v[j];
    }
    const float mg = row_sum(gg) * (1.0f / 64.0f), mgx = row_sum(t) * (1.0f / 64.0f);
    Row dx;
#pragma unroll
    for (int j = 0; j < 16; ++j) dx.v[j] = (gg.v[j] - mg - xh.v[j] * mgx) * rstd;
    row_store(a.dr + rw * AE, kq, dx);
    row_store(a.dgp + rw * a.lddg, kq, dg);
  }
  // per-wave parameter-gradient rows: sum over the 16 row slots of the wave (DPP all-reduce inside each 16-lane row), one slab row each
#pragma unroll
  for (int j = 0; j < 16; ++j) { ds1.v[j] = gsum(ds1.v[j], 16); ds2.v[j] = gsum(ds2.v[j], 16); dga.v[j] = gsum(dga.v[j], 16); dbe.v[j] = gsum(dbe.v[j], 16); }
  if (m == 0) {
    row_store(a.slab_s1 + (long)blockIdx.x * AE, kq, ds1);
    if (a.s2) row_store(a.slab_s2 + (long)blockIdx.x * AE, kq, ds2);
    row_store(a.slab_ga + (long)blockIdx.x * AE, kq, dga);
    row_store(a.slab_be + (long)blockIdx.x * AE, kq, dbe);
  }
}

}  // namespace magpo

using namespace magpo;

// ptrs_host (device pointers, host array of 34): r gp gamma beta wo_t res s1 s2 pe pos | u y o ope | w0_t b0 out0 | hs hw hb1 value |
//   q2_t[0..3] q2[0..3] | hn w1_t b1 logits | rows (NULL, or gp / res are row tables read through rows[r]).   dims_host[6] = {tail, K, npos, ldg, ld0, nq2}.
extern "C" int magpo_seg_post(const int* dims_host, long R, const void* const* p, int nptrs, hipStream_t st) {
  if (R <= 0) return MAGPO_OK;
  if (nptrs != 34) { set_error("magpo_seg_post: pointer table size mismatch"); return MAGPO_EINVAL; }
  SegArgs a;
  a.tail = dims_host[0]; a.K = dims_host[1]; a.npos = dims_host[2]; a.ldg = dims_host[3]; a.ld0 = dims_host[4]; a.nq2 = dims_host[5];
  a.R = R;
  if (a.tail < 0 || a.tail > 3 || a.nq2 < 0 || a.nq2 > 4 || a.K < 1 || a.K > 64) { set_error("magpo_seg_post: bad arguments"); return MAGPO_EINVAL; }
  int i = 0;
  a.r = (const float*)p[i++]; a.gp = (const float*)p[i++]; a.gamma = (const float*)p[i++]; a.beta = (const float*)p[i++];
  a.wo_t = (const float*)p[i++]; a.res = (const float*)p[i++]; a.s1 = (const float*)p[i++]; a.s2 = (const float*)p[i++];
  a.pe = (const float*)p[i++]; a.pos = (const int*)p[i++];
  a.u = (float*)p[i++]; a.y = (float*)p[i++]; a.o = (float*)p[i++]; a.ope = (float*)p[i++];
  a.w0_t = (const float*)p[i++]; a.b0 = (const float*)p[i++]; a.out0 = (float*)p[i++];
  a.hs = (const float*)p[i++]; a.hw = (const float*)p[i++]; a.hb1 = (const float*)p[i++]; a.value = (float*)p[i++];
  for (int k = 0; k < 4; ++k) a.q2_t[k] = (const float*)p[i++];
  for (int k = 0; k < 4; ++k) a.q2[k] = (float*)p[i++];
  a.hn = (float*)p[i++]; a.w1_t = (const float*)p[i++]; a.b1 = (const float*)p[i++]; a.logits = (float*)p[i++];
  a.rows = (const int*)p[i++];
  const long ntiles = (R + 15) / 16;
  long grid = 256 * 4;   // one wave per SIMD (the weights of the segment live in its registers)
  if (grid > ntiles) grid = ntiles;
  switch (a.tail) {
    case 0: hipLaunchKernelGGL(k_seg_post<0>, dim3((unsigned)grid), dim3(64), 0, st, a); break;
    case 1: hipLaunchKernelGGL(k_seg_post<1>, dim3((unsigned)grid), dim3(64), 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_seg_post<2>, dim3((unsigned)grid), dim3(64), 0, st, a); break;
    default: hipLaunchKernelGGL(k_seg_post<3>, dim3((unsigned)grid), dim3(64), 0, st, a); break;
  }
  return check_launch("magpo_seg_post");
}

// Number of slab rows magpo_seg_bwd writes for R rows (= its grid size).
extern "C" int magpo_seg_bwd_grid(long R) { const long nt = (R + 15) / 16; return (int)(nt < 1024 ? (nt < 1 ? 1 : nt) : 1024); }

// ptrs_host[20] (device pointers): a y s1 s2 d0 d1 d2 wo_nat r gp gamma beta | dsum dr dgp | slab_s1 slab_s2 slab_ga slab_be | rows
// (y, s2, d1, d2, slab_s2, rows may be NULL; with rows, a and gp are row tables read through rows[r]); ldg / lddg: row strides of gp / dgp.  Slabs: [magpo_seg_bwd_grid(R)][64].
extern "C" int magpo_seg_bwd(long R, long ldg, long lddg, const void* const* p, int nptrs, hipStream_t st) {
  if (R <= 0) return MAGPO_OK;
  if (nptrs != 21) { set_error("magpo_seg_bwd: pointer table size mismatch"); return MAGPO_EINVAL; }
  SegBwdArgs a;
  a.R = R; a.ldg = ldg; a.lddg = lddg;
  int i = 0;
  a.a = (const float*)p[i++]; a.y = (const float*)p[i++]; a.s1 = (const float*)p[i++]; a.s2 = (const float*)p[i++];
  a.d0 = (const float*)p[i++]; a.d1 = (const float*)p[i++]; a.d2 = (const float*)p[i++]; a.wo_nat = (const float*)p[i++];
  a.r = (const float*)p[i++]; a.gp = (const float*)p[i++]; a.gamma = (const float*)p[i++]; a.beta = (const float*)p[i++];
  a.dsum = (float*)p[i++]; a.dr = (float*)p[i++]; a.dgp = (float*)p[i++];
  a.slab_s1 = (float*)p[i++]; a.slab_s2 = (float*)p[i++]; a.slab_ga = (float*)p[i++]; a.slab_be = (float*)p[i++];
  a.rows = (const int*)p[i++];
  a.wo_t = (const float*)p[i++];
  if (a.wo_t) hipLaunchKernelGGL(k_seg_bwd<true>, dim3((unsigned)magpo_seg_bwd_grid(R)), dim3(64), 0, st, a);
  else hipLaunchKernelGGL(k_seg_bwd<false>, dim3((unsigned)magpo_seg_bwd_grid(R)), dim3(64), 0, st, a);
  return check_launch("magpo_seg_bwd");
}
