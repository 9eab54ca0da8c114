// Fused Sable acting step (SableNetwork.get_actions, mava/networks/sable_network.py:443-482) for gfx950:
// ONE launch per environment step, WAVE-AUTONOMOUS: a single-wave workgroup owns EPW environments and carries them
// through the encoder (all A tokens), the A autoregressive decoder iterations (decode.py:111-153) and the categorical
// sampling with no workgroup barrier at all.  The rollout is a chain of dependent small ops, so what matters is the
// number of dependent memory round trips per step, not bandwidth or flops (measured: ~2 us per dependent global
// access, in-kernel stage timing, profiles/):
//   * activations live in REGISTERS in a feature-major layout -- lane (env = l & 15, kq = l >> 4) holds the 16 features
//     n = 16 g + 4 kq + r (g, r in 0..3) of one env's 64-wide row.  A dense layer is computed transposed,
//     Y^T[n][env] = sum_k Wt[n][k] X^T[k][env], on v_mfma_f32_16x16x4_f32 with W as the A operand (float4 fragments
//     straight from L2, prefetched 4 column groups ahead) and the activation registers as the B operand; the
//     accumulator comes out in the same feature-major layout, so dense layers, RMSNorm (16 in-lane adds + 2
//     xor-shuffles), GELU, residuals and positional encodings chain without touching LDS or memory;
//   * the retention step works on one env at a time with the whole 64x64 state in the wave's registers (lane ->
//     columns c4..c4+3 of rows 16 rg .. 16 rg + 15), with the next states prefetched several deep; q/k/v/g rows reach
//     the state layout through a small per-wave LDS tile; GroupNorm + swish gate are fused behind it;
//   * sampling: the logits never leave registers; gumbel noise from the JAX threefry stream (rl.hip: k_sample).
// Only what later agents / the training pass need goes to (L2-resident) scratch: this step's k, v rows, obs_rep, q2.
//
// HBM traffic = the retention states, so every state is read ONCE and written ONCE per env step:
//   * encoder state: one pass (all A tokens are known up front);
//   * decoder states: the update S <- kappa S + sum_a k_a^T v_a of step t needs all A decoded tokens, the outputs of step t need
//     kappa S while the tokens are still being decoded.  The update is therefore DEFERRED to the next launch: memory holds the state
//     that ENTERED the last step plus that step's k | v rows (the scratch rows blk[].qkvg1 / blk[].kvg2 persist between launches);
//     the next launch's pre-pass loads the state once, adds the pending rows, applies the episode-end zeroing, writes it back and
//     computes from registers everything the decoder needs from kappa S:
//       - cross-retention: q2_a (kappa S) for all agents (the query is the encoder's, known before the decoder starts);
//       - self-retention, block 0, one head: q_c (kappa S) for EVERY candidate previous action c (the block-0 query is a function of
//         (previous action, step count) only: q = x_c W_q + pe W_q) -- a (K + 2) x 64 x 64 product per env on the otherwise idle
//         16x16x4 fp32 MFMA with the state registers as B operand, instead of re-reading the state once per agent;
//     the decoder iterations then only add the intra-step rank-1 terms in registers (cross_ret).  `flush` (the last launch of a
//     rollout, or a stand-alone step) adds the rows of the current launch so that memory holds the true carried states again.
//     Blocks > 0 and n_head > 1 keep the per-agent state pass (ret_pass mode 1) for the self-retention.
#pragma once
#include "fm_rows.hpp"
#include <stdlib.h>
#include <string.h>

namespace magpo {

constexpr float FMIN_ = -3.4028234663852886e38f;
constexpr int MAXB = 4;          // max blocks
constexpr int MAXA = 8;          // max agents of the fused path (token staging registers)
constexpr int QP = 272;          // LDS pitch of a [q|k|v|g] token row: 16 mod 64 -> the (env, kq) float4 pattern is conflict-minimal
constexpr int UP = 80;
           // LDS pitch of a 64-wide row

// Timing experiments only (scripts/debug/act_ab.sh; results are WRONG with them): -DMAGPO_ACT_X_NOLOAD replaces every retention-state load
// by a constant, -DMAGPO_ACT_X_NOSTORE drops the state stores -- what is left is the kernel's compute + scratch-row time.
#ifdef MAGPO_ACT_X_NOLOAD
#define XLOAD(p) make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f)
#else
#define XLOAD(p) ld4nt(p)
#endif
#ifdef MAGPO_ACT_X_NOSTORE
#define XSTORE(p, v) do { if ((v).x == 123.456f) st4nt(p, v); } while (0)
#else
#define XSTORE(p, v) st4nt(p, v)
#endif

struct ActBlk {
  const float *qkvg_t, *wo_t, *ln1, *ln2, *gn_g, *gn_b;                                     // encoder block
  const float *qkvg1_t, *wo1_t, *dln1, *gn1_g, *gn1_b;                                       // decoder self-retention
  const float *q2_t, *kvg2_t, *wo2_t, *dln2, *dln3, *gn2_g, *gn2_b;                           // decoder cross-retention
  float *qkvg1, *q2, *kvg2;                                                                   // scratch [N*A][256|64|192]
};
struct ActArgs {
  int N, A, K, F, nb, nh, hs, gs, npos, value_only, ldo;   // ldo = floats between observation rows (>= F: wide observations are padded)
  // Deferred decoder-state updates (see the header comment): pending = the k | v rows the PREVIOUS launch left in the scratch rows
  // (blk[].qkvg1 / blk[].kvg2) have not been added to S_d1 / S_d2 yet; flush = add this launch's (or, for a value-only launch, the
  // pending) rows before returning, so that the states in memory are the true carried states again.
  int pending, flush;
  // Deferred candidate pass (block 0, one head): defer = the deferring waves run the candidate pre-pass of the NEXT step at the end of this
  // launch (rows of this step applied, positional row of step count + 1, no episode-end zeroing yet); precand = the previous launch did
  // that, so those waves find S_d1 up to date and their candidate table built (envs whose episode ended in between are fixed up).
  int precand, defer;
  float* ptab;              // [N][K + 2][64] block-0 self-retention: q_c (kappa S) for every candidate previous action c, row K + 1 = the positional part
  const float* obs; const int* pos; const unsigned char* mask; const uint32_t* keys_dev; uint32_t keys[16][2];
  const float *s_obs, *W_obs, *s_encln, *W_act, *s_decln;
  const float *vh0_t, *vh0_b, *vh_s, *vh_w, *vh_b1;
  const float *h0_t, *h0_b, *h_s, *h1_t, *h1_b;
  const float* pe;
  float kappa[4];
  ActBlk blk[MAXB];
  float *S_enc, *S_d1, *S_d2;     // [nb][nh][N][4096]
  float *xn; const unsigned char* done; float *qkvg, *u, *y, *rep, *reppe, *hv;    // scratch [N*A][64 | 256]  (used: xn, qkvg, u, rep); done [N] or NULL
  float *xa, *kin1, *y1, *c, *cpe, *y2, *xo, *xope, *hp, *hn, *logits, *u1, *u2; int* prev;   // unused by this kernel (table layout kept)
  int* action; float* logp; float* value;
};

#ifdef MAGPO_ACT_PROF
__device__ unsigned long long g_act_prof[32];
#define RT_DECL() unsigned long long rt_acc[5] = {0, 0, 0, 0, 0}; unsigned long long rt_last = clock64();
#define RT(k) do { unsigned long long t_ = clock64(); rt_acc[k] += t_ - rt_last; rt_last = t_; } while (0)
#define RT_FLUSH() do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { for (int k_ = 0; k_ < 5; ++k_) atomicAdd(&g_act_prof[8 + k_], rt_acc[k_]); } } while (0)
#else
#define RT_DECL()
#define RT(k)
#define RT_FLUSH()
#endif

// ---- recurrent retention over the wave's envs, one (env, head) state at a time ------------------------------------
//   S_eff = kappa * S + sum_{a < ntok} k_a^T v_a ; u_a = swish(g_a) * GroupNorm(q_a S_eff) for a in [ret_from, ntok)
// ENC: ntok = A, all tokens staged from the global [q|k|v|g] rows `hist`; u rows -> global uout[(env*A + a)*64].
// DEC: ntok = i + 1, tokens a < i staged from the global k|v history (hist rows, columns hcol..hcol+127), token i read
//      from the wave's TQ tile; u -> LDS tile U[env].
// The first state of a pass can be PRIMED: its loads are issued by the caller before the dense phase that precedes the pass
// (prime_state), so that the memory system also has work while the wave runs MFMA / VALU code (one wave per SIMD: nothing else hides it).
__device__ __forceinline__ void prime_state(float4 (&dst)[16], const float* __restrict__ Se, int lane) {
  const int c4 = 4 * (lane & 15), rg = lane >> 4;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[r] = ld4nt(Se + (16 * rg + r) * 64 + c4);
}
__device__ __forceinline__ void prime_state_perm(float4 (&dst)[16], const float* __restrict__ Se, int lane) {   // row order of self_prepass_cand
  const int c4 = 4 * (lane & 15), kq = lane >> 4;
#pragma unroll
  for (int j = 0; j < 16; ++j) dst[j] = ld4nt(Se + (16 * (j >> 2) + 4 * kq + (j & 3)) * 64 + c4);
}

template <int MODE, int NA, int NBUF, int NH, bool PRIMED = false>
__device__ __forceinline__ void ret_pass(float* TQ, float* HK, float* U, float* __restrict__ S0 /* head 0 of this block */, long NS,
                                         const ActArgs& a, int env0, int nvalid, int i, const float* __restrict__ hist, long ldh,
                                         int hcol, float* __restrict__ uout, long ldu, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int write_state, unsigned long long dmask,
                                         const float* __restrict__ qsrc = nullptr, long ldq = 0, int apply_pending = 0,
                                         const float4* __restrict__ primed = nullptr) {
  const int lane = threadIdx.x, c4 = 4 * (lane & 15), rg = lane >> 4;
  const int A = a.A, nh = NH ? NH : a.nh, hs = NH ? AE / NH : a.hs, gs = NH ? AE / (NH * NH) : a.gs;   // NH = 0: run-time head count
  // (the team size as a compile-time constant -- token loops without uniform branches -- was measured in round 4: 484.6 vs 486.6 us, nothing)
  // MODE 0 (encoder): all A tokens staged as [q|k|v|g] rows, state update + write, gated output -> global uout
  // MODE 1 (decoder self-retention, agent i): tokens a < i staged (k|v), token i from TQ, output u_i -> LDS U, state written at the last agent
  // MODE 2 (cross-retention pre-pass): q rows of all A agents staged, RAW q_a (kappa S) -> global uout, state untouched
  // MODE 3 (flush): k|v rows of all A agents staged, state update + write, no output
  // MODE 4 (decoder pre-pass with queries): s = kappa s (+ the pending k|v rows of the previous launch); zero where the episode just
  //         ended; WRITE (the state that enters this step); then raw q_a (kappa s) -> global uout for the A staged query rows (qsrc)
  // MODE 5 (decoder pre-pass, no outputs): the same without queries
  constexpr bool ENC = MODE == 0;
  constexpr bool PRE = MODE == 4 || MODE == 5;
  constexpr bool DO_UPD = MODE != 2, DO_OUT = MODE != 3 && MODE != 5;
  const int ntok = MODE == 1 ? i + 1 : (PRE && !apply_pending ? 0 : A), ret_from = MODE == 1 ? i : 0, nstage = MODE == 1 ? i : A;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool colin = c4 < hs, rowin = 16 * rg < hs;
  const float inv_gs = 1.0f / (float)gs;
  const int cc = colin ? c4 : 0, rr = rowin ? 16 * rg : 0;   // clamped offsets for the unconditional LDS reads   // lane -> state columns c4..c4+3 of rows 16 rg .. 16 rg + 15
  float4 gam = z4, bet = z4;
  if (colin) { gam = ld4g(gamma + c4); bet = ld4g(beta + c4); }
  // NBUF state buffers rotate through the (env, head) pairs: NBUF - 1 states (16 KB each) are in flight behind the
  // one being used.  The pipeline body is branch-free (clamped prefetch indices, predicated stores) so that the
  // waits on the oldest loads leave the younger ones outstanding.
  // Issue order matters: vmcnt retires in order, so a pair's token rows are loaded right behind its state -- both are
  // first needed in the same iteration, and everything younger stays outstanding.
  float4 buf[NBUF][16], hreg[NBUF][NA];
  RT_DECL();
  const int npairs = nvalid * nh;
  auto prefetch = [&](float4 (&dst)[16], float4 (&tok)[NA], int pair, bool with_state = true) {
    pair = min(pair, npairs - 1);
    const int e = pair / nh, h = pair - e * nh;
    const float* Se = S0 + (long)h * NS + (long)(env0 + e) * 4096;
    if (with_state) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r] = XLOAD(Se + (16 * rg + r) * 64 + c4);
    }
    const long row0 = (long)(env0 + e) * A;
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      if (t < nstage) {
        if (ENC) tok[t] = ld4g(hist + (row0 + t) * ldh + 4 * lane);
        else if (MODE == 2) { if (lane < 16) tok[t] = ld4g(hist + (row0 + t) * ldh + 4 * lane); }
        else if (MODE == 4) {   // lanes 0-15: this step's query row; lanes 16-47: the pending k | v row of the previous launch
          if (lane < 16) tok[t] = ld4g(qsrc + (row0 + t) * ldq + 4 * lane);
          else if (lane < 48) tok[t] = ld4g(hist + (row0 + t) * ldh + hcol + 4 * (lane - 16));
        }
        else if (lane < 32) tok[t] = ld4g(hist + (row0 + t) * ldh + hcol + 4 * lane);
      }
    }
  };
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j) {
    if (PRIMED && j == 0) {   // pair 0's state was requested by the caller (prime_state); only its token rows are loaded here
#pragma unroll
      for (int r = 0; r < 16; ++r) buf[0][r] = primed[r];
      prefetch(buf[0], hreg[0], 0, false);
    } else {
      prefetch(buf[j], hreg[j], j);
    }
  }
  for (int base = 0; base < npairs; base += NBUF) {
#pragma unroll
    for (int j = 0; j < NBUF; ++j) {
      const int pair = base + j;
      const bool live = pair < npairs;          // tail slots recompute the last pair and store nothing
      const int pc = min(pair, npairs - 1);
      const int e = pc / nh, h = pc - e * nh, o = h * hs;
      const long row0 = (long)(env0 + e) * A;
      float* Se = S0 + (long)h * NS + (long)(env0 + e) * 4096;
      float4 (&s)[16] = buf[j];
      RT(0);
      prefetch(buf[(j + NBUF - 1) % NBUF], hreg[(j + NBUF - 1) % NBUF], pair + NBUF - 1);
      // stage this env's token rows (rewritten per head: keeps the body branch-free)
      __builtin_amdgcn_wave_barrier();   // reads of the previous pair's tokens are done (in-order DS ops of one wave)
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        if (t < nstage) {
          if (ENC) *reinterpret_cast<float4*>(HK + t * QP + 4 * lane) = hreg[j][t];
          else if (MODE == 2) { if (lane < 16) *reinterpret_cast<float4*>(HK + t * QP + 4 * lane) = hreg[j][t]; }
          else if (MODE == 4) { if (lane < 48) *reinterpret_cast<float4*>(HK + t * QP + 4 * lane) = hreg[j][t]; }   // q at 0, k | v at 64
          else if (lane < 32) *reinterpret_cast<float4*>(HK + t * QP + 64 + 4 * lane) = hreg[j][t];
        }
      }
      lsync();   // LDS only: a workgroup-scope fence here would drain the prefetches (vmcnt(0))
      RT(1);
      // episode ended on the previous step: the carried state is zero (rec_magpo.py:164-169)
      // (pre-pass: the state in memory already carries the zeroing of ITS step; this step's zeroing follows the pending update)
      // Pre-pass without pending rows (first launch of a rollout, stand-alone step): memory already holds the carried state itself, so
      // there is nothing to bring up to date -- decaying it here would decay it twice (found in round 4: the seam of consecutive rollouts)
      const float decay = PRE ? (apply_pending ? a.kappa[h] : 1.0f) : (((dmask >> e) & 1ull) ? 0.f : a.kappa[h]);
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r].x *= decay; s[r].y *= decay; s[r].z *= decay; s[r].w *= decay; }
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        if (DO_UPD && t < ntok) {
          const float* tk = (MODE != 1 || t < i) ? HK + t * QP : TQ + e * QP;
          float4 vv = *reinterpret_cast<const float4*>(tk + 128 + o + cc);   // unconditional loads (a predicated load becomes
          if (!colin) vv = z4;                                                  // an exec-masked block with its own LDS wait)
          float kk[16];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float4 k4 = *reinterpret_cast<const float4*>(tk + 64 + o + rr + 4 * r4);
            if (!rowin) k4 = z4;
            kk[4 * r4] = k4.x; kk[4 * r4 + 1] = k4.y; kk[4 * r4 + 2] = k4.z; kk[4 * r4 + 3] = k4.w;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r].x += kk[r] * vv.x; s[r].y += kk[r] * vv.y; s[r].z += kk[r] * vv.z; s[r].w += kk[r] * vv.w; }
        }
      }
      RT(2);
      if (PRE) {   // episode ended on the previous step: the state that enters this step is zero (rec_magpo.py:164-169)
        const float keep = ((dmask >> e) & 1ull) ? 0.f : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r].x *= keep; s[r].y *= keep; s[r].z *= keep; s[r].w *= keep; }
      }
      if (write_state && live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) XSTORE(Se + (16 * rg + r) * 64 + c4, s[r]);
      }
      if (MODE == 4) {   // the outputs see kappa S
        const float kp = a.kappa[h];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r].x *= kp; s[r].y *= kp; s[r].z *= kp; s[r].w *= kp; }
      }
      RT(3);
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        if (!DO_OUT || t < ret_from || t >= (MODE == 4 ? A : ntok)) continue;
        const float* tk = (MODE != 1 || t < i) ? HK + t * QP : TQ + e * QP;
        float4 p = z4;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          float4 q4 = *reinterpret_cast<const float4*>(tk + o + rr + 4 * r4);
          if (!rowin) q4 = z4;
          p.x += q4.x * s[4 * r4].x; p.y += q4.x * s[4 * r4].y; p.z += q4.x * s[4 * r4].z; p.w += q4.x * s[4 * r4].w;
          p.x += q4.y * s[4 * r4 + 1].x; p.y += q4.y * s[4 * r4 + 1].y; p.z += q4.y * s[4 * r4 + 1].z; p.w += q4.y * s[4 * r4 + 1].w;
          p.x += q4.z * s[4 * r4 + 2].x; p.y += q4.z * s[4 * r4 + 2].y; p.z += q4.z * s[4 * r4 + 2].z; p.w += q4.z * s[4 * r4 + 2].w;
          p.x += q4.w * s[4 * r4 + 3].x; p.y += q4.w * s[4 * r4 + 3].y; p.z += q4.w * s[4 * r4 + 3].z; p.w += q4.w * s[4 * r4 + 3].w;
        }
        p.x = xsum32(xsum16(p.x)); p.y = xsum32(xsum16(p.y)); p.z = xsum32(xsum16(p.z)); p.w = xsum32(xsum16(p.w));
        // fused epilogue (retention.py:289-294): GroupNorm over groups of gs channels, then the swish gate
        float s1 = (p.x + p.y) + (p.z + p.w), s2 = (p.x * p.x + p.y * p.y) + (p.z * p.z + p.w * p.w);
        s1 = gsum(s1, gs >> 2); s2 = gsum(s2, gs >> 2);
        const float mu = s1 * inv_gs, m2 = s2 * inv_gs;
        const float rstd = rsqrtf(fmaxf(m2 - mu * mu, 0.f) + EPSN);
        if (MODE == 2 || MODE == 4) {   // raw q (kappa S): the intra-step terms, GroupNorm and gate follow in registers (cross_ret)
          if (colin && rg == 0 && live) st4g(uout + (row0 + t) * ldu + o + c4, p);
          continue;
        }
        const float4 g4 = *reinterpret_cast<const float4*>(tk + 192 + o + cc);
        if (colin && rg == 0 && live) {
          float4 o4;
          o4.x = fswish(g4.x) * ((p.x - mu) * rstd * gam.x + bet.x);
          o4.y = fswish(g4.y) * ((p.y - mu) * rstd * gam.y + bet.y);
          o4.z = fswish(g4.z) * ((p.z - mu) * rstd * gam.z + bet.z);
          o4.w = fswish(g4.w) * ((p.w - mu) * rstd * gam.w + bet.w);
          if (ENC) st4g(uout + (row0 + t) * ldu + o + c4, o4);
          else *reinterpret_cast<float4*>(U + e * UP + o + c4) = o4;
        }
      }
      RT(4);
    }
  }
  RT_FLUSH();
}

// ---- decoder self-retention pre-pass, block 0, one head: the candidate table ---------------------------------------------------
// For every env of the wave: S <- kappa S (+ pending k_a^T v_a of the previous launch), zero where the episode just ended, WRITE;
// then P[c] = q_c (kappa S) for the K + 1 candidate previous actions c (q_c = x_c W_q, x_c = rms(gelu(W_act[c])) s) and
// P[K + 1] = (pe W_q) (kappa S), the positional part of the query -- the block-0 query of agent i is q = x_prev W_q + pe W_q, so the
// decoder reads rows `prev` and K + 1 of its env's table instead of the state (ptab [N][K + 2][64]).
// The product runs on v_mfma_f32_16x16x4_f32: A = candidate rows (tile row m = l & 15 -> candidate 16 mt + m, features in the Row
// layout), B = the state registers.  For that the lane (col group n = l & 15, kq = l >> 4) holds state rows rho(j) = 16 (j >> 2) +
// 4 kq + (j & 3), j < 16 -- the Row feature order -- of columns 4 n .. 4 n + 3; k-step j multiplies feature rho(j) on both sides.
// Output tile (mt, nt): lane holds candidates 16 mt + 4 kq + i (i < 4) of column 4 n + nt, i.e. one float4 of row c per lane.
template <int NA, int MT, int NB>
__device__ __forceinline__ void self_prepass_cand(float* HK, float* PEQ, float* __restrict__ S0, const ActArgs& a, int env0, int nvalid,
                                                  const float* __restrict__ pend, const Row (&xq)[MT], unsigned long long dmask,
                                                  const float4* __restrict__ primed, bool apply_pending) {
  const int lane = threadIdx.x, n16 = lane & 15, c4 = 4 * n16, kq = lane >> 4, A = a.A;
  const float kappa = a.kappa[0];
  const int pe_row = a.K + 1, pe_mt = pe_row >> 4, pe_m = pe_row & 15;
  float4 buf[NB][16], hreg[NB][NA];   // NB states rotate: NB - 1 in flight behind the one in use
  auto prefetch = [&](float4 (&dst)[16], float4 (&tok)[NA], int e, bool with_state = true) {
    e = min(e, nvalid - 1);
    const float* Se = S0 + (long)(env0 + e) * 4096;
    if (with_state) {
#pragma unroll
      for (int j = 0; j < 16; ++j) dst[j] = XLOAD(Se + (16 * (j >> 2) + 4 * kq + (j & 3)) * 64 + c4);
    }
    const long row0 = (long)(env0 + e) * A;
#pragma unroll
    for (int t = 0; t < NA; ++t)
      if (t < A && lane < 32) tok[t] = ld4g(pend + (row0 + t) * 256 + 64 + 4 * lane);   // k | v of the previous launch (qkvg1 rows)
  };
#pragma unroll
  for (int j = 0; j < 16; ++j) buf[0][j] = primed[j];   // env 0's state: requested by the caller (prime_state_perm)
  prefetch(buf[0], hreg[0], 0, false);
#pragma unroll
  for (int j = 1; j < NB - 1; ++j) prefetch(buf[j], hreg[j], j);
  for (int base = 0; base < nvalid; base += NB) {
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      const int e = min(base + jb, nvalid - 1);
      const bool live = base + jb < nvalid;
      float4 (&s)[16] = buf[jb];
      prefetch(buf[(jb + NB - 1) % NB], hreg[(jb + NB - 1) % NB], base + jb + NB - 1);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < NA; ++t)
        if (t < A && lane < 32) *reinterpret_cast<float4*>(HK + t * QP + 64 + 4 * lane) = hreg[jb][t];
      lsync();
      const float kin = apply_pending ? kappa : 1.0f;   // no pending rows: memory holds the carried state itself (see ret_pass)
#pragma unroll
      for (int j = 0; j < 16; ++j) { s[j].x *= kin; s[j].y *= kin; s[j].z *= kin; s[j].w *= kin; }
      if (apply_pending) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
          if (t < A) {
            const float4 vv = *reinterpret_cast<const float4*>(HK + t * QP + 128 + c4);
            float kk[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 k4 = *reinterpret_cast<const float4*>(HK + t * QP + 64 + 16 * g + 4 * kq);
              kk[4 * g] = k4.x; kk[4 * g + 1] = k4.y; kk[4 * g + 2] = k4.z; kk[4 * g + 3] = k4.w;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) { s[j].x += kk[j] * vv.x; s[j].y += kk[j] * vv.y; s[j].z += kk[j] * vv.z; s[j].w += kk[j] * vv.w; }
          }
        }
      }
      const float keep = ((dmask >> e) & 1ull) ? 0.f : 1.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) { s[j].x *= keep; s[j].y *= keep; s[j].z *= keep; s[j].w *= keep; }
      if (live) {
        float* Se = S0 + (long)(env0 + e) * 4096;
#pragma unroll
        for (int j = 0; j < 16; ++j) XSTORE(Se + (16 * (j >> 2) + 4 * kq + (j & 3)) * 64 + c4, s[j]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) { s[j].x *= kappa; s[j].y *= kappa; s[j].z *= kappa; s[j].w *= kappa; }
      // positional query row of this env (features in the Row order of this lane's kq)
      float qp[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t4 = *reinterpret_cast<const float4*>(PEQ + e * 64 + 16 * g + 4 * kq);
        qp[4 * g] = t4.x; qp[4 * g + 1] = t4.y; qp[4 * g + 2] = t4.z; qp[4 * g + 3] = t4.w;
      }
      float* prow = a.ptab + (long)(env0 + e) * (a.K + 2) * 64;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool pe_lane = mt == pe_mt && n16 == pe_m;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float av = pe_lane ? qp[j] : xq[mt].v[j];
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s[j].x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s[j].y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s[j].z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s[j].w, acc[3], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 16 * mt + 4 * kq + i;
          if (live && c <= pe_row) st4g(prow + (long)c * 64 + c4, make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]));
        }
      }
    }
  }
}

// ---- cross-retention of agent i in registers (feature-major rows, all envs of the wave at once) --------------------------
//   r = P2_i + sum_{a <= i} (q_i . k_a)_head v_a ,  u = swish(g_i) * GroupNorm(r)
// P2_i = q_i (kappa S) comes from the pre-pass (the cross-retention query is the encoder's, so it is known for every agent
// before the decoder starts: the state is read once per step there and once more when it is updated after the last agent,
// instead of once per agent); tokens a < i are read back from the k|v history rows (ld 256: [k | v | - | P2]).
// hist (NA > 4 only): the k | v history rows of this env (token t at hist + t * 256, v at + 64) are then streamed through two row
// pairs (one in use, one in flight) instead of being held for all NA - 1 earlier agents at once: 14 rows = 224 VGPRs for 8-agent
// teams, which the register file does not have beside the state buffers (249 values went to scratch).
#ifndef MAGPO_ACT_STREAM4
#define MAGPO_ACT_STREAM4 0   // 1: the k | v history rows of the earlier agents are streamed for teams of <= 4 agents too (register relief)
#endif
template <int NH, int NA, bool STREAM = (NA > 4 || MAGPO_ACT_STREAM4)>
__device__ __forceinline__ Row cross_ret(const ActArgs& a, const Row& q, const Row& kc, const Row& vc, const Row& gc, const Row& p2,
                                         const Row (&hk)[NA - 1], const Row (&hv)[NA - 1], int i, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int kq, const float* __restrict__ hist = nullptr) {
  const int nh = NH ? NH : a.nh, hs = AE / nh;
  Row r = p2;
  Row nk, nv;   // STREAM: rows of token t, requested while token t - 1 is being used
  if (STREAM && i > 0) { nk = row_load(hist, kq); nv = row_load(hist + 64, kq); }
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    if (t > i) continue;
    Row kt = kc, vt = vc;   // (value selects: a select between references would pin the arrays in scratch)
    if (STREAM) {
      if (t < i) {
        kt = nk; vt = nv;
        if (t + 1 < i) { nk = row_load(hist + (long)(t + 1) * 256, kq); nv = row_load(hist + (long)(t + 1) * 256 + 64, kq); }
      }
    } else if (t < NA - 1) {
      const bool old = t < i;
#pragma unroll
      for (int j = 0; j < 16; ++j) { kt.v[j] = old ? hk[t < NA - 1 ? t : 0].v[j] : kc.v[j]; vt.v[j] = old ? hv[t < NA - 1 ? t : 0].v[j] : vc.v[j]; }
    }
    float part[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) d += q.v[4 * g + c] * kt.v[4 * g + c];
      part[g] = xsum32(xsum16(d));   // over the 4 k-quarter lanes of the env: features 16 g .. 16 g + 15
    }
    float coef[4];
    if (nh == 1) { const float d = (part[0] + part[1]) + (part[2] + part[3]); coef[0] = coef[1] = coef[2] = coef[3] = d; }
    else if (nh == 2) { coef[0] = coef[1] = part[0] + part[1]; coef[2] = coef[3] = part[2] + part[3]; }
    else { coef[0] = part[0]; coef[1] = part[1]; coef[2] = part[2]; coef[3] = part[3]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) r.v[j] += coef[j >> 2] * vt.v[j];
  }
  // GroupNorm over groups of gs = hs / nh consecutive channels (retention.py:289-294), then the swish gate
  float mu[4], rstd[4];
  if (nh == 1) {
    Row sq;
#pragma unroll
    for (int j = 0; j < 16; ++j) sq.v[j] = r.v[j] * r.v[j];
    const float m1 = row_sum(r) * (1.0f / 64.0f), m2 = row_sum(sq) * (1.0f / 64.0f);
    const float rs = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + EPSN);
#pragma unroll
    for (int g = 0; g < 4; ++g) { mu[g] = m1; rstd[g] = rs; }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s1 = (r.v[4 * g] + r.v[4 * g + 1]) + (r.v[4 * g + 2] + r.v[4 * g + 3]);
      float s2 = (r.v[4 * g] * r.v[4 * g] + r.v[4 * g + 1] * r.v[4 * g + 1]) + (r.v[4 * g + 2] * r.v[4 * g + 2] + r.v[4 * g + 3] * r.v[4 * g + 3]);
      float inv = 0.25f;
      if (nh == 2) { s1 = xsum32(xsum16(s1)); s2 = xsum32(xsum16(s2)); inv = 1.0f / 16.0f; }   // gs = 16: the 4 lanes of block g
      const float m1 = s1 * inv, m2 = s2 * inv;
      mu[g] = m1;
      rstd[g] = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + EPSN);
    }
  }
  Row u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int n = 16 * (j >> 2) + 4 * kq + (j & 3), c = n % hs;   // channel inside the head
    u.v[j] = fswish(gc.v[j]) * ((r.v[j] - mu[j >> 2]) * rstd[j >> 2] * gamma[c] + beta[c]);
  }
  return u;
}

// Optional in-kernel stage timing (debug builds only: -DMAGPO_ACT_PROF): wall-clock ticks (100 MHz) per stage, summed over every 64th wave:
// 0 encoder embedding + q|k|v|g GEMM, 1 encoder state pass, 2 encoder W_o / norms / value head / q2, 4 decoder pre-pass: candidate table,
// 19 decoder pre-pass: cross-retention states, 5 decoder: action embedding + q|k|v|g GEMM + self-retention terms, 6 W_o1 + norm + k|v|g GEMM +
// cross-retention terms, 7 W_o2 + norms + head, 16 sampling, 17 / 18 per-agent state pass of blocks > 0 (8..12: sub-stages of the state passes).
#ifdef MAGPO_ACT_PROF
#define PROF(k) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { unsigned long long t_ = wall_clock64(); \
  atomicAdd(&g_act_prof[k], t_ - t_last); t_last = t_; } } while (0)
#else
#define PROF(k)
#endif

// State buffers per wave at 16 envs per wave (the -D hook exists for scripts/debug/act_nbuf.sh).  Measured on MI355X, us per launch for
// 2 / 3 / 4 buffers: 698 / 769 / 815 (A = 4, one block, 16 384 envs), 2340 / 2863 / 4893 (A = 8, two blocks): the third buffer costs 80
// VGPRs, which pushes 94 values into scratch, and every scratch reload waits with vmcnt(0), i.e. drains the very prefetches the buffer
// was meant to keep in flight.
#ifndef MAGPO_ACT_NBUF16
#define MAGPO_ACT_NBUF16 2
#endif
// A/B hook (scripts/debug/act_ab.sh): 1 = the first state of a pass is requested ahead of the dense phase in front of it instead of at the
// start of the pass.  Measured on one box, 16 384 envs: 582 / 584 us with, 574 / 575 us without (4 096 envs: 280 vs 273) -- with 1 024
// independent waves the memory system already has work while a wave runs its dense phase; the extra live registers cost more.  Off.
#ifndef MAGPO_ACT_PRIME
#define MAGPO_ACT_PRIME 0
#endif
constexpr int ACT_NBUF16 = MAGPO_ACT_NBUF16;
#ifndef MAGPO_ACT_NBUF_ENC
#define MAGPO_ACT_NBUF_ENC MAGPO_ACT_NBUF16
#endif
#ifndef MAGPO_ACT_NBUF_PRE
#define MAGPO_ACT_NBUF_PRE MAGPO_ACT_NBUF16
#endif
#ifndef MAGPO_ACT_NBUF_CAND
#define MAGPO_ACT_NBUF_CAND 2
#endif

#ifndef MAGPO_ACT_DEFER
#define MAGPO_ACT_DEFER 1   // 1: the non-EARLY waves run the candidate pre-pass of the next step at the end of the launch (see defer_wave)
#endif
#ifndef MAGPO_ACT_STAGGER_MOD
#define MAGPO_ACT_STAGGER_MOD 2
#define MAGPO_ACT_STAGGER_EARLY 1
#endif
#ifndef MAGPO_ACT_STAGGER_SHIFT
#define MAGPO_ACT_STAGGER_SHIFT 3   // workgroups alternate between the two phase orders in groups of 2^shift
#endif
#ifndef MAGPO_ACT_STAGGER
#define MAGPO_ACT_STAGGER 1   // 1: the workgroups alternate between two phase orders (see stag / defer_wave in k_sable_act)
#endif
#ifndef MAGPO_ACT_WPE
#define MAGPO_ACT_WPE 1   // waves per SIMD the register allocation aims at (A/B hook)
#endif
template <int EPW, int NA, int NH>
__global__ __launch_bounds__(64, MAGPO_ACT_WPE) void k_sable_act(ActArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* TQ = smem;                  // [EPW][QP]  this iteration's [q|k|v|g] rows, one per env
  float* HK = TQ + EPW * QP;         // [A][QP]    staged token rows of the env being processed
  float* U = HK + a.A * QP;          // [EPW][UP]  gated retention output, one row per env
  float* XS = U + EPW * UP;          // [EPW][UP]  block input x parked across the self-retention (register relief)
  float* PEQ = XS + EPW * UP;        // [EPW][64]  pe W_q of every env (candidate pre-pass)
#ifdef MAGPO_ACT_PROF
  unsigned long long t_last = wall_clock64();
#endif
  const int lane = threadIdx.x, env = lane & 15, kq = lane >> 4, m = env;
  const int env0 = blockIdx.x * EPW;
  const int nvalid = min(EPW, a.N - env0);
  const bool valid = env < nvalid;
  const int le = valid ? env : 0;
  const long ge = env0 + le;   // invalid lanes shadow env 0 of the wave (reads stay legal, nothing is stored)
  const int A = a.A, nb = a.nb, nh_ = NH ? NH : a.nh;
  const long NS = (long)a.N * 4096;
  int p_ = a.pos[ge];
  p_ = p_ < 0 ? 0 : (p_ >= a.npos ? a.npos - 1 : p_);
  const Row pe = row_load(a.pe + (long)p_ * AE, kq);
  // envs whose episode ended on the previous step (bit e): one flag load per kernel, kept out of the retention pipeline
  const unsigned long long dmask = a.done ? __ballot(valid && kq == 0 && a.done[ge] != 0) : 0ull;

  // first state of the block-0 encoder pass: in flight while the token rows are computed
  float4 pS[16];
  if (MAGPO_ACT_PRIME) prime_state(pS, a.S_enc + (long)env0 * 4096, lane);

  // candidate path of the block-0 self-retention: one head (the state tile is the whole 64 x 64 matrix)
  const bool cand = nh_ == 1;
  // The candidate pre-pass (block-0 self-retention states: a third of the launch's state traffic) depends on the previous launch's pending
  // rows and on parameters only -- not on this step's encoder.  All waves start together and walk the same phases, so the chip alternates
  // between phases in which every wave streams states (HBM-bound, ~5 TB/s) and dense phases in which none does.  Two roles, alternating
  // in groups of 8 workgroups (STAGGER):
  //   EARLY waves run the candidate pre-pass FIRST, ahead of the encoder;
  //   DEFERRING waves run it LAST, for the NEXT step (a.defer): after the decoder the step's k | v rows are known, the next step count is
  //     this one + 1 unless the episode ends, and an ended episode means a zero state and a zero table -- fixed up by the next launch
  //     (a.precand) from its `done` flags.  They stream states while the EARLY waves decode, and decode while those stream.
  const bool stag = MAGPO_ACT_STAGGER && (int)((blockIdx.x >> MAGPO_ACT_STAGGER_SHIFT) % MAGPO_ACT_STAGGER_MOD) < MAGPO_ACT_STAGGER_EARLY;
  const bool defer_wave = cand && MAGPO_ACT_DEFER && MAGPO_ACT_STAGGER && !stag;
  // (a third role -- the pre-pass between the cross-state pre-pass and the decoder for every third group -- was measured: 498 vs 499 us)
  auto cand_pass = [&](const Row& pe_q, unsigned long long dm, bool apply) __attribute__((always_inline)) {
    const ActBlk& B = a.blk[0];
    float4 pS1[16];   // the first self-retention state of the candidate pass
    // candidate query rows q_c = x_c W_q (tile row = candidate) and the positional query rows pe W_q of this wave's envs
    const int ntile = (a.K + 2 + 15) >> 4;
#define CAND_ROWS(MT_)                                                                                                        \
    {                                                                                                                        \
      prime_state_perm(pS1, a.S_d1 + (long)env0 * 4096, lane);                                                               \
      Row xin_[MT_ + 1], xout_[MT_ + 1];   /* the candidate tiles and the positional row: one pass over the weight fragments */ \
      _Pragma("unroll") for (int mt = 0; mt < MT_; ++mt) {                                                                   \
        const int c = min(16 * mt + env, a.K);                                                                               \
        xin_[mt] = row_rms(row_gelu(row_load(a.W_act + (long)c * AE, kq)), a.s_decln, kq);                                   \
      }                                                                                                                      \
      xin_[MT_] = pe_q;                                                                                                      \
      dense64_multi<MT_ + 1, true>(xin_, MT_ + 1, xout_, B.qkvg1_t, nullptr, m, kq);                                         \
      row_store(PEQ + env * 64, kq, xout_[MT_]);                                                                             \
      lsync();                                                                                                               \
      Row xq[MT_];                                                                                                           \
      _Pragma("unroll") for (int mt = 0; mt < MT_; ++mt) xq[mt] = xout_[mt];                                                 \
      self_prepass_cand<NA, MT_, (EPW == 16 ? MAGPO_ACT_NBUF_CAND : 2)>(HK, PEQ, a.S_d1, a, env0, nvalid, B.qkvg1, xq, dm, pS1, apply);              \
    }
    if (ntile == 1) CAND_ROWS(1) else if (ntile == 2) CAND_ROWS(2) else CAND_ROWS(3)
#undef CAND_ROWS
  };
  if (cand && !a.value_only) {
    if (!(defer_wave && a.precand)) {   // EARLY waves; DEFERRING waves on the first launch of a rollout / a stand-alone step
      cand_pass(pe, dmask, a.pending != 0);
      wsync();
    } else if (dmask) {                 // the previous launch built state and table for "episode goes on": where it ended, the state is zero
      for (int e = 0; e < nvalid; ++e) {
        if (!((dmask >> e) & 1ull)) continue;
        float* Se = a.S_d1 + (long)(env0 + e) * 4096;
#pragma unroll
        for (int r = 0; r < 16; ++r) st4nt(Se + r * 256 + 4 * lane, make_float4(0.f, 0.f, 0.f, 0.f));
      }
    }
    PROF(4);
  }

  // ---------------- encoder over the A tokens of the step (act_encoder_fn, encode.py:58-84)
  for (int b = 0; b < nb; ++b) {
    const ActBlk& B = a.blk[b];
    // the A tokens of the step go through every layer TOGETHER, four at a time: each weight fragment is fetched once for the four
    // rows (wgemm_multi) -- the wave streams its weights from L2, and that stream, not the MFMAs, is what the dense phases wait for
    for (int t0 = 0; t0 < A; t0 += 4) {
      const int nt = min(4, A - t0);
      Row kin[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const long row = ge * A + min(t0 + tt, A - 1);   // rows past the team shadow its last token (nothing of them is stored)
        Row x;
        if (b == 0) {   // x = rms(gelu(rmsnorm_F(obs) * s_obs @ W_obs)) * s_encln   (sable_network.py:93-101,126,132)
          // rmsnorm_F(obs) * s_obs @ W_obs = rstd * sum_f (o_f s_f) W_obs[f]: one pass over the features, four at a time with their loads
          // issued together (a run-time loop of dependent scalar loads costs one memory round trip per feature on a lone wave)
          const float* o = a.obs + row * a.ldo;
          float ms = 0.f;
          Row z;
#pragma unroll
          for (int j = 0; j < 16; ++j) z.v[j] = 0.f;
          for (int f0 = 0; f0 < a.F; f0 += 4) {
            float ov[4], sv[4];
            Row w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int fi = min(f0 + u, a.F - 1);
              ov[u] = o[fi]; sv[u] = a.s_obs[fi];
              w[u] = row_load(a.W_obs + fi * AE, kq);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float oq = f0 + u < a.F ? ov[u] : 0.f;
              ms += oq * oq;
              const float c = oq * sv[u];
#pragma unroll
              for (int j = 0; j < 16; ++j) z.v[j] += c * w[u].v[j];
            }
          }
          const float rstd = rsqrtf(ms / (float)a.F + EPSN);
#pragma unroll
          for (int j = 0; j < 16; ++j) z.v[j] *= rstd;
          x = row_rms(row_gelu(z), a.s_encln, kq);
        } else {        // x = ln(rep of the previous block) (shared self.ln, sable_network.py:150)
          x = row_rms(row_load(a.rep + row * AE, kq), a.s_encln, kq);
        }
        if (valid && tt < nt) row_store(a.xn + row * AE, kq, x);
        kin[tt] = row_add(x, pe);
      }
      wgemm_multi<16, 4, true>(kin, nt, B.qkvg_t, m, kq, [&](int tt, int g, f32x4 acc) {
        if (valid && tt < nt) st4g(a.qkvg + (ge * A + t0 + tt) * 256 + 16 * g + 4 * kq, make_float4(acc[0], acc[1], acc[2], acc[3]));
      });
    }
    wsync();
    PROF(0);
    if (b == 0) ret_pass<0, NA, (EPW == 16 ? MAGPO_ACT_NBUF_ENC : 2), NH, MAGPO_ACT_PRIME != 0>(TQ, HK, U, a.S_enc, NS, a, env0, nvalid, 0, a.qkvg, 256, 0, a.u, AE, B.gn_g, B.gn_b, a.value_only ? 0 : 1, dmask, nullptr, 0, 0, pS);
    else ret_pass<0, NA, (EPW == 16 ? ACT_NBUF16 : 2), NH>(TQ, HK, U, a.S_enc + (long)b * nh_ * NS, NS, a, env0, nvalid, 0, a.qkvg, 256, 0, a.u, AE, B.gn_g, B.gn_b, a.value_only ? 0 : 1, dmask);
    // the first cross-retention state of the decoder pre-pass rides along with the encoder's post-retention dense phase
    if (MAGPO_ACT_PRIME && b == nb - 1 && !a.value_only) prime_state(pS, a.S_d2 + (long)env0 * 4096, lane);
    wsync();
    PROF(1);
    for (int t0 = 0; t0 < A; t0 += 4) {
      const int nt = min(4, A - t0);
      Row rep[4];
      {
        Row uu[4], y[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) uu[tt] = row_load(a.u + (ge * A + min(t0 + tt, A - 1)) * AE, kq);
        dense64_multi<4, true>(uu, nt, y, B.wo_t, nullptr, m, kq);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const long row = ge * A + min(t0 + tt, A - 1);
          const Row x = row_load(a.xn + row * AE, kq);
          rep[tt] = row_rms(row_rms(row_add(x, y[tt]), B.ln1, kq), B.ln2, kq);
          if (valid && tt < nt) row_store(a.rep + row * AE, kq, rep[tt]);
        }
      }
      if (b == nb - 1) {
        {
          Row hv[4];
          dense64_multi<4, true>(rep, nt, hv, a.vh0_t, a.vh0_b, m, kq);
          const Row w = row_load(a.vh_w, kq);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            const Row h = row_rms(row_gelu(hv[tt]), a.vh_s, kq);
            Row hw;
#pragma unroll
            for (int j = 0; j < 16; ++j) hw.v[j] = h.v[j] * w.v[j];
            const float val = row_sum(hw) + a.vh_b1[0];
            if (valid && kq == 0 && tt < nt) a.value[ge * A + t0 + tt] = val;
          }
        }
        if (!a.value_only) {
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) rep[tt] = row_add(rep[tt], pe);   // reppe
          for (int db = 0; db < nb; ++db) {
            float* q2o = a.blk[db].q2;
            wgemm_multi<4, 4, true>(rep, nt, a.blk[db].q2_t, m, kq, [&](int tt, int g, f32x4 acc) {
              if (valid && tt < nt) st4g(q2o + (ge * A + t0 + tt) * AE + 16 * g + 4 * kq, make_float4(acc[0], acc[1], acc[2], acc[3]));
            });
          }
        }
      }
    }
    wsync();
    PROF(2);
  }
  PROF(3);
  constexpr int NBF = EPW == 16 ? ACT_NBUF16 : 2;
  // flush: S <- kappa S + sum_a k_a^T v_a with the rows in the scratch (this launch's, or the pending ones of the previous launch)
  auto flush_states = [&](bool d1_block0_done) {
    wsync();
    for (int b = 0; b < nb; ++b) {
      const ActBlk& B = a.blk[b];
      if (!(b == 0 && d1_block0_done))
        ret_pass<3, NA, NBF, NH>(TQ, HK, U, a.S_d1 + (long)b * nh_ * NS, NS, a, env0, nvalid, 0, B.qkvg1, 256, 64, nullptr, 0, B.gn1_g, B.gn1_b, 1, 0ull);
      ret_pass<3, NA, NBF, NH>(TQ, HK, U, a.S_d2 + (long)b * nh_ * NS, NS, a, env0, nvalid, 0, B.kvg2, 256, 0, nullptr, 0, B.gn2_g, B.gn2_b, 1, 0ull);
    }
  };
  if (a.value_only) {   // uniform: bootstrap value only (rec_magpo.py:202-208); the last launch of a rollout also settles the decoder states
    if (a.flush && a.pending) flush_states(defer_wave && a.precand);   // (the deferring waves' S_d1 already holds the last step's rows)
    return;
  }

  // decoder pre-pass: every decoder state is loaded once, brought up to date (pending rows of the previous launch, episode-end
  // zeroing), written back, and gives from registers what the decoder needs from kappa S
  for (int b = 0; b < nb; ++b) {
    const ActBlk& B = a.blk[b];
    if (b == 0) ret_pass<4, NA, (EPW == 16 ? MAGPO_ACT_NBUF_PRE : 2), NH, MAGPO_ACT_PRIME != 0>(TQ, HK, U, a.S_d2, NS, a, env0, nvalid, 0, B.kvg2, 256, 0, B.kvg2 + 192, 256, B.gn2_g, B.gn2_b, 1, dmask,
                                               B.q2, AE, a.pending, pS);
    else ret_pass<4, NA, NBF, NH>(TQ, HK, U, a.S_d2 + (long)b * nh_ * NS, NS, a, env0, nvalid, 0, B.kvg2, 256, 0, B.kvg2 + 192, 256, B.gn2_g, B.gn2_b, 1, dmask,
                                  B.q2, AE, a.pending);
    PROF(19);
    if (b == 0 && cand) {
      // (the candidate pre-pass of block 0 ran ahead of the encoder, or at the end of the previous launch)
    } else {
      ret_pass<5, NA, NBF, NH>(TQ, HK, U, a.S_d1 + (long)b * nh_ * NS, NS, a, env0, nvalid, 0, B.qkvg1, 256, 64, nullptr, 0, B.gn1_g, B.gn1_b, 1, dmask,
                               nullptr, 0, a.pending);
    }
  }
  wsync();
  PROF(4);

  // ---------------- autoregressive decoder (decode.py:111-153): token i of every env
  int prev = 0;   // 0 = start token, action + 1 afterwards
  // Weight fragments are requested one layer AHEAD of their use (wload / wgemm_pre, fm_rows.hpp): the first fragments of a layer fly
  // while the previous layer's MFMAs and row math run; those of the next agent's first layer while this agent is being sampled.
  // (Requesting a layer's first weight fragments one layer ahead -- wload / wgemm_pre, fm_rows.hpp -- was built and measured in round 4:
  // the kernel sits at its 512 registers, every 64 more end in scratch, and all variants were slower: 549 - 577 us against 537.)
  for (int i = 0; i < A; ++i) {
    const long row = ge * A + i;
    Row xo;
    for (int b = 0; b < nb; ++b) {
      const ActBlk& B = a.blk[b];
      Row xin;
      if (b == 0) xin = row_rms(row_gelu(row_load(a.W_act + (long)prev * AE, kq)), a.s_decln, kq);   // action embedding (:258-267)
      else xin = xo;
      Row cpe;
      if (b == 0 && cand) {
        // self-retention from the candidate table: r = P[prev] + P[pe] + sum_{a <= i} (q . k_a) v_a, everything in registers
        const float* pt = a.ptab + ge * (long)(a.K + 2) * 64;
        Row pc = row_load(pt + (long)prev * 64, kq), pp = row_load(pt + (long)(a.K + 1) * 64, kq);
        if (defer_wave && a.precand) {   // table built before the env step: an episode that ended since has a zero state, hence zero rows
          const float keepf = ((dmask >> le) & 1ull) ? 0.f : 1.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) { pc.v[j] *= keepf; pp.v[j] *= keepf; }
        }
        Row hk1[NA - 1], hv1[NA - 1];   // (unused: the history rows of the earlier agents are streamed, two row pairs at a time)
        const Row kin = row_add(xin, pe);
        Row q1, k1, v1, g1;
        float* hrow = B.qkvg1 + row * 256;
        wgemm<16, true>(kin, B.qkvg1_t, m, kq, [&](int g, f32x4 acc) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (g < 4) q1.v[4 * g + c] = acc[c];
            else if (g < 8) k1.v[4 * (g - 4) + c] = acc[c];
            else if (g < 12) v1.v[4 * (g - 8) + c] = acc[c];
            else g1.v[4 * (g - 12) + c] = acc[c];
          }
          if (valid && g >= 4 && g < 12) st4g(hrow + 16 * g + 4 * kq, make_float4(acc[0], acc[1], acc[2], acc[3]));   // k, v: history / pending rows
        });
        const Row u1 = cross_ret<NH, NA, true>(a, q1, k1, v1, g1, row_add(pc, pp), hk1, hv1, i, B.gn1_g, B.gn1_b, kq, B.qkvg1 + ge * A * 256 + 64);
        PROF(5);
        const Row y1 = dense64<true>(u1, B.wo1_t, nullptr, m, kq);
        cpe = row_add(row_rms(row_add(xin, y1), B.dln1, kq), pe);
      } else {
        if (valid) row_store(XS + env * UP, kq, xin);
        {
          const Row kin = row_add(xin, pe);
          float* hrow = B.qkvg1 + row * 256;
          wgemm<16, true>(kin, B.qkvg1_t, m, kq, [&](int g, f32x4 acc) {
            const float4 v4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (valid) {
              *reinterpret_cast<float4*>(TQ + env * QP + 16 * g + 4 * kq) = v4;
              if (g >= 4 && g < 12) st4g(hrow + 16 * g + 4 * kq, v4);   // k, v of this token: history for the later agents
            }
          });
        }
        wsync();
        PROF(17);
        // the state in memory is the one that entered this step (pre-pass): no zeroing, never written here
        ret_pass<1, NA, NBF, NH>(TQ, HK, U, a.S_d1 + (long)b * nh_ * NS, NS, a, env0, nvalid, i, B.qkvg1, 256, 64, nullptr, 0, B.gn1_g, B.gn1_b, 0, 0ull);
        wsync();
        PROF(18);
        const Row y1 = dense64<true>(row_load(U + le * UP, kq), B.wo1_t, nullptr, m, kq);
        cpe = row_add(row_rms(row_add(row_load(XS + le * UP, kq), y1), B.dln1, kq), pe);
      }
      {
        // cross-retention: q from the encoder (q2 row of this token), k/v/g from the decoder stream; everything but the
        // pre-pass term stays in registers, so the state is not touched until the last agent has been decoded
        // rows written by the earlier agents / the pre-pass: issued before the GEMM so that they arrive behind it
        Row hk2[NA - 1], hv2[NA - 1];
#pragma unroll
        for (int t = 0; t < NA - 1; ++t) {
          if (NA <= 4 && !MAGPO_ACT_STREAM4 && t < i) { hk2[t] = row_load(B.kvg2 + (ge * A + t) * 256, kq); hv2[t] = row_load(B.kvg2 + (ge * A + t) * 256 + 64, kq); }
        }
        const Row p2 = row_load(B.kvg2 + row * 256 + 192, kq);
        const Row q2 = row_load(B.q2 + row * AE, kq);
        Row k2, v2, g2;
        float* hrow = B.kvg2 + row * 256;
        wgemm<12, true>(cpe, B.kvg2_t, m, kq, [&](int g, f32x4 acc) {
          const float4 v4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (g < 4) k2.v[4 * g + c] = acc[c];
            else if (g < 8) v2.v[4 * (g - 4) + c] = acc[c];
            else g2.v[4 * (g - 8) + c] = acc[c];
          }
          if (valid && g < 8) st4g(hrow + 16 * g + 4 * kq, v4);   // k, v of this token: history for the later agents / the state update
        });
        const Row u2 = cross_ret<NH, NA>(a, q2, k2, v2, g2, p2, hk2, hv2, i, B.gn2_g, B.gn2_b, kq, B.kvg2 + ge * A * 256);
        PROF(6);
        const Row y2 = dense64<true>(u2, B.wo2_t, nullptr, m, kq);
        const Row repi = row_load(a.rep + row * AE, kq);
        xo = row_rms(row_rms(row_add(repi, y2), B.dln2, kq), B.dln3, kq);
      }
    }
    // head (sable_network.py:296-319) and sampling
    const Row hn = row_rms(row_gelu(dense64<true>(xo, a.h0_t, a.h0_b, m, kq)), a.h_s, kq);
    Row lg;
#pragma unroll
    for (int j = 8; j < 16; ++j) lg.v[j] = 0.f;
    wgemm<2, true>(hn, a.h1_t, m, kq, [&](int g, f32x4 acc) {   // K <= 31 actions: two column groups of 16 hold every logit
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int n = 16 * g + 4 * kq + r; lg.v[4 * g + r] = acc[r] + (n < a.K ? a.h1_b[n] : 0.f); }
    });
    PROF(7);
    const uint32_t k0 = a.keys_dev ? a.keys_dev[2 * i] : a.keys[i][0], k1 = a.keys_dev ? a.keys_dev[2 * i + 1] : a.keys[i][1];
    const unsigned char* mk = a.mask ? a.mask + (ge * A + i) * a.K : nullptr;
    float xv[16];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = 16 * (j >> 2) + 4 * kq + (j & 3);
      xv[j] = (n < a.K) ? ((mk && !mk[n]) ? FMIN_ : lg.v[j]) : -INFINITY;
      mx = fmaxf(mx, xv[j]);
    }
    mx = fmaxf(mx, xget16(mx, lane));
    mx = fmaxf(mx, xget32(mx, lane));
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = 16 * (j >> 2) + 4 * kq + (j & 3);
      if (n < a.K) se += expf(xv[j] - mx);
    }
    se = xsum32(xsum16(se));
    const float lse = mx + logf(se);
    float best = -INFINITY, best_lp = 0.f;
    int arg = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = 16 * (j >> 2) + 4 * kq + (j & 3);
      if (n < a.K) {
        const float lp = xv[j] - lse;
        const uint32_t bits = random_bits32(k0, k1, (uint32_t)(ge * a.K + n));
        const float f = __uint_as_float((bits >> 9) | 0x3f800000u) - 1.0f;
        const float uu = fmaxf(1.17549435e-38f, f + 1.17549435e-38f);
#ifdef MAGPO_X_FASTGUMBEL   // timing experiment: hardware fp32 logs (NOT the oracle's correctly rounded value)
        const float gmb = -__logf(-__logf(uu));
#else
        const float gmb = (float)(-log(-log((double)uu)));
#endif
        const float vv = gmb + lp;
        if (vv > best || (vv == best && n < arg)) { best = vv; arg = n; best_lp = lp; }
      }
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {   // first-max over the 4 lanes of the env
      const float ob = sh == 16 ? xget16(best, lane) : xget32(best, lane), olp = sh == 16 ? xget16(best_lp, lane) : xget32(best_lp, lane);
      const int oa = __float_as_int(sh == 16 ? xget16(__int_as_float(arg), lane) : xget32(__int_as_float(arg), lane));
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; best_lp = olp; }
    }
    if (valid && kq == 0) { a.action[row] = arg; a.logp[row] = best_lp; }
    prev = arg + 1;
    PROF(16);
  }
  if (defer_wave && a.defer) {   // the candidate pre-pass of the NEXT step (never together with flush: the rows would be applied twice)
    wsync();                     // this step's k | v rows are read back by other lanes
    const int pn = min(p_ + 1, a.npos - 1);
    const Row pe_n = row_load(a.pe + (long)pn * AE, kq);
    cand_pass(pe_n, 0ull, true);
  } else if (a.flush) flush_states(false);   // stand-alone step (or last step of a rollout without a value launch): settle the decoder states now
}

}  // namespace magpo


// One launcher per wave shape (envs per wave): explicitly instantiated in act_fused_epw{4,8,16}.hip so that the three sets of kernel
// instances compile in parallel (one translation unit took 130 s); act_fused.hip holds the C entry point only.
template <int EPW> void launch_act(const magpo::ActArgs& a, hipStream_t st) {
  using namespace magpo;
  const size_t lds = (size_t)((EPW + a.A) * QP + 2 * EPW * UP + EPW * 64) * sizeof(float);
  const dim3 grid((a.N + EPW - 1) / EPW), blk(64);
  if (a.A <= 4 && a.nh == 1) hipLaunchKernelGGL((k_sable_act<EPW, 4, 1>), grid, blk, lds, st, a);   // the benchmark shape: everything static
  else if (a.A <= 4) hipLaunchKernelGGL((k_sable_act<EPW, 4, 0>), grid, blk, lds, st, a);
  else hipLaunchKernelGGL((k_sable_act<EPW, 8, 0>), grid, blk, lds, st, a);
}
