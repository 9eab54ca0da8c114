// RL-side kernels of the MAGPO learner (gfx950): PRNG utilities, categorical sampling, GAE,
// minibatch gather (shuffle as index arithmetic), advantage moments and the fused guider/actor loss.
//
// Reference: rec_magpo.py:222-370 (_guider_loss_fn, _actor_loss_fn), :439-462 (shuffle + layout),
// utils/multistep.py:24-68 (GAE), networks/utils/sable/decode.py:128-149 (autoregressive sampling),
// distrax Categorical (log_prob / entropy / KL) and jax.random.categorical (gumbel-argmax).
#include "common.hpp"

namespace magpo {

constexpr float FMIN = -3.4028234663852886e38f;  // jnp.finfo(float32).min used to mask illegal actions

// ---- PRNG utilities -------------------------------------------------------------------------
__global__ void k_split(const uint32_t* __restrict__ key, uint32_t* __restrict__ out, long num) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  uint32_t a, b;
  threefry2x32(key[0], key[1], (uint32_t)(i >> 32), (uint32_t)i, a, b);
  out[2 * i] = a;
  out[2 * i + 1] = b;
}
__global__ void k_random_bits(const uint32_t* __restrict__ key, uint32_t* __restrict__ out, long num) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  uint32_t a, b;
  threefry2x32(key[0], key[1], (uint32_t)(i >> 32), (uint32_t)i, a, b);
  out[i] = a ^ b;
}

// ---- categorical sampling of one agent's action for every env ------------------------------------
// logits row n at logits + n*ld (K valid entries); mask row n at mask + n*mask_stride (nullable).
// gumbel for (n, k) uses flat counter n*K + k of the sample key (distrax sample shape (1,N,1,K)).
// The double log keeps the gumbel value correctly rounded so that the oracle and the device agree.
struct SampleArgs {
  const float* logits; long ld; const unsigned char* mask; long mask_stride;
  uint32_t k0, k1; const uint32_t* key_dev;  // key_dev (device, 2 words) overrides k0/k1: static arguments for graph replay
  int* action; long act_stride; float* logp; long logp_stride; int* next_idx; long next_stride;
  float* lp_all; long lp_ld;  // optional normalised log-probs out
  int N, K;
};
__global__ void k_sample(SampleArgs a) {
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const uint32_t k0 = a.key_dev ? a.key_dev[0] : a.k0, k1 = a.key_dev ? a.key_dev[1] : a.k1;
  const float* x = a.logits + n * a.ld;
  const unsigned char* m = a.mask ? a.mask + n * a.mask_stride : nullptr;
  float mx = -INFINITY;
  for (int k = 0; k < a.K; ++k) mx = fmaxf(mx, (m && !m[k]) ? FMIN : x[k]);
  float se = 0.f;
  for (int k = 0; k < a.K; ++k) se += expf(((m && !m[k]) ? FMIN : x[k]) - mx);
  const float lse = mx + logf(se);
  float best = -INFINITY;
  int arg = 0;
  float best_lp = 0.f;
  for (int k = 0; k < a.K; ++k) {
    const float lp = ((m && !m[k]) ? FMIN : x[k]) - lse;
    if (a.lp_all) a.lp_all[n * a.lp_ld + k] = lp;
    const uint32_t bits = random_bits32(k0, k1, (uint32_t)(n * a.K + k));
    const float f = __uint_as_float((bits >> 9) | 0x3f800000u) - 1.0f;
    const float u = fmaxf(1.17549435e-38f, f + 1.17549435e-38f);
    const float gmb = (float)(-log(-log((double)u)));
    const float v = gmb + lp;
    if (v > best) { best = v; arg = k; best_lp = lp; }
  }
  a.action[n * a.act_stride] = arg;
  a.logp[n * a.logp_stride] = best_lp;
  if (a.next_idx) a.next_idx[n * a.next_stride] = arg + 1;
}

// ---- GAE (multistep.py:24-68): one thread per (env, agent), reverse over T --------------------------
__global__ void k_gae(const float* __restrict__ reward, const float* __restrict__ value, const unsigned char* __restrict__ done,
                      const float* __restrict__ last_val, const unsigned char* __restrict__ last_done,
                      float* __restrict__ adv, float* __restrict__ targets, int T, int N, int A, float gamma, float lam) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long NA = (long)N * A;
  if (i >= NA) return;
  const long n = i / A;
  float gae = 0.f, next_value = last_val[i], next_done = last_done[n] ? 1.f : 0.f;
#pragma unroll 4
  for (int t = T - 1; t >= 0; --t) {
    const float v = value[t * NA + i];
    const float nd = 1.0f - next_done;
    const float delta = reward[t * NA + i] + gamma * next_value * nd - v;
    gae = delta + gamma * lam * nd * gae;
    adv[t * NA + i] = gae;
    targets[t * NA + i] = gae + v;
    next_value = v;
    next_done = done[(long)t * N + n] ? 1.f : 0.f;
  }
}

// ---- GAE as a wavefront prefix scan: one WAVE per (env, agent) sequence, 64 timesteps per tile, tiles from the end of the rollout.
// Step t applies the affine map f_t(x) = delta_t + c_t x with c_t = gamma lambda (1 - done_{t+1}) to the advantage of step t + 1, so
// adv_t = (f_t o f_{t+1} o ... o f_{T-1})(0): a reverse inclusive scan under composition, (a, b) o (a', b') = (a a', b + a b'), done in
// log2(64) shuffle steps per tile with the tile's result carried into the one before it.  A sequence costs T / 64 x 6 dependent steps
// instead of T, which is what matters when N * A is small (the paper's 64-env batch: 512 sequences = 512 threads of the serial kernel
// walking 128 dependent steps); its loads are strided by N * A floats per lane, so the streaming one-thread-per-sequence kernel above stays
// the choice for large batches (coalesced across sequences).  Same recurrence, other summation order: fp32 rounding differences only.
__global__ __launch_bounds__(64) void k_gae_scan(const float* __restrict__ reward, const float* __restrict__ value, const unsigned char* __restrict__ done,
                                                   const float* __restrict__ last_val, const unsigned char* __restrict__ last_done,
                                                   float* __restrict__ adv, float* __restrict__ targets, int T, int N, int A, float gamma, float lam) {
  const long NA = (long)N * A, i = blockIdx.x;
  const long n = i / A;
  const int lane = threadIdx.x;
  float carry = 0.f;   // advantage of the first step behind the tile
  for (int t0 = ((T - 1) / 64) * 64; t0 >= 0; t0 -= 64) {
    const int t = t0 + lane;
    const bool live = t < T;
    float a = 1.f, b = 0.f, v = 0.f;   // identity map for the lanes past the end of the rollout
    if (live) {
      v = value[t * NA + i];
      const bool last = t == T - 1;
      const float next_value = last ? last_val[i] : value[(long)(t + 1) * NA + i];
      const float nd = (last ? last_done[n] : done[(long)(t + 1) * N + n]) ? 0.f : 1.f;
      b = reward[t * NA + i] + gamma * next_value * nd - v;
      a = gamma * lam * nd;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {   // lane l: composition over lanes l .. min(l + 2 d - 1, 63)
      const float a2 = __shfl_down(a, d), b2 = __shfl_down(b, d);
      if (lane + d < 64) { b = b + a * b2; a = a * a2; }
    }
    const float g = b + a * carry;
    if (live) { adv[t * NA + i] = g; targets[t * NA + i] = g + v; }
    carry = __shfl(g, 0);
  }
}

// ---- minibatch gather: (T,N,A,...) trajectory -> sequence-major minibatch rows --------------------
// row = (j*T + t)*A + a'   <-   (t, env = env_idx[j], agent = agent_perm[a'])     (rec_magpo.py:441-462)
struct GatherArgs {
  const float* obs; const int* action; const int* stepcount; const unsigned char* done; const unsigned char* mask;
  const float* value; const float* logp; const float* adv; const float* targets;
  const int* env_idx; const int* agent_perm;
  float* o_obs; int* o_action; int* o_prev; int* o_pos; unsigned char* o_done; unsigned char* o_mask;
  float* o_value; float* o_logp; float* o_adv; float* o_targets; int* o_h0idx;
  int T, N, A, F, K, mb;
};
__global__ void k_gather_minibatch(GatherArgs g) {
  long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long R = (long)g.mb * g.T * g.A;
  if (r >= R) return;
  const int ap = (int)(r % g.A);
  const long jt = r / g.A;
  const int t = (int)(jt % g.T);
  const int j = (int)(jt / g.T);
  const int env = g.env_idx[j];
  const int ag = g.agent_perm[ap];
  const long src = ((long)t * g.N + env) * g.A + ag;
  for (int f = 0; f < g.F; ++f) g.o_obs[r * g.F + f] = g.obs[src * g.F + f];
  g.o_action[r] = g.action[src];
  g.o_prev[r] = ap == 0 ? 0 : g.action[((long)t * g.N + env) * g.A + g.agent_perm[ap - 1]] + 1;
  g.o_pos[r] = g.stepcount[(long)t * g.N + env];
  g.o_value[r] = g.value[src];
  g.o_logp[r] = g.logp[src];
  g.o_adv[r] = g.adv[src];
  g.o_targets[r] = g.targets[src];
  if (g.mask) for (int k = 0; k < g.K; ++k) g.o_mask[r * g.K + k] = g.mask[src * g.K + k];
  if (ap == 0) g.o_done[(long)j * g.T + t] = g.done[(long)t * g.N + env];
  if (t == 0) g.o_h0idx[(long)j * g.A + ap] = env * g.A + ag;
}

// ---- moments of the advantages (rec_magpo.py:283,356): mean and 1/(std + 1e-8), population std ------
__global__ void k_moments_partial(const float* __restrict__ x, long n, double* __restrict__ part) {
  __shared__ double sh[2][256];
  double s = 0.0, s2 = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = x[i];
    s += v;
    s2 += v * v;
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0][0]; part[2 * blockIdx.x + 1] = sh[1][0]; }
}
__global__ void k_moments_final(const double* __restrict__ part, int nb, long n, float* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  const double s = wave_sum_strided(part, nb, 2, 0), s2 = wave_sum_strided(part, nb, 2, 1);
  if (threadIdx.x != 0) return;
  const double mean = s / (double)n;
  double var = s2 / (double)n - mean * mean;
  if (var < 0.0) var = 0.0;
  out[0] = (float)mean;
  out[1] = 1.0f / ((float)sqrt(var) + 1e-8f);
}

// ---- fused MAGPO losses + gradients w.r.t. logits / value -------------------------------------------
struct LossArgs {
  const float* g_logits; const float* a_logits; long ldg, lda;   // raw logits, K valid columns
  const unsigned char* mask;                                      // [R][K] nullable
  const int* action; const float* old_logp; const float* old_value; const float* value;
  const float* adv; const float* targets; const float* adv_stats;  // [mean, 1/(std+1e-8)]
  float* dg_logits; float* da_logits; long lddg, lddda;            // gradients (columns >= K zeroed up to ld)
  float* dvalue;
  double* part;   // [grid][8] partial sums: pg, kl_masked, entropy, value_loss, actor_pg, kl, 0, 0
  long R; int K;
  float clip_eps, log_clip_gpo, ent_coef, vf_coef, alpha, inv_R;
};

__device__ __forceinline__ float min_grad_weight(float l1, float l2, bool first) {
  // d min(l1,l2)/d l1 (first) or /d l2: 1 / 0, split evenly on exact ties (JAX / torch convention)
  if (l1 == l2) return 0.5f;
  return ((l1 < l2) == first) ? 1.f : 0.f;
}
// all-reduce max over aligned groups of 16 lanes (one DPP row) on the VALU (a __shfl_xor is an LDS round trip); max is exact in any order
__device__ __forceinline__ float max16(float v) {
  v = fmaxf(v, dpp_mov_<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov_<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov_<0x141>(v));   // row_half_mirror
  v = fmaxf(v, dpp_mov_<0x140>(v));   // row_mirror
  return v;
}

// 16 lanes per row, 4 logit columns per lane (K <= 64): rows are read / written as coalesced 256-B lines.
__global__ __launch_bounds__(256) void k_magpo_loss(LossArgs a) {
  __shared__ double sh[6][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, c4 = 4 * l16;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long nrow4 = (a.R + 3) / 4;
  for (long base = (long)blockIdx.x * 4 + wave; base < nrow4; base += (long)gridDim.x * 4) {
    const long r = base * 4 + (lane >> 4);
    const bool ok = r < a.R;
    const long rr = ok ? r : 0;
    float xg[4], xa[4];
    bool legal[4];
    // one unconditional float4 per lane and operand when the rows are padded to 64 columns (a predicated load per element is an
    // exec-masked block with its own wait); the K / mask selection happens on the loaded values
    const bool vec = a.ldg >= 64 && a.lda >= 64 && ((a.ldg | a.lda) & 3) == 0;
    float lgv[4], lav[4];
    if (vec) {   // uniform branch: the element-wise (predicated) loads below are compiled only into the narrow-row path
      const float4 vg = *reinterpret_cast<const float4*>(a.g_logits + rr * a.ldg + c4);
      const float4 va = *reinterpret_cast<const float4*>(a.a_logits + rr * a.lda + c4);
      lgv[0] = vg.x; lgv[1] = vg.y; lgv[2] = vg.z; lgv[3] = vg.w;
      lav[0] = va.x; lav[1] = va.y; lav[2] = va.z; lav[3] = va.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = c4 + j < a.K;
        lgv[j] = in ? a.g_logits[rr * a.ldg + c4 + j] : 0.f;
        lav[j] = in ? a.a_logits[rr * a.lda + c4 + j] : 0.f;
      }
    }
    unsigned char mk[4] = {1, 1, 1, 1};
    if (a.mask) {
#pragma unroll
      for (int j = 0; j < 4; ++j) mk[j] = a.mask[rr * a.K + (c4 + j < a.K ? c4 + j : 0)];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = c4 + j < a.K;
      legal[j] = in && mk[j];
      xg[j] = in ? (legal[j] ? lgv[j] : FMIN) : -INFINITY;
      xa[j] = in ? (legal[j] ? lav[j] : FMIN) : -INFINITY;
    }
    const float mg = max16(fmaxf(fmaxf(xg[0], xg[1]), fmaxf(xg[2], xg[3])));
    const float ma = max16(fmaxf(fmaxf(xa[0], xa[1]), fmaxf(xa[2], xa[3])));
    float sg = 0.f, sa = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sg += expf(xg[j] - mg); sa += expf(xa[j] - ma); }
    const float lseg = mg + logf(sum16(sg)), lsea = ma + logf(sum16(sa));
    const int act = a.action[rr];
    float lpg[4], lpa[4], pg[4], pa[4];
    float ent = 0.f, kl = 0.f, g_logp = 0.f, a_logp = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lpg[j] = xg[j] - lseg; lpa[j] = xa[j] - lsea;
      pg[j] = expf(lpg[j]); pa[j] = expf(lpa[j]);
      const bool nz = pg[j] != 0.f;   // selects, not branches: an exec-masked block per element serialises the row
      ent -= nz ? pg[j] * lpg[j] : 0.f;
      kl += nz ? pg[j] * (lpg[j] - lpa[j]) : 0.f;
      const bool hit = c4 + j == act;
      g_logp = hit ? lpg[j] : g_logp;
      a_logp = hit ? lpa[j] : a_logp;
    }
    ent = sum16(ent); kl = sum16(kl); g_logp = sum16(g_logp); a_logp = sum16(a_logp);
    const float old = a.old_logp[rr];
    const float A_ = (a.adv[rr] - a.adv_stats[0]) * a.adv_stats[1];
    const float eps = a.clip_eps, ld = a.log_clip_gpo;
    // guider surrogate (rec_magpo.py:261-294)
    const float ratio = expf(g_logp - old);
    const float d = g_logp - a_logp;
    const float dcl = fminf(fmaxf(d, -ld), ld);
    const float cr = expf(dcl + a_logp - old);
    const float crc = fminf(fmaxf(cr, 1.f - eps), 1.f + eps);
    const float l1 = ratio * A_, l2 = crc * A_;
    const float pgl = -fminf(l1, l2);
    const bool kmask = (d < -ld) || (d > ld);
    const float dl2 = (cr > 1.f - eps && cr < 1.f + eps && d > -ld && d < ld) ? cr * A_ : 0.f;
    const float c_g = -(min_grad_weight(l1, l2, true) * l1 + min_grad_weight(l1, l2, false) * dl2);
    // value loss (:298-303)
    const float v = a.value[rr], vo = a.old_value[rr], tg = a.targets[rr];
    const float dv = v - vo;
    const float vcl = vo + fminf(fmaxf(dv, -eps), eps);
    const float e1 = (v - tg) * (v - tg), e2 = (vcl - tg) * (vcl - tg);
    const float vl = 0.5f * fmaxf(e1, e2);
    const float g1 = 2.f * (v - tg);
    const float g2 = (dv > -eps && dv < eps) ? 2.f * (vcl - tg) : 0.f;
    const float dvl = 0.5f * (e1 > e2 ? g1 : (e2 > e1 ? g2 : 0.5f * (g1 + g2)));
    // actor surrogate (:354-367)
    const float ra = expf(a_logp - old);
    const float rac = fminf(fmaxf(ra, 1.f - eps), 1.f + eps);
    const float m1 = ra * A_, m2 = rac * A_;
    const float apl = -fminf(m1, m2);
    const float dm2 = (ra > 1.f - eps && ra < 1.f + eps) ? ra * A_ : 0.f;
    const float c_a = -(min_grad_weight(m1, m2, true) * m1 + min_grad_weight(m1, m2, false) * dm2);
    const float km = kmask ? 1.f : 0.f;
    if (ok) {
      float4 og, oa;
      float* pog = &og.x;
      float* poa = &oa.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float onehot = (c4 + j == act) ? 1.f : 0.f;
        float gg = c_g * (onehot - pg[j]);
        gg += (pg[j] != 0.f) ? km * pg[j] * ((lpg[j] - lpa[j]) - kl) + a.ent_coef * pg[j] * (lpg[j] + ent) : 0.f;
        float ga = a.alpha * c_a * (onehot - pa[j]) + (pa[j] - pg[j]);
        if (!legal[j]) { gg = 0.f; ga = 0.f; }
        pog[j] = a.inv_R * gg;
        poa[j] = a.inv_R * ga;
      }
      if (a.lddg >= 64 && a.lddda >= 64) {   // uniform: whole padded rows
        *reinterpret_cast<float4*>(a.dg_logits + r * a.lddg + c4) = og;
        *reinterpret_cast<float4*>(a.da_logits + r * a.lddda + c4) = oa;
      } else {
        if (c4 + 3 < a.lddg) *reinterpret_cast<float4*>(a.dg_logits + r * a.lddg + c4) = og;
        else for (int j = 0; j < 4; ++j) if (c4 + j < a.lddg) a.dg_logits[r * a.lddg + c4 + j] = pog[j];
        if (c4 + 3 < a.lddda) *reinterpret_cast<float4*>(a.da_logits + r * a.lddda + c4) = oa;
        else for (int j = 0; j < 4; ++j) if (c4 + j < a.lddda) a.da_logits[r * a.lddda + c4 + j] = poa[j];
      }
      if (l16 == 0) {
        a.dvalue[r] = a.inv_R * a.vf_coef * dvl;
        acc[0] += pgl; acc[1] += km * kl; acc[2] += ent; acc[3] += vl; acc[4] += apl; acc[5] += kl;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const float w = wave_sum(acc[q]);
    if (lane == 0) sh[q][wave] = (double)w;
  }
  __syncthreads();
  if (threadIdx.x < 6) a.part[8 * blockIdx.x + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}
// out: [total, value_loss, actor_loss, guider_loss, kl_loss, entropy, actor_kl, total_guider, total_actor]
__global__ void k_loss_final(const double* __restrict__ part, int nb, float inv_R, float ent_coef, float vf_coef, float alpha,
                             float* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double s[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) s[q] = wave_sum_strided(part, nb, 8, q);
  if (threadIdx.x != 0) return;
  const float pg = (float)(s[0] * inv_R), klm = (float)(s[1] * inv_R), ent = (float)(s[2] * inv_R);
  const float vl = (float)(s[3] * inv_R), apl = (float)(s[4] * inv_R), kl = (float)(s[5] * inv_R);
  const float tg = pg + klm - ent_coef * ent + vf_coef * vl;
  const float ta = apl * alpha + kl;
  out[0] = tg + ta; out[1] = vl; out[2] = apl; out[3] = pg; out[4] = klm; out[5] = ent; out[6] = kl; out[7] = tg; out[8] = ta;
}

// misc small kernels ----------------------------------------------------------------------------
__global__ void k_copy_rows_f32(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, long R, int W) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * W) return;
  long r = i / W;
  int c = (int)(i - r * W);
  dst[r * ldd + c] = src[r * lds_ + c];
}
__global__ void k_repeat_done(const unsigned char* __restrict__ done, unsigned char* __restrict__ out, long N, int A) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * A) return;
  out[i] = done[i / A];
}

}  // namespace magpo

using namespace magpo;

extern "C" int magpo_threefry_split(const uint32_t* key, uint32_t* out, long num, hipStream_t st) {
  hipLaunchKernelGGL(k_split, dim3((unsigned)((num + 255) / 256)), dim3(256), 0, st, key, out, num);
  return check_launch("magpo_threefry_split");
}
extern "C" int magpo_threefry_random_bits(const uint32_t* key, uint32_t* out, long num, hipStream_t st) {
  hipLaunchKernelGGL(k_random_bits, dim3((unsigned)((num + 255) / 256)), dim3(256), 0, st, key, out, num);
  return check_launch("magpo_threefry_random_bits");
}
// Host-side key derivation (scalar, exact): out[i] = threefry2x32(key, (0, i)), i < num.
extern "C" int magpo_key_split_host(const uint32_t* key, int num, uint32_t* out) {
  for (int i = 0; i < num; ++i) threefry2x32(key[0], key[1], 0u, (uint32_t)i, out[2 * i], out[2 * i + 1]);
  return MAGPO_OK;
}
// jax.random.fold_in(key, data) = threefry2x32(key, (0, data)) (flax derives every parameter's init key this way: params.py)
extern "C" int magpo_key_fold_in_host(const uint32_t* key, uint32_t data, uint32_t* out) {
  threefry2x32(key[0], key[1], 0u, data, out[0], out[1]);
  return MAGPO_OK;
}
extern "C" int magpo_random_bits_host(const uint32_t* key, int num, uint32_t* out) {
  for (int i = 0; i < num; ++i) out[i] = random_bits32(key[0], key[1], (uint32_t)i);
  return MAGPO_OK;
}

extern "C" int magpo_sample_categorical(const float* logits, long ld, const unsigned char* mask, long mask_stride,
                                        uint32_t k0, uint32_t k1, const uint32_t* key_dev, int* action, long act_stride, float* logp,
                                        long logp_stride, int* next_idx, long next_stride, float* lp_all, long lp_ld, int N,
                                        int K, hipStream_t st) {
  if ((long)N * K >= (1L << 32)) { set_error("sample: N*K must be < 2^32"); return MAGPO_EINVAL; }
  SampleArgs a{logits, ld, mask, mask_stride, k0, k1, key_dev, action, act_stride, logp, logp_stride, next_idx, next_stride, lp_all, lp_ld, N, K};
  hipLaunchKernelGGL(k_sample, dim3((N + 127) / 128), dim3(128), 0, st, a);
  return check_launch("magpo_sample_categorical");
}

extern "C" int magpo_gae(const float* reward, const float* value, const unsigned char* done, const float* last_val,
                         const unsigned char* last_done, float* adv, float* targets, int T, int N, int A, float gamma,
                         float lam, hipStream_t st) {
  long NA = (long)N * A;
  if (NA <= 0 || T <= 0) return MAGPO_OK;
  // few sequences: one wave per sequence, prefix scan over time (k_gae_scan); many: one thread per sequence, coalesced across sequences
  if (NA < 8192 && T >= 16)
    hipLaunchKernelGGL(k_gae_scan, dim3((unsigned)NA), dim3(64), 0, st, reward, value, done, last_val, last_done, adv, targets, T, N, A, gamma, lam);
  else
    hipLaunchKernelGGL(k_gae, dim3((unsigned)((NA + 127) / 128)), dim3(128), 0, st, reward, value, done, last_val, last_done, adv,
                       targets, T, N, A, gamma, lam);
  return check_launch("magpo_gae");
}

extern "C" int magpo_gather_minibatch(const float* obs, const int* action, const int* stepcount, const unsigned char* done,
                                      const unsigned char* mask, const float* value, const float* logp, const float* adv,
                                      const float* targets, const int* env_idx, const int* agent_perm, float* o_obs,
                                      int* o_action, int* o_prev, int* o_pos, unsigned char* o_done, unsigned char* o_mask,
                                      float* o_value, float* o_logp, float* o_adv, float* o_targets, int* o_h0idx, int T,
                                      int N, int A, int F, int K, int mb, hipStream_t st) {
  GatherArgs g{obs, action, stepcount, done, mask, value, logp, adv, targets, env_idx, agent_perm, o_obs, o_action, o_prev,
               o_pos, o_done, o_mask, o_value, o_logp, o_adv, o_targets, o_h0idx, T, N, A, F, K, mb};
  long R = (long)mb * T * A;
  hipLaunchKernelGGL(k_gather_minibatch, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, g);
  return check_launch("magpo_gather_minibatch");
}

// workspace: >= 2*1024 doubles. out: [mean, 1/(std+1e-8)]
extern "C" int magpo_adv_moments(const float* x, long n, double* workspace, float* out, hipStream_t st) {
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_moments_partial, dim3(nb), dim3(256), 0, st, x, n, workspace);
  hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(64), 0, st, workspace, nb, n, out);
  return check_launch("magpo_adv_moments");
}

// workspace: >= 8*1024 doubles; loss_out: 9 floats (see k_loss_final)
extern "C" int magpo_loss_fwd_bwd(const float* g_logits, long ldg, const float* a_logits, long lda, const unsigned char* mask,
                                  const int* action, const float* old_logp, const float* old_value, const float* value,
                                  const float* adv, const float* targets, const float* adv_stats, float* dg_logits,
                                  long lddg, float* da_logits, long lddda, float* dvalue, double* workspace, float* loss_out,
                                  long R, int K, float clip_eps, float clip_gpo, float ent_coef, float vf_coef, float alpha,
                                  hipStream_t st) {
  if (K > 64 || K > ldg || K > lda || K > lddg || K > lddda || (lddg & 3) || (lddda & 3) || clip_gpo <= 0.f) {
    set_error("loss: need K <= 64, K <= strides, gradient strides multiples of 4, clip_gpo > 0");
    return MAGPO_EINVAL;
  }
  int nb = (int)((R + 15) / 16);
  if (nb > 1024) nb = 1024;
  LossArgs a{g_logits, a_logits, ldg, lda, mask, action, old_logp, old_value, value, adv, targets, adv_stats,
             dg_logits, da_logits, lddg, lddda, dvalue, workspace, R, K, clip_eps, logf(clip_gpo), ent_coef, vf_coef, alpha,
             1.0f / (float)R};
  hipLaunchKernelGGL(k_magpo_loss, dim3(nb), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, st, workspace, nb, 1.0f / (float)R, ent_coef, vf_coef, alpha, loss_out);
  return check_launch("magpo_loss_fwd_bwd");
}

extern "C" int magpo_copy_rows(const float* src, long lds_, float* dst, long ldd, long R, int W, hipStream_t st) {
  long n = R * W;
  hipLaunchKernelGGL(k_copy_rows_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds_, dst, ldd, R, W);
  return check_launch("magpo_copy_rows");
}
