// Level-Based Foraging env + Mava wrappers (LbfWrapper with the always-on team reward, AgentID, AutoReset,
// RecordEpisodeMetrics; mava/wrappers/jumanji.py:171-220, mava/utils/make_env.py:90-135) for gfx950.
// UNPINNED DYNAMICS: the environment itself is third-party Jumanji (absent from the reference tree); this kernel and
// oracle/lbf.py restate its published algorithm (see the oracle's module docstring for every rule) and are bit-exact with
// each other.  One thread per env: the state is a few dozen integers, so the kernel is a pure HBM stream
// (state in, state + obs [A][3 (NF + A) + A] f32 + action mask [A][6] u8 out), coalesced across the envs of a wave by the
// struct-of-arrays layout below.
#include "common.hpp"

namespace magpo {

constexpr int LBF_MAXA = 8, LBF_MAXF = 8, LBF_NACT = 6, LBF_LOAD = 5;
__device__ __constant__ int c_lbf_dr[6] = {0, -1, 1, 0, 0, 0};
__device__ __constant__ int c_lbf_dc[6] = {0, 0, 0, -1, 1, 0};

struct LbfState {
  int* agent_pos;          // [N][A][2] (row, col)
  int* agent_level;        // [N][A]
  int* food_pos;           // [N][NF][2]
  int* food_level;         // [N][NF]
  unsigned char* food_eaten;  // [N][NF]
  int* step_count;         // [N]
  uint32_t* key;           // [N][2]   LevelBasedForaging State.key
  uint32_t* metrics_key;   // [N][2]   RecordEpisodeMetricsState.key (kept, never consumed)
  float* run_ret; int* run_len; float* ep_ret; int* ep_len;  // [N]
};
struct LbfCfg { int N, A, NF, G, fov, max_level, force_coop, TLIM; };

struct LbfEnv {   // one env in registers
  int ar[LBF_MAXA], ac[LBF_MAXA], al[LBF_MAXA];
  int fr[LBF_MAXF], fc[LBF_MAXF], fl[LBF_MAXF];
  bool fe[LBF_MAXF];
};

__device__ __forceinline__ uint32_t rb32(uint32_t k0, uint32_t k1) { return random_bits32(k0, k1, 0u); }
__device__ __forceinline__ int lbf_randint(uint32_t k0, uint32_t k1, uint32_t i, int lo, int hi) {
  // jax.random.randint(key, (n,), lo, hi) element i (oracle/prng.py:randint)
  uint32_t a0, a1, b0, b1;
  threefry2x32(k0, k1, 0u, 0u, a0, a1);
  threefry2x32(k0, k1, 0u, 1u, b0, b1);
  const uint32_t span = hi > lo ? (uint32_t)(hi - lo) : 1u;
  const uint32_t h = random_bits32(a0, a1, i), l = random_bits32(b0, b1, i);
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  return lo + (int)(((h % span) * mult + (l % span)) % span);
}
__device__ __forceinline__ void clr(unsigned long long (&m)[4], int cell) { m[cell >> 6] &= ~(1ull << (cell & 63)); }

// RandomGenerator.__call__ (see oracle/lbf.py:_generate); returns the State.key
__device__ __forceinline__ void lbf_generate(const LbfCfg& c, uint32_t k0, uint32_t k1, LbfEnv& e, uint32_t& sk0, uint32_t& sk1) {
  uint32_t kf0, kf1, ka0, ka1, kfl0, kfl1, kal0, kal1;
  threefry2x32(k0, k1, 0u, 0u, kf0, kf1);     // key_food
  threefry2x32(k0, k1, 0u, 1u, ka0, ka1);     // key_agents
  threefry2x32(k0, k1, 0u, 2u, kfl0, kfl1);   // key_food_level
  threefry2x32(k0, k1, 0u, 3u, kal0, kal1);   // key_agent_level
  threefry2x32(k0, k1, 0u, 4u, sk0, sk1);     // key
  const int G = c.G;
  unsigned long long valid[4] = {0, 0, 0, 0}, freec[4] = {0, 0, 0, 0};
  for (int r = 0; r < G; ++r)
    for (int q = 0; q < G; ++q) {
      const int cell = r * G + q;
      freec[cell >> 6] |= 1ull << (cell & 63);
      if (r > 0 && r < G - 1 && q > 0 && q < G - 1) valid[cell >> 6] |= 1ull << (cell & 63);
    }
  for (int f = 0; f < c.NF; ++f) {
    uint32_t s0, s1;
    threefry2x32(kf0, kf1, 0u, (uint32_t)f, s0, s1);
    const int cell = choice_mask_cumsum(valid, s0, s1);   // take_positions: choice(key_f, G*G, (), p=mask)
    const int r = cell / G, q = cell - r * G;
    e.fr[f] = r; e.fc[f] = q; e.fe[f] = false;
    clr(valid, cell);
    if (r + 1 < G) clr(valid, cell + G);
    if (r > 0) clr(valid, cell - G);
    if (q + 1 < G) clr(valid, cell + 1);
    if (q > 0) clr(valid, cell - 1);
    clr(freec, cell);
  }
  // sample_agents: ONE choice(key_agents, G*G, (A,), replace=False, p=mask) = Gumbel top-k (oracle/prng.py:choice): cell i carries
  // g_i = gumbel(key, (G*G,))[i] + log(mask_i); the A largest in descending order, equal values by lower index.  Cells with food
  // (log 0 = -inf) come last, in index order, exactly as a stable descending sort leaves them.
  {
    float bg[LBF_MAXA];
    int bc[LBF_MAXA];
    for (int a = 0; a < c.A; ++a) { bg[a] = -INFINITY; bc[a] = -1; }
    for (int cell = 0; cell < G * G; ++cell) {
      const bool ok = (freec[cell >> 6] >> (cell & 63)) & 1ull;
      const float g = ok ? gumbel_exact_from_bits(random_bits32(ka0, ka1, (uint32_t)cell)) : -INFINITY;
      // insert behind every entry that is >= g (earlier cells win ties); empty slots (bc < 0) lose against everything
      int pos = c.A;
      for (int a = c.A - 1; a >= 0; --a) if (bc[a] < 0 || g > bg[a]) pos = a;
      for (int a = c.A - 1; a > pos; --a) { bg[a] = bg[a - 1]; bc[a] = bc[a - 1]; }
      if (pos < c.A) { bg[pos] = g; bc[pos] = cell; }
    }
    for (int a = 0; a < c.A; ++a) {
      e.ar[a] = bc[a] / G; e.ac[a] = bc[a] - e.ar[a] * G;
      e.al[a] = lbf_randint(kal0, kal1, (uint32_t)a, 1, c.max_level + 1);
    }
  }
  // sum of the (up to) three smallest agent levels
  int m1 = 1 << 30, m2 = 1 << 30, m3 = 1 << 30;
  for (int a = 0; a < c.A; ++a) {
    const int v = e.al[a];
    if (v < m1) { m3 = m2; m2 = m1; m1 = v; } else if (v < m2) { m3 = m2; m2 = v; } else if (v < m3) { m3 = v; }
  }
  const int maxfl = m1 + (c.A > 1 ? m2 : 0) + (c.A > 2 ? m3 : 0);
  for (int f = 0; f < c.NF; ++f) e.fl[f] = c.force_coop ? maxfl : lbf_randint(kfl0, kfl1, (uint32_t)f, 1, maxfl + 1);
}

__device__ __forceinline__ void lbf_store(const LbfState& s, const LbfCfg& c, long n, const LbfEnv& e) {
  for (int a = 0; a < c.A; ++a) {
    s.agent_pos[(n * c.A + a) * 2] = e.ar[a]; s.agent_pos[(n * c.A + a) * 2 + 1] = e.ac[a];
    s.agent_level[n * c.A + a] = e.al[a];
  }
  for (int f = 0; f < c.NF; ++f) {
    s.food_pos[(n * c.NF + f) * 2] = e.fr[f]; s.food_pos[(n * c.NF + f) * 2 + 1] = e.fc[f];
    s.food_level[n * c.NF + f] = e.fl[f];
    s.food_eaten[n * c.NF + f] = e.fe[f] ? 1 : 0;
  }
}
__device__ __forceinline__ void lbf_load(const LbfState& s, const LbfCfg& c, long n, LbfEnv& e) {
  for (int a = 0; a < c.A; ++a) {
    e.ar[a] = s.agent_pos[(n * c.A + a) * 2]; e.ac[a] = s.agent_pos[(n * c.A + a) * 2 + 1];
    e.al[a] = s.agent_level[n * c.A + a];
  }
  for (int f = 0; f < c.NF; ++f) {
    e.fr[f] = s.food_pos[(n * c.NF + f) * 2]; e.fc[f] = s.food_pos[(n * c.NF + f) * 2 + 1];
    e.fl[f] = s.food_level[n * c.NF + f];
    e.fe[f] = s.food_eaten[n * c.NF + f] != 0;
  }
}

// VectorObserver + compute_action_mask + AgentIDWrapper: obs [A][A + 3 (NF + A)] f32, mask [A][6] u8
__device__ __forceinline__ void lbf_observe(const LbfCfg& c, const LbfEnv& e, float* __restrict__ obs, unsigned char* __restrict__ mask) {
  const int A = c.A, NF = c.NF, F = A + 3 * (NF + A);
  for (int a = 0; a < A; ++a) {
    float* o = obs + a * F;
    for (int i = 0; i < A; ++i) o[i] = i == a ? 1.f : 0.f;
    o += A;
    const int mr = e.ar[a], mc = e.ac[a];
    const int sr = min(c.fov, mr) - mr, sc = min(c.fov, mc) - mc;
    for (int f = 0; f < NF; ++f) {
      const bool vis = abs(e.fr[f] - mr) <= c.fov && abs(e.fc[f] - mc) <= c.fov && !e.fe[f];
      o[3 * f] = vis ? (float)(e.fr[f] + sr) : -1.f;
      o[3 * f + 1] = vis ? (float)(e.fc[f] + sc) : -1.f;
      o[3 * f + 2] = vis ? (float)e.fl[f] : 0.f;
    }
    float* p = o + 3 * NF;
    p[0] = (float)(mr + sr); p[1] = (float)(mc + sc); p[2] = (float)e.al[a];
    int j = 1;
    for (int b = 0; b < A; ++b) {
      if (b == a) continue;
      const bool vis = abs(e.ar[b] - mr) <= c.fov && abs(e.ac[b] - mc) <= c.fov;
      p[3 * j] = vis ? (float)(e.ar[b] + sr) : -1.f;
      p[3 * j + 1] = vis ? (float)(e.ac[b] + sc) : -1.f;
      p[3 * j + 2] = vis ? (float)e.al[b] : 0.f;
      ++j;
    }
    bool adj = false;
    for (int f = 0; f < NF; ++f) adj |= (abs(e.fr[f] - mr) + abs(e.fc[f] - mc) == 1) && !e.fe[f];
    for (int k = 0; k < LBF_NACT; ++k) {
      const int nr = mr + c_lbf_dr[k], nc = mc + c_lbf_dc[k];
      bool bad = nr < 0 || nr >= c.G || nc < 0 || nc >= c.G;
      for (int b = 0; b < A; ++b) bad |= b != a && e.ar[b] == nr && e.ac[b] == nc;
      for (int f = 0; f < NF; ++f) bad |= e.fr[f] == nr && e.fc[f] == nc && !e.fe[f];
      mask[a * LBF_NACT + k] = (!bad && (k != LBF_LOAD || adj)) ? 1 : 0;
    }
  }
}

__global__ __launch_bounds__(64) void k_lbf_reset(LbfState s, LbfCfg c, const uint32_t* __restrict__ env_keys, float* __restrict__ obs,
                                                  int* __restrict__ obs_step, unsigned char* __restrict__ mask) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= c.N) return;
  const uint32_t e0 = env_keys[2 * n], e1 = env_keys[2 * n + 1];
  uint32_t m0, m1, r0, r1, k0, k1;
  threefry2x32(e0, e1, 0u, 0u, m0, m1);  // key, reset_key = split(key)   (episode_metrics.py:62)
  threefry2x32(e0, e1, 0u, 1u, r0, r1);
  LbfEnv e;
  lbf_generate(c, r0, r1, e, k0, k1);
  lbf_store(s, c, n, e);
  s.step_count[n] = 0;
  s.key[2 * n] = k0; s.key[2 * n + 1] = k1;
  s.metrics_key[2 * n] = m0; s.metrics_key[2 * n + 1] = m1;
  s.run_ret[n] = 0.f; s.run_len[n] = 0; s.ep_ret[n] = 0.f; s.ep_len[n] = 0;
  const int F = c.A + 3 * (c.NF + c.A);
  lbf_observe(c, e, obs + n * (long)c.A * F, mask + n * (long)c.A * LBF_NACT);
  obs_step[n] = 0;
}

struct LbfOut {
  float* reward; float* discount; unsigned char* done; float* obs; int* obs_step; unsigned char* mask;
  float* m_ep_ret; int* m_ep_len; unsigned char* m_term;
};

__global__ __launch_bounds__(64) void k_lbf_step(LbfState s, LbfCfg c, const int* __restrict__ actions, int act_stride, LbfOut o,
                                                 int auto_reset) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= c.N) return;
  const int A = c.A, NF = c.NF;
  LbfEnv e;
  lbf_load(s, c, n, e);
  int act[LBF_MAXA], nr[LBF_MAXA], nc[LBF_MAXA];
  for (int a = 0; a < A; ++a) {
    int k = actions[n * act_stride + a];
    k = k < 0 ? 0 : (k >= LBF_NACT ? LBF_NACT - 1 : k);   // (a gather with an out-of-range index clamps)
    act[a] = k;
    const int r = e.ar[a] + c_lbf_dr[k], q = e.ac[a] + c_lbf_dc[k];
    bool blocked = r < 0 || r >= c.G || q < 0 || q >= c.G;
    for (int b = 0; b < A; ++b) blocked |= b != a && e.ar[b] == r && e.ac[b] == q;
    for (int f = 0; f < NF; ++f) blocked |= e.fr[f] == r && e.fc[f] == q && !e.fe[f];
    nr[a] = blocked ? e.ar[a] : r; nc[a] = blocked ? e.ac[a] : q;
  }
  bool dup[LBF_MAXA];
  for (int a = 0; a < A; ++a) {
    dup[a] = false;
    for (int b = 0; b < A; ++b) dup[a] |= b != a && nr[a] == nr[b] && nc[a] == nc[b];
  }
  for (int a = 0; a < A; ++a) if (!dup[a]) { e.ar[a] = nr[a]; e.ac[a] = nc[a]; }
  float rew[LBF_MAXA];
  for (int a = 0; a < A; ++a) rew[a] = 0.f;
  int tfl = 0;
  for (int f = 0; f < NF; ++f) tfl += e.fl[f];
  bool eaten_new[LBF_MAXF];
  for (int f = 0; f < NF; ++f) {
    int lv[LBF_MAXA], sum = 0;
    for (int a = 0; a < A; ++a) {
      const bool adj = (abs(e.ar[a] - e.fr[f]) + abs(e.ac[a] - e.fc[f]) == 1) && act[a] == LBF_LOAD && !e.fe[f];
      lv[a] = adj ? e.al[a] : 0;
      sum += lv[a];
    }
    const bool now = sum >= e.fl[f] && !e.fe[f] && sum > 0;
    const float norm = (float)sum * (float)tfl;
    for (int a = 0; a < A; ++a) {
      const float r = (float)(lv[a] * (now ? e.fl[f] : 0));
      rew[a] += norm > 0.f ? __fdiv_rn(r, norm) : 0.f;
    }
    eaten_new[f] = e.fe[f] || now;
  }
  bool all = true;
  for (int f = 0; f < NF; ++f) { e.fe[f] = eaten_new[f]; all &= e.fe[f]; }
  float team = 0.f;
  for (int a = 0; a < A; ++a) team += rew[a];   // aggregate_rewards (jumanji.py:43-46)
  const int steps = s.step_count[n] + 1;
  const bool done = all || steps >= c.TLIM;
  int obs_step = steps;
  if (done && auto_reset) {
    uint32_t k0 = s.key[2 * n], k1 = s.key[2 * n + 1], nk0, nk1, sk0, sk1;
    threefry2x32(k0, k1, 0u, 0u, nk0, nk1);  // key, _ = split(state.key)   (auto_reset_wrapper.py:74)
    lbf_generate(c, nk0, nk1, e, sk0, sk1);
    s.key[2 * n] = sk0; s.key[2 * n + 1] = sk1;
    obs_step = 0;
  }
  lbf_store(s, c, n, e);
  s.step_count[n] = obs_step;
  const int F = A + 3 * (NF + A);
  lbf_observe(c, e, o.obs + n * (long)A * F, o.mask + n * (long)A * LBF_NACT);
  o.obs_step[n] = obs_step;
  for (int a = 0; a < A; ++a) o.reward[n * A + a] = team;
  if (o.discount) for (int a = 0; a < A; ++a) o.discount[n * A + a] = all ? 0.f : 1.f;   // termination = all food eaten; the time limit truncates (discount 1)
  o.done[n] = done ? 1 : 0;
  // episode_metrics.py:79-112: mean over agents of the (identical) team rewards, as a sum / A in fp32
  float msum = 0.f;
  for (int a = 0; a < A; ++a) msum += team;
  const float new_ret = s.run_ret[n] + __fdiv_rn(msum, (float)A);
  const int new_len = s.run_len[n] + 1;
  const float ep_ret = done ? new_ret : s.ep_ret[n];
  const int ep_len = done ? new_len : s.ep_len[n];
  s.run_ret[n] = done ? 0.f : new_ret;
  s.run_len[n] = done ? 0 : new_len;
  s.ep_ret[n] = ep_ret;
  s.ep_len[n] = ep_len;
  o.m_ep_ret[n] = ep_ret;
  o.m_ep_len[n] = ep_len;
  o.m_term[n] = done ? 1 : 0;
}

}  // namespace magpo

using namespace magpo;

static int lbf_check(const LbfCfg& c) {
  if (c.A < 1 || c.A > LBF_MAXA || c.NF < 1 || c.NF > LBF_MAXF || c.G < 3 || c.G > 16 || c.max_level < 1 || c.TLIM < 1) {
    set_error("lbf: 1 <= num_agents <= 8, 1 <= num_food <= 8, 3 <= grid_size <= 16");
    return MAGPO_EINVAL;
  }
  // every food draw must find a valid cell whatever came before it: a food blocks at most 5 interior cells; and the agents need free cells
  if ((c.G - 2) * (c.G - 2) < 5 * (c.NF - 1) + 1 || c.G * c.G - c.NF < c.A) {
    set_error("lbf: the grid cannot hold that many food items / agents (need (G - 2)^2 >= 5 (num_food - 1) + 1 and G^2 - num_food >= num_agents)");
    return MAGPO_EINVAL;
  }
  return MAGPO_OK;
}

extern "C" int magpo_lbf_reset(int* agent_pos, int* agent_level, int* food_pos, int* food_level, unsigned char* food_eaten, int* step_count,
                               uint32_t* key, uint32_t* metrics_key, float* run_ret, int* run_len, float* ep_ret, int* ep_len, int N, int A,
                               int NF, int G, int fov, int max_level, int force_coop, int time_limit, const uint32_t* env_keys, float* obs,
                               int* obs_step, unsigned char* mask, hipStream_t st) {
  LbfState s{agent_pos, agent_level, food_pos, food_level, food_eaten, step_count, key, metrics_key, run_ret, run_len, ep_ret, ep_len};
  LbfCfg c{N, A, NF, G, fov, max_level, force_coop, time_limit};
  if (int e = lbf_check(c)) return e;
  if (N <= 0) return MAGPO_OK;
  hipLaunchKernelGGL(k_lbf_reset, dim3((N + 63) / 64), dim3(64), 0, st, s, c, env_keys, obs, obs_step, mask);
  return check_launch("magpo_lbf_reset");
}

extern "C" int magpo_lbf_step(int* agent_pos, int* agent_level, int* food_pos, int* food_level, unsigned char* food_eaten, int* step_count,
                              uint32_t* key, uint32_t* metrics_key, float* run_ret, int* run_len, float* ep_ret, int* ep_len, int N, int A,
                              int NF, int G, int fov, int max_level, int force_coop, int time_limit, const int* actions, int act_stride,
                              float* reward, float* discount, unsigned char* done, float* obs, int* obs_step, unsigned char* mask, float* m_ep_ret,
                              int* m_ep_len, unsigned char* m_term, int auto_reset, hipStream_t st) {
  LbfState s{agent_pos, agent_level, food_pos, food_level, food_eaten, step_count, key, metrics_key, run_ret, run_len, ep_ret, ep_len};
  LbfCfg c{N, A, NF, G, fov, max_level, force_coop, time_limit};
  if (int e = lbf_check(c)) return e;
  if (N <= 0) return MAGPO_OK;
  LbfOut o{reward, discount, done, obs, obs_step, mask, m_ep_ret, m_ep_len, m_term};
  hipLaunchKernelGGL(k_lbf_step, dim3((N + 63) / 64), dim3(64), 0, st, s, c, actions, act_stride, o, auto_reset);
  return check_launch("magpo_lbf_step");
}
