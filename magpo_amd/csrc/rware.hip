// Robot Warehouse env + Mava wrappers (RwareWrapper: observation as float, the shared scalar reward repeated per agent;
// AgentID, AutoReset, RecordEpisodeMetrics; mava/wrappers/jumanji.py:137-168, mava/utils/make_env.py:90-135) for gfx950.
// UNPINNED DYNAMICS: the environment is third-party Jumanji (absent from the reference tree); this kernel and
// oracle/rware.py restate its published algorithm (every rule is listed in the oracle's module docstring) and are
// bit-exact with each other.  One thread per env working in place on the env's grids in HBM (two H x W int layers):
// a step touches a few dozen cells, the observation window and the outputs, so the kernel is a small latency-bound
// stream next to the acting kernel.
#include "common.hpp"

namespace magpo {

constexpr int RW_NACT = 5, RW_FORWARD = 1, RW_LEFT = 2, RW_RIGHT = 3, RW_TOGGLE = 4, RW_MAXA = 8;

struct RwState {
  int* grid_a; int* grid_s;          // [N][H*W] agents / shelves layer: 0 = empty, id + 1 otherwise
  int* agent_pos; int* agent_dir;    // [N][A][2] (row, col), [N][A] (0 up, 1 right, 2 down, 3 left)
  unsigned char* agent_carry;        // [N][A]
  unsigned char* shelf_req;          // [N][NS]
  int* queue;                        // [N][Q] requested shelf ids (1-based)
  int* step_count;                   // [N]
  unsigned char* amask;              // [N][A][5] action mask of the current state (sanitises the next actions)
  uint32_t* key; uint32_t* metrics_key;   // [N][2]
  float* run_ret; int* run_len; float* ep_ret; int* ep_len;
};
struct RwCfg { int N, A, CH, SR, SC, R, Q, TLIM, H, W, NS; };

__host__ __device__ __forceinline__ bool rw_highway(const RwCfg& c, int y, int x) {
  return (x % 3 == 0) || (y % (c.CH + 1) == 0) || (y == c.H - 1) || (y > c.H - (c.CH + 3) && (x == c.W / 2 - 1 || x == c.W / 2));
}
__device__ __forceinline__ void rw_ahead(const RwCfg& c, int r, int q, int d, int& nr, int& nc) {
  const int dr = d == 0 ? -1 : (d == 2 ? 1 : 0), dc = d == 1 ? 1 : (d == 3 ? -1 : 0);
  nr = min(max(r + dr, 0), c.H - 1);
  nc = min(max(q + dc, 0), c.W - 1);
}
__device__ __forceinline__ int rw_randint4(uint32_t k0, uint32_t k1, uint32_t i) {   // jax.random.randint(key, (n,), 0, 4) element i
  uint32_t a0, a1, b0, b1;
  threefry2x32(k0, k1, 0u, 0u, a0, a1);
  threefry2x32(k0, k1, 0u, 1u, b0, b1);
  const uint32_t h = random_bits32(a0, a1, i), l = random_bits32(b0, b1, i);
  const uint32_t mult = ((65536u % 4u) * (65536u % 4u)) % 4u;
  return (int)(((h % 4u) * mult + (l % 4u)) % 4u);
}

// jax.random.choice(key, n, (num,), replace=False) without p = permutation(key, n)[:num] (oracle/prng.py:choice / permutation): for
// n < 1626 the shuffle is ONE round -- key, sub = split(key); stable sort of 0..n-1 by random_bits(sub, n) -- so the prefix is the num
// smallest (bits, index) pairs in order.  num <= RW_MAXPICK.
constexpr int RW_MAXPICK = 16;
__device__ __forceinline__ void rw_perm_prefix(uint32_t k0, uint32_t k1, int n, int num, int* __restrict__ out) {
  uint32_t s0, s1;
  threefry2x32(k0, k1, 0u, 1u, s0, s1);   // sub = split(key)[1]
  uint32_t bb[RW_MAXPICK];
  int bi[RW_MAXPICK];
  int filled = 0;
  for (int i = 0; i < n; ++i) {
    const uint32_t b = random_bits32(s0, s1, (uint32_t)i);
    int pos = filled;                       // behind every entry with bits <= b (earlier indices win ties)
    for (int a = filled - 1; a >= 0; --a) if (b < bb[a]) pos = a;
    if (pos >= num) continue;
    const int last = filled < num ? filled : num - 1;
    for (int a = last; a > pos; --a) { bb[a] = bb[a - 1]; bi[a] = bi[a - 1]; }
    bb[pos] = b; bi[pos] = i;
    if (filled < num) ++filled;
  }
  for (int a = 0; a < num; ++a) out[a] = bi[a];
}

__device__ __forceinline__ void rw_mask(const RwCfg& c, const RwState& s, long n) {
  const int HW = c.H * c.W;
  const int* ga = s.grid_a + n * HW; const int* gs = s.grid_s + n * HW;
  for (int a = 0; a < c.A; ++a) {
    int nr, nc;
    rw_ahead(c, s.agent_pos[(n * c.A + a) * 2], s.agent_pos[(n * c.A + a) * 2 + 1], s.agent_dir[n * c.A + a], nr, nc);
    const bool bad = ga[nr * c.W + nc] > 0 || (s.agent_carry[n * c.A + a] && gs[nr * c.W + nc] > 0);
    unsigned char* m = s.amask + (n * c.A + a) * RW_NACT;
    m[0] = 1; m[1] = bad ? 0 : 1; m[2] = 1; m[3] = 1; m[4] = 1;
  }
}

// RandomGenerator.__call__ (oracle/rware.py:_generate); writes the whole env state, returns nothing
__device__ __forceinline__ void rw_generate(const RwCfg& c, const RwState& s, long n, uint32_t k0, uint32_t k1) {
  const int HW = c.H * c.W;
  int* ga = s.grid_a + n * HW; int* gs = s.grid_s + n * HW;
  uint32_t kp0, kp1, kd0, kd1, kq0, kq1, ks0, ks1;
  threefry2x32(k0, k1, 0u, 0u, kp0, kp1);   // key_pos
  threefry2x32(k0, k1, 0u, 1u, kd0, kd1);   // key_dir
  threefry2x32(k0, k1, 0u, 2u, kq0, kq1);   // key_queue
  threefry2x32(k0, k1, 0u, 3u, ks0, ks1);   // key
  int sid = 0;
  for (int y = 0; y < c.H; ++y)
    for (int x = 0; x < c.W; ++x) {
      ga[y * c.W + x] = 0;
      gs[y * c.W + x] = rw_highway(c, y, x) ? 0 : ++sid;
    }
  int picks[RW_MAXPICK];
  rw_perm_prefix(kp0, kp1, HW, c.A, picks);   // agent cells: choice(key_pos, H*W, (A,), replace=False)
  for (int a = 0; a < c.A; ++a) {
    const int cell = picks[a];
    ga[cell] = a + 1;
    s.agent_pos[(n * c.A + a) * 2] = cell / c.W;
    s.agent_pos[(n * c.A + a) * 2 + 1] = cell % c.W;
    s.agent_dir[n * c.A + a] = rw_randint4(kd0, kd1, (uint32_t)a);
    s.agent_carry[n * c.A + a] = 0;
  }
  unsigned char* req = s.shelf_req + n * c.NS;
  for (int i = 0; i < c.NS; ++i) req[i] = 0;
  rw_perm_prefix(kq0, kq1, c.NS, c.Q, picks);   // request queue: choice(key_queue, shelf ids, (Q,), replace=False)
  for (int q = 0; q < c.Q; ++q) {
    req[picks[q]] = 1;
    s.queue[n * c.Q + q] = picks[q] + 1;
  }
  s.step_count[n] = 0;
  s.key[2 * n] = ks0; s.key[2 * n + 1] = ks1;
  rw_mask(c, s, n);
}

// observation [A][ldo] f32 = [one-hot id | row, col, carrying, one-hot dir, highway | per window cell: agent, its dir one-hot, shelf,
// requested]; action mask [A][5] u8 (copied from the state)
__device__ __forceinline__ void rw_observe(const RwCfg& c, const RwState& s, long n, float* __restrict__ obs, long ldo, unsigned char* __restrict__ mask) {
  const int HW = c.H * c.W, A = c.A;
  const int* ga = s.grid_a + n * HW; const int* gs = s.grid_s + n * HW;
  const int nf = A + 8 + 7 * (2 * c.R + 1) * (2 * c.R + 1);
  for (int a = 0; a < A; ++a) {
    float* o = obs + a * ldo;
    for (int i = 0; i < nf; ++i) o[i] = 0.f;
    o[a] = 1.f;
    o += A;
    const int r = s.agent_pos[(n * A + a) * 2], q = s.agent_pos[(n * A + a) * 2 + 1];
    o[0] = (float)r; o[1] = (float)q; o[2] = s.agent_carry[n * A + a] ? 1.f : 0.f;
    o[3 + s.agent_dir[n * A + a]] = 1.f;
    o[7] = rw_highway(c, r, q) ? 1.f : 0.f;
    int j = 8;
    for (int dr = -c.R; dr <= c.R; ++dr)
      for (int dc = -c.R; dc <= c.R; ++dc) {
        const int rr = r + dr, cc = q + dc;
        if (rr >= 0 && rr < c.H && cc >= 0 && cc < c.W) {
          const int ida = ga[rr * c.W + cc], ids = gs[rr * c.W + cc];
          if (ida > 0) { o[j] = 1.f; o[j + 1 + s.agent_dir[n * A + ida - 1]] = 1.f; }
          if (ids > 0) { o[j + 5] = 1.f; o[j + 6] = s.shelf_req[n * c.NS + ids - 1] ? 1.f : 0.f; }
        }
        j += 7;
      }
    for (int k = 0; k < RW_NACT; ++k) mask[a * RW_NACT + k] = s.amask[(n * A + a) * RW_NACT + k];
  }
}

__global__ __launch_bounds__(64) void k_rware_reset(RwState s, RwCfg c, const uint32_t* __restrict__ env_keys, float* __restrict__ obs, long ldo,
                                                    int* __restrict__ obs_step, unsigned char* __restrict__ mask) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= c.N) return;
  const uint32_t e0 = env_keys[2 * n], e1 = env_keys[2 * n + 1];
  uint32_t m0, m1, r0, r1;
  threefry2x32(e0, e1, 0u, 0u, m0, m1);  // key, reset_key = split(key)   (episode_metrics.py:62)
  threefry2x32(e0, e1, 0u, 1u, r0, r1);
  rw_generate(c, s, n, r0, r1);
  s.metrics_key[2 * n] = m0; s.metrics_key[2 * n + 1] = m1;
  s.run_ret[n] = 0.f; s.run_len[n] = 0; s.ep_ret[n] = 0.f; s.ep_len[n] = 0;
  rw_observe(c, s, n, obs + n * (long)c.A * ldo, ldo, mask + n * (long)c.A * RW_NACT);
  obs_step[n] = 0;
}

struct RwOut {
  float* reward; float* discount; unsigned char* done; float* obs; long ldo; int* obs_step; unsigned char* mask;
  float* m_ep_ret; int* m_ep_len; unsigned char* m_term;
};

__global__ __launch_bounds__(64) void k_rware_step(RwState s, RwCfg c, const int* __restrict__ actions, int act_stride, RwOut o, int auto_reset) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= c.N) return;
  const int A = c.A, HW = c.H * c.W;
  int* ga = s.grid_a + n * HW; int* gs = s.grid_s + n * HW;
  int act[RW_MAXA];
  for (int a = 0; a < A; ++a) {   // get_valid_actions: a masked action becomes NOOP
    int k = actions[n * act_stride + a];
    k = k < 0 ? 0 : (k >= RW_NACT ? RW_NACT - 1 : k);
    act[a] = s.amask[(n * A + a) * RW_NACT + k] ? k : 0;
  }
  for (int a = 0; a < A; ++a) {   // agents are updated one after the other, in id order
    const int r = s.agent_pos[(n * A + a) * 2], q = s.agent_pos[(n * A + a) * 2 + 1], d = s.agent_dir[n * A + a];
    if (act[a] == RW_FORWARD) {
      int nr, nc;
      rw_ahead(c, r, q, d, nr, nc);
      ga[r * c.W + q] = 0;
      ga[nr * c.W + nc] = a + 1;
      s.agent_pos[(n * A + a) * 2] = nr; s.agent_pos[(n * A + a) * 2 + 1] = nc;
      if (s.agent_carry[n * A + a]) {
        const int sid = gs[r * c.W + q];
        gs[r * c.W + q] = 0;
        gs[nr * c.W + nc] = sid;
      }
    } else if (act[a] == RW_LEFT) {
      s.agent_dir[n * A + a] = (d + 3) & 3;
    } else if (act[a] == RW_RIGHT) {
      s.agent_dir[n * A + a] = (d + 1) & 3;
    } else if (act[a] == RW_TOGGLE) {
      if (!s.agent_carry[n * A + a]) s.agent_carry[n * A + a] = gs[r * c.W + q] > 0 ? 1 : 0;
      else if (!rw_highway(c, r, q)) s.agent_carry[n * A + a] = 0;
    }
  }
  bool collision = false;
  for (int a = 0; a < A; ++a) collision |= ga[s.agent_pos[(n * A + a) * 2] * c.W + s.agent_pos[(n * A + a) * 2 + 1]] != a + 1;
  float reward = 0.f;
  uint32_t k0 = s.key[2 * n], k1 = s.key[2 * n + 1];
  unsigned char* req = s.shelf_req + n * c.NS;
  for (int g = 0; g < 2; ++g) {   // goals (H - 1, W/2 - 1), (H - 1, W/2) in order
    const int sid = gs[(c.H - 1) * c.W + c.W / 2 - 1 + g];
    if (sid > 0 && req[sid - 1]) {
      reward += 1.0f;
      uint32_t n0, n1, s0, s1;
      threefry2x32(k0, k1, 0u, 0u, n0, n1);   // key, sub = split(key)
      threefry2x32(k0, k1, 0u, 1u, s0, s1);
      k0 = n0; k1 = n1;
      // the replacement request: choice(sub, NS, (), replace=False, p=not requested) = Gumbel top-1 (oracle/prng.py:choice): the first
      // maximum of gumbel(sub, (NS,))[i] over the shelves that are not in the queue (the delivered one still counts as requested)
      int pick = 0;
      float best = -INFINITY;
      bool have = false;
      for (int i = 0; i < c.NS; ++i) {
        if (req[i]) continue;
        const float g = gumbel_exact_from_bits(random_bits32(s0, s1, (uint32_t)i));
        if (!have || g > best) { best = g; pick = i; have = true; }
      }
      for (int q = 0; q < c.Q; ++q) {
        if (s.queue[n * c.Q + q] == sid) { s.queue[n * c.Q + q] = pick + 1; break; }
      }
      req[sid - 1] = 0;
      req[pick] = 1;
    }
  }
  s.key[2 * n] = k0; s.key[2 * n + 1] = k1;
  const int steps = s.step_count[n] + 1;
  const bool done = collision || steps >= c.TLIM;
  int obs_step = steps;
  if (done && auto_reset) {
    uint32_t nk0, nk1;
    threefry2x32(k0, k1, 0u, 0u, nk0, nk1);  // key, _ = split(state.key)   (auto_reset_wrapper.py:74)
    rw_generate(c, s, n, nk0, nk1);
    obs_step = 0;
  } else {
    s.step_count[n] = steps;
    rw_mask(c, s, n);
  }
  rw_observe(c, s, n, o.obs + n * (long)A * o.ldo, o.ldo, o.mask + n * (long)A * RW_NACT);
  o.obs_step[n] = obs_step;
  for (int a = 0; a < A; ++a) o.reward[n * A + a] = reward;
  if (o.discount) for (int a = 0; a < A; ++a) o.discount[n * A + a] = done ? 0.f : 1.f;   // collision or horizon: termination
  o.done[n] = done ? 1 : 0;
  float msum = 0.f;   // episode_metrics.py:79-112: mean over agents of the repeated reward, as a sum / A in fp32
  for (int a = 0; a < A; ++a) msum += reward;
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

static int rw_cfg(RwCfg& c, int N, int A, int CH, int SR, int SC, int R, int Q, int TLIM) {
  c = RwCfg{N, A, CH, SR, SC, R, Q, TLIM, (CH + 1) * SR + 2, 3 * SC + 1, 0};
  if (A < 1 || A > RW_MAXA || CH < 1 || SR < 1 || SC < 1 || R < 1 || R > 2 || Q < 1 || TLIM < 1) {
    set_error("rware: 1 <= num_agents <= 8, sensor_range in {1, 2}, positive layout parameters");
    return MAGPO_EINVAL;
  }
  int ns = 0;
  for (int y = 0; y < c.H; ++y)
    for (int x = 0; x < c.W; ++x) ns += rw_highway(c, y, x) ? 0 : 1;
  c.NS = ns;
  if (Q >= ns || A >= c.H * c.W) { set_error("rware: request queue / agents do not fit the layout"); return MAGPO_EINVAL; }
  if (Q > RW_MAXPICK || c.H * c.W > 1625) { set_error("rware: request_queue_size <= 16 and at most 1625 cells (one-round shuffle)"); return MAGPO_EINVAL; }
  return MAGPO_OK;
}

// layout sizes for the caller's buffers: out[0] = H, out[1] = W, out[2] = number of shelves
extern "C" int magpo_rware_layout(int column_height, int shelf_rows, int shelf_columns, int* out) {
  RwCfg c;
  if (int e = rw_cfg(c, 1, 1, column_height, shelf_rows, shelf_columns, 1, 1, 1)) return e;
  out[0] = c.H; out[1] = c.W; out[2] = c.NS;
  return MAGPO_OK;
}

extern "C" int magpo_rware_reset(int* grid_a, int* grid_s, int* agent_pos, int* agent_dir, unsigned char* agent_carry, unsigned char* shelf_req,
                                 int* queue, int* step_count, unsigned char* amask, uint32_t* key, uint32_t* metrics_key, float* run_ret,
                                 int* run_len, float* ep_ret, int* ep_len, int N, int A, int column_height, int shelf_rows, int shelf_columns,
                                 int sensor_range, int queue_size, int time_limit, const uint32_t* env_keys, float* obs, long ldo, int* obs_step,
                                 unsigned char* mask, hipStream_t st) {
  RwCfg c;
  if (int e = rw_cfg(c, N, A, column_height, shelf_rows, shelf_columns, sensor_range, queue_size, time_limit)) return e;
  if (N <= 0) return MAGPO_OK;
  RwState s{grid_a, grid_s, agent_pos, agent_dir, agent_carry, shelf_req, queue, step_count, amask, key, metrics_key, run_ret, run_len, ep_ret, ep_len};
  hipLaunchKernelGGL(k_rware_reset, dim3((N + 63) / 64), dim3(64), 0, st, s, c, env_keys, obs, ldo, obs_step, mask);
  return check_launch("magpo_rware_reset");
}

extern "C" int magpo_rware_step(int* grid_a, int* grid_s, int* agent_pos, int* agent_dir, unsigned char* agent_carry, unsigned char* shelf_req,
                                int* queue, int* step_count, unsigned char* amask, uint32_t* key, uint32_t* metrics_key, float* run_ret,
                                int* run_len, float* ep_ret, int* ep_len, int N, int A, int column_height, int shelf_rows, int shelf_columns,
                                int sensor_range, int queue_size, int time_limit, const int* actions, int act_stride, float* reward,
                                float* discount, unsigned char* done, float* obs, long ldo, int* obs_step, unsigned char* mask, float* m_ep_ret, int* m_ep_len,
                                unsigned char* m_term, int auto_reset, hipStream_t st) {
  RwCfg c;
  if (int e = rw_cfg(c, N, A, column_height, shelf_rows, shelf_columns, sensor_range, queue_size, time_limit)) return e;
  if (N <= 0) return MAGPO_OK;
  RwState s{grid_a, grid_s, agent_pos, agent_dir, agent_carry, shelf_req, queue, step_count, amask, key, metrics_key, run_ret, run_len, ep_ret, ep_len};
  RwOut o{reward, discount, done, obs, ldo, obs_step, mask, m_ep_ret, m_ep_len, m_term};
  hipLaunchKernelGGL(k_rware_step, dim3((N + 63) / 64), dim3(64), 0, st, s, c, actions, act_stride, o, auto_reset);
  return check_launch("magpo_rware_step");
}
