"""Level-Based Foraging + Mava wrapper stack, batched numpy restatement (oracle; test infrastructure only).

Wrapper order (mava/utils/make_env.py:90-104,107-135):
  RecordEpisodeMetrics (wrappers/episode_metrics.py:60-112)
    -> AutoResetWrapper (wrappers/auto_reset_wrapper.py:60-101)      [train env only]
      -> AgentIDWrapper (wrappers/observation.py:42-54)
        -> LbfWrapper (wrappers/jumanji.py:171-220; aggregate_rewards is always on)
          -> jumanji LevelBasedForaging-v0 with RandomGenerator(**task_config)

**UNPINNED DYNAMICS.**  The environment itself lives in third-party Jumanji (1.1.0 @ git 9ced6b8, uv.lock:1217-1219), whose
source is NOT in /root/reference and which cannot be installed here.  What follows restates Jumanji's published algorithm
(jumanji/environments/routing/lbf: env.py step / get_reward, utils.py update_agent_positions / fix_collisions / eat_food /
compute_action_mask, observer.py VectorObserver, generator.py RandomGenerator) from memory:
  * actions NOOP, UP (-1,0), DOWN (+1,0), LEFT (0,-1), RIGHT (0,+1), LOAD; positions are (row, col)
  * move: an agent stays put if its target cell is outside the grid, holds an uneaten food or holds another agent (at its
    position BEFORE the move); afterwards every agent whose new cell is shared with another agent returns to its old cell
  * loading = (action == LOAD); a food is eaten when the summed levels of the adjacent (|dr| + |dc| == 1) loading agents
    reach its level
  * reward of agent a for food f = level_a [adjacent & loading] * eaten_now * level_f / (sum of adjacent loading levels *
    total food level)  (normalize_reward = True, penalty = 0); the wrapper sums it over agents (team reward)
  * episode ends when all food is eaten (termination) or step_count >= time_limit (truncation)
  * vector observation per agent: (x, y, level) of every food, then of itself, then of the other agents in id order;
    entities outside the field of view (|d| > fov on an axis) and eaten food read (-1, -1, 0); coordinates are shifted by
    min(fov, own position) - own position (absolute coordinates when fov >= grid_size)
  * action mask: a move is legal if the target cell is inside the grid and free; NOOP always; LOAD iff an uneaten food is adjacent
  * generator: food on non-border cells, no two foods in the same or in 4-adjacent cells; agents on free cells; agent levels
    uniform in [1, max_agent_level]; force_coop: every food level = sum of the (up to three) smallest agent levels
The generator's cell draws follow jax.random.choice's published algorithm (oracle/prng.py:choice) with the call forms of
Jumanji's RandomGenerator as recalled: every food cell is one `choice(key_f, G*G, shape=(), p=mask)` (replace=True: cumulative
counts of the boolean mask, one uniform, searchsorted) with key_f = split(key_food, NF)[f] and the mask updated between the
draws; the agents are ONE `choice(key_agents, G*G, shape=(A,), replace=False, p=mask)` (Gumbel top-k over the cells that hold
no food).  Levels use the exact jax.random.randint restatement (oracle/prng.py).  A mask with no valid cell left yields cell 0
(what choice returns for an all-zero p); make_lbf_env / magpo_lbf_* reject configurations where that can happen.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import prng

STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2
MOVES = np.array([[0, 0], [-1, 0], [1, 0], [0, -1], [0, 1], [0, 0]], np.int32)
LOAD = 5
NUM_ACTIONS = 6


class LbfSpec:
    def __init__(self, grid_size=8, fov=8, num_agents=2, num_food=2, max_agent_level=2, force_coop=True, time_limit=100):
        self.grid_size, self.fov, self.num_agents, self.num_food = int(grid_size), int(fov), int(num_agents), int(num_food)
        self.max_agent_level, self.force_coop, self.time_limit = int(max_agent_level), bool(force_coop), int(time_limit)
        self.num_actions = NUM_ACTIONS
        self.add_agent_id = True   # system.add_agent_id (make_env.py:90-104): False = AgentIDWrapper is not applied

    @property
    def obs_dim(self) -> int:   # vector observation + one-hot agent id (AgentIDWrapper)
        return 3 * (self.num_food + self.num_agents) + (self.num_agents if self.add_agent_id else 0)


def _generate(spec: LbfSpec, key: np.ndarray) -> Dict[str, np.ndarray]:
    """RandomGenerator.__call__ for one key."""
    G, A, NF = spec.grid_size, spec.num_agents, spec.num_food
    ks = prng.split(key, 5)   # key_food, key_agents, key_food_level, key_agent_level, key
    key_food, key_agents, key_food_level, key_agent_level, key_state = ks
    valid = np.ones((G, G), bool)
    valid[0, :] = valid[-1, :] = valid[:, 0] = valid[:, -1] = False
    food_pos = np.zeros((NF, 2), np.int32)
    fkeys = prng.split(key_food, NF)
    for f in range(NF):
        c = int(prng.choice(fkeys[f], G * G, 1, True, valid.reshape(-1))[0])   # take_positions: choice(key, flat_size, (), p=mask)
        r, q = divmod(c, G)
        food_pos[f] = (r, q)
        for dr, dc in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)):
            if 0 <= r + dr < G and 0 <= q + dc < G:
                valid[r + dr, q + dc] = False
    free = np.ones((G, G), bool)
    free[food_pos[:, 0], food_pos[:, 1]] = False
    cells = prng.choice(key_agents, G * G, A, False, free.reshape(-1))   # sample_agents: choice(key, G*G, (A,), replace=False, p=mask)
    agent_pos = np.stack(np.divmod(cells, G), axis=1).astype(np.int32)
    agent_level = prng.randint(key_agent_level, A, 1, spec.max_agent_level + 1).astype(np.int32)
    max_food_level = int(np.sort(agent_level)[:3].sum())
    if spec.force_coop:
        food_level = np.full(NF, max_food_level, np.int32)
    else:
        food_level = prng.randint(key_food_level, NF, 1, max_food_level + 1).astype(np.int32)
    return dict(agent_pos=agent_pos, agent_level=agent_level, agent_loading=np.zeros(A, bool), food_pos=food_pos, food_level=food_level,
                food_eaten=np.zeros(NF, bool), step_count=np.int32(0), key=key_state.copy())


def _core_reset(spec: LbfSpec, keys: np.ndarray) -> Dict[str, np.ndarray]:
    sts = [_generate(spec, k) for k in keys]
    return {f: np.stack([s[f] for s in sts]) for f in sts[0]}


def _observe(spec: LbfSpec, st: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """VectorObserver.state_to_observation + compute_action_mask for a batch: agents_view (N, A, 3 (NF + A)) int32,
    action_mask (N, A, 6) bool."""
    N, A, NF, G, fov = st["agent_pos"].shape[0], spec.num_agents, spec.num_food, spec.grid_size, spec.fov
    view = np.zeros((N, A, 3 * (NF + A)), np.int32)
    mask = np.zeros((N, A, NUM_ACTIONS), bool)
    ap, al, fp, fl, fe = st["agent_pos"], st["agent_level"], st["food_pos"], st["food_level"], st["food_eaten"]
    for a in range(A):
        me = ap[:, a]                                   # (N, 2)
        shift = np.minimum(fov, me) - me                # transform_positions
        for f in range(NF):
            vis = (np.abs(fp[:, f] - me) <= fov).all(1) & ~fe[:, f]
            view[:, a, 3 * f] = np.where(vis, fp[:, f, 0] + shift[:, 0], -1)
            view[:, a, 3 * f + 1] = np.where(vis, fp[:, f, 1] + shift[:, 1], -1)
            view[:, a, 3 * f + 2] = np.where(vis, fl[:, f], 0)
        o = 3 * NF
        view[:, a, o] = me[:, 0] + shift[:, 0]
        view[:, a, o + 1] = me[:, 1] + shift[:, 1]
        view[:, a, o + 2] = al[:, a]
        j = 1
        for b in range(A):
            if b == a:
                continue
            vis = (np.abs(ap[:, b] - me) <= fov).all(1)
            view[:, a, o + 3 * j] = np.where(vis, ap[:, b, 0] + shift[:, 0], -1)
            view[:, a, o + 3 * j + 1] = np.where(vis, ap[:, b, 1] + shift[:, 1], -1)
            view[:, a, o + 3 * j + 2] = np.where(vis, al[:, b], 0)
            j += 1
        for k in range(NUM_ACTIONS):
            nxt = me + MOVES[k]
            oob = ((nxt < 0) | (nxt >= G)).any(1)
            occ_a = np.zeros(N, bool)
            for b in range(A):
                if b != a:
                    occ_a |= (ap[:, b] == nxt).all(1)
            occ_f = np.zeros(N, bool)
            for f in range(NF):
                occ_f |= (fp[:, f] == nxt).all(1) & ~fe[:, f]
            mask[:, a, k] = ~(oob | occ_a | occ_f)
        adj = np.zeros(N, bool)
        for f in range(NF):
            adj |= (np.abs(fp[:, f] - me).sum(1) == 1) & ~fe[:, f]
        mask[:, a, LOAD] &= adj
    return view, mask


def make_obs(spec: LbfSpec, st: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """LbfWrapper.modify_timestep (agents_view as float) + AgentIDWrapper (one-hot id in front, observation.py:42-54)."""
    view, mask = _observe(spec, st)
    N, A = view.shape[0], spec.num_agents
    ids = np.broadcast_to(np.eye(A, dtype=np.float32)[None], (N, A, A))
    full = np.concatenate([ids, view.astype(np.float32)], axis=-1) if getattr(spec, "add_agent_id", True) else view.astype(np.float32)
    return dict(agents_view=full, action_mask=mask,
                step_count=np.repeat(st["step_count"][:, None], A, axis=1).astype(np.int32))


def reset(spec: LbfSpec, env_keys: np.ndarray) -> Tuple[Dict, Dict]:
    """RecordEpisodeMetrics.reset (episode_metrics.py:60-77) around LevelBasedForaging.reset."""
    ks = prng.split(env_keys, 2)   # key (kept, unused), reset_key
    core = _core_reset(spec, ks[:, 1, :])
    n, a = env_keys.shape[0], spec.num_agents
    state = dict(core, metrics_key=ks[:, 0, :].copy(), running_return=np.zeros(n, np.float32), running_length=np.zeros(n, np.int32),
                 episode_return=np.zeros(n, np.float32), episode_length=np.zeros(n, np.int32))
    timestep = dict(step_type=np.full(n, STEP_FIRST, np.int8), reward=np.zeros((n, a), np.float32), discount=np.ones((n, a), np.float32),
                    observation=make_obs(spec, core),
                    episode_metrics=dict(episode_return=np.zeros(n, np.float32), episode_length=np.zeros(n, np.int32),
                                         is_terminal_step=np.zeros(n, bool)))
    return state, timestep


_CORE = ("agent_pos", "agent_level", "agent_loading", "food_pos", "food_level", "food_eaten", "step_count", "key")


def _core_step(spec: LbfSpec, st: Dict[str, np.ndarray], actions: np.ndarray):
    """LevelBasedForaging.step for a batch; returns (new core state, per-agent reward (N, A) float32, terminate, truncate)."""
    N, A, NF, G = actions.shape[0], spec.num_agents, spec.num_food, spec.grid_size
    ap, fp, fe, fl, al = st["agent_pos"], st["food_pos"], st["food_eaten"], st["food_level"], st["agent_level"]
    new = ap.copy()
    for a in range(A):   # simulate_agent_movement against the positions BEFORE the move
        nxt = ap[:, a] + MOVES[actions[:, a]]
        blocked = ((nxt < 0) | (nxt >= G)).any(1)
        for b in range(A):
            if b != a:
                blocked |= (ap[:, b] == nxt).all(1)
        for f in range(NF):
            blocked |= (fp[:, f] == nxt).all(1) & ~fe[:, f]
        new[:, a] = np.where(blocked[:, None], ap[:, a], nxt)
    dup = np.zeros((N, A), bool)   # fix_collisions
    for a in range(A):
        for b in range(A):
            if b != a:
                dup[:, a] |= (new[:, a] == new[:, b]).all(1)
    new = np.where(dup[:, :, None], ap, new)
    loading = actions == LOAD
    reward = np.zeros((N, A), np.float32)
    eaten = fe.copy()
    total_food_level = fl.sum(1).astype(np.float32)
    for f in range(NF):   # eat_food + get_reward_per_food
        adj = (np.abs(new - fp[:, f][:, None, :]).sum(2) == 1) & loading & ~fe[:, f][:, None]
        lv = np.where(adj, al, 0)                       # adj_loading_agents_levels (N, A)
        s = lv.sum(1)
        eaten_now = (s >= fl[:, f]) & ~fe[:, f] & (s > 0)
        # (jumanji: eaten_this_step = sum >= level; an already eaten food has adj levels 0, so the sum is 0 < level)
        norm = s.astype(np.float32) * total_food_level
        r = (lv * (eaten_now[:, None] * fl[:, f][:, None])).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(norm[:, None] > 0, r / norm[:, None], np.float32(0.0))   # nan_to_num(0 / 0) = 0
        reward += r.astype(np.float32)
        eaten[:, f] |= eaten_now
    steps = st["step_count"] + 1
    core = dict(agent_pos=new.astype(np.int32), agent_level=al, agent_loading=loading, food_pos=fp, food_level=fl, food_eaten=eaten,
                step_count=steps.astype(np.int32), key=st["key"])
    return core, reward, eaten.all(1), steps >= spec.time_limit


def step(spec: LbfSpec, state: Dict, actions: np.ndarray, auto_reset: bool = True) -> Tuple[Dict, Dict]:
    """One step of the wrapped train env (auto_reset=True) or eval env (False)."""
    actions = np.asarray(actions, np.int32)
    a = spec.num_agents
    core, reward, terminate, truncate = _core_step(spec, {k: state[k] for k in _CORE}, actions)
    done = terminate | truncate
    tsum = np.zeros(reward.shape[0], np.float32)
    for i in range(a):   # aggregate_rewards (jumanji.py:43-46); explicit left-to-right fp32 sum (numpy's pairwise sum reorders at 8 terms)
        tsum = (tsum + reward[:, i]).astype(np.float32)
    team = np.repeat(tsum[:, None], a, axis=1)
    obs_state = core
    if auto_reset and done.any():   # auto_reset_wrapper.py:60-83: key, _ = split(state.key); reset(key); keep reward etc.
        idx = np.nonzero(done)[0]
        fresh = _core_reset(spec, prng.split(core["key"][idx], 2)[:, 0, :])
        core = {k: v.copy() for k, v in core.items()}
        for k in _CORE:
            core[k][idx] = fresh[k]
        obs_state = core
    discount = np.repeat(np.where(terminate, 0.0, 1.0).astype(np.float32)[:, None], a, axis=1)
    not_done = (~done).astype(np.float32)
    msum = np.zeros(reward.shape[0], np.float32)
    for i in range(a):
        msum = (msum + team[:, i]).astype(np.float32)
    new_ret = (state["running_return"] + msum / np.float32(a)).astype(np.float32)   # episode_metrics.py:91-96: mean over agents
    new_len = state["running_length"] + 1
    ep_ret = (state["episode_return"] * not_done + new_ret * done).astype(np.float32)
    ep_len = np.where(done, new_len, state["episode_length"]).astype(np.int32)
    new_state = dict(core, metrics_key=state["metrics_key"], running_return=(new_ret * not_done).astype(np.float32),
                     running_length=np.where(done, 0, new_len).astype(np.int32), episode_return=ep_ret, episode_length=ep_len)
    timestep = dict(step_type=np.where(done, STEP_LAST, STEP_MID).astype(np.int8), reward=team, discount=discount,
                    observation=make_obs(spec, obs_state),
                    episode_metrics=dict(episode_return=ep_ret, episode_length=ep_len, is_terminal_step=done.copy()))
    return new_state, timestep
