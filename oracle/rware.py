"""Robot Warehouse + Mava wrapper stack, batched numpy restatement (oracle; test infrastructure only).

Wrapper order (mava/utils/make_env.py:90-104,107-135):
  RecordEpisodeMetrics (wrappers/episode_metrics.py:60-112)
    -> AutoResetWrapper (wrappers/auto_reset_wrapper.py:60-101)      [train env only]
      -> AgentIDWrapper (wrappers/observation.py:42-54)
        -> RwareWrapper (wrappers/jumanji.py:137-168: obs as float, the scalar reward / discount repeated per agent)
          -> jumanji RobotWarehouse-v0 with RandomGenerator(**task_config), time_limit 500 (configs/env/rware.yaml:20)

**UNPINNED DYNAMICS.**  The environment lives in third-party Jumanji (1.1.0 @ git 9ced6b8), whose source is NOT in
/root/reference and cannot be installed here; this restates its published algorithm (jumanji/environments/routing/
robot_warehouse: env.py, utils*.py, generator.py -- itself a JAX port of github.com/semitable/robotic-warehouse) from memory:
  * layout (rware `_make_layout_from_params`): H = (column_height + 1) * shelf_rows + 2 rows, W = 3 * shelf_columns + 1 columns;
    a cell (row y, col x) is a highway if x % 3 == 0, y % (column_height + 1) == 0, y == H - 1, or it lies in the goal corridor
    (y > H - (column_height + 3) and x in {W//2 - 1, W//2}); every other cell holds a shelf (ids 1.. in row-major order);
    goals = (H - 1, W//2 - 1), (H - 1, W//2)
  * state: agents layer / shelves layer of the grid (0 = empty, id + 1 otherwise), agent position / direction (0 up, 1 right,
    2 down, 3 left) / carrying flag, shelf requested flags, request queue, step count, action mask, key (a shelf's position is
    where its id stands in the shelves layer)
  * actions NOOP, FORWARD, LEFT (dir - 1), RIGHT (dir + 1), TOGGLE_LOAD; an action whose mask entry is False becomes NOOP
  * the agents are updated ONE AFTER THE OTHER in id order: FORWARD moves to the (grid-clipped) cell ahead, writing the agents
    layer (old cell 0, new cell id + 1) and, if carrying, the shelf with it; TOGGLE_LOAD picks up the shelf under a free agent
    or puts a carried shelf down unless the cell is a highway
  * collision = some agent's cell of the agents layer no longer holds its id (two agents entered one cell, or one entered a
    cell its owner left in the same step); a collision or step_count >= time_limit ends the episode (termination, discount 0)
  * reward: for each goal in order, a requested shelf standing on it gives +1 to the shared reward, leaves the request queue
    and a shelf that is not requested takes its slot (key, sub = split(key) per delivery)
  * action mask (computed on the state AFTER the step, used to sanitise the NEXT step's actions): only FORWARD can be
    illegal -- when the cell ahead (after clipping, so the agent's own cell at a wall) holds an agent, or holds a shelf while
    the agent is carrying one
  * observation per agent, sensor_range r: [row, col, carrying, one-hot direction (4), on highway] then, for every cell of the
    (2r + 1)^2 window in row-major order, [agent present, one-hot direction of that agent (4), shelf present, shelf requested];
    cells outside the grid read zeros; 8 + 7 (2r + 1)^2 features (71 at r = 1)
The generator's and the request queue's draws follow jax.random.choice's published algorithm (oracle/prng.py:choice) with the
call forms of Jumanji's code as recalled: agent cells = `choice(key_pos, H*W, (A,), replace=False)` (no p: the first A entries of
permutation(key, H*W)), the initial queue = `choice(key_queue, NS, (Q,), replace=False)` (same form over the shelf ids), and the
shelf that replaces a delivered request = `choice(sub, NS, (), replace=False, p=not_requested)` (p given, replace=False: Gumbel
top-1 over the shelves that are not in the queue).  Directions use the exact jax.random.randint restatement.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import prng

STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2
NOOP, FORWARD, LEFT, RIGHT, TOGGLE = 0, 1, 2, 3, 4
NUM_ACTIONS = 5
DR = np.array([-1, 0, 1, 0], np.int32)   # up, right, down, left
DC = np.array([0, 1, 0, -1], np.int32)


class RwareSpec:
    def __init__(self, column_height=8, shelf_rows=1, shelf_columns=3, num_agents=4, sensor_range=1, request_queue_size=4, time_limit=500):
        self.column_height, self.shelf_rows, self.shelf_columns = int(column_height), int(shelf_rows), int(shelf_columns)
        self.num_agents, self.sensor_range, self.request_queue_size, self.time_limit = int(num_agents), int(sensor_range), int(request_queue_size), int(time_limit)
        self.num_actions = NUM_ACTIONS
        self.H = (self.column_height + 1) * self.shelf_rows + 2
        self.W = 3 * self.shelf_columns + 1
        H, W = self.H, self.W
        hw = np.zeros((H, W), bool)
        for y in range(H):
            for x in range(W):
                hw[y, x] = (x % 3 == 0) or (y % (self.column_height + 1) == 0) or (y == H - 1) or \
                           (y > H - (self.column_height + 3) and x in (W // 2 - 1, W // 2))
        self.highway = hw
        self.goals = [(H - 1, W // 2 - 1), (H - 1, W // 2)]
        self.shelf_cells = np.argwhere(~hw).astype(np.int32)      # (NS, 2) row-major: shelf id = index + 1
        self.num_shelves = self.shelf_cells.shape[0]

    @property
    def obs_dim(self) -> int:   # vector observation + one-hot agent id (AgentIDWrapper)
        return 8 + 7 * (2 * self.sensor_range + 1) ** 2 + self.num_agents


def _generate(spec: RwareSpec, key: np.ndarray) -> Dict[str, np.ndarray]:
    H, W, A, NS, Q = spec.H, spec.W, spec.num_agents, spec.num_shelves, spec.request_queue_size
    ks = prng.split(key, 4)   # key_pos, key_dir, key_queue, key
    key_pos, key_dir, key_queue, key_state = ks
    cells = prng.choice(key_pos, H * W, A, False)             # choice(key, H*W, (A,), replace=False)
    pos = np.stack(np.divmod(cells, W), axis=1).astype(np.int32)
    direction = prng.randint(key_dir, A, 0, 4).astype(np.int32)
    picks = prng.choice(key_queue, NS, Q, False)              # choice(key, shelf_ids, (Q,), replace=False)
    requested = np.zeros(NS, bool)
    requested[picks] = True
    queue = (picks + 1).astype(np.int32)
    grid_a = np.zeros((H, W), np.int32)
    for a in range(A):
        grid_a[pos[a, 0], pos[a, 1]] = a + 1
    grid_s = np.zeros((H, W), np.int32)
    grid_s[spec.shelf_cells[:, 0], spec.shelf_cells[:, 1]] = np.arange(1, NS + 1)
    st = dict(grid_a=grid_a, grid_s=grid_s, agent_pos=pos, agent_dir=direction, agent_carry=np.zeros(A, bool),
              shelf_req=requested, queue=queue, step_count=np.int32(0), key=key_state.copy())
    st["action_mask"] = _action_mask(spec, st)
    return st


def _ahead(spec: RwareSpec, pos, d):
    return min(max(pos[0] + DR[d], 0), spec.H - 1), min(max(pos[1] + DC[d], 0), spec.W - 1)


def _action_mask(spec: RwareSpec, st) -> np.ndarray:
    A = spec.num_agents
    m = np.ones((A, NUM_ACTIONS), bool)
    for a in range(A):
        r, c = _ahead(spec, st["agent_pos"][a], st["agent_dir"][a])
        m[a, FORWARD] = not (st["grid_a"][r, c] > 0 or (st["agent_carry"][a] and st["grid_s"][r, c] > 0))
    return m


def _observe(spec: RwareSpec, st) -> np.ndarray:
    A, R = spec.num_agents, spec.sensor_range
    nf = 8 + 7 * (2 * R + 1) ** 2
    out = np.zeros((A, nf), np.int32)
    for a in range(A):
        r, c = st["agent_pos"][a]
        o = out[a]
        o[0], o[1], o[2] = r, c, int(st["agent_carry"][a])
        o[3 + st["agent_dir"][a]] = 1
        o[7] = int(spec.highway[r, c])
        j = 8
        for dr in range(-R, R + 1):
            for dc in range(-R, R + 1):
                rr, cc = r + dr, c + dc
                if 0 <= rr < spec.H and 0 <= cc < spec.W:
                    ida, ids = st["grid_a"][rr, cc], st["grid_s"][rr, cc]
                    if ida > 0:
                        o[j] = 1
                        o[j + 1 + st["agent_dir"][ida - 1]] = 1
                    if ids > 0:
                        o[j + 5] = 1
                        o[j + 6] = int(st["shelf_req"][ids - 1])
                j += 7
    return out


def _step_one(spec: RwareSpec, st, actions):
    """RobotWarehouse.step for one env (state dict of arrays without the env axis); returns (new state, reward, done)."""
    A = spec.num_agents
    st = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}
    ga, gs = st["grid_a"], st["grid_s"]
    acts = [int(actions[a]) if st["action_mask"][a, min(max(int(actions[a]), 0), NUM_ACTIONS - 1)] else NOOP for a in range(A)]
    for a in range(A):
        act, (r, c), d = acts[a], st["agent_pos"][a], st["agent_dir"][a]
        if act == FORWARD:
            nr, nc = _ahead(spec, (r, c), d)
            ga[r, c] = 0
            ga[nr, nc] = a + 1
            st["agent_pos"][a] = (nr, nc)
            if st["agent_carry"][a]:
                sid = gs[r, c]
                gs[r, c] = 0
                gs[nr, nc] = sid   # (the shelf's own position record follows the grid; nothing reads it)
        elif act == LEFT:
            st["agent_dir"][a] = (d + 3) % 4
        elif act == RIGHT:
            st["agent_dir"][a] = (d + 1) % 4
        elif act == TOGGLE:
            if not st["agent_carry"][a]:
                st["agent_carry"][a] = gs[r, c] > 0
            elif not spec.highway[r, c]:
                st["agent_carry"][a] = False
    collision = any(ga[st["agent_pos"][a][0], st["agent_pos"][a][1]] != a + 1 for a in range(A))
    reward = np.float32(0.0)
    key = st["key"]
    for (gr, gc) in spec.goals:
        sid = gs[gr, gc]
        if sid > 0 and st["shelf_req"][sid - 1]:
            reward = np.float32(reward + np.float32(1.0))
            ks = prng.split(key, 2)
            key, sub = ks[0], ks[1]
            new = int(prng.choice(sub, spec.num_shelves, 1, False, ~st["shelf_req"])[0])   # not_in_queue mask still holds the delivered shelf as requested
            slot = int(np.nonzero(st["queue"] == sid)[0][0])
            st["queue"][slot] = new + 1
            st["shelf_req"][sid - 1] = False
            st["shelf_req"][new] = True
    st["key"] = key
    steps = st["step_count"] + 1
    st["step_count"] = np.int32(steps)
    st["action_mask"] = _action_mask(spec, st)
    return st, reward, bool(collision or steps >= spec.time_limit)


_CORE = ("grid_a", "grid_s", "agent_pos", "agent_dir", "agent_carry", "shelf_req", "queue", "step_count", "key", "action_mask")


def _stack(sts):
    return {f: np.stack([s[f] for s in sts]) for f in _CORE}


def _unstack(st, n):
    return {f: st[f][n] for f in _CORE}


def make_obs(spec: RwareSpec, st: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """RwareWrapper.modify_timestep (agents_view as float) + AgentIDWrapper (one-hot id in front)."""
    N, A = st["agent_pos"].shape[0], spec.num_agents
    view = np.stack([_observe(spec, _unstack(st, n)) for n in range(N)]).astype(np.float32)
    ids = np.broadcast_to(np.eye(A, dtype=np.float32)[None], (N, A, A))
    return dict(agents_view=np.concatenate([ids, view], axis=-1), action_mask=st["action_mask"].copy(),
                step_count=np.repeat(st["step_count"][:, None], A, axis=1).astype(np.int32))


def reset(spec: RwareSpec, env_keys: np.ndarray) -> Tuple[Dict, Dict]:
    ks = prng.split(env_keys, 2)   # key (kept, unused), reset_key  (episode_metrics.py:62)
    core = _stack([_generate(spec, k) for k in ks[:, 1, :]])
    n, a = env_keys.shape[0], spec.num_agents
    state = dict(core, metrics_key=ks[:, 0, :].copy(), running_return=np.zeros(n, np.float32), running_length=np.zeros(n, np.int32),
                 episode_return=np.zeros(n, np.float32), episode_length=np.zeros(n, np.int32))
    timestep = dict(step_type=np.full(n, STEP_FIRST, np.int8), reward=np.zeros((n, a), np.float32), discount=np.ones((n, a), np.float32),
                    observation=make_obs(spec, core),
                    episode_metrics=dict(episode_return=np.zeros(n, np.float32), episode_length=np.zeros(n, np.int32),
                                         is_terminal_step=np.zeros(n, bool)))
    return state, timestep


def step(spec: RwareSpec, state: Dict, actions: np.ndarray, auto_reset: bool = True) -> Tuple[Dict, Dict]:
    actions = np.asarray(actions, np.int32)
    N, a = actions.shape[0], spec.num_agents
    res = [_step_one(spec, _unstack(state, n), actions[n]) for n in range(N)]
    core = _stack([r[0] for r in res])
    reward = np.array([r[1] for r in res], np.float32)
    done = np.array([r[2] for r in res], bool)
    if auto_reset and done.any():   # auto_reset_wrapper.py:60-83: key, _ = split(state.key); reset(key); reward etc. kept
        idx = np.nonzero(done)[0]
        fresh = _stack([_generate(spec, k) for k in prng.split(core["key"][idx], 2)[:, 0, :]])
        for k in _CORE:
            core[k][idx] = fresh[k]
    rewards = np.repeat(reward[:, None], a, axis=1)
    discount = np.repeat(np.where(done, 0.0, 1.0).astype(np.float32)[:, None], a, axis=1)
    not_done = (~done).astype(np.float32)
    msum = np.zeros(N, np.float32)
    for i in range(a):
        msum = (msum + rewards[:, i]).astype(np.float32)
    new_ret = (state["running_return"] + msum / np.float32(a)).astype(np.float32)   # episode_metrics.py:91-96
    new_len = state["running_length"] + 1
    ep_ret = (state["episode_return"] * not_done + new_ret * done).astype(np.float32)
    ep_len = np.where(done, new_len, state["episode_length"]).astype(np.int32)
    new_state = dict(core, metrics_key=state["metrics_key"], running_return=(new_ret * not_done).astype(np.float32),
                     running_length=np.where(done, 0, new_len).astype(np.int32), episode_return=ep_ret, episode_length=ep_len)
    timestep = dict(step_type=np.where(done, STEP_LAST, STEP_MID).astype(np.int8), reward=rewards, discount=discount,
                    observation=make_obs(spec, core),
                    episode_metrics=dict(episode_return=ep_ret, episode_length=ep_len, is_terminal_step=done.copy()))
    return new_state, timestep
