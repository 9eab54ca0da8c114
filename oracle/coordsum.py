"""CoordSum environment + Mava wrapper stack, batched numpy restatement (oracle).

Follows, in wrapper order (mava/utils/make_env.py:90-104, 202-218):
  RecordEpisodeMetrics (wrappers/episode_metrics.py:60-112)
    -> AutoResetWrapper (wrappers/auto_reset_wrapper.py:60-101)      [train env only]
      -> AgentIDWrapper (wrappers/observation.py:42-54)
        -> CoordSumWrapper (wrappers/matrax.py:104-142)
          -> CoordSum (coordsum/env.py:55-139)

All envs of a batch are stepped together; state is a dict of arrays with a leading
env axis.  JAX out-of-bounds semantics are restated explicitly (gather indices and
dynamic_update_slice starts are clamped): coordsum/env.py:85,105-109,115.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import prng

STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2


class CoordSumSpec:
    def __init__(self, num_agents: int, num_actions: int, time_limit: int = 100, maxval=None):
        # coordsum/env.py:40-53
        self.num_agents = int(num_agents)
        self.num_actions = int(num_actions)
        self.time_limit = int(time_limit)
        self.maxval = int(maxval) if maxval else int(num_actions)
        self.add_agent_id = True   # system.add_agent_id (make_env.py:90-104): False = AgentIDWrapper is not applied

    @property
    def obs_dim(self) -> int:  # AgentIDWrapper: one-hot id + 1 feature
        return (self.num_agents if self.add_agent_id else 0) + 1


REGISTRY = {  # coordsum/__init__.py:6-45
    "5x20-80-v0": dict(num_agents=5, num_actions=20, time_limit=100, maxval=80),
    "3x30-50-v0": dict(num_agents=3, num_actions=30, time_limit=100, maxval=50),
    "3x10-30-v0": dict(num_agents=3, num_actions=10, time_limit=100, maxval=30),
    "8x15-100-v0": dict(num_agents=8, num_actions=15, time_limit=100, maxval=100),
}


def _core_reset(spec: CoordSumSpec, keys: np.ndarray) -> Dict[str, np.ndarray]:
    """CoordSum.reset for a batch of keys (N,2). coordsum/env.py:55-74."""
    n = keys.shape[0]
    ks = prng.split(keys, 2)  # (N,2,2): key, target_key
    target = prng.randint(ks[:, 1, :], spec.time_limit + 1, 0, spec.maxval)  # (N, T_lim+1)
    return dict(
        step_count=np.zeros(n, np.int32),
        target=target.astype(np.int32),
        record=-np.ones((n, spec.num_actions, spec.time_limit), np.int32),
        key=ks[:, 0, :].copy(),
    )


def make_obs(spec: CoordSumSpec, target_val: np.ndarray, step_count: np.ndarray) -> Dict[str, np.ndarray]:
    """CoordSumWrapper.modify_timestep + AgentIDWrapper._add_agent_ids.
    agents_view (N, A, A+1) int32 = [eye(A) | target]; matrax.py:117-134, observation.py:42-54."""
    n, a = target_val.shape[0], spec.num_agents
    view = np.zeros((n, a, a + 1), np.int32)
    view[:, :, :a] = np.eye(a, dtype=np.int32)[None]
    view[:, :, a] = target_val[:, None]
    if not getattr(spec, "add_agent_id", True):   # no AgentIDWrapper: the CoordSumWrapper's view alone (matrax.py:117-134)
        view = view[:, :, a:]
    return dict(
        agents_view=view,
        action_mask=np.ones((n, a, spec.num_actions), bool),
        step_count=np.repeat(step_count[:, None], a, axis=1).astype(np.int32),
    )


def reset(spec: CoordSumSpec, env_keys: np.ndarray) -> Tuple[Dict, Dict]:
    """Full train-env reset for per-env keys (N,2): RecordEpisodeMetrics.reset
    (episode_metrics.py:60-77) around CoordSum.reset."""
    ks = prng.split(env_keys, 2)  # key (kept, unused), reset_key
    core = _core_reset(spec, ks[:, 1, :])
    n = env_keys.shape[0]
    state = dict(
        core,
        metrics_key=ks[:, 0, :].copy(),
        running_return=np.zeros(n, np.float32),
        running_length=np.zeros(n, np.int32),
        episode_return=np.zeros(n, np.float32),
        episode_length=np.zeros(n, np.int32),
    )
    timestep = dict(
        step_type=np.full(n, STEP_FIRST, np.int8),
        reward=np.zeros((n, spec.num_agents), np.float32),
        discount=np.ones((n, spec.num_agents), np.float32),
        observation=make_obs(spec, core["target"][:, 0], core["step_count"]),
        episode_metrics=dict(
            episode_return=np.zeros(n, np.float32),
            episode_length=np.zeros(n, np.int32),
            is_terminal_step=np.zeros(n, bool),
        ),
    )
    return state, timestep


def _core_step(spec: CoordSumSpec, st: Dict, actions: np.ndarray):
    """CoordSum.step (coordsum/env.py:76-139) for a batch."""
    n = actions.shape[0]
    K, TL = spec.num_actions, spec.time_limit
    ar = np.arange(n)
    t = st["step_count"]
    g = st["target"][ar, np.minimum(t, TL)]  # gather clamps
    sum_match = actions.sum(axis=1) == g
    row_idx = np.minimum(g, K - 1)  # gather clamps
    row = st["record"][ar, row_idx]  # (N, TL)
    valid = (row != -1) & (row < TL)  # bincount(length=TL) drops out-of-range values
    # bincount(length=TL) of the valid entries, argmax takes the first maximum
    counts = np.zeros((n, TL), np.float32)
    safe = np.where(valid, row, 0)
    np.add.at(counts, (np.repeat(ar, TL), safe.reshape(-1)), valid.reshape(-1).astype(np.float32))
    guess = np.argmax(counts, axis=1)
    hit = guess == actions[:, 0]
    reward = np.where(sum_match, np.where(hit, 1.0, 2.0), 0.0).astype(np.float32)
    record = st["record"].copy()
    record[ar, row_idx, np.minimum(t, TL - 1)] = actions[:, 0]  # dynamic_update_slice clamps
    steps = t + 1
    done = steps >= TL
    next_target = st["target"][ar, np.minimum(steps, TL)]
    new = dict(step_count=steps.astype(np.int32), target=st["target"], record=record, key=st["key"])
    return new, reward, done, next_target


def step(spec: CoordSumSpec, state: Dict, actions: np.ndarray, auto_reset: bool = True) -> Tuple[Dict, Dict]:
    """One step of the wrapped train env (auto_reset=True) or eval env (False)."""
    actions = np.asarray(actions, np.int32)
    a = spec.num_agents
    core_in = {k: state[k] for k in ("step_count", "target", "record", "key")}
    core, reward, done, next_target = _core_step(spec, core_in, actions)
    obs_target, obs_step = next_target, core["step_count"]
    if auto_reset and done.any():
        # auto_reset_wrapper.py:60-83: key,_ = split(state.key); reset(key); keep reward etc.
        idx = np.nonzero(done)[0]
        new_keys = prng.split(core["key"][idx], 2)[:, 0, :]
        fresh = _core_reset(spec, new_keys)
        core = {k: v.copy() for k, v in core.items()}
        for k in ("step_count", "target", "record", "key"):
            core[k][idx] = fresh[k]
        obs_target = obs_target.copy()
        obs_step = obs_step.copy()
        obs_target[idx] = fresh["target"][:, 0]
        obs_step[idx] = 0
    rewards = np.repeat(reward[:, None], a, axis=1)
    discount = np.repeat(np.where(done, 0.0, 1.0).astype(np.float32)[:, None], a, axis=1)
    # episode_metrics.py:79-112
    not_done = (~done).astype(np.float32)
    new_ret = state["running_return"] + rewards.mean(axis=1, dtype=np.float32)
    new_len = state["running_length"] + 1
    ep_ret = (state["episode_return"] * not_done + new_ret * done).astype(np.float32)
    ep_len = np.where(done, new_len, state["episode_length"]).astype(np.int32)
    new_state = dict(
        core,
        metrics_key=state["metrics_key"],
        running_return=(new_ret * not_done).astype(np.float32),
        running_length=np.where(done, 0, new_len).astype(np.int32),
        episode_return=ep_ret,
        episode_length=ep_len,
    )
    timestep = dict(
        step_type=np.where(done, STEP_LAST, STEP_MID).astype(np.int8),
        reward=rewards,
        discount=discount,
        observation=make_obs(spec, obs_target, obs_step),
        episode_metrics=dict(episode_return=ep_ret, episode_length=ep_len, is_terminal_step=done.copy()),
    )
    return new_state, timestep
