"""Anakin evaluator, CPU restatement (oracle; test infrastructure only).

Follows mava/evaluator.py:66-171 (get_num_eval_envs, get_eval_fn: _episode / _env_step) and
make_rec_eval_act_fn :188-208, on the non-auto-reset eval env of mava/utils/make_env.py:100-102
(AgentID -> RecordEpisodeMetrics; SURVEY B15: the scan runs time_limit + 1 steps and steps past
termination, metrics are read at the first ``last()``).

PRNG chain (evaluator.py:128,136-137,141):
  per episode loop   key, reset_key = split(key); reset_keys = split(reset_key, n_envs)
  per env step       step_key, act_key = split(step_key), where the scan carry STARTS from the loop's ``key``
                     but the carried key is discarded: ``_episode`` returns the key taken right after the reset
                     split, so the next loop continues from there (not from the end of the step chain).

PARITY UNPINNED: ``pi.sample(seed=key)`` runs through TFP 0.25's Categorical sampler inside the
IdentityTransformation wrapper (heads.py:63, distributions.py:133-152); restated here as
jax.random.categorical's gumbel-argmax with the gumbel tensor laid out row-major over (env, agent, action)
(from memory of tfp.substrates.jax ``_categorical_jax``: gumbel of shape logits_2d.shape + (n,)).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import coordsum as cs
from . import networks as nets
from . import prng


def get_num_eval_envs(num_envs: int, eval_episodes: int, n_devices: int = 1) -> int:
    """evaluator.py:66-79."""
    if eval_episodes <= num_envs * n_devices:
        return math.ceil(eval_episodes / n_devices)
    return num_envs


def rec_eval_act(ap, timestep, key, hidden, greedy: bool = False):
    """make_rec_eval_act_fn (evaluator.py:188-208): last_done = timestep.last() per agent, leading time dim of 1."""
    ob = timestep["observation"]
    obs = torch.from_numpy(ob["agents_view"])
    mask = torch.from_numpy(ob["action_mask"])
    n_agents = obs.shape[1]
    last = torch.from_numpy(timestep["step_type"] == cs.STEP_LAST)
    last_done = last[:, None].repeat(1, n_agents)
    hidden, logp, _ = nets.actor_apply(ap, hidden, obs[None], last_done[None], mask[None])
    logp = logp[0]
    if greedy:
        action = logp.argmax(-1).to(torch.int32).numpy()
    else:
        action = prng.categorical(key, logp.to(torch.float32).numpy())
    return action, hidden


@torch.no_grad()
def evaluate(spec, ap, key: np.ndarray, num_envs: int, eval_episodes: int, hidden: int = 128,
             greedy: bool = False, dtype=torch.float32, env=cs) -> Dict[str, np.ndarray]:
    """eval_fn (evaluator.py:113-157) on one device: returns flattened per-episode metric arrays."""
    n = get_num_eval_envs(num_envs, eval_episodes)
    loops = math.ceil(eval_episodes / n)
    ap = {k: v.to(dtype) for k, v in ap.items()}
    rets, lens = [], []
    key = np.asarray(key, np.uint32)
    for _ in range(loops):
        ks = prng.split(key, 2)                                   # :136
        key, reset_key = ks[0], ks[1]
        state, ts = env.reset(spec, prng.split(reset_key, n))      # :137-138
        h = torch.zeros(n, spec.num_agents, hidden, dtype=dtype)  # init_act_state (rec_magpo.py:745-748)
        step_key = key                                            # :140 (the carried copy is thrown away by :150)
        last_t, m_ret, m_len = [], [], []
        for _t in range(spec.time_limit + 1):                     # :141
            ks = prng.split(step_key, 2)                          # :128
            step_key, act_key = ks[0], ks[1]
            action, h = rec_eval_act(ap, ts, act_key, h, greedy)
            state, ts = env.step(spec, state, action, auto_reset=False)
            last_t.append(ts["step_type"] == cs.STEP_LAST)
            m_ret.append(ts["episode_metrics"]["episode_return"].copy())
            m_len.append(ts["episode_metrics"]["episode_length"].copy())
        done_idx = np.argmax(np.stack(last_t), axis=0)            # first done (:147)
        ar = np.arange(n)
        rets.append(np.stack(m_ret)[done_idx, ar])
        lens.append(np.stack(m_len)[done_idx, ar])
    return {"episode_return": np.concatenate(rets), "episode_length": np.concatenate(lens)}
