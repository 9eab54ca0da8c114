"""Anakin evaluator (mava/evaluator.py:66-208) on the device kernels: sampled (or greedy) GRU-actor episodes on
non-auto-reset envs for time_limit + 1 steps, metrics taken at the first done of every env.

``pi.sample(seed=key)`` goes through TFP's categorical sampler in the reference; here it is the same
gumbel-argmax kernel as the rollout with the gumbel tensor laid out row-major over (env, agent, action)
-- PARITY UNPINNED against TFP's internal layout."""
from __future__ import annotations

import math
import time
import warnings
from typing import Callable, Dict

import numpy as np
import torch

from ._lib import lib
from .actor import GruActor
from .learner import host_split


def get_num_eval_envs(config, absolute_metric: bool, n_devices: int = 1) -> int:
    n_parallel = config.arch.num_envs * n_devices
    episodes = config.arch.num_absolute_metric_eval_episodes if absolute_metric else config.arch.num_eval_episodes
    if episodes <= n_parallel:
        return math.ceil(episodes / n_devices)
    return config.arch.num_envs


def make_rec_eval_act_fn(actor_apply_fn: GruActor, config) -> Callable:
    """Makes ``EvalActFn(params, timestep, key, actor_state) -> (action, actor_state)`` for a recurrent actor (evaluator.py:50-63,
    188-208).  ``actor_apply_fn`` is the object that owns the actor's kernels (the reference passes ``actor_network.apply``);
    ``timestep`` is the env's TimeStep (observation.agents_view [N, A, F], observation.action_mask [N, A, K], ``last()`` [N]),
    ``key`` one PRNG key ([2] uint32), ``actor_state`` = {"hidden_state": [N * A, 128]}."""
    actor = actor_apply_fn
    greedy = bool(config.arch.evaluation_greedy)
    L = lib()
    loaded = {"params": None}
    _hidden_state = "hidden_state"

    def eval_act_fn(params: Dict[str, torch.Tensor], timestep, key: np.ndarray, actor_state):
        if params is not None and params is not actor.named and params is not loaded["params"]:
            actor.load_named(params)
            loaded["params"] = params
        view, mask = timestep.observation.agents_view, timestep.observation.action_mask
        N, A = view.shape[0], view.shape[1]
        if view.stride(2) != 1 or view.stride(1) != actor.Fld or view.stride(0) != A * actor.Fld:
            raise ValueError(f"agents_view rows must be {actor.Fld} floats apart (got strides {tuple(view.stride())})")
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        last_done = timestep.last().to(torch.uint8)          # repeated over the agents inside the kernel (evaluator.py:200)
        h_in = actor_state[_hidden_state]
        h_out = actor_state.get("_spare")
        if h_out is None or h_out.shape != h_in.shape or h_out.data_ptr() == h_in.data_ptr():
            h_out = torch.empty_like(h_in)
        logits = actor.step(view, h_in, last_done, h_out, want_logits=True)
        action = torch.empty(N, A, dtype=torch.int32, device=view.device)
        if greedy:   # pi.mode() of the masked categorical (heads.py:56-63: illegal logits -> finfo.min)
            lg = logits[:, :actor.K]
            if mask is not None:
                lg = torch.where(mask.view(N * A, actor.K) != 0, lg, torch.full_like(lg, torch.finfo(torch.float32).min))
            action.copy_(lg.argmax(-1).view(N, A))
        else:
            key = np.asarray(key, dtype=np.uint32).reshape(2)
            logp = torch.empty(N * A, device=view.device)
            L.call("magpo_sample_categorical", logits, 64, mask, 0 if mask is None else actor.K, int(key[0]), int(key[1]), None, action, 1,
                   logp, 1, None, 0, None, 0, N * A, actor.K, torch.cuda.current_stream().cuda_stream)
        return action, {_hidden_state: h_out, "_spare": h_in}

    return eval_act_fn


def get_eval_fn(env, act_fn: Callable, config, absolute_metric: bool, device=None, n_devices: int = 1):
    """``EvalFn(params, key, init_act_state) -> metrics`` (evaluator.py:66-185) over the MarlEnv contract: ``env.reset(keys)``,
    then ``time_limit + 1`` times ``act_fn(params, timestep, act_key, actor_state)`` and ``env.step(env_state, action)``; the
    metrics are those of every env's FIRST terminal step.  ``n_devices`` = number of ranks (the reference's jax.device_count())."""
    episodes = config.arch.num_absolute_metric_eval_episodes if absolute_metric else config.arch.num_eval_episodes
    n_envs = get_num_eval_envs(config, absolute_metric, n_devices)
    n_parallel = n_envs * n_devices
    loops = math.ceil(episodes / n_parallel)
    if episodes % n_parallel:
        warnings.warn(f"Number of evaluation episodes ({episodes}) is not divisible by num_envs * num_devices "
                      f"({n_parallel}); running {loops * n_parallel} episodes.", stacklevel=2)
    if device is not None and getattr(env, "device", None) is None:
        env.device = device
    L = lib()

    def eval_fn(params, key: np.ndarray, init_act_state) -> Dict[str, np.ndarray]:
        rets, lens = [], []
        for _ in range(loops):   # _episode (evaluator.py:125-148)
            ks = host_split(key, 2)
            key, reset_key = ks[0], ks[1]
            dev = env._dev()
            kd = torch.from_numpy(reset_key.view(np.int32).copy()).to(dev)
            reset_keys = torch.empty(n_envs, 2, dtype=torch.int32, device=dev)
            L.call("magpo_threefry_split", kd, reset_keys, n_envs, torch.cuda.current_stream().cuda_stream)
            env_state, ts = env.reset(reset_keys)
            actor_state = {"hidden_state": init_act_state["hidden_state"].clone()}
            got = torch.zeros(n_envs, dtype=torch.bool, device=dev)
            ep_ret = torch.zeros(n_envs, device=dev)
            ep_len = torch.zeros(n_envs, dtype=torch.int32, device=dev)
            step_key = key   # the scan's carried key is discarded by _episode (evaluator.py:140,150): loop 2 continues from ``key``
            for _t in range(env.time_limit + 1):   # _env_step (evaluator.py:113-123)
                ks = host_split(step_key, 2)
                step_key, act_key = ks[0], ks[1]
                action, actor_state = act_fn(params, ts, act_key, actor_state)
                env_state, ts = env.step(env_state, action)
                m = ts.extras["episode_metrics"]
                first = ts.last() & ~got
                ep_ret = torch.where(first, m["episode_return"], ep_ret)
                ep_len = torch.where(first, m["episode_length"], ep_len)
                got |= first
            rets.append(ep_ret.cpu().numpy())
            lens.append(ep_len.cpu().numpy())
        return {"episode_return": np.concatenate(rets), "episode_length": np.concatenate(lens)}

    def timed_eval_fn(params, key, init_act_state):
        t0 = time.time()
        metrics = eval_fn(params, key, init_act_state)
        torch.cuda.synchronize()
        metrics["steps_per_second"] = float(np.sum(metrics["episode_length"])) / (time.time() - t0)
        return metrics

    return timed_eval_fn
