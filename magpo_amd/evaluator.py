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
from .learner import host_split, make_env_batch, obs_row_stride


def get_num_eval_envs(config, absolute_metric: bool, n_devices: int = 1) -> int:
    n_parallel = config.arch.num_envs * n_devices
    episodes = config.arch.num_absolute_metric_eval_episodes if absolute_metric else config.arch.num_eval_episodes
    if episodes <= n_parallel:
        return math.ceil(episodes / n_devices)
    return config.arch.num_envs


def make_rec_eval_act_fn(actor: GruActor, config) -> Callable:
    """EvalActFn(params, obs, last_done, key, actor_state) -> (action, actor_state) (evaluator.py:188-208)."""
    greedy = bool(config.arch.evaluation_greedy)
    L = lib()
    loaded = {"params": None}

    def eval_act_fn(params: Dict[str, torch.Tensor], obs: torch.Tensor, last_done: torch.Tensor, key: np.ndarray, actor_state, mask=None):
        if params is not None and params is not actor.named and params is not loaded["params"]:
            actor.load_named(params)
            loaded["params"] = params
        h_in = actor_state["hidden_state"]
        h_out = actor_state.get("_spare")
        if h_out is None or h_out.shape != h_in.shape:
            h_out = torch.empty_like(h_in)
        logits = actor.step(obs, h_in, last_done, h_out, want_logits=True)
        N, A = obs.shape[0], obs.shape[1]
        action = torch.empty(N, A, dtype=torch.int32, device=obs.device)
        if greedy:   # pi.mode() of the masked categorical (heads.py:56-63: illegal logits -> finfo.min)
            lg = logits[:, :actor.K]
            if mask is not None:
                lg = torch.where(mask.view(N * A, actor.K) != 0, lg, torch.full_like(lg, torch.finfo(torch.float32).min))
            action.copy_(lg.argmax(-1).view(N, A))
        else:
            logp = torch.empty(N * A, device=obs.device)
            L.call("magpo_sample_categorical", logits, 64, mask, 0 if mask is None else actor.K, int(key[0]), int(key[1]), None, action, 1,
                   logp, 1, None, 0, None, 0, N * A, actor.K, torch.cuda.current_stream().cuda_stream)
        return action, {"hidden_state": h_out, "_spare": h_in}

    return eval_act_fn


def get_eval_fn(eval_env, act_fn: Callable, config, absolute_metric: bool, device="cuda", n_devices: int = 1):
    episodes = config.arch.num_absolute_metric_eval_episodes if absolute_metric else config.arch.num_eval_episodes
    n_envs = get_num_eval_envs(config, absolute_metric, n_devices)
    n_parallel = n_envs * n_devices
    loops = math.ceil(episodes / n_parallel)
    if episodes % n_parallel:
        warnings.warn(f"Number of evaluation episodes ({episodes}) is not divisible by num_envs * num_devices "
                      f"({n_parallel}); running {loops * n_parallel} episodes.", stacklevel=2)
    cfg = eval_env.cfg
    A, TL = cfg.num_agents, cfg.time_limit
    env = make_env_batch(cfg, n_envs, device)
    f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
    i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
    obs, obs_step = f32(n_envs, A, obs_row_stride(cfg.obs_dim)), i32(n_envs)
    mask = torch.zeros(n_envs, A, cfg.num_actions, dtype=torch.uint8, device=device) if cfg.has_mask else None
    reward, done = f32(n_envs, A), torch.zeros(n_envs, dtype=torch.uint8, device=device)
    m_ret, m_len, m_term = f32(n_envs), i32(n_envs), torch.zeros(n_envs, dtype=torch.uint8, device=device)
    L = lib()

    def eval_fn(params, key: np.ndarray, init_act_state) -> Dict[str, np.ndarray]:
        rets, lens = [], []
        for _ in range(loops):
            ks = host_split(key, 2)
            key, reset_key = ks[0], ks[1]
            kd = torch.from_numpy(reset_key.view(np.int32).copy()).to(device)
            rk = torch.empty(n_envs, 2, dtype=torch.int32, device=device)
            L.call("magpo_threefry_split", kd, rk, n_envs, torch.cuda.current_stream().cuda_stream)
            env.reset(rk, obs, obs_step, mask)
            done.zero_()
            state = {"hidden_state": init_act_state["hidden_state"].clone()}
            got = torch.zeros(n_envs, dtype=torch.bool, device=device)
            ep_ret, ep_len = f32(n_envs), i32(n_envs)
            step_key = key   # the scan's carried key is discarded by _episode (evaluator.py:140,150): loop 2 continues from ``key``
            for _t in range(TL + 1):
                ks = host_split(step_key, 2)
                step_key, act_key = ks[0], ks[1]
                action, state = act_fn(params, obs, done, act_key, state) if mask is None else act_fn(params, obs, done, act_key, state, mask)
                env.step(action, reward, done, obs, obs_step, m_ret, m_len, m_term, auto_reset=False, mask=mask)
                first = done.bool() & ~got
                ep_ret = torch.where(first, m_ret, ep_ret)
                ep_len = torch.where(first, m_len, ep_len)
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
