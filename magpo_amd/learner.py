"""MAGPO Anakin learner on the MI355X kernels: rollout -> GAE -> epochs x minibatches -> clip+Adam.

Host-side mirror of ``get_learner_fn`` (mava/systems/gpo/anakin/rec_magpo.py:91-530).  One instance is
one *group* (the reference's (device, update-batch) replica of ``num_envs`` envs); groups are
independent except for the mean of the gradients (rec_magpo.py:395-409), which the caller injects
through ``grad_sync`` (an RCCL all-reduce over xGMI in the multi-GPU launcher).

Data layout in HBM (no physical shuffle of activations; the minibatch gather moves only per-token
scalars, rec_magpo.py:441-462 becomes index arithmetic):
  trajectory  obs[T+1,N,A,F] f32, step_count[T+1,N] i32, done[T+1,N] u8 (slot t = "obs at step t starts an
              episode"), action/value/log_prob/reward/adv/targets [T,N,A]
  states      3 x [N,64,64] fp32 retention states, policy hidden [N*A,128]
  minibatch   rows (j, t, a') sequence-major, R = mb*T*A
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from ._lib import lib
from .actor import GruActor
from .sable import SableGuider


@dataclass
class SystemConfig:
    rollout_length: int = 128
    ppo_epochs: int = 4
    num_minibatches: int = 2
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_eps: float = 0.2
    ent_coef: float = 0.01
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    clip_gpo: float = 1.5
    alpha: float = 1.0
    actor_lr: float = 2.5e-4
    # make_learning_rate (mava/utils/training.py:20-64): linear decay lr * (1 - (count // (ppo_epochs * num_minibatches)) / num_updates)
    # with the optimiser step count BEFORE the step; lr_num_updates is config.system.num_updates as the schedule reads it when the learner
    # is TRACED (first learn() call), i.e. the value check_total_timesteps derived: the system file refreshes it on every learn() call
    decay_learning_rates: bool = False
    lr_num_updates: int = 1000
    # not a reference key: every minibatch is trained in this many equal slabs of sequences whose gradients are accumulated before the
    # ONE optimiser step (same gradient up to fp32 summation order; advantage statistics stay those of the whole minibatch).
    # Activations in HBM scale with the slab, so large teams run at the reference's num_minibatches within the memory of one GPU.
    micro_batches: int = 1


@dataclass
class CoordSumConfig:
    num_agents: int
    num_actions: int
    time_limit: int = 100
    maxval: Optional[int] = None
    add_agent_id: bool = True  # system.add_agent_id (AgentIDWrapper, make_env.py:90-104): False = the networks read the rows behind the one-hot id
    has_mask = False          # action_mask is all-True (matrax.py:117-134): never stored
    class_tables = True       # observations take few distinct values: first-layer class tables apply (csrc/classtab.hip)

    def __post_init__(self):
        if not self.maxval:
            self.maxval = self.num_actions  # coordsum/env.py:49-53

    @property
    def obs_dim(self) -> int:   # AgentIDWrapper (observation.py:42-54): [one-hot id | target]
        return self.num_agents + 1


@dataclass
class LbfConfig:
    """jumanji LevelBasedForaging-v0 + RandomGenerator(**task_config) (configs/env/scenario/*-coop.yaml) under LbfWrapper."""
    grid_size: int = 8
    fov: int = 8
    num_agents: int = 2
    num_food: int = 2
    max_agent_level: int = 2
    force_coop: bool = True
    time_limit: int = 100
    add_agent_id: bool = True
    has_mask = True
    class_tables = False
    num_actions = 6

    @property
    def obs_dim(self) -> int:   # vector observation 3 (num_food + num_agents) + one-hot agent id
        return 3 * (self.num_food + self.num_agents) + self.num_agents


def net_obs(cfg) -> Tuple[int, int]:
    """(features the networks read, column offset of the first one inside an observation row).  The env kernels always write
    [one-hot agent id | features] rows (AgentIDWrapper, observation.py:42-54); with ``system.add_agent_id: False`` (make_env.py:90-104: the
    wrapper is not applied) the networks are built for the features alone and every consumer gets the row pointer advanced by num_agents
    floats with the row stride unchanged.  Narrow observations only (the 128-float padded rows of wide observations are read with
    16-byte vector loads that a column offset would misalign)."""
    if getattr(cfg, "add_agent_id", True):
        return cfg.obs_dim, 0
    if cfg.obs_dim > 32:
        raise NotImplementedError("system.add_agent_id=False with wide observations (obs_dim > 32: Robot Warehouse)")
    return cfg.obs_dim - cfg.num_agents, cfg.num_agents


def host_split(key: np.ndarray, num: int = 2) -> np.ndarray:
    """jax.random.split of one key on the host (exact; C ABI magpo_key_split_host)."""
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty((num, 2), np.uint32)
    lib().raw("magpo_key_split_host")(key.ctypes.data, num, out.ctypes.data)
    return out


def prng_key(seed: int) -> np.ndarray:
    return np.array([(int(seed) >> 32) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFF], np.uint32)


class CoordSumEnvBatch:
    """Device-resident batch of wrapped CoordSum envs (state surface of coordsum/env.py:17-26 plus the
    RecordEpisodeMetrics counters, episode_metrics.py:35-48)."""

    def __init__(self, cfg: CoordSumConfig, N: int, device):
        self.cfg, self.N, self.dev = cfg, N, device
        A, K, TL = cfg.num_agents, cfg.num_actions, cfg.time_limit
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        self.step_count, self.target, self.record = i32(N), i32(N, TL + 1), i32(N, K, TL)
        self.key, self.metrics_key = i32(N, 2), i32(N, 2)
        self.run_ret, self.run_len = torch.zeros(N, device=device), i32(N)
        self.ep_ret, self.ep_len = torch.zeros(N, device=device), i32(N)
        self.L = lib()

    def _state(self):
        return (self.step_count, self.target, self.record, self.key, self.metrics_key, self.run_ret, self.run_len, self.ep_ret, self.ep_len)

    def _cfg(self):
        c = self.cfg
        return (self.N, c.num_agents, c.num_actions, c.time_limit, c.maxval)

    state_fields = ("step_count", "target", "record", "key", "metrics_key", "run_ret", "run_len", "ep_ret", "ep_len")

    def reset(self, env_keys: torch.Tensor, obs, obs_step, mask=None):
        self.L.call("magpo_coordsum_reset", *self._state(), *self._cfg(), env_keys, obs, obs_step, torch.cuda.current_stream().cuda_stream)

    def step(self, actions, reward, done, obs, obs_step, m_ret, m_len, m_term, auto_reset=True, mask=None, discount=None):
        self.L.call("magpo_coordsum_step", *self._state(), *self._cfg(), actions, self.cfg.num_agents, reward, discount, done, obs, obs_step,
                    m_ret, m_len, m_term, 1 if auto_reset else 0, torch.cuda.current_stream().cuda_stream)


class LbfEnvBatch:
    """Device-resident batch of wrapped Level-Based Foraging envs (csrc/lbf.hip; UNPINNED dynamics, see oracle/lbf.py)."""
    state_fields = ("agent_pos", "agent_level", "food_pos", "food_level", "food_eaten", "step_count", "key", "metrics_key",
                    "run_ret", "run_len", "ep_ret", "ep_len")

    def __init__(self, cfg: LbfConfig, N: int, device):
        self.cfg, self.N, self.dev = cfg, N, device
        A, NF = cfg.num_agents, cfg.num_food
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        self.agent_pos, self.agent_level, self.food_pos, self.food_level = i32(N, A, 2), i32(N, A), i32(N, NF, 2), i32(N, NF)
        self.food_eaten = torch.zeros(N, NF, dtype=torch.uint8, device=device)
        self.step_count, self.key, self.metrics_key = i32(N), i32(N, 2), i32(N, 2)
        self.run_ret, self.run_len = torch.zeros(N, device=device), i32(N)
        self.ep_ret, self.ep_len = torch.zeros(N, device=device), i32(N)
        self.L = lib()

    def _args(self):
        c = self.cfg
        return (self.agent_pos, self.agent_level, self.food_pos, self.food_level, self.food_eaten, self.step_count, self.key, self.metrics_key,
                self.run_ret, self.run_len, self.ep_ret, self.ep_len, self.N, c.num_agents, c.num_food, c.grid_size, c.fov, c.max_agent_level,
                1 if c.force_coop else 0, c.time_limit)

    def reset(self, env_keys: torch.Tensor, obs, obs_step, mask=None):
        self.L.call("magpo_lbf_reset", *self._args(), env_keys, obs, obs_step, mask, torch.cuda.current_stream().cuda_stream)

    def step(self, actions, reward, done, obs, obs_step, m_ret, m_len, m_term, auto_reset=True, mask=None, discount=None):
        self.L.call("magpo_lbf_step", *self._args(), actions, self.cfg.num_agents, reward, discount, done, obs, obs_step, mask, m_ret, m_len, m_term,
                    1 if auto_reset else 0, torch.cuda.current_stream().cuda_stream)


@dataclass
class RwareConfig:
    """jumanji RobotWarehouse-v0 + RandomGenerator(**task_config) (configs/env/scenario/tiny-4ag.yaml ...) under RwareWrapper."""
    column_height: int = 8
    shelf_rows: int = 1
    shelf_columns: int = 3
    num_agents: int = 4
    sensor_range: int = 1
    request_queue_size: int = 4
    time_limit: int = 500
    has_mask = True
    class_tables = False
    num_actions = 5

    @property
    def obs_dim(self) -> int:   # 8 + 7 (2 r + 1)^2 vector observation + one-hot agent id
        return 8 + 7 * (2 * self.sensor_range + 1) ** 2 + self.num_agents


class RwareEnvBatch:
    """Device-resident batch of wrapped Robot Warehouse envs (csrc/rware.hip; UNPINNED dynamics, see oracle/rware.py)."""
    state_fields = ("grid_a", "grid_s", "agent_pos", "agent_dir", "agent_carry", "shelf_req", "queue", "step_count", "amask", "key",
                    "metrics_key", "run_ret", "run_len", "ep_ret", "ep_len")

    def __init__(self, cfg: RwareConfig, N: int, device):
        self.cfg, self.N, self.dev = cfg, N, device
        self.L = lib()
        lay = np.zeros(3, np.int32)
        self.L.call("magpo_rware_layout", cfg.column_height, cfg.shelf_rows, cfg.shelf_columns, lay.ctypes.data)
        self.H, self.W, self.NS = int(lay[0]), int(lay[1]), int(lay[2])
        A = cfg.num_agents
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        u8 = lambda *s: torch.zeros(*s, dtype=torch.uint8, device=device)
        self.grid_a, self.grid_s = i32(N, self.H, self.W), i32(N, self.H, self.W)
        self.agent_pos, self.agent_dir, self.agent_carry = i32(N, A, 2), i32(N, A), u8(N, A)
        self.shelf_req, self.queue = u8(N, self.NS), i32(N, cfg.request_queue_size)
        self.step_count, self.amask, self.key, self.metrics_key = i32(N), u8(N, A, 5), i32(N, 2), i32(N, 2)
        self.run_ret, self.run_len = torch.zeros(N, device=device), i32(N)
        self.ep_ret, self.ep_len = torch.zeros(N, device=device), i32(N)
        self.ldo = obs_row_stride(cfg.obs_dim)

    def _args(self):
        c = self.cfg
        return (self.grid_a, self.grid_s, self.agent_pos, self.agent_dir, self.agent_carry, self.shelf_req, self.queue, self.step_count, self.amask,
                self.key, self.metrics_key, self.run_ret, self.run_len, self.ep_ret, self.ep_len, self.N, c.num_agents, c.column_height,
                c.shelf_rows, c.shelf_columns, c.sensor_range, c.request_queue_size, c.time_limit)

    def reset(self, env_keys: torch.Tensor, obs, obs_step, mask=None):
        self.L.call("magpo_rware_reset", *self._args(), env_keys, obs, self.ldo, obs_step, mask, torch.cuda.current_stream().cuda_stream)

    def step(self, actions, reward, done, obs, obs_step, m_ret, m_len, m_term, auto_reset=True, mask=None, discount=None):
        self.L.call("magpo_rware_step", *self._args(), actions, self.cfg.num_agents, reward, discount, done, obs, self.ldo, obs_step, mask, m_ret, m_len,
                    m_term, 1 if auto_reset else 0, torch.cuda.current_stream().cuda_stream)


def obs_row_stride(obs_dim: int) -> int:
    """Floats between observation rows: obs_dim for small observations, 128 (zero-padded) for wide ones (csrc/wideobs.hip)."""
    return obs_dim if obs_dim <= 32 else 128


def make_env_batch(cfg, N: int, device):
    if isinstance(cfg, RwareConfig):
        return RwareEnvBatch(cfg, N, device)
    return LbfEnvBatch(cfg, N, device) if isinstance(cfg, LbfConfig) else CoordSumEnvBatch(cfg, N, device)


class EnvGroup:
    """Per-group rollout state: envs, trajectory, retention / GRU states, PRNG key.  A group is the
    reference's (device, update-batch) replica (rec_magpo.py:519, :648-653); all groups of a process share
    the parameters and the training workspaces."""

    def __init__(self, env_cfg, N: int, T: int, device, n_block: int = 1, n_tile: int = 1):
        A, F = env_cfg.num_agents, obs_row_stride(env_cfg.obs_dim)
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        u8 = lambda *s: torch.zeros(*s, dtype=torch.uint8, device=device)
        self.env = make_env_batch(env_cfg, N, device)
        self.traj = dict(obs=f32(T + 1, N, A, F), step_count=i32(T + 1, N), done=u8(T + 1, N), action=i32(T, N, A), value=f32(T, N, A),
                         reward=f32(T, N, A), log_prob=f32(T, N, A), adv=f32(T, N, A), targets=f32(T, N, A))
        # action masks (Observation.action_mask) only for envs that have illegal actions; None = every action legal
        self.traj["mask"] = u8(T + 1, N, A, env_cfg.num_actions) if env_cfg.has_mask else None
        self.metrics = dict(episode_return=f32(T, N), episode_length=i32(T, N), is_terminal_step=u8(T, N))
        # (encoder, decoder self, decoder cross) retention states as 64 x 64 tiles: one (zero-padded) tile per head, or the four blocks of
        # the one 128-wide head (SableGuider.ntile)
        self.sable_hs = tuple(f32(n_block, n_tile, N, 64, 64) for _ in range(3))
        self.prev_sable_hs = tuple(f32(n_block, n_tile, N, 64, 64) for _ in range(3))
        self.policy_h = [f32(N * A, 128), f32(N * A, 128)]
        self.policy_h0 = f32(N * A, 128)
        self.last_val = f32(N, A)
        self.key = prng_key(0)
        self.cur = 0
        self.skeys_host = np.zeros((T, A, 2), np.uint32)
        self.skeys_dev = torch.zeros(T, A, 2, dtype=torch.int32, device=device)
        self.graph = None
        self.graph_failed = False


class MagpoLearner:
    def __init__(self, env_cfg, num_envs: int, sys: SystemConfig, device, *, net_seed: Optional[int] = 0,
                 decay_scaling_factor: float = 0.8, use_pe: bool = True, wgrad_groups: int = 512, num_groups: int = 1,
                 n_block: int = 1, n_head: int = 1, embed_dim: int = 64, tuning=None, guider: Optional[SableGuider] = None,
                 actor: Optional[GruActor] = None, optims=None, apply_fns=None, update_fns=None):
        """``guider`` / ``actor`` / ``optims`` = (guider ClipAdam, actor ClipAdam): networks and optimisers built by the caller
        (rec_magpo.learner_setup hands them to get_learner_fn as its apply / update functions); by default the learner builds its own.
        ``apply_fns`` = (sable_action_select_fn, sable_apply_fn, actor_apply_fn), ``update_fns`` = (sable_update_fn, actor_update_fn)
        (rec_magpo.py:99-100): the callables the loop CALLS for acting, the two training forwards and the two optimiser steps --
        by default the bound methods of the objects above; get_learner_fn passes on what it was given (thin adaptors included)."""
        from .tuning import Tuning
        self.tuning = tuning if tuning is not None else (guider.tuning if guider is not None else Tuning.from_env())   # ONE object shared by both networks (tuning.py)
        self.env_cfg, self.N, self.sys, self.dev = env_cfg, num_envs, sys, device
        A, K = env_cfg.num_agents, env_cfg.num_actions
        F, self.obs_off = net_obs(env_cfg)   # what the networks read: the whole row with the AgentIDWrapper's one-hot id, or the part behind it
        self.A, self.K, self.F, self.T = A, K, F, sys.rollout_length
        self.Fld = obs_row_stride(env_cfg.obs_dim)   # floats between rows as the env kernels write them
        if num_envs % sys.num_minibatches:
            raise ValueError("num_envs must be divisible by num_minibatches")
        self.L = lib()
        # one contiguous buffer [guider grads | actor grads | loss scalars] = one all-reduce message (rec_magpo.py:395-409)
        from .optim import ClipAdam
        from .params import FlatParams, actor_layout, guider_layout
        if guider is not None:
            n_block, n_head, embed_dim = guider.nb, guider.nh, guider.EL
        self.nb, self.nh = int(n_block), int(n_head)
        gn = FlatParams(guider_layout(int(embed_dim), F, K, self.nb, self.nh), "cpu").numel
        an = FlatParams(actor_layout(F, 128, K), "cpu").numel
        self.grad_all = torch.zeros(gn + an + 16, dtype=torch.float32, device=device)
        self.grad_acc = torch.zeros_like(self.grad_all) if num_groups > 1 else None
        self.grad_mu = torch.zeros_like(self.grad_all) if sys.micro_batches > 1 else None
        if guider is None:
            guider = SableGuider(A, K, F, device, decay_scaling_factor=decay_scaling_factor, use_pe=use_pe,
                                 max_pos=env_cfg.time_limit + 1, wgrad_groups=wgrad_groups, n_block=self.nb, n_head=self.nh, embed_dim=int(embed_dim),
                                 seed=None if net_seed is None else net_seed, grads=self.grad_all[:gn], tuning=self.tuning, obs_ld=self.Fld)
        else:
            guider.bind_grads(self.grad_all[:gn])
        if actor is None:
            actor = GruActor(A, K, F, device, wgrad_groups=wgrad_groups, seed=None if net_seed is None else net_seed + 1,
                             grads=self.grad_all[gn:gn + an], tuning=self.tuning, obs_ld=self.Fld)
        else:
            actor.bind_grads(self.grad_all[gn:gn + an])
        if guider.F != F or actor.F != F or guider.Fld != self.Fld or actor.Fld != self.Fld:
            raise ValueError(f"networks built for {guider.F} / {actor.F} observation features with row stride {guider.Fld} / {actor.Fld}, "
                             f"the env provides {F} with row stride {self.Fld}")
        self.guider, self.actor = guider, actor
        self.g_opt, self.a_opt = optims if optims is not None else (ClipAdam(guider, sys), ClipAdam(actor, sys))
        assert self.g_opt.net is guider and self.a_opt.net is actor
        self.sable_action_select_fn, self.sable_apply_fn, self.actor_apply_fn = apply_fns if apply_fns is not None else \
            (guider.get_actions, guider.apply, actor.apply)
        self.sable_update_fn, self.actor_update_fn = update_fns if update_fns is not None else (self.g_opt.update, self.a_opt.update)
        self.loss_out = self.grad_all[gn + an:gn + an + 9]
        self.nt = self.guider.ntile
        self.groups: List[EnvGroup] = [EnvGroup(env_cfg, num_envs, self.T, device, self.nb, self.nt) for _ in range(num_groups)]
        # rollout-start states of all groups in ONE tensor each (group g = envs g*N .. g*N + N - 1), so that the minibatches of all
        # local groups train as one batch of sequences (update(): the groups differ only in their advantage statistics)
        U_, N_ = num_groups, num_envs
        self._prev_hs = tuple(torch.zeros(self.nb, self.nt, U_ * N_, 64, 64, device=device) for _ in range(3))
        self._policy_h0 = torch.zeros(U_ * N_ * A, 128, device=device)
        for gi, g in enumerate(self.groups):
            g.prev_sable_hs = tuple(t[:, :, gi * N_:(gi + 1) * N_] for t in self._prev_hs)
            g.policy_h0 = self._policy_h0[gi * N_ * A:(gi + 1) * N_ * A]
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        # optimiser state (optax adam: count, mu, nu) lives in the two ClipAdam objects: g_mu / g_nu / g_count ... below are views of it
        self.ws64 = torch.zeros(8 * 1024, dtype=torch.float64, device=device)
        self.gnorm = f32(2)
        self.adv_stats = f32(2)
        self._mb: Dict[str, torch.Tensor] = {}
        # First-layer class tables (csrc/classtab.hip): a wrapped CoordSum token is one of A*maxval*npos distinct inputs, so the
        # layers in front of the GRU / of the first retention run on the distinct rows only.  MAGPO_CLASS_TABLES=0 = dense path.
        import os
        # (the tables are read in place by the 64-wide fused kernels: a 128-wide net takes the dense first layers)
        # (... and need the one-hot id in the network input: the class rows are [id | target])
        self.class_tables = env_cfg.class_tables and os.environ.get("MAGPO_CLASS_TABLES", "1") != "0" and int(embed_dim) <= 64 and self.obs_off == 0
        self._cls = None
        # the actor's forward / backward run on a second HIP stream next to the guider's (independent until the loss)
        self.overlap_actor = False  # opt-in (bench.py --overlap): ~3 %, but per-kernel timings then include contention
        self._actor_stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None

    g_mu = property(lambda self: self.g_opt.mu)
    g_nu = property(lambda self: self.g_opt.nu)
    a_mu = property(lambda self: self.a_opt.mu)
    a_nu = property(lambda self: self.a_opt.nu)
    g_count = property(lambda self: self.g_opt.count, lambda self, v: setattr(self.g_opt, "count", int(v)))
    a_count = property(lambda self: self.a_opt.count, lambda self, v: setattr(self.a_opt, "count", int(v)))

    # group-0 shortcuts (single-group callers and the parity tests)
    env = property(lambda self: self.groups[0].env)
    traj = property(lambda self: self.groups[0].traj)
    metrics = property(lambda self: self.groups[0].metrics)
    sable_hs = property(lambda self: self.groups[0].sable_hs)
    policy_h = property(lambda self: self.groups[0].policy_h)
    last_val = property(lambda self: self.groups[0].last_val)
    _cur = property(lambda self: self.groups[0].cur)

    @property
    def key(self):
        return self.groups[0].key

    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def _net_view(self, obs: torch.Tensor) -> torch.Tensor:
        """The part of the observation rows the networks read (net_obs): same rows, same stride, pointer behind the one-hot id."""
        return obs if self.obs_off == 0 else obs[..., self.obs_off:]

    # ------------------------------------------------------------------ setup (rec_magpo.py:642-660)
    def setup(self, key: np.ndarray, n_groups: int = 1, group: int = 0):
        """``n_groups`` = total number of groups in the job (ranks x local groups), ``group`` = global index of
        this process's first group.  Reset keys are rows 1.. of split(key, n_groups*N + 1) laid out row-major
        over (group, env); ONE step key is shared by every group (rec_magpo.py:660-671)."""
        N = self.N
        if group < 0 or group + len(self.groups) > n_groups:   # (a short key table would send the env-reset kernel out of bounds)
            raise ValueError(f"setup: this learner holds {len(self.groups)} env group(s) starting at group {group}, but the job has n_groups={n_groups}")
        total = n_groups * N + 1
        kd = torch.from_numpy(np.ascontiguousarray(key, np.uint32).view(np.int32)).to(self.dev)
        allk = torch.empty(total, 2, dtype=torch.int32, device=self.dev)
        self.L.call("magpo_threefry_split", kd, allk, total, self._st())
        key0 = allk[0].cpu().numpy().view(np.uint32)
        ks = host_split(key0, 2)
        self.setup_key = ks[0]
        for gi, g in enumerate(self.groups):
            env_keys = allk[1 + (group + gi) * N: 1 + (group + gi + 1) * N].contiguous()
            g.env.reset(env_keys, g.traj["obs"][0], g.traj["step_count"][0], None if g.traj["mask"] is None else g.traj["mask"][0])
            g.traj["done"][0].zero_()
            g.key = ks[1].copy()
            for h in g.sable_hs:
                h.zero_()
            g.policy_h[0].zero_()
            g.cur = 0

    # ------------------------------------------------------------------ rollout (rec_magpo.py:126-212)
    overlap_actor_step = False  # (measured slower: the acting kernel already fills every wave slot) actor hidden-state carry on a side stream beside the guider's acting kernel
    batched_actor_carry = True  # actor hidden-state carry as ONE scan over the finished trajectory (not T single steps)
    fused_act = True  # one launch per env step for the whole Sable acting step (csrc/act_fused_kernel.hpp)
    use_graph = True  # replay the whole rollout as one HIP graph (removes ~11K host launches per rollout)
    batch_groups = True  # update_batch_size > 1: the minibatches of all local groups train as one batch of sequences

    def rollout(self):
        if len(self.groups) > 1 and self.use_graph and self.fused_act and self.A <= 8 and self.class_tables and self.batched_actor_carry \
                and all(g.graph is not None for g in self.groups):
            # the groups' rollouts are independent: replay their graphs side by side (at small num_envs a rollout is a latency
            # chain that leaves most of the chip idle)
            main = torch.cuda.current_stream()
            for g in self.groups:
                self._rollout_keys(g)
                self._upload_keys(g)
            for gi, g in enumerate(self.groups):
                st = self._group_stream(gi)
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    g.graph.replay()
            for gi in range(len(self.groups)):
                main.wait_stream(self._group_stream(gi))
            return
        for g in self.groups:
            self._rollout_keys(g)
            if not self.use_graph or g.graph_failed:
                self._rollout_body(g, g.skeys_host)            # eager: keys by value
            elif g.graph is not None:
                self._upload_keys(g)
                g.graph.replay()
            elif not getattr(g, "warmed", False):
                self._rollout_body(g, g.skeys_host)            # first call allocates every workspace eagerly
                g.warmed = True
            else:
                self._upload_keys(g)
                self._capture(g)

    def _group_stream(self, gi: int):
        if not hasattr(self, "_gstreams"):
            self._gstreams = {}
        if gi not in self._gstreams:
            self._gstreams[gi] = torch.cuda.Stream(device=self.dev)
        return self._gstreams[gi]

    def _upload_keys(self, g: EnvGroup):
        # pageable source: the runtime stages the 8 KB immediately, so the host table can be reused right away
        g.skeys_dev.copy_(torch.from_numpy(g.skeys_host.view(np.int32).copy()))

    def _capture(self, g: EnvGroup):
        cur0 = g.cur
        try:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._rollout_body(g, g.skeys_dev)
            g.graph = graph
            graph.replay()
        except Exception as e:  # capture is an optimisation: never let it change results
            import warnings
            warnings.warn(f"HIP graph capture of the rollout failed ({e!r}); running eagerly")
            g.graph, g.graph_failed, g.cur = None, True, cur0
            torch.cuda.synchronize()
            self._rollout_body(g, g.skeys_host)

    def _rollout_keys(self, g: EnvGroup):
        """Host key chain of one rollout (pure function of the carried key): key, policy_key = split(key) per env
        step (rec_magpo.py:135); inside get_actions key, sample_key = split(key) per agent (decode.py:141); one
        more split for the bootstrap value (:202).  The A sample keys per step go to a device table."""
        T, A = self.T, self.A
        tab = g.skeys_host
        key = g.key
        for t in range(T):
            ks = host_split(key, 2)
            key, k = ks[0], ks[1]
            for i in range(A):
                kk = host_split(k, 2)
                k, tab[t, i] = kk[0], kk[1]
        g.key = host_split(key, 2)[0]

    def _rollout_body(self, g: EnvGroup, skeys):
        L, st, T, N, A = self.L, self._st(), self.T, self.N, self.A
        tr = g.traj
        for d, s in zip(g.prev_sable_hs, g.sable_hs):
            d.copy_(s)
        g.policy_h0.copy_(g.policy_h[g.cur])
        main = torch.cuda.current_stream()
        side = self._actor_stream if self.overlap_actor_step else None
        fused = self.fused_act
        if fused:   # fragment-major weight copies of the acting kernel from the current parameters (a node of the captured graph as well)
            self.guider.build_act_weights()
        gtag = str(self.groups.index(g))
        act = (lambda *a, **k: self.sable_action_select_fn(*a, tag=gtag, **k)) if fused else self.guider.act

        def zero_done(done):
            for k in range(self.nb):
                for h in range(self.nt):
                    L.call("magpo_zero_states_where_done", g.sable_hs[0][k][h], g.sable_hs[1][k][h], g.sable_hs[2][k][h], done, N, st)

        for t in range(T):
            obs, pos, done_prev = self._net_view(tr["obs"][t]), tr["step_count"][t], tr["done"][t]
            if not self.batched_actor_carry:
                # the actor's hidden-state carry is a pure function of (obs, done); per step it can run on a side stream
                h_in, h_out = g.policy_h[g.cur], g.policy_h[1 - g.cur]
                if side is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        self.actor.step(obs, h_in, done_prev, h_out)
                else:
                    self.actor.step(obs, h_in, done_prev, h_out)
                g.cur = 1 - g.cur
            mk = None if tr["mask"] is None else tr["mask"][t]
            if fused:   # states of envs whose episode just ended read as zero inside the kernel (rec_magpo.py:164-169); the decoder-state
                # update of a step is deferred to the next launch (each state read + written once per step, csrc/act_fused.hip)
                act(obs, pos, g.sable_hs, skeys[t], tr["action"][t], tr["log_prob"][t], tr["value"][t], done=done_prev, mask=mk,
                    pending=t > 0, flush=False, precand=t > 0, defer=True)
            else:
                act(obs, pos, g.sable_hs, skeys[t], tr["action"][t], tr["log_prob"][t], tr["value"][t], mask=mk)
            g.env.step(tr["action"][t], tr["reward"][t], tr["done"][t + 1], tr["obs"][t + 1], tr["step_count"][t + 1],
                       g.metrics["episode_return"][t], g.metrics["episode_length"][t], g.metrics["is_terminal_step"][t],
                       mask=None if tr["mask"] is None else tr["mask"][t + 1])
            if not fused:
                zero_done(tr["done"][t + 1])
        if side is not None:
            main.wait_stream(side)
        if self.batched_actor_carry:   # one scan over the finished trajectory instead of T single steps (same result)
            ccl = None
            if self.class_tables:   # input side of the GRU on the A*maxval distinct (agent, target) rows
                if getattr(g, "traj_cls", None) is None:
                    g.traj_cls = torch.empty(T * N * A, dtype=torch.int32, device=self.dev)
                L.call("magpo_coordsum_classes", tr["obs"], self.F, None, None, A, self.env_cfg.maxval, 1, g.traj_cls, None, T * N * A, st)
                ccl = (self._class_rows()["obs_act"], g.traj_cls)
            self.actor.carry(self._net_view(tr["obs"][:T]), g.policy_h[g.cur], tr["done"][:T], g.policy_h[1 - g.cur], classes=ccl, tag=gtag)
            g.cur = 1 - g.cur
        if g.cur != 0:  # keep the buffer roles identical from rollout to rollout (static graph arguments)
            g.policy_h[0].copy_(g.policy_h[1])
            g.cur = 0
        if fused:   # bootstrap value (encoder states of just-ended episodes read as zero) + the last step's pending decoder-state update
            act(self._net_view(tr["obs"][T]), tr["step_count"][T], g.sable_hs, None, None, None, g.last_val, value_only=True, done=tr["done"][T], pending=True, flush=True, precand=True)
            zero_done(tr["done"][T])
        else:
            act(self._net_view(tr["obs"][T]), tr["step_count"][T], g.sable_hs, None, None, None, g.last_val, value_only=True)
        L.call("magpo_gae", tr["reward"], tr["value"], tr["done"], g.last_val, tr["done"][T], tr["adv"], tr["targets"], T, N, A,
               self.sys.gamma, self.sys.gae_lambda, st)

    def _carry_over(self):
        """Slot T of the trajectory becomes slot 0 of the next rollout."""
        for g in self.groups:
            tr = g.traj
            tr["obs"][0].copy_(tr["obs"][self.T]); tr["step_count"][0].copy_(tr["step_count"][self.T]); tr["done"][0].copy_(tr["done"][self.T])
            if tr["mask"] is not None:
                tr["mask"][0].copy_(tr["mask"][self.T])

    # ------------------------------------------------------------------ shuffles (jax.random.permutation)
    def _permutation(self, key: np.ndarray, n: int) -> torch.Tensor:
        rounds = int(math.ceil(3 * math.log(max(1, n)) / math.log(2 ** 32 - 1)))
        x = torch.arange(n, dtype=torch.int32, device=self.dev)
        bits = torch.empty(n, dtype=torch.int32, device=self.dev)
        for _ in range(rounds):
            ks = host_split(key, 2)
            key, sub = ks[0], ks[1]
            kd = torch.from_numpy(sub.view(np.int32).copy()).to(self.dev)
            self.L.call("magpo_threefry_random_bits", kd, bits, n, self._st())
            order = torch.sort(bits.to(torch.int64) & 0xFFFFFFFF, stable=True).indices
            x = x[order]
        return x.contiguous()

    # ------------------------------------------------------------------ one minibatch (rec_magpo.py:217-435)
    def _gather(self, groups: List[int], env_idx: torch.Tensor, agent_perm: torch.Tensor):
        """Minibatch rows (j, t, a') of the listed groups, group after group, in sequence-major order."""
        T, N, A, F, K = self.T, self.N, self.A, self.Fld, self.K   # (observation rows are copied with their padding)
        mb, U = env_idx.numel(), len(groups)
        R1 = mb * T * A
        R = U * R1
        m = self._mb
        if m.get("R") != R:
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
            i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=self.dev)
            m.update(R=R, obs=f32(R, F), action=i32(R), prev=i32(R), pos=i32(R), done=torch.empty(U * mb, T, dtype=torch.uint8, device=self.dev),
                     value=f32(R), logp=f32(R), adv=f32(R), targets=f32(R), h0idx=i32(U * mb * A),
                     dg=f32(R, 64), da=f32(R, 64), dv=f32(R),
                     mask=torch.empty(R, K, dtype=torch.uint8, device=self.dev) if self.env_cfg.has_mask else None)
        for u, gi in enumerate(groups):
            tr = self.groups[gi].traj
            r = slice(u * R1, (u + 1) * R1)
            h0 = m["h0idx"][u * mb * A:(u + 1) * mb * A]
            self.L.call("magpo_gather_minibatch", tr["obs"], tr["action"], tr["step_count"], tr["done"], tr["mask"], tr["value"], tr["log_prob"],
                        tr["adv"], tr["targets"], env_idx, agent_perm, m["obs"][r], m["action"][r], m["prev"][r], m["pos"][r],
                        m["done"][u * mb:(u + 1) * mb], None if m["mask"] is None else m["mask"][r], m["value"][r], m["logp"][r], m["adv"][r], m["targets"][r], h0, T, N, A, F, K, mb,
                        self._st())
            if gi:
                h0.add_(gi * N * A)      # rows of the stacked start states
        return m

    def _class_rows(self):
        """Distinct first-layer inputs of wrapped CoordSum tokens, in class order (built once)."""
        if self._cls is None:
            A, K, mv, npos = self.A, self.K, self.env_cfg.maxval, self.env_cfg.time_limit + 1
            Ce, Cd = A * mv * npos, (K + 1) * npos
            c = dict(Ce=Ce, Cd=Cd, Ca=A * mv, npos=npos, obs_enc=torch.empty(Ce, self.F, device=self.dev),
                     pos_enc=torch.empty(Ce, dtype=torch.int32, device=self.dev), prev_dec=torch.empty(Cd, dtype=torch.int32, device=self.dev),
                     pos_dec=torch.empty(Cd, dtype=torch.int32, device=self.dev))
            self.L.call("magpo_coordsum_class_rows", A, mv, npos, K, c["obs_enc"], c["pos_enc"], c["prev_dec"], c["pos_dec"], self._st())
            c["obs_act"] = c["obs_enc"][::npos].contiguous()     # actor class (agent, target) = encoder class // npos
            c["zero"] = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self._cls = c
        return self._cls

    def _classes(self, m):
        """Class index of every minibatch row and the stable row order per class (one sort per network side; the actor's
        classes are a coarsening of the encoder's, so it shares that order)."""
        c = self._class_rows()
        R = m["R"]
        if m.get("cls_R") != R:
            m.update(cls_R=R, cls_enc=torch.empty(R, dtype=torch.int32, device=self.dev), cls_dec=torch.empty(R, dtype=torch.int32, device=self.dev))
        self.L.call("magpo_coordsum_classes", m["obs"], self.F, m["prev"], m["pos"], self.A, self.env_cfg.maxval, c["npos"],
                    m["cls_enc"], m["cls_dec"], R, self._st())
        out = {}
        for side, C in (("enc", c["Ce"]), ("dec", c["Cd"])):
            cls = m["cls_" + side]
            vals, order = torch.sort(cls, stable=True)
            # class boundaries in the sorted order without a host synchronisation (torch.bincount sizes its output on the host)
            if ("bounds_" + side) not in c:
                c["bounds_" + side] = torch.arange(C + 1, dtype=torch.int32, device=self.dev)
            offsets = torch.searchsorted(vals, c["bounds_" + side])
            out[side] = (cls, order, offsets)
        cls_act = torch.div(m["cls_enc"], c["npos"], rounding_mode="floor").to(torch.int32)
        out["act"] = (c["obs_act"], cls_act, out["enc"][1], out["enc"][2][::c["npos"]].contiguous())
        return out

    def _minibatch_adv_stats(self, group, env_idx: torch.Tensor) -> torch.Tensor:
        """(mean, 1 / (std + eps)) of the advantages of the minibatch's envs per group [U, 2] (rec_magpo.py:283,356): what
        minibatch_grads computes from its gathered rows, here for the WHOLE minibatch ahead of its micro-batches."""
        groups = [group] if isinstance(group, int) else list(group)
        out = torch.zeros(len(groups), 2, device=self.dev)
        for u, gi in enumerate(groups):
            sel = self.groups[gi].traj["adv"][:, env_idx.long(), :].contiguous()
            self.L.call("magpo_adv_moments", sel, sel.numel(), self.ws64, out[u], self._st())
        return out

    def minibatch_grads(self, env_idx: torch.Tensor, agent_perm: torch.Tensor, group=0, hs_idx: Optional[torch.Tensor] = None,
                        adv_stats: Optional[torch.Tensor] = None):
        """Forward + loss + backward of both networks for one minibatch; gradients land in guider.grads / actor.grads, loss
        scalars in self.loss_out (all inside self.grad_all, on device).  ``group``: one group index, or a list of groups that
        train as ONE batch of sequences -- every group uses the same env / agent permutation and the loss is a mean
        over rows, so the batch gradient is the unweighted mean of the groups' gradients (the pmean over the "batch" axis,
        rec_magpo.py:395-397); only the advantage normalisation stays per group (rec_magpo.py:283,356).
        ``hs_idx`` [mb]: env whose rollout-start Sable states sequence j trains on (quirk B19: in the reference
        it differs from ``env_idx`` after the first PPO epoch); default = ``env_idx``.  ``adv_stats`` [U, 2]: advantage statistics to
        use instead of those of the rows at hand (micro-batches: the statistics of the whole minibatch)."""
        s, T, A, K, N = self.sys, self.T, self.A, self.K, self.N
        groups = [group] if isinstance(group, int) else list(group)
        U = len(groups)
        m = self._gather(groups, env_idx, agent_perm)
        mb, R = env_idx.numel(), m["R"]
        nseq, R1 = U * mb, R // U
        hidx = env_idx if hs_idx is None else hs_idx
        if U > 1 or groups[0]:
            hidx = torch.cat([hidx + gi * N for gi in groups])
        cl = self._classes(m) if self.class_tables else None
        acl = None if cl is None else cl["act"]
        gcl = None if cl is None else dict(rows=(self._cls["obs_enc"], self._cls["pos_enc"], self._cls["prev_dec"], self._cls["pos_dec"]),
                                           enc=cl["enc"], dec=cl["dec"])
        side = self._actor_stream if self.overlap_actor else None
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)  # minibatch gather (and the previous optimiser step) are complete for the actor
            with torch.cuda.stream(side):
                a_logits = self.actor_apply_fn(self._net_view(m["obs"]), m["done"], self._policy_h0, m["h0idx"], nseq, T, classes=acl)
        g_logits, value = self.sable_apply_fn(self._net_view(m["obs"]), m["prev"], m["pos"], m["done"], self._prev_hs, hidx, nseq, T, classes=gcl)
        if side is not None:
            main.wait_stream(side)
        else:
            a_logits = self.actor_apply_fn(self._net_view(m["obs"]), m["done"], self._policy_h0, m["h0idx"], nseq, T, classes=acl)
        st = self._st()
        if U == 1:
            if adv_stats is None:
                self.L.call("magpo_adv_moments", m["adv"], R, self.ws64, self.adv_stats, st)
            stats = self.adv_stats if adv_stats is None else adv_stats[0]
        else:   # per-group statistics, applied in place with the loss kernel's own expression (adv - mean) * rstd; identity stats after
            if getattr(self, "_adv_stats_u", None) is None or self._adv_stats_u.shape[0] != U:
                self._adv_stats_u = torch.zeros(U, 2, device=self.dev)
                self._adv_ident = torch.tensor([0.0, 1.0], device=self.dev)
            su = self._adv_stats_u if adv_stats is None else adv_stats
            if adv_stats is None:
                for u in range(U):
                    self.L.call("magpo_adv_moments", m["adv"][u * R1:(u + 1) * R1], R1, self.ws64, su[u], st)
            a2 = m["adv"].view(U, R1)
            a2.sub_(su[:, 0:1]).mul_(su[:, 1:2])
            stats = self._adv_ident
        self.L.call("magpo_loss_fwd_bwd", g_logits, 64, a_logits, 64, m["mask"], m["action"], m["logp"], m["value"], value, m["adv"], m["targets"],
                    stats, m["dg"], 64, m["da"], 64, m["dv"], self.ws64, self.loss_out, R, K, s.clip_eps, s.clip_gpo,
                    s.ent_coef, s.vf_coef, s.alpha, st)
        if side is not None:
            side.wait_stream(main)  # loss gradients are ready
            with torch.cuda.stream(side):
                self.actor.seq_bwd(m["da"])
        self.guider.train_bwd(m["dg"], m["dv"])
        if side is not None:
            main.wait_stream(side)
        else:
            self.actor.seq_bwd(m["da"])

    def apply_grads(self, grad_scale: float = 1.0):
        """optax clip_by_global_norm + adam + apply_updates on both flat buffers (rec_magpo.py:412-420): the two update functions."""
        self.sable_update_fn(grad_scale, self.ws64, self.gnorm[0:1])
        self.last_lr = self.actor_update_fn(grad_scale, self.ws64, self.gnorm[1:2])

    # ------------------------------------------------------------------ update (rec_magpo.py:214-487)
    def update(self, grad_sync: Optional[Callable[["MagpoLearner"], float]] = None) -> torch.Tensor:
        """ppo_epochs x num_minibatches optimisation steps; returns the loss table [P, M, 9] (device), already
        averaged over groups (and ranks when grad_sync all-reduces)."""
        s, N, A = self.sys, self.N, self.A
        M = s.num_minibatches
        mbs = N // M
        U = len(self.groups)
        losses = torch.zeros(s.ppo_epochs, M, 9, device=self.dev)
        # Quirk B19 (rec_magpo.py:437,447,471): the reference shuffles prev_hstates by batch_perm and carries the SHUFFLED
        # arrays into the next epoch, so epoch e reads state row hs_idx_e[i] = hs_idx_{e-1}[batch_perm_e[i]] for
        # sequence i while the trajectory is gathered by batch_perm_e alone.  Only the index is composed; the 48 KiB
        # states never move.
        hs_idx = None
        mu = max(1, int(s.micro_batches))
        if mu > 1 and mbs % mu:
            raise ValueError(f"micro_batches={mu} must divide the minibatch of {mbs} envs")

        def grads(idx, group, hidx):
            """Gradients of one minibatch into grad_all: in one pass, or as the mean over ``micro_batches`` equal slabs."""
            if mu == 1:
                self.minibatch_grads(idx, agent_perm, group, hidx)
                return
            stats = self._minibatch_adv_stats(group, idx)
            step = mbs // mu
            self.grad_mu.zero_()
            for j in range(mu):
                self.minibatch_grads(idx[j * step:(j + 1) * step].contiguous(), agent_perm, group, hidx[j * step:(j + 1) * step].contiguous(),
                                     adv_stats=stats)
                self.grad_mu.add_(self.grad_all)
            self.grad_all.copy_(self.grad_mu).mul_(1.0 / mu)

        for e in range(s.ppo_epochs):
            # every group holds the same key => one permutation serves all groups
            ks = host_split(self.groups[0].key, 4)
            kb, ka, ke = ks[1], ks[2], ks[3]
            for g in self.groups:
                g.key = ks[0].copy()
            batch_perm = self._permutation(kb, N)
            agent_perm = self._permutation(ka, A)
            hs_idx = batch_perm if hs_idx is None else hs_idx[batch_perm.long()].contiguous()
            for mi in range(M):
                ke = host_split(ke, 2)[0]  # key, entropy_key = split(key): unused for discrete actions (:373)
                idx = batch_perm[mi * mbs:(mi + 1) * mbs].contiguous()
                hidx = hs_idx[mi * mbs:(mi + 1) * mbs].contiguous()
                if U == 1:
                    grads(idx, 0, hidx)
                    scale = 1.0
                elif self.batch_groups:   # all local groups as one batch of sequences: the row mean IS the pmean over "batch"
                    grads(idx, list(range(U)), hidx)
                    scale = 1.0
                else:  # group by group: accumulate, the 1/U goes into grad_scale
                    self.grad_acc.zero_()
                    for gi in range(U):
                        grads(idx, gi, hidx)
                        self.grad_acc.add_(self.grad_all)
                    self.grad_all.copy_(self.grad_acc)
                    scale = 1.0 / U
                scale *= grad_sync(self) if grad_sync is not None else 1.0
                self.apply_grads(scale)
                losses[e, mi].copy_(self.loss_out)
                losses[e, mi].mul_(scale)
        return losses

    def update_step(self, grad_sync=None):
        """One ``_update_step`` (rec_magpo.py:106-499): rollout + GAE + training."""
        self.rollout()
        losses = self.update(grad_sync)
        self._carry_over()
        return losses
