"""MAGPO Anakin learner, CPU restatement (oracle; test infrastructure only).

Follows mava/systems/gpo/anakin/rec_magpo.py:91-530 (get_learner_fn: _env_step, GAE,
_update_epoch, _update_minibatch, _guider_loss_fn, _actor_loss_fn), learner_setup :533-685
(PRNG layout), mava/utils/multistep.py:24-68, mava/utils/jax_utils.py:70-83,
and optax 0.2.4 clip_by_global_norm + adam(eps=1e-5) (rec_magpo.py:581-589).
One "group" (= one (device, update-batch) replica of N envs) is simulated; groups are
independent except for the gradient mean (rec_magpo.py:395-409), which ``update`` exposes
through the ``grad_hook`` argument.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import coordsum as cs
from . import networks as nets
from . import prng


class SystemCfg:
    def __init__(self, **kw):
        d = dict(rollout_length=128, ppo_epochs=4, num_minibatches=2, gamma=0.99, gae_lambda=0.95,
                 clip_eps=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, clip_gpo=1.5, alpha=1.0,
                 actor_lr=2.5e-4, hidden=128, decay_learning_rates=False, lr_num_updates=1000)
        d.update(kw)
        self.__dict__.update(d)


# ----------------------------------------------------------------------------- GAE
def calculate_gae(reward, value, done, last_val, last_done, gamma, lam):
    """multistep.py:24-68. reward/value/done (T,N,A); last_* (N,A)."""
    T = reward.shape[0]
    adv = torch.zeros_like(value)
    gae = torch.zeros_like(last_val)
    next_value, next_done = last_val, last_done.to(value.dtype)
    for t in range(T - 1, -1, -1):
        delta = reward[t] + gamma * next_value * (1 - next_done) - value[t]
        gae = delta + gamma * lam * (1 - next_done) * gae
        adv[t] = gae
        next_value, next_done = value[t], done[t].to(value.dtype)
    return adv, adv + value


# ----------------------------------------------------------------------------- optimiser
def adam_init(params):
    return dict(count=0, mu={k: torch.zeros_like(v) for k, v in params.items()},
                nu={k: torch.zeros_like(v) for k, v in params.items()})


def clip_adam_step(params, grads, opt, lr, max_norm, b1=0.9, b2=0.999, eps=1e-5):
    """optax.chain(clip_by_global_norm(max_norm), adam(lr, eps=1e-5)) + apply_updates."""
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(next(iter(grads.values())).dtype)
    count = opt["count"] + 1
    new_p, mu, nu = {}, {}, {}
    for k, p in params.items():
        g = grads[k]
        if gnorm >= max_norm:
            g = (g / gnorm) * max_norm
        mu[k] = b1 * opt["mu"][k] + (1 - b1) * g
        nu[k] = b2 * opt["nu"][k] + (1 - b2) * g * g
        mu_hat = mu[k] / (1 - b1 ** count)
        nu_hat = nu[k] / (1 - b2 ** count)
        new_p[k] = p + (-lr) * (mu_hat / (torch.sqrt(nu_hat) + eps))
    return new_p, dict(count=count, mu=mu, nu=nu), gnorm


# ----------------------------------------------------------------------------- losses
def _kl(lp1, lp2):
    """distrax _kl_divergence_categorical_categorical: sum p1 (logp1 - logp2), 0 where p1 == 0."""
    p1 = lp1.exp()
    return torch.where(p1 == 0, torch.zeros_like(p1), p1 * (lp1 - lp2)).sum(-1)


def guider_loss(sys: SystemCfg, value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp, mb):
    """_guider_loss_fn rec_magpo.py:222-311 given network outputs (actor side is constant)."""
    ld = math.log(sys.clip_gpo)
    kl = _kl(g_lp_all, a_lp_all.detach())
    a_logp = a_logp.detach()
    ratio = torch.exp(g_logp - mb["log_prob"])
    d = g_logp - a_logp
    clipped_ratio = torch.exp(torch.clamp(d, -ld, ld) + a_logp - mb["log_prob"])
    mask = ((d < -ld) | (d > ld)).to(value.dtype)
    kl_loss = (kl * mask).mean()
    gae = mb["adv"]
    gae = (gae - gae.mean()) / (gae.std(unbiased=False) + 1e-8)
    l1 = ratio * gae
    l2 = torch.clamp(clipped_ratio, 1 - sys.clip_eps, 1 + sys.clip_eps) * gae
    pg = -torch.minimum(l1, l2).mean()
    ent = g_ent.mean()
    vclip = mb["value"] + (value - mb["value"]).clamp(-sys.clip_eps, sys.clip_eps)
    vl = 0.5 * torch.maximum((value - mb["targets"]) ** 2, (vclip - mb["targets"]) ** 2).mean()
    total = pg + kl_loss - sys.ent_coef * ent + sys.vf_coef * vl
    return total, dict(guider_loss=pg, entropy=ent, value_loss=vl, kl_loss=kl_loss)


def actor_loss(sys: SystemCfg, g_lp_all, a_lp_all, a_logp, mb):
    """_actor_loss_fn rec_magpo.py:313-370."""
    kl = _kl(g_lp_all.detach(), a_lp_all).mean()
    ratio = torch.exp(a_logp - mb["log_prob"])
    gae = mb["adv"]
    gae = (gae - gae.mean()) / (gae.std(unbiased=False) + 1e-8)
    l1 = ratio * gae
    l2 = torch.clamp(ratio, 1 - sys.clip_eps, 1 + sys.clip_eps) * gae
    al = -torch.minimum(l1, l2).mean()
    return al * sys.alpha + kl, dict(actor_loss=al, actor_kl=kl)


def minibatch_forward(sys, scfg, gp, ap, mb):
    """Network forwards of one minibatch (shared by both losses; the reference runs the Sable
    forward twice with identical inputs, rec_magpo.py:233,322)."""
    value, g_logp, g_ent, g_lp_all = nets.sable_train(
        gp, scfg, mb["obs"], mb["action"], mb["mask"], mb["step_count"], mb["prev_hs"], mb["done"])
    A = scfg.A
    n, ta = mb["action"].shape
    T = ta // A

    def fwd(x):  # forward_reshape rec_magpo.py:60-75
        return x.reshape(n, T, A, *x.shape[2:]).transpose(0, 1)

    _, a_lp, _ = nets.actor_apply(ap, mb["policy_h0"], fwd(mb["obs"]), fwd(mb["done"]), fwd(mb["mask"]))
    a_lp_all = a_lp.transpose(0, 1).reshape(n, ta, -1)  # backward_reshape :78-88
    a_logp = torch.gather(a_lp_all, -1, mb["action"].long()[..., None])[..., 0]
    return value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp


# ----------------------------------------------------------------------------- learner
class OracleLearner:
    """Single-group learner; state lives in numpy/torch on the CPU."""

    def __init__(self, spec, num_envs: int, sys: SystemCfg, scfg: nets.SableCfg,
                 guider_params, actor_params, dtype=torch.float32, env=cs):
        """``env``: the wrapped-env module (oracle.coordsum or oracle.lbf: reset(spec, keys) / step(spec, state, actions))."""
        self.spec, self.N, self.sys, self.scfg, self.dtype, self.env = spec, num_envs, sys, scfg, dtype, env
        self.gp = {k: v.clone().to(dtype) for k, v in guider_params.items()}
        self.ap = {k: v.clone().to(dtype) for k, v in actor_params.items()}
        self.g_opt = adam_init(self.gp)
        self.a_opt = adam_init(self.ap)

    def setup(self, key: np.ndarray, n_groups: int = 1, group: int = 0):
        """learner_setup PRNG layout (rec_magpo.py:642-660): env keys = split(key, G*N+1)[1:],
        row-major over (group, env); one step key shared by all groups."""
        N = self.N
        ks = prng.split(key, n_groups * N + 1)
        key = ks[0]
        env_keys = ks[1 + group * N: 1 + (group + 1) * N]
        self.env_state, self.timestep = self.env.reset(self.spec, env_keys)
        ks = prng.split(key, 2)
        self.setup_key, self.key = ks[0], ks[1]
        self.dones = np.zeros((N, self.spec.num_agents), bool)
        self.sable_hs = nets.init_sable_hstates(N, self.scfg, self.dtype)
        self.policy_h = torch.zeros(N, self.spec.num_agents, self.sys.hidden, dtype=self.dtype)

    # -- rollout ---------------------------------------------------------------------------
    @torch.no_grad()
    def rollout(self, T: Optional[int] = None, record_logits: bool = False):
        """_env_step x T (rec_magpo.py:126-197) + bootstrap value (:202-208)."""
        sys, scfg, spec = self.sys, self.scfg, self.spec
        T = T or sys.rollout_length
        traj = {k: [] for k in ("done", "action", "value", "reward", "log_prob", "obs", "step_count", "mask")}
        metrics = {k: [] for k in ("episode_return", "episode_length", "is_terminal_step")}
        logits_rec = []
        self.prev_sable_hs = tuple(h.clone() for h in self.sable_hs)
        self.policy_h0 = self.policy_h.clone()
        for _ in range(T):
            ks = prng.split(self.key, 2)
            self.key, policy_key = ks[0], ks[1]
            ob = self.timestep["observation"]
            obs = torch.from_numpy(ob["agents_view"])
            mask = torch.from_numpy(ob["action_mask"])
            sc = torch.from_numpy(ob["step_count"])
            action, logp, value, new_hs, lp_all = nets.sable_get_actions(
                self.gp, scfg, obs, mask, sc, self.sable_hs, policy_key)
            last_done = torch.from_numpy(self.dones)
            self.policy_h, _, _ = nets.actor_apply(self.ap, self.policy_h, obs[None], last_done[None], mask[None])
            prev_done = self.dones.copy()
            self.env_state, self.timestep = self.env.step(spec, self.env_state, action.numpy(), auto_reset=True)
            done = self.timestep["step_type"] == cs.STEP_LAST
            dmask = torch.from_numpy(done)[:, None, None, None, None]
            self.sable_hs = tuple(torch.where(dmask, torch.zeros_like(h), h) for h in new_hs)
            self.dones = np.repeat(done[:, None], spec.num_agents, axis=1)
            traj["done"].append(torch.from_numpy(prev_done))
            traj["action"].append(action)
            traj["value"].append(value)
            traj["reward"].append(torch.from_numpy(self.timestep["reward"]).to(self.dtype))
            traj["log_prob"].append(logp)
            traj["obs"].append(obs)
            traj["step_count"].append(sc)
            traj["mask"].append(mask)
            for k in metrics:
                metrics[k].append(self.timestep["episode_metrics"][k].copy())
            if record_logits:
                logits_rec.append(lp_all)
        ks = prng.split(self.key, 2)
        self.key, last_val_key = ks[0], ks[1]
        ob = self.timestep["observation"]
        _, _, last_val, _, _ = nets.sable_get_actions(
            self.gp, scfg, torch.from_numpy(ob["agents_view"]), torch.from_numpy(ob["action_mask"]),
            torch.from_numpy(ob["step_count"]), self.sable_hs, last_val_key)
        traj = {k: torch.stack(v, dim=0) for k, v in traj.items()}
        adv, targets = calculate_gae(traj["reward"], traj["value"], traj["done"], last_val,
                                     torch.from_numpy(self.dones), sys.gamma, sys.gae_lambda)
        traj["adv"], traj["targets"] = adv, targets
        self.traj = traj
        self.last_val = last_val
        out = {k: np.stack(v, axis=0) for k, v in metrics.items()}
        if record_logits:
            out["logits"] = torch.stack(logits_rec, dim=0)
        return out

    # -- training ---------------------------------------------------------------------------
    def make_minibatches(self, batch_perm: np.ndarray, agent_perm: np.ndarray, prev_hstates=None):
        """rec_magpo.py:441-462: take env axis, take agent axis, concat time & agents, split.

        ``prev_hstates`` is the epoch carry of rec_magpo.py:437: the reference rebinds
        ``prev_hstates = take(prev_hstates, batch_perm)`` (:447) and puts the SHUFFLED arrays back into
        ``update_state`` (:471), so the rollout-start Sable states are permuted cumulatively across the
        PPO epochs while the trajectory is always shuffled from its original order (quirk B19).  The
        shuffled states are left in ``self._epoch_prev_hs`` for the next epoch; None = first epoch."""
        M = self.sys.num_minibatches
        bp = torch.from_numpy(batch_perm.astype(np.int64))
        apm = torch.from_numpy(agent_perm.astype(np.int64))
        tr = self.traj
        N = self.N

        def prep(x):  # (T,N,A,...) -> (M, mb, T*A, ...)
            x = x.index_select(1, bp).index_select(2, apm)
            x = x.transpose(0, 1)
            x = x.reshape(N, x.shape[1] * x.shape[2], *x.shape[3:])
            return x.reshape(M, N // M, *x.shape[1:])

        fields = {k: prep(tr[k]) for k in ("done", "action", "value", "log_prob", "obs", "step_count", "mask", "adv", "targets")}
        carried = self.prev_sable_hs if prev_hstates is None else prev_hstates
        self._epoch_prev_hs = tuple(h.index_select(0, bp) for h in carried)          # :447, carried by :471
        prev = tuple(h.reshape(M, N // M, *h.shape[1:]) for h in self._epoch_prev_hs)
        h0 = self.policy_h0.index_select(0, bp).index_select(1, apm).reshape(M, N // M, *self.policy_h0.shape[1:])
        mbs = []
        for m in range(M):
            mb = {k: v[m] for k, v in fields.items()}
            mb["prev_hs"] = tuple(h[m] for h in prev)
            mb["policy_h0"] = h0[m]
            mbs.append(mb)
        return mbs

    def minibatch_grads(self, mb):
        gp = {k: v.detach().clone().requires_grad_(True) for k, v in self.gp.items()}
        ap = {k: v.detach().clone().requires_grad_(True) for k, v in self.ap.items()}
        value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp = minibatch_forward(self.sys, self.scfg, gp, ap, mb)
        gl, ginfo = guider_loss(self.sys, value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp, mb)
        al, ainfo = actor_loss(self.sys, g_lp_all, a_lp_all, a_logp, mb)
        g_grads = torch.autograd.grad(gl, list(gp.values()), retain_graph=True, allow_unused=True)
        a_grads = torch.autograd.grad(al, list(ap.values()), allow_unused=True)
        gg = {k: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(gp.items(), g_grads)}
        ag = {k: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(ap.items(), a_grads)}
        info = {k: float(v.detach()) for k, v in {**ginfo, **ainfo}.items()}
        info["total_loss"] = float(gl.detach()) + float(al.detach())
        inter = dict(value=value.detach(), g_logp=g_logp.detach(), g_ent=g_ent.detach(),
                     g_lp_all=g_lp_all.detach(), a_lp_all=a_lp_all.detach())
        return gg, ag, info, inter

    def update(self, grad_hook: Optional[Callable] = None):
        """_update_epoch x ppo_epochs (rec_magpo.py:214-487)."""
        sys = self.sys
        infos = []
        prev_hstates = None   # update_state[-1] = prev_sable_hstates (:474-482)
        for _ in range(sys.ppo_epochs):
            ks = prng.split(self.key, 4)
            self.key, kb, ka, ke = ks[0], ks[1], ks[2], ks[3]
            batch_perm = prng.permutation(kb, self.N)
            agent_perm = prng.permutation(ka, self.spec.num_agents)
            mbs = self.make_minibatches(batch_perm, agent_perm, prev_hstates)
            prev_hstates = self._epoch_prev_hs
            for mb in mbs:
                ke = prng.split(ke, 2)[0]  # key, entropy_key = split(key) (:373), unused for discrete
                gg, ag, info, _ = self.minibatch_grads(mb)
                if grad_hook is not None:
                    gg, ag = grad_hook(gg, ag)
                lr_g, lr_a = self._lr(self.g_opt["count"]), self._lr(self.a_opt["count"])
                self.gp, self.g_opt, _ = clip_adam_step(self.gp, gg, self.g_opt, lr_g, sys.max_grad_norm)
                self.ap, self.a_opt, _ = clip_adam_step(self.ap, ag, self.a_opt, lr_a, sys.max_grad_norm)
                infos.append(info)
        return infos

    def _lr(self, count: int) -> float:
        """make_learning_rate (mava/utils/training.py:20-64): constant, or the linear schedule evaluated at the optax step count."""
        sys = self.sys
        if not sys.decay_learning_rates:
            return sys.actor_lr
        return sys.actor_lr * (1.0 - (count // (sys.ppo_epochs * sys.num_minibatches)) / sys.lr_num_updates)

    def update_step(self):
        m = self.rollout()
        infos = self.update()
        return m, infos
