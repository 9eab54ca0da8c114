"""Sable guider + GRU actor, torch-CPU restatement (oracle; test infrastructure only).

Follows mava/networks/sable_network.py, retention.py, utils/sable/{encode,decode,
positional_encoding,get_init_hstates}.py, base.py:121-184, torsos.py:24-47,79-99,
heads.py:26-63.  Third-party semantics (flax 0.10.3 Dense / RMSNorm / GroupNorm / GRUCell,
jax.nn.gelu(approximate=True), distrax/tfp Categorical) are restated from their published
definitions -- PARITY UNPINNED (see oracle/__init__.py).

Parameters are plain dicts name -> tensor in Flax's natural layouts (Dense kernels [in, out]).
dtype is whatever the parameter tensors carry (float32 for parity runs, float64 for
finite-difference / tight reference checks).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from . import prng

FMIN = float(np.finfo(np.float32).min)


# ----------------------------------------------------------------------------- parameter tables
def guider_param_shapes(E: int, F: int, K: int, nh: int = 1, nb: int = 1) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of the Sable guider parameters (sable_network.py,
    retention.py:48-64,226-263, torsos.py:88-95)."""
    hs = E // nh
    s: Dict[str, Tuple[int, ...]] = {}
    s["enc.ln.scale"] = (E,)
    s["enc.obs.norm.scale"] = (F,)
    s["enc.obs.dense.kernel"] = (F, E)
    s["enc.head.dense0.kernel"] = (E, E)
    s["enc.head.dense0.bias"] = (E,)
    s["enc.head.norm.scale"] = (E,)
    s["enc.head.dense1.kernel"] = (E, 1)
    s["enc.head.dense1.bias"] = (1,)
    for b in range(nb):
        p = f"enc.block{b}."
        s[p + "ln1.scale"] = (E,)
        s[p + "ln2.scale"] = (E,)
        _retn_shapes(s, p + "retn.", E, nh, hs)
        _ffn_shapes(s, p + "ffn.", E)
    s["dec.ln.scale"] = (E,)
    s["dec.act.kernel"] = (K + 1, E)
    s["dec.head.dense0.kernel"] = (E, E)
    s["dec.head.dense0.bias"] = (E,)
    s["dec.head.norm.scale"] = (E,)
    s["dec.head.dense1.kernel"] = (E, K)
    s["dec.head.dense1.bias"] = (K,)
    for b in range(nb):
        p = f"dec.block{b}."
        s[p + "ln1.scale"] = (E,)
        s[p + "ln2.scale"] = (E,)
        s[p + "ln3.scale"] = (E,)
        _retn_shapes(s, p + "retn1.", E, nh, hs)
        _retn_shapes(s, p + "retn2.", E, nh, hs)
        _ffn_shapes(s, p + "ffn.", E)
    return s


def _retn_shapes(s, p, E, nh, hs):
    s[p + "w_q"] = (nh, E, hs)
    s[p + "w_k"] = (nh, E, hs)
    s[p + "w_v"] = (nh, E, hs)
    s[p + "w_g"] = (E, E)
    s[p + "w_o"] = (E, E)
    s[p + "gn.scale"] = (hs,)
    s[p + "gn.bias"] = (hs,)


def _ffn_shapes(s, p, E):
    s[p + "W_linear"] = (E, E)
    s[p + "W_gate"] = (E, E)
    s[p + "W_output"] = (E, E)


def actor_param_shapes(F: int, H: int, K: int) -> Dict[str, Tuple[int, ...]]:
    """RecurrentActor parameters (base.py:152-184; flax GRUCell: ir,iz,in with bias; hr,hz no
    bias; hn with bias)."""
    s: Dict[str, Tuple[int, ...]] = {}
    s["pre.kernel"] = (F, H)
    s["pre.bias"] = (H,)
    for g in ("ir", "iz", "in"):
        s[f"gru.{g}.kernel"] = (H, H)
        s[f"gru.{g}.bias"] = (H,)
    s["gru.hr.kernel"] = (H, H)
    s["gru.hz.kernel"] = (H, H)
    s["gru.hn.kernel"] = (H, H)
    s["gru.hn.bias"] = (H,)
    s["post.kernel"] = (H, H)
    s["post.bias"] = (H,)
    s["head.kernel"] = (H, K)
    s["head.bias"] = (K,)
    return s


def _orthogonal(gen, shape, gain, dtype):
    rows, cols = shape
    a = torch.randn((max(rows, cols), min(rows, cols)), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))[None, :]
    if rows < cols:
        q = q.T
    return (gain * q).to(dtype)


def init_guider_params(seed: int, E: int, F: int, K: int, nh: int = 1, nb: int = 1, dtype=torch.float32,
                       randomize_ffn: bool = False) -> Dict[str, torch.Tensor]:
    """Same init *distributions* as the reference (orthogonal(sqrt2 / 0.01), normal(1/E),
    zeros for SwiGLU, ones for norm scales); not the same bits (JAX PRNG + QR unavailable)."""
    gen = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    for name, shape in guider_param_shapes(E, F, K, nh, nb).items():
        if name.endswith("scale"):
            p[name] = torch.ones(shape, dtype=dtype)
        elif name.endswith("bias"):
            p[name] = torch.zeros(shape, dtype=dtype)
        elif ".ffn." in name:
            p[name] = (torch.randn(shape, generator=gen, dtype=torch.float64) * 0.1).to(dtype) if randomize_ffn \
                else torch.zeros(shape, dtype=dtype)
        elif name.split(".")[-1] in ("w_q", "w_k", "w_v", "w_g", "w_o"):
            p[name] = (torch.randn(shape, generator=gen, dtype=torch.float64) / E).to(dtype)
        elif name.endswith("dense1.kernel"):
            p[name] = _orthogonal(gen, shape, 0.01, dtype)
        else:
            p[name] = _orthogonal(gen, shape, math.sqrt(2.0), dtype)
    return p


def init_actor_params(seed: int, F: int, H: int, K: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    for name, shape in actor_param_shapes(F, H, K).items():
        if name.endswith("bias"):
            p[name] = torch.zeros(shape, dtype=dtype)
        elif name in ("gru.hr.kernel", "gru.hz.kernel", "gru.hn.kernel"):
            p[name] = _orthogonal(gen, shape, 1.0, dtype)
        elif name.startswith("gru."):
            std = math.sqrt(1.0 / shape[0]) / 0.87962566103423978  # lecun_normal (truncated)
            w = torch.empty(shape, dtype=torch.float64)
            torch.nn.init.trunc_normal_(w, 0.0, 1.0, -2.0, 2.0, generator=gen)
            p[name] = (w * std).to(dtype)
        elif name == "head.kernel":
            p[name] = _orthogonal(gen, shape, 0.01, dtype)
        else:
            p[name] = _orthogonal(gen, shape, math.sqrt(2.0), dtype)
    return p


# ----------------------------------------------------------------------------- primitives
def rmsnorm(x, scale, eps: float = 1e-6):
    """flax.linen.RMSNorm (eps 1e-6, scale only): x * rsqrt(mean(x^2) + eps) * scale."""
    var = (x * x).mean(dim=-1, keepdim=True)
    return x * (torch.rsqrt(var + eps) * scale)


def groupnorm_rows(x, gamma, beta, num_groups: int, eps: float = 1e-6):
    """flax.linen.GroupNorm(num_groups) applied to a 2-D (rows, hs) array: per-row statistics
    over hs/num_groups channel groups, fast variance E[x^2]-E[x]^2 clamped at 0
    (retention.py:247,289,317)."""
    shp = x.shape
    g = x.reshape(*shp[:-1], num_groups, shp[-1] // num_groups)
    mean = g.mean(dim=-1, keepdim=True)
    mean2 = (g * g).mean(dim=-1, keepdim=True)
    var = torch.clamp(mean2 - mean * mean, min=0.0)
    y = ((g - mean) * torch.rsqrt(var + eps)).reshape(shp)
    return y * gamma + beta


def gelu(x):
    """jax.nn.gelu(approximate=True)."""
    return torch.nn.functional.gelu(x, approximate="tanh")


def swish(x):
    return x * torch.sigmoid(x)


def positional_encoding(position, E: int, dtype):
    """positional_encoding.py:24-60: pe[..., 0::2] = sin(pos*div), pe[..., 1::2] = cos(pos*div);
    computed in float32 like the reference, then cast."""
    div = torch.exp(torch.arange(0, E, 2, dtype=torch.float32) * (-math.log(10000.0) / E))
    x = position.to(torch.float32)[..., None] * div
    pe = torch.zeros(*position.shape, E, dtype=torch.float32)
    pe[..., 0::2] = torch.sin(x)
    pe[..., 1::2] = torch.cos(x)
    return pe.to(dtype)


def decay_kappas(nh: int, scaling: float) -> np.ndarray:
    """sable_network.py:366-369 / retention.py:231-234 (float32 arithmetic)."""
    k = 1.0 - np.exp(np.linspace(np.log(np.float32(1 / 32)), np.log(np.float32(1 / 512)), nh, dtype=np.float32))
    return (k.astype(np.float32) * np.float32(scaling)).astype(np.float32)


def swiglu(p, pre, x):
    return (swish(x @ p[pre + "W_gate"]) * (x @ p[pre + "W_linear"])) @ p[pre + "W_output"]


# ----------------------------------------------------------------------------- retention
def _timestep_mask_reference(ts_dones):
    """retention.py:145-168 verbatim semantics (python loop over T)."""
    B, T = ts_dones.shape
    mask = torch.zeros(B, T, T, dtype=torch.bool)
    for i in range(T):
        d = ts_dones[:, i, None, None]
        xs = torch.zeros(B, T, T, dtype=torch.bool)
        ys = torch.zeros(B, T, T, dtype=torch.bool)
        xs[:, i:, :] = d
        ys[:, :, :i] = d
        mask |= xs & ys
    return ~mask


def decay_matrix(dones, n_agents: int, kappa: float, masked: bool, dtype):
    """SimpleRetention.get_decay_matrix (retention.py:117-136, 170-187)."""
    ts = dones[:, ::n_agents].bool()
    B, T = ts.shape
    n = torch.arange(T)[:, None]
    m = torch.arange(T)[None, :]
    base = torch.where(n >= m, torch.tensor(float(kappa), dtype=torch.float64) ** (n - m).clamp(min=0).double(),
                       torch.zeros((), dtype=torch.float64))
    D = base[None].expand(B, T, T) * _timestep_mask_reference(ts)
    D = D.repeat_interleave(n_agents, dim=1).repeat_interleave(n_agents, dim=2)
    if masked:
        C = D.shape[1]
        D = D * torch.tril(torch.ones(C, C, dtype=torch.float64))[None]
    return D.to(dtype)


def xi_vector(dones, n_agents: int, kappa: float, dtype):
    """SimpleRetention.get_xi (retention.py:189-213)."""
    ts = dones[:, ::n_agents].bool()
    B, T = ts.shape
    anyd = ts.any(dim=1, keepdim=True)
    first = torch.where(anyd, ts.float().argmax(dim=1, keepdim=True), torch.full((B, 1), T))
    i = torch.arange(T)[None, :]
    xi = (torch.tensor(float(kappa), dtype=torch.float64) ** (i + 1).double()) * (i < first)
    return xi.repeat_interleave(n_agents, dim=1)[..., None].to(dtype)


def msr_chunk(p, pre, key, query, value, hstate, dones, step_count, *, n_agents, nh, masked, kappas, use_pe=True):
    """MultiScaleRetention.__call__ (retention.py:265-295) + SimpleRetention.__call__ (:66-100).
    hstate (B, nh, hs, hs). Returns (out, next_hstate, ret_pre_groupnorm)."""
    B, C, E = value.shape
    hs = E // nh
    if use_pe:
        pe = positional_encoding(step_count, E, value.dtype)
        key, query, value = key + pe, query + pe, value + pe
    rets, new_h = [], []
    for h in range(nh):
        kap = float(kappas[h])
        q = query @ p[pre + "w_q"][h]
        k = key @ p[pre + "w_k"][h]
        v = value @ p[pre + "w_v"][h]
        D = decay_matrix(dones, n_agents, kap, masked, value.dtype)
        xi = xi_vector(dones, n_agents, kap, value.dtype)
        chunk_decay = kap ** (C // n_agents)
        delta = (~dones[:, ::n_agents].bool().any(dim=1))[:, None, None].to(value.dtype)
        nxt = k.transpose(1, 2) @ (v * D[:, -1].reshape(B, C, 1)) + hstate[:, h] * chunk_decay * delta
        cross = (q @ hstate[:, h]) * xi
        inner = ((q @ k.transpose(1, 2)) * D) @ v
        rets.append(inner + cross)
        new_h.append(nxt)
    ret = torch.cat(rets, dim=-1)
    rn = groupnorm_rows(ret.reshape(-1, hs), p[pre + "gn.scale"], p[pre + "gn.bias"], nh).reshape(ret.shape)
    out = (swish(key @ p[pre + "w_g"]) * rn) @ p[pre + "w_o"]
    return out, torch.stack(new_h, dim=1), ret


def msr_recurrent(p, pre, key, query, value, hstate, step_count, *, nh, use_pe=True):
    """MultiScaleRetention.recurrent (retention.py:297-323) + SimpleRetention.recurrent (:102-115)."""
    B, S, E = value.shape
    hs = E // nh
    if use_pe:
        pe = positional_encoding(step_count, E, value.dtype)
        key, query, value = key + pe, query + pe, value + pe
    rets, new_h = [], []
    for h in range(nh):
        q = query @ p[pre + "w_q"][h]
        k = key @ p[pre + "w_k"][h]
        v = value @ p[pre + "w_v"][h]
        upd = hstate[:, h] + k.transpose(1, 2) @ v
        rets.append(q @ upd)
        new_h.append(upd)
    ret = torch.cat(rets, dim=-1)
    rn = groupnorm_rows(ret.reshape(-1, hs), p[pre + "gn.scale"], p[pre + "gn.bias"], nh).reshape(ret.shape)
    out = (swish(key @ p[pre + "w_g"]) * rn) @ p[pre + "w_o"]
    return out, torch.stack(new_h, dim=1)


# ----------------------------------------------------------------------------- Sable
class SableCfg:
    def __init__(self, n_agents, action_dim, obs_dim, embed_dim=64, n_head=1, n_block=1,
                 decay_scaling_factor=0.8, use_pe=True, chunk_timesteps=None):
        self.A, self.K, self.F = n_agents, action_dim, obs_dim
        self.E, self.nh, self.nb = embed_dim, n_head, n_block
        self.kappas = decay_kappas(n_head, decay_scaling_factor)
        self.use_pe = use_pe
        self.chunk_timesteps = chunk_timesteps


def _obs_encoder(p, obs):
    x = rmsnorm(obs, p["enc.obs.norm.scale"])
    return gelu(x @ p["enc.obs.dense.kernel"])


def _value_head(p, rep):
    h = gelu(rep @ p["enc.head.dense0.kernel"] + p["enc.head.dense0.bias"])
    h = rmsnorm(h, p["enc.head.norm.scale"])
    return h @ p["enc.head.dense1.kernel"] + p["enc.head.dense1.bias"]


def _logit_head(p, x):
    h = gelu(x @ p["dec.head.dense0.kernel"] + p["dec.head.dense0.bias"])
    h = rmsnorm(h, p["dec.head.norm.scale"])
    return h @ p["dec.head.dense1.kernel"] + p["dec.head.dense1.bias"]


def encoder_chunk(p, cfg: SableCfg, obs, hstate, dones, step_count):
    """Encoder.__call__ (sable_network.py:121-137). hstate (B, nh, nb, hs, hs)."""
    rep = _obs_encoder(p, obs)
    new_h = torch.zeros_like(hstate)
    for b in range(cfg.nb):
        pre = f"enc.block{b}."
        x = rmsnorm(rep, p["enc.ln.scale"])
        ret, hs_new, _ = msr_chunk(p, pre + "retn.", x, x, x, hstate[:, :, b], dones, step_count,
                                   n_agents=cfg.A, nh=cfg.nh, masked=False, kappas=cfg.kappas, use_pe=cfg.use_pe)
        x = rmsnorm(x + ret, p[pre + "ln1.scale"])
        rep = rmsnorm(x + swiglu(p, pre + "ffn.", x), p[pre + "ln2.scale"])
        new_h[:, :, b] = hs_new
    return _value_head(p, rep), rep, new_h


def encoder_recurrent(p, cfg: SableCfg, obs, hstate, step_count):
    """Encoder.recurrent (sable_network.py:139-156)."""
    rep = _obs_encoder(p, obs)
    new_h = torch.zeros_like(hstate)
    for b in range(cfg.nb):
        pre = f"enc.block{b}."
        x = rmsnorm(rep, p["enc.ln.scale"])
        ret, hs_new = msr_recurrent(p, pre + "retn.", x, x, x, hstate[:, :, b], step_count, nh=cfg.nh, use_pe=cfg.use_pe)
        x = rmsnorm(x + ret, p[pre + "ln1.scale"])
        rep = rmsnorm(x + swiglu(p, pre + "ffn.", x), p[pre + "ln2.scale"])
        new_h[:, :, b] = hs_new
    return _value_head(p, rep), rep, new_h


def decoder_chunk(p, cfg: SableCfg, action_onehot, obs_rep, hs1, hs2, dones, step_count):
    """Decoder.__call__ (sable_network.py:296-319) + DecodeBlock.__call__ (:188-217)."""
    x = rmsnorm(gelu(action_onehot @ p["dec.act.kernel"]), p["dec.ln.scale"])
    n1, n2 = torch.zeros_like(hs1), torch.zeros_like(hs2)
    for b in range(cfg.nb):
        pre = f"dec.block{b}."
        kw = dict(n_agents=cfg.A, nh=cfg.nh, masked=True, kappas=cfg.kappas, use_pe=cfg.use_pe)
        ret, h1n, _ = msr_chunk(p, pre + "retn1.", x, x, x, hs1[:, :, b], dones, step_count, **kw)
        ret = rmsnorm(x + ret, p[pre + "ln1.scale"])
        ret2, h2n, _ = msr_chunk(p, pre + "retn2.", ret, obs_rep, ret, hs2[:, :, b], dones, step_count, **kw)
        y = rmsnorm(obs_rep + ret2, p[pre + "ln2.scale"])
        x = rmsnorm(y + swiglu(p, pre + "ffn.", y), p[pre + "ln3.scale"])
        n1[:, :, b], n2[:, :, b] = h1n, h2n
    return _logit_head(p, x), n1, n2


def decoder_recurrent(p, cfg: SableCfg, action_onehot, obs_rep, hs1, hs2, step_count):
    """Decoder.recurrent (sable_network.py:321-343) + DecodeBlock.recurrent (:219-242)."""
    x = rmsnorm(gelu(action_onehot @ p["dec.act.kernel"]), p["dec.ln.scale"])
    n1, n2 = torch.zeros_like(hs1), torch.zeros_like(hs2)
    for b in range(cfg.nb):
        pre = f"dec.block{b}."
        ret, h1n = msr_recurrent(p, pre + "retn1.", x, x, x, hs1[:, :, b], step_count, nh=cfg.nh, use_pe=cfg.use_pe)
        ret = rmsnorm(x + ret, p[pre + "ln1.scale"])
        ret2, h2n = msr_recurrent(p, pre + "retn2.", ret, obs_rep, ret, hs2[:, :, b], step_count, nh=cfg.nh, use_pe=cfg.use_pe)
        y = rmsnorm(obs_rep + ret2, p[pre + "ln2.scale"])
        x = rmsnorm(y + swiglu(p, pre + "ffn.", y), p[pre + "ln3.scale"])
        n1[:, :, b], n2[:, :, b] = h1n, h2n
    return _logit_head(p, x), n1, n2


def shifted_actions(action, K: int, n_agents: int, dtype):
    """get_shifted_discrete_actions (decode.py:86-108): one-hot into slots 1..K, roll one token,
    start token (slot 0) at every first agent."""
    B, S = action.shape
    sh = torch.zeros(B, S, K + 1, dtype=dtype)
    sh[:, :, 1:] = torch.nn.functional.one_hot(action.long(), K).to(dtype)
    sh = torch.roll(sh, shifts=1, dims=1)
    start = torch.zeros(K + 1, dtype=dtype)
    start[0] = 1
    sh[:, ::n_agents, :] = start
    return sh


def masked_log_softmax(logits, mask):
    ml = torch.where(mask, logits, torch.full_like(logits, FMIN))
    return ml - torch.logsumexp(ml, dim=-1, keepdim=True)


def sable_train(p, cfg: SableCfg, obs, action, mask, step_count, hstates, dones):
    """SableNetwork.__call__ (sable_network.py:412-441) with train_encoder_fn (encode.py:27-55)
    and discrete_train_decoder_fn (decode.py:36-83).
    obs (B,S,F), action (B,S), mask (B,S,K) bool, step_count (B,S), dones (B,S) bool,
    hstates = (enc, dec1, dec2) each (B,nh,nb,hs,hs).
    Returns value (B,S), log_prob (B,S), entropy (B,S), normalised log-probs (B,S,K)."""
    dt = p["enc.ln.scale"].dtype
    obs = obs.to(dt)
    B, S = obs.shape[:2]
    chunk = S if not cfg.chunk_timesteps else cfg.chunk_timesteps * cfg.A
    enc_h, d1, d2 = hstates
    vals, reps = [], []
    for c0 in range(0, S, chunk):
        sl = slice(c0, c0 + chunk)
        v, r, enc_h = encoder_chunk(p, cfg, obs[:, sl], enc_h, dones[:, sl], step_count[:, sl])
        vals.append(v)
        reps.append(r)
    value = torch.cat(vals, dim=1)
    obs_rep = torch.cat(reps, dim=1)
    sh = shifted_actions(action, cfg.K, cfg.A, dt)
    lgs = []
    for c0 in range(0, S, chunk):
        sl = slice(c0, c0 + chunk)
        lg, d1, d2 = decoder_chunk(p, cfg, sh[:, sl], obs_rep[:, sl], d1, d2, dones[:, sl], step_count[:, sl])
        lgs.append(lg)
    logits = torch.cat(lgs, dim=1)
    logp_all = masked_log_softmax(logits, mask)  # distrax.Categorical normalises on construction
    logp = torch.gather(logp_all, -1, action.long()[..., None])[..., 0]
    pr = logp_all.exp()
    ent = -torch.where(pr == 0, torch.zeros_like(pr), pr * logp_all).sum(-1)
    return value[..., 0], logp, ent, logp_all


def sable_get_actions(p, cfg: SableCfg, obs, mask, step_count, hstates, key, forced_actions=None):
    """SableNetwork.get_actions (sable_network.py:443-482) with act_encoder_fn (encode.py:58-84)
    and discrete_autoregressive_act (decode.py:111-153).
    obs (B,A,F), mask (B,A,K), step_count (B,A); hstates 3x(B,nh,nb,hs,hs); key (2,) uint32.
    Returns action (B,A) int32, log_prob, value, new hstates, per-agent masked-normalised logits."""
    dt = p["enc.ln.scale"].dtype
    obs = obs.to(dt)
    B, A = obs.shape[:2]
    kap = torch.tensor(cfg.kappas, dtype=dt)[None, :, None, None, None]
    enc_h, d1, d2 = (h * kap for h in hstates)
    value, rep, enc_h = encoder_recurrent(p, cfg, obs, enc_h, step_count)
    sh = torch.zeros(B, A, cfg.K + 1, dtype=dt)
    sh[:, 0, 0] = 1
    acts = torch.zeros(B, A, dtype=torch.int32)
    logps = torch.zeros(B, A, dtype=dt)
    all_lp = []
    key = np.asarray(key, np.uint32)
    for i in range(A):
        lg, d1, d2 = decoder_recurrent(p, cfg, sh[:, i:i + 1], rep[:, i:i + 1], d1, d2, step_count[:, i:i + 1])
        lp = masked_log_softmax(lg, mask[:, i:i + 1])  # (B,1,K)
        ks = prng.split(key, 2)
        key, sample_key = ks[0], ks[1]
        if forced_actions is None:
            a = torch.from_numpy(prng.categorical(sample_key, lp.detach().to(torch.float32).numpy()))[:, 0]
        else:
            a = forced_actions[:, i]
        acts[:, i] = a
        logps[:, i] = torch.gather(lp[:, 0], -1, a.long()[:, None])[:, 0]
        all_lp.append(lp[:, 0])
        if i + 1 < A:
            sh[:, i + 1, 1:] = torch.nn.functional.one_hot(a.long(), cfg.K).to(dt)
    return acts, logps, value[..., 0], (enc_h, d1, d2), torch.stack(all_lp, dim=1)


def init_sable_hstates(B, cfg: SableCfg, dtype=torch.float32):
    """get_init_hidden_state (get_init_hstates.py:20-43)."""
    hs = cfg.E // cfg.nh
    z = lambda: torch.zeros(B, cfg.nh, cfg.nb, hs, hs, dtype=dtype)
    return (z(), z(), z())


# ----------------------------------------------------------------------------- GRU actor
def gru_cell(p, h, x):
    """flax.linen.GRUCell.__call__."""
    r = torch.sigmoid(x @ p["gru.ir.kernel"] + p["gru.ir.bias"] + h @ p["gru.hr.kernel"])
    z = torch.sigmoid(x @ p["gru.iz.kernel"] + p["gru.iz.bias"] + h @ p["gru.hz.kernel"])
    n = torch.tanh(x @ p["gru.in.kernel"] + p["gru.in.bias"] + r * (h @ p["gru.hn.kernel"] + p["gru.hn.bias"]))
    return (1.0 - z) * n + z * h


def actor_apply(p, hidden, obs, done, mask):
    """RecurrentActor.__call__ (base.py:161-184) + ScannedRNN (:121-142).
    hidden (N,A,H); obs (T,N,A,F); done (T,N,A) bool; mask (T,N,A,K).
    Returns new hidden (N,A,H), normalised log-probs (T,N,A,K), per-step hidden (T,N,A,H)."""
    dt = p["pre.kernel"].dtype
    emb = torch.relu(obs.to(dt) @ p["pre.kernel"] + p["pre.bias"])
    outs = []
    h = hidden
    for t in range(obs.shape[0]):
        h = torch.where(done[t][..., None], torch.zeros_like(h), h)
        h = gru_cell(p, h, emb[t])
        outs.append(h)
    ys = torch.stack(outs, dim=0)
    y = torch.relu(ys @ p["post.kernel"] + p["post.bias"])
    logits = y @ p["head.kernel"] + p["head.bias"]
    return h, masked_log_softmax(logits, mask), ys


# ---------------------------------------------------------------------------------------------------------------------------
# Same-seed initialisation: the parameters flax would create from rec_magpo.py:598-604 (sable_network.init(net_key, ...)) and :623
# (actor_network.init(actor_net_key, ...)).  Module paths as the reference's modules name their children (sable_network.py:46-60 setup
# attributes ln1 / ln2 / retn / ffn; :97-125 Sequential children layers_<i>; :118-125,283-291 encoder_block_<i> / decoder_block_<i>;
# retention.py:237-262 w_g, w_o, group_norm, retention_heads_<h>; :50-64 w_q, w_k, w_v; base.py:152-195 pre_torso / ScannedRNN_0 /
# GRUCell_0 / post_torso / action_head with auto-named Dense_0 children), per-scope parameter counters in creation order (Dense: kernel 1,
# bias 2).  PARITY UNPINNED (oracle/prng.py: samplers and flax's key derivation are restated from memory).


def init_guider_params_from_key(net_key: np.ndarray, E: int, F: int, K: int, nh: int = 1, nb: int = 1, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    s2 = math.sqrt(2.0)
    key = lambda path, c: prng.flax_param_key(net_key, path, c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    p: Dict[str, torch.Tensor] = {}
    for name, shape in guider_param_shapes(E, F, K, nh, nb).items():   # ones / zeros first; the random tensors overwrite below
        p[name] = torch.ones(shape, dtype=dtype) if name.endswith("scale") else torch.zeros(shape, dtype=dtype)
    p["enc.obs.dense.kernel"] = t(prng.init_orthogonal(key(("encoder", "obs_encoder", "layers_1"), 1), (F, E), s2))
    p["enc.head.dense0.kernel"] = t(prng.init_orthogonal(key(("encoder", "head", "layers_0"), 1), (E, E), s2))
    p["enc.head.dense1.kernel"] = t(prng.init_orthogonal(key(("encoder", "head", "layers_3"), 1), (E, 1), 0.01))
    p["dec.act.kernel"] = t(prng.init_orthogonal(key(("decoder", "action_encoder", "layers_0"), 1), (K + 1, E), s2))
    p["dec.head.dense0.kernel"] = t(prng.init_orthogonal(key(("decoder", "head", "layers_0"), 1), (E, E), s2))
    p["dec.head.dense1.kernel"] = t(prng.init_orthogonal(key(("decoder", "head", "layers_3"), 1), (E, K), 0.01))
    hs = E // nh

    def retn(prefix, path):
        p[prefix + "w_g"] = t(prng.init_normal(key(path, 1), (E, E), 1.0 / E))
        p[prefix + "w_o"] = t(prng.init_normal(key(path, 2), (E, E), 1.0 / E))
        for c, n in enumerate(("w_q", "w_k", "w_v"), start=1):
            p[prefix + n] = t(np.stack([prng.init_normal(key(path + (f"retention_heads_{h}",), c), (E, hs), 1.0 / E) for h in range(nh)]))

    for b in range(nb):
        retn(f"enc.block{b}.retn.", ("encoder", f"encoder_block_{b}", "retn"))
        retn(f"dec.block{b}.retn1.", ("decoder", f"decoder_block_{b}", "retn1"))
        retn(f"dec.block{b}.retn2.", ("decoder", f"decoder_block_{b}", "retn2"))
    return p


def init_actor_params_from_key(actor_net_key: np.ndarray, F: int, H: int, K: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    s2 = math.sqrt(2.0)
    key = lambda path, c: prng.flax_param_key(actor_net_key, path, c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    p = {name: torch.zeros(shape, dtype=dtype) for name, shape in actor_param_shapes(F, H, K).items()}
    p["pre.kernel"] = t(prng.init_orthogonal(key(("pre_torso", "Dense_0"), 1), (F, H), s2))
    cell = ("ScannedRNN_0", "GRUCell_0")
    for g in ("ir", "iz", "in"):
        p[f"gru.{g}.kernel"] = t(prng.init_lecun_normal(key(cell + (g,), 1), (H, H)))
    for g in ("hr", "hz", "hn"):
        p[f"gru.{g}.kernel"] = t(prng.init_orthogonal(key(cell + (g,), 1), (H, H), 1.0))
    p["post.kernel"] = t(prng.init_orthogonal(key(("post_torso", "Dense_0"), 1), (H, H), s2))
    p["head.kernel"] = t(prng.init_orthogonal(key(("action_head", "Dense_0"), 1), (H, K), 0.01))
    return p
