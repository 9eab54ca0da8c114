"""Flat fp32 parameter buffers of the guider (Sable) and the actor (GRU), with named views.

The reference keeps Flax FrozenDict pytrees (mava/systems/gpo/types.py:25-29).
Here each network owns ONE flat device buffer (so clip+Adam and the gradient all-reduce are single
launches / one RCCL message); the entries below are views into it, in Flax's natural layouts
(Dense kernels [in, out]).  Projections that always run together share one fused matrix
(w_qkvg = [w_q | w_k | w_v | w_g]) and are exposed under their reference names as column views.
Every entry starts on a 16-byte boundary (the kernels load float4).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch


def _round4(n: int) -> int:
    return (n + 3) // 4 * 4


class FlatParams:
    def __init__(self, shapes: "OrderedDict[str, Tuple[int, ...]]", device):
        self.shapes = shapes
        self.offsets: Dict[str, int] = {}
        off = 0
        for name, shp in shapes.items():
            self.offsets[name] = off
            off += _round4(math.prod(shp))
        self.numel = off
        self.device = device
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)

    def view_of(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        shp = self.shapes[name]
        o = self.offsets[name]
        return flat[o:o + math.prod(shp)].view(*shp)

    def views(self, flat: torch.Tensor = None) -> Dict[str, torch.Tensor]:
        flat = self.flat if flat is None else flat
        return {n: self.view_of(flat, n) for n in self.shapes}


def guider_layout(E: int, F: int, K: int, nb: int = 1, nh: int = 1) -> "OrderedDict[str, Tuple[int, ...]]":
    assert E in (16, 32, 64, 128) and E % nh == 0, "embed_dim must be 16, 32, 64 or 128 (nets narrower than 64 run embedded in the 64-wide kernels)"
    hs = E // nh  # GroupNorm scale / bias are per head channel (retention.py:247)
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
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
        s[p + "retn.w_qkvg"] = (E, 4 * E)
        s[p + "retn.w_o"] = (E, E)
        s[p + "retn.gn.scale"] = (hs,)
        s[p + "retn.gn.bias"] = (hs,)
        for w in ("W_linear", "W_gate", "W_output"):
            s[p + f"ffn.{w}"] = (E, E)
    s["dec.ln.scale"] = (E,)
    s["dec.act.kernel"] = (K + 1, E)
    s["dec.head.dense0.kernel"] = (E, E)
    s["dec.head.dense0.bias"] = (E,)
    s["dec.head.norm.scale"] = (E,)
    s["dec.head.dense1.kernel"] = (E, K)
    s["dec.head.dense1.bias"] = (K,)
    for b in range(nb):
        p = f"dec.block{b}."
        for n in ("ln1", "ln2", "ln3"):
            s[p + f"{n}.scale"] = (E,)
        s[p + "retn1.w_qkvg"] = (E, 4 * E)
        s[p + "retn1.w_o"] = (E, E)
        s[p + "retn1.gn.scale"] = (hs,)
        s[p + "retn1.gn.bias"] = (hs,)
        s[p + "retn2.w_q"] = (E, E)
        s[p + "retn2.w_kvg"] = (E, 3 * E)
        s[p + "retn2.w_o"] = (E, E)
        s[p + "retn2.gn.scale"] = (hs,)
        s[p + "retn2.gn.bias"] = (hs,)
        for w in ("W_linear", "W_gate", "W_output"):
            s[p + f"ffn.{w}"] = (E, E)
    return s


def guider_named_views(views: Dict[str, torch.Tensor], E: int = 64, nh: int = 1) -> Dict[str, torch.Tensor]:
    """Reference-named parameter dict (names / shapes as oracle.networks.guider_param_shapes): per-head projection
    kernels [n_head, E, hs] are strided views of the fused [E, 4E] / [E, 3E] matrices (head h = columns h*hs..)."""
    hs = E // nh
    heads = lambda m: m.unflatten(1, (nh, hs)).permute(1, 0, 2)   # [E, nh*hs] -> [nh, E, hs] (view)
    out: Dict[str, torch.Tensor] = {}
    for n, v in views.items():
        if n.endswith("w_qkvg"):
            p = n[: -len("w_qkvg")]
            out[p + "w_q"] = heads(v[:, 0:E])
            out[p + "w_k"] = heads(v[:, E:2 * E])
            out[p + "w_v"] = heads(v[:, 2 * E:3 * E])
            out[p + "w_g"] = v[:, 3 * E:4 * E]
        elif n.endswith("retn2.w_q"):
            out[n] = heads(v)
        elif n.endswith("w_kvg"):
            p = n[: -len("w_kvg")]
            out[p + "w_k"] = heads(v[:, 0:E])
            out[p + "w_v"] = heads(v[:, E:2 * E])
            out[p + "w_g"] = v[:, 2 * E:3 * E]
        else:
            out[n] = v
    return out


class WidthEmbedding:
    """Exact embedding of an embed_dim = E < 64 Sable network in the 64-wide kernels (E = 32: every feature twice, E = 16: four
    times).  Activations are carried DUPLICATED, feature i at device positions m*i .. m*i + m - 1 (m = 64 / E): then every
    mean over the 64 device channels (RMSNorm, GroupNorm over m * gs channels) equals the mean over the E logical ones, heads
    stay contiguous, and element-wise ops are unchanged.  Parameters expand accordingly (`d` = duplicate along an axis, `z` =
    value at position m*i, zeros behind it):
        norm / GroupNorm scales, biases, positional-encoding rows            d
        Dense F->E, (K+1)->E (embeddings)                                    columns d
        Dense E->E read from duplicated activations (w_o, w_g, w_v, heads)   rows z, columns d
        w_q, w_k                                                             rows z, columns z   (q.k over the logical features only)
        Dense E->1 / E->K (value / logit heads)                              rows z
    so that the 64-wide network computes the E-wide one exactly (up to fp32 summation order).  The device parameters are TIED
    copies: the optimiser works on the logical buffer, ``expand`` rebuilds the device buffer after every step and ``fold`` sums
    the device gradients of the copies (at most m addends; pure gathers, so bit-stable)."""

    def __init__(self, E: int, F: int, K: int, nb: int, nh: int, device):
        self.E, self.m = E, 64 // E
        self.L = FlatParams(guider_layout(E, F, K, nb, nh), "cpu")
        self.D = FlatParams(guider_layout(64, F, K, nb, nh), "cpu")
        m = self.m
        iota = torch.arange(1, self.L.numel + 1, dtype=torch.float64)
        lv = self.L.views(iota)
        dflat = torch.zeros(self.D.numel, dtype=torch.float64)
        dv = self.D.views(dflat)

        def d(x, ax):
            return x.repeat_interleave(m, dim=ax)

        def z(x, ax):
            shp = list(x.shape)
            shp[ax] *= m
            out = torch.zeros(shp, dtype=x.dtype)
            idx = [slice(None)] * x.dim()
            idx[ax] = slice(0, None, m)
            out[tuple(idx)] = x
            return out

        def blocks(x, kinds):   # fused projection [E, n*E]: column blocks of E with their own column rule, rows z
            cols = [(d if kd == "d" else z)(x[:, i * E:(i + 1) * E], 1) for i, kd in enumerate(kinds)]
            return z(torch.cat(cols, 1), 0)

        for n, x in lv.items():
            if n == "enc.obs.norm.scale" or n.endswith("dense1.bias"):
                y = x
            elif x.dim() == 1:
                y = d(x, 0)
            elif n in ("enc.obs.dense.kernel", "dec.act.kernel"):
                y = d(x, 1)
            elif n.endswith("dense1.kernel"):
                y = z(x, 0)
            elif n.endswith("w_qkvg"):
                y = blocks(x, "zzdd")
            elif n.endswith("w_kvg"):
                y = blocks(x, "zdd")
            elif n.endswith("retn2.w_q"):
                y = z(z(x, 1), 0)
            else:   # Dense E -> E on duplicated activations
                y = z(d(x, 1), 0)
            dv[n].copy_(y)
        src = dflat.long() - 1                      # device element <- logical element (-1: structural zero)
        self.gather = src.clamp(min=0).to(device)
        self.mask = (src >= 0).to(torch.float32).to(device)
        # fold: logical element <- its (up to m) device copies
        pos = torch.nonzero(src >= 0).squeeze(1)
        order = torch.argsort(src[pos], stable=True)
        pos, owner = pos[order], src[pos][order]
        counts = torch.bincount(owner, minlength=self.L.numel)
        start = torch.cumsum(counts, 0) - counts
        self.copies = []
        for j in range(int(counts.max())):
            has = counts > j
            idx = torch.zeros(self.L.numel, dtype=torch.long)
            idx[has] = pos[(start + j)[has]]
            self.copies.append((idx.to(device), has.to(torch.float32).to(device)))

    def expand(self, logical: torch.Tensor, out: torch.Tensor) -> None:
        torch.mul(logical[self.gather], self.mask, out=out)

    def fold(self, dev_grads: torch.Tensor, out: torch.Tensor) -> None:
        idx, has = self.copies[0]
        acc = dev_grads[idx] * has
        for idx, has in self.copies[1:]:
            acc = acc + dev_grads[idx] * has
        out.copy_(acc)

    def expand_rows(self, x: torch.Tensor) -> torch.Tensor:
        """[*, E] rows (positional encodings) -> [*, 64] duplicated."""
        return x.repeat_interleave(self.m, dim=-1).contiguous()


def actor_layout(F: int, H: int, K: int) -> "OrderedDict[str, Tuple[int, ...]]":
    assert H == 128, "the gfx950 GRU kernels are specialised for hidden_state_dim = 128"
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["pre.kernel"] = (F, H)
    s["pre.bias"] = (H,)
    s["gru.wi"] = (H, 3 * H)     # [ir | iz | in] kernels
    s["gru.bi"] = (3 * H,)
    s["gru.wh"] = (H, 3 * H)     # [hr | hz | hn] kernels
    s["gru.hn.bias"] = (H,)
    s["post.kernel"] = (H, H)
    s["post.bias"] = (H,)
    s["head.kernel"] = (H, K)
    s["head.bias"] = (K,)
    return s


def actor_named_views(views: Dict[str, torch.Tensor], H: int = 128) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for n, v in views.items():
        if n == "gru.wi":
            for i, g in enumerate(("ir", "iz", "in")):
                out[f"gru.{g}.kernel"] = v[:, i * H:(i + 1) * H]
        elif n == "gru.bi":
            for i, g in enumerate(("ir", "iz", "in")):
                out[f"gru.{g}.bias"] = v[i * H:(i + 1) * H]
        elif n == "gru.wh":
            for i, g in enumerate(("hr", "hz", "hn")):
                out[f"gru.{g}.kernel"] = v[:, i * H:(i + 1) * H]
        else:
            out[n] = v
    return out


def _orthogonal(gen, shape, gain):
    rows, cols = shape
    a = torch.randn((max(rows, cols), min(rows, cols)), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))[None, :]
    if rows < cols:
        q = q.T
    return (gain * q).float()


def init_guider(named: Dict[str, torch.Tensor], seed: int, E: int = 64) -> None:
    """Reference init distributions (sable_network.py:97,104,107,263,279,282 orthogonal(sqrt2 / 0.01);
    retention.py:50-64,237-246 normal(1/E); torsos.py:88-95 zeros; RMSNorm / GroupNorm ones/zeros)."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, v in named.items():
            if name.endswith("scale"):
                v.fill_(1.0)
            elif name.endswith("bias") or ".ffn." in name:
                v.zero_()
            elif name.split(".")[-1] in ("w_q", "w_k", "w_v", "w_g", "w_o"):
                v.copy_((torch.randn(v.shape, generator=gen, dtype=torch.float64) / E).float())
            elif name.endswith("dense1.kernel"):
                v.copy_(_orthogonal(gen, tuple(v.shape), 0.01))
            else:
                v.copy_(_orthogonal(gen, tuple(v.shape), math.sqrt(2.0)))


def init_actor(named: Dict[str, torch.Tensor], seed: int) -> None:
    """torsos.py:42 orthogonal(sqrt2); heads.py:53 orthogonal(0.01); flax GRUCell defaults
    (lecun_normal input kernels, orthogonal recurrent kernels, zero biases)."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, v in named.items():
            if name.endswith("bias"):
                v.zero_()
            elif name in ("gru.hr.kernel", "gru.hz.kernel", "gru.hn.kernel"):
                v.copy_(_orthogonal(gen, tuple(v.shape), 1.0))
            elif name.startswith("gru."):
                w = torch.empty(tuple(v.shape), dtype=torch.float64)
                torch.nn.init.trunc_normal_(w, 0.0, 1.0, -2.0, 2.0, generator=gen)
                v.copy_((w * (math.sqrt(1.0 / v.shape[0]) / 0.87962566103423978)).float())
            elif name == "head.kernel":
                v.copy_(_orthogonal(gen, tuple(v.shape), 0.01))
            else:
                v.copy_(_orthogonal(gen, tuple(v.shape), math.sqrt(2.0)))
