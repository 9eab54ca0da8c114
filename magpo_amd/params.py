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

import numpy as np
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


# ---------------------------------------------------------------------------------------------------------------------------
# Same-seed initialisation (rec_magpo.py:598-604: sable_network.init(net_key, ...); :623: actor_network.init(actor_net_key, ...)).
# The reference's parameters are a function of (net_key, actor_net_key): flax derives one key per parameter from the 'params' key and the
# module path, and jax's initialisers draw from it.  This restates both on the host (threefry through the C ABI's host entry points,
# float32 numpy for the samplers, LAPACK for the QR) so that ``system.seed`` selects the same run here as there -- up to what cannot be
# checked without JAX (the restatement is from memory; the last bits of XLA's erf_inv / QR): UNPINNED, see oracle/prng.py.
# tests/test_oracle_prng.py compares these arrays with the oracle's restatement bit for bit and checks the distributions.

def _fold_in(key: np.ndarray, data: int) -> np.ndarray:
    from ._lib import lib
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(2, np.uint32)
    lib().raw("magpo_key_fold_in_host")(key.ctypes.data, int(data) & 0xFFFFFFFF, out.ctypes.data)
    return out


def _param_key(params_key: np.ndarray, path: Tuple[str, ...], counter: int) -> np.ndarray:
    """flax/core/scope.py: LazyRng(params_key, path + (counter,)) -> fold_in(params_key, sha1(names and integer bytes)[:4] big endian)."""
    import hashlib
    m = hashlib.sha1()
    for x in tuple(path) + (int(counter),):
        m.update(x.encode("utf-8") if isinstance(x, str) else int(x).to_bytes((int(x).bit_length() + 7) // 8, byteorder="big"))
    return _fold_in(params_key, int.from_bytes(m.digest()[:4], byteorder="big"))


def _uniform(key: np.ndarray, n: int, lo: np.float32, hi: np.float32) -> np.ndarray:
    from ._lib import lib
    key = np.ascontiguousarray(key, dtype=np.uint32)
    bits = np.empty(n, np.uint32)
    lib().raw("magpo_random_bits_host")(key.ctypes.data, int(n), bits.ctypes.data)
    u = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    return np.maximum(lo, u * (hi - lo) + lo).astype(np.float32)


def _erf_inv(x: np.ndarray) -> np.ndarray:
    """float32 erf_inv: M. Giles' single-precision polynomial (central branch for -log(1 - x^2) < 5, tail branch beyond)."""
    f = np.float32
    w = -np.log1p(-(x * x)).astype(np.float32)
    a = (w - f(2.5)).astype(np.float32)
    p = f(2.81022636e-08)
    for c in (3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941):
        p = (f(c) + p * a).astype(np.float32)
    b = (np.sqrt(np.maximum(w, f(5.0))).astype(np.float32) - f(3.0)).astype(np.float32)
    q = f(-0.000200214257)
    for c in (0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682):
        q = (f(c) + q * b).astype(np.float32)
    return (np.where(w < f(5.0), p, q) * x).astype(np.float32)


def _normal(key: np.ndarray, shape) -> np.ndarray:
    n = int(np.prod(shape))
    u = _uniform(key, n, np.nextafter(np.float32(-1.0), np.float32(0.0)), np.float32(1.0))
    return (np.float32(np.sqrt(2)) * _erf_inv(u)).astype(np.float32).reshape(shape)


def _trunc_normal(key: np.ndarray, shape, lower: float = -2.0, upper: float = 2.0) -> np.ndarray:
    f = np.float32
    s2 = f(np.sqrt(2))
    a, b = f(math.erf(float(f(lower) / s2))), f(math.erf(float(f(upper) / s2)))
    out = (s2 * _erf_inv(_uniform(key, int(np.prod(shape)), a, b))).astype(np.float32)
    return np.clip(out, np.nextafter(f(lower), f(np.inf)), np.nextafter(f(upper), f(-np.inf))).astype(np.float32).reshape(shape)


def _orth(key: np.ndarray, shape, scale: float) -> np.ndarray:
    rows, cols = int(shape[0]), int(shape[1])
    a = _normal(key, (cols, rows) if rows < cols else (rows, cols))
    q, r = np.linalg.qr(a)
    q = (q * np.sign(np.diag(r))[None, :]).astype(np.float32)
    return (np.float32(scale) * (q.T if rows < cols else q)).astype(np.float32)


def init_guider_from_key(named: Dict[str, torch.Tensor], net_key: np.ndarray, E: int, n_head: int) -> None:
    """The Sable parameters flax creates from ``net_key`` (module paths: oracle/networks.py:init_guider_params_from_key and the
    reference lines cited there).  ``named``: the logical named views (retention projections as [n_head, E, E / n_head])."""
    s2 = math.sqrt(2.0)
    key = lambda path, c: _param_key(net_key, path, c)
    hs = E // n_head
    with torch.no_grad():
        def put(name, arr):
            named[name].copy_(torch.from_numpy(np.ascontiguousarray(arr)).reshape(named[name].shape))
        for name, v in named.items():
            v.fill_(1.0) if name.endswith("scale") else v.zero_()
        F, K = named["enc.obs.dense.kernel"].shape[0], named["dec.head.dense1.kernel"].shape[-1]
        put("enc.obs.dense.kernel", _orth(key(("encoder", "obs_encoder", "layers_1"), 1), (F, E), s2))
        put("enc.head.dense0.kernel", _orth(key(("encoder", "head", "layers_0"), 1), (E, E), s2))
        put("enc.head.dense1.kernel", _orth(key(("encoder", "head", "layers_3"), 1), (E, 1), 0.01))
        put("dec.act.kernel", _orth(key(("decoder", "action_encoder", "layers_0"), 1), (K + 1, E), s2))
        put("dec.head.dense0.kernel", _orth(key(("decoder", "head", "layers_0"), 1), (E, E), s2))
        put("dec.head.dense1.kernel", _orth(key(("decoder", "head", "layers_3"), 1), (E, K), 0.01))
        b = 0
        while f"enc.block{b}.retn.w_g" in named:
            for prefix, path in ((f"enc.block{b}.retn.", ("encoder", f"encoder_block_{b}", "retn")),
                                 (f"dec.block{b}.retn1.", ("decoder", f"decoder_block_{b}", "retn1")),
                                 (f"dec.block{b}.retn2.", ("decoder", f"decoder_block_{b}", "retn2"))):
                put(prefix + "w_g", _normal(key(path, 1), (E, E)) * np.float32(1.0 / E))
                put(prefix + "w_o", _normal(key(path, 2), (E, E)) * np.float32(1.0 / E))
                for c, n in enumerate(("w_q", "w_k", "w_v"), start=1):
                    put(prefix + n, np.stack([_normal(key(path + (f"retention_heads_{h}",), c), (E, hs)) * np.float32(1.0 / E) for h in range(n_head)]))
            b += 1


def init_actor_from_key(named: Dict[str, torch.Tensor], actor_net_key: np.ndarray) -> None:
    """The RecurrentActor parameters flax creates from ``actor_net_key`` (oracle/networks.py:init_actor_params_from_key)."""
    s2 = math.sqrt(2.0)
    key = lambda path, c: _param_key(actor_net_key, path, c)
    with torch.no_grad():
        def put(name, arr):
            named[name].copy_(torch.from_numpy(np.ascontiguousarray(arr)).reshape(named[name].shape))
        for v in named.values():
            v.zero_()
        F, H = named["pre.kernel"].shape
        K = named["head.kernel"].shape[1]
        put("pre.kernel", _orth(key(("pre_torso", "Dense_0"), 1), (F, H), s2))
        cell = ("ScannedRNN_0", "GRUCell_0")
        std = np.float32(np.sqrt(np.float32(1.0) / np.float32(H))) / np.float32(0.87962566103423978)
        for g in ("ir", "iz", "in"):
            put(f"gru.{g}.kernel", _trunc_normal(key(cell + (g,), 1), (H, H)) * std)
        for g in ("hr", "hz", "hn"):
            put(f"gru.{g}.kernel", _orth(key(cell + (g,), 1), (H, H), 1.0))
        put("post.kernel", _orth(key(("post_torso", "Dense_0"), 1), (H, H), s2))
        put("head.kernel", _orth(key(("action_head", "Dense_0"), 1), (H, K), 0.01))
