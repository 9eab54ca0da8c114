"""Sable guider on the MI355X kernels: recurrent acting and chunkwise training forward/backward.

Host-side composition of the C-ABI kernels (include/magpo.h); mirrors ``SableNetwork.get_actions``
(mava/networks/sable_network.py:443-482) and ``SableNetwork.__call__`` (:412-441).  The backward pass
is hand-derived (no autograd): it walks the forward graph in reverse, one kernel per node.

Supported configuration (asserted): embed_dim 64, n_head 1, n_block 1, discrete actions, one chunk
per rollout, SwiGLU weights at their zero init (then the FFN branch and its gradients are exactly 0,
SURVEY B5 -- verified on the host at construction / load time).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ._lib import lib
from .params import FlatParams, guider_layout, guider_named_views, init_guider

E = 64


def decay_kappa(n_head: int, scaling: float) -> float:
    """sable_network.py:366-369 in float32 for n_head = 1."""
    k = np.float32(1.0) - np.exp(np.log(np.float32(1 / 32)))
    return float(np.float32(k) * np.float32(scaling))


class _Bufs:
    def __init__(self, device):
        self.device = device
        self.t: Dict[str, torch.Tensor] = {}

    def get(self, name, shape, dtype=torch.float32, zero=False):
        t = self.t.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.device)
            self.t[name] = t
        return t


class SableGuider:
    def __init__(self, n_agents: int, action_dim: int, obs_dim: int, device, *, embed_dim: int = 64, n_head: int = 1,
                 n_block: int = 1, decay_scaling_factor: float = 0.8, use_pe: bool = True, max_pos: int = 101,
                 wgrad_groups: int = 512, seed: Optional[int] = None, grads: Optional[torch.Tensor] = None):
        if embed_dim != 64 or n_head != 1 or n_block != 1:
            raise NotImplementedError("gfx950 Sable kernels: embed_dim=64, n_head=1, n_block=1 only (SURVEY 8f rank 3)")
        if obs_dim > 32 or action_dim > 31:
            raise NotImplementedError("obs_dim <= 32 and action_dim <= 31 required")
        self.A, self.K, self.F = n_agents, action_dim, obs_dim
        self.dev = device
        self.L = lib()
        self.kappa = decay_kappa(1, decay_scaling_factor)
        self.G = wgrad_groups
        self.P = FlatParams(guider_layout(E, obs_dim, action_dim), device)
        self.grads = torch.zeros_like(self.P.flat) if grads is None else grads
        assert self.grads.numel() == self.P.numel
        self.v = self.P.views()
        self.gv = self.P.views(self.grads)
        self.named = guider_named_views(self.v)
        self.named_grads = guider_named_views(self.gv)
        if seed is not None:
            init_guider(self.named, seed)
        self.npos = max_pos
        self.pe = torch.zeros(max_pos, E, device=device)
        if use_pe:
            self.L.call("magpo_pe_table", self.pe, max_pos, E, self._st())
        self.wt: Dict[str, torch.Tensor] = {}
        self.b = _Bufs(device)
        # weight-gradient GEMMs run on a side stream: they are off the critical path of the backward chain
        self.wgrad_stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self.overlap_wgrad = False  # opt-in (bench.py --overlap): ~0.5 %, but per-kernel timings then include contention
        self.wg_ws = torch.empty(self.L.call("magpo_wgrad_workspace_floats", E, 4 * E, self.G), device=device)
        self.refresh()

    # ------------------------------------------------------------------ plumbing
    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def check_ffn_zero(self):
        for n, v in self.v.items():
            if ".ffn." in n and bool(v.any().item()):
                raise NotImplementedError("non-zero SwiGLU weights: the FFN branch is not evaluated by the HIP path")

    def load_named(self, params: Dict[str, torch.Tensor]):
        with torch.no_grad():
            for n, v in self.named.items():
                v.copy_(params[n].to(self.dev, torch.float32).reshape(v.shape))
        self.check_ffn_zero()
        self.refresh()

    def _tp(self, name, W, npad=None):
        K_, N_ = W.shape
        npad = npad or (N_ + 31) // 32 * 32
        t = self.wt.get(name)
        if t is None:
            t = torch.zeros(npad, K_, device=self.dev)
            self.wt[name] = t
        self.L.call("magpo_transpose_pad", W, t, K_, N_, npad, self._st())
        return t

    def refresh(self):
        """Rebuild the transposed (forward-GEMM) weight copies after a parameter update."""
        v = self.v
        for key, name in [("qkvg", "enc.block0.retn.w_qkvg"), ("wo", "enc.block0.retn.w_o"), ("vh0", "enc.head.dense0.kernel"),
                          ("qkvg1", "dec.block0.retn1.w_qkvg"), ("wo1", "dec.block0.retn1.w_o"), ("q2", "dec.block0.retn2.w_q"),
                          ("kvg2", "dec.block0.retn2.w_kvg"), ("wo2", "dec.block0.retn2.w_o"), ("h0", "dec.head.dense0.kernel")]:
            self._tp(key, v[name])
        h1t = self._tp("h1", v["dec.head.dense1.kernel"], 64)        # [64 (K padded)][64]
        self._tp("h1_nat_pad", h1t, E)                                # [64][64]: natural W padded to 64 columns

    def lin(self, X, ldx, Wt, bias, Y, ldy, R, KIN, NOUT, act=0, Ypre=None):
        self.L.call("magpo_linear", X, ldx, Wt, bias, Y, ldy, Ypre, R, KIN, NOUT, act, self._st())

    def wgrad(self, X, ldx, dY, ldy, R, KIN, NOUT, dW, db=None, krows=None):
        """dW = X^T dY.  With overlap_wgrad the GEMM is queued on the side stream behind everything the calling stream
        has queued so far (so X and dY are complete); the caller must not overwrite dY before train_bwd joins."""
        side = self.wgrad_stream if self.overlap_wgrad else None
        if side is None:
            self.L.call("magpo_wgrad", X, ldx, dY, ldy, R, KIN, krows or KIN, NOUT, dW, db, self.wg_ws, self.G, 1.0, 0, self._st())
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.L.call("magpo_wgrad", X, ldx, dY, ldy, R, KIN, krows or KIN, NOUT, dW, db, self.wg_ws, self.G, 1.0, 0, self._st())

    def reduce(self, slab, out, P=64, stride=None):
        self.L.call("magpo_reduce_slabs", slab, out, slab.shape[0], P, stride or slab.shape[1], 1.0, 0, self._st())

    # ------------------------------------------------------------------ acting (recurrent form)
    def _pro(self, pro, a, lda, y, ldy_in, s1, s2, use_pe, pos, pos_stride, W, idx, idx_stride, out, ldout, outpe, ldoutpe,
             Wt, bias, Y, ldy, R, NOUT):
        """Dense layer with the preceding row-wise op fused into its prologue (csrc/linear.hip: k_linear_pro)."""
        self.L.call("magpo_linear_pro", pro, a, lda, y, ldy_in, s1, s2, self.pe, pos, pos_stride, self.npos, 1 if use_pe else 0,
                    W, idx, idx_stride, self.v["enc.obs.norm.scale"], self.F, out, ldout, outpe, ldoutpe, Wt, bias, Y, ldy, R, NOUT,
                    self._st())

    def act(self, obs, pos, states, sample_keys, action_out, logp_out, value_out, mask=None, value_only=False):
        """One env step for N envs (SableNetwork.get_actions, sable_network.py:443-482).  obs [N,A,F] f32, pos [N] i32
        (step_count), states = (S_enc, S_d1, S_d2) each [N,64,64] updated in place, sample_keys = [A,2] uint32 (host
        array: keys by value; device tensor: static arguments for graph replay).
        Writes action [N,A] i32, logp [N,A], value [N,A].  Row-wise ops are fused into the prologue of the dense layer
        that follows them and the GroupNorm + swish gate into the recurrent retention kernel: 6 + 9*A launches."""
        L, st, A, K, F = self.L, self._st(), self.A, self.K, self.F
        N = obs.shape[0]
        R = N * A
        v, b = self.v, self.b
        xn = b.get("a_xn", (R, E)); qkvg = b.get("a_qkvg", (R, 4 * E)); u = b.get("a_u", (R, E)); y = b.get("a_y", (R, E))
        rep = b.get("a_rep", (R, E)); reppe = b.get("a_reppe", (R, E)); hv = b.get("a_hv", (R, E))
        s_enc, s_d1, s_d2 = states
        if value_only:  # bootstrap value (rec_magpo.py:202-208): states must not change
            s_enc = b.get("a_senc_tmp", tuple(s_enc.shape)).copy_(s_enc)
        pos_tok = b.get("a_pos", (R,), torch.int32)
        pos_tok.view(N, A).copy_(pos.view(N, 1).expand(N, A))
        # encoder over the A tokens of this timestep (act_encoder_fn, encode.py:58-84)
        self._pro(2, obs, F, None, 0, v["enc.ln.scale"], None, True, pos_tok, 1, v["enc.obs.dense.kernel"], None, 0, xn, E, None, 0,
                  self.wt["qkvg"], None, qkvg, 4 * E, R, 4 * E)
        L.call("magpo_retention_recurrent", s_enc, qkvg, 4 * E, qkvg[:, E:], 4 * E, qkvg[:, 2 * E:], 4 * E, A, u, E, N, A, 0,
               self.kappa, 1, qkvg[:, 3 * E:], 4 * E, v["enc.block0.retn.gn.scale"], v["enc.block0.retn.gn.bias"], st)
        self.lin(u, E, self.wt["wo"], None, y, E, R, E, E)
        self._pro(3, xn, E, y, E, v["enc.block0.ln1.scale"], v["enc.block0.ln2.scale"], False, pos_tok, 1, None, None, 0, rep, E, reppe, E,
                  self.wt["vh0"], v["enc.head.dense0.bias"], hv, E, R, E)
        L.call("magpo_headmid_fwd", hv, E, v["enc.head.norm.scale"], None, 0, v["enc.head.dense1.kernel"],
               v["enc.head.dense1.bias"], value_out, 1, R, st)
        if value_only:
            return
        # autoregressive decoder (decode.py:111-153): one token per env per iteration.  Per-agent projections stay
        # resident ([N, A, .]) so that a retention state is read once per agent and written once per step.
        prev = b.get("a_prev", (N, A), torch.int32, zero=True)
        xa = b.get("d_xa", (N, E)); qkvg1 = b.get("d_qkvg1", (R, 4 * E)); u1 = b.get("d_u1", (R, E)); y1 = b.get("d_y1", (N, E))
        q2 = b.get("d_q2", (R, E)); kvg2 = b.get("d_kvg2", (R, 3 * E)); u2 = b.get("d_u2", (R, E)); y2 = b.get("d_y2", (N, E))
        hp = b.get("d_hp", (N, E)); logits = b.get("d_logits", (N, E), zero=True)
        self.lin(reppe, E, self.wt["q2"], None, q2, E, R, E, E)   # cross-retention queries of all agents at once
        for i in range(A):
            last = 1 if i == A - 1 else 0
            self._pro(1, None, 0, None, 0, v["dec.ln.scale"], None, True, pos, 1, v["dec.act.kernel"], prev[:, i:], A, xa, E, None, 0,
                      self.wt["qkvg1"], None, qkvg1[i:], A * 4 * E, N, 4 * E)
            L.call("magpo_retention_recurrent", s_d1, qkvg1, 4 * E, qkvg1[:, E:], 4 * E, qkvg1[:, 2 * E:], 4 * E, A, u1, E, N, i + 1, i,
                   self.kappa, last, qkvg1[:, 3 * E:], 4 * E, v["dec.block0.retn1.gn.scale"], v["dec.block0.retn1.gn.bias"], st)
            self.lin(u1[i:], A * E, self.wt["wo1"], None, y1, E, N, E, E)
            self._pro(3, xa, E, y1, E, v["dec.block0.ln1.scale"], None, True, pos, 1, None, None, 0, None, 0, None, 0,
                      self.wt["kvg2"], None, kvg2[i:], A * 3 * E, N, 3 * E)
            L.call("magpo_retention_recurrent", s_d2, q2, E, kvg2, 3 * E, kvg2[:, E:], 3 * E, A, u2, E, N, i + 1, i, self.kappa, last,
                   kvg2[:, 2 * E:], 3 * E, v["dec.block0.retn2.gn.scale"], v["dec.block0.retn2.gn.bias"], st)
            self.lin(u2[i:], A * E, self.wt["wo2"], None, y2, E, N, E, E)
            self._pro(3, rep[i:], A * E, y2, E, v["dec.block0.ln2.scale"], v["dec.block0.ln3.scale"], False, pos, 1, None, None, 0,
                      None, 0, None, 0, self.wt["h0"], v["dec.head.dense0.bias"], hp, E, N, E)
            self._pro(4, hp, E, None, 0, v["dec.head.norm.scale"], None, False, pos, 1, None, None, 0, None, 0, None, 0,
                      self.wt["h1"], v["dec.head.dense1.bias"], logits, E, N, K)
            if torch.is_tensor(sample_keys):   # device key table [A, 2] (static arguments: HIP-graph replay)
                k0, k1, kdev = 0, 0, sample_keys[i]
            else:
                k0, k1, kdev = int(sample_keys[i][0]), int(sample_keys[i][1]), None
            L.call("magpo_sample_categorical", logits, E, None if mask is None else mask[:, i], (A * K if mask is not None else 0),
                   k0, k1, kdev, action_out[:, i:], A, logp_out[:, i:], A, prev[:, i + 1:] if i + 1 < A else None, A, None, 0, N, K, st)

    # ------------------------------------------------------------------ training forward (chunkwise form)
    def train_fwd(self, obs, prev_idx, pos, dones, s0, seq_env, nseq: int, T: int):
        """obs [R,F], prev_idx [R] (0 = start token, a+1 otherwise), pos [R] step counts, dones [nseq,T] u8,
        s0 = three [N,64,64] rollout-start states indexed through seq_env [nseq].
        Returns (logits [R,64] raw with K valid columns, value [R])."""
        L, st, A, K, F, v, b = self.L, self._st(), self.A, self.K, self.F, self.v, self.b
        R = nseq * T * A
        nch = L.call("magpo_retention_num_chunks", T, A)
        g = lambda n, w=E: b.get("t_" + n, (R, w))
        z, xn, kin, qkvg, r, u, y = g("z"), g("xn"), g("kin"), g("qkvg", 4 * E), g("r"), g("u"), g("y")
        rep, reppe, hv, value = g("rep"), g("reppe"), g("hv"), b.get("t_value", (R,))
        za, xa, kin1, qkvg1, r1, u1, y1 = g("za"), g("xa"), g("kin1"), g("qkvg1", 4 * E), g("r1"), g("u1"), g("y1")
        c, cpe, q2, kvg2, r2, u2, y2 = g("c"), g("cpe"), g("q2"), g("kvg2", 3 * E), g("r2"), g("u2"), g("y2")
        out, hp, hn = g("out"), g("hp"), g("hn")
        logits = b.get("t_logits", (R, E), zero=True)
        st_e = b.get("t_st_e", (nseq, nch, E, E)); st_1 = b.get("t_st_1", (nseq, nch, E, E)); st_2 = b.get("t_st_2", (nseq, nch, E, E))
        self._saved = dict(obs=obs, prev_idx=prev_idx, pos=pos, dones=dones, nseq=nseq, T=T, R=R)
        L.call("magpo_embed_fwd", 0, obs, F, F, v["enc.obs.norm.scale"], v["enc.obs.dense.kernel"], None, 0, v["enc.ln.scale"],
               self.pe, pos, 1, self.npos, z, E, xn, E, kin, E, R, st)
        self.lin(kin, E, self.wt["qkvg"], None, qkvg, 4 * E, R, E, 4 * E)
        L.call("magpo_retention_chunk_fwd", qkvg, 4 * E, qkvg[:, E:], 4 * E, qkvg[:, 2 * E:], 4 * E, r, E, s0[0], seq_env, dones,
               st_e, None, nseq, T, A, 0, self.kappa, st)
        L.call("magpo_retpost_fwd", r, E, qkvg[:, 3 * E:], 4 * E, v["enc.block0.retn.gn.scale"], v["enc.block0.retn.gn.bias"], u, E, R, st)
        self.lin(u, E, self.wt["wo"], None, y, E, R, E, E)
        L.call("magpo_resnorm_fwd", xn, E, y, E, v["enc.block0.ln1.scale"], v["enc.block0.ln2.scale"], self.pe, pos, 1, self.npos,
               rep, E, reppe, E, R, st)
        self.lin(rep, E, self.wt["vh0"], v["enc.head.dense0.bias"], hv, E, R, E, E)
        L.call("magpo_headmid_fwd", hv, E, v["enc.head.norm.scale"], None, 0, v["enc.head.dense1.kernel"], v["enc.head.dense1.bias"],
               value, 1, R, st)
        L.call("magpo_embed_fwd", 1, None, 0, 0, None, v["dec.act.kernel"], prev_idx, 1, v["dec.ln.scale"], self.pe, pos, 1, self.npos,
               za, E, xa, E, kin1, E, R, st)
        self.lin(kin1, E, self.wt["qkvg1"], None, qkvg1, 4 * E, R, E, 4 * E)
        L.call("magpo_retention_chunk_fwd", qkvg1, 4 * E, qkvg1[:, E:], 4 * E, qkvg1[:, 2 * E:], 4 * E, r1, E, s0[1], seq_env, dones,
               st_1, None, nseq, T, A, 1, self.kappa, st)
        L.call("magpo_retpost_fwd", r1, E, qkvg1[:, 3 * E:], 4 * E, v["dec.block0.retn1.gn.scale"], v["dec.block0.retn1.gn.bias"], u1, E, R, st)
        self.lin(u1, E, self.wt["wo1"], None, y1, E, R, E, E)
        L.call("magpo_resnorm_fwd", xa, E, y1, E, v["dec.block0.ln1.scale"], None, self.pe, pos, 1, self.npos, c, E, cpe, E, R, st)
        self.lin(reppe, E, self.wt["q2"], None, q2, E, R, E, E)
        self.lin(cpe, E, self.wt["kvg2"], None, kvg2, 3 * E, R, E, 3 * E)
        L.call("magpo_retention_chunk_fwd", q2, E, kvg2, 3 * E, kvg2[:, E:], 3 * E, r2, E, s0[2], seq_env, dones, st_2, None,
               nseq, T, A, 1, self.kappa, st)
        L.call("magpo_retpost_fwd", r2, E, kvg2[:, 2 * E:], 3 * E, v["dec.block0.retn2.gn.scale"], v["dec.block0.retn2.gn.bias"], u2, E, R, st)
        self.lin(u2, E, self.wt["wo2"], None, y2, E, R, E, E)
        L.call("magpo_resnorm_fwd", rep, E, y2, E, v["dec.block0.ln2.scale"], v["dec.block0.ln3.scale"], None, None, 0, 0, out, E,
               None, 0, R, st)
        self.lin(out, E, self.wt["h0"], v["dec.head.dense0.bias"], hp, E, R, E, E)
        L.call("magpo_headmid_fwd", hp, E, v["dec.head.norm.scale"], hn, E, None, None, None, 0, R, st)
        self.lin(hn, E, self.wt["h1"], v["dec.head.dense1.bias"], logits, E, R, E, K)
        return logits, value

    # ------------------------------------------------------------------ training backward
    def train_bwd(self, dlogits, dvalue):
        """dlogits [R,64] (columns >= K zero), dvalue [R]; fills self.grads (every entry written once)."""
        L, st, A, K, F, v, gv, b = self.L, self._st(), self.A, self.K, self.F, self.v, self.gv, self.b
        sv = self._saved
        R, nseq, T = sv["R"], sv["nseq"], sv["T"]
        obs, prev_idx, pos, dones = sv["obs"], sv["prev_idx"], sv["pos"], sv["dones"]
        t = lambda n, w=E: b.t["t_" + n]
        g = lambda n, w=E: b.get("g_" + n, (R, w))
        grid = L.call("magpo_row_grid", R)
        slab = lambda n, w=64: b.get("s_" + n, (grid, w))
        # ---- logit head
        self.wgrad(t("hn"), E, dlogits, E, R, E, K, gv["dec.head.dense1.kernel"], gv["dec.head.dense1.bias"])
        dhn = g("dhn")
        self.lin(dlogits, E, self.wt["h1_nat_pad"], None, dhn, E, R, E, E)
        dhp = g("dhp_l")
        L.call("magpo_headmid_bwd", t("hp"), E, v["dec.head.norm.scale"], dhn, E, None, None, 0, dhp, E, slab("a"), None, None, R, st)
        self.reduce(slab("a"), gv["dec.head.norm.scale"])
        self.wgrad(t("out"), E, dhp, E, R, E, E, gv["dec.head.dense0.kernel"], gv["dec.head.dense0.bias"])
        dout = g("dout")
        self.lin(dhp, E, v["dec.head.dense0.kernel"], None, dout, E, R, E, E)
        # ---- decoder block tail: out = rms(rms(rep + y2) * ln2) * ln3
        dsum2 = g("dsum2")
        L.call("magpo_resnorm_bwd", t("rep"), E, t("y2"), E, v["dec.block0.ln2.scale"], v["dec.block0.ln3.scale"], dout, E, None, 0,
               None, 0, dsum2, E, slab("a"), slab("b"), R, st)
        self.reduce(slab("a"), gv["dec.block0.ln2.scale"]); self.reduce(slab("b"), gv["dec.block0.ln3.scale"])
        self.wgrad(t("u2"), E, dsum2, E, R, E, E, gv["dec.block0.retn2.w_o"])
        du2 = g("du")
        self.lin(dsum2, E, v["dec.block0.retn2.w_o"], None, du2, E, R, E, E)
        dr2 = g("dr"); dq2 = g("dq2"); dkvg2 = g("dkvg2", 3 * E)
        L.call("magpo_retpost_bwd", t("r2"), E, t("kvg2")[:, 2 * E:], 3 * E, v["dec.block0.retn2.gn.scale"], v["dec.block0.retn2.gn.bias"],
               du2, E, dr2, E, dkvg2[:, 2 * E:], 3 * E, slab("a"), slab("b"), R, st)
        self.reduce(slab("a"), gv["dec.block0.retn2.gn.scale"]); self.reduce(slab("b"), gv["dec.block0.retn2.gn.bias"])
        kvg2 = t("kvg2")
        L.call("magpo_retention_chunk_bwd", t("q2"), E, kvg2, 3 * E, kvg2[:, E:], 3 * E, dr2, E, dq2, E, dkvg2, 3 * E, dkvg2[:, E:], 3 * E,
               dones, b.t["t_st_2"], nseq, T, A, 1, self.kappa, st)
        self.wgrad(t("reppe"), E, dq2, E, R, E, E, gv["dec.block0.retn2.w_q"])
        self.wgrad(t("cpe"), E, dkvg2, 3 * E, R, E, 3 * E, gv["dec.block0.retn2.w_kvg"])
        dreppe = g("dreppe"); dcpe = g("dcpe")
        self.lin(dq2, E, v["dec.block0.retn2.w_q"], None, dreppe, E, R, E, E)
        self.lin(dkvg2, 3 * E, v["dec.block0.retn2.w_kvg"], None, dcpe, E, R, 3 * E, E)
        # ---- decoder self-retention: c = rms(xa + y1) * ln1
        dsum1 = g("dsum1")
        L.call("magpo_resnorm_bwd", t("xa"), E, t("y1"), E, v["dec.block0.ln1.scale"], None, dcpe, E, None, 0, None, 0, dsum1, E,
               slab("a"), None, R, st)
        self.reduce(slab("a"), gv["dec.block0.ln1.scale"])
        self.wgrad(t("u1"), E, dsum1, E, R, E, E, gv["dec.block0.retn1.w_o"])
        du1 = g("du")
        self.lin(dsum1, E, v["dec.block0.retn1.w_o"], None, du1, E, R, E, E)
        dr1 = g("dr"); dqkvg1 = g("dqkvg1", 4 * E)
        qkvg1 = t("qkvg1")
        L.call("magpo_retpost_bwd", t("r1"), E, qkvg1[:, 3 * E:], 4 * E, v["dec.block0.retn1.gn.scale"], v["dec.block0.retn1.gn.bias"],
               du1, E, dr1, E, dqkvg1[:, 3 * E:], 4 * E, slab("a"), slab("b"), R, st)
        self.reduce(slab("a"), gv["dec.block0.retn1.gn.scale"]); self.reduce(slab("b"), gv["dec.block0.retn1.gn.bias"])
        L.call("magpo_retention_chunk_bwd", qkvg1, 4 * E, qkvg1[:, E:], 4 * E, qkvg1[:, 2 * E:], 4 * E, dr1, E, dqkvg1, 4 * E,
               dqkvg1[:, E:], 4 * E, dqkvg1[:, 2 * E:], 4 * E, dones, b.t["t_st_1"], nseq, T, A, 1, self.kappa, st)
        self.wgrad(t("kin1"), E, dqkvg1, 4 * E, R, E, 4 * E, gv["dec.block0.retn1.w_qkvg"])
        dkin1 = g("dkin")
        self.lin(dqkvg1, 4 * E, v["dec.block0.retn1.w_qkvg"], None, dkin1, E, R, 4 * E, E)
        dza = g("dz")
        L.call("magpo_embed_bwd", 1, t("za"), E, dsum1, E, dkin1, E, None, 0, v["dec.ln.scale"], dza, E, slab("a"), slab("w", 32 * E),
               K + 1, None, 0, 0, None, None, None, prev_idx, 1, R, st)
        self.reduce(slab("a"), gv["dec.ln.scale"])
        self.reduce(slab("w", 32 * E), gv["dec.act.kernel"], P=(K + 1) * E, stride=32 * E)
        # ---- value head
        dhv = g("dhv")
        L.call("magpo_headmid_bwd", t("hv"), E, v["enc.head.norm.scale"], None, 0, v["enc.head.dense1.kernel"], dvalue, 1, dhv, E,
               slab("a"), slab("b"), slab("c", 1), R, st)
        self.reduce(slab("a"), gv["enc.head.norm.scale"]); self.reduce(slab("b"), gv["enc.head.dense1.kernel"])
        self.reduce(slab("c", 1), gv["enc.head.dense1.bias"], P=1, stride=1)
        self.wgrad(t("rep"), E, dhv, E, R, E, E, gv["enc.head.dense0.kernel"], gv["enc.head.dense0.bias"])
        drep_v = g("dout")
        self.lin(dhv, E, v["enc.head.dense0.kernel"], None, drep_v, E, R, E, E)
        # ---- encoder block: rep = rms(rms(xn + y) * ln1) * ln2 ; d(rep) = value head + cross-retention query + decoder residual
        dsum0 = g("dsum0")
        L.call("magpo_resnorm_bwd", t("xn"), E, t("y"), E, v["enc.block0.ln1.scale"], v["enc.block0.ln2.scale"], drep_v, E, dreppe, E,
               dsum2, E, dsum0, E, slab("a"), slab("b"), R, st)
        self.reduce(slab("a"), gv["enc.block0.ln1.scale"]); self.reduce(slab("b"), gv["enc.block0.ln2.scale"])
        self.wgrad(t("u"), E, dsum0, E, R, E, E, gv["enc.block0.retn.w_o"])
        du = g("du")
        self.lin(dsum0, E, v["enc.block0.retn.w_o"], None, du, E, R, E, E)
        dr = g("dr"); dqkvg = g("dqkvg", 4 * E)
        qkvg = t("qkvg")
        L.call("magpo_retpost_bwd", t("r"), E, qkvg[:, 3 * E:], 4 * E, v["enc.block0.retn.gn.scale"], v["enc.block0.retn.gn.bias"], du, E,
               dr, E, dqkvg[:, 3 * E:], 4 * E, slab("a"), slab("b"), R, st)
        self.reduce(slab("a"), gv["enc.block0.retn.gn.scale"]); self.reduce(slab("b"), gv["enc.block0.retn.gn.bias"])
        L.call("magpo_retention_chunk_bwd", qkvg, 4 * E, qkvg[:, E:], 4 * E, qkvg[:, 2 * E:], 4 * E, dr, E, dqkvg, 4 * E, dqkvg[:, E:], 4 * E,
               dqkvg[:, 2 * E:], 4 * E, dones, b.t["t_st_e"], nseq, T, A, 0, self.kappa, st)
        self.wgrad(t("kin"), E, dqkvg, 4 * E, R, E, 4 * E, gv["enc.block0.retn.w_qkvg"])
        dkin = g("dkin")
        self.lin(dqkvg, 4 * E, v["enc.block0.retn.w_qkvg"], None, dkin, E, R, 4 * E, E)
        dz = g("dz")
        L.call("magpo_embed_bwd", 0, t("z"), E, dsum0, E, dkin, E, None, 0, v["enc.ln.scale"], dz, E, slab("a"), slab("w", 32 * E), F,
               obs, F, F, v["enc.obs.norm.scale"], v["enc.obs.dense.kernel"], slab("d", 32), None, 0, R, st)
        self.reduce(slab("a"), gv["enc.ln.scale"])
        self.reduce(slab("d", 32), gv["enc.obs.norm.scale"], P=F, stride=32)
        self.reduce(slab("w", 32 * E), gv["enc.obs.dense.kernel"], P=F * E, stride=32 * E)
        if self.overlap_wgrad and self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
