"""Decentralised GRU actor (RecurrentActor, mava/networks/base.py:152-184) on the MI355X kernels.

pre-torso Dense(F->128)+ReLU -> scanned GRU(128) with done-resets -> post-torso Dense(128->128)+ReLU
-> Dense(128->K) logits (masked categorical head, heads.py:26-63).  ``step`` pushes the carry during
the rollout (rec_magpo.py:146-159) and serves the evaluator; ``seq_fwd`` / ``seq_bwd`` are the
training forward and the hand-derived BPTT backward.
"""
from __future__ import annotations

from typing import Dict, Optional

import contextlib

import numpy as np
import torch

from ._lib import lib
from .params import FlatParams, actor_layout, actor_named_views, init_actor
from .sable import _Bufs
from .tuning import Tuning

H = 128


class GruActor:
    def __init__(self, n_agents: int, action_dim: int, obs_dim: int, device, *, hidden: int = 128, wgrad_groups: int = 512,
                 seed: Optional[int] = None, grads: Optional[torch.Tensor] = None, tuning: Optional[Tuning] = None, obs_ld: Optional[int] = None):
        self.tuning = tuning if tuning is not None else Tuning.from_env()   # per-call kernel knobs (tuning.py); the library keeps none
        if hidden != 128:
            raise NotImplementedError("gfx950 GRU kernels: hidden_state_dim = 128 only")
        if obs_dim > 128 or action_dim > 32:
            raise NotImplementedError("obs_dim <= 128 and action_dim <= 32 required")
        self.A, self.K, self.F = n_agents, action_dim, obs_dim
        self.wide = obs_dim > 32            # wide observations: rows padded to 128, pre-torso on the MFMA dense kernel (csrc/wideobs.hip)
        self.Fld = 128 if self.wide else obs_dim      # floats between observation rows
        if obs_ld is not None and int(obs_ld) != self.Fld:   # rows wider than the features read (system.add_agent_id: False, learner.net_obs)
            if self.wide or int(obs_ld) < obs_dim:
                raise ValueError(f"obs_ld={obs_ld} with obs_dim={obs_dim}: a separate row stride is supported for narrow observations only")
            self.Fld = int(obs_ld)
        self.dev = device
        self.L = lib()
        self.G = wgrad_groups
        self.P = FlatParams(actor_layout(obs_dim, H, action_dim), device)
        self.grads = torch.zeros_like(self.P.flat) if grads is None else grads
        assert self.grads.numel() == self.P.numel
        self.v = self.P.views()
        self.gv = self.P.views(self.grads)
        self.named = actor_named_views(self.v)
        self.named_grads = actor_named_views(self.gv)
        if isinstance(seed, np.ndarray):   # a PRNG key: the parameters flax creates from it (rec_magpo.py:623; params.init_actor_from_key)
            from .params import init_actor_from_key
            init_actor_from_key(self.named, seed)
        elif seed is not None:
            init_actor(self.named, seed)
        self.wt: Dict[str, torch.Tensor] = {}
        self.b = _Bufs(device)
        self.wgrad_stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self.overlap_wgrad = False  # opt-in (bench.py --overlap): ~0.5 %, but per-kernel timings then include contention
        self.wg_ws = torch.empty(self.L.call("magpo_wgrad_workspace_floats", H, 3 * H, self.G), device=device)
        self.refresh()

    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def bind_grads(self, grads: torch.Tensor) -> None:
        """Make ``grads`` (flat, P.numel floats, e.g. a slice of the learner's all-reduce message) the gradient buffer."""
        assert grads.numel() == self.P.numel
        self.grads = grads
        self.gv = self.P.views(self.grads)
        self.named_grads = actor_named_views(self.gv)

    def load_named(self, params):
        with torch.no_grad():
            for n, v in self.named.items():
                v.copy_(params[n].to(self.dev, torch.float32).reshape(v.shape))
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
        v = self.v
        if self.wide:   # W_pre [F, 128] as [128][128] with zero columns beyond F
            if "pre" not in self.wt:
                self.wt["pre"] = torch.zeros(H, 128, device=self.dev)
            self.wt["pre"][:, :self.F].copy_(v["pre.kernel"].t())
        self._tp("wi", v["gru.wi"]); self._tp("wh", v["gru.wh"]); self._tp("post", v["post.kernel"])
        ht = self._tp("head", v["head.kernel"], 64)      # [64][128]
        self._tp("head_nat_pad", ht, H)                   # [128][64]
        # W_i with its gate columns in the order of the backward scan's gradient matrix (n | r | z), see seq_bwd
        if "wi_nrz" not in self.wt:
            self.wt["wi_nrz"] = torch.empty(H, 3 * H, device=self.dev)
        self._cols_nrz(v["gru.wi"], self.wt["wi_nrz"], H, inverse=True)

    def _cols_nrz(self, src, dst, rows, inverse=False):
        """[rows, 3H] matrices, gate column blocks (n | r | z) -> (r | z | n) (inverse: the other way), on the current stream."""
        L, st = self.L, self._st()
        if inverse:
            L.call("magpo_copy_rows", src[:, 2 * H:], 3 * H, dst, 3 * H, rows, H, st)
            L.call("magpo_copy_rows", src, 3 * H, dst[:, H:], 3 * H, rows, 2 * H, st)
        else:
            L.call("magpo_copy_rows", src, 3 * H, dst[:, 2 * H:], 3 * H, rows, H, st)
            L.call("magpo_copy_rows", src[:, H:], 3 * H, dst, 3 * H, rows, 2 * H, st)

    def lin(self, X, ldx, Wt, bias, Y, ldy, R, KIN, NOUT, act=0, Ypre=None):
        self.L.call("magpo_linear", X, ldx, Wt, bias, Y, ldy, Ypre, R, KIN, NOUT, act, self.tuning.actor_linear_variant, self._st())

    def pre_torso(self, obs, emb, R):
        """emb = relu(obs W_pre + b) (MLPTorso, torsos.py:36-47) for R observation rows (stride self.Fld)."""
        if self.wide:
            self.lin(obs, 128, self.wt["pre"], self.v["pre.bias"], emb, H, R, 128, H, act=1)
        else:
            self.L.call("magpo_small_linear", obs, self.Fld, self.F, self.v["pre.kernel"], self.v["pre.bias"], emb, H, H, R, 1, self._st())

    def _groups(self, R):
        """Row slabs of a split weight gradient: no more than one per 256 rows (small minibatches: fewer partials to reduce)."""
        return max(1, min(self.G, R // 256))

    def wgrad(self, X, ldx, dY, ldy, R, KIN, NOUT, dW, db=None, krows=None):
        """dW = X^T dY, queued on the side stream (off the critical path of the backward chain)."""
        side = self.wgrad_stream if self.overlap_wgrad else None
        if side is None:
            self.L.call("magpo_wgrad", X, ldx, dY, ldy, R, KIN, krows or KIN, NOUT, dW, db, self.wg_ws, self._groups(R), 1.0, 0, self.tuning.wgrad_variant, self._st())
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.L.call("magpo_wgrad", X, ldx, dY, ldy, R, KIN, krows or KIN, NOUT, dW, db, self.wg_ws, self._groups(R), 1.0, 0, self.tuning.wgrad_variant, self._st())

    def _wgrad_nrz(self, emb, dg, R, gw, dW, db):
        """dW_i, db_i from dg's columns 0..3H (gate blocks n | r | z) into W_i's order, on the weight-gradient stream."""
        side = self.wgrad_stream if self.overlap_wgrad else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            st = self._st()
            self.L.call("magpo_wgrad", emb, H, dg, 4 * H, R, H, H, 3 * H, gw[:H], gw[H], self.wg_ws, self._groups(R), 1.0, 0, self.tuning.wgrad_variant, st)
            self._cols_nrz(gw[:H], dW, H)
            self._cols_nrz(gw[H:], db.view(1, 3 * H), 1)

    # one step for N envs: returns new hidden [N*A,128]; logits [N*A,64] if want_logits
    def step(self, obs, h_in, reset_env, h_out, want_logits: bool = False):
        """obs [N,A,F] f32, h_in / h_out [N*A,128], reset_env [N] u8 (reset-before-step flag per env)."""
        L, st, A, F, v, b = self.L, self._st(), self.A, self.F, self.v, self.b
        N = obs.shape[0]
        R = N * A
        emb = b.get("s_emb", (R, H)); xi = b.get("s_xi", (R, 3 * H))
        self.pre_torso(obs, emb, R)
        self.lin(emb, H, self.wt["wi"], v["gru.bi"], xi, 3 * H, R, H, 3 * H)
        L.call("magpo_gru_scan_fwd", xi, self.wt["wh"], v["gru.hn.bias"], h_in, None, reset_env, h_out, None, None, N, 1, A, None, 0, self.tuning.gru_block_rows, st)
        if not want_logits:
            return None
        y = b.get("s_y", (R, H)); logits = b.get("s_logits", (R, 64), zero=True)
        self.lin(h_out, H, self.wt["post"], v["post.bias"], y, H, R, H, H, act=1)
        self.lin(y, H, self.wt["head"], v["head.bias"], logits, 64, R, H, self.K)
        return logits

    def carry(self, obs_tm, h_in, reset_tm, h_out, classes=None, tag=""):
        """Hidden-state carry over a whole rollout at once: obs_tm [T,N,A,F] (time-major trajectory), reset_tm [T,N] u8
        reset-before-step flags, h_in / h_out [N*A,128].  Same result as T calls of :meth:`step` (ScannedRNN, base.py:121-149):
        the carry depends on (obs, done) only, never on the sampled actions.  ``classes`` = (obs_tab [C,F], cls [T*N*A] i32):
        the input side on the distinct rows only (csrc/classtab.hip)."""
        L, st, A, F, v, b = self.L, self._st(), self.A, self.F, self.v, self.b
        T, N = obs_tm.shape[0], obs_tm.shape[1]
        R = T * N * A
        if classes is not None:
            _, xi_tab = self.input_table(classes[0], "r" + tag)
            L.call("magpo_gru_carry", xi_tab, self.wt["wh"], v["gru.hn.bias"], h_in, reset_tm, h_out, N, T, A, classes[1], self.tuning.gru_block_rows, st)
            return
        emb = b.get("c_emb", (R, H)); xi = b.get("c_xi", (R, 3 * H))
        self.pre_torso(obs_tm, emb, R)
        self.lin(emb, H, self.wt["wi"], v["gru.bi"], xi, 3 * H, R, H, 3 * H)
        L.call("magpo_gru_carry", xi, self.wt["wh"], v["gru.hn.bias"], h_in, reset_tm, h_out, N, T, A, None, self.tuning.gru_block_rows, st)

    def input_table(self, obs_tab: torch.Tensor, tag: str = ""):
        """xi of every distinct observation row: pre-torso + GRU input projection on obs_tab [C,F] -> [C,384] (the rows of a
        minibatch then take their xi by class index; see csrc/classtab.hip)."""
        L, st, F, v, b = self.L, self._st(), self.F, self.v, self.b
        C = obs_tab.shape[0]
        emb_tab = b.get("c_embtab" + tag, (C, H)); xi_tab = b.get("c_xitab" + tag, (C, 3 * H))
        L.call("magpo_small_linear", obs_tab, F, F, v["pre.kernel"], v["pre.bias"], emb_tab, H, H, C, 1, st)
        self.lin(emb_tab, H, self.wt["wi"], v["gru.bi"], xi_tab, 3 * H, C, H, 3 * H)
        return emb_tab, xi_tab

    def seq_fwd(self, obs, dones, h0, h0_idx, nseq: int, T: int, classes=None):
        """obs [R,F] rows (seq, t, agent); dones [nseq,T] u8 resets; h0 [*,128] gathered through h0_idx [nseq*A].
        ``classes`` (optional) = (obs_tab [C,F], cls [R] i32, order [R] i64, offsets [C+1] i64): the rows' observations are
        obs_tab[cls]; the input side of the GRU is then evaluated on the C distinct rows only.
        Returns raw logits [R,64] (K valid columns)."""
        L, st, A, F, v, b = self.L, self._st(), self.A, self.F, self.v, self.b
        R = nseq * T * A
        hs = b.get("t_hs", (R, H))
        gates = b.get("t_gates", (R, 4 * H)); hprev = b.get("t_hprev", (R, H)); y = b.get("t_y", (R, H))
        logits = b.get("t_logits", (R, 64), zero=True)
        self._saved = dict(obs=obs, dones=dones, nseq=nseq, T=T, R=R, classes=classes)
        if classes is not None:   # the scan reads xi rows straight from the (L2-resident) class table
            _, xi = self.input_table(classes[0])
            xi_cls = classes[1]
        else:
            emb = b.get("t_emb", (R, H)); xi = b.get("t_xi", (R, 3 * H))
            xi_cls = None
            self.pre_torso(obs, emb, R)
            self.lin(emb, H, self.wt["wi"], v["gru.bi"], xi, 3 * H, R, H, 3 * H)
        L.call("magpo_gru_scan_fwd", xi, self.wt["wh"], v["gru.hn.bias"], h0, h0_idx, dones, hs, gates, hprev, nseq, T, A, xi_cls, self.tuning.gru_split_bf16, self.tuning.gru_block_rows, st)
        self.lin(hs, H, self.wt["post"], v["post.bias"], y, H, R, H, H, act=1)
        self.lin(y, H, self.wt["head"], v["head.bias"], logits, 64, R, H, self.K)
        return logits

    def seq_bwd(self, dlogits):
        """dlogits [R,64] (columns >= K zero); fills self.grads."""
        L, st, A, F, K, v, gv, b = self.L, self._st(), self.A, self.F, self.K, self.v, self.gv, self.b
        sv = self._saved
        R, nseq, T, obs, dones = sv["R"], sv["nseq"], sv["T"], sv["obs"], sv["dones"]
        t = lambda n: b.t["t_" + n]
        self.wgrad(t("y"), H, dlogits, 64, R, H, K, gv["head.kernel"], gv["head.bias"])
        dy = b.get("g_dy", (R, H))
        # dy = (dlogits @ W_head^T) masked by the forward ReLU (fused epilogue: act 4 takes the mask in the Ypre slot)
        self.lin(dlogits, 64, self.wt["head_nat_pad"], None, dy, H, R, 64, H, act=4, Ypre=t("y"))
        self.wgrad(t("hs"), H, dy, H, R, H, H, gv["post.kernel"], gv["post.bias"])
        dhs = b.get("g_dhs", (R, H))
        self.lin(dy, H, v["post.kernel"], None, dhs, H, R, H, H)
        # one gradient matrix for both projections (include/magpo.h): dg = (dn_in | dr | dz | dn_hid); the hidden side is its columns
        # H..4H in W_h's own gate order, the input side its columns 0..3H in the order (n | r | z)
        dg = b.get("g_dg", (R, 4 * H))
        nblk = (nseq * A + 63) // 64
        slab = b.get("g_slab", (nblk, H))
        L.call("magpo_gru_scan_bwd", t("gates"), t("hprev"), dones, dhs, v["gru.wh"], dg, slab, nseq, T, A, self.tuning.gru_split_bf16, self.tuning.gru_block_rows, st)
        L.call("magpo_reduce_slabs", slab, gv["gru.hn.bias"], nblk, H, H, 1.0, 0, st)
        self.wgrad(t("hprev"), H, dg[:, H:], 4 * H, R, H, 3 * H, gv["gru.wh"])
        if sv["classes"] is not None:
            # input side on the class table: S[c] = sum of the rows of class c, gate blocks back into W_i's order on the C rows, then the
            # layers' backward on C rows
            obs, _, order, offsets = sv["classes"]
            C = obs.shape[0]
            part = b.get("g_cpart", (L.call("magpo_class_sum_slots", C), C, 3 * H))
            s_nrz = b.get("g_dxic_nrz", (C, 3 * H)); dxi = b.get("g_dxic", (C, 3 * H))
            L.call("magpo_class_sum", dg, 4 * H, order, offsets, C, 3 * H, part, s_nrz, st)
            self._cols_nrz(s_nrz, dxi, C)
            emb, R = b.t["c_embtab"], C
            self.wgrad(emb, H, dxi, 3 * H, R, H, 3 * H, gv["gru.wi"], gv["gru.bi"])
            ldx, wi_t = 3 * H, v["gru.wi"]
        else:
            # per token row: weight gradient with its gate blocks in dg's order, put back into W_i's order on the [H, 3H] result; the dX
            # GEMM contracts over the gates in dg's order against the equally permuted copy of W_i
            emb = t("emb")
            gw = b.get("g_gwi_nrz", (H + 1, 3 * H))
            self._wgrad_nrz(emb, dg, R, gw, gv["gru.wi"], gv["gru.bi"])
            dxi, ldx, wi_t = dg, 4 * H, self.wt["wi_nrz"]
        demb = b.get("g_demb", (R, H))
        if self.wide:   # ReLU backward fused into the dX GEMM (act 4 takes the mask), then dW_pre = obs^T demb on the dense kernel
            self.lin(dxi, ldx, wi_t, None, demb, H, R, 3 * H, H, act=4, Ypre=emb)
            self.wgrad(obs, 128, demb, H, R, 128, H, gv["pre.kernel"], gv["pre.bias"], krows=F)
            if self.overlap_wgrad and self.wgrad_stream is not None:
                torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            return
        self.lin(dxi, ldx, wi_t, None, demb, H, R, 3 * H, H)
        grid = L.call("magpo_row_grid", R)
        sw = b.get("g_slabw", (grid, 33 * H))
        L.call("magpo_small_relu_wgrad", obs, self.Fld, F, emb, demb, sw, R, st)
        L.call("magpo_reduce_slabs", sw, gv["pre.kernel"], grid, F * H, 33 * H, 1.0, 0, st)
        L.call("magpo_reduce_slabs", sw[:, 32 * H:], gv["pre.bias"], grid, H, 33 * H, 1.0, 0, st)
        if self.overlap_wgrad and self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)


GruActor.apply = GruActor.seq_fwd   # actor_network.apply (rec_magpo.py:631): the scanned training forward (its backward: seq_bwd)
