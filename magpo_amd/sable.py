"""Sable guider on the MI355X kernels: recurrent acting and chunkwise training forward/backward.

Host-side composition of the C-ABI kernels (include/magpo.h); mirrors ``SableNetwork.get_actions``
(mava/networks/sable_network.py:443-482) and ``SableNetwork.__call__`` (:412-441).  The backward pass
is hand-derived (no autograd): it walks the forward graph in reverse, one kernel per node.

Supported configuration (asserted): embed_dim 16 / 32 / 64 / 128, n_head 1 / 2 / 4, any n_block, discrete actions, one chunk
per rollout, SwiGLU weights at their zero init (then the FFN branch and its gradients are exactly 0,
SURVEY B5 -- verified on the host at construction / load time).

Widths.  ``EL`` = the reference's embed_dim; ``E`` = width of the device network: 64 (EL <= 64: narrower nets run embedded, see
params.WidthEmbedding) or 128.  The fused kernels (csrc/seg_fused.hip, csrc/act_fused.hip, first-layer class tables read in place)
exist for E = 64; an E = 128 network is composed kernel by kernel: dense layers on k_linear_lds<128 | 256 | 384,*>, row kernels at 32
lanes per row (csrc/rowops.hip), retention on the 64-wide tile kernels -- n_head 2 / 4 have 64- / 32-wide heads (real tiles); the
one 128-wide head of n_head = 1 is evaluated blockwise, r[:, J] = sum_I ret(q[:, I], k[:, I], v[:, J]) over the 64-column halves
I, J with the 128 x 128 state kept as four 64 x 64 tiles S[I][J] (exact: retention is bilinear in (q.k) and v).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ._lib import lib
from .params import FlatParams, WidthEmbedding, guider_layout, guider_named_views, init_guider
from .tuning import Tuning

E = 64      # width of the fused (64-wide) kernel family, of a retention state tile and of a logit row
TILE = 64   # retention states live in 64 x 64 tiles on the device
LW = 64     # logit rows (K <= 31 valid columns)


def decay_kappas(n_head: int, scaling: float):
    """sable_network.py:366-369 / retention.py:231-234 in float32: one kappa per head."""
    k = 1.0 - np.exp(np.linspace(np.log(np.float32(1 / 32)), np.log(np.float32(1 / 512)), n_head, dtype=np.float32))
    return [float(x) for x in (k.astype(np.float32) * np.float32(scaling)).astype(np.float32)]


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
                 wgrad_groups: int = 512, seed: Optional[int] = None, grads: Optional[torch.Tensor] = None,
                 tuning: Optional[Tuning] = None, obs_ld: Optional[int] = None):
        self.tuning = tuning if tuning is not None else Tuning.from_env()   # per-call kernel knobs (tuning.py); the library keeps none
        if embed_dim not in (16, 32, 64, 128) or n_head not in (1, 2, 4) or n_block < 1 or embed_dim % n_head or 64 // n_head // n_head < 4:
            raise NotImplementedError("gfx950 Sable kernels: embed_dim in {16, 32, 64, 128} (nets narrower than 64 run embedded in the 64-wide "
                                      "kernels, params.WidthEmbedding) and n_head in {1, 2, 4} (SURVEY 8f rank 3)")
        self.nb, self.nh = int(n_block), int(n_head)
        self.EL = int(embed_dim)        # logical embed_dim (what the optimiser, checkpoints and the reference see)
        self.E = E = 128 if self.EL == 128 else 64   # width of the device network
        self.hs = E // self.nh          # head width of the device network
        self.gs = self.hs // self.nh    # flax GroupNorm(num_groups=n_head) on (token*head, hs) rows: hs / n_head channels per group
        # retention state tiles [n_block, ntile, N, 64, 64]: one per head, or the four 64 x 64 blocks (I, J) of the one 128-wide head
        self.blockwise = self.hs > TILE
        self.ntile = 4 if self.blockwise else self.nh
        if obs_dim > 128 or action_dim > 31:
            raise NotImplementedError("obs_dim <= 128 and action_dim <= 31 required")
        self.A, self.K, self.F = n_agents, action_dim, obs_dim
        # observation rows: F floats apart for small observations (row kernels), padded to 128 for wide ones (obs_dim > 32, e.g.
        # Robot Warehouse: the observation-side first layer then runs on the MFMA dense kernels, csrc/wideobs.hip)
        self.wide = obs_dim > 32
        self.Fld = 128 if self.wide else obs_dim      # floats between observation rows
        if obs_ld is not None and int(obs_ld) != self.Fld:   # rows wider than the features read (system.add_agent_id: False, learner.net_obs)
            if self.wide or int(obs_ld) < obs_dim:
                raise ValueError(f"obs_ld={obs_ld} with obs_dim={obs_dim}: a separate row stride is supported for narrow observations only")
            self.Fld = int(obs_ld)
        self.dev = device
        self.L = lib()
        self.kappas = decay_kappas(self.nh, decay_scaling_factor)
        self.kappa = self.kappas[0]
        self.G = wgrad_groups
        # P / grads: the LOGICAL parameters (optimiser, all-reduce, checkpoints); PD / grads_D: what the kernels read and write.
        # For embed_dim = 64 they are the same buffers.
        self.P = FlatParams(guider_layout(self.EL, obs_dim, action_dim, self.nb, self.nh), device)
        self.grads = torch.zeros_like(self.P.flat) if grads is None else grads
        assert self.grads.numel() == self.P.numel
        if self.EL == E:
            self.emb, self.PD, self.grads_D = None, self.P, self.grads
        else:
            self.emb = WidthEmbedding(self.EL, obs_dim, action_dim, self.nb, self.nh, device)
            self.PD = FlatParams(guider_layout(E, obs_dim, action_dim, self.nb, self.nh), device)
            self.grads_D = torch.zeros_like(self.PD.flat)
        self.v = self.PD.views()
        self.gv = self.PD.views(self.grads_D)
        self.named = guider_named_views(self.P.views(), self.EL, self.nh)
        self.named_grads = guider_named_views(self.P.views(self.grads), self.EL, self.nh)
        if isinstance(seed, np.ndarray):   # a PRNG key: the parameters flax creates from it (rec_magpo.py:598-604; params.init_guider_from_key)
            from .params import init_guider_from_key
            init_guider_from_key(self.named, seed, self.EL, self.nh)
        elif seed is not None:
            init_guider(self.named, seed, self.EL)
        self.npos = max_pos
        self.pe = torch.zeros(max_pos, E, device=device)
        if use_pe:
            if self.emb is None:
                self.L.call("magpo_pe_table", self.pe, max_pos, E, self._st())
            else:   # positional_encoding.py:24-60 at the logical width, every feature duplicated like the activations
                pe_l = torch.zeros(max_pos, self.EL, device=device)
                self.L.call("magpo_pe_table", pe_l, max_pos, self.EL, self._st())
                self.pe.copy_(self.emb.expand_rows(pe_l))
        self.wt: Dict[str, torch.Tensor] = {}
        self.wa: Dict[str, torch.Tensor] = {}   # fragment-major copies for the fused acting kernel (build_act_weights)
        self._act_w_dirty = True
        self.b = _Bufs(device)
        self._act_tabs: Dict[tuple, tuple] = {}
        self._seg_tabs: Dict[tuple, tuple] = {}
        self.fused_segments = self.nh == 1 and E == 64   # token-local parts between retention ops as single launches (csrc/seg_fused.hip)
        # weight-gradient GEMMs run on a side stream: they are off the critical path of the backward chain
        self.wgrad_stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self.overlap_wgrad = False  # opt-in (bench.py --overlap): ~0.5 %, but per-kernel timings then include contention
        self.wg_ws = torch.empty(self.L.call("magpo_wgrad_workspace_floats", E, 4 * E, self.G), device=device)
        self.refresh()

    # ------------------------------------------------------------------ plumbing
    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def bind_grads(self, grads: torch.Tensor) -> None:
        """Make ``grads`` (a flat fp32 buffer of P.numel floats, e.g. a slice of the learner's all-reduce message) the gradient buffer."""
        assert grads.numel() == self.P.numel
        self.grads = grads
        if self.emb is None:
            self.grads_D = grads
        self.gv = self.PD.views(self.grads_D)
        self.named_grads = guider_named_views(self.P.views(self.grads), self.EL, self.nh)

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
        """Rebuild the device parameters (embed_dim < 64) and the transposed (forward-GEMM) weight copies after a parameter update."""
        if self.emb is not None:
            self.emb.expand(self.P.flat, self.PD.flat)
        v, E = self.v, self.E
        if self.wide:   # W_obs [F, 64] as [64][128] (forward) and [128][64] (dOn = dz W_obs^T), zero beyond F
            for name, shape in (("wobs", (E, 128)), ("wobs_nat_pad", (128, E))):
                if name not in self.wt:
                    self.wt[name] = torch.zeros(*shape, device=self.dev)
            self.wt["wobs"][:, :self.F].copy_(v["enc.obs.dense.kernel"].t())
            self.wt["wobs_nat_pad"][:self.F].copy_(v["enc.obs.dense.kernel"])
        self._tp("vh0", v["enc.head.dense0.kernel"])
        self._tp("h0", v["dec.head.dense0.kernel"])
        for b in range(self.nb):
            e, d = f"enc.block{b}.", f"dec.block{b}."
            for key, name in [(f"qkvg{b}", e + "retn.w_qkvg"), (f"wo{b}", e + "retn.w_o"), (f"qkvg1{b}", d + "retn1.w_qkvg"),
                              (f"wo1{b}", d + "retn1.w_o"), (f"q2{b}", d + "retn2.w_q"), (f"kvg2{b}", d + "retn2.w_kvg"),
                              (f"wo2{b}", d + "retn2.w_o")]:
                self._tp(key, v[name])
        h1t = self._tp("h1", v["dec.head.dense1.kernel"], LW)        # [64 (K padded)][E]
        self._tp("h1_nat_pad", h1t, E)                                # [E][64]: natural W padded to 64 columns
        if E > 64:   # dX = dY W^T with KIN = 4 E = 512 runs as two K-halves (k_linear_lds<256,*>): contiguous copies of the row halves
            for bk in range(self.nb):
                for name in (f"enc.block{bk}.retn.w_qkvg", f"dec.block{bk}.retn1.w_qkvg"):
                    W = v[name]
                    for half in (0, 1):
                        key = f"{name}.nat{half}"
                        if key not in self.wt:
                            self.wt[key] = torch.empty(E, 2 * E, device=self.dev)
                        self.wt[key].copy_(W[:, half * 2 * E:(half + 1) * 2 * E])
        self._act_w_dirty = True   # the acting kernel's fragment-major copies are rebuilt before the next acting step (build_act_weights)

    def _act_weight_names(self):
        names = ["vh0", "h0", "h1"]
        for b in range(self.nb):
            names += [f"qkvg{b}", f"wo{b}", f"qkvg1{b}", f"wo1{b}", f"q2{b}", f"kvg2{b}", f"wo2{b}"]
        return names

    def build_act_weights(self):
        """Fragment-major copies of the transposed weights for the fused acting kernel (csrc/fm_rows.hpp: wfrag<true>):
        Wf[g][gk][lane = m + 16 kq][4] = Wt[16 g + m][16 gk + 4 kq .. + 3], so that a wave's weight-fragment load is 1 KB contiguous.
        In-place copies into persistent buffers (static pointers: safe inside a HIP-graph capture); the rollout calls this at its start
        (MagpoLearner._rollout_body), a stand-alone ``act_fused`` when the parameters changed since the last build."""
        if self.E != 64:
            return
        for name in self._act_weight_names():
            t = self.wt[name]
            d = self.wa.get(name)
            if d is None:
                d = self.wa[name] = torch.empty_like(t)
            self.L.call("magpo_act_weight_layout", t, d, t.shape[0], self._st())
        self._act_w_dirty = False

    def lin(self, X, ldx, Wt, bias, Y, ldy, R, KIN, NOUT, act=0, Ypre=None):
        self.L.call("magpo_linear", X, ldx, Wt, bias, Y, ldy, Ypre, R, KIN, NOUT, act, self.tuning.linear_variant, self._st())

    def _groups(self, R):
        """Row slabs of a split weight gradient: no more than one per 256 rows (small minibatches: fewer partials to reduce)."""
        return max(1, min(self.G, R // 256))

    def wgrad(self, X, ldx, dY, ldy, R, KIN, NOUT, dW, db=None, krows=None):
        """dW = X^T dY.  With overlap_wgrad the GEMM is queued on the side stream behind everything the calling stream
        has queued so far (so X and dY are complete); the caller must not overwrite dY before train_bwd joins."""
        # (with n_block > 1 the d(obs_rep) sums are accumulated in place, so the side stream is not used)
        side = self.wgrad_stream if (self.overlap_wgrad and self.nb == 1) else None
        if side is None:
            self.L.call("magpo_wgrad", X, ldx, dY, ldy, R, KIN, krows or KIN, NOUT, dW, db, self.wg_ws, self._groups(R), 1.0, 0, self.tuning.wgrad_variant, self._st())
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.L.call("magpo_wgrad", X, ldx, dY, ldy, R, KIN, krows or KIN, NOUT, dW, db, self.wg_ws, self._groups(R), 1.0, 0, self.tuning.wgrad_variant, self._st())

    def reduce(self, slab, out, P=None, stride=None, accumulate=False):
        self.L.call("magpo_reduce_slabs", slab, out, slab.shape[0], P or slab.shape[1], stride or slab.shape[1], 1.0, 1 if accumulate else 0, self._st())

    def lin_dx_qkvg(self, dY, name, dX, R):
        """dX [R, E] = dY [R, 4E] W^T for a fused q|k|v|g projection ``name`` ([E, 4E]): one launch up to KIN = 256, two K-halves + add for E = 128."""
        E = self.E
        if E == 64:
            self.lin(dY, 4 * E, self.v[name], None, dX, E, R, 4 * E, E)
            return
        tmp = self.b.get("g_dx_half", (R, E))
        self.lin(dY, 4 * E, self.wt[name + ".nat0"], None, dX, E, R, 2 * E, E)
        self.lin(dY[:, 2 * E:], 4 * E, self.wt[name + ".nat1"], None, tmp, E, R, 2 * E, E)
        self.add_(dX, tmp)

    def add_(self, dst, src):
        self.L.call("magpo_add_inplace", dst, src, dst.numel(), self._st())

    # ------------------------------------------------------------------ acting (recurrent form)
    def _pro(self, pro, a, lda, y, ldy_in, s1, s2, use_pe, pos, pos_stride, W, idx, idx_stride, out, ldout, outpe, ldoutpe,
             Wt, bias, Y, ldy, R, NOUT):
        """Dense layer behind a row-wise op: pro 1 embed-action, 2 embed-observation, 3 residual + norm(s), 4 gelu + norm; the dense input is
        the row (+ pe when ``use_pe``); ``out`` / ``outpe`` (optional) receive the row / the row + pe.  E = 64: one launch with the row op in
        the prologue of the GEMM (csrc/linear.hip: k_linear_pro); E = 128: row kernel + k_linear_lds<128,*>."""
        E = self.E
        if E == 64:
            self.L.call("magpo_linear_pro", pro, a, lda, y, ldy_in, s1, s2, self.pe, pos, pos_stride, self.npos, 1 if use_pe else 0,
                        W, idx, idx_stride, self.v["enc.obs.norm.scale"], self.F, out, ldout, outpe, ldoutpe, Wt, bias, Y, ldy, R, NOUT,
                        self._st())
            return
        L, st, b = self.L, self._st(), self.b
        if out is None:
            out, ldout = b.get("p_row", (R, E)), E
        need_pe = use_pe or outpe is not None or pro in (1, 2)
        if need_pe and outpe is None:
            outpe, ldoutpe = b.get("p_rowpe", (R, E)), E
        if pro == 1:
            L.call("magpo_embed_fwd", 1, None, 0, 0, None, W, idx, idx_stride, s1, self.pe, pos, pos_stride, self.npos, None, 0, out, ldout,
                   outpe, ldoutpe, R, E, st)
        elif pro == 2:
            L.call("magpo_embed_fwd", 0, a, lda, self.F, self.v["enc.obs.norm.scale"], W, None, 0, s1, self.pe, pos, pos_stride, self.npos, None, 0,
                   out, ldout, outpe, ldoutpe, R, E, st)
        elif pro == 3:
            L.call("magpo_resnorm_fwd", a, lda, y, ldy_in, s1, s2, self.pe, pos, pos_stride, self.npos, out, ldout, outpe if need_pe else None,
                   ldoutpe if need_pe else 0, R, E, st)
        else:
            L.call("magpo_headmid_fwd", a, lda, s1, out, ldout, None, None, None, 0, R, E, st)
        if use_pe:
            self.lin(outpe, ldoutpe, Wt, bias, Y, ldy, R, E, NOUT)
        else:
            self.lin(out, ldout, Wt, bias, Y, ldy, R, E, NOUT)

    # ------------------------------------------------------------------ retention helpers (per state tile)
    def _tiles(self):
        """State tiles of one retention layer as (tile, q / k column offset, v / r column offset, width, kappa): one per head, or -- one
        128-wide head -- the four 64 x 64 blocks S[I][J] (q, k columns of half I meet v, r columns of half J; outputs add over I)."""
        if self.blockwise:
            return [(2 * i + j, TILE * i, TILE * j, TILE, self.kappas[0]) for i in (0, 1) for j in (0, 1)]
        return [(h, h * self.hs, h * self.hs, self.hs, self.kappas[h]) for h in range(self.nh)]

    def add_rows(self, dst, ldd, src, lds, R, W):
        self.L.call("magpo_add_rows", dst, ldd, src, lds, R, W, self._st())

    def _ret_rec(self, S, q, ldq, k, ldk, v, ldv, env_rows, u, ldu, N, ntok, ret_from, write, gp, ldg, gamma, beta):
        """Recurrent retention + GroupNorm / gate for every state tile; S [ntile, N, 64, 64].  Rows: token a of env e at e * env_rows + a."""
        E = self.E
        if not self.blockwise:   # the kernel's fused epilogue covers a whole head
            for ti, oq, ov, w, kap in self._tiles():
                self.L.call("magpo_retention_recurrent", S[ti], q[:, oq:], ldq, k[:, oq:], ldk, v[:, ov:], ldv, env_rows, u[:, ov:], ldu, N, ntok,
                            ret_from, kap, write, gp[:, ov:], ldg, gamma, beta, w, self.gs, self._st())
            return
        R = u.shape[0]
        raw, tmp = self.b.get("rr_raw", (R, E)), self.b.get("rr_tmp", (R, TILE))
        for ti, oq, ov, w, kap in self._tiles():
            first = oq == 0
            out, ldo = (raw[:, ov:], E) if first else (tmp, TILE)
            self.L.call("magpo_retention_recurrent", S[ti], q[:, oq:], ldq, k[:, oq:], ldk, v[:, ov:], ldv, env_rows, out, ldo, N, ntok,
                        ret_from, kap, write, None, 0, None, None, w, w, self._st())
            if not first:
                self.add_rows(raw[:, ov:], E, tmp, TILE, R, TILE)   # (rows outside [ret_from, ntok) hold stale values: never read)
        for a in range(ret_from, ntok):   # GroupNorm + swish gate on the 128-wide rows of token a of every env
            self.L.call("magpo_retpost_fwd", raw[a:], env_rows * E, gp[a:], env_rows * ldg, gamma, beta, u[a:], env_rows * ldu, N, self.hs, self.gs,
                        E, self._st())

    def _ret_fwd(self, q, ldq, k, ldk, v, ldv, r, s0, seq_env, dones, name, nseq, T, masked, rows=None):
        """``rows`` (i32 [R], optional): q | k | v are row tables and token row r reads table row rows[r] (csrc/classtab.hip)."""
        ct = self.tuning.ret_chunk_tokens
        nch = self.L.call("magpo_retention_num_chunks", T, self.A, ct)
        E, R = self.E, nseq * T * self.A
        for ti, oq, ov, w, kap in self._tiles():
            stt = self.b.get(f"t_{name}_{ti}", (nseq, nch, TILE, TILE))
            partial = self.blockwise and oq != 0   # second q / k half of the 128-wide head: its contribution adds to the first one's
            out, ldo = (self.b.get("rf_tmp", (R, TILE)), TILE) if partial else (r[:, ov:], E)
            self.L.call("magpo_retention_chunk_fwd", q[:, oq:], ldq, k[:, oq:], ldk, v[:, ov:], ldv, out, ldo, s0[ti], seq_env, dones, stt,
                        None, nseq, T, self.A, masked, kap, w, rows, ct, self._st())
            if partial:
                self.add_rows(r[:, ov:], E, out, TILE, R, TILE)

    def _ret_bwd(self, q, ldq, k, ldk, v, ldv, dr, dq, lddq, dk, lddk, dv, lddv, dones, name, nseq, T, masked, rows=None):
        ct = self.tuning.ret_chunk_tokens
        E, R = self.E, nseq * T * self.A
        for ti, oq, ov, w, kap in self._tiles():
            stt = self.b.t[f"t_{name}_{ti}"]
            if not self.blockwise:
                self.L.call("magpo_retention_chunk_bwd", q[:, oq:], ldq, k[:, oq:], ldk, v[:, ov:], ldv, dr[:, ov:], E, dq[:, oq:], lddq, dk[:, oq:],
                            lddk, dv[:, ov:], lddv, dones, stt, nseq, T, self.A, masked, kap, w, rows, ct, self._st())
                continue
            # block (I, J) of the 128-wide head: dq_I, dk_I add over J; dv_J adds over I
            tq, tk, tv = (self.b.get(f"rb_t{c}", (R, TILE)) for c in "qkv")
            qk_first, v_first = ov == 0, oq == 0
            oq_, ldq_ = (dq[:, oq:], lddq) if qk_first else (tq, TILE)
            ok_, ldk_ = (dk[:, oq:], lddk) if qk_first else (tk, TILE)
            ov_, ldv_ = (dv[:, ov:], lddv) if v_first else (tv, TILE)
            self.L.call("magpo_retention_chunk_bwd", q[:, oq:], ldq, k[:, oq:], ldk, v[:, ov:], ldv, dr[:, ov:], E, oq_, ldq_, ok_, ldk_,
                        ov_, ldv_, dones, stt, nseq, T, self.A, masked, kap, w, rows, ct, self._st())
            if not qk_first:
                self.add_rows(dq[:, oq:], lddq, tq, TILE, R, TILE)
                self.add_rows(dk[:, oq:], lddk, tk, TILE, R, TILE)
            if not v_first:
                self.add_rows(dv[:, ov:], lddv, tv, TILE, R, TILE)

    def _retpost_fwd(self, r, gp, ldg, gamma, beta, u, R):
        E = self.E
        self.L.call("magpo_retpost_fwd", r, E, gp, ldg, gamma, beta, u, E, R, self.hs, self.gs, E, self._st())

    def _retpost_bwd(self, r, gp, ldg, pfx, du, dr, dgp, lddg, R, sa, sb):
        v, gv, E = self.v, self.gv, self.E
        self.L.call("magpo_retpost_bwd", r, E, gp, ldg, v[pfx + "gn.scale"], v[pfx + "gn.bias"], du, E, dr, E, dgp, lddg, sa, sb, R,
                    self.hs, self.gs, E, self._st())
        for h in range(self.nh):   # scale / bias [hs] are shared by the heads: fold the per-column slabs
            self.reduce(sa[:, h * self.hs:], gv[pfx + "gn.scale"], P=self.hs, stride=E, accumulate=h > 0)
            self.reduce(sb[:, h * self.hs:], gv[pfx + "gn.bias"], P=self.hs, stride=E, accumulate=h > 0)

    def act(self, obs, pos, states, sample_keys, action_out, logp_out, value_out, mask=None, value_only=False):
        """One env step for N envs (SableNetwork.get_actions, sable_network.py:443-482).  obs [N,A,F] f32, pos [N] i32
        (step_count), states = (S_enc, S_d1, S_d2) each [n_block, N, 64, 64] updated in place, sample_keys = [A,2]
        uint32 (host array: keys by value; device tensor: static arguments for graph replay).
        Writes action [N,A] i32, logp [N,A], value [N,A].  Row-wise ops are fused into the prologue of the dense layer
        that follows them and the GroupNorm + swish gate into the recurrent retention kernel."""
        L, st, A, K, F, nb, E = self.L, self._st(), self.A, self.K, self.F, self.nb, self.E
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
        for blk in range(nb):
            e = f"enc.block{blk}."
            if blk == 0:
                self._pro(2, obs, self.Fld, None, 0, v["enc.ln.scale"], None, True, pos_tok, 1, v["enc.obs.dense.kernel"], None, 0, xn, E, None, 0,
                          self.wt["qkvg0"], None, qkvg, 4 * E, R, 4 * E)
            else:  # x = ln(rep of the previous block) (shared self.ln, sable_network.py:150)
                self._pro(3, rep, E, None, 0, v["enc.ln.scale"], None, True, pos_tok, 1, None, None, 0, xn, E, None, 0,
                          self.wt[f"qkvg{blk}"], None, qkvg, 4 * E, R, 4 * E)
            self._ret_rec(s_enc[blk], qkvg, 4 * E, qkvg[:, E:], 4 * E, qkvg[:, 2 * E:], 4 * E, A, u, E, N, A, 0, 1,
                          qkvg[:, 3 * E:], 4 * E, v[e + "retn.gn.scale"], v[e + "retn.gn.bias"])
            self.lin(u, E, self.wt[f"wo{blk}"], None, y, E, R, E, E)
            if blk == nb - 1:
                self._pro(3, xn, E, y, E, v[e + "ln1.scale"], v[e + "ln2.scale"], False, pos_tok, 1, None, None, 0, rep, E, reppe, E,
                          self.wt["vh0"], v["enc.head.dense0.bias"], hv, E, R, E)
            else:
                L.call("magpo_resnorm_fwd", xn, E, y, E, v[e + "ln1.scale"], v[e + "ln2.scale"], None, None, 0, 0, rep, E, None, 0, R, E, st)
        L.call("magpo_headmid_fwd", hv, E, v["enc.head.norm.scale"], None, 0, v["enc.head.dense1.kernel"],
               v["enc.head.dense1.bias"], value_out, 1, R, E, st)
        if value_only:
            return
        # autoregressive decoder (decode.py:111-153): one token per env per iteration.  Per-agent projections stay
        # resident ([N, A, .]) so that a retention state is read once per agent and written once per step.
        prev = b.get("a_prev", (N, A), torch.int32, zero=True)
        xa = b.get("d_xa", (N, E)); y1 = b.get("d_y1", (N, E)); y2 = b.get("d_y2", (N, E)); xo = b.get("d_xo", (N, E))
        xope = b.get("d_xope", (N, E))
        hp = b.get("d_hp", (N, E)); logits = b.get("d_logits", (N, LW), zero=True)
        qkvg1 = [b.get(f"d_qkvg1_{k}", (R, 4 * E)) for k in range(nb)]
        q2 = [b.get(f"d_q2_{k}", (R, E)) for k in range(nb)]
        kvg2 = [b.get(f"d_kvg2_{k}", (R, 3 * E)) for k in range(nb)]
        u1 = b.get("d_u1", (R, E)); u2 = b.get("d_u2", (R, E))
        for blk in range(nb):   # cross-retention queries of all agents at once
            self.lin(reppe, E, self.wt[f"q2{blk}"], None, q2[blk], E, R, E, E)
        for i in range(A):
            last = 1 if i == A - 1 else 0
            for blk in range(nb):
                d = f"dec.block{blk}."
                if blk == 0:
                    self._pro(1, None, 0, None, 0, v["dec.ln.scale"], None, True, pos, 1, v["dec.act.kernel"], prev[:, i:], A, xa, E, None, 0,
                              self.wt["qkvg10"], None, qkvg1[0][i:], A * 4 * E, N, 4 * E)
                    xin = xa
                else:  # block input = previous block's output x (xo); key = query = value = x + pe (xope)
                    self.lin(xope, E, self.wt[f"qkvg1{blk}"], None, qkvg1[blk][i:], A * 4 * E, N, E, 4 * E)
                    xin = xo
                self._ret_rec(s_d1[blk], qkvg1[blk], 4 * E, qkvg1[blk][:, E:], 4 * E, qkvg1[blk][:, 2 * E:], 4 * E, A, u1, E, N, i + 1, i, last,
                              qkvg1[blk][:, 3 * E:], 4 * E, v[d + "retn1.gn.scale"], v[d + "retn1.gn.bias"])
                self.lin(u1[i:], A * E, self.wt[f"wo1{blk}"], None, y1, E, N, E, E)
                self._pro(3, xin, E, y1, E, v[d + "ln1.scale"], None, True, pos, 1, None, None, 0, None, 0, None, 0,
                          self.wt[f"kvg2{blk}"], None, kvg2[blk][i:], A * 3 * E, N, 3 * E)
                self._ret_rec(s_d2[blk], q2[blk], E, kvg2[blk], 3 * E, kvg2[blk][:, E:], 3 * E, A, u2, E, N, i + 1, i, last,
                              kvg2[blk][:, 2 * E:], 3 * E, v[d + "retn2.gn.scale"], v[d + "retn2.gn.bias"])
                self.lin(u2[i:], A * E, self.wt[f"wo2{blk}"], None, y2, E, N, E, E)
                if blk == nb - 1:
                    self._pro(3, rep[i:], A * E, y2, E, v[d + "ln2.scale"], v[d + "ln3.scale"], False, pos, 1, None, None, 0,
                              None, 0, None, 0, self.wt["h0"], v["dec.head.dense0.bias"], hp, E, N, E)
                else:
                    L.call("magpo_resnorm_fwd", rep[i:], A * E, y2, E, v[d + "ln2.scale"], v[d + "ln3.scale"], self.pe, pos, 1, self.npos,
                           xo, E, xope, E, N, E, st)
            self._pro(4, hp, E, None, 0, v["dec.head.norm.scale"], None, False, pos, 1, None, None, 0, None, 0, None, 0,
                      self.wt["h1"], v["dec.head.dense1.bias"], logits, LW, N, K)
            if torch.is_tensor(sample_keys):   # device key table [A, 2] (static arguments: HIP-graph replay)
                k0, k1, kdev = 0, 0, sample_keys[i]
            else:
                k0, k1, kdev = int(sample_keys[i][0]), int(sample_keys[i][1]), None
            L.call("magpo_sample_categorical", logits, LW, None if mask is None else mask[:, i], (A * K if mask is not None else 0),
                   k0, k1, kdev, action_out[:, i:], A, logp_out[:, i:], A, prev[:, i + 1:] if i + 1 < A else None, A, None, 0, N, K, st)

    def act_fused(self, obs, pos, states, sample_keys, action_out, logp_out, value_out, mask=None, value_only=False, done=None, tag="",
                  pending=False, flush=True, precand=False, defer=False):
        """Same contract as :meth:`act`, ONE launch per env step (csrc/act_fused_kernel.hpp: k_sable_act): a wave carries 4 - 16 envs
        through encoder, the A decoder iterations and the sampling.  states [n_block, n_head, N, 64, 64].  ``done`` [N] u8
        (optional): envs whose episode ended on the previous step -- their carried states are read as zero
        (rec_magpo.py:164-169), which replaces a separate zeroing pass between steps.
        ``pending`` / ``flush``: the kernel defers a step's decoder-state update to the next launch so that every state is read and
        written once per step (include/magpo.h).  The defaults {False, True} are a stand-alone step (states settled on return); a
        rollout passes pending = (t > 0), flush = False and ends with a launch that has flush = True (the bootstrap-value launch).
        The scratch rows that carry the pending k | v rows belong to ``tag``.
        ``defer`` / ``precand`` (include/magpo.h): inside a rollout half of the workgroups run the block-0 candidate pre-pass of the NEXT
        step at the end of the launch (defer) and skip it in the next one (precand), so that they stream states while the others decode."""
        A, K, F, nb, nh = self.A, self.K, self.F, self.nb, self.nh
        if A > 8 and self.wide:
            raise NotImplementedError("wide observations (obs_dim > 32) with more than 8 agents")
        if A > 8 or self.E != 64:   # token staging registers / 64-wide register rows of the fused kernel: larger teams take the kernel-by-kernel path (states settled on return)
            if done is not None:
                for k in range(nb):
                    for h in range(self.ntile):
                        self.L.call("magpo_zero_states_where_done", states[0][k][h], states[1][k][h], states[2][k][h], done, obs.shape[0], self._st())
            return self.act(obs, pos, states, sample_keys, action_out, logp_out, value_out, mask=mask, value_only=value_only)
        N = obs.shape[0]
        R = N * A
        v, b = self.v, self.b
        s_enc, s_d1, s_d2 = states   # value_only (bootstrap value, rec_magpo.py:202-208): the kernel writes no state
        kdev = sample_keys if torch.is_tensor(sample_keys) else None
        cache_key = (N, bool(value_only), bool(pending), bool(flush), bool(precand), bool(defer), self.tuning.act_envs_per_wave, obs.data_ptr(), pos.data_ptr(), None if mask is None else mask.data_ptr(),
                     None if kdev is None else kdev.data_ptr(), s_enc.data_ptr(), s_d1.data_ptr(), s_d2.data_ptr(),
                     None if action_out is None else action_out.data_ptr(), None if logp_out is None else logp_out.data_ptr(),
                     value_out.data_ptr(), None if done is None else done.data_ptr())
        if self._act_w_dirty and not torch.cuda.is_current_stream_capturing():
            self.build_act_weights()   # (a captured rollout rebuilds them itself, as graph nodes: MagpoLearner._rollout_body)
        tabs = self._act_tabs.get(cache_key)
        if tabs is None:
            if not self.wa:
                self.build_act_weights()
            wa = self.wa
            g = lambda n, w=E, rows=R: b.get(f"f{tag}_" + n, (rows, w))   # scratch per caller tag (env groups may act concurrently)
            ptr = lambda t: 0 if t is None else t.data_ptr()
            glob = [obs, pos, mask, kdev,
                    v["enc.obs.norm.scale"], v["enc.obs.dense.kernel"], v["enc.ln.scale"], v["dec.act.kernel"], v["dec.ln.scale"],
                    wa["vh0"], v["enc.head.dense0.bias"], v["enc.head.norm.scale"], v["enc.head.dense1.kernel"], v["enc.head.dense1.bias"],
                    wa["h0"], v["dec.head.dense0.bias"], v["dec.head.norm.scale"], wa["h1"], v["dec.head.dense1.bias"],
                    self.pe, s_enc, s_d1, s_d2,
                    g("xn"), done, g("qkvg", 4 * E), g("u"), g("y"), g("rep"), g("reppe"), g("hv"),
                    g("xa", E, N), g("kin1", E, N), g("y1", E, N), g("c", E, N), g("cpe", E, N), g("y2", E, N), g("xo", E, N),
                    g("xope", E, N), g("hp", E, N), g("hn", E, N), g("logits", E, N), g("u1"), g("u2"),
                    b.get(f"f{tag}_prev", (N, A), torch.int32, zero=True), action_out, logp_out, value_out,
                    b.get(f"f{tag}_ptab", (N, (K + 2) * E)) if nh == 1 else None]
            blk = []
            for k in range(nb):
                e, d = f"enc.block{k}.", f"dec.block{k}."
                blk += [wa[f"qkvg{k}"], wa[f"wo{k}"], v[e + "ln1.scale"], v[e + "ln2.scale"], v[e + "retn.gn.scale"], v[e + "retn.gn.bias"],
                        wa[f"qkvg1{k}"], wa[f"wo1{k}"], v[d + "ln1.scale"], v[d + "retn1.gn.scale"], v[d + "retn1.gn.bias"],
                        wa[f"q2{k}"], wa[f"kvg2{k}"], wa[f"wo2{k}"], v[d + "ln2.scale"], v[d + "ln3.scale"],
                        v[d + "retn2.gn.scale"], v[d + "retn2.gn.bias"],
                        g(f"qkvg1_{k}", 4 * E), g(f"q2_{k}"), g(f"kvg2_{k}", 4 * E)]   # kvg2 rows: [k | v | - | P2] (ld 256)
            tabs = (np.array([N, A, K, F, nb, nh, self.hs, self.gs, self.npos, 1 if value_only else 0, self.Fld, self.tuning.act_envs_per_wave,
                              1 if pending else 0, 1 if flush else 0, 1 if precand else 0, 1 if defer else 0], dtype=np.int32),
                    np.array((self.kappas + [0.0] * 4)[:4], dtype=np.float32),
                    np.array([ptr(t) for t in glob], dtype=np.uint64), np.array([ptr(t) for t in blk], dtype=np.uint64))
            if len(self._act_tabs) > 4096:
                self._act_tabs.clear()
            self._act_tabs[cache_key] = tabs
        dims, kap, gp, bp = tabs
        keys = None
        if kdev is None and not value_only:
            keys = np.ascontiguousarray(np.asarray(sample_keys, dtype=np.uint32).reshape(A, 2))
        self.L.call("magpo_sable_act", dims.ctypes.data, kap.ctypes.data, None if keys is None else keys.ctypes.data,
                    gp.ctypes.data, int(gp.size), bp.ctypes.data, int(bp.size), self._st())

    # the reference's names for the two apply functions of the Sable network (rec_magpo.py:624-628)
    get_actions = act_fused          # partial(sable_network.apply, method="get_actions"): execution  (apply = train_fwd: end of the file)

    def _seg_post(self, tail, r, gp, ldg, gamma, beta, wo_t, res, s1, s2, pos, u, y, o, ope, R, w0_t=None, b0=None, out0=None, ld0=0,
                  hs=None, hw=None, hb1=None, value=None, q2_t=(), q2=(), hn=None, w1_t=None, b1=None, logits=None, rows=None):
        """One fused launch for the token-local part between two retention ops (csrc/seg_fused.hip)."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        q2_t, q2 = list(q2_t) + [None] * (4 - len(q2_t)), list(q2) + [None] * (4 - len(q2))
        tab = [r, gp, gamma, beta, wo_t, res, s1, s2, self.pe, pos, u, y, o, ope, w0_t, b0, out0, hs, hw, hb1, value, *q2_t, *q2, hn, w1_t, b1, logits, rows]
        key = (tail, R, tuple(ptr(t) for t in tab))
        ent = self._seg_tabs.get(key)
        if ent is None:
            nq2 = sum(1 for t in q2 if t is not None)
            ent = (np.array([tail, self.K, self.npos, ldg, ld0, nq2], dtype=np.int32), np.array([ptr(t) for t in tab], dtype=np.uint64))
            if len(self._seg_tabs) > 256:
                self._seg_tabs.clear()
            self._seg_tabs[key] = ent
        self.L.call("magpo_seg_post", ent[0].ctypes.data, R, ent[1].ctypes.data, int(ent[1].size), self._st())

    def _seg_bwd(self, a, y, s1, s2, d0, d1, d2, wo_nat, r, gp, ldg, pfx, dsum, dr, dgp, lddg, R, g_s1, g_s2, acc_s1=False, rows=None, wo_t=None):
        """Backward of the front of a post-retention segment in one launch (csrc/seg_fused.hip: k_seg_bwd): d(res + y) through the
        RMSNorm(s), dsum W_o^T, GroupNorm + gate backward, and the four parameter-gradient rows (reduced from per-wave slabs)."""
        v, gv, b = self.v, self.gv, self.b
        ptr = lambda t: 0 if t is None else t.data_ptr()
        G = self.L.call("magpo_seg_bwd_grid", R)
        sl = [b.get(f"sb_{i}", (G, E)) for i in range(4)]
        tab = [a, y, s1, s2, d0, d1, d2, wo_nat, r, gp, v[pfx + "gn.scale"], v[pfx + "gn.bias"], dsum, dr, dgp, sl[0], sl[1] if s2 is not None else None, sl[2], sl[3], rows, wo_t]
        key = ("bwd", R, ldg, lddg, tuple(ptr(t) for t in tab))
        ent = self._seg_tabs.get(key)
        if ent is None:
            ent = np.array([ptr(t) for t in tab], dtype=np.uint64)
            self._seg_tabs[key] = ent
        self.L.call("magpo_seg_bwd", R, ldg, lddg, ent.ctypes.data, int(ent.size), self._st())
        self.reduce(sl[0], g_s1, accumulate=acc_s1)
        if s2 is not None:
            self.reduce(sl[1], g_s2)
        self.reduce(sl[2], gv[pfx + "gn.scale"]); self.reduce(sl[3], gv[pfx + "gn.bias"])

    # ------------------------------------------------------------------ training forward (chunkwise form)
    def train_fwd(self, obs, prev_idx, pos, dones, s0, seq_env, nseq: int, T: int, classes=None):
        """obs [R,F], prev_idx [R] (0 = start token, a+1 otherwise), pos [R] step counts, dones [nseq,T] u8,
        s0 = three [n_block, N, 64, 64] rollout-start states indexed through seq_env [nseq].
        ``classes`` (optional, csrc/classtab.hip) = dict(rows=(obs_enc [Ce,F], pos_enc [Ce], prev_dec [Cd], pos_dec [Cd]),
        enc=(cls, order, offsets), dec=(cls, order, offsets)): the embeddings and the first q|k|v|g projections of encoder
        and decoder are then evaluated on the distinct input rows only and gathered by class.
        Returns (logits [R,64] raw with K valid columns, value [R])."""
        L, st, A, K, F, v, b, nb, E = self.L, self._st(), self.A, self.K, self.F, self.v, self.b, self.nb, self.E
        R = nseq * T * A
        g = lambda n, w=E: b.get("t_" + n, (R, w))
        direct = classes is not None and self.fused_segments   # block-0 consumers read the class tables through the class index
        self._saved = dict(obs=obs, prev_idx=prev_idx, pos=pos, dones=dones, nseq=nseq, T=T, R=R, classes=classes, direct=direct)
        rep, reppe, hv, value = g("rep"), g("reppe"), g("hv"), b.get("t_value", (R,))
        logits = b.get("t_logits", (R, LW), zero=True)
        # ---- encoder
        if classes is not None:   # embedding + first projection on the Ce distinct (agent, target, step) rows, gathered by class
            obs_c, pos_c = classes["rows"][0], classes["rows"][1]
            Ce = obs_c.shape[0]
            xn_c, kin_c, qkvg_c = b.get("c_xn0", (Ce, E)), b.get("c_kin0", (Ce, E)), b.get("c_qkvg0", (Ce, 4 * E))
            L.call("magpo_embed_fwd", 0, obs_c, F, F, v["enc.obs.norm.scale"], v["enc.obs.dense.kernel"], None, 0, v["enc.ln.scale"],
                   self.pe, pos_c, 1, self.npos, None, 0, xn_c, E, kin_c, E, Ce, E, st)
            self.lin(kin_c, E, self.wt["qkvg0"], None, qkvg_c, 4 * E, Ce, E, 4 * E)
            if not direct:   # per-token copies for the unfused segment kernels; the fused ones read the tables through the class index
                L.call("magpo_gather_rows", xn_c, E, classes["enc"][0], g("xn0"), E, R, E, st)
                L.call("magpo_gather_rows", qkvg_c, 4 * E, classes["enc"][0], g("qkvg0", 4 * E), 4 * E, R, 4 * E, st)
        elif self.wide:   # obs_encoder + ln on padded rows: RMSNorm_F -> Dense on the MFMA kernel -> GELU + RMSNorm (sable_network.py:93-101,132)
            on, z0 = b.get("t_on", (R, 128)), g("z0")
            L.call("magpo_obsnorm_fwd", obs, self.Fld, F, v["enc.obs.norm.scale"], on, R, st)
            self.lin(on, 128, self.wt["wobs"], None, z0, E, R, 128, E)
            L.call("magpo_headmid_fwd", z0, E, v["enc.ln.scale"], g("xn0"), E, None, None, None, 0, R, E, st)
            L.call("magpo_add_pe", g("xn0"), E, self.pe, pos, 1, self.npos, g("kin0"), E, R, E, st)
        else:
            L.call("magpo_embed_fwd", 0, obs, self.Fld, F, v["enc.obs.norm.scale"], v["enc.obs.dense.kernel"], None, 0, v["enc.ln.scale"],
                   self.pe, pos, 1, self.npos, None, 0, g("xn0"), E, g("kin0"), E, R, E, st)   # z is recomputed by the backward
        for k in range(nb):
            e = f"enc.block{k}."
            rows = classes["enc"][0] if direct and k == 0 else None
            if rows is not None:
                xn, qkvg = xn_c, qkvg_c
            else:
                xn, qkvg = g(f"xn{k}"), g(f"qkvg{k}", 4 * E)
            r, u = g(f"r{k}"), g(f"u{k}")
            y = None if self.fused_segments else g(f"y{k}")   # (fused segments: y = u W_o is recomputed by the backward, never stored)
            if k > 0 or classes is None:
                self.lin(g(f"kin{k}"), E, self.wt[f"qkvg{k}"], None, qkvg, 4 * E, R, E, 4 * E)
            self._ret_fwd(qkvg, 4 * E, qkvg[:, E:], 4 * E, qkvg[:, 2 * E:], 4 * E, r, s0[0][k], seq_env, dones, f"st_e{k}", nseq, T, 0, rows=rows)
            if self.fused_segments and k == nb - 1:
                # GroupNorm + gate, W_o, residual + norms, value head and the cross-retention queries of every decoder block: one launch
                self._seg_post(1, r, qkvg[:, 3 * E:], 4 * E, v[e + "retn.gn.scale"], v[e + "retn.gn.bias"], self.wt[f"wo{k}"], xn,
                               v[e + "ln1.scale"], v[e + "ln2.scale"], pos, u, y, rep, reppe, R, w0_t=self.wt["vh0"], b0=v["enc.head.dense0.bias"],
                               out0=hv, ld0=E, hs=v["enc.head.norm.scale"], hw=v["enc.head.dense1.kernel"], hb1=v["enc.head.dense1.bias"],
                               value=value, q2_t=[self.wt[f"q2{j}"] for j in range(nb)], q2=[g(f"q2{j}") for j in range(nb)], rows=rows)
                continue
            if self.fused_segments:
                repb = g(f"repb{k}")
                self._seg_post(0, r, qkvg[:, 3 * E:], 4 * E, v[e + "retn.gn.scale"], v[e + "retn.gn.bias"], self.wt[f"wo{k}"], xn,
                               v[e + "ln1.scale"], v[e + "ln2.scale"], pos, u, y, repb, None, R, rows=rows)
                L.call("magpo_resnorm_fwd", repb, E, None, 0, v["enc.ln.scale"], None, self.pe, pos, 1, self.npos,
                       g(f"xn{k + 1}"), E, g(f"kin{k + 1}"), E, R, E, st)
                continue
            self._retpost_fwd(r, qkvg[:, 3 * E:], 4 * E, v[e + "retn.gn.scale"], v[e + "retn.gn.bias"], u, R)
            self.lin(u, E, self.wt[f"wo{k}"], None, y, E, R, E, E)
            if k == nb - 1:
                L.call("magpo_resnorm_fwd", xn, E, y, E, v[e + "ln1.scale"], v[e + "ln2.scale"], self.pe, pos, 1, self.npos,
                       rep, E, reppe, E, R, E, st)
            else:
                repb = g(f"repb{k}")
                L.call("magpo_resnorm_fwd", xn, E, y, E, v[e + "ln1.scale"], v[e + "ln2.scale"], None, None, 0, 0, repb, E, None, 0, R, E, st)
                L.call("magpo_resnorm_fwd", repb, E, None, 0, v["enc.ln.scale"], None, self.pe, pos, 1, self.npos,
                       g(f"xn{k + 1}"), E, g(f"kin{k + 1}"), E, R, E, st)
        if not self.fused_segments:
            self.lin(rep, E, self.wt["vh0"], v["enc.head.dense0.bias"], hv, E, R, E, E)
            L.call("magpo_headmid_fwd", hv, E, v["enc.head.norm.scale"], None, 0, v["enc.head.dense1.kernel"], v["enc.head.dense1.bias"],
                   value, 1, R, E, st)
        # ---- decoder
        if classes is not None:   # action embedding + first projection on the Cd distinct (previous action, step) rows
            prev_c, posd_c = classes["rows"][2], classes["rows"][3]
            Cd = prev_c.shape[0]
            x_c, xpe_c, qkvg1_c = b.get("c_x0", (Cd, E)), b.get("c_xpe0", (Cd, E)), b.get("c_qkvg10", (Cd, 4 * E))
            L.call("magpo_embed_fwd", 1, None, 0, 0, None, v["dec.act.kernel"], prev_c, 1, v["dec.ln.scale"], self.pe, posd_c, 1, self.npos,
                   None, 0, x_c, E, xpe_c, E, Cd, E, st)
            self.lin(xpe_c, E, self.wt["qkvg10"], None, qkvg1_c, 4 * E, Cd, E, 4 * E)
            if not direct:
                L.call("magpo_gather_rows", x_c, E, classes["dec"][0], g("x0"), E, R, E, st)
                L.call("magpo_gather_rows", qkvg1_c, 4 * E, classes["dec"][0], g("qkvg10", 4 * E), 4 * E, R, 4 * E, st)
        else:
            L.call("magpo_embed_fwd", 1, None, 0, 0, None, v["dec.act.kernel"], prev_idx, 1, v["dec.ln.scale"], self.pe, pos, 1, self.npos,
                   None, 0, g("x0"), E, g("xpe0"), E, R, E, st)   # za = W_act[prev] is gathered again by the backward
        for k in range(nb):
            d = f"dec.block{k}."
            rows = classes["dec"][0] if direct and k == 0 else None
            if rows is not None:
                x, qkvg1 = x_c, qkvg1_c
            else:
                x, qkvg1 = g(f"x{k}"), g(f"qkvg1{k}", 4 * E)
            r1, u1 = g(f"r1{k}"), g(f"u1{k}")
            cpe, q2, kvg2, r2, u2 = g(f"cpe{k}"), g(f"q2{k}"), g(f"kvg2{k}", 3 * E), g(f"r2{k}"), g(f"u2{k}")
            y1, y2 = (None, None) if self.fused_segments else (g(f"y1{k}"), g(f"y2{k}"))
            if k > 0 or classes is None:
                self.lin(g(f"xpe{k}"), E, self.wt[f"qkvg1{k}"], None, qkvg1, 4 * E, R, E, 4 * E)
            self._ret_fwd(qkvg1, 4 * E, qkvg1[:, E:], 4 * E, qkvg1[:, 2 * E:], 4 * E, r1, s0[1][k], seq_env, dones, f"st_1{k}", nseq, T, 1, rows=rows)
            if self.fused_segments:
                # after the self-retention: gate, W_o, residual + norm (+ pe) and the k | v | g projection of the cross-retention
                self._seg_post(2, r1, qkvg1[:, 3 * E:], 4 * E, v[d + "retn1.gn.scale"], v[d + "retn1.gn.bias"], self.wt[f"wo1{k}"], x,
                               v[d + "ln1.scale"], None, pos, u1, y1, None, cpe, R, w0_t=self.wt[f"kvg2{k}"], out0=kvg2, ld0=3 * E, rows=rows)
                self._ret_fwd(q2, E, kvg2, 3 * E, kvg2[:, E:], 3 * E, r2, s0[2][k], seq_env, dones, f"st_2{k}", nseq, T, 1)
                if k == nb - 1:   # ... and after the cross-retention of the last block the logit head
                    self._seg_post(3, r2, kvg2[:, 2 * E:], 3 * E, v[d + "retn2.gn.scale"], v[d + "retn2.gn.bias"], self.wt[f"wo2{k}"], rep,
                                   v[d + "ln2.scale"], v[d + "ln3.scale"], pos, u2, y2, g(f"x{nb}"), None, R, w0_t=self.wt["h0"],
                                   b0=v["dec.head.dense0.bias"], out0=g("hp"), ld0=E, hs=v["dec.head.norm.scale"], hn=g("hn"),
                                   w1_t=self.wt["h1"], b1=v["dec.head.dense1.bias"], logits=logits)
                else:
                    self._seg_post(0, r2, kvg2[:, 2 * E:], 3 * E, v[d + "retn2.gn.scale"], v[d + "retn2.gn.bias"], self.wt[f"wo2{k}"], rep,
                                   v[d + "ln2.scale"], v[d + "ln3.scale"], pos, u2, y2, g(f"x{k + 1}"), g(f"xpe{k + 1}"), R)
                continue
            self._retpost_fwd(r1, qkvg1[:, 3 * E:], 4 * E, v[d + "retn1.gn.scale"], v[d + "retn1.gn.bias"], u1, R)
            self.lin(u1, E, self.wt[f"wo1{k}"], None, y1, E, R, E, E)
            L.call("magpo_resnorm_fwd", x, E, y1, E, v[d + "ln1.scale"], None, self.pe, pos, 1, self.npos, None, 0, cpe, E, R, E, st)   # only c + pe is consumed
            self.lin(reppe, E, self.wt[f"q2{k}"], None, q2, E, R, E, E)
            self.lin(cpe, E, self.wt[f"kvg2{k}"], None, kvg2, 3 * E, R, E, 3 * E)
            self._ret_fwd(q2, E, kvg2, 3 * E, kvg2[:, E:], 3 * E, r2, s0[2][k], seq_env, dones, f"st_2{k}", nseq, T, 1)
            self._retpost_fwd(r2, kvg2[:, 2 * E:], 3 * E, v[d + "retn2.gn.scale"], v[d + "retn2.gn.bias"], u2, R)
            self.lin(u2, E, self.wt[f"wo2{k}"], None, y2, E, R, E, E)
            if k == nb - 1:
                L.call("magpo_resnorm_fwd", rep, E, y2, E, v[d + "ln2.scale"], v[d + "ln3.scale"], None, None, 0, 0, g(f"x{nb}"), E,
                       None, 0, R, E, st)
            else:
                L.call("magpo_resnorm_fwd", rep, E, y2, E, v[d + "ln2.scale"], v[d + "ln3.scale"], self.pe, pos, 1, self.npos,
                       g(f"x{k + 1}"), E, g(f"xpe{k + 1}"), E, R, E, st)
        if not self.fused_segments:
            hp, hn = g("hp"), g("hn")
            self.lin(g(f"x{nb}"), E, self.wt["h0"], v["dec.head.dense0.bias"], hp, E, R, E, E)
            L.call("magpo_headmid_fwd", hp, E, v["dec.head.norm.scale"], hn, E, None, None, None, 0, R, E, st)
            self.lin(hn, E, self.wt["h1"], v["dec.head.dense1.bias"], logits, LW, R, E, K)
        return logits, value

    # ------------------------------------------------------------------ training backward
    def train_bwd(self, dlogits, dvalue):
        """dlogits [R,64] (columns >= K zero), dvalue [R]; fills self.grads (every entry written exactly once, the
        shared encoder ln scale accumulates over blocks)."""
        L, st, A, K, F, v, gv, b, nb, E = self.L, self._st(), self.A, self.K, self.F, self.v, self.gv, self.b, self.nb, self.E
        sv = self._saved
        R, nseq, T = sv["R"], sv["nseq"], sv["T"]
        obs, prev_idx, pos, dones = sv["obs"], sv["prev_idx"], sv["pos"], sv["dones"]
        cl, Rd = sv["classes"], R
        t = lambda n: b.t["t_" + n]
        g = lambda n, w=E: b.get("g_" + n, (R, w))
        grid = L.call("magpo_row_grid", R)
        slab = lambda n, w=E: b.get("s_" + n, (grid, w))
        # ---- logit head
        self.wgrad(t("hn"), E, dlogits, LW, R, E, K, gv["dec.head.dense1.kernel"], gv["dec.head.dense1.bias"])
        dhn = g("dhn")
        self.lin(dlogits, LW, self.wt["h1_nat_pad"], None, dhn, E, R, LW, E)
        dhp = g("dhp_l")
        L.call("magpo_headmid_bwd", t("hp"), E, v["dec.head.norm.scale"], dhn, E, None, None, 0, dhp, E, slab("a"), None, None, R, E, st)
        self.reduce(slab("a"), gv["dec.head.norm.scale"])
        self.wgrad(t(f"x{nb}"), E, dhp, E, R, E, E, gv["dec.head.dense0.kernel"], gv["dec.head.dense0.bias"])
        dout = g("dout")
        self.lin(dhp, E, v["dec.head.dense0.kernel"], None, dout, E, R, E, E)
        # ---- decoder blocks, last to first.  incoming gradient of block k's output x_{k+1}: (din0 [+ din1])
        din0, din1 = dout, None
        drep_q, drep_r = None, None      # running sums of d(obs_rep): cross-retention query path / residual path
        for k in reversed(range(nb)):
            d = f"dec.block{k}."
            dsum2 = g(f"dsum2_{k}")
            dr2 = g("dr"); dq2 = g(f"dq2_{k}"); dkvg2 = g(f"dkvg2_{k}", 3 * E)
            kvg2 = t(f"kvg2{k}")
            if self.fused_segments:
                self._seg_bwd(t("rep"), None, v[d + "ln2.scale"], v[d + "ln3.scale"], din0, din1, None, v[d + "retn2.w_o"], t(f"r2{k}"),
                              kvg2[:, 2 * E:], 3 * E, d + "retn2.", dsum2, dr2, dkvg2[:, 2 * E:], 3 * E, R, gv[d + "ln2.scale"], gv[d + "ln3.scale"],
                              wo_t=self.wt[f"wo2{k}"])
                self.wgrad(t(f"u2{k}"), E, dsum2, E, R, E, E, gv[d + "retn2.w_o"])
            else:
                L.call("magpo_resnorm_bwd", t("rep"), E, t(f"y2{k}"), E, v[d + "ln2.scale"], v[d + "ln3.scale"], din0, E, din1, E if din1 is not None else 0,
                       None, 0, dsum2, E, slab("a"), slab("b"), R, E, st)
                self.reduce(slab("a"), gv[d + "ln2.scale"]); self.reduce(slab("b"), gv[d + "ln3.scale"])
                self.wgrad(t(f"u2{k}"), E, dsum2, E, R, E, E, gv[d + "retn2.w_o"])
                du2 = g("du")
                self.lin(dsum2, E, v[d + "retn2.w_o"], None, du2, E, R, E, E)
                self._retpost_bwd(t(f"r2{k}"), kvg2[:, 2 * E:], 3 * E, d + "retn2.", du2, dr2, dkvg2[:, 2 * E:], 3 * E, R, slab("a"), slab("b"))
            self._ret_bwd(t(f"q2{k}"), E, kvg2, 3 * E, kvg2[:, E:], 3 * E, dr2, dq2, E, dkvg2, 3 * E, dkvg2[:, E:], 3 * E, dones,
                          f"st_2{k}", nseq, T, 1)
            self.wgrad(t("reppe"), E, dq2, E, R, E, E, gv[d + "retn2.w_q"])
            self.wgrad(t(f"cpe{k}"), E, dkvg2, 3 * E, R, E, 3 * E, gv[d + "retn2.w_kvg"])
            dreppe = g(f"dreppe_{k}"); dcpe = g("dcpe")
            self.lin(dq2, E, v[d + "retn2.w_q"], None, dreppe, E, R, E, E)
            self.lin(dkvg2, 3 * E, v[d + "retn2.w_kvg"], None, dcpe, E, R, 3 * E, E)
            if drep_q is None:
                drep_q, drep_r = dreppe, dsum2
            else:
                self.add_(drep_q, dreppe); self.add_(drep_r, dsum2)
            # self-retention: c = rms(x_k + y1) * ln1
            dsum1 = g(f"dsum1_{k}")
            dr1 = g("dr"); dqkvg1 = g(f"dqkvg1_{k}", 4 * E)
            rows = cl["dec"][0] if sv["direct"] and k == 0 else None
            qkvg1, xk = (b.t["c_qkvg10"], b.t["c_x0"]) if rows is not None else (t(f"qkvg1{k}"), t(f"x{k}"))
            if self.fused_segments:
                self._seg_bwd(xk, None, v[d + "ln1.scale"], None, dcpe, None, None, v[d + "retn1.w_o"], t(f"r1{k}"),
                              qkvg1[:, 3 * E:], 4 * E, d + "retn1.", dsum1, dr1, dqkvg1[:, 3 * E:], 4 * E, R, gv[d + "ln1.scale"], None, rows=rows,
                              wo_t=self.wt[f"wo1{k}"])
                self.wgrad(t(f"u1{k}"), E, dsum1, E, R, E, E, gv[d + "retn1.w_o"])
            else:
                L.call("magpo_resnorm_bwd", t(f"x{k}"), E, t(f"y1{k}"), E, v[d + "ln1.scale"], None, dcpe, E, None, 0, None, 0, dsum1, E,
                       slab("a"), None, R, E, st)
                self.reduce(slab("a"), gv[d + "ln1.scale"])
                self.wgrad(t(f"u1{k}"), E, dsum1, E, R, E, E, gv[d + "retn1.w_o"])
                du1 = g("du")
                self.lin(dsum1, E, v[d + "retn1.w_o"], None, du1, E, R, E, E)
                self._retpost_bwd(t(f"r1{k}"), qkvg1[:, 3 * E:], 4 * E, d + "retn1.", du1, dr1, dqkvg1[:, 3 * E:], 4 * E, R, slab("a"), slab("b"))
            self._ret_bwd(qkvg1, 4 * E, qkvg1[:, E:], 4 * E, qkvg1[:, 2 * E:], 4 * E, dr1, dqkvg1, 4 * E, dqkvg1[:, E:], 4 * E,
                          dqkvg1[:, 2 * E:], 4 * E, dones, f"st_1{k}", nseq, T, 1, rows=rows)
            if k == 0 and cl is not None:   # block 0 on the class table: per-class sums of both gradient paths, then Cd rows
                _, order, offsets = cl["dec"]
                Cd = cl["rows"][2].shape[0]
                dq_c, ds_c = b.get("gc_dqkvg1", (Cd, 4 * E)), b.get("gc_dsum1", (Cd, E))
                part = b.get("gc_part_d", (L.call("magpo_class_sum_slots", Cd), Cd, 4 * E))
                L.call("magpo_class_sum", dqkvg1, 4 * E, order, offsets, Cd, 4 * E, part, dq_c, st)
                L.call("magpo_class_sum", dsum1, E, order, offsets, Cd, E, part, ds_c, st)
                self.wgrad(b.t["c_xpe0"], E, dq_c, 4 * E, Cd, E, 4 * E, gv[d + "retn1.w_qkvg"])
                dkin1 = b.get("gc_dkin1", (Cd, E))
                self.lin_dx_qkvg(dq_c, d + "retn1.w_qkvg", dkin1, Cd)
                din0, din1, prev_idx, Rd = ds_c, dkin1, cl["rows"][2], Cd
                break
            self.wgrad(t(f"xpe{k}"), E, dqkvg1, 4 * E, R, E, 4 * E, gv[d + "retn1.w_qkvg"])
            dkin1 = g(f"dkin1_{k}")
            self.lin_dx_qkvg(dqkvg1, d + "retn1.w_qkvg", dkin1, R)
            din0, din1 = dsum1, dkin1      # gradient of x_k (block input): residual path + key/query/value path
        gridd = L.call("magpo_row_grid", Rd)
        sa_d, sw_d = b.get("s_a_d", (gridd, E)), b.get("s_w_d", (gridd, 32 * E))
        L.call("magpo_embed_bwd", 1, None, 0, din0, E, din1, E, None, 0, v["dec.ln.scale"], None, 0, sa_d, sw_d,
               K + 1, None, 0, 0, None, v["dec.act.kernel"], None, prev_idx, 1, Rd, E, st)
        self.reduce(sa_d, gv["dec.ln.scale"])
        self.reduce(sw_d, gv["dec.act.kernel"], P=(K + 1) * E, stride=32 * E)
        # ---- value head
        dhv = g("dhv")
        L.call("magpo_headmid_bwd", t("hv"), E, v["enc.head.norm.scale"], None, 0, v["enc.head.dense1.kernel"], dvalue, 1, dhv, E,
               slab("a"), slab("b"), slab("c", 1), R, E, st)
        self.reduce(slab("a"), gv["enc.head.norm.scale"]); self.reduce(slab("b"), gv["enc.head.dense1.kernel"])
        self.reduce(slab("c", 1), gv["enc.head.dense1.bias"], P=1, stride=1)
        self.wgrad(t("rep"), E, dhv, E, R, E, E, gv["enc.head.dense0.kernel"], gv["enc.head.dense0.bias"])
        drep_v = g("dout")
        self.lin(dhv, E, v["enc.head.dense0.kernel"], None, drep_v, E, R, E, E)
        # ---- encoder blocks, last to first; d(rep) = value head + cross-retention queries + decoder residuals
        e0, e1, e2 = drep_v, drep_q, drep_r
        first_ln = True
        for k in reversed(range(nb)):
            e = f"enc.block{k}."
            dsum0 = g(f"dsum0_{k}")
            dr = g("dr"); dqkvg = g(f"dqkvg_{k}", 4 * E)
            rows = cl["enc"][0] if sv["direct"] and k == 0 else None
            qkvg, xnk = (b.t["c_qkvg0"], b.t["c_xn0"]) if rows is not None else (t(f"qkvg{k}"), t(f"xn{k}"))
            if self.fused_segments:
                self._seg_bwd(xnk, None, v[e + "ln1.scale"], v[e + "ln2.scale"], e0, e1, e2, v[e + "retn.w_o"], t(f"r{k}"),
                              qkvg[:, 3 * E:], 4 * E, e + "retn.", dsum0, dr, dqkvg[:, 3 * E:], 4 * E, R, gv[e + "ln1.scale"], gv[e + "ln2.scale"], rows=rows,
                              wo_t=self.wt[f"wo{k}"])
                self.wgrad(t(f"u{k}"), E, dsum0, E, R, E, E, gv[e + "retn.w_o"])
            else:
                L.call("magpo_resnorm_bwd", t(f"xn{k}"), E, t(f"y{k}"), E, v[e + "ln1.scale"], v[e + "ln2.scale"], e0, E, e1, E if e1 is not None else 0,
                       e2, E if e2 is not None else 0, dsum0, E, slab("a"), slab("b"), R, E, st)
                self.reduce(slab("a"), gv[e + "ln1.scale"]); self.reduce(slab("b"), gv[e + "ln2.scale"])
                self.wgrad(t(f"u{k}"), E, dsum0, E, R, E, E, gv[e + "retn.w_o"])
                du = g("du")
                self.lin(dsum0, E, v[e + "retn.w_o"], None, du, E, R, E, E)
                self._retpost_bwd(t(f"r{k}"), qkvg[:, 3 * E:], 4 * E, e + "retn.", du, dr, dqkvg[:, 3 * E:], 4 * E, R, slab("a"), slab("b"))
            self._ret_bwd(qkvg, 4 * E, qkvg[:, E:], 4 * E, qkvg[:, 2 * E:], 4 * E, dr, dqkvg, 4 * E, dqkvg[:, E:], 4 * E, dqkvg[:, 2 * E:], 4 * E,
                          dones, f"st_e{k}", nseq, T, 0, rows=rows)
            if k == 0 and cl is not None:
                _, order, offsets = cl["enc"]
                obs_c = cl["rows"][0]
                Ce = obs_c.shape[0]
                dq_c, ds_c = b.get("gc_dqkvg0", (Ce, 4 * E)), b.get("gc_dsum0", (Ce, E))
                part = b.get("gc_part_e", (L.call("magpo_class_sum_slots", Ce), Ce, 4 * E))
                L.call("magpo_class_sum", dqkvg, 4 * E, order, offsets, Ce, 4 * E, part, dq_c, st)
                L.call("magpo_class_sum", dsum0, E, order, offsets, Ce, E, part, ds_c, st)
                self.wgrad(b.t["c_kin0"], E, dq_c, 4 * E, Ce, E, 4 * E, gv[e + "retn.w_qkvg"])
                dkin = b.get("gc_dkin0", (Ce, E))
                self.lin_dx_qkvg(dq_c, e + "retn.w_qkvg", dkin, Ce)
                gride = L.call("magpo_row_grid", Ce)
                sa, sw, sd = b.get("s_a_e", (gride, E)), b.get("s_w_e", (gride, 32 * E)), b.get("s_d_e", (gride, 32))
                L.call("magpo_embed_bwd", 0, None, 0, ds_c, E, dkin, E, None, 0, v["enc.ln.scale"], None, 0, sa, sw, F,
                       obs_c, F, F, v["enc.obs.norm.scale"], v["enc.obs.dense.kernel"], sd, None, 0, Ce, E, st)
                self.reduce(sa, gv["enc.ln.scale"], accumulate=not first_ln)
                self.reduce(sd, gv["enc.obs.norm.scale"], P=F, stride=32)
                self.reduce(sw, gv["enc.obs.dense.kernel"], P=F * E, stride=32 * E)
                break
            self.wgrad(t(f"kin{k}"), E, dqkvg, 4 * E, R, E, 4 * E, gv[e + "retn.w_qkvg"])
            dkin = g(f"dkin_{k}")
            self.lin_dx_qkvg(dqkvg, e + "retn.w_qkvg", dkin, R)
            if k > 0:  # xn_k = rms(rep_{k-1}) * enc.ln (shared scale): d(rep_{k-1})
                drepb = g(f"drepb_{k}")
                L.call("magpo_resnorm_bwd", t(f"repb{k - 1}"), E, None, 0, v["enc.ln.scale"], None, dsum0, E, dkin, E, None, 0, drepb, E,
                       slab("a"), None, R, E, st)
                self.reduce(slab("a"), gv["enc.ln.scale"], accumulate=not first_ln)
                first_ln = False
                e0, e1, e2 = drepb, None, None
            elif self.wide:
                dxn, dz, don = g("dxn0"), g("dz0"), b.get("g_don", (R, 128))
                L.call("magpo_copy_rows", dsum0, E, dxn, E, R, E, st)
                self.add_(dxn, dkin)
                L.call("magpo_headmid_bwd", t("z0"), E, v["enc.ln.scale"], dxn, E, None, None, 0, dz, E, slab("a"), None, None, R, E, st)
                self.reduce(slab("a"), gv["enc.ln.scale"], accumulate=not first_ln)
                self.wgrad(b.t["t_on"], 128, dz, E, R, 128, E, gv["enc.obs.dense.kernel"], krows=F)
                self.lin(dz, E, self.wt["wobs_nat_pad"], None, don, 128, R, E, 128)
                so = b.get("s_obsn", (L.call("magpo_obsnorm_grid", R), 128))
                L.call("magpo_obsnorm_bwd", obs, self.Fld, F, don, so, R, st)
                self.reduce(so, gv["enc.obs.norm.scale"], P=F, stride=128)
            else:
                L.call("magpo_embed_bwd", 0, None, 0, dsum0, E, dkin, E, None, 0, v["enc.ln.scale"], None, 0, slab("a"), slab("w", 32 * E), F,
                       obs, self.Fld, F, v["enc.obs.norm.scale"], v["enc.obs.dense.kernel"], slab("d", 32), None, 0, R, E, st)
                self.reduce(slab("a"), gv["enc.ln.scale"], accumulate=not first_ln)
                self.reduce(slab("d", 32), gv["enc.obs.norm.scale"], P=F, stride=32)
                self.reduce(slab("w", 32 * E), gv["enc.obs.dense.kernel"], P=F * E, stride=32 * E)
        if self.overlap_wgrad and self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
        if self.emb is not None:   # gradient of a logical parameter = sum over its tied device copies
            self.emb.fold(self.grads_D, self.grads)


SableGuider.apply = SableGuider.train_fwd   # sable_network.apply: the chunkwise training forward (its backward: train_bwd)
