#!/usr/bin/env python3
"""Headline benchmark: MAGPO env-steps/sec on CoordSum-4ag, 16384 envs per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full ``_update_step`` of the reference learner (rec_magpo.py:106-499): a 128-step rollout of
all envs (guider acting, actor push, env step), GAE, then ppo_epochs x num_minibatches optimisation steps of
both networks (forward, fused losses, hand-written backward, gradient all-reduce across GPUs, clip + Adam).
Nothing is skipped.  value = n_gpus * num_envs * rollout_length * K / wall (the reference's steps_per_second,
rec_magpo.py:720-726,761), inputs resident in HBM, weak scaling (per-GPU envs fixed).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_HBM_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
PEAK_F32_MFMA_TF = 157.3   # fp32-input MFMA dense peak (v_mfma_f32_32x32x2_f32)
PMC_FILE = "r04_pmc_hbm_traffic.json"   # offline PMC passes (scripts/pmc_kernels.py + scripts/pmc_collect.py)


class KernelTimer:
    """HIP-event timing of the heavy kernel families inside the timed region (events on the stream the
    kernels are launched on = torch's current stream).  Cost model = ALGORITHMIC bytes / flops per launch
    (DESIGN.md section 4), so re-reads and padding do not count as work."""

    def __init__(self, min_rows: int):
        self.min_rows = min_rows
        self.rec = {}
        self.pending = []
        self.enabled = False

    def _model(self, name, a):
        if name == "magpo_linear":
            R, KIN, NOUT = a[7], a[8], a[9]
            if R < self.min_rows:
                return None
            if a[6] is None or a[10] == 4:   # no pre-activation copy (act 4: the slot carries the mask): shared-tile kernel
                nw = 4 if ((NOUT + 31) // 32) % 4 == 0 else 2
                key = f"k_linear_lds<{KIN}, {nw}, {'true' if (a[11] & 4 and nw == 4 and KIN in (128, 192)) else 'false'}>"
            else:
                key = f"k_linear_w<{KIN}>"
            return key, 4.0 * R * (KIN + NOUT), 2.0 * R * KIN * NOUT
        if name == "magpo_wgrad":
            R, KIN, NOUT = a[4], a[5], a[7]
            if R < self.min_rows:
                return None
            full = R >= 64 * 256 and ((KIN, NOUT) == (128, 384) or (R % 64 == 0 and (KIN, NOUT) in ((128, 128), (64, 256))))
            if R >= 64 * 256 and R % 64 == 0 and (KIN, NOUT) in ((64, 192), (128, 128)):
                key = f"k_wgrad_full_g<{KIN // 64}, {3 if NOUT == 192 else 2}>"
            elif full and R % 64 == 0:
                key = f"k_wgrad_full_x<{KIN // 32}, {NOUT // 128}, {0 if (KIN, NOUT) == (64, 256) else 4}, false>"
            elif full:
                key = f"k_wgrad_full<{KIN // 32}, {NOUT // 128}>"
            else:
                key = "k_wgrad<2>" if NOUT >= 128 else "k_wgrad<1>"
            return key, 4.0 * R * (KIN + NOUT), 2.0 * R * KIN * NOUT
        if name in ("magpo_retention_chunk_fwd", "magpo_retention_chunk_bwd"):
            fwd = name.endswith("fwd")
            nseq, T, A = (a[13], a[14], a[15]) if fwd else (a[16], a[17], a[18])
            from magpo_amd._lib import lib as _lib
            nch = int(_lib().raw("magpo_retention_num_chunks")(int(T), int(A), int(a[20] if fwd else a[23])))
            ct = 64 if nch == (T + 64 // A - 1) // (64 // A) else 32     # tokens per chunk of the kernels in use (csrc/retention32.hpp)
            rows = nseq * T * A
            byts = 4.0 * rows * 64 * (4 if fwd else 7) + 4.0 * nseq * nch * 4096
            if ct == 64:
                return ("k_ret_chunk_fwd" if fwd else "k_ret_chunk_bwd"), byts, (4 if fwd else 9) * 2.0 * 64 ** 3 * nseq * nch
            # executed MACs per 32-token chunk: fwd QK^T 32x32x64 + QS 32x64x64 + PV 32x64x32 + state 64x64x32; bwd P, dP 2 x 32x32x64,
            # dQ / dK / dV 3 x (32x64x32 + 32x64x64), G 64x64x32
            macs = 393216 if fwd else 851968
            # <FAST (every chunk full), ROWS (q | k | v rows read through the block-0 class table)>: the instance retention.hip launches
            fast = (32 % A == 0) and T % (32 // A) == 0 and int(a[18] if fwd else a[21]) == 64 and nch <= 32
            rows_t = (a[19] if fwd else a[22]) is not None
            if rows_t:   # the q | k | v rows of the table are L2-resident: only r out (fwd) / the gradient rows (bwd) and the states are HBM bytes
                byts = 4.0 * rows * 64 * (1 if fwd else 4) + 4.0 * nseq * nch * 4096
            tf = lambda b: "true" if b else "false"
            return (f"k_ret32_fwd<{tf(fast)}, {tf(rows_t)}>" if fwd else f"k_ret32_bwd<{tf(fast)}, {tf(rows_t)}>"), byts, 2.0 * macs * nseq * nch
        if name == "magpo_retention_recurrent":
            nenv, ntok, wr = a[10], a[11], a[14]
            return "k_ret_recurrent", 4.0 * nenv * ((2 if wr else 1) * 4096 + 4 * ntok * 64), 4.0 * nenv * ntok * 4096 + 2.0 * nenv * 4096
        if name == "magpo_gru_scan_fwd":
            nseq, T, A = a[9], a[10], a[11]
            if T == 1:
                return None
            rows = nseq * T * A
            # HBM bytes: gates [R, 512] + h [R, 128] + h_prev [R, 128] written; the input projection xi is read through the class table
            # (L2-resident rows, not HBM) when a class list is given, from [R, 384] otherwise
            xi_b = 0 if a[12] is not None else 384
            key = "k_gru_scan_fwd_bf3<true, 3>" if a[13] == 2 else ("k_gru_scan_fwd_bf<true>" if a[13] == 1 else "k_gru_scan_fwd<true, 0, 64>")
            return key, 4.0 * rows * (xi_b + 128 + 512 + 128), 2.0 * rows * 128 * 384
        if name == "magpo_gru_scan_bwd":
            nseq, T, A = a[7], a[8], a[9]
            rows = nseq * T * A
            return "k_gru_scan_bwd<true, 64>", 4.0 * rows * (512 + 128 + 128 + 512), 2.0 * rows * 384 * 128
        if name == "magpo_gru_carry":   # the rollout's actor carry as one scan over the time-major trajectory (xi from the class table)
            nenv, T, A = a[6], a[7], a[8]
            rows = nenv * T * A
            return "k_gru_scan_fwd<true, 1, 64>", 4.0 * rows * ((0 if a[9] is not None else 384) + 1) + 8.0 * nenv * A * 128, 2.0 * rows * 128 * 384
        if name == "magpo_seg_post":
            import ctypes
            d = (ctypes.c_int * 6).from_address(a[0])
            tail, K, nq2 = d[0], d[1], d[5]
            R = a[1]
            if R < self.min_rows:
                return None
            pt = (ctypes.c_uint64 * a[3]).from_address(a[2])
            tab = pt[a[3] - 1] != 0          # rows index given: gate / residual rows come from the L2-resident class table, not from HBM
            # HBM row streams of 256 B: r in, gate + residual in (unless table rows), u + o (+ ope) out, then the tail's outputs
            streams = 1 + (0 if tab else 2) + 2 + {0: 0.0, 1: 1.0 + nq2 + 1.0 / 64, 2: 1.0 + 3.0, 3: 1.0 + 1.0 + K / 64.0}[tail]
            dense = {0: 1, 1: 2 + nq2, 2: 4, 3: 2 + K / 64.0}[tail]
            return f"k_seg_post<{tail}>", 256.0 * R * streams, 2.0 * R * 64 * 64 * dense
        if name == "magpo_class_sum":   # bit-stable per-class row sums: every row of X [R, W] read once (classtab.hip)
            R, W = a[2].numel(), a[5]
            if R < self.min_rows:
                return None
            return "k_class_sum", 4.0 * R * W + 8.0 * R, 1.0 * R * W
        if name == "magpo_headmid_bwd":
            R, E = a[13], a[14]
            if R < self.min_rows:
                return None
            return f"k_headmid_bwd<{E}>", 4.0 * R * E * 3, 30.0 * R * E
        if name == "magpo_seg_bwd":
            import ctypes
            R = a[0]
            pt = (ctypes.c_uint64 * a[4]).from_address(a[3])
            tab = pt[19] != 0   # rows index given: the gate / residual rows of block 0 come from the L2-resident class table
            # HBM row streams of 256 B: a, y (recomputed from u: no), d0..d2 in, r, gate in (unless table rows), dsum, dr, dgp out
            return "k_seg_bwd<true>", 256.0 * R * (7 if tab else 9), 2.0 * R * 64 * 64 * 2
        if name == "magpo_sable_act":
            import ctypes
            d = (ctypes.c_int * 10).from_address(a[0])
            N, A, nb, nh, hs, value_only = d[0], d[1], d[4], d[5], d[6], d[9]
            if value_only or N < 1024:
                return None
            # the three retention states read + written once per env step = 2 * 3 * nb * nh * hs^2 * 4 B per env
            # (98 304 B at E = 64, one block, one head).  Everything else the launch touches (obs, weights, outputs) is < 1 %.
            state = 4.0 * nb * nh * hs * hs
            d14 = (ctypes.c_int * 14).from_address(a[0])
            from magpo_amd._lib import lib as _lib
            epw = _lib().call("magpo_sable_act_envs_per_wave", int(N), int(A), int(d14[11]))   # the instance the library launches (act_fused.hip)
            return f"k_sable_act<{epw}, {4 if A <= 4 else 8}, {1 if (nh == 1 and A <= 4) else 0}>", 6.0 * state * N, N * A * (46.0 * 64 * 64 + 12.0 * 64 * 64 / nh)
        if name == "magpo_coordsum_step":   # ~0.5 KB per env-step (record row + targets + obs / reward / metrics out)
            N, A, TL = a[9], a[10], a[12]
            return "k_coordsum_step", float(N) * (4.0 * TL + 4.0 * A * (A + 2) + 64.0), 0.0
        if name == "magpo_lbf_step":        # state in + out, observation + action mask + reward / metrics out
            N, A, NF = a[12], a[13], a[14]
            return "k_lbf_step", float(N) * (2.0 * (12.0 * A + 13.0 * NF + 40.0) + 4.0 * A * (A + 3 * (NF + A)) + 6.0 * A + 4.0 * A + 12.0), 0.0
        if name == "magpo_rware_step":      # the cells a step touches + observation rows (padded to 128 floats) + masks / reward / metrics
            N, A = a[15], a[16]
            return "k_rware_step", float(N) * (4.0 * A * 128 + 2.0 * 4.0 * (6 * A + 9 * A) + 5.0 * A + 4.0 * A + 64.0), 0.0
        if name == "magpo_loss_fwd_bwd":
            R, K = a[19], a[20]
            return "k_magpo_loss", 4.0 * R * (4 * K + 8), 60.0 * R * K
        return None

    only = None       # restrict timing to these entry points
    all_calls = False  # time EVERY entry point (those without a cost model as "other: <entry point>"): the eager set-up step only --
    #                    an event pair around each of the ~700 small launches of an update step would cost the timed region ~0.5 %

    def begin(self, name, args):
        if not self.enabled or torch.cuda.is_current_stream_capturing() or (self.only is not None and name not in self.only):
            return None
        m = self._model(name, args)
        if m is None:
            if not self.all_calls or name in ("magpo_sable_act_envs_per_wave", "magpo_class_sum_slots", "magpo_row_grid", "magpo_seg_bwd_grid",
                                              "magpo_retention_num_chunks", "magpo_obsnorm_grid", "magpo_wgrad_workspace_floats", "magpo_abi_version"):
                return None
            m = (f"other: {name}", 0.0, 0.0)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (m, e0, e1)

    def end(self, tok):
        m, e0, e1 = tok
        e1.record()
        self.pending.append((m, e0, e1))

    def collect(self):
        torch.cuda.synchronize()
        for (key, byts, flops), e0, e1 in self.pending:
            r = self.rec.setdefault(key, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
            r["calls"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["bytes"] += byts
            r["flops"] += flops
        self.pending = []

    def dominant(self, steps: int = 1, once_per_step=()):
        """``once_per_step``: keys whose record covers exactly ONE update step (the acting kernel, timed on one eager rollout
        in the untimed set-up because inside the timed region it runs inside the rollout's HIP graph); the others cover
        ``steps`` update steps of the timed region.  The dominant kernel is the one with the most time per update step."""
        if not self.rec:
            return None, None
        per_step = lambda k: self.rec[k]["ms"] / (1 if k in once_per_step else max(1, steps))
        key = max(self.rec, key=per_step)
        r = self.rec[key]
        avg_s = r["ms"] / r["calls"] / 1e3
        gbs = r["bytes"] / r["calls"] / avg_s / 1e9
        tfs = r["flops"] / r["calls"] / avg_s / 1e12
        # the bound is whichever resource the algorithmic work needs longer at peak
        if gbs / PEAK_HBM_GBS >= tfs / PEAK_F32_MFMA_TF:
            roof = dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
        else:
            roof = dict(bound="mfma", achieved=round(tfs, 2), peak=PEAK_F32_MFMA_TF, unit="TFLOP/s", frac=round(tfs / PEAK_F32_MFMA_TF, 4))
        roof.update(kernel=key, avg_us=round(avg_s * 1e6, 1), calls=r["calls"], ms_per_update_step=round(per_step(key), 2), traffic=None,
                    traffic_source=None, algorithmic_bytes_per_launch=round(r["bytes"] / r["calls"]),
                    timing=("HIP events around each launch of one eager rollout in the untimed set-up (inside the timed region this kernel "
                            "runs inside the rollout's HIP graph, whose replays are timed as a whole: see rollout_graph_ms_per_step)"
                            if key in once_per_step else "HIP events around every launch inside the timed region"))
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes; same
        # workload, profiles/r01_pmc_hbm_traffic.json, made by scripts/pmc_kernels.py + scripts/pmc_collect.py).  Only attached when the kernel family matches.
        try:
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
                pmc = json.load(f)
            name = key.split("<")[0] if key.startswith("k_sable_act") else key   # timer keys are the rocprof kernel names
            cand = [k for k in pmc if k == name or (name == "k_sable_act" and k.startswith("k_sable_act"))]
            if cand and getattr(self, "attach_traffic", True):
                roof["traffic"] = round(max(pmc[k]["total"] for k in cand))
                roof["traffic_source"] = (f"profiles/{PMC_FILE}: offline rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, separate runs) "
                                          "of this workload; NOT measured in this run")
        except (OSError, ValueError):
            pass
        table = {k: dict(calls=v["calls"], ms=round(v["ms"], 2), ms_per_step=round(per_step(k), 2), gbs=round(v["bytes"] / v["ms"] / 1e6, 1),
                         tflops=round(v["flops"] / v["ms"] / 1e9, 2)) for k, v in sorted(self.rec.items(), key=lambda kv: -per_step(kv[0]))}
        return roof, table


def cpu_baseline():
    """The CPU oracle (a torch-CPU port of the reference loop; the reference's own JAX path cannot run here:
    jax / flax / jumanji are not installed) timed on this box's host cores on a bounded sample of the same
    workload: CoordSum-4ag, same network sizes, rollout_length=128, 4 epochs x 2 minibatches, 256 envs (SURVEY 8(d) asks for
    N in {4, 64, 1024}; at 256 envs the torch-CPU kernels are no longer launch-bound and one update step takes ~10 s)."""
    from oracle import coordsum as ocs
    from oracle import learner as olearn
    from oracle import networks as onets
    from oracle import prng as oprng
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the GPU box gives one GPU a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    A, K, N = 4, 20, 256
    ol = olearn.OracleLearner(ocs.CoordSumSpec(A, K, 100, 60), N, olearn.SystemCfg(), onets.SableCfg(A, K, A + 1),
                              onets.init_guider_params(1, 64, A + 1, K), onets.init_actor_params(2, A + 1, 128, K))
    ol.setup(oprng.split(oprng.prng_key(42), 4)[0])
    times = []
    for _ in range(3):   # three full update steps, the MEDIAN step reported (~10 s each on the GPU box's 16-core share)
        t0 = time.time()
        ol.update_step()
        times.append(time.time() - t0)
    med = sorted(times)[1]
    return dict(value=round(N * 128 / med, 1), unit="env-steps/s", cores=cores, kind="port",
                sample=f"median of 3 full update steps of CoordSum-4ag at num_envs={N} (rollout 128, 4 epochs x 2 minibatches), "
                       f"torch-CPU fp32 oracle, {cores} threads, {' / '.join(f'{t:.1f}' for t in times)} s per step")


def launch_ranks(n: int) -> int:
    """Start ``n`` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as a child of
    this GPU-free process; forward their stderr, relay rank 0's ONE JSON line from stdout.  Returns the exit status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    for line in proc.stdout:
        st = line.strip()
        if st.startswith("{") and st.endswith("}"):
            lines.append(st)
        elif st:
            print(st, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        print(f"[bench] a rank failed (torch.distributed.run exit status {rc})", file=sys.stderr, flush=True)
        return rc if 0 < rc < 256 else 1
    if len(lines) != 1:
        print(f"[bench] expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr, flush=True)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--num-envs", type=int, default=16384, help="envs per GPU (weak scaling)")
    ap.add_argument("--workload", default="coordsum-4ag", choices=["coordsum-4ag", "coordsum-8x15", "lbf-8x8-2p-2f", "rware-tiny-4ag"],
                    help="coordsum-4ag = BASELINE.json configs[1] (headline); coordsum-8x15 = configs[4] per GPU (registered 8x15-100, n_block=2, 8 minibatches); "
                         "lbf-8x8-2p-2f = configs[2] (Level-Based Foraging 8x8-2p-2f-coop) and rware-tiny-4ag = configs[3] per GPU (Robot Warehouse tiny-4ag, run it with "
                         "--num-envs 4096): UNPINNED dynamics, csrc/lbf.hip / csrc/rware.hip restate Jumanji's published algorithm")
    ap.add_argument("--micro-batches", type=int, default=0, help="train every minibatch in this many slabs with accumulated gradients (same update, "
                    "activations in HBM scale with the slab); 0 = the workload's default (1; coordsum-8x15: 4)")
    ap.add_argument("--embed-dim", type=int, default=64, help="Sable embed_dim (16 / 32 / 64 / 128); the headline is the reference default 64")
    ap.add_argument("--n-head", type=int, default=1)
    ap.add_argument("--n-block", type=int, default=0, help="0 = the workload's default (1; coordsum-8x15: 2)")
    ap.add_argument("--ppo-epochs", type=int, default=0, help="0 = the reference default 4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the fp32-MFMA / dense-first-layers variant measurements after the timed region")
    ap.add_argument("--overlap", action="store_true", help="run actor / weight-gradient kernels on side streams (experiment: -1 %% with the current kernels, which fill the chip; kernel timings then include contention)")
    ap.add_argument("--gru-split-bf16", type=int, default=2, choices=[0, 1, 2],
                    help="recurrent GEMMs of the GRU training scans: 2 (default) = forward scan on bf16 MFMA with operands split in three pieces (24 mantissa "
                    "bits, six products, fp32 accumulate: error against fp64 no larger than the fp32-MFMA scan's, test_gru_scan_bf16_triples_keep_fp32_accuracy), "
                    "0 = fp32 MFMA everywhere, 1 = pairs (16 bits, forward and backward; A/B only).  The JSON's dtype names the mode")
    ap.add_argument("--fp32-mfma", action="store_true", help="exact fp32 MFMA everywhere: --gru-split-bf16 0 and the 128 / 192-input dense layers on fp32 MFMA "
                    "instead of bf16 MFMA with three-piece operand splits")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--check-replicas", action="store_true", help="assert that parameters stayed identical on all ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without an outer launcher: this process has not touched the GPU (importing torch does not) and
        # never will -- it starts N fresh ranks under torch.distributed.run as a CHILD, relays rank 0's JSON line and exits with
        # the child's status.  (A process that has initialised the GPU must never exec / re-exec on this pool.)
        raise SystemExit(launch_ranks(args.gpus))

    from magpo_amd import distributed as mdist
    from magpo_amd._lib import lib
    from magpo_amd.learner import CoordSumConfig, LbfConfig, MagpoLearner, RwareConfig, SystemConfig, host_split, prng_key
    import torch.distributed as dist

    from magpo_amd.tuning import Tuning
    rank, world, local = mdist.init_from_env(args.backend)
    gru_split = int(args.gru_split_bf16)
    tuning = Tuning.from_env()
    if args.fp32_mfma:
        gru_split = 0
        tuning.linear_variant &= ~4
        tuning.actor_linear_variant &= ~4
    tuning.gru_split_bf16 = gru_split   # (the command line decides, whatever MAGPO_GRU_SPLIT_BF16 says)
    lin_bf3 = ([] if not tuning.actor_linear_variant & 4 else ["the GRU actor's 128-input dense layers"]) + \
              ([] if not tuning.linear_variant & 4 else ["the guider's 128 / 192-input dense layers"])
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)
    if world != args.gpus:
        if args.gpus != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    N = args.num_envs
    if args.workload == "coordsum-8x15":
        # 8 agents, two blocks: the reference's num_minibatches = 2 trained as 4 slabs per minibatch (R = 2.1 M rows per pass in HBM)
        sysc = SystemConfig(micro_batches=args.micro_batches or 4)
        env_cfg = CoordSumConfig(num_agents=8, num_actions=15, time_limit=100, maxval=100)
        n_block = 2
    elif args.workload == "rware-tiny-4ag":
        sysc = SystemConfig(micro_batches=args.micro_batches or 1)
        env_cfg = RwareConfig(column_height=8, shelf_rows=1, shelf_columns=3, num_agents=4, sensor_range=1, request_queue_size=4, time_limit=500)
        n_block = 1
    elif args.workload == "lbf-8x8-2p-2f":
        sysc = SystemConfig(micro_batches=args.micro_batches or 1)
        env_cfg = LbfConfig(grid_size=8, fov=8, num_agents=2, num_food=2, max_agent_level=2, force_coop=True, time_limit=100)
        n_block = 1
    else:
        sysc = SystemConfig(micro_batches=args.micro_batches or 1)  # reference defaults (configs/system/gpo/rec_magpo.yaml)
        env_cfg = CoordSumConfig(num_agents=4, num_actions=20, time_limit=100, maxval=60)
        n_block = 1
    n_block = args.n_block or n_block
    if args.ppo_epochs:
        sysc.ppo_epochs = args.ppo_epochs
    learner = MagpoLearner(env_cfg, N, sysc, dev, net_seed=0, n_block=n_block, n_head=args.n_head, embed_dim=args.embed_dim, tuning=tuning)  # same seed => replicated parameters on every rank
    key = host_split(prng_key(42), 4)[0]
    learner.setup(key, n_groups=world, group=rank)
    if args.overlap:
        learner.overlap_actor = True
        learner.guider.overlap_wgrad = learner.actor.overlap_wgrad = True
    grad_sync = mdist.make_grad_sync(world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # setup (untimed, not part of W): the first update step allocates every workspace, the second one captures the
    # 128-step rollout into a HIP graph; from then on every step does identical work
    for i in range(2):
        learner.update_step(grad_sync)
    torch.cuda.synchronize()
    log("setup done (workspaces allocated, rollout graph captured)")
    for i in range(args.warmup):
        learner.update_step(grad_sync)
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    timer = KernelTimer(min_rows=1 << 16)
    timer.attach_traffic = args.workload == "coordsum-4ag" and N == 16384  # the PMC passes were taken on that workload
    timing = not args.no_kernel_timing and rank == 0
    act_keys = ()
    eager = KernelTimer(min_rows=1 << 16)
    if timing:
        # One EAGER update step, still outside the timed region, with an event pair around EVERY C-ABI launch: (a) the acting and env kernels
        # run inside the rollout's HIP graph in the timed region, where per-launch events cannot be recorded (same kernels, shapes and data
        # distribution here); (b) the ~700 small launches per step without a cost model ("other: <entry point>") are timed here only, so
        # that the kernel table accounts for the whole step without costing the timed region an event pair per small launch.
        # Every rank runs this extra update step so that the replicas stay in lock-step.
        lib().timer, eager.enabled, eager.all_calls = eager, True, True
    learner.use_graph = False
    learner.update_step(grad_sync)
    learner.use_graph = True
    if timing:
        eager.collect()
        eager.enabled = False
    roll_events = []
    if timing:
        lib().timer, timer.enabled = timer, True
        orig_rollout = learner.rollout

        def timed_rollout():   # events around the graph replay(s) of the rollout (acting + env + actor carry + GAE)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); orig_rollout(); e1.record()
            roll_events.append((e0, e1))
        learner.rollout = timed_rollout
    # exposed gradient-exchange time per rank: events on the compute stream around the one all-reduce of a minibatch (the stream waits
    # for the collective before the optimiser kernel runs)
    ar_events = []
    sync = grad_sync
    if grad_sync is not None:
        def sync(l):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = grad_sync(l); e1.record()
            ar_events.append((e0, e1))
            return r
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        learner.update_step(sync)
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    lib().timer = None
    elapsed = reduce_max(elapsed)
    ar_ms = sum(e0.elapsed_time(e1) for e0, e1 in ar_events) / max(1, args.steps) if ar_events else 0.0
    ar_all = None
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        t[rank] = ar_ms
        dist.all_reduce(t)
        ar_all = [round(float(x), 3) for x in t.tolist()]
    if args.check_replicas and world > 1:
        cs = float(learner.guider.P.flat.double().sum().item() + learner.actor.P.flat.double().abs().sum().item())
        assert reduce_max(cs) == -reduce_max(-cs), "parameters diverged across ranks"
        log(f"replica check ok (checksum {cs:.9f}); env observations differ per rank: {float(learner.traj['obs'].double().sum().item()):.1f}")
    log(f"timed region done: {elapsed:.3f}s for {args.steps} steps")
    if rank == 0:
        timer.collect()
        # kernels that the timed region could not time (inside the rollout graph) or did not (no cost model): from the eager step
        act_keys = tuple(k for k in eager.rec if k not in timer.rec)
        for k in act_keys:
            timer.rec[k] = eager.rec[k]
        roof, table = timer.dominant(args.steps, act_keys)
        if table:
            for k, row in table.items():
                row["timed_in"] = "eager set-up step" if k in act_keys else "timed region"
                if k.startswith("other: "):
                    row["gbs"] = row["tflops"] = None
        rollout_ms = round(sum(e0.elapsed_time(e1) for e0, e1 in roll_events) / max(1, len(roll_events)), 2) if roll_events else None
        env_steps = world * N * sysc.rollout_length * args.steps
        if isinstance(env_cfg, RwareConfig):
            env_name = "RobotWarehouse"
            env_desc = ("Robot Warehouse tiny-4ag (11 x 10 grid, 32 shelves, sensor range 1, 75-wide observations, time_limit=500; UNPINNED dynamics "
                        "restated from Jumanji's published algorithm)")
        elif isinstance(env_cfg, LbfConfig):
            env_name = "LevelBasedForaging"
            env_desc = (f"Level-Based Foraging {env_cfg.grid_size}x{env_cfg.grid_size}-{env_cfg.num_agents}p-{env_cfg.num_food}f-coop (fov {env_cfg.fov}, "
                        "time_limit=100; UNPINNED dynamics restated from Jumanji's published algorithm)")
        else:
            env_name = "CoordSum"
            env_desc = f"CoordSum {env_cfg.num_agents}-agent (num_actions={env_cfg.num_actions}, maxval={env_cfg.maxval}, time_limit=100)"
        out = {
            "metric": f"env-steps/sec (all agents stepping jointly), {env_name}-{env_cfg.num_agents}ag, full MAGPO update loop",
            "value": round(env_steps / elapsed, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,   # from the initialised process group (RCCL / gloo), not from argv
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" + ("" if not (gru_split or lin_bf3) else " (fp32 MFMA; as bf16 x3 operand splits = 24 mantissa bits, six products, fp32 accumulate: "
                              + ", ".join(([] if gru_split != 2 else ["the GRU forward scan's recurrent GEMM"]) + lin_bf3)
                              + ("; GRU recurrent GEMMs as bf16 PAIRS = 16 mantissa bits" if gru_split == 1 else "") + ")"),
            "data": f"synthetic (fixed-seed {env_name} episodes, random-init networks)",
            "config": {"workload": f"{env_desc}, {N} envs/GPU x {world} GPU, rollout_length=128, ppo_epochs={sysc.ppo_epochs}, "
                                   f"num_minibatches={sysc.num_minibatches}" + (f" (each in {sysc.micro_batches} slabs, gradients accumulated)" if sysc.micro_batches > 1 else "")
                                   + f", Sable embed {args.embed_dim} / {args.n_head} head / {n_block} block, GRU 128",
                       "agent_steps_per_s": round(env_steps * env_cfg.num_agents / elapsed, 1),
                       "first_layer_class_tables": bool(learner.class_tables),   # exact (no caching across updates); MAGPO_CLASS_TABLES=0 = dense path
                       "parallelism": f"dp{world} (envs sharded, one flat grad all-reduce per minibatch)"},
            "roofline": roof,
            "rollout_graph_ms_per_step": rollout_ms,
            "hbm_resident_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),   # peak of torch's allocator on rank 0 (all buffers of the path)
            "kernel_table": table,
            # share of the step the table accounts for (launches through the C ABI; torch's own sort / searchsorted / copies are not in it)
            "kernel_table_ms_per_step": None if not table else round(sum(r["ms_per_step"] for r in table.values()), 2),
            "allreduce_wait_ms_per_step": ar_all,   # per rank, exposed on the compute stream (None for one rank)
        }
        if table:
            out["kernel_table_coverage"] = round(out["kernel_table_ms_per_step"] / out["ms_per_step"], 3)
        if world == 1 and not args.no_variants:
            # The same workload under the two switches that separate the headline from "plain": measured in this run, on this box, right
            # after the timed region (2 untimed + 3 timed update steps each).  fp32_mfma = exact fp32 MFMA everywhere (no bf16 x3 operand
            # splits); dense_first_layers = the first layers of both networks evaluated on every token row instead of on the distinct
            # (agent, target, step) rows (an exact algebraic rewrite that only CoordSum's finite observation alphabet allows).
            def variant(setup):
                try:
                    setup()
                    for g in learner.groups:   # the rollout graph holds the kernels of the previous mode
                        g.graph, g.warmed, g.graph_failed = None, False, False
                    for _ in range(3):
                        learner.update_step(grad_sync)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        learner.update_step(grad_sync)
                    torch.cuda.synchronize()
                    return round((time.perf_counter() - t1) / 3 * 1e3, 2)
                except Exception as e:   # a variant must never take the headline line down
                    return f"failed: {e!r}"
            variants = {}
            split0, lv0, alv0 = tuning.gru_split_bf16, tuning.linear_variant, tuning.actor_linear_variant
            if split0 or (lv0 | alv0) & 4:
                def fp32():
                    tuning.gru_split_bf16 = 0; tuning.linear_variant &= ~4; tuning.actor_linear_variant &= ~4
                variants["fp32_mfma_ms_per_step"] = variant(fp32)
                tuning.gru_split_bf16, tuning.linear_variant, tuning.actor_linear_variant = split0, lv0, alv0
            if learner.class_tables:
                def dense():
                    learner.class_tables = False
                variants["dense_first_layers_ms_per_step"] = variant(dense)
                learner.class_tables = True
            out["variants"] = variants
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle baseline ...")
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
