#!/usr/bin/env python3
"""BASELINE metric 2: MAGPO return curves on the four registered CoordSum scenarios with the tuned hyper-parameters of the
reference (experiment_data/params.csv:61-64, MAGPO rows: num_envs 64, num_updates 1220, num_evaluation 122 -- with the
default update_batch_size 2 and rollout_length 128 that is the 20 M steps of configs/system/gpo/rec_magpo.yaml:3; the
evaluation at 10 M steps is interval 61 of 122), several seeds in parallel processes on one GPU.

    python scripts/run_sweep.py --out gpurun_out/sweep --scenarios 3x10-30 8x15-100 --seeds 0 1 2 3 4

Writes <out>/json/<scenario>_s<seed>/metrics.json (marl-eval layout, magpo_amd/utils/logger.py) + <out>/<scenario>_s<seed>.log.
The reference's own curves (experiment_data.json) are not available offline: the summary (scripts/sweep_summary.py) reports
ours only."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# experiment_data/params.csv:61-64 (columns: num_minibatches, max_grad_norm, n_embd, ppo_epochs, clip_eps, decay_scaling_factor,
# n_head, ent_coef, n_block, lr (critic_lr column; MAGPO has one learning rate, system.actor_lr), alpha, delta = clip_gpo)
TUNED = {
    "3x10-30": dict(M=8, mgn=0.5, E=32, P=10, clip=0.05, dsf=1.0, nh=1, ent=0.01, nb=2, lr=0.00025, alpha=2, delta=1.3),
    "3x30-50": dict(M=4, mgn=0.5, E=64, P=15, clip=0.1, dsf=1.0, nh=1, ent=0.01, nb=1, lr=0.0005, alpha=8, delta=1.2),
    "5x20-80": dict(M=8, mgn=0.5, E=64, P=15, clip=0.2, dsf=0.5, nh=2, ent=0.01, nb=2, lr=0.0005, alpha=4, delta=1.1),
    "8x15-100": dict(M=8, mgn=0.5, E=64, P=15, clip=0.2, dsf=1.0, nh=1, ent=0.01, nb=2, lr=0.001, alpha=8, delta=1.3),
    # params.csv:88,98 (LBF rows; lr = actor_lr column) and :83 (RWARE tiny-4ag: n_embd 128, n_head 2, n_block 3)
    "8x8-2p-2f-coop": dict(env="lbf", M=4, mgn=10, E=32, P=15, clip=0.2, dsf=0.3, nh=4, ent=0.001, nb=2, lr=0.00025, alpha=2, delta=1.5),
    "2s-8x8-2p-2f-coop": dict(env="lbf", M=4, mgn=0.5, E=32, P=5, clip=0.2, dsf=0.3, nh=4, ent=0.001, nb=2, lr=0.0005, alpha=2, delta=1.5),
    "tiny-4ag": dict(env="rware", M=2, mgn=0.5, E=128, P=5, clip=0.2, dsf=0.5, nh=2, ent=0.01, nb=3, lr=0.0005, alpha=8, delta=1.3),
}


def overrides(scen, seed, out, num_updates, num_evaluation, resumable=False):
    h = TUNED[scen]
    extra = []
    if resumable:   # checkpoint the full learner + loop state at every evaluation; continue from it when one exists
        uid = f"{scen}_s{seed}"
        have = os.path.isdir(os.path.join(out, "checkpoints", "rec_magpo", uid)) and any(
            f.endswith(".pt") for f in os.listdir(os.path.join(out, "checkpoints", "rec_magpo", uid)))
        extra = ["logger.checkpointing.save_model=True", "logger.checkpointing.save_args.keep_latest=True",
                 f"logger.checkpointing.save_args.checkpoint_uid={uid}", f"logger.checkpointing.load_model={have}",
                 f"logger.checkpointing.load_args.checkpoint_uid={uid}"]
    return extra + [f"env={h.get('env', 'coordsum')}", f"env/scenario={scen}", "arch.num_envs=64", f"arch.num_evaluation={num_evaluation}", "system.total_timesteps=~",
            f"system.num_updates={num_updates}", f"system.seed={seed}", f"system.num_minibatches={h['M']}", f"system.max_grad_norm={h['mgn']}",
            f"system.ppo_epochs={h['P']}", f"system.clip_eps={h['clip']}", f"system.ent_coef={h['ent']}", f"system.actor_lr={h['lr']}",
            f"system.alpha={h['alpha']}", f"system.clip_gpo={h['delta']}", f"network.net_config.embed_dim={h['E']}",
            f"network.net_config.n_head={h['nh']}", f"network.net_config.n_block={h['nb']}",
            f"network.memory_config.decay_scaling_factor={h['dsf']}", "logger.loggers.json.enabled=True",
            f"logger.base_exp_path={out}/", f"logger.loggers.json.path={scen}_s{seed}"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sweep")
    ap.add_argument("--scenarios", nargs="+", default=list(TUNED))
    ap.add_argument("--seeds", nargs="+", type=int, default=[0, 1, 2, 3, 4])
    ap.add_argument("--num-updates", type=int, default=1220)
    ap.add_argument("--num-evaluation", type=int, default=122)
    ap.add_argument("--jobs", nargs="+", default=None, help="explicit job list scenario:seed (overrides --scenarios / --seeds)")
    ap.add_argument("--resumable", action="store_true", help="checkpoint every evaluation and continue from the latest checkpoint under --out")
    ap.add_argument("--parallel", type=int, default=5, help="runs at once on the GPU (the box allows 6 GPU processes)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    jobs = [(s, seed) for s in args.scenarios for seed in args.seeds]
    if args.jobs:
        jobs = [(j.rsplit(":", 1)[0], int(j.rsplit(":", 1)[1])) for j in args.jobs]
    running = []
    t0 = time.time()
    while jobs or running:
        while jobs and len(running) < args.parallel:
            scen, seed = jobs.pop(0)
            log = open(os.path.join(args.out, f"{scen}_s{seed}.log"), "w")
            cmd = [sys.executable, "-m", "magpo_amd.systems.gpo.anakin.rec_magpo", *overrides(scen, seed, args.out, args.num_updates, args.num_evaluation, args.resumable)]
            running.append((subprocess.Popen(cmd, cwd=ROOT, stdout=log, stderr=subprocess.STDOUT), scen, seed, time.time()))
        time.sleep(5)
        for r in list(running):
            if r[0].poll() is not None:
                running.remove(r)
                print(f"[sweep] {r[1]} seed {r[2]} finished rc={r[0].returncode} in {time.time() - r[3]:.0f}s (elapsed {time.time() - t0:.0f}s)", flush=True)
                if args.resumable and r[0].returncode == 0:   # a finished run needs no checkpoint any more
                    import shutil
                    shutil.rmtree(os.path.join(args.out, "checkpoints", "rec_magpo", f"{r[1]}_s{r[2]}"), ignore_errors=True)
        if int(time.time() - t0) % 60 < 5:
            print(f"[sweep] {len(running)} running, {len(jobs)} queued, {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
