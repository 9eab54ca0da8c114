"""N > 1 path on the GPU box: two processes (one env group each, torch.distributed) must end with the parameters of a
single process that owns both groups (update_batch_size = 2) -- the gradient mean of rec_magpo.py:395-409 is the only
coupling.  One GPU is available to the tests, so the two ranks share it and the all-reduce runs over gloo (staged
through host memory, magpo_amd/distributed.py); on a multi-GPU node the same code path runs over RCCL."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(A=3, K=10, TL=6, maxval=30, N=6, T=8, P=2, M=2, steps=2, seed=31)


def _mk(num_groups):
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig
    c = CFG
    return MagpoLearner(CoordSumConfig(c["A"], c["K"], c["TL"], c["maxval"]), c["N"],
                        SystemConfig(rollout_length=c["T"], ppo_epochs=c["P"], num_minibatches=c["M"]), "cuda", net_seed=5, wgrad_groups=4,
                        num_groups=num_groups)


def _key():
    from magpo_amd.learner import host_split, prng_key
    return host_split(prng_key(CFG["seed"]), 4)[0]


def _rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from magpo_amd import distributed as mdist
    mdist.init_from_env("gloo")
    l = _mk(1)
    l.setup(_key(), n_groups=world, group=rank)
    sync = mdist.make_grad_sync(world)
    losses = [l.update_step(sync).cpu() for _ in range(CFG["steps"])]
    torch.cuda.synchronize()
    q.put((rank, l.guider.P.flat.cpu().numpy(), l.actor.P.flat.cpu().numpy(), torch.stack(losses).numpy(), l.traj["action"].cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_with_two_groups():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = _mk(2)
    one.setup(_key(), n_groups=2, group=0)
    losses = torch.stack([one.update_step().cpu() for _ in range(CFG["steps"])]).numpy()
    gp, ap = one.guider.P.flat.cpu().numpy(), one.actor.P.flat.cpu().numpy()
    # replicas stay identical without a broadcast (identical optimiser step on identical averaged gradients)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    # ... and equal the single-process two-group learner (summation order of the two gradient halves may differ)
    assert np.allclose(res[0][1], gp, rtol=0, atol=2e-6), float(np.abs(res[0][1] - gp).max())
    assert np.allclose(res[0][2], ap, rtol=0, atol=2e-6), float(np.abs(res[0][2] - ap).max())
    assert np.allclose(res[0][3], losses, rtol=1e-4, atol=1e-6)
    for r in range(2):   # rank r rolled out group r's envs
        assert np.array_equal(res[r][4], one.groups[r].traj["action"].cpu().numpy())
    assert not np.array_equal(res[0][4], res[1][4])


def _rccl_worker(port, q):
    """RCCL itself (backend "nccl" on ROCm) on the one GPU of the box: a single-rank group, the same all_reduce call on the same
    flat gradient buffer and stream ordering as the multi-GPU path."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=0, world_size=1)
    calls = []

    def sync(learner):   # what distributed.make_grad_sync returns for world > 1, forced here for world = 1
        before = learner.grad_all.clone()
        dist.all_reduce(learner.grad_all, op=dist.ReduceOp.SUM)
        calls.append(bool(torch.equal(before, learner.grad_all)))
        return 1.0

    a, b = _mk(1), _mk(1)
    a.setup(_key()); b.setup(_key())
    for _ in range(2):
        a.update_step(sync)
        b.update_step(None)
    torch.cuda.synchronize()
    q.put((len(calls), all(calls), bool(torch.equal(a.guider.P.flat, b.guider.P.flat) and torch.equal(a.actor.P.flat, b.actor.P.flat))))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_all_reduce_on_the_gradient_buffer():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29900 + os.getpid() % 90, q))
    p.start()
    ncalls, identity, same = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert ncalls == 2 * CFG["P"] * CFG["M"] and identity and same


def _resume_worker(rank, world, port, tmp, q):
    """Two ranks: learn, checkpoint (rank-aware), learn again; then a fresh job built from ANOTHER seed restores the checkpoint and
    must continue exactly like the uninterrupted one -- with every rank on its OWN envs (a resume that hands rank 0's
    rollout state to every rank shrinks the effective batch silently)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from magpo_amd import distributed as mdist
    from magpo_amd.config import compose
    from magpo_amd.learner import host_split, prng_key
    from magpo_amd.systems.gpo.anakin import rec_magpo
    from magpo_amd.utils import make_env as environments
    from magpo_amd.utils.checkpointing import Checkpointer, latest_valid_checkpoint, restore_learner_state
    from magpo_amd.utils.config import check_total_timesteps
    mdist.init_from_env("gloo")

    def setup(seed):
        cfg = compose("rec_magpo", ["env=coordsum", "env/scenario=3x10-30", "arch.num_envs=6", "system.total_timesteps=~", "system.num_updates=4",
                                    "system.rollout_length=8", "system.ppo_epochs=2", "env.kwargs.time_limit=5", f"system.seed={seed}",
                                    f"logger.base_exp_path={tmp}/"])
        env, _ = environments.make(cfg)
        ks = host_split(prng_key(seed), 4)
        learn, _, state = rec_magpo.learner_setup(env, (ks[0], ks[2], ks[3]), cfg, torch.device("cuda"), rank, world)
        cfg = check_total_timesteps(cfg, world)
        cfg.system.num_updates_per_eval = 1
        return learn, state

    learn, s0 = setup(42)
    s1 = learn(s0).learner_state
    ck = Checkpointer("rec_magpo", base_path=tmp, checkpoint_uid="mr", rank=rank, world=world)
    ck.save(1, s1, episode_return=0.0)
    dist.barrier()
    s2 = learn(s1).learner_state
    snap = lambda l: np.concatenate([l.traj["action"].cpu().numpy().reshape(-1).astype(np.float32), l.traj["obs"].cpu().numpy().reshape(-1)])
    act_ref = snap(learn.learner)
    learn2, _ = setup(7)
    path = latest_valid_checkpoint(os.path.join(tmp, "checkpoints", "rec_magpo", "mr"), rank, world)
    restored, _ = restore_learner_state(path, "cuda", rank, world)
    r2 = learn2(restored).learner_state
    act_res = snap(learn2.learner)   # actions and observations of the rank's first group over the whole rollout
    same = all(torch.equal(a, b) for a, b in zip(r2.params.guider_params.values(), s2.params.guider_params.values())) \
        and torch.equal(r2.env_state["target"], s2.env_state["target"]) and np.array_equal(r2.key, s2.key)
    q.put((rank, act_ref, act_res, bool(same)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_resume_keeps_every_ranks_own_rollout_state(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 90
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=400) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.array_equal(res[r][1], res[r][2]), f"rank {r}: resumed rollout differs from the uninterrupted one"
        assert res[r][3], f"rank {r}: resumed state differs"
    # (with a near-uniform initial policy and the shared step key the sampled ACTIONS of two groups can coincide: compare the observations too)
    assert not np.array_equal(res[0][2], res[1][2]), "after the resume both ranks roll out the same envs"
