"""The rec_magpo entry point end to end on the GPU: config compose -> learner_setup -> learn -> evaluator -> logs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_run_experiment_small(tmp_path):
    from magpo_amd.config import compose
    from magpo_amd.systems.gpo.anakin import rec_magpo
    cfg = compose("rec_magpo", ["env=coordsum", "env/scenario=3x10-30", "arch.num_envs=8", "arch.num_evaluation=2",
                                "arch.num_eval_episodes=8", "arch.num_absolute_metric_eval_episodes=16", "system.total_timesteps=~",
                                "system.num_updates=4", "system.rollout_length=16", "system.ppo_epochs=2", "env.kwargs.time_limit=10",
                                "logger.loggers.json.enabled=True", f"logger.base_exp_path={tmp_path}/", "logger.loggers.json.path=run",
                                "logger.checkpointing.save_model=True", "network.memory_config.timestep_chunk_size=4",
                                "network.net_config.n_block=2"])
    perf = rec_magpo.run_experiment(cfg)
    assert np.isfinite(perf) and 0.0 <= perf <= 20.0
    data = json.load(open(os.path.join(tmp_path, "json", "run", "metrics.json")))
    run = data["CoordSum"]["3x10-30-v0"]["rec_magpo"]["seed_42"]
    assert "step_0" in run and "step_1" in run and "absolute_metrics" in run
    assert "mean_episode_return" in run["step_0"] and "steps_per_second" in run["step_0"]
    ckdir = os.path.join(tmp_path, "checkpoints", "rec_magpo")
    pts = [f for d in os.listdir(ckdir) for f in os.listdir(os.path.join(ckdir, d)) if f.endswith(".pt")]
    assert len(pts) == 1  # max_to_keep: 1


def test_two_groups_share_parameters_and_average_gradients():
    """update_batch_size = 2: two env groups, one parameter set, gradient = mean over groups (rec_magpo.py:395-397)."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    sysc = SystemConfig(rollout_length=8, ppo_epochs=1, num_minibatches=1)
    cfg = CoordSumConfig(3, 10, 6, 30)
    key = host_split(prng_key(1), 4)[0]
    two = MagpoLearner(cfg, 4, sysc, "cuda", net_seed=3, wgrad_groups=4, num_groups=2)
    two.setup(key, n_groups=2, group=0)
    singles = []
    for gi in range(2):
        s = MagpoLearner(cfg, 4, sysc, "cuda", net_seed=3, wgrad_groups=4)
        s.setup(key, n_groups=2, group=gi)
        singles.append(s)
    two.rollout()
    for gi, s in enumerate(singles):
        s.rollout()
        assert torch.equal(s.traj["action"], two.groups[gi].traj["action"])
    assert not torch.equal(two.groups[0].traj["obs"], two.groups[1].traj["obs"])
    ks = host_split(two.key, 4)
    bp, ap = two._permutation(ks[1], 4), two._permutation(ks[2], 3)
    grads = []
    for s in singles:
        s.minibatch_grads(bp, ap)
        grads.append(s.grad_all.clone())
    mean = (grads[0] + grads[1]) / 2
    two.update()
    singles[0].grad_all.copy_(grads[0] + grads[1])
    singles[0].apply_grads(0.5)
    assert torch.allclose(two.guider.P.flat, singles[0].guider.P.flat, atol=1e-7)
    assert torch.allclose(two.actor.P.flat, singles[0].actor.P.flat, atol=1e-7)
    assert mean.abs().max() > 0
