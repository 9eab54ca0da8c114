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
    # (the two groups train as one batch of sequences: same gradient up to fp32 summation order; Adam turns a relative gradient
    # difference d into an update difference of ~lr * d)
    assert torch.allclose(two.guider.P.flat, singles[0].guider.P.flat, atol=2e-6)
    assert torch.allclose(two.actor.P.flat, singles[0].actor.P.flat, atol=2e-6)
    assert mean.abs().max() > 0
    # group-by-group accumulation (batch_groups = False) gives the same update
    seq = MagpoLearner(cfg, 4, sysc, "cuda", net_seed=3, wgrad_groups=4, num_groups=2)
    seq.batch_groups = False
    seq.setup(key, n_groups=2, group=0)
    seq.rollout()
    seq.update()
    assert torch.allclose(two.guider.P.flat, seq.guider.P.flat, atol=2e-6) and torch.allclose(two.actor.P.flat, seq.actor.P.flat, atol=2e-6)


def _small_cfg(tmp_path, seed, extra=()):
    from magpo_amd.config import compose
    return compose("rec_magpo", ["env=coordsum", "env/scenario=3x10-30", "arch.num_envs=6", "arch.num_evaluation=2", "arch.num_eval_episodes=6",
                                 "system.total_timesteps=~", "system.num_updates=4", "system.rollout_length=12", "system.ppo_epochs=2",
                                 "system.update_batch_size=2", "env.kwargs.time_limit=7", f"system.seed={seed}", f"logger.base_exp_path={tmp_path}/",
                                 *extra])


def _setup(cfg):
    from magpo_amd.learner import host_split, prng_key
    from magpo_amd.systems.gpo.anakin import rec_magpo
    from magpo_amd.utils import make_env as environments
    from magpo_amd.utils.config import check_total_timesteps
    env, _ = environments.make(cfg)
    ks = host_split(prng_key(int(cfg.system.seed)), 4)
    learn, _, state = rec_magpo.learner_setup(env, (ks[0], ks[2], ks[3]), cfg, torch.device("cuda"))
    cfg = check_total_timesteps(cfg, 1)
    cfg.system.num_updates_per_eval = 1
    return learn, state


def _flat(state):
    out = [state.params.guider_params[k] for k in sorted(state.params.guider_params)] + [state.params.actor_params[k] for k in sorted(state.params.actor_params)]
    out += [state.opt_states.guider_opt_state["mu"], state.opt_states.actor_opt_state["nu"], state.hstates.policy_hidden_state,
            *state.hstates.sable_hidden_state, state.dones, state.timestep["agents_view"], *[state.env_state[k] for k in sorted(state.env_state)]]
    return [t.detach().cpu() for t in out]


def test_learn_is_a_function_of_its_state_and_checkpoints_resume(tmp_path):
    """LearnerFn contract (mava/types.py:207, rec_magpo.py:501-528): state in, state out.  (1) An OLD state can be passed again
    and gives the same successor; (2) a checkpoint (mava/utils/checkpointing.py:108-145) restored into a learner that was set
    up from a different seed continues bit-identically to the uninterrupted run: parameters, Adam moments, PRNG key, env state,
    observation / dones and both hidden states all travel."""
    from magpo_amd.utils.checkpointing import Checkpointer, restore_learner_state
    learn, s0 = _setup(_small_cfg(tmp_path, 42))
    s1 = learn(s0).learner_state
    s2 = learn(s1).learner_state
    ck = Checkpointer("rec_magpo", base_path=str(tmp_path), checkpoint_uid="resume")
    ck.save(2, s2, episode_return=1.0)
    s3 = learn(s2).learner_state
    # (1) rewind: the learner's buffers hold s3 now, learn(s1) must reproduce s2 exactly
    s2b = learn(s1).learner_state
    assert np.array_equal(s2b.key, s2.key)
    for a, b in zip(_flat(s2b), _flat(s2)):
        assert torch.equal(a, b)
    # (2) resume in a learner built from another seed (other parameters, env keys, step key)
    learn2, t0 = _setup(_small_cfg(tmp_path, 7))
    assert not torch.equal(_flat(t0)[0], _flat(s0)[0])
    restored, ts = restore_learner_state(os.path.join(tmp_path, "checkpoints", "rec_magpo", "resume", "2.pt"), "cuda")
    assert ts == 2
    r3 = learn2(restored).learner_state
    assert np.array_equal(r3.key, s3.key)
    assert r3.opt_states.guider_opt_state["count"] == s3.opt_states.guider_opt_state["count"] == 12
    for a, b in zip(_flat(r3), _flat(s3)):
        assert torch.equal(a, b), "resumed run differs from the uninterrupted one"


def test_bench_prints_one_json_line_with_the_contract_fields():
    """`python bench.py` (the driver's command, here at a small shape and without the CPU baseline leg): exactly one JSON line on
    stdout carrying the contract's fields, the roofline object of the dominant kernel and the workload description."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--num-envs", "256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"].startswith("f32")   # (+ the GRU forward scan's operand-split mode, named in brackets)
    assert d["unit"] == "env-steps/s" and d["value"] > 0 and abs(d["value"] - 256 * 128 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3


def test_lr_decay_uses_the_num_updates_check_total_timesteps_derives(tmp_path):
    """linear_scedule (mava/utils/training.py:37-43) reads config.system.num_updates when the learner is traced, i.e. AFTER
    check_total_timesteps rewrote it (rec_magpo.py:581 builds the optimiser before :717, but the closure holds the same config): with
    total_timesteps set the rate decays over the DERIVED number of updates -- never below zero, ~0 after the last update."""
    from magpo_amd.config import compose
    from magpo_amd.learner import MagpoLearner
    from magpo_amd.systems.gpo.anakin import rec_magpo
    n_env, T, U, updates = 4, 8, 2, 6
    cfg = compose("rec_magpo", ["env=coordsum", "env/scenario=3x10-30", f"arch.num_envs={n_env}", "arch.num_evaluation=2", "arch.num_eval_episodes=4",
                                "arch.absolute_metric=False", f"system.total_timesteps={n_env * T * U * updates}", f"system.rollout_length={T}",
                                "system.ppo_epochs=2", "system.decay_learning_rates=True", "env.kwargs.time_limit=5", f"logger.base_exp_path={tmp_path}/"])
    assert int(cfg.system.num_updates) != updates        # the composed default (1000) is NOT what the schedule may use
    lrs = []
    orig = MagpoLearner.apply_grads

    def spy(self, *a, **k):
        out = orig(self, *a, **k)
        lrs.append(self.last_lr)
        return out
    MagpoLearner.apply_grads = spy
    try:
        rec_magpo.run_experiment(cfg)
    finally:
        MagpoLearner.apply_grads = orig
    per_update = 2 * int(cfg.system.num_minibatches)
    assert len(lrs) == updates * per_update
    base = float(cfg.system.actor_lr)
    want = [base * (1.0 - (i // per_update) / updates) for i in range(len(lrs))]
    assert np.allclose(lrs, want, rtol=1e-12, atol=0) and min(lrs) > 0 and abs(lrs[-1] - base / updates) < 1e-12


def test_get_learner_fn_calls_the_functions_it_is_given_and_rejects_foreign_callables():
    """rec_magpo.py:91-100: get_learner_fn(env, (sable_action_select_fn, sable_apply_fn, actor_apply_fn), (sable_update_fn, actor_update_fn),
    config).  Here the five callables must be (adaptors around) methods of the objects that own the device buffers: (1) thin adaptors
    are accepted and are what the loop CALLS (rollout, both training forwards, both optimiser steps), with the same result as the bound
    methods; (2) a free function raises a TypeError that states the requirement."""
    import functools
    from magpo_amd.actor import GruActor
    from magpo_amd.config import compose
    from magpo_amd.learner import host_split, prng_key
    from magpo_amd.optim import ClipAdam
    from magpo_amd.sable import SableGuider
    from magpo_amd.systems.gpo.anakin import rec_magpo
    from magpo_amd.utils import make_env as environments
    cfg = compose("rec_magpo", ["env=coordsum", "env/scenario=3x10-30", "arch.num_envs=6", "system.total_timesteps=~", "system.num_updates=2",
                                "system.rollout_length=8", "system.ppo_epochs=2", "env.kwargs.time_limit=5"])
    cfg.system.num_updates_per_eval = 1
    env, _ = environments.make(cfg)
    sysc = rec_magpo._system_config(cfg)
    key = host_split(prng_key(3), 4)[0]

    def build(adapt):
        g = SableGuider(env.cfg.num_agents, env.cfg.num_actions, env.cfg.obs_dim, "cuda", max_pos=env.cfg.time_limit + 1, seed=5)
        a = GruActor(env.cfg.num_agents, env.cfg.num_actions, env.cfg.obs_dim, "cuda", seed=6, tuning=g.tuning)
        go, ao = ClipAdam(g, sysc), ClipAdam(a, sysc)
        fns = [g.get_actions, g.apply, a.apply, go.update, ao.update]
        calls = [0] * 5
        if adapt:
            def wrap(i, f):
                @functools.wraps(f)
                def w(*args, **kw):
                    calls[i] += 1
                    return f(*args, **kw)
                return w
            fns = [wrap(i, f) for i, f in enumerate(fns)]
            fns[3] = functools.partial(fns[3])   # a partial around a wrapper around the bound method
        learn = rec_magpo.get_learner_fn(env, tuple(fns[:3]), tuple(fns[3:]), cfg)
        learn.learner.setup(key, n_groups=len(learn.learner.groups))   # update_batch_size = 2 groups (configs/system/gpo/rec_magpo.yaml)
        learn.learner._live_state = rec_magpo._snapshot_state(learn.learner)
        out = learn(learn.learner._live_state)
        return out, calls

    plain, _ = build(False)
    adapted, calls = build(True)
    T, P, M, U = 8, 2, int(cfg.system.num_minibatches), int(cfg.system.update_batch_size)
    assert calls[0] == U * (T + 1), "the rollout of every env group must call the execution function it was given (T steps + the bootstrap value)"
    assert calls[1] == P * M and calls[2] == P * M and calls[4] == P * M, calls
    for k, v in plain.learner_state.params.guider_params.items():
        assert torch.equal(v, adapted.learner_state.params.guider_params[k]), k
    g = SableGuider(env.cfg.num_agents, env.cfg.num_actions, env.cfg.obs_dim, "cuda", max_pos=env.cfg.time_limit + 1, seed=5)
    a = GruActor(env.cfg.num_agents, env.cfg.num_actions, env.cfg.obs_dim, "cuda", seed=6, tuning=g.tuning)
    go, ao = ClipAdam(g, sysc), ClipAdam(a, sysc)
    with pytest.raises(TypeError, match="bound method SableGuider.apply"):
        rec_magpo.get_learner_fn(env, (g.get_actions, lambda *x, **k: None, a.apply), (go.update, ao.update), cfg)
    with pytest.raises(TypeError, match="bound method ClipAdam.update"):
        rec_magpo.get_learner_fn(env, (g.get_actions, g.apply, a.apply), (print, ao.update), cfg)
    with pytest.raises(TypeError, match="GruActor.apply"):
        rec_magpo.get_learner_fn(env, (g.get_actions, g.apply, g.apply), (go.update, ao.update), cfg)
