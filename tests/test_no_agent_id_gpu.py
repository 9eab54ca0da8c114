"""``system.add_agent_id: False`` (configs/system/gpo/rec_magpo.yaml:9; make_env.py:90-104: the AgentIDWrapper is not applied): the
networks are built for the observation WITHOUT the one-hot agent id.  The env kernels keep writing ``[id | features]`` rows; every
network-side consumer reads them through a pointer advanced behind the id with the row stride unchanged (magpo_amd/learner.py:net_obs).
Narrow observations (CoordSum, Level-Based Foraging); Robot Warehouse raises.  Checked against the oracle built without the wrapper:
rollout (actions bit-exact), minibatch gradients, a full update, the evaluator, and the entry point end to end."""
import numpy as np
import pytest
import torch

from oracle import coordsum as ocs
from oracle import evaluator as oeval
from oracle import lbf as olbf
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} (ref scale {ref:.3e})"


def _pair(kind, N, T, nb=1):
    from magpo_amd.learner import CoordSumConfig, LbfConfig, MagpoLearner, SystemConfig
    if kind == "coordsum":
        spec, cfg, env = ocs.CoordSumSpec(3, 10, 7, 30), CoordSumConfig(3, 10, 7, 30, add_agent_id=False), ocs
    else:
        args = (8, 8, 2, 2, 2, True, 12)
        spec, cfg, env = olbf.LbfSpec(*args), LbfConfig(*args, add_agent_id=False), olbf
    spec.add_agent_id = False
    A, K, F = spec.num_agents, spec.num_actions, spec.obs_dim
    assert F == cfg.obs_dim - A
    gp = onets.init_guider_params(1, 64, F, K, nb=nb)
    ap = onets.init_actor_params(2, F, 128, K)
    gp["dec.head.dense1.kernel"] = gp["dec.head.dense1.kernel"] * 30   # logits with a visible spread
    ap["head.kernel"] = ap["head.kernel"] * 30
    osys = olearn.SystemCfg(rollout_length=T, ppo_epochs=2, num_minibatches=2)
    ol = olearn.OracleLearner(spec, N, osys, onets.SableCfg(A, K, F, n_block=nb), gp, ap, env=env)
    key = oprng.split(oprng.prng_key(31), 4)[0]
    ol.setup(key)
    dl = MagpoLearner(cfg, N, SystemConfig(rollout_length=T, ppo_epochs=2, num_minibatches=2), "cuda", net_seed=None, wgrad_groups=4, n_block=nb)
    assert dl.F == F and dl.obs_off == A and dl.Fld == cfg.obs_dim and not dl.class_tables
    dl.guider.load_named(gp); dl.actor.load_named(ap)
    dl.setup(key)
    return ol, dl, A, F


@pytest.mark.parametrize("kind,N,T,nb", [("coordsum", 8, 12, 1), ("coordsum", 6, 11, 2), ("lbf", 8, 16, 1)])
def test_learner_parity_without_agent_ids(kind, N, T, nb):
    ol, dl, A, F = _pair(kind, N, T, nb)
    for step in range(2):
        om = ol.rollout()
        dl.rollout()
        tr, otr = dl.traj, ol.traj
        assert np.array_equal(tr["action"].cpu().numpy(), otr["action"].numpy()), f"update step {step}: sampled actions differ"
        # the env rows keep their id columns; the oracle's observation is the part behind them
        assert np.array_equal(tr["obs"][:T, :, :, A:A + F].cpu().numpy(), otr["obs"].numpy().astype(np.float32))
        assert np.array_equal(tr["obs"][:T, :, :, :A].cpu().numpy(), np.broadcast_to(np.eye(A, dtype=np.float32), (T, N, A, A)))
        assert np.array_equal(tr["reward"].cpu().numpy(), otr["reward"].numpy())
        _close(tr["value"], otr["value"], 1e-4, 1e-6, "value")
        _close(tr["log_prob"], otr["log_prob"], 1e-4, 1e-6, "log_prob")
        _close(dl.policy_h[dl._cur], ol.policy_h.reshape(N * A, 128), 1e-4, 1e-6, "policy hidden")
        assert om["is_terminal_step"].any()
        if step == 0:
            ks = oprng.split(ol.key, 4)
            bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
            gg, ag, info, inter = ol.minibatch_grads(ol.make_minibatches(bp, apm)[1])
            dl.minibatch_grads(dl._permutation(ks[1], N)[N // 2:].contiguous(), dl._permutation(ks[2], A))
            for n, g in dl.guider.named_grads.items():
                scale = max(gg[n].abs().max().item(), 1e-6)
                _close(g / scale, gg[n].reshape(g.shape) / scale, 0, 2e-3, f"guider grad {n}")
            for n, g in dl.actor.named_grads.items():
                scale = max(ag[n].abs().max().item(), 1e-6)
                _close(g / scale, ag[n].reshape(g.shape) / scale, 0, 2e-3, f"actor grad {n}")
        ol.update()
        dl.update()
        dl._carry_over()
        assert np.array_equal(dl.key, ol.key)
        for net, ref in ((dl.guider, ol.gp), (dl.actor, ol.ap)):
            for n, v in net.named.items():
                _close(v, ref[n].reshape(v.shape), 0, 3e-5 * (step + 1), f"param {n} (update step {step})")


def test_entry_point_and_evaluator_without_agent_ids(tmp_path):
    from magpo_amd.actor import GruActor
    from magpo_amd.config import compose
    from magpo_amd.evaluator import get_eval_fn, get_num_eval_envs, make_rec_eval_act_fn
    from magpo_amd.learner import obs_row_stride
    from magpo_amd.systems.gpo.anakin import rec_magpo
    from magpo_amd.utils import make_env as environments
    cfg = compose("rec_magpo", ["env=coordsum", "env/scenario=3x10-30", "system.add_agent_id=False", "arch.num_envs=5", "arch.num_eval_episodes=10",
                                "env.kwargs.time_limit=8"])
    env, eval_env = environments.make(cfg)
    A, K = env.num_agents, env.action_dim
    assert env.obs_dim == 1 and env.observation_spec.agents_view.shape == (A, 1)
    state, ts = env.reset(oprng.split(oprng.prng_key(3), 5))
    assert ts.observation.agents_view.shape == (5, A, 1) and ts.observation.agents_view.stride(1) == A + 1
    ap = onets.init_actor_params(17, 1, 128, K)
    ap["head.kernel"] = ap["head.kernel"] * 60
    actor = GruActor(A, K, env.obs_dim, "cuda", obs_ld=obs_row_stride(env.cfg.obs_dim))
    evaluator = get_eval_fn(eval_env, make_rec_eval_act_fn(actor, cfg), cfg, absolute_metric=False, device="cuda")
    n = get_num_eval_envs(cfg, False)
    key = oprng.split(oprng.prng_key(23), 3)[1]
    got = evaluator({k: v.cuda() for k, v in ap.items()}, key, {"hidden_state": torch.zeros(n * A, 128, device="cuda")})
    spec = ocs.CoordSumSpec(A, K, 8, env.cfg.maxval)
    spec.add_agent_id = False
    want = oeval.evaluate(spec, ap, key, 5, 10)
    assert np.array_equal(got["episode_length"], want["episode_length"]) and np.array_equal(got["episode_return"], want["episode_return"])
    # the entry point end to end (CoordSum and LBF), and the wide-observation env refusing the switch
    for extra in (["env=coordsum", "env/scenario=3x10-30", "env.kwargs.time_limit=8"], ["env=lbf", "env/scenario=8x8-2p-2f-coop", "env.kwargs.time_limit=12"]):
        c = compose("rec_magpo", extra + ["system.add_agent_id=False", "arch.num_envs=8", "arch.num_evaluation=2", "arch.num_eval_episodes=8",
                                          "arch.num_absolute_metric_eval_episodes=16", "system.total_timesteps=~", "system.num_updates=4",
                                          "system.rollout_length=16", "system.ppo_epochs=2", f"logger.base_exp_path={tmp_path}/"])
        assert np.isfinite(rec_magpo.run_experiment(c))
    with pytest.raises(NotImplementedError, match="Robot Warehouse"):
        environments.make(compose("rec_magpo", ["env=rware", "env/scenario=tiny-2ag", "system.add_agent_id=False"]))
