"""Level-Based Foraging on the GPU (csrc/lbf.hip) against oracle/lbf.py -- both restate Jumanji's published algorithm (UNPINNED
dynamics) and must agree bit for bit: env state, observations, action masks, rewards, episode metrics, auto-reset; then the
MAGPO learner (masked sampling, masked losses) and the evaluator on top of it."""
import numpy as np
import pytest
import torch

from oracle import evaluator as oeval
from oracle import lbf as olbf
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,fov,A,NF,maxl,coop,TL,N", [(8, 8, 2, 2, 2, True, 25, 70), (8, 2, 2, 2, 2, True, 30, 33), (10, 10, 3, 3, 3, False, 40, 65),
                                                       (15, 5, 4, 5, 2, False, 35, 20)])
def test_lbf_env_matches_oracle(G, fov, A, NF, maxl, coop, TL, N):
    from magpo_amd.learner import LbfConfig, LbfEnvBatch
    spec = olbf.LbfSpec(G, fov, A, NF, maxl, coop, TL)
    cfg = LbfConfig(G, fov, A, NF, maxl, coop, TL)
    keys = oprng.split(oprng.prng_key(G * 100 + A), N)
    st, ts = olbf.reset(spec, keys)
    env = LbfEnvBatch(cfg, N, "cuda")
    F = cfg.obs_dim
    obs, obs_step = torch.zeros(N, A, F, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda")
    mask = torch.zeros(N, A, 6, dtype=torch.uint8, device="cuda")
    reward, done = torch.zeros(N, A, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda")
    m_ret, m_len, m_term = torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda")
    env.reset(torch.from_numpy(keys.view(np.int32)).cuda(), obs, obs_step, mask)

    def check(tag):
        for f in ("agent_pos", "agent_level", "food_pos", "food_level", "step_count"):
            assert np.array_equal(getattr(env, f).cpu().numpy(), st[f]), (tag, f)
        assert np.array_equal(env.food_eaten.cpu().numpy().astype(bool), st["food_eaten"]), tag
        assert np.array_equal(env.key.cpu().numpy().view(np.uint32), st["key"]), tag
        assert np.array_equal(obs.cpu().numpy(), ts["observation"]["agents_view"]), tag
        assert np.array_equal(mask.cpu().numpy().astype(bool), ts["observation"]["action_mask"]), tag
        assert np.array_equal(obs_step.cpu().numpy(), ts["observation"]["step_count"][:, 0]), tag
    check("reset")
    rng = np.random.default_rng(3)
    eaten_any, resets = False, 0
    for t in range(3 * TL):
        m = ts["observation"]["action_mask"]
        a = np.zeros((N, A), np.int32)
        for n in range(N):
            for i in range(A):
                legal = np.nonzero(m[n, i])[0]
                # prefer LOAD when legal (so that food gets eaten), else a random legal move
                a[n, i] = 5 if (m[n, i, 5] and rng.random() < 0.7) else rng.choice(legal)
        st, ts = olbf.step(spec, st, a, auto_reset=True)
        env.step(torch.from_numpy(a).cuda(), reward, done, obs, obs_step, m_ret, m_len, m_term, auto_reset=True, mask=mask)
        check(t)
        assert np.array_equal(reward.cpu().numpy(), ts["reward"]), t
        assert np.array_equal(done.cpu().numpy().astype(bool), ts["step_type"] == olbf.STEP_LAST), t
        assert np.array_equal(m_ret.cpu().numpy(), ts["episode_metrics"]["episode_return"]), t
        assert np.array_equal(m_len.cpu().numpy(), ts["episode_metrics"]["episode_length"]), t
        eaten_any |= bool((ts["reward"] > 0).any())
        resets += int((ts["step_type"] == olbf.STEP_LAST).sum())
    assert eaten_any and resets > N, "the test must see food eaten and episodes ending"


def _mk(cfg_args, N, T, P=2, M=2, seed=5):
    from magpo_amd.learner import LbfConfig, MagpoLearner, SystemConfig
    spec = olbf.LbfSpec(*cfg_args)
    cfg = LbfConfig(*cfg_args)
    A, K, F = spec.num_agents, 6, spec.obs_dim
    scfg = onets.SableCfg(A, K, F)
    gp = onets.init_guider_params(1, 64, F, K)
    ap = onets.init_actor_params(2, F, 128, K)
    # logits with a visible spread: at init the head is ~uniform and masked sampling would hardly be exercised
    gp["dec.head.dense1.kernel"] = gp["dec.head.dense1.kernel"] * 30
    ap["head.kernel"] = ap["head.kernel"] * 30
    ol = olearn.OracleLearner(spec, N, olearn.SystemCfg(rollout_length=T, ppo_epochs=P, num_minibatches=M), scfg, gp, ap, env=olbf)
    key = oprng.split(oprng.prng_key(seed), 4)[0]
    ol.setup(key)
    dl = MagpoLearner(cfg, N, SystemConfig(rollout_length=T, ppo_epochs=P, num_minibatches=M), "cuda", net_seed=None, wgrad_groups=4)
    dl.guider.load_named(gp); dl.actor.load_named(ap)
    dl.setup(key)
    return ol, dl


def _close(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} (ref scale {ref:.3e})"


@pytest.mark.parametrize("cfg_args,N,T", [((8, 8, 2, 2, 2, True, 12), 8, 16), ((8, 2, 3, 2, 2, False, 9), 6, 12)])
def test_lbf_learner_parity(cfg_args, N, T):
    ol, dl = _mk(cfg_args, N, T)
    om = ol.rollout()
    dl.rollout()
    tr, otr = dl.traj, ol.traj
    assert np.array_equal(tr["action"].cpu().numpy(), otr["action"].numpy()), "sampled actions differ"
    assert np.array_equal(tr["obs"][:T].cpu().numpy(), otr["obs"].numpy())
    assert np.array_equal(tr["mask"][:T].cpu().numpy().astype(bool), otr["mask"].numpy())
    assert np.array_equal(tr["reward"].cpu().numpy(), otr["reward"].numpy())
    assert not otr["mask"].numpy().all(), "the rollout must meet illegal actions"
    # sampled actions are always legal
    assert bool(torch.gather(tr["mask"][:T], -1, tr["action"].long().unsqueeze(-1)).all())
    _close(tr["value"], otr["value"], 1e-4, 1e-6, "value")
    _close(tr["log_prob"], otr["log_prob"], 1e-4, 1e-6, "log_prob")
    for k in ("episode_return", "episode_length"):
        assert np.array_equal(dl.metrics[k].cpu().numpy(), om[k]), k
    assert om["is_terminal_step"].any()
    ol.update()
    dl.update()
    assert np.array_equal(dl.key, ol.key)
    for net, ref in ((dl.guider, ol.gp), (dl.actor, ol.ap)):
        for n, v in net.named.items():
            _close(v, ref[n].reshape(v.shape), 0, 3e-5, f"param {n}")


def test_lbf_evaluator_matches_oracle():
    from magpo_amd.actor import GruActor
    from magpo_amd.config import compose
    from magpo_amd.evaluator import get_eval_fn, get_num_eval_envs, make_rec_eval_act_fn
    from magpo_amd.utils import make_env as environments
    cfg = compose("rec_magpo", ["env=lbf", "env/scenario=8x8-2p-2f-coop", "arch.num_envs=6", "arch.num_eval_episodes=12", "env.kwargs.time_limit=15"])
    env, eval_env = environments.make(cfg)
    A, K, F = env.num_agents, env.action_dim, env.obs_dim
    assert (A, K, F) == (2, 6, 14)
    ap = onets.init_actor_params(17, F, 128, K)
    ap["head.kernel"] = ap["head.kernel"] * 40
    actor = GruActor(A, K, F, "cuda")
    evaluator = get_eval_fn(eval_env, make_rec_eval_act_fn(actor, cfg), cfg, absolute_metric=False, device="cuda")
    n = get_num_eval_envs(cfg, False)
    key = oprng.split(oprng.prng_key(2), 3)[1]
    got = evaluator({k: v.cuda() for k, v in ap.items()}, key, {"hidden_state": torch.zeros(n * A, 128, device="cuda")})
    want = oeval.evaluate(olbf.LbfSpec(8, 8, 2, 2, 2, True, 15), ap, key, 6, 12, env=olbf)
    assert np.array_equal(got["episode_length"], want["episode_length"])
    assert np.array_equal(got["episode_return"], want["episode_return"])
