"""Robot Warehouse on the GPU (csrc/rware.hip) against oracle/rware.py -- both restate Jumanji's published algorithm (UNPINNED
dynamics) and must agree bit for bit; then the MAGPO learner on its 75-wide observations (wide-observation path: rows padded to
128 floats, first layers on the MFMA dense kernels, csrc/wideobs.hip) and the evaluator."""
import numpy as np
import pytest
import torch

from oracle import evaluator as oeval
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng
from oracle import rware as orw

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("CH,SR,SC,A,Q,TL,N", [(8, 1, 3, 4, 4, 40, 70), (8, 1, 3, 2, 2, 25, 33), (8, 2, 3, 4, 4, 30, 20), (4, 1, 5, 8, 8, 30, 17)])
def test_rware_env_matches_oracle(CH, SR, SC, A, Q, TL, N):
    from magpo_amd.learner import RwareConfig, RwareEnvBatch
    spec = orw.RwareSpec(CH, SR, SC, A, 1, Q, TL)
    cfg = RwareConfig(CH, SR, SC, A, 1, Q, TL)
    keys = oprng.split(oprng.prng_key(CH * 10 + A), N)
    st, ts = orw.reset(spec, keys)
    env = RwareEnvBatch(cfg, N, "cuda")
    assert (env.H, env.W, env.NS) == (spec.H, spec.W, spec.num_shelves)
    F = cfg.obs_dim
    obs, obs_step = torch.zeros(N, A, 128, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda")
    mask = torch.zeros(N, A, 5, dtype=torch.uint8, device="cuda")
    reward, done = torch.zeros(N, A, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda")
    m_ret, m_len, m_term = torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda")
    env.reset(torch.from_numpy(keys.view(np.int32)).cuda(), obs, obs_step, mask)

    def check(tag):
        for f in ("grid_a", "grid_s", "agent_pos", "agent_dir", "queue", "step_count"):
            assert np.array_equal(getattr(env, f).cpu().numpy(), st[f]), (tag, f)
        assert np.array_equal(env.agent_carry.cpu().numpy().astype(bool), st["agent_carry"]), tag
        assert np.array_equal(env.shelf_req.cpu().numpy().astype(bool), st["shelf_req"]), tag
        assert np.array_equal(env.key.cpu().numpy().view(np.uint32), st["key"]), tag
        assert np.array_equal(obs[:, :, :F].cpu().numpy(), ts["observation"]["agents_view"]), tag
        assert float(obs[:, :, F:].abs().max()) == 0.0
        assert np.array_equal(mask.cpu().numpy().astype(bool), ts["observation"]["action_mask"]), tag
    check("reset")
    rng = np.random.default_rng(4)
    total_reward, resets, early = 0.0, 0, 0
    for t in range(4 * TL):
        m = ts["observation"]["action_mask"]
        a = np.zeros((N, A), np.int32)
        for n in range(N):
            for i in range(A):
                # mostly forward / toggle so that shelves travel; illegal FORWARDs are sent on purpose (they must become NOOPs)
                a[n, i] = rng.choice([1, 1, 1, 4, 2, 3, 0]) if rng.random() < 0.9 else 1
        st, ts = orw.step(spec, st, a, auto_reset=True)
        env.step(torch.from_numpy(a).cuda(), reward, done, obs, obs_step, m_ret, m_len, m_term, auto_reset=True, mask=mask)
        check(t)
        assert np.array_equal(reward.cpu().numpy(), ts["reward"]), t
        d = ts["step_type"] == orw.STEP_LAST
        assert np.array_equal(done.cpu().numpy().astype(bool), d), t
        assert np.array_equal(m_ret.cpu().numpy(), ts["episode_metrics"]["episode_return"]), t
        assert np.array_equal(m_len.cpu().numpy(), ts["episode_metrics"]["episode_length"]), t
        total_reward += float(ts["reward"][:, 0].sum())
        resets += int(d.sum())
        early += int((ts["episode_metrics"]["episode_length"][d] < TL).sum())
    assert resets > N and early > 0, "the test must see horizon endings and collision endings"


def _mk(cfg_args, N, T, P=2, M=2, seed=5, E=64, nh=1, nb=1):
    from magpo_amd.learner import MagpoLearner, RwareConfig, SystemConfig
    spec = orw.RwareSpec(*cfg_args)
    cfg = RwareConfig(*cfg_args)
    A, K, F = spec.num_agents, 5, spec.obs_dim
    scfg = onets.SableCfg(A, K, F, embed_dim=E, n_head=nh, n_block=nb)
    gp = onets.init_guider_params(1, E, F, K, nh=nh, nb=nb)
    ap = onets.init_actor_params(2, F, 128, K)
    gp["dec.head.dense1.kernel"] = gp["dec.head.dense1.kernel"] * 30
    ap["head.kernel"] = ap["head.kernel"] * 30
    ol = olearn.OracleLearner(spec, N, olearn.SystemCfg(rollout_length=T, ppo_epochs=P, num_minibatches=M), scfg, gp, ap, env=orw)
    key = oprng.split(oprng.prng_key(seed), 4)[0]
    ol.setup(key)
    dl = MagpoLearner(cfg, N, SystemConfig(rollout_length=T, ppo_epochs=P, num_minibatches=M), "cuda", net_seed=None, wgrad_groups=4,
                      embed_dim=E, n_head=nh, n_block=nb)
    dl.guider.load_named(gp); dl.actor.load_named(ap)
    dl.setup(key)
    return ol, dl


def _close(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} (ref scale {ref:.3e})"


@pytest.mark.parametrize("E,nh,nb", [(64, 1, 1), (128, 2, 3), (128, 1, 1)])
def test_rware_learner_parity(E, nh, nb):
    """Wide observations (75 features) through rollout, minibatch gradients and a full update, against the oracle.  (128, 2, 3) is the tuned
    MAGPO network of RWARE tiny-4ag (experiment_data/params.csv: n_embd 128, n_head 2, n_block 3), (128, 1, 1) that of tiny-2ag / medium-4ag."""
    N, T = 8, 16
    ol, dl = _mk((8, 1, 3, 4, 1, 4, 11), N, T, E=E, nh=nh, nb=nb)
    F = 75
    om = ol.rollout()
    dl.rollout()
    tr, otr = dl.traj, ol.traj
    assert np.array_equal(tr["action"].cpu().numpy(), otr["action"].numpy()), "sampled actions differ"
    assert np.array_equal(tr["obs"][:T, :, :, :F].cpu().numpy(), otr["obs"].numpy())
    assert np.array_equal(tr["mask"][:T].cpu().numpy().astype(bool), otr["mask"].numpy())
    assert np.array_equal(tr["reward"].cpu().numpy(), otr["reward"].numpy())
    _close(tr["value"], otr["value"], 1e-4, 1e-6, "value")
    _close(tr["log_prob"], otr["log_prob"], 1e-4, 1e-6, "log_prob")
    _close(dl.policy_h[dl._cur], ol.policy_h.reshape(N * 4, 128), 1e-4, 1e-6, "policy hidden")
    assert om["is_terminal_step"].any()
    ks = oprng.split(ol.key, 4)
    bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], 4)
    gg, ag, info, inter = ol.minibatch_grads(ol.make_minibatches(bp, apm)[1])
    dl.minibatch_grads(dl._permutation(ks[1], N)[N // 2:].contiguous(), dl._permutation(ks[2], 4))
    for n, g in dl.guider.named_grads.items():
        scale = max(gg[n].abs().max().item(), 1e-6)
        _close(g / scale, gg[n].reshape(g.shape) / scale, 0, 2e-3, f"guider grad {n}")
    for n, g in dl.actor.named_grads.items():
        scale = max(ag[n].abs().max().item(), 1e-6)
        _close(g / scale, ag[n].reshape(g.shape) / scale, 0, 2e-3, f"actor grad {n}")
    ol.update()
    dl.update()
    assert np.array_equal(dl.key, ol.key)
    # Parameters after the update's four clip + Adam steps.  Adam divides every element by its own gradient magnitude, so an element whose
    # gradient sits at the fp32 noise floor of its tensor (|g| ~ 1e-5 x the largest, where the gradient check above allows 100 % relative
    # error) moves by a different fraction of lr on the two sides: with the three-block net 2 of the 16 384 elements of enc.block2.retn.w_k
    # differ by 0.3 lr, everything else by < 0.02 lr (scripts/debug/rware_update_err.py; mechanism: profiles/r03_step2_sensitivity_fp64.txt).
    # Bound: 3e-5 flat (as for CoordSum) for the one-block nets; ONLY the three-block net, and there only the retention projections of
    # its last encoder block (the stiff direction of DESIGN 2b), may hold elements within ONE Adam step (lr), at most 0.1 % of a tensor.
    lr = 2.5e-4
    for net, ref in ((dl.guider, ol.gp), (dl.actor, ol.ap)):
        for n, v in net.named.items():
            d = (v.detach().cpu().double().reshape(-1) - ref[n].reshape(v.shape).double().reshape(-1)).abs()
            if nb == 3 and net is dl.guider and n.startswith("enc.block2.retn."):
                assert d.max().item() <= lr, f"param {n}: max err {d.max().item():.3e}"
                assert int((d > 3e-5).sum()) <= max(0, d.numel() // 1000), f"param {n}: {int((d > 3e-5).sum())} of {d.numel()} elements beyond 3e-5"
            else:
                assert d.max().item() <= 3e-5, f"param {n}: max err {d.max().item():.3e}"


def test_rware_evaluator_and_entry_point(tmp_path):
    from magpo_amd.actor import GruActor
    from magpo_amd.config import compose
    from magpo_amd.evaluator import get_eval_fn, get_num_eval_envs, make_rec_eval_act_fn
    from magpo_amd.systems.gpo.anakin import rec_magpo
    from magpo_amd.utils import make_env as environments
    cfg = compose("rec_magpo", ["env=rware", "env/scenario=tiny-2ag", "arch.num_envs=6", "arch.num_eval_episodes=12", "env.kwargs.time_limit=15"])
    env, eval_env = environments.make(cfg)
    A, K, F = env.num_agents, env.action_dim, env.obs_dim
    assert (A, K, F) == (2, 5, 73)
    ap = onets.init_actor_params(17, F, 128, K)
    ap["head.kernel"] = ap["head.kernel"] * 40
    actor = GruActor(A, K, F, "cuda")
    evaluator = get_eval_fn(eval_env, make_rec_eval_act_fn(actor, cfg), cfg, absolute_metric=False, device="cuda")
    n = get_num_eval_envs(cfg, False)
    key = oprng.split(oprng.prng_key(2), 3)[1]
    got = evaluator({k: v.cuda() for k, v in ap.items()}, key, {"hidden_state": torch.zeros(n * A, 128, device="cuda")})
    want = oeval.evaluate(orw.RwareSpec(8, 1, 3, 2, 1, 2, 15), ap, key, 6, 12, env=orw)
    assert np.array_equal(got["episode_length"], want["episode_length"])
    assert np.array_equal(got["episode_return"], want["episode_return"])
    # the reference's DEFAULT experiment (configs/default/rec_magpo.yaml: env rware, scenario tiny-2ag) end to end, shortened
    cfg = compose("rec_magpo", ["arch.num_envs=8", "arch.num_evaluation=2", "arch.num_eval_episodes=8", "arch.num_absolute_metric_eval_episodes=16",
                                "system.total_timesteps=~", "system.num_updates=4", "system.rollout_length=16", "system.ppo_epochs=2",
                                "env.kwargs.time_limit=20", f"logger.base_exp_path={tmp_path}/"])
    perf = rec_magpo.run_experiment(cfg)
    assert np.isfinite(perf)
