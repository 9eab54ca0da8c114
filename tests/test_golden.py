"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
the CPU test guards the oracle against drift; the GPU tests check the HIP path against the same vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import coordsum as ocs
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_prng_and_env_fixtures():
    f = np.load(os.path.join(G, "prng.npz"))
    ks = oprng.split(f["key"], 6)
    assert np.array_equal(ks, f["split6"]) and np.array_equal(oprng.random_bits(ks[1], 32), f["bits32"])
    assert np.array_equal(oprng.randint(ks[2], 101, 0, 60), f["randint_0_60"])
    assert np.array_equal(oprng.permutation(ks[3], 16), f["perm16"]) and np.array_equal(oprng.permutation(ks[4], 1000), f["perm1000"])
    assert np.array_equal(oprng.categorical(ks[5], f["cat_logits"]), f["cat_sample"])
    m = f["choice_mask"]
    assert np.array_equal(np.concatenate([oprng.choice(k, 64, 1, True, m) for k in oprng.split(ks[0], 16)]), f["choice_p_replace"])
    assert m[f["choice_p_replace"]].all() and m[f["choice_p_noreplace"]].all() and len(set(f["choice_p_noreplace"].tolist())) == 5
    assert np.array_equal(oprng.choice(ks[1], 64, 5, False, m), f["choice_p_noreplace"])
    assert np.array_equal(oprng.choice(ks[2], 110, 4, False), f["choice_noreplace"]) and np.array_equal(oprng.choice(ks[3], 64, 7, True), f["choice_replace"])
    e = np.load(os.path.join(G, "coordsum.npz"))
    A, K, TL, mv = e["cfg"]
    spec = ocs.CoordSumSpec(A, K, TL, mv)
    st, _ = ocs.reset(spec, e["env_keys"])
    for i in range(e["actions"].shape[0]):
        st, ts = ocs.step(spec, st, e["actions"][i])
        assert np.array_equal(ts["reward"][:, 0], e["reward"][i]) and np.array_equal(ts["step_type"] == ocs.STEP_LAST, e["done"][i])
    assert np.array_equal(st["record"], e["final_record"]) and np.array_equal(st["key"], e["final_key"])


def _grid_env(name):
    from oracle import lbf as olbf
    from oracle import rware as orw
    f = np.load(os.path.join(G, name + ".npz"))
    c = [int(x) for x in f["cfg"]]
    if name == "lbf":
        return f, olbf, olbf.LbfSpec(c[0], c[1], c[2], c[3], c[4], bool(c[5]), c[6]), c
    return f, orw, orw.RwareSpec(*c), c


@pytest.mark.parametrize("name", ["lbf", "rware"])
def test_oracle_reproduces_grid_env_fixtures(name):
    f, mod, spec, _ = _grid_env(name)
    st, ts = mod.reset(spec, f["env_keys"])
    assert np.array_equal(ts["observation"]["agents_view"], f["obs0"])
    for i in range(f["actions"].shape[0]):
        st, ts = mod.step(spec, st, f["actions"][i])
        assert np.array_equal(ts["reward"][:, 0], f["reward"][i]) and np.array_equal(ts["step_type"] == 2, f["done"][i]), i
        assert np.array_equal(ts["episode_metrics"]["episode_return"], f["episode_return"][i])
    assert np.array_equal(ts["observation"]["agents_view"], f["final_obs"]) and np.array_equal(st["key"], f["final_key"])
    assert f["done"].any() and (name == "rware" or f["reward"].max() > 0)   # (a random policy does not deliver shelves: the reward path of
    # Robot Warehouse is covered by tests/test_oracle_rware.py and the bit-exact GPU comparison over long random rollouts)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lbf", "rware"])
def test_hip_grid_envs_match_golden(name):
    from magpo_amd.learner import LbfConfig, RwareConfig, make_env_batch, obs_row_stride
    f, _, spec, c = _grid_env(name)
    cfg = LbfConfig(c[0], c[1], c[2], c[3], c[4], bool(c[5]), c[6]) if name == "lbf" else RwareConfig(*c)
    N, A, K, F = 6, cfg.num_agents, cfg.num_actions, cfg.obs_dim
    env = make_env_batch(cfg, N, "cuda")
    obs, obs_step = torch.zeros(N, A, obs_row_stride(F), device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda")
    mask = torch.zeros(N, A, K, dtype=torch.uint8, device="cuda")
    reward, done = torch.zeros(N, A, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda")
    m_ret, m_len, m_term = torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda")
    env.reset(torch.from_numpy(f["env_keys"].view(np.int32)).cuda(), obs, obs_step, mask)
    assert np.array_equal(obs[:, :, :F].cpu().numpy(), f["obs0"])
    for i in range(f["actions"].shape[0]):
        env.step(torch.from_numpy(f["actions"][i]).cuda(), reward, done, obs, obs_step, m_ret, m_len, m_term, auto_reset=True, mask=mask)
        assert np.array_equal(reward[:, 0].cpu().numpy(), f["reward"][i]) and np.array_equal(done.cpu().numpy().astype(bool), f["done"][i]), i
        assert np.array_equal(m_ret.cpu().numpy(), f["episode_return"][i])
    assert np.array_equal(obs[:, :, :F].cpu().numpy(), f["final_obs"])
    assert np.array_equal(mask.cpu().numpy().astype(bool), f["final_mask"])
    assert np.array_equal(env.key.cpu().numpy().view(np.uint32), f["final_key"])


def _stat(v):
    x = v.detach().cpu().double().reshape(-1)
    return np.concatenate([[x.sum().item(), x.abs().sum().item()], x[:8].numpy(), np.zeros(max(0, 8 - x.numel()))])


def _load_learner():
    f = np.load(os.path.join(G, "learner.npz"))
    A, K = int(f["cfg"][0]), int(f["cfg"][1])
    gp0 = onets.init_guider_params(11, 64, A + 1, K)
    ap0 = onets.init_actor_params(12, A + 1, 128, K)
    for n, v in gp0.items():  # the seeds must regenerate the exact initial parameters the fixture was made with
        assert np.allclose(_stat(v), f["gp0/" + n], rtol=0, atol=1e-9), n
    for n, v in ap0.items():
        assert np.allclose(_stat(v), f["ap0/" + n], rtol=0, atol=1e-9), n
    return f, gp0, ap0


def test_oracle_reproduces_learner_fixture():
    torch.set_num_threads(1)
    f, gp0, ap0 = _load_learner()
    A, K, TL, mv, N, T = (int(x) for x in f["cfg"])
    ol = olearn.OracleLearner(ocs.CoordSumSpec(A, K, TL, mv), N, olearn.SystemCfg(rollout_length=T, ppo_epochs=2, num_minibatches=2),
                              onets.SableCfg(A, K, A + 1), gp0, ap0)
    ol.setup(f["key"])
    for s in range(1, int(f["n_steps"]) + 1):
        ol.rollout()
        assert np.array_equal(ol.traj["action"].numpy(), f[f"s{s}_traj_action"])
        assert np.allclose(ol.traj["value"].numpy(), f[f"s{s}_traj_value"], atol=1e-6)
        ol.update()
        assert np.array_equal(ol.key, f[f"s{s}_key_after"])
        for n, v in ol.gp.items():
            assert np.allclose(_stat(v), f[f"gp{s}/" + n], rtol=1e-5, atol=1e-4), (s, n)


@pytest.mark.gpu
def test_hip_env_and_prng_match_golden(L, stream):
    from tests.test_kernels_gpu import DevEnv, dev
    f = np.load(os.path.join(G, "prng.npz"))
    out = torch.zeros(6, 2, dtype=torch.int32, device="cuda")
    L.call("magpo_threefry_split", dev(f["key"].view(np.int32)), out, 6, stream)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), f["split6"])
    e = np.load(os.path.join(G, "coordsum.npz"))
    A, K, TL, mv = (int(x) for x in e["cfg"])
    env = DevEnv(L, stream, ocs.CoordSumSpec(A, K, TL, mv), 6)
    env.reset(e["env_keys"])
    for i in range(e["actions"].shape[0]):
        env.step(dev(e["actions"][i]))
        assert np.array_equal(env.reward[:, 0].cpu().numpy(), e["reward"][i]), i
        assert np.array_equal(env.obs[:, 0, -1].cpu().numpy(), e["obs_target"][i].astype(np.float32))
        assert np.array_equal(env.m_ret.cpu().numpy(), e["episode_return"][i])
    assert np.array_equal(env.record.cpu().numpy(), e["final_record"])
    assert np.array_equal(env.key.cpu().numpy().view(np.uint32), e["final_key"])


@pytest.mark.gpu
def test_hip_learner_matches_golden():
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig
    f, gp0, ap0 = _load_learner()
    A, K, TL, mv, N, T = (int(x) for x in f["cfg"])
    dl = MagpoLearner(CoordSumConfig(A, K, TL, mv), N, SystemConfig(rollout_length=T, ppo_epochs=2, num_minibatches=2), "cuda",
                      net_seed=None, wgrad_groups=4)
    dl.guider.load_named(gp0)
    dl.actor.load_named(ap0)
    dl.setup(f["key"])
    for s in range(1, int(f["n_steps"]) + 1):   # steps 2.. start from non-zero retention states (quirk B19 is visible)
        dl.rollout()
        assert np.array_equal(dl.traj["action"].cpu().numpy(), f[f"s{s}_traj_action"]), f"step {s}: sampled actions must be bit-exact"
        assert np.array_equal(dl.traj["reward"].cpu().numpy(), f[f"s{s}_traj_reward"])
        for k in ("value", "log_prob", "adv", "targets"):
            assert np.allclose(dl.traj[k].cpu().numpy(), f[f"s{s}_traj_" + k], rtol=1e-4, atol=1e-5), (s, k)
        dl.update()
        dl._carry_over()
        assert np.array_equal(dl.key, f[f"s{s}_key_after"])
        tol = 3e-5 + 1e-4 * (s - 1)   # Adam turns tiny gradient differences into ~lr-sized steps (see test_learner_gpu.py)
        for n, v in dl.guider.named.items():
            st = _stat(v)
            assert np.allclose(st[2:], f[f"gp{s}/" + n][2:], atol=tol), (s, n)          # leading elements
            assert abs(st[0] - f[f"gp{s}/" + n][0]) <= tol * v.numel(), (s, n)          # checksum
        for n, v in dl.actor.named.items():
            st = _stat(v)
            assert np.allclose(st[2:], f[f"ap{s}/" + n][2:], atol=tol), (s, n)
            assert abs(st[0] - f[f"ap{s}/" + n][0]) <= tol * v.numel(), (s, n)
