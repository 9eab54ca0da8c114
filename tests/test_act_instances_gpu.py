"""Parity of EVERY instance of the fused acting kernel k_sable_act<envs per wave, NA, NH> (csrc/act_fused_kernel.hpp), in particular
the ones the bench shapes dispatch: 16 envs per wave above 8 192 envs, 8 up to 8 192, 4 up to 4 096 (act_fused.hip: magpo_sable_act_envs_per_wave).
Their state-buffer depths, LDS carve-up and `nvalid` tails differ, so each is checked on its own (Weak 1):

(a) forced instances on small ragged batches against the CPU oracle (sable_network.py:443-482, decode.py:111-153 restated in
    oracle/networks.py:sable_get_actions) and against the kernel-by-kernel composition, with the rollout replayed as a HIP graph
    (so the per-step `pending` / `flush` arguments of every instance are the captured ones);
(b) the size-based dispatch itself at BASELINE.json's env counts: fused == composed acting path over six env steps (two episode ends).
"""
import numpy as np
import pytest
import torch

from oracle import coordsum as ocs
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, rtol, atol, what):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} (ref scale {ref:.3e})"


def _instance(N, A, nh, forced=0):
    from magpo_amd._lib import lib
    epw = lib().call("magpo_sable_act_envs_per_wave", N, A, forced)
    return f"k_sable_act<{epw},{4 if A <= 4 else 8},{1 if (nh == 1 and A <= 4) else 0}>"


def _tuning(epw):
    from magpo_amd.tuning import Tuning
    t = Tuning()
    t.act_envs_per_wave = epw
    return t


# A, K, N (ragged against every wave size), n_block, n_head
CASES = [(4, 20, 70, 1, 1), (2, 10, 33, 2, 1), (8, 15, 33, 2, 1), (5, 15, 21, 1, 2), (3, 6, 19, 1, 4)]


@pytest.mark.parametrize("epw", [4, 8, 16])
@pytest.mark.parametrize("A,K,N,nb,nh", CASES)
def test_forced_instance_equals_kernel_composition_under_graph_replay(epw, A, K, N, nb, nh):
    """Three rollouts (eager, captured, replayed) of the forced instance against the eager kernel-by-kernel path: same actions,
    values / log-probs / carried states to fp32 rounding; episodes end inside every rollout (time limit 7 < T = 12)."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    sysc = SystemConfig(rollout_length=12, ppo_epochs=1, num_minibatches=1)
    key = host_split(prng_key(11), 4)[0]
    ls = []
    for fused in (False, True):
        l = MagpoLearner(CoordSumConfig(A, K, 7, 3 * K), N, sysc, DEV, net_seed=9, wgrad_groups=4, n_block=nb, n_head=nh, tuning=_tuning(epw))
        l.fused_act, l.use_graph = fused, fused
        l.setup(key)
        ls.append(l)
    a, b = ls
    for it in range(3):
        for l in ls:
            l.rollout()
        assert torch.equal(a.traj["action"], b.traj["action"]), f"rollout {it}: actions of {_instance(N, A, nh, epw)} differ from the composition"
        assert bool(a.traj["done"][1:].any())
        # two fp32 implementations with different summation orders: 1e-5 on the first rollout (as test_learner_gpu.py), 3e-5 once both
        # run from their own carried states (measured 1.1e-5 relative on rollout 1) -- an order below the 1e-4 bar against the oracle
        rt = 1e-5 if it == 0 else 3e-5
        for k in ("value", "log_prob", "adv"):
            close(b.traj[k], a.traj[k], rt, 1e-6, f"{k} (rollout {it})")
        close(b.last_val, a.last_val, rt, 1e-6, "last_val")
        for x, y in zip(a.sable_hs, b.sable_hs):
            close(y, x, rt, 1e-6, f"sable state (rollout {it})")
        for l in ls:
            l._carry_over()
    assert b.groups[0].graph is not None and not b.groups[0].graph_failed, "rollouts 2 and 3 must have been a HIP-graph capture / replay"


@pytest.mark.parametrize("epw", [4, 8, 16])
@pytest.mark.parametrize("A,K,TL,maxval,N,T,nb,nh", [(4, 20, 7, 60, 33, 12, 1, 1), (2, 10, 7, 15, 21, 12, 2, 1), (8, 15, 6, 100, 19, 10, 2, 1),
                                                     (5, 20, 6, 80, 18, 10, 1, 2),
                                                     # every episode ends exactly ON the rollout seam (T = 2 x time limit): the bootstrap-value
                                                     # launch flushes, the host zeroes, and the next rollout's first launch sees done for every env
                                                     (4, 20, 5, 60, 17, 10, 1, 1)])
def test_forced_instance_against_the_oracle(epw, A, K, TL, maxval, N, T, nb, nh):
    """Three consecutive rollouts of the forced instance (eager, captured as a HIP graph, replayed;, non-zero carried states, episode
    ends) against the oracle: sampled actions bit-exact, values / log-probs / carried retention states <= 1e-4."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig
    spec = ocs.CoordSumSpec(A, K, TL, maxval)
    gp = onets.init_guider_params(1, 64, A + 1, K, nb=nb, nh=nh)
    ap = onets.init_actor_params(2, A + 1, 128, K)
    ol = olearn.OracleLearner(spec, N, olearn.SystemCfg(rollout_length=T, ppo_epochs=1, num_minibatches=1),
                              onets.SableCfg(A, K, A + 1, embed_dim=64, n_block=nb, n_head=nh), gp, ap)
    key = oprng.split(oprng.prng_key(17), 4)[0]
    ol.setup(key)
    dl = MagpoLearner(CoordSumConfig(A, K, TL, maxval), N, SystemConfig(rollout_length=T, ppo_epochs=1, num_minibatches=1), DEV,
                      net_seed=None, wgrad_groups=4, n_block=nb, n_head=nh, tuning=_tuning(epw))
    dl.guider.load_named(gp)
    dl.actor.load_named(ap)
    dl.setup(key)
    for it in range(3):
        om = ol.rollout()
        dl.rollout()
        what = f"{_instance(N, A, nh, epw)} rollout {it}"
        assert np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy()), f"{what}: sampled actions differ from the oracle"
        assert np.array_equal(dl.traj["reward"].cpu().numpy(), ol.traj["reward"].numpy())
        assert om["is_terminal_step"].any(), "the rollout must cross an episode boundary"
        close(dl.traj["value"], ol.traj["value"], 1e-4, 2e-6, f"{what}: value")
        close(dl.traj["log_prob"], ol.traj["log_prob"], 1e-4, 2e-6, f"{what}: log_prob")
        close(dl.last_val, ol.last_val, 1e-4, 2e-6, f"{what}: last_val")
        hs = 64 // nh
        for d, o in zip(dl.sable_hs, ol.sable_hs):   # device [nb, nh, N, 64, 64] zero-padded tiles; oracle (N, nh, nb, hs, hs)
            close(d[:, :, :, :hs, :hs], o.permute(2, 1, 0, 3, 4), 1e-4, 2e-6, f"{what}: sable state")
        # the next rollout starts from the carried env / hidden state of this one (no parameter update in between on either side;
        # the oracle carries its timestep itself)
        dl._carry_over()
    assert dl.groups[0].graph is not None and not dl.groups[0].graph_failed


@pytest.mark.parametrize("A,K,N,nb,nh,epw", [(4, 20, 21, 1, 1, 4), (4, 20, 21, 1, 1, 8), (4, 20, 21, 1, 1, 16), (3, 10, 18, 2, 1, 4), (3, 10, 18, 2, 1, 8),
                                             (3, 10, 18, 2, 1, 16), (5, 15, 9, 1, 2, 8), (5, 15, 9, 1, 2, 16), (8, 15, 10, 2, 1, 4), (8, 15, 10, 2, 1, 16)])
def test_carried_decoder_states_across_rollout_seams(A, K, N, nb, nh, epw):
    """NO episode ends inside or between the rollouts (time limit 100 > 4 x 6 steps), so every rollout starts from non-zero decoder states
    that nothing resets: the first launch of a rollout has no pending rows and must take the carried states as they are (round 4 found
    them decayed twice there -- invisible wherever an episode end zeroes the states before they are compared).  Fused instance against
    the oracle AND the kernel-by-kernel path over four rollouts (the third and fourth as HIP-graph capture / replay)."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig
    T, TL, maxval = 6, 100, 3 * K
    gp = onets.init_guider_params(1, 64, A + 1, K, nb=nb, nh=nh)
    for n in gp:   # larger output projections: the decoder states must matter for the logits (at init W_o ~ 1 / 64 hides them)
        if n.endswith("w_o"):
            gp[n] = gp[n] * 8.0
    ap = onets.init_actor_params(2, A + 1, 128, K)
    ol = olearn.OracleLearner(ocs.CoordSumSpec(A, K, TL, maxval), N, olearn.SystemCfg(rollout_length=T, ppo_epochs=1, num_minibatches=1),
                              onets.SableCfg(A, K, A + 1, embed_dim=64, n_block=nb, n_head=nh), gp, ap)
    key = oprng.split(oprng.prng_key(29), 4)[0]
    ol.setup(key)
    dls = []
    for fused in (True, False):
        dl = MagpoLearner(CoordSumConfig(A, K, TL, maxval), N, SystemConfig(rollout_length=T, ppo_epochs=1, num_minibatches=1), DEV,
                          net_seed=None, wgrad_groups=4, n_block=nb, n_head=nh, tuning=_tuning(epw))
        dl.fused_act, dl.use_graph = fused, fused
        dl.guider.load_named(gp)
        dl.actor.load_named(ap)
        dl.setup(key)
        dls.append(dl)
    hs = 64 // nh
    for it in range(4):
        om = ol.rollout()
        assert not om["is_terminal_step"].any()
        for dl in dls:
            dl.rollout()
        f, c = dls
        what = f"{_instance(N, A, nh, epw)} rollout {it}"
        assert np.array_equal(f.traj["action"].cpu().numpy(), ol.traj["action"].numpy()), f"{what}: sampled actions differ from the oracle"
        assert torch.equal(f.traj["action"], c.traj["action"])
        close(f.traj["log_prob"], ol.traj["log_prob"], 1e-4, 2e-6, f"{what}: log_prob")
        for d, d2, o in zip(f.sable_hs, c.sable_hs, ol.sable_hs):
            assert float(o.abs().max()) > 1e-2
            close(d[:, :, :, :hs, :hs], o.permute(2, 1, 0, 3, 4), 1e-4, 2e-6, f"{what}: carried state vs the oracle")
            close(d, d2, 3e-5, 1e-6, f"{what}: carried state vs the kernel composition")
        # the GRU actor's hidden state is carried across the seams as well (one scan per rollout from the carried state)
        close(f.policy_h[f._cur], ol.policy_h.reshape(N * A, 128), 1e-4, 2e-6, f"{what}: policy hidden state")
        for dl in dls:
            dl._carry_over()
    assert dls[0].groups[0].graph is not None and not dls[0].groups[0].graph_failed


FULL = [(16384, 4, 20, 1, 1), (4096, 4, 20, 1, 1), (8192, 4, 5, 1, 1), (16384, 8, 15, 2, 1), (16384, 2, 6, 1, 1), (4096, 8, 15, 2, 1)]


@pytest.mark.parametrize("N,A,K,nb,nh", FULL, ids=[f"{n}envs-{a}ag-{_n}" for n, a, _n in
                                                   [(16384, 4, "k_sable_act<16,4,1>"), (4096, 4, "k_sable_act<4,4,1>"), (8192, 4, "k_sable_act<8,4,1>"),
                                                    (16384, 8, "k_sable_act<16,8,0>"), (16384, 2, "k_sable_act<16,4,1>"), (4096, 8, "k_sable_act<4,8,0>")]])
def test_size_dispatched_instance_equals_composition_at_bench_sizes(N, A, K, nb, nh, request):
    """BASELINE.json's env counts (16 384 CoordSum-4ag / LBF-2p, 8 192 and 4 096 per GPU for RWARE-4ag / the 32 768-env target, 8-agent
    teams of the 131 072-env config): the instance magpo_sable_act picks BY SIZE (named in the test id, checked here) runs six env
    steps + the bootstrap-value launch and must agree with the kernel-by-kernel acting path.

    Both are fp32 with different summation orders (logits agree to ~3e-5), and a Gumbel-max draw whose two best perturbed values lie
    closer than that may legitimately fall either way: ~1e-5 of the draws at this tolerance, i.e. a handful of the 10^5 - 10^6 here
    (at the <= 70-env sizes above, and against the oracle everywhere, the actions are identical).  So the statement checked is:
    every env whose actions all agree (>= 99.98 % of them) has values / log-probs / carried states within 3e-5; every other env is
    shown to be such a near tie at its first differing draw (Gumbel noise recomputed from the step's sample key with the oracle's
    PRNG, perturbed values of the two chosen actions within 3e-4), after which its trajectory is its own."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    assert _instance(N, A, nh) in request.node.callspec.id, f"dispatch changed: {_instance(N, A, nh)}"
    T = 6
    sysc = SystemConfig(rollout_length=T, ppo_epochs=1, num_minibatches=1)
    key = host_split(prng_key(23), 4)[0]
    res = []
    for fused in (False, True):
        # time limit 2: every episode ends after two steps, so steps 3 and 5 run on zeroed states (and pending rows of ended episodes)
        l = MagpoLearner(CoordSumConfig(A, K, 2, 3 * K), N, sysc, DEV, net_seed=9, wgrad_groups=4, n_block=nb, n_head=nh)
        assert l.tuning.act_envs_per_wave == 0
        l.fused_act, l.use_graph = fused, False
        l.setup(key)
        l.rollout()
        torch.cuda.synchronize()
        res.append({"action": l.traj["action"].cpu(), "value": l.traj["value"].cpu(), "logp": l.traj["log_prob"].cpu(),
                    "last_val": l.last_val.cpu(), "hs": [h.cpu() for h in l.sable_hs], "done": l.traj["done"].cpu(),
                    "skeys": l.groups[0].skeys_host.copy()})
        del l
        torch.cuda.empty_cache()
    a, b = res
    assert bool(a["done"][1:].any()) and np.array_equal(a["skeys"], b["skeys"])
    diff = a["action"] != b["action"]                       # [T, N, A]
    bad = diff.any(dim=2).any(dim=0)                        # envs with at least one differing action
    nbad = int(bad.sum())
    assert nbad <= max(3, int(2e-4 * N)), f"{nbad} of {N} envs differ between {_instance(N, A, nh)} and the composition"
    for e in torch.nonzero(bad).flatten().tolist():
        t, i = [int(v) for v in torch.nonzero(diff[:, e, :])[0]]
        g = oprng.gumbel(a["skeys"][t, i], N * K)           # decode.py:141-149: categorical over [N, K] logits, flat index env * K + action
        xa, xb = int(a["action"][t, e, i]), int(b["action"][t, e, i])
        va, vb = float(g[e * K + xa]) + float(a["logp"][t, e, i]), float(g[e * K + xb]) + float(b["logp"][t, e, i])
        assert abs(va - vb) <= 3e-4, f"env {e} step {t} agent {i}: actions {xa} / {xb} differ without a near tie ({va:.6f} vs {vb:.6f})"
    ok = ~bad
    close(b["value"][:, ok], a["value"][:, ok], 3e-5, 1e-6, "value")
    close(b["logp"][:, ok], a["logp"][:, ok], 3e-5, 1e-6, "log_prob")
    close(b["last_val"].reshape(N, A)[ok], a["last_val"].reshape(N, A)[ok], 3e-5, 1e-6, "last_val")
    for x, y in zip(a["hs"], b["hs"]):
        close(y[:, :, ok], x[:, :, ok], 3e-5, 1e-6, "sable state")
