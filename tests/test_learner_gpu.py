"""End-to-end GPU parity of the MAGPO learner (rollout, GAE, minibatch gradients, optimiser step)
against the CPU oracle on identical seeds, parameters and PRNG keys."""
import numpy as np
import pytest
import torch

from oracle import coordsum as ocs
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(A, K, TL, maxval, N, T, P=2, M=2, seed=42, nb=1, nh=1, E=64):
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig
    spec = ocs.CoordSumSpec(A, K, TL, maxval)
    scfg = onets.SableCfg(A, K, A + 1, embed_dim=E, n_block=nb, n_head=nh)
    osys = olearn.SystemCfg(rollout_length=T, ppo_epochs=P, num_minibatches=M)
    gp = onets.init_guider_params(1, E, A + 1, K, nb=nb, nh=nh)
    ap = onets.init_actor_params(2, A + 1, 128, K)
    ol = olearn.OracleLearner(spec, N, osys, scfg, gp, ap)
    key = oprng.split(oprng.prng_key(seed), 4)[0]
    ol.setup(key)
    dl = MagpoLearner(CoordSumConfig(A, K, TL, maxval), N, SystemConfig(rollout_length=T, ppo_epochs=P, num_minibatches=M), DEV,
                      net_seed=None, wgrad_groups=8, n_block=nb, n_head=nh, embed_dim=E)
    dl.guider.load_named(gp)
    dl.actor.load_named(ap)
    dl.setup(key)
    return ol, dl


def close(a, b, rtol, atol, what):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} (ref scale {ref:.3e})"


@pytest.mark.parametrize("A,K,TL,maxval,N,T,nb,nh,E", [(4, 20, 10, 60, 8, 16, 1, 1, 64), (2, 10, 7, 15, 4, 12, 1, 1, 64), (3, 10, 9, 30, 6, 11, 1, 1, 64),
                                                       (4, 20, 10, 60, 8, 16, 2, 1, 64), (8, 15, 9, 100, 4, 11, 3, 1, 64),
                                                       (5, 20, 9, 80, 4, 11, 2, 2, 64), (3, 10, 9, 30, 6, 10, 1, 4, 64),
                                                       # embed_dim 32 / 16 (tuned CoordSum 3x10 and both LBF rows of experiment_data/params.csv:
                                                       # n_embd 32 with 1 / 4 heads): the narrow net runs embedded in the 64-wide kernels
                                                       (3, 10, 9, 30, 6, 11, 2, 1, 32), (4, 20, 10, 60, 4, 12, 1, 2, 32), (2, 10, 7, 15, 4, 12, 1, 2, 16),
                                                       # the default rollout length on BASELINE's shapes: 8 / 16 / 10 retention chunks per
                                                       # sequence (more than 8 takes the per-chunk bookkeeping branch), narrow heads over 10 chunks
                                                       (4, 20, 100, 60, 2, 128, 1, 1, 64), (8, 15, 100, 100, 2, 128, 1, 1, 64), (5, 20, 100, 80, 2, 128, 1, 2, 64),
                                                       # embed_dim 128 (20 of the 22 tuned RWARE / LBF rows of experiment_data/params.csv): n_head 1 = one
                                                       # 128-wide head evaluated blockwise on four 64 x 64 state tiles, n_head 2 / 4 = real 64- / 32-wide heads
                                                       (4, 20, 10, 60, 8, 16, 1, 1, 128), (3, 10, 9, 30, 6, 11, 2, 2, 128), (4, 20, 10, 60, 4, 12, 1, 4, 128),
                                                       (2, 10, 7, 15, 4, 12, 3, 1, 128)])
def test_rollout_and_update_parity(A, K, TL, maxval, N, T, nb, nh, E):
    # (the T = 128 cases are about the chunk bookkeeping of a full-length rollout; they spend their time in the oracle's autograd, so one
    # PPO epoch there -- the cumulative prev_hstates permutation across epochs, quirk B19, is covered by the short cases)
    ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh, E=E, P=2 if T < 100 else 1)
    assert np.array_equal(dl.env.target.cpu().numpy(), ol.env_state["target"])
    assert np.array_equal(dl.key, ol.key)
    om = ol.rollout(record_logits=True)
    dl.rollout()
    tr, otr = dl.traj, ol.traj
    # sampled action indices: bit-exact (north_star); logits agree far below the gumbel gaps at this size
    assert np.array_equal(tr["action"].cpu().numpy(), otr["action"].numpy()), "sampled actions differ"
    assert np.array_equal(tr["obs"][:T].cpu().numpy(), otr["obs"].numpy().astype(np.float32))
    assert np.array_equal(tr["done"][:T].cpu().numpy().astype(bool), otr["done"][:, :, 0].numpy())
    assert np.array_equal(tr["reward"].cpu().numpy(), otr["reward"].numpy())
    close(tr["value"], otr["value"], 1e-4, 1e-6, "value")
    close(tr["log_prob"], otr["log_prob"], 1e-5, 1e-6, "log_prob")
    close(dl.last_val, ol.last_val, 1e-4, 1e-6, "last_val")
    close(tr["adv"], otr["adv"], 1e-4, 1e-5, "adv")
    close(tr["targets"], otr["targets"], 1e-4, 1e-5, "targets")
    for k in ("episode_return", "episode_length"):
        assert np.array_equal(dl.metrics[k].cpu().numpy(), om[k]), k
    assert om["is_terminal_step"].any(), "the test must cross an episode boundary"
    for d, o in zip(dl.sable_hs, ol.sable_hs):
        # device head states are zero-padded to 64 x 64; oracle layout (N, nh, nb, hs, hs).  With embed_dim < 64 every feature is
        # carried m = 64 / E times (params.WidthEmbedding): S'[m i, m j + c] = S[i, j], the rows between are zero
        if E == 128 and nh == 1:   # one 128-wide head = tiles S[I][J] at index 2 I + J
            full = torch.cat([torch.cat([d[:, 0], d[:, 1]], -1), torch.cat([d[:, 2], d[:, 3]], -1)], -2)   # [nb, N, 128, 128]
            close(full, o[:, 0].permute(1, 0, 2, 3), 1e-4, 1e-6, "sable state (128-wide head)")
            continue
        hs, m = max(E, 64) // nh, max(1, 64 // E)
        close(d[:, :, :, :hs:m, :hs:m], o.permute(2, 1, 0, 3, 4), 1e-4, 1e-6, "sable state")
        assert float(d[:, :, :, hs:, :].abs().max() if hs < 64 else 0.0) == 0.0
        if m > 1:
            assert float(d[:, :, :, 1:hs:m, :].abs().max()) == 0.0 and torch.equal(d[:, :, :, :hs:m, 1:hs:m], d[:, :, :, :hs:m, :hs:m])
    close(dl.policy_h[dl._cur], ol.policy_h.reshape(N * A, 128), 1e-4, 1e-6, "policy hidden")
    assert np.array_equal(dl.key, ol.key)

    # ---- one minibatch: losses and gradients (hand-written backward vs autograd of the oracle)
    ks = oprng.split(ol.key, 4)
    bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
    bpd = dl._permutation(ks[1], N)
    apd = dl._permutation(ks[2], A)
    assert np.array_equal(bpd.cpu().numpy(), bp) and np.array_equal(apd.cpu().numpy(), apm)
    mbs = ol.make_minibatches(bp, apm)
    gg, ag, info, inter = ol.minibatch_grads(mbs[1])
    mbsz = N // ol.sys.num_minibatches
    dl.minibatch_grads(bpd[mbsz:2 * mbsz].contiguous(), apd)
    lo = dl.loss_out.cpu()
    for i, k in [(1, "value_loss"), (2, "actor_loss"), (3, "guider_loss"), (4, "kl_loss"), (5, "entropy"), (6, "actor_kl")]:
        close(lo[i], torch.tensor(info[k]), 1e-3, 2e-6, k)
    close(dl.guider.b.t["t_value"], inter["value"], 1e-4, 1e-6, "train value")
    for n, g in dl.guider.named_grads.items():
        scale = max(gg[n].abs().max().item(), 1e-6)
        close(g / scale, gg[n].reshape(g.shape) / scale, 0, 2e-3, f"guider grad {n}")
    for n, g in dl.actor.named_grads.items():
        scale = max(ag[n].abs().max().item(), 1e-6)
        close(g / scale, ag[n].reshape(g.shape) / scale, 0, 2e-3, f"actor grad {n}")

    # ---- full update (epochs x minibatches, clip + Adam) and parameter parity afterwards
    oinfos = ol.update()
    losses = dl.update().cpu()
    assert np.array_equal(dl.key, ol.key)
    for n, v in dl.guider.named.items():
        close(v, ol.gp[n].reshape(v.shape), 0, 3e-5, f"guider param {n}")
    for n, v in dl.actor.named.items():
        close(v, ol.ap[n].reshape(v.shape), 0, 3e-5, f"actor param {n}")
    close(losses[-1, -1, 1], torch.tensor(oinfos[-1]["value_loss"]), 5e-3, 1e-5, "final value loss")
    # ---- update steps 2 and 3 against the oracle: carried-over env / hidden state, non-zero rollout-start retention
    # states (so the cumulative prev_hstates permutation of rec_magpo.py:437-471, quirk B19, changes the result),
    # Adam counts > P*M, and from step 2 on the HIP-graph capture / replay of the rollout.
    # The update is a chaotic map of the parameters: for the 3-block net a 3e-6 parameter difference after step 1 changes
    # the step-2 gradient of enc.block2.retn.w_k by 47 % -- reproduced inside the oracle alone by evaluating it at the
    # device's parameters, where device and oracle agree to 8e-6 (scripts/debug/step2_sens.py).  So each further step is
    # compared from a COMMON starting point: the oracle takes over the device's parameters and Adam moments (the drift
    # up to there is bounded separately), while env state, PRNG keys, hidden states and trajectories keep running unsynced.
    dl._carry_over()
    for s in ((2, 3) if T < 100 else (2,)):   # (the T = 128 cases spend their time in the oracle: one further step there)
        drift = max((v.cpu() - ref[n].reshape(v.shape)).abs().max().item() for net, ref in ((dl.guider, ol.gp), (dl.actor, ol.ap))
                    for n, v in net.named.items())
        assert drift <= 3e-5, f"parameter drift entering update step {s}: {drift:.2e}"
        for net, ref, opt, mu, nu in ((dl.guider, ol.gp, ol.g_opt, dl.g_mu, dl.g_nu), (dl.actor, ol.ap, ol.a_opt, dl.a_mu, dl.a_nu)):
            mv, nv = net.P.views(mu), net.P.views(nu)
            from magpo_amd.params import actor_named_views, guider_named_views
            named = (lambda v: guider_named_views(v, E, nh)) if net is dl.guider else actor_named_views
            mn, nn = named(mv), named(nv)
            for n in ref:
                ref[n] = net.named[n].detach().cpu().reshape(ref[n].shape).clone()
                opt["mu"][n] = mn[n].detach().cpu().reshape(ref[n].shape).clone()
                opt["nu"][n] = nn[n].detach().cpu().reshape(ref[n].shape).clone()
        om = ol.rollout()
        dl.rollout()
        assert np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy()), f"update step {s}: sampled actions differ"
        assert np.array_equal(dl.traj["reward"].cpu().numpy(), ol.traj["reward"].numpy())
        close(dl.traj["value"], ol.traj["value"], 1e-4, 2e-6, f"value (step {s})")
        close(dl.traj["adv"], ol.traj["adv"], 1e-4, 2e-5, f"adv (step {s})")
        for k in ("episode_return", "episode_length"):
            assert np.array_equal(dl.metrics[k].cpu().numpy(), om[k]), (s, k)
        for d, o in zip(dl.groups[0].prev_sable_hs, ol.prev_sable_hs):
            assert float(o.abs().max()) > 0, "rollout-start retention states must be non-zero from step 2 on"
        ol.update()
        dl.update()
        dl._carry_over()
        assert np.array_equal(dl.key, ol.key)
        for net, ref in ((dl.guider, ol.gp), (dl.actor, ol.ap)):
            for n, v in net.named.items():
                close(v, ref[n].reshape(v.shape), 0, 3e-5, f"param {n} (step {s})")
    assert dl.groups[0].graph is not None, "the rollouts of steps 2 and 3 should have been a HIP-graph capture / replay"


def test_graph_replay_equals_eager_rollout():
    """The HIP-graph replay of the rollout must reproduce the eager rollout bit for bit (same kernels, same order)."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    sysc = SystemConfig(rollout_length=9, ppo_epochs=1, num_minibatches=1)   # odd T: exercises the buffer-role fix-up
    key = host_split(prng_key(5), 4)[0]
    ls = []
    for use_graph in (False, True):
        l = MagpoLearner(CoordSumConfig(3, 10, 7, 30), 8, sysc, "cuda", net_seed=4, wgrad_groups=4)
        l.use_graph = use_graph
        l.setup(key)
        ls.append(l)
    for it in range(4):
        for l in ls:
            l.update_step()
        assert ls[1].groups[0].graph is not None or it < 1, "rollout should be captured from the second call on"
        for k in ("action", "value", "log_prob", "reward", "adv"):
            assert torch.equal(ls[0].traj[k], ls[1].traj[k]), (it, k)
        assert torch.equal(ls[0].guider.P.flat, ls[1].guider.P.flat) and np.array_equal(ls[0].key, ls[1].key)
    assert not ls[1].groups[0].graph_failed


@pytest.mark.parametrize("A,K,N,nb,nh", [(4, 20, 70, 1, 1), (5, 15, 33, 2, 2), (16, 6, 40, 1, 4)])
def test_fused_act_equals_kernel_composition(A, K, N, nb, nh):
    """k_sable_act (one launch per env step) against the kernel-by-kernel acting path: same actions, and values /
    log-probs / retention states to fp32 rounding, over ragged workgroups (N not a multiple of 32) and episode ends."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    sysc = SystemConfig(rollout_length=12, ppo_epochs=1, num_minibatches=1)
    key = host_split(prng_key(11), 4)[0]
    ls = []
    for fused in (False, True):
        l = MagpoLearner(CoordSumConfig(A, K, 7, 3 * K), N, sysc, "cuda", net_seed=9, wgrad_groups=4, n_block=nb, n_head=nh)
        l.fused_act, l.use_graph = fused, False
        l.setup(key)
        l.rollout()
        ls.append(l)
    a, b = ls
    assert torch.equal(a.traj["action"], b.traj["action"])
    assert bool(a.traj["done"].any())
    for k in ("value", "log_prob", "adv"):
        close(b.traj[k], a.traj[k], 1e-5, 1e-6, k)
    close(b.last_val, a.last_val, 1e-5, 1e-6, "last_val")
    for x, y in zip(a.sable_hs, b.sable_hs):
        close(y, x, 1e-5, 1e-6, "sable state")


@pytest.mark.parametrize("A,K,N,T,nb", [(4, 20, 12, 16, 1), (3, 10, 7, 9, 2)])
def test_fused_segments_equal_kernel_composition(A, K, N, T, nb):
    """k_seg_post / k_seg_bwd (token-local parts between the retention ops as single launches) against the kernel-by-kernel
    training path: same data, same parameters -> same losses and gradients to fp32 rounding (ragged row counts included)."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    sysc = SystemConfig(rollout_length=T, ppo_epochs=1, num_minibatches=1)
    key = host_split(prng_key(21), 4)[0]
    ls = []
    for fused in (False, True):
        l = MagpoLearner(CoordSumConfig(A, K, 7, 3 * K), N, sysc, "cuda", net_seed=3, wgrad_groups=4, n_block=nb)
        l.guider.fused_segments = fused
        l.use_graph = False
        l.setup(key)
        l.rollout()
        env_idx = torch.arange(N, device="cuda", dtype=torch.int32)
        agent_perm = torch.arange(A, device="cuda", dtype=torch.int32)
        l.minibatch_grads(env_idx, agent_perm)
        torch.cuda.synchronize()
        ls.append(l)
    a, b = ls
    assert torch.equal(a.traj["action"], b.traj["action"])
    close(b.loss_out, a.loss_out, 1e-5, 1e-6, "losses")
    ga, gb = a.guider.grads, b.guider.grads
    scale = float(ga.abs().max())
    assert float((ga - gb).abs().max()) <= 2e-5 * scale + 1e-7, "guider gradients differ between the fused and the composed path"
    close(b.actor.grads, a.actor.grads, 1e-5, 1e-7, "actor gradients")


@pytest.mark.parametrize("A,K,N,T,nb", [(4, 20, 12, 16, 1), (3, 10, 7, 9, 2), (8, 15, 5, 12, 1)])
def test_class_tables_equal_dense_path(A, K, N, T, nb):
    """First-layer class tables (csrc/classtab.hip: layers in front of the GRU / the first retention evaluated on the distinct
    (agent, target[, step]) / (previous action, step) rows only, parameter gradients from per-class sums) against the dense
    path on every row: same losses and gradients to fp32 summation order."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    sysc = SystemConfig(rollout_length=T, ppo_epochs=1, num_minibatches=1)
    key = host_split(prng_key(33), 4)[0]
    ls = []
    for tables in (False, True):
        l = MagpoLearner(CoordSumConfig(A, K, 7, 3 * K), N, sysc, "cuda", net_seed=3, wgrad_groups=4, n_block=nb)
        l.class_tables = tables
        l.use_graph = False
        l.setup(key)
        l.rollout()
        g = torch.Generator().manual_seed(1)
        l.minibatch_grads(torch.randperm(N, generator=g).int().cuda(), torch.randperm(A, generator=g).int().cuda())
        torch.cuda.synchronize()
        ls.append(l)
    a, b = ls
    assert torch.equal(a.traj["action"], b.traj["action"])
    close(b.loss_out, a.loss_out, 1e-5, 1e-6, "losses")
    for net in ("guider", "actor"):
        for n, ga in getattr(a, net).named_grads.items():
            gb = getattr(b, net).named_grads[n]
            scale = float(ga.abs().max())
            assert float((ga - gb).abs().max()) <= 3e-5 * scale + 1e-9, f"{net} gradient {n} differs between class tables and the dense path"


@pytest.mark.parametrize("U,mu", [(1, 2), (1, 4), (2, 2)])
def test_micro_batches_equal_one_pass(U, mu):
    """``system.micro_batches``: every minibatch trained in mu slabs whose gradients are accumulated before the one optimiser
    step == the minibatch in one pass (same parameters after a whole update up to fp32 summation order; the advantage statistics
    are those of the whole minibatch in both).  U = 2: two local groups trained as one batch of sequences."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
    A, K, N, T = 4, 20, 16, 12
    key = host_split(prng_key(21), 4)[0]
    ls = []
    for m in (1, mu):
        sysc = SystemConfig(rollout_length=T, ppo_epochs=2, num_minibatches=2, micro_batches=m)
        l = MagpoLearner(CoordSumConfig(A, K, 7, 3 * K), N, sysc, "cuda", net_seed=5, wgrad_groups=4, num_groups=U)
        l.setup(key, n_groups=U)
        losses = l.update_step()
        torch.cuda.synchronize()
        ls.append((l, losses))
    (a, la), (b, lb) = ls
    for ga, gb in zip(a.groups, b.groups):
        assert torch.equal(ga.traj["action"], gb.traj["action"])
    close(lb, la, 2e-4, 1e-6, "loss table")
    for net in ("guider", "actor"):
        pa, pb = getattr(a, net).P.flat, getattr(b, net).P.flat
        assert float((pa - pb).abs().max()) <= 2e-6, f"{net} parameters differ between {mu} micro-batches and one pass"


def test_embed32_four_heads_lbf_setting():
    """n_embd = 32 with 4 heads (both LBF rows of experiment_data/params.csv): head width 8, GroupNorm groups of TWO channels.
    The normalised output is then +-gamma (a - b) / (2 sqrt(var + eps)) with a ~ b at init, which is ill-conditioned in fp32
    for every implementation: the fp32 oracle at embed_dim 32 and the fp32 oracle run on the expanded 64-wide parameters differ
    by ~1 % in the w_v gradients although they agree to 1e-12 in fp64 (scripts/debug/emb_grads.py).  So: rollout bit-exact /
    1e-4 as everywhere, gradients against the oracle at a 5 % bound here; the 2e-3 bound is kept by the 1- and 2-head cases."""
    A, K, TL, maxval, N, T, nb, nh, E = 4, 20, 10, 60, 4, 12, 1, 4, 32
    ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh, E=E)
    ol.rollout(); dl.rollout()
    assert np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy())
    close(dl.traj["value"], ol.traj["value"], 1e-4, 1e-6, "value")
    close(dl.traj["log_prob"], ol.traj["log_prob"], 1e-5, 1e-6, "log_prob")
    ks = oprng.split(ol.key, 4)
    bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
    gg, ag, info, inter = ol.minibatch_grads(ol.make_minibatches(bp, apm)[0])
    dl.minibatch_grads(dl._permutation(ks[1], N)[:N // 2].contiguous(), dl._permutation(ks[2], A))
    close(dl.guider.b.t["t_value"], inter["value"], 1e-4, 1e-6, "train value")
    for n, g in dl.guider.named_grads.items():
        scale = max(gg[n].abs().max().item(), 1e-6)
        close(g / scale, gg[n].reshape(g.shape) / scale, 0, 5e-2, f"guider grad {n}")


def test_linear_lr_decay_matches_oracle():
    """system.decay_learning_rates (mava/utils/training.py:20-64): lr * (1 - (count // (P * M)) / num_updates), count = optax step count."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig
    A, K, TL, maxval, N, T = 3, 10, 9, 30, 6, 10
    spec = ocs.CoordSumSpec(A, K, TL, maxval)
    gp = onets.init_guider_params(1, 64, A + 1, K)
    ap = onets.init_actor_params(2, A + 1, 128, K)
    kw = dict(rollout_length=T, ppo_epochs=2, num_minibatches=2, decay_learning_rates=True, lr_num_updates=3, actor_lr=1e-3)
    ol = olearn.OracleLearner(spec, N, olearn.SystemCfg(**kw), onets.SableCfg(A, K, A + 1), gp, ap)
    key = oprng.split(oprng.prng_key(8), 4)[0]
    ol.setup(key)
    dl = MagpoLearner(CoordSumConfig(A, K, TL, maxval), N, SystemConfig(**kw), DEV, net_seed=None, wgrad_groups=4)
    dl.guider.load_named(gp); dl.actor.load_named(ap)
    dl.setup(key)
    for s in range(2):   # update 1 runs at lr, update 2 at 2/3 lr
        ol.rollout(); dl.rollout()
        assert np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy())
        before = {n: v.clone() for n, v in ol.gp.items()}
        ol.update(); dl.update(); dl._carry_over()
        for n, v in dl.guider.named.items():
            close(v, ol.gp[n].reshape(v.shape), 0, 6e-5 * (s + 1), f"guider param {n} (update {s + 1})")
    assert ol._lr(4) == pytest.approx(1e-3 * 2 / 3) and ol._lr(3) == 1e-3


@pytest.mark.parametrize("E,nh,nb", [(64, 1, 1)])   # (embed-128 nets: measured once, numbers in the docstring; 50 - 75 s of oracle fp64 time each)
def test_dense_bf16_triples_whole_minibatch_gradient_error_vs_fp64(E, nh, nb):
    """Acceptance test for the dense layers on bf16 MFMA with three-piece operand splits (Tuning.linear_variant bit 2; VERDICT r3 item 3 i):
    the gradient of a WHOLE minibatch (rollout of 128 steps, both networks, both losses) against the oracle evaluated in fp64 on the same
    trajectory and parameters.  Measured (round 4): fp32 MFMA 1.043e-7 / bf16 x3 1.047e-7 for the default net; embed-128 / 2-head / 2-block
    nets 8.19e-8 / 8.84e-8 (16 envs) and 1.044e-7 / 1.205e-7 (8 envs) -- both at fp32 rounding level, the triples NOT more accurate (six of
    the nine partial products): "as good as fp32 MFMA" holds to within 0.4 % at embed 64 and 8 - 15 % at embed 128.  With the unexplained
    -3.0 +- 1.3 of the ten-seed learning check (profiles/r03_sweep_return_at_10M.md) that settles it: the dense-layer triples stay OPT-IN
    (MAGPO_LINEAR_BF3=1, ~1 % of the headline step); this test pins their accuracy at <= 1.25 x the fp32-MFMA gradient error."""
    from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig
    from magpo_amd.tuning import Tuning
    A, K, TL, maxval, N, T = 4, 20, 100, 60, 8, 128
    spec = ocs.CoordSumSpec(A, K, TL, maxval)
    scfg = onets.SableCfg(A, K, A + 1, embed_dim=E, n_block=nb, n_head=nh)
    osys = olearn.SystemCfg(rollout_length=T, ppo_epochs=1, num_minibatches=1)
    gp = onets.init_guider_params(1, E, A + 1, K, nb=nb, nh=nh)
    ap = onets.init_actor_params(2, A + 1, 128, K)
    ol = olearn.OracleLearner(spec, N, osys, scfg, gp, ap)
    key = oprng.split(oprng.prng_key(7), 4)[0]
    ol.setup(key)
    ol.rollout()
    ol.rollout()   # second rollout: non-zero rollout-start states
    ks = oprng.split(ol.key, 4)
    bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
    mb = ol.make_minibatches(bp, apm)[0]
    to64 = lambda x: x.double() if torch.is_tensor(x) and x.is_floating_point() else (tuple(to64(y) for y in x) if isinstance(x, tuple) else x)
    ol64 = olearn.OracleLearner(spec, N, osys, scfg, gp, ap, dtype=torch.float64)
    gg, ag, _, _ = ol64.minibatch_grads({k: to64(v) for k, v in mb.items()})
    errs = {}
    for name, variant in (("fp32_mfma", 0), ("bf16x3", 4)):
        t = Tuning()
        t.linear_variant = t.actor_linear_variant = variant
        dl = MagpoLearner(CoordSumConfig(A, K, TL, maxval), N, SystemConfig(rollout_length=T, ppo_epochs=1, num_minibatches=1), DEV,
                          net_seed=None, wgrad_groups=8, n_block=nb, n_head=nh, embed_dim=E, tuning=t)
        dl.guider.load_named(gp); dl.actor.load_named(ap)
        dl.setup(key)
        dl.rollout(); dl._carry_over(); dl.rollout()
        assert np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy())
        dl.minibatch_grads(dl._permutation(ks[1], N), dl._permutation(ks[2], A))
        num = den = 0.0
        worst = ("", 0.0)
        for net, ref in ((dl.guider, gg), (dl.actor, ag)):
            for n, g in net.named_grads.items():
                r = ref[n].reshape(g.shape)
                d = float((g.detach().cpu().double() - r).pow(2).sum())
                num += d; den += float(r.pow(2).sum())
                rel = (d / max(float(r.pow(2).sum()), 1e-30)) ** 0.5
                if rel > worst[1]:
                    worst = (n, rel)
        errs[name] = (num / den) ** 0.5
        print(f"[{E}-{nh}-{nb}] {name}: relative gradient error vs fp64 {errs[name]:.3e} (worst tensor {worst[0]}: {worst[1]:.3e})")
        del dl
        torch.cuda.empty_cache()
    assert errs["fp32_mfma"] < 1e-4, errs
    assert errs["bf16x3"] <= 1.25 * errs["fp32_mfma"] + 1e-9, f"bf16 x3 dense layers are less accurate than fp32 MFMA: {errs}"
