"""Oracle learner: the reference quirks that only show over several PPO epochs / update steps."""
import numpy as np
import torch

from oracle import coordsum as ocs
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng


def _mk(P=3, M=2, N=6, T=6, A=2, K=6):
    gp = onets.init_guider_params(5, 64, A + 1, K)
    ap = onets.init_actor_params(6, A + 1, 128, K)
    ol = olearn.OracleLearner(ocs.CoordSumSpec(A, K, 5, 9), N, olearn.SystemCfg(rollout_length=T, ppo_epochs=P, num_minibatches=M),
                              onets.SableCfg(A, K, A + 1), gp, ap)
    ol.setup(oprng.split(oprng.prng_key(9), 4)[0])
    return ol


def test_prev_hstates_permutation_compounds_across_epochs():
    """rec_magpo.py:437 unpacks prev_hstates from the epoch carry, :447 rebinds it to take(prev_hstates, batch_perm) and
    :471 carries the SHUFFLED arrays: epoch e trains sequence i on the trajectory of env p_e[i] but on the rollout-start
    state of env p_1[p_2[... p_e[i]]] (quirk B19)."""
    torch.set_num_threads(1)
    ol = _mk()
    ol.update_step()          # step 1 leaves non-zero retention states
    ol.rollout()
    key = ol.key
    perms, carried = [], None
    for e in range(3):
        ks = oprng.split(key, 4)
        key = ks[0]
        bp, apm = oprng.permutation(ks[1], ol.N), oprng.permutation(ks[2], 2)
        perms.append(bp)
        mbs = ol.make_minibatches(bp, apm, carried)
        carried = ol._epoch_prev_hs
        idx = perms[0]
        for p in perms[1:]:
            idx = idx[p]          # take(take(x, p1), p2) = x[p1[p2]]
        got = torch.cat([mb["prev_hs"][0] for mb in mbs], dim=0)
        assert torch.equal(got, ol.prev_sable_hs[0][torch.from_numpy(idx.astype(np.int64))]), e
        # the trajectory side is NOT compounded: every epoch shuffles the original trajectory with this epoch's permutation
        obs = torch.cat([mb["obs"] for mb in mbs], dim=0)
        want = ol.traj["obs"].index_select(1, torch.from_numpy(bp.astype(np.int64))).index_select(2, torch.from_numpy(apm.astype(np.int64)))
        want = want.transpose(0, 1).reshape(ol.N, -1, want.shape[-1])
        assert torch.equal(obs, want)
    assert any(float(h.abs().max()) > 0 for h in ol.prev_sable_hs), "the test needs non-zero rollout-start states"


def test_compounding_changes_the_update_from_step_two_on():
    """Sensitivity: with the per-epoch (non-compounded) indexing the parameters after update step 2 differ, so the
    multi-step parity tests and the golden fixture can see the quirk."""
    torch.set_num_threads(1)
    a, b = _mk(), _mk()
    orig = b.make_minibatches
    b.make_minibatches = lambda bp, apm, prev=None: orig(bp, apm, None)     # always index the original states
    a.update_step(); b.update_step()
    for n in a.gp:    # step 1 starts from zero states: identical
        assert torch.equal(a.gp[n], b.gp[n]), n
    a.update_step(); b.update_step()
    assert max(float((a.gp[n] - b.gp[n]).abs().max()) for n in a.gp) > 1e-7
