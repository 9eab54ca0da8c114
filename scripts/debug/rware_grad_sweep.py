"""Guider-gradient error of the device against the oracle on the RWARE wide-observation path over network shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_rware_gpu import _mk
from oracle import prng as oprng
for E, nh, nb in ((64, 1, 3), (64, 2, 2), (128, 2, 1), (128, 2, 2), (128, 2, 3), (128, 1, 3), (128, 4, 3)):
    N, T = 8, 16
    ol, dl = _mk((8, 1, 3, 4, 1, 4, 11), N, T, E=E, nh=nh, nb=nb)
    ol.rollout(); dl.rollout()
    same = np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy())
    ks = oprng.split(ol.key, 4)
    bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], 4)
    gg = ol.minibatch_grads(ol.make_minibatches(bp, apm)[1])[0]
    dl.minibatch_grads(dl._permutation(ks[1], N)[N // 2:].contiguous(), dl._permutation(ks[2], 4))
    errs = sorted(((float((g.cpu() - gg[n].reshape(g.shape)).abs().max()) / max(float(gg[n].abs().max()), 1e-9), n) for n, g in dl.guider.named_grads.items()), reverse=True)
    print(f"E={E} nh={nh} nb={nb}: actions identical {same}; worst gradients:", [(f"{e:.1e}", n) for e, n in errs[:4]])
