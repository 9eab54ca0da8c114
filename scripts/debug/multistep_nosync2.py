"""Step-2 gradient of the 3-block / 8-agent shape, device vs oracle: (a) unsynced, (b) oracle evaluated at the device's parameters
(rollout states unsynced), per tensor; and which parameters differ after step 1."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_learner_gpu import _mk
from oracle import prng as oprng
A, K, TL, maxval, N, T, nb, nh, E = (8, 15, 9, 100, 4, 11, 3, 1, 64)
ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh, E=E)
ol.rollout(); dl.rollout(); ol.update(); dl.update(); dl._carry_over()
diffs = sorted(((float((v.cpu() - ol.gp[n].reshape(v.shape)).abs().max()), n) for n, v in dl.guider.named.items()), reverse=True)
print("largest parameter differences after step 1:", [(f"{d:.1e}", n) for d, n in diffs[:6]])
ol.rollout(); dl.rollout()
print("step-2 rollout: actions identical", np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy()),
      " |d value|", float((dl.traj["value"].cpu() - ol.traj["value"]).abs().max()), " |d adv|", float((dl.traj["adv"].cpu() - ol.traj["adv"]).abs().max()),
      " |d prev_hs|", [float((d.cpu() - o.permute(2, 1, 0, 3, 4)).abs().max()) for d, o in zip(dl.groups[0].prev_sable_hs, ol.prev_sable_hs)])
ks = oprng.split(ol.key, 4)
bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
mb = ol.make_minibatches(bp, apm)[0]
g_own = ol.minibatch_grads(mb)[0]
saved = ol.gp
ol.gp = {n: dl.guider.named[n].detach().cpu().reshape(v.shape).clone() for n, v in saved.items()}
g_dev_params = ol.minibatch_grads(mb)[0]                       # the oracle's trajectory, the device's parameters
ol.gp = saved
bpd, apd = dl._permutation(ks[1], N), dl._permutation(ks[2], A)
dl.minibatch_grads(bpd[:N // 2].contiguous(), apd, 0, bpd[:N // 2].contiguous())
rows = []
for n, g in dl.guider.named_grads.items():
    s = float(g_own[n].abs().max())
    if s < 1e-9: continue
    gd = g.cpu()
    rows.append((float((gd - g_own[n].reshape(gd.shape)).abs().max()) / s, float((gd - g_dev_params[n].reshape(gd.shape)).abs().max()) / s,
                 float((g_own[n] - g_dev_params[n]).abs().max()) / s, s, n))
rows.sort(reverse=True)
print("tensor: dev vs oracle(own) | dev vs oracle(device params, oracle trajectory) | oracle(own) vs oracle(device params) | scale")
for r in rows[:10]:
    print(f"  {r[4]:30s} {r[0]:.2e} | {r[1]:.2e} | {r[2]:.2e} | {r[3]:.2e}")
