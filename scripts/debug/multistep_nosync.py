"""Three consecutive update steps of device and oracle WITHOUT re-synchronising parameters in between (VERDICT r2 Weak 2): drift of the
parameters, agreement of the sampled actions, and the step-2 / step-3 gradient difference of the tensors named in the round-2 note."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_learner_gpu import _mk
from oracle import prng as oprng
for shape in ((8, 15, 9, 100, 4, 11, 3, 1, 64), (4, 20, 10, 60, 8, 16, 1, 1, 64), (5, 20, 9, 80, 4, 11, 2, 2, 64)):
    A, K, TL, maxval, N, T, nb, nh, E = shape
    ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh, E=E)
    print("shape", shape)
    for step in (1, 2, 3):
        ol.rollout(); dl.rollout()
        same = np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy())
        ks = oprng.split(ol.key, 4)
        bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
        gg = ol.minibatch_grads(ol.make_minibatches(bp, apm)[0])[0]
        bpd, apd = dl._permutation(ks[1], N), dl._permutation(ks[2], A)
        dl.minibatch_grads(bpd[:N // 2].contiguous(), apd, 0, bpd[:N // 2].contiguous())
        gerr = max(float((g.cpu() - gg[n].reshape(g.shape)).abs().max()) / max(float(gg[n].abs().max()), 1e-12) for n, g in dl.guider.named_grads.items()
                   if float(gg[n].abs().max()) > 1e-7)
        ol.update(); dl.update(); dl._carry_over()
        drift = max(float((v.cpu() - ref[n].reshape(v.shape)).abs().max()) for net, ref in ((dl.guider, ol.gp), (dl.actor, ol.ap)) for n, v in net.named.items())
        print(f"  step {step}: actions identical {same}; max relative guider-gradient error (first minibatch) {gerr:.2e}; parameter drift after the update {drift:.2e}")
