"""Debug: is the step-2 gradient difference a property of the parameters (oracle evaluated at the device's parameters)?"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_learner_gpu import _mk
from oracle import prng as oprng
A, K, TL, maxval, N, T, nb, nh = (8, 15, 9, 100, 4, 11, 3, 1)
ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh)
ol.rollout(); dl.rollout(); ol.update(); dl.update(); dl._carry_over()
# second oracle that continues from the DEVICE's parameters
ol2 = copy.deepcopy(ol)
ol2.gp = {n: dl.guider.named[n].detach().cpu().reshape(v.shape).clone() for n, v in ol.gp.items()}
ol2.ap = {n: dl.actor.named[n].detach().cpu().reshape(v.shape).clone() for n, v in ol.ap.items()}
ol.rollout(); ol2.rollout(); dl.rollout()
ks = oprng.split(ol.key, 4)
bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
g1 = ol.minibatch_grads(ol.make_minibatches(bp, apm)[0])[0]
g2 = ol2.minibatch_grads(ol2.make_minibatches(bp, apm)[0])[0]
bpd = torch.from_numpy(bp).cuda().int(); apd = torch.from_numpy(apm).cuda().int()
dl.minibatch_grads(bpd[:N // 2].contiguous(), apd, 0, bpd[:N // 2].contiguous())
for n in ("enc.block2.retn.w_k", "enc.block2.retn.w_q", "enc.block1.retn.w_k", "dec.head.dense0.kernel"):
    gd = dl.guider.named_grads[n].cpu()
    s = g1[n].abs().max().item()
    print(n, "scale", f"{s:.2e}", "oracle(own) vs oracle(dev params)", f"{(g1[n]-g2[n]).abs().max().item()/s:.2e}",
          "dev vs oracle(dev params)", f"{(gd-g2[n].reshape(gd.shape)).abs().max().item()/s:.2e}",
          "dev vs oracle(own)", f"{(gd-g1[n].reshape(gd.shape)).abs().max().item()/s:.2e}")
d = max((ol.gp[n] - ol2.gp[n]).abs().max().item() for n in ol.gp)
print("max param diff oracle vs device", d)
for k in ("value", "log_prob", "adv"):
    print(k, (ol.traj[k] - ol2.traj[k]).abs().max().item(), (ol2.traj[k] - dl.traj[k].cpu()).abs().max().item())
