"""Debug: parameter error after every optimiser step of update step 2 (test flow, no parameter sync)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_learner_gpu import _mk
from oracle import prng as oprng
from oracle import learner as olearn
A, K, TL, maxval, N, T, nb, nh = (8, 15, 9, 100, 4, 11, 3, 1)
ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh)
ol.rollout(); dl.rollout(); ol.update(); dl.update(); dl._carry_over()
def perr():
    worst = max(((v.cpu() - ol.gp[n].reshape(v.shape)).abs().max().item(), n) for n, v in dl.guider.named.items())
    worsta = max(((v.cpu() - ol.ap[n].reshape(v.shape)).abs().max().item(), n) for n, v in dl.actor.named.items())
    return worst, worsta
print("after step 1", perr())
ol.rollout(); dl.rollout()
print("actions equal", np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy()))
key = ol.key
carried, hs_idx = None, None
sys_ = ol.sys
for e in range(2):
    ks = oprng.split(key, 4); key = ks[0]
    bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
    mbs = ol.make_minibatches(bp, apm, carried); carried = ol._epoch_prev_hs
    bpd = torch.from_numpy(bp).cuda().int(); apd = torch.from_numpy(apm).cuda().int()
    hs_idx = bpd if hs_idx is None else hs_idx[bpd.long()].contiguous()
    mbsz = N // 2
    for mi in range(2):
        gg, ag, info, inter = ol.minibatch_grads(mbs[mi])
        dl.minibatch_grads(bpd[mi * mbsz:(mi + 1) * mbsz].contiguous(), apd, 0, hs_idx[mi * mbsz:(mi + 1) * mbsz].contiguous())
        worst = max(((g.cpu() - gg[n].reshape(g.shape)).abs().max().item() / max(gg[n].abs().max().item(), 1e-12), n, gg[n].abs().max().item()) for n, g in dl.guider.named_grads.items())
        gn_o = torch.sqrt(sum((g.double() ** 2).sum() for g in gg.values())).item()
        ol.gp, ol.g_opt, _ = olearn.clip_adam_step(ol.gp, gg, ol.g_opt, sys_.actor_lr, sys_.max_grad_norm)
        ol.ap, ol.a_opt, _ = olearn.clip_adam_step(ol.ap, ag, ol.a_opt, sys_.actor_lr, sys_.max_grad_norm)
        dl.apply_grads(1.0)
        print(f"epoch {e} mb {mi}: worst rel grad err {worst[0]:.2e} ({worst[1]}, scale {worst[2]:.1e}) gnorm oracle {gn_o:.3e} dev {dl.gnorm.cpu().tolist()} -> param err {perr()}")
# where is the error: which elements, what are nu there
n = perr()[0][1]
v = dl.guider.named[n].cpu(); o = ol.gp[n].reshape(v.shape)
i = (v - o).abs().argmax()
print(n, "worst idx", int(i), "dev", v.reshape(-1)[i].item(), "oracle", o.reshape(-1)[i].item(), "nu oracle", ol.g_opt["nu"][n].reshape(-1)[i].item(), "mu oracle", ol.g_opt["mu"][n].reshape(-1)[i].item())
