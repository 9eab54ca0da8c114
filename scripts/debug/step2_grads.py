"""Debug: gradient parity of the minibatches of update step 2 (non-zero rollout-start states) against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_learner_gpu import _mk
from oracle import prng as oprng
A, K, TL, maxval, N, T, nb, nh = [int(x) for x in sys.argv[1:9]] if len(sys.argv) > 8 else (8, 15, 9, 100, 4, 11, 3, 1)
ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh)
ol.rollout(); dl.rollout(); ol.update(); dl.update(); dl._carry_over()
def perr(tag):
    e = max((v.cpu() - ol.gp[n].reshape(v.shape)).abs().max().item() for n, v in dl.guider.named.items())
    print(tag, "max guider param err", e)
perr("after step 1")
# sync parameters exactly so that step 2 isolates the gradient path
dl.guider.load_named(ol.gp); dl.actor.load_named(ol.ap)
ol.rollout(); dl.rollout()
print("actions equal", np.array_equal(dl.traj["action"].cpu().numpy(), ol.traj["action"].numpy()))
for d, o in zip(dl.groups[0].prev_sable_hs, ol.prev_sable_hs):
    hs = 64 // nh
    print("prev hs err", (d[:, :, :, :hs, :hs].cpu() - o.permute(2, 1, 0, 3, 4)).abs().max().item(), "scale", o.abs().max().item())
key = ol.key
carried, hs_idx = None, None
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
        worst = []
        for n, g in dl.guider.named_grads.items():
            scale = max(gg[n].abs().max().item(), 1e-12)
            worst.append(((g.cpu() - gg[n].reshape(g.shape)).abs().max().item() / scale, n, scale))
        worst.sort(reverse=True)
        print(f"epoch {e} mb {mi}: worst rel grad errs", [(f"{w[0]:.2e}", w[1], f"{w[2]:.1e}") for w in worst[:4]])
