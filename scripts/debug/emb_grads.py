"""Debug: device-level gradients of the embedded (E=32 in 64-wide kernels) network vs the oracle run on the expanded parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_learner_gpu import _mk
from oracle import prng as oprng, networks as onets, learner as olearn
from magpo_amd.params import guider_named_views
A, K, TL, maxval, N, T, nb, nh, E = (4, 20, 10, 60, 4, 12, 1, 4, 32)
ol, dl = _mk(A, K, TL, maxval, N, T, nb=nb, nh=nh, E=E)
ol.rollout(); dl.rollout()
ks = oprng.split(ol.key, 4)
bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
mbs = ol.make_minibatches(bp, apm)
gg, ag, info, inter = ol.minibatch_grads(mbs[1])
bpd = torch.from_numpy(bp).cuda().int(); apd = torch.from_numpy(apm).cuda().int()
mbsz = N // 2
dl.minibatch_grads(bpd[mbsz:].contiguous(), apd)
# oracle on the expanded 64-wide parameters (duplicated PE)
gp64 = {n: v.detach().cpu().clone() for n, v in guider_named_views(dl.guider.v, 64, nh).items()}
orig = onets.positional_encoding
onets.positional_encoding = lambda pos, E_, dt: orig(pos, E, dt).repeat_interleave(64 // E, dim=-1)
ol64 = olearn.OracleLearner(ol.spec, N, ol.sys, onets.SableCfg(A, K, A + 1, embed_dim=64, n_block=nb, n_head=nh), gp64, ol.ap)
mb = dict(mbs[1])
m = 64 // E
def exp_state(h):   # (mb, nh, nb, hs, hs) logical -> device layout
    hs = 64 // nh
    d = torch.zeros(*h.shape[:3], hs, hs)
    for c in range(m):
        d[..., ::m, c::m] = h
    return d
mb["prev_hs"] = tuple(exp_state(h) for h in mb["prev_hs"])
g64, _, info64, _ = ol64.minibatch_grads(mb)
print("loss oracle32 vs oracle64", info["guider_loss"], info64["guider_loss"], info["value_loss"], info64["value_loss"])
dn = guider_named_views(dl.guider.gv, 64, nh)
for n, g in dn.items():
    ref = g64[n].reshape(g.shape)
    s = ref.abs().max().item()
    e = (g.cpu() - ref).abs().max().item()
    if s > 0 and e / s > 1e-3:
        print(n, "scale %.2e rel err %.2e" % (s, e / s))
print("done")
