"""Debug: training-forward encoder stages of the device vs fp64 recomputation at the basic parity shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_learner_gpu import _mk
from oracle import prng as oprng, networks as onets
A, K, TL, maxval, N, T = 4, 20, 10, 60, 8, 16
ol, dl = _mk(A, K, TL, maxval, N, T)
dl.class_tables = False
ol.rollout(); dl.rollout()
ks = oprng.split(ol.key, 4)
bpd, apd = dl._permutation(ks[1], N), dl._permutation(ks[2], A)
mbsz = N // 2
dl.minibatch_grads(bpd[mbsz:2 * mbsz].contiguous(), apd)
bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
_, _, _, inter = ol.minibatch_grads(ol.make_minibatches(bp, apm)[1])
m, b = dl._mb, dl.guider.b.t
p = {k: v.double() for k, v in ol.gp.items()}
R = mbsz * T * A
obs = m["obs"].cpu().double().reshape(mbsz, T * A, -1)
pos = m["pos"].cpu().long().reshape(mbsz, T * A)
dones = m["done"].cpu().bool()[:, :, None].expand(mbsz, T, A).reshape(mbsz, T * A)
x0 = onets._obs_encoder(p, obs)
xn = onets.rmsnorm(x0, p["enc.ln.scale"])
pe = onets.positional_encoding(pos, 64, torch.float64)
def rep_err(name, dev, ref):
    e = (dev.cpu().double().reshape(ref.shape) - ref).abs().reshape(mbsz, T, A, -1).amax((0, 2, 3))
    print(f"{name:8s} max {e.max():.2e}  per t:", " ".join(f"{x:.0e}" for x in e.tolist()))
rep_err("xn0", b["t_xn0"], xn)
rep_err("kin0", b["t_kin0"], xn + pe)
kin = xn + pe
q = kin @ p["enc.block0.retn.w_q"][0]; k = kin @ p["enc.block0.retn.w_k"][0]; v = kin @ p["enc.block0.retn.w_v"][0]
rep_err("q", b["t_qkvg0"][:, 0:64], q); rep_err("k", b["t_qkvg0"][:, 64:128], k); rep_err("v", b["t_qkvg0"][:, 128:192], v)
rep_err("g", b["t_qkvg0"][:, 192:256], kin @ p["enc.block0.retn.w_g"])
cfg = onets.SableCfg(A, K, A + 1)
h0 = torch.zeros(mbsz, 1, 64, 64, dtype=torch.float64)
out, _, ret = onets.msr_chunk(p, "enc.block0.retn.", xn, xn, xn, h0, dones, pos, n_agents=A, nh=1, masked=False, kappas=cfg.kappas)
rep_err("r0", b["t_r0"], ret)
rn = onets.groupnorm_rows(ret.reshape(-1, 64), p["enc.block0.retn.gn.scale"], p["enc.block0.retn.gn.bias"], 1).reshape(ret.shape)
u = onets.swish(kin @ p["enc.block0.retn.w_g"]) * rn
rep_err("u0", b["t_u0"], u)
rep_err("y0", b["t_y0"], out)
x1 = onets.rmsnorm(xn + out, p["enc.block0.ln1.scale"]); rep = onets.rmsnorm(x1, p["enc.block0.ln2.scale"])
rep_err("rep", b["t_rep"], rep)
rep_err("value", b["t_value"], onets._value_head(p, rep))
print("var(r) median per t:", " ".join(f"{y:.0e}" for y in ret.var(-1, unbiased=False).reshape(mbsz, T, A).median(0).values.median(1).values.tolist()))
print("pos seq0 agent0:", pos[0, ::A].tolist())
print("reference value vs the oracle's own:", float((onets._value_head(p, rep).reshape(-1) - inter["value"].double().reshape(-1)).abs().max()))
D = b["t_r0"].cpu().double().reshape(ret.shape); print("r0 dev/ref ratio sample:", (D[0, 4:8, :3] / ret[0, 4:8, :3]).tolist())
