"""VERDICT r2 Weak 2: "a 3e-6 parameter difference changes a step-2 gradient by 47 %" -- where does that come from?  CPU only.
The oracle learner runs update step 1 in fp32 (3-block net, the shape of the multi-step parity test).  Then the step-2 minibatch gradient of
the guider is evaluated, on ONE fixed trajectory (the step-2 rollout of the unperturbed parameters), at parameters P and P + d with |d| <= 3e-6:
  (a) in fp32 and (b) in fp64, and the change is attributed to the loss terms (value / clipped surrogate / entropy) and to the rows of the batch.
    python scripts/debug/step2_sens_fp64.py"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import coordsum as ocs, learner as olearn, networks as onets, prng as oprng
torch.set_num_threads(4)
A, K, TL, maxval, N, T, nb, nh = 8, 15, 9, 100, 4, 11, 3, 1
spec = ocs.CoordSumSpec(A, K, TL, maxval)
scfg = onets.SableCfg(A, K, A + 1, embed_dim=64, n_block=nb, n_head=nh)
osys = olearn.SystemCfg(rollout_length=T, ppo_epochs=2, num_minibatches=2)
ol = olearn.OracleLearner(spec, N, osys, scfg, onets.init_guider_params(1, 64, A + 1, K, nb=nb, nh=nh), onets.init_actor_params(2, A + 1, 128, K))
ol.setup(oprng.split(oprng.prng_key(42), 4)[0])
ol.rollout(); ol.update()
ol.rollout()                                   # the step-2 trajectory (fixed from here on)
ks = oprng.split(ol.key, 4)
bp, apm = oprng.permutation(ks[1], N), oprng.permutation(ks[2], A)
mb = ol.make_minibatches(bp, apm)[0]
g = torch.Generator().manual_seed(0)
delta = {n: (torch.rand(v.shape, generator=g) * 2 - 1) * 3e-6 for n, v in ol.gp.items()}

def grads(dtype, perturbed, term=None):
    o = copy.copy(ol)
    o.gp = {n: (v.double() + (delta[n].double() if perturbed else 0)).to(dtype) for n, v in ol.gp.items()}
    o.ap = {n: v.to(dtype) for n, v in ol.ap.items()}
    m = {k: (tuple(h.to(dtype) for h in v) if isinstance(v, tuple) else (v.to(dtype) if torch.is_floating_point(v) else v)) for k, v in mb.items()}
    gp = {k: v.detach().clone().requires_grad_(True) for k, v in o.gp.items()}
    ap = {k: v.detach().clone().requires_grad_(True) for k, v in o.ap.items()}
    value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp = olearn.minibatch_forward(o.sys, o.scfg, gp, ap, m)
    gl, ginfo = olearn.guider_loss(o.sys, value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp, m)
    gr = torch.autograd.grad(gl, list(gp.values()), allow_unused=True)
    return {k: (x if x is not None else torch.zeros_like(v)).double() for (k, v), x in zip(gp.items(), gr)}, {k: float(v) for k, v in ginfo.items()}, value.detach().double(), g_logp.detach().double()

names = ("enc.block2.retn.w_k", "enc.block2.retn.w_q", "enc.block0.retn.w_k", "dec.head.dense0.kernel", "enc.head.dense0.kernel")
for dtype in (torch.float32, torch.float64):
    g0, i0, v0, l0 = grads(dtype, False)
    g1, i1, v1, l1 = grads(dtype, True)
    print(f"--- {dtype}: |d value| {float((v1 - v0).abs().max()):.2e}  |d logp| {float((l1 - l0).abs().max()):.2e}   losses {i0}")
    for n in names:
        s = g0[n].abs().max().item()
        print(f"   {n:28s} |grad| {s:.3e}   relative change under the 3e-6 perturbation {float((g1[n] - g0[n]).abs().max()) / s:.3e}")
g32, _, _, _ = grads(torch.float32, False); g64, _, _, _ = grads(torch.float64, False)
print("--- fp32 gradient vs fp64 gradient at the SAME parameters (rounding alone):")
for n in names:
    s = g64[n].abs().max().item()
    print(f"   {n:28s} {float((g32[n] - g64[n]).abs().max()) / s:.3e}")

# (c) the same with the perturbed learner rolling out ITS OWN step-2 trajectory (what scripts/debug/step2_sens.py compared)
print("--- own step-2 rollouts at P and P + d (fp32):")
ola = olearn.OracleLearner(spec, N, osys, scfg, onets.init_guider_params(1, 64, A + 1, K, nb=nb, nh=nh), onets.init_actor_params(2, A + 1, 128, K))
ola.setup(oprng.split(oprng.prng_key(42), 4)[0]); ola.rollout(); ola.update()
olb = olearn.OracleLearner(spec, N, osys, scfg, onets.init_guider_params(1, 64, A + 1, K, nb=nb, nh=nh), onets.init_actor_params(2, A + 1, 128, K))
olb.setup(oprng.split(oprng.prng_key(42), 4)[0]); olb.rollout(); olb.update()
olb.gp = {n: v + delta[n] for n, v in olb.gp.items()}
ola.rollout(); olb.rollout()
same = bool((ola.traj["action"] == olb.traj["action"]).all())
print("   sampled actions identical:", same, " max |d value|", float((ola.traj["value"] - olb.traj["value"]).abs().max()),
      " max |d adv|", float((ola.traj["adv"] - olb.traj["adv"]).abs().max()), " |adv| scale", float(ola.traj["adv"].abs().max()))
ga = ola.minibatch_grads(ola.make_minibatches(bp, apm)[0])[0]; gb = olb.minibatch_grads(olb.make_minibatches(bp, apm)[0])[0]
for n in names:
    s = ga[n].abs().max().item()
    print(f"   {n:28s} relative change {float((ga[n] - gb[n]).abs().max()) / s:.3e}")

# (d) the parameter difference that actually arises between two implementations: Adam turns a RELATIVE gradient error into an update difference
# of lr * err * g / (|g| + eps)-ish, coherent over a whole tensor -- not a random +-3e-6.  Two oracle learners, the second with the step-1 gradients
# multiplied by (1 + 1e-4 * N(0, 1)) (what fp32 summation order does); step-2 gradients on ONE trajectory, in fp32 and in fp64.
print("--- step 1 with gradients perturbed by 1e-4 relative (two fp32 implementations), then the step-2 gradient at both parameter sets:")
def learner():
    o = olearn.OracleLearner(spec, N, osys, scfg, onets.init_guider_params(1, 64, A + 1, K, nb=nb, nh=nh), onets.init_actor_params(2, A + 1, 128, K))
    o.setup(oprng.split(oprng.prng_key(42), 4)[0]); o.rollout(); return o
oa, ob = learner(), learner()
gen = torch.Generator().manual_seed(1)
noise = lambda gg, ag: ({k: v * (1 + 1e-4 * torch.randn(v.shape, generator=gen)) for k, v in gg.items()}, ag)
oa.update(); ob.update(grad_hook=noise)
dif = sorted(((float((oa.gp[n] - ob.gp[n]).abs().max()), n) for n in oa.gp), reverse=True)[:4]
print("   largest parameter differences after step 1:", [(f"{d:.1e}", n) for d, n in dif])
oa.rollout()
mb = oa.make_minibatches(bp, apm)[0]
for dtype in (torch.float32, torch.float64):
    res = []
    for o in (oa, ob):
        gp = {k: v.to(dtype).detach().clone().requires_grad_(True) for k, v in o.gp.items()}
        ap = {k: v.to(dtype).detach().clone().requires_grad_(True) for k, v in oa.ap.items()}
        m = {k: (tuple(h.to(dtype) for h in v) if isinstance(v, tuple) else (v.to(dtype) if torch.is_floating_point(v) else v)) for k, v in mb.items()}
        value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp = olearn.minibatch_forward(oa.sys, oa.scfg, gp, ap, m)
        gl, _ = olearn.guider_loss(oa.sys, value, g_logp, g_ent, g_lp_all, a_lp_all, a_logp, m)
        gr = torch.autograd.grad(gl, list(gp.values()), allow_unused=True)
        res.append({k: (x if x is not None else torch.zeros_like(v)).double() for (k, v), x in zip(gp.items(), gr)})
    for n in ("enc.block2.retn.w_k", "enc.block2.retn.w_q", "enc.block2.retn.w_o", "enc.ln.scale", "dec.head.dense0.kernel"):
        s = float(res[0][n].abs().max())
        print(f"   {str(dtype):14s} {n:26s} |grad| {s:.2e}  relative change {float((res[0][n] - res[1][n]).abs().max()) / s:.2e}")
