"""Why single-ulp differences show up as 1e-5 .. 1e-3 errors behind a retention layer (VERDICT r2 Weak 2 / Weak 4): at tokens whose
carried state is zero (first timestep of a sequence with a zero rollout-start state, first timestep after an episode end) the retention
output r is only the intra-timestep term, |r| ~ 1e-4 at initialisation, and the GroupNorm behind it (retention.py:289) divides by
sqrt(var(r) + 1e-6) with var(r) ~ 1e-8 << eps... the output is r / 1e-3: every absolute error of r is amplified ~1000 x, and the following
swish gate * W_o feed it into the residual stream.  CPU only (the oracle in fp32 against the oracle in fp64 on identical inputs):
    python scripts/debug/groupnorm_conditioning.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import coordsum as ocs, networks as onets, prng as oprng
torch.manual_seed(0)
A, K, TL, maxval, B, T = 4, 20, 10, 60, 4, 16
cfg = onets.SableCfg(A, K, A + 1)
p32 = onets.init_guider_params(1, 64, A + 1, K)
p64 = {k: v.double() for k, v in p32.items()}
spec = ocs.CoordSumSpec(A, K, TL, maxval)
st, ts = ocs.reset(spec, oprng.split(oprng.prng_key(3), B))
obs, steps, dones = [], [], []
rng = np.random.default_rng(0)
for t in range(T):
    obs.append(ts["observation"]["agents_view"].astype(np.float32)); steps.append(ts["observation"]["step_count"]); dones.append(ts["step_type"] == 2 if t else np.zeros(B, bool))
    st, ts = ocs.step(spec, st, rng.integers(0, K, (B, A)).astype(np.int32))
obs = torch.from_numpy(np.stack(obs, 1).reshape(B, T * A, -1)); steps = torch.from_numpy(np.stack(steps, 1).reshape(B, T * A)).long()
dones = torch.from_numpy(np.stack(dones, 1))[:, :, None].expand(B, T, A).reshape(B, T * A)
out = {}
for name, p, dt in (("fp32", p32, torch.float32), ("fp64", p64, torch.float64)):
    h = torch.zeros(B, 1, 1, 64, 64, dtype=dt)
    x = onets.rmsnorm(onets._obs_encoder(p, obs.to(dt)), p["enc.ln.scale"])
    _, _, ret = onets.msr_chunk(p, "enc.block0.retn.", x, x, x, h[:, :, 0], dones, steps, n_agents=A, nh=1, masked=False, kappas=cfg.kappas)
    value, rep, _ = onets.encoder_chunk(p, cfg, obs.to(dt), h, dones, steps)
    out[name] = (ret.double(), value.double())
r64, v64 = out["fp64"]; r32, v32 = out["fp32"]
pt = lambda x: " ".join(f"{y:.1e}" for y in x.reshape(B, T, A, -1).abs().amax((0, 2, 3)).tolist())
print("episode starts (pos = 0) at t =", sorted(set((steps[:, ::A] == 0).nonzero()[:, 1].tolist())))
print("|r| (retention output before GroupNorm), max per t :", pt(r64))
print("var(r) per row, median per t                       :", " ".join(f"{y:.1e}" for y in r64.var(-1, unbiased=False).reshape(B, T, A).median(0).values.median(1).values.tolist()))
print("fp32 oracle - fp64 oracle, r     max per t         :", pt(r32 - r64))
print("fp32 oracle - fp64 oracle, value max per t         :", pt((v32 - v64)))
print("=> at the state-less timesteps var(r) << eps = 1e-6: GroupNorm returns ~ r / sqrt(eps), an fp32 rounding error of r (1e-9 .. 1e-8) becomes")
print("   1e-6 .. 1e-5 in the normalised row and in everything behind it; the same holds for a 3e-6 parameter difference (step-2 gradients).")
