"""Algebraic pins of the network oracle: recurrent == chunkwise retention (the reference's two forms must
agree for any done pattern and chunking), Appendix D2-D4 worked examples, fp64 finite differences."""
import numpy as np
import torch

from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng


def test_decay_matrix_and_xi_appendix_d3():
    f64 = torch.float64
    d = lambda *x: torch.tensor([list(x)], dtype=torch.bool)
    D = onets.decay_matrix(d(0, 0, 0), 1, 0.5, True, f64)[0]
    assert torch.allclose(D, torch.tensor([[1, 0, 0], [.5, 1, 0], [.25, .5, 1]], dtype=f64))
    assert torch.allclose(onets.xi_vector(d(0, 0, 0), 1, 0.5, f64)[0, :, 0], torch.tensor([.5, .25, .125], dtype=f64))
    D = onets.decay_matrix(d(0, 0, 1), 1, 0.5, True, f64)[0]
    assert torch.allclose(D, torch.tensor([[1, 0, 0], [.5, 1, 0], [0, 0, 1]], dtype=f64))
    assert torch.allclose(onets.xi_vector(d(0, 0, 1), 1, 0.5, f64)[0, :, 0], torch.tensor([.5, .25, 0], dtype=f64))
    assert torch.allclose(onets.decay_matrix(d(1, 0, 0), 1, 0.5, True, f64)[0], onets.decay_matrix(d(0, 0, 0), 1, 0.5, True, f64)[0])
    assert torch.allclose(onets.xi_vector(d(1, 0, 0), 1, 0.5, f64)[0, :, 0], torch.zeros(3, dtype=f64))
    # A = 2: encoder keeps the full 2x2 diagonal blocks, decoder zeroes the strictly-upper entry
    dn = torch.tensor([[0, 0, 0, 0]], dtype=torch.bool)
    De = onets.decay_matrix(dn, 2, 0.5, False, f64)[0]
    Dd = onets.decay_matrix(dn, 2, 0.5, True, f64)[0]
    assert De[0, 1] == 1 and Dd[0, 1] == 0 and De[2, 1] == 0.5 and Dd[3, 2] == 1


def test_shifted_actions_appendix_d4():
    sh = onets.shifted_actions(torch.tensor([[1, 2, 0, 1]]), 3, 2, torch.float32)[0]
    assert sh.tolist() == [[1, 0, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0]]


def test_gae_appendix_d2_and_direct_sum():
    r = torch.tensor([1.0, 0.0], dtype=torch.float64).reshape(2, 1, 1)
    v = torch.tensor([0.5, 0.2], dtype=torch.float64).reshape(2, 1, 1)
    d = torch.zeros(2, 1, 1, dtype=torch.bool)
    lv = torch.full((1, 1), 0.1, dtype=torch.float64)
    adv, tg = olearn.calculate_gae(r, v, d, lv, torch.zeros(1, 1, dtype=torch.bool), 0.99, 0.95)
    assert torch.allclose(adv.flatten(), torch.tensor([0.6030095, -0.101], dtype=torch.float64))
    assert torch.allclose(tg.flatten(), torch.tensor([1.1030095, 0.099], dtype=torch.float64))
    adv, tg = olearn.calculate_gae(r, v, d, lv, torch.ones(1, 1, dtype=torch.bool), 0.99, 0.95)
    assert torch.allclose(adv.flatten(), torch.tensor([0.5099, -0.2], dtype=torch.float64))
    # O(T^2) direct sum on random data
    g = torch.Generator().manual_seed(0)
    T = 9
    r, v = torch.randn(T, 3, 2, generator=g, dtype=torch.float64), torch.randn(T, 3, 2, generator=g, dtype=torch.float64)
    d = torch.rand(T, 3, 2, generator=g) < 0.3
    lv, ld = torch.randn(3, 2, generator=g, dtype=torch.float64), torch.rand(3, 2, generator=g) < 0.5
    adv, _ = olearn.calculate_gae(r, v, d, lv, ld, 0.9, 0.8)
    nv = torch.cat([v[1:], lv[None]]); nd = torch.cat([d[1:], ld[None]]).double()
    delta = r + 0.9 * nv * (1 - nd) - v
    ref = torch.zeros_like(adv)
    for t in range(T):
        w = torch.ones(3, 2, dtype=torch.float64)
        for s in range(t, T):
            ref[t] += w * delta[s]
            w = w * 0.9 * 0.8 * (1 - nd[s])
    assert torch.allclose(adv, ref)


def _random_case(A=3, K=5, B=2, T=7, seed=0, ffn=True):
    g = torch.Generator().manual_seed(seed)
    dt = torch.float64
    cfg = onets.SableCfg(A, K, A + 1)
    gp = onets.init_guider_params(seed + 1, 64, A + 1, K, dtype=dt, randomize_ffn=ffn)
    obs = torch.randint(0, 30, (B, T, A, A + 1), generator=g).to(dt)
    act = torch.randint(0, K, (B, T, A), generator=g)
    sc = torch.randint(0, 50, (B, T, 1), generator=g).expand(B, T, A)
    dones_t = torch.rand(B, T, generator=g) < 0.3
    mask = torch.ones(B, T, A, K, dtype=torch.bool)
    hs = tuple(torch.randn(B, 1, 1, 64, 64, generator=g, dtype=dt) * 0.1 for _ in range(3))
    return cfg, gp, obs, act, sc, dones_t, mask, hs


def test_recurrent_equals_chunkwise_with_dones_and_chunks():
    cfg, gp, obs, act, sc, dones_t, mask, hs = _random_case()
    B, T, A = act.shape
    dones = dones_t[:, :, None].expand(B, T, A)
    flat = lambda x: x.reshape(B, T * A, *x.shape[3:])
    v, lp, ent, _ = onets.sable_train(gp, cfg, flat(obs), flat(act), flat(mask), flat(sc), hs, flat(dones))
    h = hs
    vs, lps = [], []
    for t in range(T):
        h = tuple(torch.where(dones_t[:, t][:, None, None, None, None], torch.zeros_like(x), x) for x in h)
        _, l, val, h, _ = onets.sable_get_actions(gp, cfg, obs[:, t], mask[:, t], sc[:, t], h, prng.prng_key(0), forced_actions=act[:, t])
        vs.append(val); lps.append(l)
    assert torch.allclose(torch.stack(vs, 1).reshape(B, -1), v, atol=1e-12)
    assert torch.allclose(torch.stack(lps, 1).reshape(B, -1), lp, atol=1e-12)
    for cts in (1, 2, 7):
        if T % cts:
            continue
        cfg2 = onets.SableCfg(cfg.A, cfg.K, cfg.F, chunk_timesteps=cts)
        v2, lp2, _, _ = onets.sable_train(gp, cfg2, flat(obs), flat(act), flat(mask), flat(sc), hs, flat(dones))
        assert torch.allclose(v2, v, atol=1e-12) and torch.allclose(lp2, lp, atol=1e-12)


def test_swiglu_zero_init_is_identically_zero_with_zero_grads():
    cfg, gp, obs, act, sc, dones_t, mask, hs = _random_case(ffn=False)
    B, T, A = act.shape
    flat = lambda x: x.reshape(B, T * A, *x.shape[3:])
    p = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    v, lp, ent, _ = onets.sable_train(p, cfg, flat(obs), flat(act), flat(mask), flat(sc), hs, flat(dones_t[:, :, None].expand(B, T, A)))
    (v.sum() + lp.sum() + ent.sum()).backward()
    for n, t in p.items():
        if ".ffn." in n:
            assert t.grad is None or float(t.grad.abs().max()) == 0.0, n


def test_guider_loss_gradient_finite_difference():
    cfg, gp, obs, act, sc, dones_t, mask, hs = _random_case(A=2, K=4, B=2, T=4, ffn=False)
    B, T, A = act.shape
    flat = lambda x: x.reshape(B, T * A, *x.shape[3:])
    sysc = olearn.SystemCfg()
    g = torch.Generator().manual_seed(3)
    mb = dict(log_prob=torch.randn(B, T * A, generator=g, dtype=torch.float64) * 0.1 - 1.4, adv=torch.randn(B, T * A, generator=g, dtype=torch.float64),
              value=torch.randn(B, T * A, generator=g, dtype=torch.float64), targets=torch.randn(B, T * A, generator=g, dtype=torch.float64))
    a_lp = torch.log_softmax(torch.randn(B, T * A, cfg.K, generator=g, dtype=torch.float64), -1)
    a_logp = a_lp.gather(-1, flat(act)[..., None])[..., 0]

    def loss(p):
        v, lp, ent, lpa = onets.sable_train(p, cfg, flat(obs), flat(act), flat(mask), flat(sc), hs, flat(dones_t[:, :, None].expand(B, T, A)))
        return olearn.guider_loss(sysc, v, lp, ent, lpa, a_lp, a_logp, mb)[0]

    name = "dec.block0.retn2.w_k"
    p = {k: v.clone() for k, v in gp.items()}
    p[name].requires_grad_(True)
    (gr,) = torch.autograd.grad(loss(p), [p[name]])
    idx = (0, 5, 7)
    eps = 1e-6
    pp = {k: v.clone() for k, v in gp.items()}; pp[name][idx] += eps
    pm = {k: v.clone() for k, v in gp.items()}; pm[name][idx] -= eps
    fd = (loss(pp) - loss(pm)) / (2 * eps)
    assert abs(float(fd) - float(gr[idx])) < 1e-6 * max(1.0, abs(float(fd)))


def test_adam_and_clip_closed_forms():
    p = {"w": torch.tensor([1.0, -2.0, 3.0], dtype=torch.float64)}
    g = {"w": torch.tensor([0.1, -0.2, 0.05], dtype=torch.float64)}
    newp, opt, gn = olearn.clip_adam_step(p, g, olearn.adam_init(p), 1e-3, 0.5)
    assert abs(float(gn) - float(g["w"].norm())) < 1e-12
    # step 1: mu_hat = g, nu_hat = g^2  =>  u = -lr * g / (|g| + eps)
    assert torch.allclose(newp["w"], p["w"] - 1e-3 * g["w"] / (g["w"].abs() + 1e-5))
    big = {"w": g["w"] * 100}
    newp2, _, gn2 = olearn.clip_adam_step(p, big, olearn.adam_init(p), 1e-3, 0.5)
    clipped = big["w"] / gn2 * 0.5
    assert abs(float(clipped.norm()) - 0.5) < 1e-12
    assert torch.allclose(newp2["w"], p["w"] - 1e-3 * clipped / (clipped.abs() + 1e-5))
