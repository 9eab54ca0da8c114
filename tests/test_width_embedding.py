"""params.WidthEmbedding: an embed_dim = E < 64 Sable network embedded in the 64-wide device network (every feature carried 64 / E
times, tied parameter copies).  Checked with the ORACLE alone in fp64, independent of any kernel: the oracle at width 64 run on the
expanded parameters (duplicated positional encodings, duplicated start states) must reproduce the E-wide oracle, and the folded
gradients must equal the E-wide gradients."""
import pytest
import torch

from magpo_amd.params import FlatParams, WidthEmbedding, guider_layout, guider_named_views
from oracle import networks as onets


@pytest.mark.parametrize("E,nh,nb", [(32, 1, 2), (32, 4, 1), (16, 2, 1)])
def test_expanded_network_equals_the_narrow_one(E, nh, nb, monkeypatch):
    A, K, F, T, N = 3, 10, 4, 5, 2
    m = 64 // E
    gp = onets.init_guider_params(1, E, F, K, nh=nh, nb=nb)
    g = torch.Generator().manual_seed(3)
    for n in gp:
        if ".ffn." not in n:
            gp[n] = gp[n] + 0.05 * torch.randn(gp[n].shape, generator=g)
    w = WidthEmbedding(E, F, K, nb, nh, "cpu")
    PL = FlatParams(guider_layout(E, F, K, nb, nh), "cpu")
    for n, v in guider_named_views(PL.views(), E, nh).items():
        v.copy_(gp[n].reshape(v.shape))
    PD = FlatParams(guider_layout(64, F, K, nb, nh), "cpu")
    w.expand(PL.flat, PD.flat)
    gp64 = {n: v.clone() for n, v in guider_named_views(PD.views(), 64, nh).items()}

    def run(p, cfg):
        p = {k: v.detach().clone().double().requires_grad_(True) for k, v in p.items()}
        obs = torch.randint(0, 5, (N, T * A, F), generator=torch.Generator().manual_seed(5)).double()
        act = torch.randint(0, K, (N, T * A), generator=torch.Generator().manual_seed(6))
        mask = torch.ones(N, T * A, K, dtype=torch.bool)
        sc = torch.arange(T).repeat_interleave(A)[None].repeat(N, 1)
        hl = E // nh
        states = [torch.randn(N, nh, nb, hl, hl, generator=torch.Generator().manual_seed(7 + i)).double() * 0.2 for i in range(3)]
        if cfg.E == 64 and E < 64:   # device layout of a head state: S'[m i, m j + c] = S[i, j], the rows between are zero
            exp = []
            for s_ in states:
                d = torch.zeros(N, nh, nb, 64 // nh, 64 // nh, dtype=torch.float64)
                for c in range(m):
                    d[..., ::m, c::m] = s_
                exp.append(d)
            states = exp
        dones = torch.zeros(N, T * A, dtype=torch.bool)
        dones[1, 2 * A:3 * A] = True
        value, logp, ent, _ = onets.sable_train(p, cfg, obs, act, mask, sc, tuple(states), dones)
        loss = (value * torch.linspace(0.5, 1.5, value.numel()).reshape(value.shape)).sum() + (logp * torch.linspace(-1, 1, logp.numel()).reshape(logp.shape)).sum() + 0.3 * ent.sum()
        grads = torch.autograd.grad(loss, list(p.values()), allow_unused=True)
        return value.detach(), logp.detach(), {k: (gr if gr is not None else torch.zeros_like(v)) for (k, v), gr in zip(p.items(), grads)}

    v_n, l_n, g_n = run(gp, onets.SableCfg(A, K, F, embed_dim=E, n_head=nh, n_block=nb))
    orig = onets.positional_encoding
    monkeypatch.setattr(onets, "positional_encoding", lambda pos, E_, dt: orig(pos, E, dt).repeat_interleave(m, dim=-1))
    v_w, l_w, g_w = run(gp64, onets.SableCfg(A, K, F, embed_dim=64, n_head=nh, n_block=nb))
    assert float((v_n - v_w).abs().max()) < 1e-10 and float((l_n - l_w).abs().max()) < 1e-10
    GD = FlatParams(guider_layout(64, F, K, nb, nh), "cpu")
    for n, v in guider_named_views(GD.views(), 64, nh).items():
        v.copy_(g_w[n].reshape(v.shape).float())
    GL = torch.empty(PL.numel)
    w.fold(GD.flat, GL)
    for n, v in guider_named_views(PL.views(GL), E, nh).items():
        ref = g_n[n].reshape(v.shape)
        assert float((v.double() - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1e-9) + 1e-12, n
