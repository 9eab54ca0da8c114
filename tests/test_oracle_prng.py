"""Oracle PRNG pinned against the Random123 threefry2x32 known-answer vectors (the only golden
vectors that exist for this path: the reference ships no tests)."""
import numpy as np

from oracle import prng


def test_threefry_known_answers():
    kat = [((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6B200159, 0x99BA4EFE)),
           ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
           ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]
    for key, ctr, exp in kat:
        x0, x1 = prng.threefry2x32(np.uint32(key[0]), np.uint32(key[1]), np.uint32(ctr[0]), np.uint32(ctr[1]))
        assert (int(x0), int(x1)) == exp


def test_split_is_threefry_of_counter():
    k = prng.prng_key(42)
    assert k.tolist() == [0, 42]
    s = prng.split(k, 3)
    for i in range(3):
        x0, x1 = prng.threefry2x32(k[0], k[1], np.uint32(0), np.uint32(i))
        assert s[i].tolist() == [int(x0), int(x1)]
    # batched keys
    sb = prng.split(s, 2)
    assert sb.shape == (3, 2, 2) and np.array_equal(sb[1], prng.split(s[1], 2))


def test_uniform_gumbel_randint_permutation():
    k = prng.prng_key(7)
    bits = prng.random_bits(k, 1000)
    u = prng.bits_to_uniform(bits)
    assert u.dtype == np.float32 and (u >= 0).all() and (u < 1).all()
    assert prng.bits_to_uniform(np.array([0], np.uint32), np.finfo(np.float32).tiny, 1.0)[0] == np.finfo(np.float32).tiny
    g = prng.bits_to_gumbel(bits)
    assert np.isfinite(g).all()
    r = prng.randint(k, 5000, 0, 60)
    assert r.min() >= 0 and r.max() < 60 and len(np.unique(r)) == 60
    for n in (4, 8, 16384):
        p = prng.permutation(k, n)
        assert np.array_equal(np.sort(p), np.arange(n))
    assert np.array_equal(prng.permutation(k, 16), prng.permutation(k, 16))


def test_categorical_matches_manual_gumbel_argmax():
    k = prng.prng_key(3)
    logits = np.random.default_rng(0).normal(size=(50, 1, 7)).astype(np.float32)
    a = prng.categorical(k, logits)
    g = prng.bits_to_gumbel(prng.random_bits(k, logits.size)).reshape(logits.shape)
    assert np.array_equal(a, np.argmax(g + logits, -1))


def test_choice_hand_worked_vector_and_branches():
    """jax.random.choice restated (oracle/prng.py:choice), the four branches of the published algorithm.

    Hand-worked p / replace=True case, key = PRNGKey(7): random_bits(key, 1) = 2895194379 = 0xAC9129CB, so
    uniform = bitcast((bits >> 9) | 0x3F800000) - 1 = 0.6740899; mask = [0,1,1,0,1,0,0,1] has cumulative counts [0,1,2,2,3,3,3,4];
    r = 4 * (1 - 0.6740899) = 1.3036404; searchsorted(left) = first index whose count >= r = index 2."""
    k = prng.prng_key(7)
    assert int(prng.random_bits(k, 1)[0]) == 2895194379
    u = prng.uniform(k, 1)[0]
    assert abs(float(u) - 0.6740899) < 1e-7
    mask = np.array([0, 1, 1, 0, 1, 0, 0, 1], bool)
    assert prng.choice(k, 8, 1, True, mask).tolist() == [2]
    # the draw is "the ceil(r)-th set cell": exhaustively over keys, always a set cell and all of them reachable
    seen = set()
    for s in range(200):
        kk = prng.prng_key(1000 + s)
        c = int(prng.choice(kk, 8, 1, True, mask)[0])
        r = 4 * (1.0 - float(prng.uniform(kk, 1)[0]))
        assert mask[c] and c == np.nonzero(mask)[0][int(np.ceil(r)) - 1]
        seen.add(c)
    assert seen == {1, 2, 4, 7}
    assert prng.choice(k, 8, 1, True, np.zeros(8, bool)).tolist() == [0]          # all-zero p: searchsorted of 0 in zeros
    # p, replace=False: Gumbel top-k = the k largest of gumbel + log p in descending order, masked cells never before set ones
    g = prng.gumbel(k, 8)
    want = [i for i in np.argsort(-g, kind="stable") if mask[i]][:3]
    assert prng.choice(k, 8, 3, False, mask).tolist() == want
    full = prng.choice(k, 8, 8, False, mask).tolist()
    assert full[:4] == [i for i in np.argsort(-g, kind="stable") if mask[i]] and full[4:] == [0, 3, 5, 6]
    # no p: randint / permutation prefix
    assert np.array_equal(prng.choice(k, 8, 5, True), prng.randint(k, 5, 0, 8))
    assert np.array_equal(prng.choice(k, 110, 4, False), prng.permutation(k, 110)[:4])
    import pytest
    with pytest.raises(ValueError):
        prng.choice(k, 3, 4, False)


def test_same_seed_initialisation_product_equals_oracle_and_has_the_reference_distributions():
    """rec_magpo.py:598-604,623: the networks' parameters are a function of (net_key, actor_net_key).  The product's host-side initialiser
    (magpo_amd/params.py: flax's per-parameter keys -> jax.random.normal / truncated_normal / orthogonal, restated) must give the oracle's
    arrays (oracle/prng.py, oracle/networks.py) bit for bit, and both must have the reference's distributions: normal(1 / E) retention
    projections, orthogonal(sqrt 2 | 0.01 | 1) dense kernels, lecun-normal GRU input kernels, ones / zeros elsewhere.  UNPINNED against
    JAX / flax themselves (absent here)."""
    import math
    import torch
    from magpo_amd.params import (FlatParams, actor_layout, actor_named_views, guider_layout, guider_named_views, init_actor_from_key,
                                  init_guider_from_key)
    from oracle import networks as onets
    ks = prng.split(prng.prng_key(42), 4)
    actor_net_key, net_key = ks[2], ks[3]
    for E, nh, nb, F, K in ((64, 1, 1, 5, 20), (32, 4, 2, 14, 6), (128, 2, 3, 75, 5)):
        og = onets.init_guider_params_from_key(net_key, E, F, K, nh, nb)
        P = FlatParams(guider_layout(E, F, K, nb, nh), "cpu")
        named = guider_named_views(P.views(), E, nh)
        init_guider_from_key(named, net_key, E, nh)
        assert set(named) == set(og)
        for n, v in named.items():
            assert np.array_equal(v.numpy().reshape(-1), og[n].numpy().reshape(-1)), f"guider {n} (E={E})"
        wq = og["enc.block0.retn.w_q"].numpy()
        assert abs(float(wq.std()) * E - 1.0) < 0.05 and abs(float(wq.mean())) * E < 0.05
        for n, gain in (("enc.obs.dense.kernel", math.sqrt(2)), ("dec.head.dense0.kernel", math.sqrt(2)), ("dec.head.dense1.kernel", 0.01),
                        ("dec.act.kernel", math.sqrt(2)), ("enc.head.dense1.kernel", 0.01)):
            w = og[n].numpy().astype(np.float64)
            g = w @ w.T if w.shape[0] < w.shape[1] else w.T @ w
            assert np.abs(g - gain * gain * np.eye(g.shape[0])).max() < 1e-5 * max(1.0, gain * gain), n
        assert float(og["enc.block0.ffn.w1"].abs().max() if "enc.block0.ffn.w1" in og else 0.0) == 0.0
        assert all(float((v - 1).abs().max()) == 0.0 for n, v in og.items() if n.endswith("scale"))
    F, H, K = 5, 128, 20
    oa = onets.init_actor_params_from_key(actor_net_key, F, H, K)
    PA = FlatParams(actor_layout(F, H, K), "cpu")
    an = actor_named_views(PA.views())
    init_actor_from_key(an, actor_net_key)
    for n, v in an.items():
        assert np.array_equal(v.numpy().reshape(-1), oa[n].numpy().reshape(-1)), f"actor {n}"
    wi = oa["gru.ir.kernel"].numpy()
    assert abs(float(wi.std()) * math.sqrt(H) - 1.0) < 0.03 and float(np.abs(wi).max()) <= 2.0 / math.sqrt(H) / 0.87962566103423978 + 1e-6
    wh = oa["gru.hz.kernel"].numpy().astype(np.float64)
    assert np.abs(wh.T @ wh - np.eye(H)).max() < 1e-5
    assert not np.array_equal(oa["gru.hr.kernel"].numpy(), oa["gru.hz.kernel"].numpy())     # one key per parameter path
    # the two sampler pieces that have closed forms: fold_in and Giles' erf_inv against scipy
    from scipy.special import erfinv
    u = np.linspace(-0.99, 0.99, 1001).astype(np.float32)
    assert np.max(np.abs(prng.erf_inv_f32(u) - erfinv(u.astype(np.float64)))) < 2e-6
    assert np.array_equal(prng.fold_in(net_key, 7), prng.split(net_key, 8)[7])                # both are threefry(key, (0, i))
