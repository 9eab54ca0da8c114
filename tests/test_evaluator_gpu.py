"""Anakin evaluator (magpo_amd/evaluator.py) against the CPU restatement of mava/evaluator.py:82-208
(oracle/evaluator.py): per-episode episode_return / episode_length arrays for fixed actor parameters and key,
over several episode loops (PRNG chain across loops) and both sampled and greedy acting."""
import numpy as np
import pytest
import torch

from oracle import coordsum as ocs
from oracle import evaluator as oeval
from oracle import networks as onets
from oracle import prng as oprng

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scenario,TL,num_envs,episodes,greedy", [("3x10-30", 9, 4, 12, False), ("5x20-80", 7, 6, 6, False),
                                                                  ("8x15-100", 6, 3, 7, True), ("3x30-50", 11, 5, 10, False)])
def test_evaluator_matches_oracle(scenario, TL, num_envs, episodes, greedy):
    from magpo_amd.actor import GruActor
    from magpo_amd.config import compose
    from magpo_amd.evaluator import get_eval_fn, get_num_eval_envs, make_rec_eval_act_fn
    from magpo_amd.utils import make_env as environments
    cfg = compose("rec_magpo", ["env=coordsum", f"env/scenario={scenario}", f"arch.num_envs={num_envs}", f"arch.num_eval_episodes={episodes}",
                                f"env.kwargs.time_limit={TL}", f"arch.evaluation_greedy={greedy}"])
    env, eval_env = environments.make(cfg)
    A, K = env.num_agents, env.action_dim
    ap = onets.init_actor_params(17, A + 1, 128, K)
    # biases away from zero and a head with a visible spread so that sampling is not uniform
    g = torch.Generator().manual_seed(3)
    ap["head.kernel"] = ap["head.kernel"] * 60
    for n in ("gru.ir.bias", "gru.iz.bias", "gru.in.bias", "gru.hn.bias", "pre.bias", "post.bias"):
        ap[n] = torch.randn(ap[n].shape, generator=g) * 0.1
    actor = GruActor(A, K, A + 1, "cuda")
    act_fn = make_rec_eval_act_fn(actor, cfg)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        evaluator = get_eval_fn(eval_env, act_fn, cfg, absolute_metric=False, device="cuda")
    n = get_num_eval_envs(cfg, False)
    key = oprng.split(oprng.prng_key(23), 3)[1]
    got = evaluator({k: v.cuda() for k, v in ap.items()}, key, {"hidden_state": torch.zeros(n * A, 128, device="cuda")})
    spec = ocs.CoordSumSpec(A, K, TL, env.cfg.maxval)
    want = oeval.evaluate(spec, ap, key, num_envs, episodes, greedy=greedy)
    assert got["episode_return"].shape == want["episode_return"].shape
    assert np.array_equal(got["episode_length"], want["episode_length"])
    assert np.array_equal(got["episode_return"], want["episode_return"]), (got["episode_return"], want["episode_return"])
    if not greedy:   # (8 greedy agents essentially never hit the target sum)
        assert want["episode_return"].max() > 0, "the test should see at least one rewarded step"
    assert got["steps_per_second"] > 0
