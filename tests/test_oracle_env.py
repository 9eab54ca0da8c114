"""Oracle CoordSum stack against the hand-worked episode of SURVEY Appendix D1 and the wrapper rules."""
import numpy as np

from oracle import coordsum as cs
from oracle import prng


def _forced(spec, target):
    st, ts = cs.reset(spec, prng.split(prng.prng_key(0), 1))
    st["target"][:] = np.asarray(target, np.int32)[None]
    return st, ts


def test_appendix_d1_episode():
    spec = cs.CoordSumSpec(2, 3, 4, 5)
    st, _ = _forced(spec, [4, 1, 4, 4, 2])
    rewards, obs = [], []
    for a in ([2, 2], [1, 0], [2, 2], [1, 2]):
        st, ts = cs.step(spec, st, np.array([a]), auto_reset=False)
        rewards.append(float(ts["reward"][0, 0]))
        obs.append(int(ts["observation"]["agents_view"][0, 0, -1]))
    assert rewards == [2.0, 2.0, 1.0, 0.0]          # clamped row, first-max guess, hit/miss
    assert obs == [1, 4, 4, 2]
    assert ts["step_type"][0] == cs.STEP_LAST and ts["discount"][0, 0] == 0.0
    assert ts["episode_metrics"]["episode_return"][0] == 5.0 and ts["episode_metrics"]["episode_length"][0] == 4
    assert st["record"][0].tolist() == [[-1, -1, -1, -1], [-1, 1, -1, -1], [2, -1, 2, 1]]


def test_auto_reset_keeps_reward_and_swaps_obs():
    spec = cs.CoordSumSpec(2, 3, 2, 5)
    keys = prng.split(prng.prng_key(5), 3)
    st, ts0 = cs.reset(spec, keys)
    assert ts0["observation"]["agents_view"].shape == (3, 2, 3)
    assert np.array_equal(ts0["observation"]["agents_view"][:, :, :2], np.broadcast_to(np.eye(2, dtype=np.int32), (3, 2, 2)))
    old_key = st["key"].copy()
    st, ts = cs.step(spec, st, np.zeros((3, 2), np.int32))
    st, ts = cs.step(spec, st, np.zeros((3, 2), np.int32))
    assert (ts["step_type"] == cs.STEP_LAST).all()
    assert (st["step_count"] == 0).all() and (ts["observation"]["step_count"] == 0).all()
    assert (st["record"] == -1).all()
    # key, _ = split(state.key); reset(key): key2, target_key = split(key)
    k1 = prng.split(old_key, 2)[:, 0]
    assert np.array_equal(st["key"], prng.split(k1, 2)[:, 0])
    assert np.array_equal(ts["observation"]["agents_view"][:, 0, -1], st["target"][:, 0])
    assert (ts["episode_metrics"]["episode_length"] == 2).all() and (st["running_length"] == 0).all()


def test_eval_env_steps_past_termination():
    spec = cs.CoordSumSpec(2, 3, 3, 5)
    st, _ = _forced(spec, [1, 2, 3, 4])
    for _ in range(5):  # evaluator scans time_limit + 1 steps without auto-reset (evaluator.py:141-148)
        st, ts = cs.step(spec, st, np.array([[0, 1]]), auto_reset=False)
    assert st["step_count"][0] == 5 and ts["step_type"][0] == cs.STEP_LAST
    assert int(ts["observation"]["agents_view"][0, 0, -1]) == 4  # clamped target index
