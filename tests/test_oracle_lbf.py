"""Level-Based Foraging restatement (oracle/lbf.py; UNPINNED dynamics: Jumanji's source is not available, see the module docstring):
hand-worked cases for every rule the restatement lists."""
import numpy as np

from oracle import lbf, prng


def _state(spec, agent_pos, agent_level, food_pos, food_level, eaten=None, step=0):
    A, NF = spec.num_agents, spec.num_food
    core = dict(agent_pos=np.array([agent_pos], np.int32), agent_level=np.array([agent_level], np.int32), agent_loading=np.zeros((1, A), bool),
                food_pos=np.array([food_pos], np.int32), food_level=np.array([food_level], np.int32),
                food_eaten=np.array([eaten if eaten is not None else [False] * NF], bool), step_count=np.array([step], np.int32),
                key=np.array([[1, 2]], np.uint32))
    return dict(core, metrics_key=np.zeros((1, 2), np.uint32), running_return=np.zeros(1, np.float32), running_length=np.zeros(1, np.int32),
                episode_return=np.zeros(1, np.float32), episode_length=np.zeros(1, np.int32))


def test_generator_respects_the_placement_rules():
    spec = lbf.LbfSpec(8, 8, 2, 2, 2, True, 100)
    st, ts = lbf.reset(spec, prng.split(prng.prng_key(0), 200))
    fp, ap = st["food_pos"], st["agent_pos"]
    assert ((fp >= 1) & (fp <= 6)).all(), "food never on the border"
    assert (np.abs(fp[:, 0] - fp[:, 1]).sum(1) > 1).all(), "foods never in the same or in 4-adjacent cells"
    for a in range(2):
        for f in range(2):
            assert (ap[:, a] != fp[:, f]).any(1).all(), "agents never on food"
    assert (ap[:, 0] != ap[:, 1]).any(1).all()
    assert set(np.unique(st["agent_level"])) <= {1, 2} and len(np.unique(st["agent_level"])) == 2
    assert (st["food_level"] == st["agent_level"].sum(1, keepdims=True)).all(), "force_coop: food level = sum of the agent levels"
    assert ts["observation"]["agents_view"].shape == (200, 2, 14) and (ts["step_type"] == lbf.STEP_FIRST).all()
    assert len({tuple(r) for r in fp.reshape(200, -1)}) > 50, "placements vary with the key"


def test_moves_blocking_and_collisions():
    spec = lbf.LbfSpec(8, 8, 3, 1, 2, True, 100)
    # agent 0 walks into food (blocked), agent 1 walks off the grid (blocked), agent 2 moves freely
    st = _state(spec, [[2, 3], [0, 0], [5, 5]], [1, 1, 2], [[3, 3]], [4])
    st2, ts = lbf.step(spec, st, np.array([[2, 1, 4]]), auto_reset=False)   # DOWN, UP, RIGHT
    assert st2["agent_pos"][0].tolist() == [[2, 3], [0, 0], [5, 6]]
    # two agents target the same cell: both return; a third one moving into a vacated cell is fine
    st = _state(spec, [[2, 2], [2, 4], [6, 6]], [1, 1, 1], [[5, 1]], [3])
    st2, _ = lbf.step(spec, st, np.array([[4, 3, 0]]), auto_reset=False)    # RIGHT, LEFT -> both want (2, 3)
    assert st2["agent_pos"][0].tolist() == [[2, 2], [2, 4], [6, 6]]
    # moving into the cell another agent occupies BEFORE the move is blocked even if that agent moves away
    st = _state(spec, [[2, 2], [2, 3], [6, 6]], [1, 1, 1], [[5, 1]], [3])
    st2, _ = lbf.step(spec, st, np.array([[4, 4, 0]]), auto_reset=False)
    assert st2["agent_pos"][0].tolist() == [[2, 2], [2, 4], [6, 6]]


def test_loading_reward_and_termination():
    spec = lbf.LbfSpec(8, 8, 2, 2, 2, True, 100)
    # food 0 (level 3) at (3, 3), agents of level 1 and 2 adjacent; food 1 (level 3) far away
    st = _state(spec, [[2, 3], [3, 4]], [1, 2], [[3, 3], [6, 6]], [3, 3])
    _, ts = lbf.step(spec, st, np.array([[5, 0]]), auto_reset=False)        # only agent 0 loads: 1 < 3, nothing happens
    assert ts["reward"][0].tolist() == [0.0, 0.0] and ts["step_type"][0] == lbf.STEP_MID
    st2, ts = lbf.step(spec, st, np.array([[5, 5]]), auto_reset=False)      # both load: 3 >= 3 -> eaten
    # agent rewards 1*3/(3*6) and 2*3/(3*6), team reward 0.5 for each agent (LbfWrapper aggregates)
    assert np.allclose(ts["reward"][0], [0.5, 0.5]) and st2["food_eaten"][0].tolist() == [True, False]
    assert ts["observation"]["agents_view"][0, 0, 2:5].tolist() == [-1.0, -1.0, 0.0], "eaten food reads (-1, -1, 0)"
    # second food: termination with discount 0 and the episode return 1.0
    st3 = _state(spec, [[5, 6], [6, 5]], [1, 2], [[3, 3], [6, 6]], [3, 3], eaten=[True, False], step=7)
    st3["running_return"][:] = 0.5; st3["running_length"][:] = 7
    st4, ts = lbf.step(spec, st3, np.array([[5, 5]]), auto_reset=False)
    assert ts["step_type"][0] == lbf.STEP_LAST and ts["discount"][0].tolist() == [0.0, 0.0]
    assert ts["episode_metrics"]["episode_return"][0] == 1.0 and ts["episode_metrics"]["episode_length"][0] == 8
    # truncation at the time limit keeps discount 1
    spec2 = lbf.LbfSpec(8, 8, 2, 2, 2, True, 3)
    st5 = _state(spec2, [[0, 0], [7, 7]], [1, 2], [[3, 3], [5, 5]], [3, 3], step=2)
    _, ts = lbf.step(spec2, st5, np.array([[0, 0]]), auto_reset=False)
    assert ts["step_type"][0] == lbf.STEP_LAST and ts["discount"][0].tolist() == [1.0, 1.0]


def test_observation_layout_mask_and_fov():
    spec = lbf.LbfSpec(8, 8, 2, 2, 2, True, 100)
    st = _state(spec, [[2, 3], [0, 0]], [1, 2], [[3, 3], [6, 6]], [3, 3])
    ob = lbf.make_obs(spec, st)
    # [one-hot id | food0 (x, y, l) | food1 | self | other]
    assert ob["agents_view"][0, 0].tolist() == [1, 0, 3, 3, 3, 6, 6, 3, 2, 3, 1, 0, 0, 2]
    assert ob["agents_view"][0, 1].tolist() == [0, 1, 3, 3, 3, 6, 6, 3, 0, 0, 2, 2, 3, 1]
    # agent 0 at (2, 3): DOWN hits the food, everything else is legal, LOAD legal (food adjacent)
    assert ob["action_mask"][0, 0].tolist() == [True, True, False, True, True, True]
    # agent 1 in the corner: UP and LEFT leave the grid, no food adjacent -> no LOAD
    assert ob["action_mask"][0, 1].tolist() == [True, False, True, False, True, False]
    # fov 2: coordinates relative to the clipped window, far entities hidden
    spec2 = lbf.LbfSpec(8, 2, 2, 2, 2, True, 100)
    st = _state(spec2, [[4, 4], [0, 0]], [1, 2], [[3, 3], [7, 7]], [3, 3])
    ob = lbf.make_obs(spec2, st)
    assert ob["agents_view"][0, 0, 2:].tolist() == [1, 1, 3, -1, -1, 0, 2, 2, 1, -1, -1, 0]
    assert ob["agents_view"][0, 1, 2:].tolist() == [-1, -1, 0, -1, -1, 0, 0, 0, 2, -1, -1, 0]


def test_auto_reset_draws_a_new_level_and_keeps_the_terminal_reward():
    spec = lbf.LbfSpec(8, 8, 2, 2, 2, True, 100)
    st = _state(spec, [[5, 6], [6, 5]], [1, 2], [[3, 3], [6, 6]], [3, 3], eaten=[True, False], step=7)
    st2, ts = lbf.step(spec, st, np.array([[5, 5]]), auto_reset=True)
    assert ts["step_type"][0] == lbf.STEP_LAST and np.allclose(ts["reward"][0], [0.5, 0.5])
    assert st2["step_count"][0] == 0 and not st2["food_eaten"][0].any()
    assert ts["observation"]["step_count"][0].tolist() == [0, 0]
    want = lbf._core_reset(spec, prng.split(np.array([[1, 2]], np.uint32), 2)[:, 0, :])   # key, _ = split(state.key)
    assert np.array_equal(st2["food_pos"], want["food_pos"]) and np.array_equal(st2["agent_pos"], want["agent_pos"])
    assert np.array_equal(st2["key"], want["key"])
