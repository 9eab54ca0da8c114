"""Robot Warehouse restatement (oracle/rware.py; UNPINNED dynamics: Jumanji's source is not available, see the module docstring):
hand-worked cases for the rules the restatement lists."""
import numpy as np

from oracle import prng, rware


def _one(spec, agent_pos, agent_dir, carry=None, requested=(), step=0):
    A, NS = spec.num_agents, spec.num_shelves
    ga = np.zeros((spec.H, spec.W), np.int32)
    for a, (r, c) in enumerate(agent_pos):
        ga[r, c] = a + 1
    gs = np.zeros((spec.H, spec.W), np.int32)
    gs[spec.shelf_cells[:, 0], spec.shelf_cells[:, 1]] = np.arange(1, NS + 1)
    req = np.zeros(NS, bool)
    req[list(requested)] = True
    st = dict(grid_a=ga, grid_s=gs, agent_pos=np.array(agent_pos, np.int32), agent_dir=np.array(agent_dir, np.int32),
              agent_carry=np.array(carry if carry is not None else [False] * A, bool), shelf_req=req,
              queue=np.array([i + 1 for i in requested], np.int32), step_count=np.int32(step), key=np.array([3, 4], np.uint32))
    st["action_mask"] = rware._action_mask(spec, st)
    return st


def test_layout_of_the_tiny_warehouse():
    spec = rware.RwareSpec(8, 1, 3, 4, 1, 4, 500)
    assert (spec.H, spec.W, spec.num_shelves, spec.obs_dim) == (11, 10, 32, 75)
    assert spec.goals == [(10, 4), (10, 5)]
    # shelves: rows 1..8 of columns 1, 2, 7, 8 (the centre block is the goal corridor), ids row-major
    assert spec.shelf_cells[:5].tolist() == [[1, 1], [1, 2], [1, 7], [1, 8], [2, 1]]
    assert spec.highway[:, [0, 3, 4, 5, 6, 9]].all() and spec.highway[[0, 9, 10]].all() and not spec.highway[1:9][:, [1, 2, 7, 8]].any()
    assert rware.RwareSpec(8, 2, 3, 4, 1, 4).num_shelves == 80   # small: 20 x 10 grid, two shelf rows minus the corridor of the lower one


def test_generator_and_auto_reset():
    spec = rware.RwareSpec(8, 1, 3, 4, 1, 4, 500)
    st, ts = rware.reset(spec, prng.split(prng.prng_key(2), 50))
    pos = st["agent_pos"]
    assert len({tuple(p) for n in range(50) for p in [tuple(map(tuple, pos[n]))]}) > 25
    for n in range(50):
        assert len({tuple(p) for p in pos[n]}) == 4, "agents start on distinct cells"
        assert len(set(st["queue"][n])) == 4 and st["shelf_req"][n].sum() == 4
        assert sorted(np.nonzero(st["shelf_req"][n])[0] + 1) == sorted(st["queue"][n])
    assert set(np.unique(st["agent_dir"])) == {0, 1, 2, 3}
    assert ts["observation"]["agents_view"].shape == (50, 4, 75)


def test_moves_turns_collisions_and_mask():
    spec = rware.RwareSpec(8, 1, 3, 2, 1, 2, 500)
    # agent 0 at the top-left corner facing up: FORWARD is masked (the clipped cell ahead is its own), so it becomes a NOOP
    st = _one(spec, [(0, 0), (5, 3)], [0, 1])
    assert st["action_mask"].tolist() == [[True, False, True, True, True], [True, True, True, True, True]]
    st2, r, done = rware._step_one(spec, st, np.array([1, 1]))
    assert st2["agent_pos"].tolist() == [[0, 0], [5, 4]] and not done and r == 0
    # turns: LEFT = dir - 1, RIGHT = dir + 1
    st3, _, _ = rware._step_one(spec, st2, np.array([2, 3]))
    assert st3["agent_dir"].tolist() == [3, 2]
    # two agents entering the same free cell in one step collide: the episode ends
    st = _one(spec, [(5, 3), (5, 5)], [1, 3])
    _, _, done = rware._step_one(spec, st, np.array([1, 1]))
    assert done
    # an agent next to another one facing it: FORWARD masked
    st = _one(spec, [(5, 3), (5, 4)], [1, 0])
    assert not st["action_mask"][0, 1] and st["action_mask"][1, 1]


def test_loading_carrying_and_delivery():
    spec = rware.RwareSpec(8, 1, 3, 2, 1, 2, 500)
    shelf = 1                                   # shelf id 1 stands at (1, 1)
    st = _one(spec, [(1, 1), (9, 9)], [2, 0], requested=(0, 5))
    st, _, _ = rware._step_one(spec, st, np.array([4, 0]))       # TOGGLE_LOAD on a shelf cell: pick it up
    assert st["agent_carry"].tolist() == [True, False]
    # carrying: moving onto another shelf (2, 1) is masked; turn right (-> left ... here dir 2 down -> 3 left) and walk to the highway
    assert not st["action_mask"][0, 1]
    st, _, _ = rware._step_one(spec, st, np.array([3, 0]))
    st, _, _ = rware._step_one(spec, st, np.array([1, 0]))
    assert st["agent_pos"][0].tolist() == [1, 0] and st["grid_s"][1, 0] == shelf and st["grid_s"][1, 1] == 0
    st, _, _ = rware._step_one(spec, st, np.array([4, 0]))       # cannot put a shelf down on a highway
    assert st["agent_carry"][0]
    # teleport the carried, requested shelf next to the goal and step onto it: +1 reward, a new request replaces it
    st = _one(spec, [(10, 3), (0, 9)], [1, 0], carry=[True, False], requested=(0, 5))
    st["grid_s"][1, 1] = 0
    st["grid_s"][10, 3] = shelf
    st["action_mask"] = rware._action_mask(spec, st)
    st2, r, done = rware._step_one(spec, st, np.array([1, 0]))
    assert r == 1.0 and not done and st2["grid_s"][10, 4] == shelf
    assert not st2["shelf_req"][0] and st2["shelf_req"].sum() == 2 and shelf not in st2["queue"] and 6 in st2["queue"]
    assert not np.array_equal(st2["key"], st["key"]), "a delivery consumes a split of the state key"
    # standing on the goal with a shelf that is no longer requested gives nothing
    _, r, _ = rware._step_one(spec, st2, np.array([0, 0]))
    assert r == 0.0


def test_observation_layout():
    spec = rware.RwareSpec(8, 1, 3, 2, 1, 2, 500)
    st = _one(spec, [(1, 0), (2, 0)], [2, 1], requested=(0, 6))   # shelf 1 at (1, 1) requested; agent 1 right below agent 0
    o = rware._observe(spec, st)[0]
    assert o[:8].tolist() == [1, 0, 0, 0, 0, 1, 0, 1]             # row, col, not carrying, facing down, on a highway
    cells = o[8:].reshape(9, 7)
    assert cells[0].tolist() == [0] * 7 and cells[3].tolist() == [0] * 7      # window cells left of column 0: outside the grid
    assert cells[4].tolist() == [1, 0, 0, 1, 0, 0, 0]             # centre: itself (facing down), no shelf
    assert cells[5].tolist() == [0, 0, 0, 0, 0, 1, 1]             # (1, 1): shelf 1, requested
    assert cells[7].tolist() == [1, 0, 1, 0, 0, 0, 0]             # (2, 0): agent 1 facing right
    assert cells[8].tolist() == [0, 0, 0, 0, 0, 1, 0]             # (2, 1): shelf 5, not requested


def test_wrapped_step_reward_and_metrics():
    spec = rware.RwareSpec(8, 1, 3, 2, 1, 2, 3)
    st, ts = rware.reset(spec, prng.split(prng.prng_key(5), 3))
    for t in range(3):
        st, ts = rware.step(spec, st, np.zeros((3, 2), np.int32), auto_reset=True)
    assert (ts["step_type"] == rware.STEP_LAST).all() and (ts["discount"] == 0).all()       # horizon: termination
    assert (ts["episode_metrics"]["episode_length"] == 3).all() and (st["step_count"] == 0).all()
    assert ts["reward"].shape == (3, 2)
