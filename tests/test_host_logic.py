"""Host-side logic that needs no GPU: parameter layout, config composer, env factory, logger, grad sync (gloo, world 2)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from magpo_amd import params as P
from magpo_amd.config import compose
from oracle import networks as onets


def test_flat_layout_matches_reference_parameter_tree():
    E, F, K = 64, 5, 20
    fp = P.FlatParams(P.guider_layout(E, F, K), "cpu")
    named = P.guider_named_views(fp.views())
    ref = onets.guider_param_shapes(E, F, K)
    assert set(named) == set(ref)
    for n, shp in ref.items():
        assert tuple(named[n].shape) == tuple(shp), n
    assert sum(int(np.prod(s)) for s in ref.values()) == 98330
    for off in fp.offsets.values():
        assert off % 4 == 0
    named["enc.block0.retn.w_k"][0, 3, 7] = 5.0  # views alias the flat buffer
    assert fp.flat[fp.offsets["enc.block0.retn.w_qkvg"] + 3 * 256 + 64 + 7] == 5.0
    fa = P.FlatParams(P.actor_layout(F, 128, K), "cpu")
    an = P.actor_named_views(fa.views())
    aref = onets.actor_param_shapes(F, 128, K)
    assert set(an) == set(aref) and all(tuple(an[n].shape) == tuple(s) for n, s in aref.items())
    assert sum(int(np.prod(s)) for s in aref.values()) == 118676


def test_init_distributions():
    fp = P.FlatParams(P.guider_layout(64, 5, 20), "cpu")
    named = P.guider_named_views(fp.views())
    P.init_guider(named, 0)
    assert float(named["enc.block0.ffn.W_gate"].abs().max()) == 0.0 and float(named["enc.ln.scale"].min()) == 1.0
    w = named["enc.head.dense0.kernel"]
    assert torch.allclose(w.T @ w, 2.0 * torch.eye(64), atol=1e-4)       # orthogonal(sqrt 2)
    assert abs(float(named["dec.block0.retn1.w_g"].std()) - 1 / 64) < 2e-3  # normal(1/E)
    assert float(named["dec.head.dense1.kernel"].abs().max()) < 0.011


def test_config_compose_and_overrides():
    c = compose("rec_magpo", ["env=coordsum", "env/scenario=8x15-100", "arch.num_envs=64", "+env.kwargs.num_agents=4",
                              "system.total_timesteps=~", "system.ppo_epochs=15"])
    assert c.env.scenario.task_name == "8x15-100-v0" and c.arch.num_envs == 64 and c.system.ppo_epochs == 15
    assert c.env.kwargs.to_container() == {"time_limit": 100, "num_agents": 4}
    assert c.system.total_timesteps is None and c.system.clip_gpo == 1.5 and c.system.update_batch_size == 2
    assert c.logger.loggers.json.task_name == "8x15-100-v0"  # ${env.scenario.task_name}
    assert c.network.memory_config.timestep_chunk_size is None and c.network.net_config.embed_dim == 64
    c.system.num_agents = 8  # struct mode off (rec_magpo.py:826)
    assert c.system.num_agents == 8
    d = compose("rec_magpo")
    assert d.env.env_name == "RobotWarehouse"  # reference default env is rware (configs/default/rec_magpo.yaml)
    with pytest.raises(KeyError):
        compose("rec_magpo", ["system.not_a_key=1"])
    # the one system key that is not the reference's: a minibatch in slabs with accumulated gradients (default: one pass)
    from magpo_amd.systems.gpo.anakin.rec_magpo import _system_config
    assert _system_config(c).micro_batches == 1
    assert _system_config(compose("rec_magpo", ["env=coordsum", "system.micro_batches=4"])).micro_batches == 4


def test_env_factory():
    from magpo_amd.utils import make_env
    c = compose("rec_magpo", ["env=coordsum", "env/scenario=5x20-80"])
    tr, ev = make_env.make(c)
    assert (tr.num_agents, tr.action_dim, tr.time_limit, tr.cfg.maxval, tr.obs_dim) == (5, 20, 100, 80, 6)
    assert tr.auto_reset and not ev.auto_reset
    c = compose("rec_magpo", ["env=coordsum", "+env.kwargs.num_agents=4", "+env.kwargs.num_actions=20", "+env.kwargs.maxval=60"])
    tr, _ = make_env.make(c)
    assert (tr.num_agents, tr.action_dim, tr.cfg.maxval) == (4, 20, 60)
    # the default env of configs/default/rec_magpo.yaml is rware with scenario tiny-2ag (mava/configs/env/rware.yaml:4): 71 vector
    # features + 2 agent-id features
    tr, ev = make_env.make(compose("rec_magpo"))
    assert (tr.num_agents, tr.action_dim, tr.time_limit, tr.obs_dim) == (2, 5, 500, 73) and tr.cfg.has_mask
    tr4, _ = make_env.make(compose("rec_magpo", ["env/scenario=tiny-4ag"]))
    assert (tr4.num_agents, tr4.obs_dim) == (4, 75)
    # the MarlEnv surface (mava/types.py:45-123): attributes + specs without touching the GPU
    assert tr.observation_spec.agents_view.shape == (2, 73) and tr.action_spec.shape == (2,) and int(tr.action_spec.num_values[0]) == 5
    assert callable(tr.reset) and callable(tr.step) and tr.unwrapped is tr.cfg
    with pytest.raises(ValueError):   # a 4x4 grid cannot be guaranteed to hold 3 non-adjacent interior food items
        make_env.make(compose("rec_magpo", ["env=lbf", "env.scenario.task_config.grid_size=4", "env.scenario.task_config.num_food=3"]))
    tr, _ = make_env.make(compose("rec_magpo", ["env=lbf", "env/scenario=15x15-4p-5f"]))
    assert (tr.num_agents, tr.action_dim, tr.time_limit, tr.obs_dim, tr.cfg.fov) == (4, 6, 100, 31, 15)
    with pytest.raises(NotImplementedError):
        make_env.make(compose("rec_magpo", ["env=rware", "env.scenario.task_config.sensor_range=2"]))


def test_check_total_timesteps_and_logger(tmp_path, capsys):
    from magpo_amd.utils.config import check_total_timesteps
    from magpo_amd.utils.logger import LogEvent, MavaLogger
    c = compose("rec_magpo", ["env=coordsum", "logger.loggers.json.enabled=True", f"logger.base_exp_path={tmp_path}/", "logger.loggers.json.path=x"])
    c = check_total_timesteps(c, n_devices=1)
    assert c.system.num_updates == 20_000_000 // 128 // 2 // 16 == 4882
    c.logger.system_name = "rec_magpo"
    lg = MavaLogger(c)
    lg.log({"episode_return": np.array([1.0, 3.0]), "episode_length": np.array([100, 100]), "steps_per_second": 5.0}, 4096, 0, LogEvent.EVAL)
    lg.log({"total_loss": np.ones((2, 2))}, 4096, 0, LogEvent.TRAIN)
    run = json.load(open(os.path.join(tmp_path, "json", "x", "metrics.json")))["CoordSum"]["3x10-30-v0"]["rec_magpo"]["seed_42"]
    assert run["step_0"]["mean_episode_return"] == [2.0] and run["step_0"]["step_count"] == 4096
    assert "EVALUATOR" in capsys.readouterr().out


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from magpo_amd import distributed as mdist
    r, w, _ = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)

    class Fake:  # the only attribute grad_sync touches
        grad_all = torch.full((1000,), float(rank + 1))

    sync = mdist.make_grad_sync(w)
    scale = sync(Fake)
    keys = torch.arange(2 * (world * 4 + 1)).view(-1, 2)
    mine = mdist.shard_env_keys(keys, 4, rank)
    q.put((rank, float(Fake.grad_all[0]) * scale, mine[0, 0].item(), mine.shape[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_world2_gloo():
    """N > 1 path on CPU: sum all-reduce of the flat gradient buffer, mean via the returned scale; env-key sharding."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 1.5           # mean of 1 and 2
    assert res[0][2] == 2 and res[1][2] == 10 and res[0][3] == 4  # rank r owns rows 1 + r*N ...


def test_make_grad_sync_single_group_is_none():
    from magpo_amd import distributed as mdist
    assert mdist.make_grad_sync(1) is None


def test_checkpointer_keeps_best(tmp_path):
    from magpo_amd.types import GPOLearnerState, OptStates, Params
    from magpo_amd.utils.checkpointing import Checkpointer
    ck = Checkpointer("rec_magpo", metadata={"a": 1}, base_path=str(tmp_path), max_to_keep=1, checkpoint_uid="u")
    mk = lambda v: GPOLearnerState(Params({"w": torch.full((3,), float(v))}, {"k": torch.zeros(2)}),
                                   OptStates(dict(count=1, mu=torch.zeros(3), nu=torch.zeros(3)), dict(count=1, mu=torch.zeros(2), nu=torch.zeros(2))),
                                   np.array([1, 2], np.uint32), [], [], torch.zeros(1), None)
    ck.save(100, mk(1), episode_return=5.0)
    ck.save(200, mk(2), episode_return=9.0)
    ck.save(300, mk(3), episode_return=7.0)
    files = sorted(f for f in os.listdir(os.path.join(tmp_path, "checkpoints", "rec_magpo", "u")) if f.endswith(".pt"))
    assert files == ["200.pt"]
    st = torch.load(os.path.join(tmp_path, "checkpoints", "rec_magpo", "u", "200.pt"), weights_only=True)
    assert st["episode_return"] == 9.0 and float(st["learner_state"]["params"]["guider_params"]["w"][0]) == 2.0
    assert json.load(open(os.path.join(tmp_path, "checkpoints", "rec_magpo", "u", "metadata.json")))["checkpointer_version"] == 2.0


def test_checkpoints_are_safe_atomic_rank_aware_and_resumable(tmp_path):
    """(1) files load with torch.load(weights_only=True): tensors + plain containers, numpy keys round-trip; (2) written through a
    temporary file (no ``.tmp`` left, a truncated newest file falls back to the one before it); (3) a relaunch seeds the retention from
    the files on disk; (4) ranks > 0 write / read their own rollout state, a rank-count mismatch raises."""
    from magpo_amd.types import GPOLearnerState, HiddenStates, OptStates, Params, SableHiddenStates
    from magpo_amd.utils.checkpointing import Checkpointer, latest_valid_checkpoint, load_checkpoint, restore_learner_state

    def mk(v, r=0):
        hs = HiddenStates(SableHiddenStates(*[torch.full((1, 1, 1, 2, 4, 4), float(v + r))] * 3), torch.full((1, 6, 128), float(r)))
        return GPOLearnerState(Params({"w": torch.full((3,), float(v))}, {"k": torch.zeros(2)}),
                               OptStates(dict(count=7, mu=torch.zeros(3), nu=torch.zeros(3)), dict(count=7, mu=torch.zeros(2), nu=torch.zeros(2))),
                               np.array([v, 100 + r], np.uint32), {"step_count": torch.full((1, 2), r, dtype=torch.int32)},
                               {"agents_view": torch.full((1, 2, 3, 4), float(r))}, torch.zeros(1, 2, dtype=torch.uint8), hs)
    d = os.path.join(tmp_path, "checkpoints", "rec_magpo", "u")
    cks = [Checkpointer("rec_magpo", base_path=str(tmp_path), max_to_keep=2, keep_latest=True, checkpoint_uid="u", rank=r, world=2) for r in range(2)]
    for t in (100, 200):
        for r in (1, 0):
            cks[r].save(t, mk(t, r), episode_return=float(t), extras=dict(eval_step=t // 100, key_e=np.array([5, 6], np.uint32), best_params=None))
    assert sorted(os.listdir(d)) == ["100.pt", "100.rank1.pt", "200.pt", "200.rank1.pt", "metadata.json"]
    raw = torch.load(os.path.join(d, "200.pt"), weights_only=True)          # (1) no pickled code objects
    assert raw["world"] == 2 and raw["learner_state"]["opt_states"]["guider_opt_state"]["count"] == 7
    s0, t0 = restore_learner_state(os.path.join(d, "200.pt"), "cpu", rank=0, world=2)
    s1, _ = restore_learner_state(os.path.join(d, "200.pt"), "cpu", rank=1, world=2)
    assert t0 == 200 and s0.key.dtype == np.uint32 and s0.key.tolist() == [200, 100] and s1.key.tolist() == [200, 101]
    assert torch.equal(s0.params.guider_params["w"], s1.params.guider_params["w"])                 # replicated parts from rank 0's file
    assert float(s1.env_state["step_count"][0, 0]) == 1 and float(s0.env_state["step_count"][0, 0]) == 0   # rollout state per rank
    assert float(s1.hstates.policy_hidden_state.max()) == 1.0 and float(s1.timestep["agents_view"].max()) == 1.0
    assert load_checkpoint(os.path.join(d, "200.pt"))["extras"]["key_e"].tolist() == [5, 6]
    with pytest.raises(ValueError):
        restore_learner_state(os.path.join(d, "200.pt"), "cpu", rank=0, world=4)
    # (2) a kill while writing 300.pt: the truncated file is skipped, and so is a timestep whose rank file is missing
    with open(os.path.join(d, "300.pt"), "wb") as f:
        f.write(open(os.path.join(d, "200.pt"), "rb").read()[:200])
    assert latest_valid_checkpoint(d, 0, 2).endswith("200.pt")
    os.remove(os.path.join(d, "200.rank1.pt"))
    assert latest_valid_checkpoint(d, 1, 2).endswith("100.pt")
    # (3) relaunch: the file that does not load is set aside (never deleted), the old files are ranked, retention removes a
    # timestep's files together -- and only once the newest timestep's file set is complete
    ck = Checkpointer("rec_magpo", base_path=str(tmp_path), max_to_keep=2, keep_latest=True, checkpoint_uid="u", rank=0, world=2)
    assert [k[1] for k in ck.kept] == [100, 200] and not os.path.exists(os.path.join(d, "300.pt")) and os.path.exists(os.path.join(d, "300.pt.corrupt"))
    ck.save(400, mk(400), episode_return=1.0)
    ck.prune()
    assert {"100.pt", "100.rank1.pt", "200.pt", "400.pt"} <= set(os.listdir(d)), "rank 1 has not written 400 yet: nothing may be pruned"
    Checkpointer("rec_magpo", base_path=str(tmp_path), checkpoint_uid="u", rank=1, world=2).save(400, mk(400, 1))
    ck.prune()
    assert sorted(os.listdir(d)) == ["200.pt", "300.pt.corrupt", "400.pt", "400.rank1.pt", "metadata.json"]
    assert not [f for f in os.listdir(d) if f.endswith(".tmp")]
    # (4b) the rank count is checked for a single process too
    with pytest.raises(ValueError):
        restore_learner_state(os.path.join(d, "400.pt"), "cpu", rank=0, world=1)
    with pytest.raises(ValueError):
        latest_valid_checkpoint(d, 0, 1)


def test_checkpointer_never_deletes_what_it_cannot_load_and_survives_a_kill_between_the_ranks_saves(tmp_path):
    """(1) a pre-existing checkpoint in another format (pickled numpy keys: not weights_only-loadable) survives the
    construction of a Checkpointer in its directory, stale ``*.tmp`` files of interrupted writes are what gets cleaned;
    (2) max_to_keep = 1, two ranks, the job is killed after rank 0 wrote timestep 200 and before rank 1 did: the complete
    checkpoint 100 must still be there and is what a resume finds; (3) orphan rank files of pruned timesteps are collected."""
    from magpo_amd.types import GPOLearnerState, HiddenStates, OptStates, Params, SableHiddenStates
    from magpo_amd.utils.checkpointing import Checkpointer, latest_valid_checkpoint

    def mk(v, r=0):
        hs = HiddenStates(SableHiddenStates(*[torch.full((1, 1, 1, 2, 4, 4), float(v + r))] * 3), torch.full((1, 6, 128), float(r)))
        return GPOLearnerState(Params({"w": torch.full((3,), float(v))}, {"k": torch.zeros(2)}),
                               OptStates(dict(count=7, mu=torch.zeros(3), nu=torch.zeros(3)), dict(count=7, mu=torch.zeros(2), nu=torch.zeros(2))),
                               np.array([v, 100 + r], np.uint32), {"step_count": torch.full((1, 2), r, dtype=torch.int32)},
                               {"agents_view": torch.full((1, 2, 3, 4), float(r))}, torch.zeros(1, 2, dtype=torch.uint8), hs)
    d = os.path.join(tmp_path, "checkpoints", "rec_magpo", "u")
    os.makedirs(d)
    torch.save({"learner_state": {"key": np.array([1, 2], np.uint32)}, "timestep": 50}, os.path.join(d, "50.pt"))   # r02 format
    open(os.path.join(d, "60.pt.tmp"), "wb").write(b"partial")
    open(os.path.join(d, "metadata.json.tmp"), "w").write("{")
    with pytest.warns(UserWarning, match="does not load"):
        cks = [Checkpointer("rec_magpo", base_path=str(tmp_path), max_to_keep=1, checkpoint_uid="u", rank=r, world=2) for r in range(2)][:2]
    assert os.path.exists(os.path.join(d, "50.pt.corrupt")) and not [f for f in os.listdir(d) if f.endswith(".tmp")]
    old = torch.load(os.path.join(d, "50.pt.corrupt"), weights_only=False)
    assert old["timestep"] == 50
    for r in (0, 1):
        cks[r].save(100, mk(100, r), episode_return=1.0)
    cks[0].prune()
    cks[0].save(200, mk(200, 0), episode_return=2.0)      # ... and the job dies before rank 1 writes 200.rank1.pt
    cks[0].prune()                                        # (even if rank 0 got this far)
    assert {"100.pt", "100.rank1.pt", "200.pt"} <= set(os.listdir(d))
    assert latest_valid_checkpoint(d, 1, 2).endswith("100.pt")
    # relaunch; this time both ranks get through timestep 300, and 300 (better return) replaces 100; the half-written 200 goes too
    cks = [Checkpointer("rec_magpo", base_path=str(tmp_path), max_to_keep=1, checkpoint_uid="u", rank=r, world=2) for r in range(2)]
    open(os.path.join(d, "10.rank1.pt"), "wb").write(b"orphan of a pruned timestep")
    for r in (1, 0):
        cks[r].save(300, mk(300, r), episode_return=3.0)
    cks[0].prune()
    assert sorted(f for f in os.listdir(d) if f.endswith(".pt")) == ["300.pt", "300.rank1.pt"]
    # a checkpoint that is NOT the best is itself the victim: its rank files go with it, nothing is orphaned
    for r in (0, 1):
        cks[r].save(400, mk(400, r), episode_return=0.5)
    cks[0].prune()
    assert sorted(f for f in os.listdir(d) if f.endswith(".pt")) == ["300.pt", "300.rank1.pt"]


def test_tuning_defaults_and_environment_switches():
    """Host-side tuning object (the C ABI keeps no state): the shipped defaults, and every arithmetic mode one switch away."""
    from magpo_amd.tuning import Tuning
    d = Tuning.from_env({})
    assert d == Tuning() and d.gru_split_bf16 == 2 and d.linear_variant == 0 and d.actor_linear_variant == 0 and d.wgrad_variant == 0
    f = Tuning.from_env({"MAGPO_GRU_SPLIT_BF16": "0"})
    assert f.gru_split_bf16 == 0 and f.linear_variant == 0 and f.actor_linear_variant == 0          # exact fp32 MFMA everywhere
    o = Tuning.from_env({"MAGPO_LINEAR_BF3": "1", "MAGPO_WGRAD_BF3": "1", "MAGPO_GRU_SPLIT_BF16": "1", "MAGPO_RET_CHUNK": "64"})
    assert o.linear_variant & 4 and o.actor_linear_variant & 4 and o.wgrad_variant & 64 and o.gru_split_bf16 == 1 and o.ret_chunk_tokens == 64
    e = Tuning.from_env({"MAGPO_LINEAR_BF3": "", "MAGPO_GRU_SPLIT_BF16": ""})                         # empty variables = unset
    assert e == Tuning()


def test_net_obs_column_offsets_for_add_agent_id_false():
    """system.add_agent_id (make_env.py:90-104): with the AgentIDWrapper the networks read the whole [one-hot id | features] row, without it the
    features behind the id (same rows, pointer offset); wide observations (Robot Warehouse) refuse the offset."""
    from magpo_amd.learner import CoordSumConfig, LbfConfig, RwareConfig, net_obs
    assert net_obs(CoordSumConfig(4, 20)) == (5, 0) and net_obs(CoordSumConfig(4, 20, add_agent_id=False)) == (1, 4)
    lbf = LbfConfig(8, 8, 2, 2, 2, True, 100)
    assert net_obs(lbf) == (14, 0) and net_obs(LbfConfig(8, 8, 2, 2, 2, True, 100, add_agent_id=False)) == (12, 2)
    rw = RwareConfig()
    assert net_obs(rw) == (rw.obs_dim, 0)
    rw.add_agent_id = False
    with pytest.raises(NotImplementedError):
        net_obs(rw)
