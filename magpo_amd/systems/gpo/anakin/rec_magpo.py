"""MAGPO system entry point on MI355X -- drop-in for mava/systems/gpo/anakin/rec_magpo.py.

Same public names and call contract as the reference system file:
    hydra_entry_point / main(overrides)      rec_magpo.py:818-831
    run_experiment(config) -> float          rec_magpo.py:688-815
    learner_setup(env, keys, config) -> (learn, actor_network, init_learner_state)   rec_magpo.py:533-685
    get_learner_fn(env, apply_fns, update_fn, config) -> LearnerFn                   rec_magpo.py:91-530
The bodies drive the HIP kernels (magpo_amd.learner.MagpoLearner); there is no JAX, no XLA, no Triton.

    python -m magpo_amd.systems.gpo.anakin.rec_magpo env=coordsum env/scenario=8x15-100 arch.num_envs=64

Multi-GPU: launch one process per GPU with torch.distributed.run; ``n_devices`` = world size, each rank owns
``update_batch_size`` groups of ``arch.num_envs`` envs, gradients are averaged with one RCCL all-reduce.
"""
from __future__ import annotations

import copy
import sys
import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from magpo_amd import distributed as mdist
from magpo_amd.actor import GruActor
from magpo_amd.config import Config, compose
from magpo_amd.evaluator import get_eval_fn, get_num_eval_envs, make_rec_eval_act_fn
from magpo_amd.learner import MagpoLearner, SystemConfig, host_split, prng_key
from magpo_amd.optim import ClipAdam
from magpo_amd.sable import SableGuider
from magpo_amd.types import ExperimentOutput, GPOLearnerState, HiddenStates, OptStates, Params, SableHiddenStates
from magpo_amd.utils import make_env as environments
from magpo_amd.utils.checkpointing import Checkpointer, latest_valid_checkpoint, load_checkpoint, restore_learner_state
from magpo_amd.utils.config import check_total_timesteps
from magpo_amd.utils.logger import LogEvent, MavaLogger

LearnerState = GPOLearnerState


def _system_config(config) -> SystemConfig:
    s = config.system
    return SystemConfig(rollout_length=int(s.rollout_length), ppo_epochs=int(s.ppo_epochs), num_minibatches=int(s.num_minibatches),
                        gamma=float(s.gamma), gae_lambda=float(s.gae_lambda), clip_eps=float(s.clip_eps), ent_coef=float(s.ent_coef),
                        vf_coef=float(s.vf_coef), max_grad_norm=float(s.max_grad_norm), clip_gpo=float(s.clip_gpo),
                        alpha=float(s.alpha), actor_lr=float(s.actor_lr), decay_learning_rates=bool(s.get("decay_learning_rates", False)),
                        lr_num_updates=int(s.num_updates) if s.get("num_updates") else 1000, micro_batches=int(s.get("micro_batches", 1) or 1))


def _snapshot_state(learner: MagpoLearner) -> GPOLearnerState:
    """LearnerState of the learner as an independent COPY (rec_magpo.py:488-497): the state a caller holds stays readable
    and re-usable after later learn() calls (the harness evaluates the pre-interval parameters, rec_magpo.py:770, SURVEY
    B12; a checkpoint of it can be resumed).  Leaves carry a leading group axis (the reference's update-batch axis)."""
    gs = learner.groups
    params = Params({k: v.clone() for k, v in learner.guider.named.items()}, {k: v.clone() for k, v in learner.actor.named.items()})
    opt = OptStates(dict(count=learner.g_count, mu=learner.g_mu.clone(), nu=learner.g_nu.clone()),
                    dict(count=learner.a_count, mu=learner.a_mu.clone(), nu=learner.a_nu.clone()))
    # The state carries the reference's [embed_dim / n_head, embed_dim / n_head] head states (get_init_hstates.py:20-43).  On the
    # device a head state sits in a zero-padded 64 x 64 tile, and a narrow net (embed_dim < 64, params.WidthEmbedding) keeps logical
    # entry (i, j) at device rows m i (q / k live in the first copy) and columns m j .. m j + m - 1 (v is duplicated), m = 64 / embed_dim.
    # The one 128-wide head of embed_dim 128 / n_head 1 lives in four 64 x 64 tiles S[I][J] (tile 2 I + J).
    gd = learner.guider
    hw, m = gd.hs, max(1, 64 // gd.EL)

    def logical(t):   # [n_block, ntile, N, 64, 64] -> [n_block, n_head, N, hs_logical, hs_logical]
        if gd.blockwise:
            return torch.cat([torch.cat([t[:, 0], t[:, 1]], -1), torch.cat([t[:, 2], t[:, 3]], -1)], -2).unsqueeze(1)
        return t[..., :hw:m, :hw:m]
    hs = HiddenStates(SableHiddenStates(*[torch.stack([logical(g.sable_hs[i]) for g in gs]) for i in range(3)]),
                      torch.stack([g.policy_h[g.cur] for g in gs]))
    env_state = {f: torch.stack([getattr(g.env, f) for g in gs]) for f in gs[0].env.state_fields}
    timestep = dict(agents_view=torch.stack([g.traj["obs"][0] for g in gs]), step_count=torch.stack([g.traj["step_count"][0] for g in gs]))
    if gs[0].traj["mask"] is not None:
        timestep["action_mask"] = torch.stack([g.traj["mask"][0] for g in gs])
    dones = torch.stack([g.traj["done"][0] for g in gs])
    return GPOLearnerState(params, opt, gs[0].key.copy(), env_state, timestep, dones, hs)


def load_learner_state(learner: MagpoLearner, state: GPOLearnerState) -> None:
    """Inverse of ``_snapshot_state``: write every leaf of ``state`` into the learner's (static, graph-captured) buffers."""
    as_dict = lambda x: x if isinstance(x, dict) else x._asdict()
    params, opt, hst = as_dict(state.params), as_dict(state.opt_states), as_dict(state.hstates)
    learner.guider.load_named(params["guider_params"])
    learner.actor.load_named(params["actor_params"])
    g, a = opt["guider_opt_state"], opt["actor_opt_state"]
    learner.g_mu.copy_(g["mu"]); learner.g_nu.copy_(g["nu"]); learner.g_count = int(g["count"])
    learner.a_mu.copy_(a["mu"]); learner.a_nu.copy_(a["nu"]); learner.a_count = int(a["count"])
    sable = as_dict(hst["sable_hidden_state"])
    sable = (sable["encoder"], sable["decoder_self_retn"], sable["decoder_cross_retn"])
    if state.dones.shape[0] != len(learner.groups):
        raise ValueError(f"learner state holds {state.dones.shape[0]} env groups, the learner {len(learner.groups)}")
    for gi, grp in enumerate(learner.groups):
        for f in grp.env.state_fields:
            getattr(grp.env, f).copy_(state.env_state[f][gi])
        grp.traj["obs"][0].copy_(state.timestep["agents_view"][gi])
        if grp.traj["mask"] is not None:
            grp.traj["mask"][0].copy_(state.timestep["action_mask"][gi])
        grp.traj["step_count"][0].copy_(state.timestep["step_count"][gi])
        grp.traj["done"][0].copy_(state.dones[gi])
        gd = learner.guider
        hw, m = gd.hs, max(1, 64 // gd.EL)
        for i in range(3):
            grp.sable_hs[i].zero_()
            if gd.blockwise:   # [n_block, 1, N, 128, 128] -> tiles (I, J)
                full = sable[i][gi][:, 0]
                for ti in range(4):
                    grp.sable_hs[i][:, ti].copy_(full[..., 64 * (ti // 2):64 * (ti // 2) + 64, 64 * (ti % 2):64 * (ti % 2) + 64])
                continue
            for c in range(m):   # rows m i, every column copy (inverse of the collapse in _snapshot_state)
                grp.sable_hs[i][..., :hw:m, c:hw:m].copy_(sable[i][gi])
        grp.policy_h[grp.cur].copy_(hst["policy_hidden_state"][gi])
        grp.key = np.array(state.key, dtype=np.uint32).copy()


def _owner(fn, cls, method: str, what: str):
    """The object whose device buffers ``fn`` acts on: ``fn`` must be the bound method ``cls.<method>`` or a thin adaptor around it
    (``functools.partial(...).func`` / ``functools.wraps(...).__wrapped__`` chains are followed)."""
    f, seen = fn, 0
    while not hasattr(f, "__self__") and seen < 8:
        f = getattr(f, "__wrapped__", None) or getattr(f, "func", None)
        seen += 1
        if f is None:
            break
    obj = getattr(f, "__self__", None)
    names = {method} | ({"act_fused"} if method == "get_actions" else {"train_fwd", "seq_fwd"} if method == "apply" else set())
    if not isinstance(obj, cls) or getattr(f, "__name__", None) not in names:
        raise TypeError(
            f"get_learner_fn: {what} must be the bound method {cls.__name__}.{method} of a network / optimiser object (or a functools.wraps / "
            f"functools.partial adaptor around it), got {fn!r}.  Unlike the reference's pure functions of parameter pytrees, the HIP path keeps "
            "parameters, activations and optimiser moments in device buffers owned by these objects and pairs each forward with a "
            "hand-written backward, so a free function cannot stand in for them.")
    return obj


def get_learner_fn(env, apply_fns, update_fn, config):
    """Returns ``learn(learner_state) -> ExperimentOutput``: ``config.system.num_updates_per_eval`` update steps
    (rec_magpo.py:91-530).  Same contract as the reference:

        apply_fns = (sable_action_select_fn, sable_apply_fn, actor_apply_fn)     rec_magpo.py:99   (execution / training / training)
        update_fn = (sable_update_fn, actor_update_fn)                           rec_magpo.py:100  (the two optimisers' update functions)

    In the reference these are pure functions of parameter pytrees.  Here the parameters, activations and optimiser moments live in device
    buffers OWNED by objects (``SableGuider``, ``GruActor``, ``ClipAdam``), and a hand-written backward pairs every training forward, so
    the five callables must be methods of such objects: ``SableGuider.get_actions`` / ``SableGuider.apply`` / ``GruActor.apply`` and
    ``ClipAdam.update``, either the bound methods themselves or thin adaptors around them that expose the method as ``__wrapped__``
    (``functools.wraps``) or ``func`` (``functools.partial``).  The loop CALLS exactly the callables it is given (rollout ->
    ``sable_action_select_fn``, minibatch forward -> ``sable_apply_fn`` / ``actor_apply_fn``, optimiser step -> the update functions) and
    reaches the owners' buffers / backward passes through them; anything else raises a ``TypeError`` that says so (``_owner``).
    ``env``: the MarlEnv whose batched kernels the rollout steps.

    State in, state out: the learner's device buffers are a cache of the last state it produced.  When ``learner_state``
    is that state (the normal host loop, rec_magpo.py:754,792) nothing is copied; any other state (an older one, a restored
    checkpoint) is loaded into the buffers first, so ``learn`` is a function of its argument."""
    sable_action_select_fn, sable_apply_fn, actor_apply_fn = apply_fns
    sable_update_fn, actor_update_fn = update_fn
    guider = _owner(sable_apply_fn, SableGuider, "apply", "apply_fns[1] (sable_apply_fn)")
    actor = _owner(actor_apply_fn, GruActor, "apply", "apply_fns[2] (actor_apply_fn)")
    if _owner(sable_action_select_fn, SableGuider, "get_actions", "apply_fns[0] (sable_action_select_fn)") is not guider:
        raise ValueError("the execution and the training function must belong to one Sable network")
    g_opt = _owner(sable_update_fn, ClipAdam, "update", "update_fn[0] (sable_update_fn)")
    a_opt = _owner(actor_update_fn, ClipAdam, "update", "update_fn[1] (actor_update_fn)")
    rank, world = mdist.rank_world()
    U = int(config.system.update_batch_size)
    learner = MagpoLearner(env.cfg, int(config.arch.num_envs), g_opt.sys, guider.dev, num_groups=U, guider=guider, actor=actor, optims=(g_opt, a_opt),
                           apply_fns=tuple(apply_fns), update_fns=tuple(update_fn))
    grad_sync = mdist.make_grad_sync(world)   # the pmean over ("batch", "device") of rec_magpo.py:395-409: one all-reduce of the flat buffer

    def learner_fn(learner_state: GPOLearnerState) -> ExperimentOutput:
        if learner_state is not getattr(learner, "_live_state", None):
            load_learner_state(learner, learner_state)
        n_up = int(config.system.num_updates_per_eval)
        # linear_scedule reads config.system.num_updates when the learner is traced, i.e. at the first learn() call -- AFTER
        # check_total_timesteps has rewritten it on the same config object (mava/utils/training.py:37-43; rec_magpo.py:581 vs :717)
        learner.sys.lr_num_updates = int(config.system.num_updates)
        ep: Dict[str, List[np.ndarray]] = {"episode_return": [], "episode_length": [], "is_terminal_step": []}
        train = []
        for _ in range(n_up):
            losses = learner.update_step(grad_sync)
            train.append(losses)
            for k in ep:
                ep[k].append(torch.stack([g.metrics[k] for g in learner.groups]).cpu().numpy())
        tl = torch.stack(train).cpu().numpy()  # (updates, P, M, 9)
        names = ["total_loss", "value_loss", "actor_loss", "guider_loss", "kl_loss", "entropy"]
        train_metrics = {n: tl[..., i] for i, n in enumerate(names)}
        episode_metrics = {k: np.stack(v) for k, v in ep.items()}
        episode_metrics["is_terminal_step"] = episode_metrics["is_terminal_step"].astype(bool)
        learner._live_state = _snapshot_state(learner)
        return ExperimentOutput(learner._live_state, episode_metrics, train_metrics)

    learner_fn.learner = learner
    return learner_fn


def learner_setup(env, keys, config, device=None, rank: int = 0, world: int = 1):
    """Initialise learner_fn, networks, optimiser, environments and states (rec_magpo.py:533-685)."""
    key, actor_net_key, net_key = keys
    config.system.num_agents = env.num_agents
    nc, mc = config.network.net_config, config.network.memory_config
    # memory_config.timestep_chunk_size only changes HOW the reference evaluates the chunkwise retention (smaller chunks with a
    # carried state, rec_magpo.py:552-557); the function is the same for every chunk size (recurrent == chunkwise, tested in
    # tests/test_oracle_networks.py).  The HIP kernel always walks 64-token tiles with the state on chip, so the key is honoured
    # as a pure memory/speed knob with no effect here.
    if mc.timestep_chunk_size:
        mc.chunk_size = int(mc.timestep_chunk_size) * env.num_agents
    else:
        mc.chunk_size = config.system.rollout_length * env.num_agents
    if mc.type != "rec_sable":
        raise NotImplementedError("memory_config.type must be rec_sable")
    if int(nc.embed_dim) not in (16, 32, 64, 128) or int(nc.n_head) not in (1, 2, 4) or int(config.network.hidden_state_dim) != 128:
        raise NotImplementedError("HIP kernels support embed_dim in {16,32,64,128}, n_head in {1,2,4}, hidden_state_dim=128 (any n_block)")
    device = device or torch.device("cuda", torch.cuda.current_device())
    U = int(config.system.update_batch_size)
    # networks (rec_magpo.py:559-579), optimisers (:581-589) -- objects that own their kernels' device buffers
    cfg, sysc = env.cfg, _system_config(config)
    # parameters = what flax creates from net_key / actor_net_key (rec_magpo.py:598-604,623; magpo_amd/params.py, UNPINNED restatement)
    from magpo_amd.learner import obs_row_stride
    obs_ld = obs_row_stride(cfg.obs_dim)   # floats between the rows the env kernels write; env.obs_dim = the features the networks read (add_agent_id)
    import os
    g_seed, a_seed = np.asarray(net_key, np.uint32), np.asarray(actor_net_key, np.uint32)
    if os.environ.get("MAGPO_LEGACY_INIT") == "1":   # A/B only: the torch-generator initialisation of rounds 1-3 (same distributions, other draws)
        g_seed = int(net_key[1]) & 0x7FFFFFFF
        a_seed = g_seed + 1
    sable_network = SableGuider(cfg.num_agents, cfg.num_actions, env.obs_dim, device, obs_ld=obs_ld, embed_dim=int(nc.embed_dim), n_head=int(nc.n_head),
                                n_block=int(nc.n_block), decay_scaling_factor=float(mc.decay_scaling_factor),
                                use_pe=bool(mc.timestep_positional_encoding), max_pos=cfg.time_limit + 1, seed=g_seed)
    actor_network = GruActor(cfg.num_agents, cfg.num_actions, env.obs_dim, device, obs_ld=obs_ld, seed=a_seed, tuning=sable_network.tuning)
    guider_optim, actor_optim = ClipAdam(sable_network, sysc), ClipAdam(actor_network, sysc)
    # Pack apply and update functions (rec_magpo.py:624-632)
    apply_fns = (sable_network.get_actions, sable_network.apply, actor_network.apply)
    update_fns = (guider_optim.update, actor_optim.update)
    learn = get_learner_fn(env, apply_fns, update_fns, config)
    learner = learn.learner
    learner.setup(key, n_groups=world * U, group=rank * U)
    learner._live_state = _snapshot_state(learner)
    return learn, learner.actor, learner._live_state


def run_experiment(_config) -> float:
    """Runs experiment (rec_magpo.py:688-815)."""
    _config.logger.system_name = "rec_magpo"
    config = copy.deepcopy(_config)
    rank, world, local = mdist.init_from_env()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    n_devices = world

    env, eval_env = environments.make(config)
    ks = host_split(prng_key(int(config.system.seed)), 4)
    key, key_e, actor_net_key, net_key = ks[0], ks[1], ks[2], ks[3]
    learn, actor_network, learner_state = learner_setup(env, (key, actor_net_key, net_key), config, device, rank, world)

    from magpo_amd.learner import obs_row_stride
    eval_actor = GruActor(env.num_agents, env.action_dim, env.obs_dim, device, obs_ld=obs_row_stride(env.cfg.obs_dim))
    eval_act_fn = make_rec_eval_act_fn(eval_actor, config)
    evaluator = get_eval_fn(eval_env, eval_act_fn, config, absolute_metric=False, device=device, n_devices=n_devices)

    config = check_total_timesteps(config, n_devices)
    assert config.system.num_updates > config.arch.num_evaluation, \
        "Number of updates per evaluation must be less than total number of updates."
    config.system.num_updates_per_eval = config.system.num_updates // config.arch.num_evaluation
    steps_per_rollout = (n_devices * config.system.num_updates_per_eval * config.system.rollout_length
                         * config.system.update_batch_size * config.arch.num_envs)
    logger = MavaLogger(config) if rank == 0 else None
    # every rank saves: rank 0 the full state, the others their own rollout state (their envs, keys and hidden states differ)
    save_checkpoint = bool(config.logger.checkpointing.save_model)
    if save_checkpoint:
        sa = config.logger.checkpointing.save_args.to_container()
        if world > 1 and not sa.get("checkpoint_uid"):   # one directory for all ranks
            sa["checkpoint_uid"] = mdist.broadcast_object(time.strftime("%Y%m%d%H%M%S"))
        checkpointer = Checkpointer(metadata=config.to_container(), model_name=config.logger.system_name,
                                    base_path=config.logger.base_exp_path, rank=rank, world=world, **sa)
    if bool(config.logger.checkpointing.load_model):
        # Resume from the latest loadable checkpoint of load_args.checkpoint_uid (the reference saves the full learner state,
        # checkpointing.py:108-145, but rec_magpo.py never reads it back: this closes the loop for long sweeps).  Rank-aware:
        # parameters / optimiser state from rank 0's file, env state / keys / hidden states from the rank's own file.
        import os
        la = config.logger.checkpointing.load_args
        cdir = os.path.join(config.logger.base_exp_path, la.rel_dir, config.logger.system_name, str(la.checkpoint_uid))
        latest = latest_valid_checkpoint(cdir, rank, world)
        learner_state, _ = restore_learner_state(latest, device, rank, world)
        resume = load_checkpoint(latest).get("extras") or {}
    else:
        resume = {}
    eval_batch = get_num_eval_envs(config, absolute_metric=False, n_devices=n_devices)
    eval_hs = {"hidden_state": torch.zeros(eval_batch * env.num_agents, 128, device=device)}

    max_episode_return = -np.inf
    best_params = None
    eval_metrics: Dict[str, Any] = {}
    start_eval = 0
    if resume:   # a checkpoint written by this loop: continue the evaluation counter, the evaluator's key chain and the best-params record
        start_eval = int(resume["eval_step"]) + 1
        key_e = np.asarray(resume["key_e"], np.uint32)
        max_episode_return = float(resume["max_episode_return"])
        best_params = None if resume["best_params"] is None else {k: v.to(device) for k, v in resume["best_params"].items()}
    for eval_step in range(start_eval, int(config.arch.num_evaluation)):
        start = time.time()
        learner_output = learn(learner_state)
        torch.cuda.synchronize()
        elapsed = time.time() - start
        t = int(steps_per_rollout * (eval_step + 1))
        em = learner_output.episode_metrics
        term = em["is_terminal_step"]
        ep_completed = bool(term.any())
        if logger:
            logger.log({"timestep": t}, t, eval_step, LogEvent.MISC)
            if ep_completed:
                logger.log({"episode_return": em["episode_return"][term], "episode_length": em["episode_length"][term],
                            "steps_per_second": steps_per_rollout / elapsed}, t, eval_step, LogEvent.ACT)
            logger.log(learner_output.train_metrics, t, eval_step, LogEvent.TRAIN)
        # evaluate the PRE-interval actor parameters, as the reference does (rec_magpo.py:770)
        trained_params = learner_state.params.actor_params
        ks = host_split(key_e, n_devices + 1)
        key_e, eval_key = ks[0], ks[1 + rank]
        eval_metrics = evaluator(trained_params, eval_key, eval_hs)
        if logger:
            logger.log(eval_metrics, t, eval_step, LogEvent.EVAL)
        episode_return = float(np.mean(eval_metrics["episode_return"]))
        if config.arch.absolute_metric and max_episode_return <= episode_return:
            best_params = {k: v.clone() for k, v in trained_params.items()}
            max_episode_return = episode_return
        if save_checkpoint:  # rec_magpo.py:779-785 (+ what run_experiment itself needs to continue: its loop state)
            mdist.barrier()
            checkpointer.save(timestep=t, unreplicated_learner_state=learner_output.learner_state, episode_return=episode_return,
                              extras=dict(eval_step=eval_step, key_e=key_e.copy(), max_episode_return=max_episode_return,
                                          best_params=None if best_params is None else {k: v.cpu() for k, v in best_params.items()}))
            mdist.barrier()      # every rank's file of this timestep is on disk: only now may older checkpoints go
            checkpointer.prune()
        learner_state = learner_output.learner_state

    eval_performance = float(np.mean(eval_metrics[config.env.eval_metric])) if eval_metrics else float("nan")
    if config.arch.absolute_metric:
        eb = get_num_eval_envs(config, absolute_metric=True, n_devices=n_devices)
        abs_hs = {"hidden_state": torch.zeros(eb * env.num_agents, 128, device=device)}
        abs_eval = get_eval_fn(eval_env, eval_act_fn, config, absolute_metric=True, device=device, n_devices=n_devices)
        abs_key = host_split(key, n_devices)[rank]
        m = abs_eval(best_params, abs_key, abs_hs)
        if logger:
            logger.log(m, int(steps_per_rollout * config.arch.num_evaluation), int(config.arch.num_evaluation) - 1, LogEvent.ABSOLUTE)
    if logger:
        logger.stop()
    return eval_performance


def hydra_entry_point(overrides: Optional[List[str]] = None) -> float:
    """Experiment entry point (rec_magpo.py:818-831): compose configs/default/rec_magpo.yaml + CLI overrides."""
    cfg = compose("rec_magpo", sys.argv[1:] if overrides is None else overrides)
    perf = run_experiment(cfg)
    print("MAGPO experiment completed")
    return perf


if __name__ == "__main__":
    hydra_entry_point()
