"""Checkpoint saving (mava/utils/checkpointing.py:34-145) without Orbax: ``torch.save`` of the LearnerState pytree.

Mirrors the reference's behaviour for rec_magpo: save-only (the system never restores, rec_magpo.py:733-739, 779-785),
best-by-``episode_return`` retention with ``max_to_keep``, ``save_interval_steps`` counted in evaluations, config stored as
metadata next to the state.  ``restore_learner_state`` is provided for resuming sweeps."""
from __future__ import annotations

import json
import os
import time
from typing import Any, Dict, List, Optional, Tuple

import torch

CHECKPOINTER_VERSION = 2.0


def _to_cpu(x: Any) -> Any:
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _to_cpu(v) for k, v in x.items()}
    if isinstance(x, tuple) and hasattr(x, "_fields"):
        return {f: _to_cpu(getattr(x, f)) for f in x._fields}
    if isinstance(x, (list, tuple)):
        return [_to_cpu(v) for v in x]
    return x


class Checkpointer:
    def __init__(self, model_name: str, metadata: Optional[Dict] = None, base_path: str = "results/", rel_dir: str = "checkpoints",
                 checkpoint_uid: Optional[str] = None, save_interval_steps: int = 1, max_to_keep: Optional[int] = 1,
                 keep_period: Optional[int] = None, keep_latest: bool = False):
        uid = checkpoint_uid or time.strftime("%Y%m%d%H%M%S")
        self.dir = os.path.join(base_path, rel_dir, model_name, uid)
        os.makedirs(self.dir, exist_ok=True)
        self.interval = max(1, int(save_interval_steps))
        self.max_to_keep = max_to_keep
        self.keep_period = keep_period
        self.keep_latest = bool(keep_latest)   # extension: rank by timestep instead of episode_return (resumable sweeps)
        self.kept: List[Tuple[float, int, str]] = []   # (episode_return, timestep, path)
        self.calls = 0
        with open(os.path.join(self.dir, "metadata.json"), "w") as f:
            json.dump({"checkpointer_version": CHECKPOINTER_VERSION, **(metadata or {})}, f, indent=1, default=str)

    def save(self, timestep: int, unreplicated_learner_state: Any, episode_return: float = 0.0, extras: Optional[Dict] = None) -> bool:
        self.calls += 1
        if (self.calls - 1) % self.interval:
            return False
        path = os.path.join(self.dir, f"{int(timestep)}.pt")
        torch.save({"learner_state": _to_cpu(unreplicated_learner_state), "timestep": int(timestep),
                    "episode_return": float(episode_return), "extras": extras}, path)
        self.kept.append((float(episode_return), int(timestep), path))
        if self.max_to_keep:
            # keep the best `max_to_keep` by episode_return (ties: latest), like orbax best_fn / best_mode="max"
            self.kept.sort(key=(lambda e: e[1]) if self.keep_latest else (lambda e: (e[0], e[1])))
            while len(self.kept) > int(self.max_to_keep):
                ret, ts, victim = self.kept[0]
                if self.keep_period and ts % int(self.keep_period) == 0:
                    break
                self.kept.pop(0)
                if os.path.exists(victim):
                    os.remove(victim)
        return True


def restore_learner_state(path: str, device="cuda"):
    """Load a checkpoint written by ``Checkpointer.save`` and rebuild the full GPOLearnerState on ``device``: parameters,
    optimiser moments and counters, PRNG key, env state, last timestep / dones and both hidden states -- everything
    ``learn(state)`` needs to continue bit-identically (mava/utils/checkpointing.py:108-145 saves exactly this pytree; the
    reference's own ``restore_params`` :147-198 reads back only params / hidden states).  Returns (state, timestep)."""
    from ..types import GPOLearnerState, HiddenStates, OptStates, Params, SableHiddenStates
    ck = torch.load(path, map_location="cpu", weights_only=False)
    st = ck["learner_state"]

    def dev(x):
        if torch.is_tensor(x):
            return x.to(device)
        if isinstance(x, dict):
            return {k: dev(v) for k, v in x.items()}
        return x

    hs = st["hstates"]
    state = GPOLearnerState(Params(dev(st["params"]["guider_params"]), dev(st["params"]["actor_params"])),
                            OptStates(dev(st["opt_states"]["guider_opt_state"]), dev(st["opt_states"]["actor_opt_state"])),
                            st["key"], dev(st["env_state"]), dev(st["timestep"]), dev(st["dones"]),
                            HiddenStates(SableHiddenStates(**dev(hs["sable_hidden_state"])), dev(hs["policy_hidden_state"])))
    return state, int(ck["timestep"])


def restore_params(path: str, restore_hstates: bool = False, device="cuda"):
    """Checkpointer.restore_params (mava/utils/checkpointing.py:147-198): the parameters (and optionally the hidden states)
    of a checkpoint."""
    state, _ = restore_learner_state(path, device)
    return state.params, (state.hstates if restore_hstates else None)
