"""Checkpoint saving (mava/utils/checkpointing.py:34-145) without Orbax: ``torch.save`` of the LearnerState pytree.

Mirrors the reference's behaviour for rec_magpo: save-only (the system never restores, rec_magpo.py:733-739, 779-785),
best-by-``episode_return`` retention with ``max_to_keep``, ``save_interval_steps`` counted in evaluations, config stored as
metadata next to the state.  ``restore_learner_state`` is provided for resuming sweeps."""
from __future__ import annotations

import glob
import json
import os
import time
import warnings
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

CHECKPOINTER_VERSION = 2.0


def _to_cpu(x: Any) -> Any:
    """Checkpoints hold only tensors, plain containers and numbers, so that they load with ``weights_only=True`` (no pickle code
    execution from the results directory): numpy arrays (PRNG keys) become int64 / float64 tensors tagged with their dtype."""
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, np.ndarray):
        return {"__ndarray__": str(x.dtype), "data": torch.from_numpy(x.astype(np.float64 if x.dtype.kind == "f" else np.int64))}
    if isinstance(x, np.generic):
        return x.item()
    if isinstance(x, dict):
        return {k: _to_cpu(v) for k, v in x.items()}
    if isinstance(x, tuple) and hasattr(x, "_fields"):
        return {f: _to_cpu(getattr(x, f)) for f in x._fields}
    if isinstance(x, (list, tuple)):
        return [_to_cpu(v) for v in x]
    return x


def _atomic_save(obj: Any, path: str) -> None:
    """Write next to the target, flush to disk and rename: a kill (or a power loss) during the write leaves the previous file set
    intact and never a renamed-but-empty file (resumable sweeps)."""
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        torch.save(obj, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def _rank_path(path: str, rank: int) -> str:
    return path if rank == 0 else path[:-3] + f".rank{rank}.pt"


ROLLOUT_FIELDS = ("key", "env_state", "timestep", "dones", "hstates")   # what differs between ranks (params / opt_states are replicated)


class Checkpointer:
    """``rank`` / ``world``: every rank of a multi-GPU job owns different envs, so rank 0 writes the full learner state to
    ``{t}.pt`` and every other rank its ROLLOUT state (env state, last timestep, dones, hidden states, key) to ``{t}.rank{r}.pt``;
    retention (rank 0) removes a timestep's files together."""

    def __init__(self, model_name: str, metadata: Optional[Dict] = None, base_path: str = "results/", rel_dir: str = "checkpoints",
                 checkpoint_uid: Optional[str] = None, save_interval_steps: int = 1, max_to_keep: Optional[int] = 1,
                 keep_period: Optional[int] = None, keep_latest: bool = False, rank: int = 0, world: int = 1):
        uid = checkpoint_uid or time.strftime("%Y%m%d%H%M%S")
        self.dir = os.path.join(base_path, rel_dir, model_name, uid)
        os.makedirs(self.dir, exist_ok=True)
        self.interval = max(1, int(save_interval_steps))
        self.max_to_keep = max_to_keep
        self.keep_period = keep_period
        self.keep_latest = bool(keep_latest)   # extension: rank by timestep instead of episode_return (resumable sweeps)
        self.rank, self.world = int(rank), int(world)
        self.kept: List[Tuple[float, int, str]] = []   # (episode_return, timestep, path)
        self.calls = 0
        if self.rank == 0:
            # a relaunch into an existing directory (resume): the files of the previous run take part in the retention.  A file that
            # does not load is NEVER deleted (the atomic write cannot leave a truncated {t}.pt, so a failure means something else: an
            # older format, another torch version, a transient I/O error): it is set aside as {t}.pt.corrupt, like JsonLogger does.
            for f in list_checkpoints(self.dir):
                try:
                    head = torch.load(f, map_location="cpu", weights_only=True)
                    self.kept.append((float(head.get("episode_return", 0.0)), int(head["timestep"]), f))
                except Exception as e:
                    warnings.warn(f"checkpoint {f} does not load ({e!r}); kept aside as {os.path.basename(f)}.corrupt")
                    os.replace(f, f + ".corrupt")
            # what a kill really leaves behind: temporaries of interrupted writes
            for f in glob.glob(os.path.join(self.dir, "*.tmp")):
                os.remove(f)
            self._sweep_orphans()
            tmp = os.path.join(self.dir, "metadata.json.tmp")
            with open(tmp, "w") as f:
                json.dump({"checkpointer_version": CHECKPOINTER_VERSION, **(metadata or {})}, f, indent=1, default=str)
            os.replace(tmp, os.path.join(self.dir, "metadata.json"))

    def save(self, timestep: int, unreplicated_learner_state: Any, episode_return: float = 0.0, extras: Optional[Dict] = None) -> bool:
        """WRITE this rank's file(s) of ``timestep``.  Older checkpoints are removed by ``prune()``: a single-rank job prunes right
        here; a multi-rank job calls ``prune()`` after a barrier behind every rank's ``save`` (run_experiment does), so that the
        previous complete checkpoint set disappears only once the new one is complete."""
        self.calls += 1
        if (self.calls - 1) % self.interval:
            return False
        path = os.path.join(self.dir, f"{int(timestep)}.pt")
        state = _to_cpu(unreplicated_learner_state)
        if self.rank != 0:
            _atomic_save({"learner_state": {f: state[f] for f in ROLLOUT_FIELDS}, "timestep": int(timestep), "rank": self.rank},
                         _rank_path(path, self.rank))
            return True
        _atomic_save({"learner_state": state, "timestep": int(timestep), "episode_return": float(episode_return), "world": self.world,
                      "extras": _to_cpu(extras)}, path)
        self.kept.append((float(episode_return), int(timestep), path))
        if self.world == 1:
            self.prune()
        return True

    def _complete(self, path: str) -> bool:
        return os.path.exists(path) and all(os.path.exists(_rank_path(path, r)) for r in range(1, self.world))

    def prune(self) -> None:
        """Rank 0: keep the best ``max_to_keep`` checkpoints by episode_return (ties: latest), like orbax best_fn / best_mode="max".
        Nothing is removed while the newest timestep's file set is incomplete (a rank has not written its file yet, or was killed
        before it could): until then the previous complete set is the one a resume needs."""
        if self.rank != 0 or not self.max_to_keep or not self.kept:
            return
        newest = max(self.kept, key=lambda e: e[1])[2]
        if not self._complete(newest):
            return
        self.kept.sort(key=(lambda e: e[1]) if self.keep_latest else (lambda e: (e[0], e[1])))
        while len(self.kept) > int(self.max_to_keep):
            ret, ts, victim = self.kept[0]
            if self.keep_period and ts % int(self.keep_period) == 0:
                break
            self.kept.pop(0)
            for f in [victim] + glob.glob(victim[:-3] + ".rank*.pt"):
                if os.path.exists(f):
                    os.remove(f)
        self._sweep_orphans()

    def _sweep_orphans(self) -> None:
        """Rank files {t}.rank{r}.pt whose {t}.pt is gone (pruned before that rank had renamed its file) and that are older than
        the newest rank-0 file: never part of a loadable checkpoint again (~0.8 GB each at 16 384 envs)."""
        heads = {int(os.path.basename(f)[:-3]) for f in list_checkpoints(self.dir)}
        newest = max(heads) if heads else -1
        for f in glob.glob(os.path.join(self.dir, "*.rank*.pt")):
            t = os.path.basename(f).split(".")[0]
            if t.isdigit() and int(t) not in heads and int(t) < newest:
                os.remove(f)


def list_checkpoints(cdir: str) -> List[str]:
    """The rank-0 checkpoint files ``{timestep}.pt`` of a directory, oldest first."""
    out = [f for f in glob.glob(os.path.join(cdir, "*.pt")) if os.path.basename(f)[:-3].isdigit()]
    return sorted(out, key=lambda f: int(os.path.basename(f)[:-3]))


def _from_saved(x: Any) -> Any:
    if isinstance(x, dict):
        if "__ndarray__" in x:
            return x["data"].numpy().astype(np.dtype(x["__ndarray__"]))
        return {k: _from_saved(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_from_saved(v) for v in x]
    return x


def load_checkpoint(path: str) -> Dict[str, Any]:
    """``torch.load(weights_only=True)`` + the numpy leaves restored."""
    return _from_saved(torch.load(path, map_location="cpu", weights_only=True))


def _check_world(ck: Dict[str, Any], path: str, world: int) -> None:
    """A checkpoint belongs to the job size that wrote it -- checked for EVERY world size: a single process resuming an N-rank
    checkpoint would silently continue rank 0's envs with 1/N of the batch and a different gradient mean."""
    if int(ck.get("world", 1)) != int(world):
        raise ValueError(f"{path} was written by a {ck.get('world', 1)}-rank job, this one has {world} rank(s)")


def latest_valid_checkpoint(cdir: str, rank: int = 0, world: int = 1) -> str:
    """Newest checkpoint of ``cdir`` that loads (a truncated newest file -- the run was killed while writing -- falls back to the
    one before it) and, for a multi-rank job, whose rank files are all present."""
    for f in reversed(list_checkpoints(cdir)):
        try:
            ck = load_checkpoint(f)
            _check_world(ck, f, world)
            for r in range(1, world):
                load_checkpoint(_rank_path(f, r))
            return f
        except ValueError:
            raise
        except Exception:
            continue
    raise FileNotFoundError(f"no loadable checkpoint under {cdir}")


def restore_learner_state(path: str, device="cuda", rank: int = 0, world: int = 1):
    """Load a checkpoint written by ``Checkpointer.save`` and rebuild the full GPOLearnerState on ``device``: parameters,
    optimiser moments and counters, PRNG key, env state, last timestep / dones and both hidden states -- everything
    ``learn(state)`` needs to continue bit-identically (mava/utils/checkpointing.py:108-145 saves exactly this pytree; the
    reference's own ``restore_params`` :147-198 reads back only params / hidden states).  Rank r > 0 of a multi-rank job takes the
    replicated parts (params, optimiser state) from ``path`` and its own rollout state from ``{t}.rank{r}.pt``.
    Format: a torch file of tensors and plain containers (not Orbax); loaded with ``weights_only=True``.  Returns (state, timestep)."""
    from ..types import GPOLearnerState, HiddenStates, OptStates, Params, SableHiddenStates
    ck = load_checkpoint(path)
    st = ck["learner_state"]
    _check_world(ck, path, world)
    if rank != 0:
        mine = load_checkpoint(_rank_path(path, rank))
        if int(mine["timestep"]) != int(ck["timestep"]):
            raise ValueError(f"{_rank_path(path, rank)} belongs to another timestep")
        st = {**st, **mine["learner_state"]}

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
