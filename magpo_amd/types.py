"""Pytree surface of the learner on torch tensors (mava/types.py:126-214, mava/systems/gpo/types.py:25-83).

Same names and fields as the reference NamedTuples; leaves are device tensors (no (device, batch)
replication dims: a process owns its groups, parameters are shared by construction).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Generic, NamedTuple, Optional, TypeVar

import torch

Metrics = Dict[str, Any]


class Observation(NamedTuple):  # mava/types.py:126-136
    agents_view: torch.Tensor   # (N, A, F) f32 (a view of rows ``stride(1)`` floats apart: wide observations are padded to 128)
    action_mask: torch.Tensor   # (N, A, K) u8 / bool (all ones for CoordSum, matrax.py:117-134)
    step_count: torch.Tensor    # (N, A) i32


class Params(NamedTuple):  # gpo/types.py:25-29
    guider_params: Dict[str, torch.Tensor]
    actor_params: Dict[str, torch.Tensor]


class OptStates(NamedTuple):  # gpo/types.py:32-36
    guider_opt_state: Dict[str, Any]
    actor_opt_state: Dict[str, Any]


class SableHiddenStates(NamedTuple):  # gpo/types.py:47-52
    encoder: torch.Tensor
    decoder_self_retn: torch.Tensor
    decoder_cross_retn: torch.Tensor


class HiddenStates(NamedTuple):  # gpo/types.py:55-59
    sable_hidden_state: SableHiddenStates
    policy_hidden_state: torch.Tensor


class StepType:  # jumanji.types.StepType
    FIRST, MID, LAST = 0, 1, 2


class TimeStep(NamedTuple):  # jumanji.types.TimeStep as the wrappers leave it (mava/wrappers/episode_metrics.py:79-112)
    step_type: torch.Tensor     # (N,) i8: 0 first, 1 mid, 2 last
    reward: torch.Tensor        # (N, A) f32
    discount: torch.Tensor      # (N, A) f32
    observation: Observation
    extras: Dict[str, Any]      # {"episode_metrics": {episode_return, episode_length, is_terminal_step}, "env_metrics": {}}

    def first(self) -> torch.Tensor:
        return self.step_type == StepType.FIRST

    def mid(self) -> torch.Tensor:
        return self.step_type == StepType.MID

    def last(self) -> torch.Tensor:
        return self.step_type == StepType.LAST


class GPOLearnerState(NamedTuple):  # gpo/types.py:62-71
    params: Params
    opt_states: OptStates
    key: Any
    env_state: Any
    timestep: Any
    dones: Any
    hstates: Any


class GPOTransition(NamedTuple):  # gpo/types.py:74-83
    done: torch.Tensor
    action: torch.Tensor
    value: torch.Tensor
    reward: torch.Tensor
    log_prob: torch.Tensor
    obs: Any
    hstates: Any


_S = TypeVar("_S")


class ExperimentOutput(NamedTuple):  # mava/types.py:199-204
    learner_state: Any
    episode_metrics: Metrics
    train_metrics: Metrics


LearnerFn = Callable[[Any], ExperimentOutput]  # mava/types.py:207
LearnerState = GPOLearnerState
