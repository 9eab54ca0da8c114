"""Pytree surface of the learner on torch tensors (mava/types.py:126-214, mava/systems/gpo/types.py:25-83).

Same names and fields as the reference NamedTuples; leaves are device tensors (no (device, batch)
replication dims: a process owns its groups, parameters are shared by construction).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Generic, NamedTuple, Optional, TypeVar

import torch

Metrics = Dict[str, Any]


class Observation(NamedTuple):  # mava/types.py:126-136
    agents_view: torch.Tensor   # (N, A, F)
    action_mask: Optional[torch.Tensor]  # (N, A, K) bool; None = every action legal (CoordSum)
    step_count: torch.Tensor    # (N, A) or (N,)


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


class TimeStep(NamedTuple):
    step_type: torch.Tensor
    reward: torch.Tensor
    discount: torch.Tensor
    observation: Observation
    extras: Dict[str, Any]

    def last(self) -> torch.Tensor:
        return self.step_type == 2


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
