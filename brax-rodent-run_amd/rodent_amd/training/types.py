"""`TrainingState` of `brax.training.agents.ppo.train` [UP; SURVEY.md a28]: what one training step carries over."""
from __future__ import annotations

import dataclasses
from typing import Any

import torch


@dataclasses.dataclass
class PPONetworkParams:
    policy: Any      # the policy MLP (nn.Module; its parameters are the leaves)
    value: Any       # the value MLP


@dataclasses.dataclass
class TrainingState:
    """Contains training state for the learner."""
    optimizer_state: Any             # torch.optim.Adam (optax.adam(lr): b1 .9, b2 .999, eps 1e-8)
    params: PPONetworkParams
    normalizer_params: Any           # running_statistics.RunningStatisticsState
    env_steps: torch.Tensor          # int64 scalar, += env_steps_per_training_step per training step (upstream: int32)
