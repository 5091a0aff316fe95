"""Observation normaliser with the update rule of `brax.training.acme.running_statistics`
(SURVEY.md Appendix E): state {count, mean, summed_variance, std}, init {0, 0, 0, 1};
batched Welford update with the three sums all-reduced over ranks (`psum`)."""
from __future__ import annotations

import dataclasses

import torch

from . import distributed as D


@dataclasses.dataclass
class RunningStatisticsState:
    count: torch.Tensor            # scalar float64-like (kept in float32 as upstream uses float32/int32)
    mean: torch.Tensor             # [obs]
    summed_variance: torch.Tensor  # [obs]
    std: torch.Tensor              # [obs]

    def clone(self):
        return RunningStatisticsState(self.count.clone(), self.mean.clone(), self.summed_variance.clone(), self.std.clone())


def init_state(size: int, device) -> RunningStatisticsState:
    z = torch.zeros(size, device=device)
    return RunningStatisticsState(torch.zeros((), device=device), z, z.clone(), torch.ones(size, device=device))


def update(state: RunningStatisticsState, batch: torch.Tensor, std_min_value=1e-6, std_max_value=1e6) -> RunningStatisticsState:
    """batch: [..., obs]; all leading dims are batch dims.  Collective: 1 + 2*obs floats per call."""
    x = batch.reshape(-1, batch.shape[-1])
    n = torch.tensor(float(x.shape[0]), device=x.device)
    D.all_reduce_sum_(n)
    count = state.count + n
    diff_to_old_mean = x - state.mean
    mean_update = diff_to_old_mean.sum(0)
    D.all_reduce_sum_(mean_update)
    mean = state.mean + mean_update / count
    diff_to_new_mean = x - mean
    variance_update = (diff_to_old_mean * diff_to_new_mean).sum(0)
    D.all_reduce_sum_(variance_update)
    summed_variance = state.summed_variance + variance_update
    std = torch.sqrt(torch.clamp(summed_variance, min=0) / count).clamp(std_min_value, std_max_value)
    return RunningStatisticsState(count, mean, summed_variance, std)


def normalize(batch: torch.Tensor, state: RunningStatisticsState) -> torch.Tensor:
    return (batch - state.mean) / state.std
