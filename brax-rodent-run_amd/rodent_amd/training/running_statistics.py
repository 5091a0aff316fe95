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


_BUFS = {}


def update_from_sums(state: RunningStatisticsState, n_local: float, sums: torch.Tensor, std_min_value=1e-6, std_max_value=1e6) -> RunningStatisticsState:
    """The same update from S1 = sum (x - mean_old) and S2 = sum (x - mean_old)^2 of the local batch (float64 [2, obs]):
    sum (x - mean_old)(x - mean_new) = S2 - (mean_new - mean_old) S1.  Collective: ONE all-reduce of 1 + 2*obs doubles."""
    K = sums.shape[1]
    pack = torch.cat([torch.tensor([float(n_local)], dtype=torch.float64, device=sums.device), sums.reshape(-1)])
    D.all_reduce_sum_(pack)
    count = state.count.double() + pack[0]
    s1, s2 = pack[1:1 + K], pack[1 + K:]
    delta = s1 / count
    mean = state.mean.double() + delta
    summed_variance = state.summed_variance.double() + (s2 - delta * s1)
    std = torch.sqrt(torch.clamp(summed_variance, min=0) / count).clamp(std_min_value, std_max_value)
    return RunningStatisticsState(count.float(), mean.float(), summed_variance.float(), std.float())


def update_from_unroll_buffer(state: RunningStatisticsState, obs: torch.Tensor, T: int) -> RunningStatisticsState:
    """`update(state, obs[..., :T, :])` for the learner's unroll buffer obs [..., T + 1, K] (the rows t < T of every sequence are the
    transitions' observations).  On a GPU: the sums in one pass of the hand-written kernel (`rr_obs_moments`; no temporaries -- the
    tensor-expression form below makes two 6.6 GB ones per training step at the launcher's sizes); elsewhere the generic update."""
    if not obs.is_cuda:
        return update(state, obs[..., :T, :])
    from .. import hip
    n_local = obs.numel() // (obs.shape[-1] * obs.shape[-2]) * T
    return update_from_sums(state, n_local, hip.obs_moments(obs.contiguous(), T, state.mean.contiguous(), _BUFS))


def normalize(batch: torch.Tensor, state: RunningStatisticsState) -> torch.Tensor:
    return (batch - state.mean) / state.std
