"""One-process-per-GPU data parallelism (the reference: `jax.pmap(axis_name='i')` -> NCCL
[UP brax.training.agents.ppo.train]).  Here: torch.distributed, backend nccl (= RCCL over xGMI) on
GPUs, gloo on CPU for the tests.  Only all-reduces are used, on flat persistent buffers."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_mean_(t: torch.Tensor) -> torch.Tensor:
    """`jax.lax.pmean(x, 'i')`."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(world_size())
    return t


class FlatGrads:
    """All parameters' gradients live in ONE flat buffer so the per-minibatch `pmean(grads)`
    [UP brax.training.gradients.gradient_update_fn] is a single all-reduce (2.5 MB, latency-bound
    on xGMI: one collective per minibatch instead of one per tensor)."""

    def __init__(self, params):
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero_(self):
        self.flat.zero_()

    def pmean_(self):
        all_reduce_mean_(self.flat)
