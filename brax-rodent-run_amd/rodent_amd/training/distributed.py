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


# Time spent in the collectives (config 3 / 4 report rollout / learner / all-reduce).  On a GPU the all-reduce is
# bracketed by events on the CURRENT stream -- torch issues the RCCL kernel on its own stream and makes the current stream wait for
# it, so the pair spans the collective as the learner's stream sees it (incl. waiting for the slowest rank); on the CPU, wall time.
_TIMER = {"on": False, "events": [], "wall": 0.0}


def time_collectives(enable: bool):
    _TIMER["on"] = bool(enable)
    _TIMER["events"].clear()
    _TIMER["wall"] = 0.0


def pop_collective_seconds() -> float:
    """Seconds inside all-reduces since the last call (the caller has synchronised the device)."""
    s = _TIMER["wall"] + sum(e0.elapsed_time(e1) for e0, e1 in _TIMER["events"]) * 1e-3
    _TIMER["events"].clear()
    _TIMER["wall"] = 0.0
    return s


def _all_reduce(t: torch.Tensor):
    if not _TIMER["on"]:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    elif t.is_cuda and not torch.cuda.is_current_stream_capturing():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        e1.record()
        _TIMER["events"].append((e0, e1))
    else:
        import time
        t0 = time.perf_counter()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        _TIMER["wall"] += time.perf_counter() - t0


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        _all_reduce(t)
    return t


def all_reduce_mean_(t: torch.Tensor) -> torch.Tensor:
    """`jax.lax.pmean(x, 'i')`."""
    if world_size() > 1:
        _all_reduce(t)
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
