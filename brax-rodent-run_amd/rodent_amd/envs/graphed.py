"""Replay of a step function as a HIP graph: what `jax.jit(env.step)` / `lax.scan` over it are to the reference
[REF brax_rodent_run_ppo.py:141-142 jit_reset / jit_step; UP brax.training.acting.generate_unroll's scan].

`GraphedSteps(step_fn, state, steps_per_replay)` captures `steps_per_replay` consecutive calls `state = step_fn(state)` --
every launch they make (action sampling, the fused env-step kernel, the wrapper kernel) -- once, on the stream the env's
batch is bound to, and `replay()` re-issues them with one host call.  The state pytree lives in fixed buffers: the graph ends
by copying the last state back over the first (leaves that pass through unchanged are skipped), so replays chain.

Why it matters here: one env step is three launches and ~0.7 ms of host work (Python wrappers, output allocation); a 2048-env
step is 1.4 ms on the GPU.  Stepping the batch as several sub-batches on separate streams -- so that one sub-batch's slowest
environments overlap the others' bulk -- needs the host cost per step well below that, which replay gives (~20 us).
"""
from __future__ import annotations

import dataclasses
from typing import Callable, Iterable, List

import torch


def tree_leaves(tree, out=None) -> List[torch.Tensor]:
    """Tensors of a pytree (dataclass / dict / list / tuple / tensor / None), in field / insertion order."""
    out = [] if out is None else out
    if tree is None:
        return out
    if torch.is_tensor(tree):
        out.append(tree)
    elif dataclasses.is_dataclass(tree):
        for f in dataclasses.fields(tree):
            tree_leaves(getattr(tree, f.name), out)
    elif isinstance(tree, dict):
        for k in tree:
            tree_leaves(tree[k], out)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            tree_leaves(v, out)
    return out


def tree_map(fn, tree):
    if tree is None:
        return None
    if torch.is_tensor(tree):
        return fn(tree)
    if dataclasses.is_dataclass(tree):
        return type(tree)(**{f.name: tree_map(fn, getattr(tree, f.name)) for f in dataclasses.fields(tree)})
    if isinstance(tree, dict):
        return {k: tree_map(fn, v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(tree_map(fn, v) for v in tree)
    return tree


class GraphedSteps:
    def __init__(self, step_fn: Callable, state, steps_per_replay: int, stream: torch.cuda.Stream,
                 generators: Iterable[torch.Generator] = ()):
        """`state` must already be an OUTPUT of `step_fn` (same pytree structure in and out) and `step_fn` must have run
        eagerly before (lazy allocations done).  `stream`: the non-default stream the env's batch was created on."""
        self.steps_per_replay = steps_per_replay
        self.stream = stream
        with torch.cuda.stream(stream):
            seen = {}

            def own(t):          # one private buffer per distinct tensor (aliases inside the tree stay aliases)
                key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()))
                if key not in seen:
                    seen[key] = t.clone()
                return seen[key]
            self.state = tree_map(own, state)
        stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        for g in generators:
            self.graph.register_generator_state(g)
        # thread_local: a process group's watchdog thread may query events while this thread captures (multi-GPU runs)
        with torch.cuda.graph(self.graph, stream=stream, capture_error_mode="thread_local"):
            s = self.state
            for _ in range(steps_per_replay):
                s = step_fn(s)
            src, dst = tree_leaves(s), tree_leaves(self.state)
            if len(src) != len(dst):
                raise ValueError("step_fn changes the structure of the state pytree")
            done = set()
            for d, x in zip(dst, src):
                if d.data_ptr() != x.data_ptr() and d.data_ptr() not in done:
                    d.copy_(x)
                    done.add(d.data_ptr())

    def replay(self):
        """Advance the state by `steps_per_replay` steps (asynchronously, on the graph's stream); returns the state buffers."""
        with torch.cuda.stream(self.stream):
            self.graph.replay()
        return self.state
