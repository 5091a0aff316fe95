"""Training wrappers with the semantics of `brax.envs.wrappers.training`, batched.

The env is batched by construction, so `VmapWrapper` only validates shapes.  `EpisodeWrapper`
maintains `info['steps']` / `info['truncation']`; `AutoResetWrapper` selects the stored first
state back on `done` -- no re-randomisation and `info` (hence `cur_frame`) is NOT restored, exactly
as upstream.
"""
from __future__ import annotations

import dataclasses

import torch

from .base import PipelineState, State


def _tree_map2(fn, a, b):
    """Apply fn leaf-wise over two identically structured pytrees (dataclass / dict / tensor / None)."""
    if a is None:
        return None
    if torch.is_tensor(a):
        return fn(a, b)
    if dataclasses.is_dataclass(a):
        return type(a)(**{f.name: _tree_map2(fn, getattr(a, f.name), getattr(b, f.name)) for f in dataclasses.fields(a)})
    if isinstance(a, dict):
        return {k: _tree_map2(fn, a[k], b[k]) for k in a}
    raise TypeError(type(a))


class Wrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name == "__setstate__":
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, rng):
        return self.env.reset(rng)

    def step(self, state, action):
        return self.env.step(state, action)

    @property
    def unwrapped(self):
        return self.env.unwrapped if hasattr(self.env, "unwrapped") else self.env


class VmapWrapper(Wrapper):
    """No-op: the HIP env already carries the leading env axis."""

    def __init__(self, env, batch_size=None):
        super().__init__(env)
        if batch_size is not None and batch_size != env.num_envs:
            raise ValueError(f"batch_size {batch_size} != env.num_envs {env.num_envs}")


class EpisodeWrapper(Wrapper):
    """Maintains episode step count and sets done at episode end."""

    def __init__(self, env, episode_length: int, action_repeat: int):
        super().__init__(env)
        self.episode_length = episode_length
        self.action_repeat = action_repeat

    def reset(self, rng):
        state = self.env.reset(rng)
        state.info["steps"] = torch.zeros_like(state.reward)
        state.info["truncation"] = torch.zeros_like(state.reward)
        return state

    def step(self, state, action):
        total = None
        nstate = state
        for _ in range(self.action_repeat):
            nstate = self.env.step(nstate, action)
            total = nstate.reward if total is None else total + nstate.reward
        state = nstate.replace(reward=total)
        steps = state.info["steps"] + self.action_repeat
        one, zero = torch.ones_like(state.done), torch.zeros_like(state.done)
        over = steps >= self.episode_length
        done = torch.where(over, one, state.done)
        state.info["truncation"] = torch.where(over, 1 - state.done, zero)
        state.info["steps"] = steps
        return state.replace(done=done)


class AutoResetWrapper(Wrapper):
    """Automatically resets Brax envs that are done (to the FIRST state of the batch member)."""

    def reset(self, rng):
        state = self.env.reset(rng)
        state.info["first_pipeline_state"] = state.pipeline_state
        state.info["first_obs"] = state.obs
        return state

    def step(self, state, action):
        if "steps" in state.info:
            steps = state.info["steps"]
            steps = torch.where(state.done.bool(), torch.zeros_like(steps), steps)
            state.info.update(steps=steps)
        state = state.replace(done=torch.zeros_like(state.done))
        state = self.env.step(state, action)
        done = state.done.bool()

        def where_done(x, y):
            d = done.reshape((-1,) + (1,) * (x.dim() - 1))
            return torch.where(d, x, y)

        new_ps = _tree_map2(where_done, state.info["first_pipeline_state"], state.pipeline_state)
        obs = where_done(state.info["first_obs"], state.obs)
        return state.replace(pipeline_state=new_ps, obs=obs)


class EvalWrapper(Wrapper):
    """Accumulates episode metrics for evaluation (brax EvalWrapper)."""

    def reset(self, rng):
        rs = self.env.reset(rng)
        rs.metrics["reward"] = rs.reward
        z = torch.zeros_like(rs.reward)
        rs.info["eval_metrics"] = dict(episode_metrics={k: torch.zeros_like(v) for k, v in rs.metrics.items()},
                                       active_episodes=torch.ones_like(rs.reward), episode_steps=z.clone())
        return rs

    def step(self, state, action):
        em = state.info["eval_metrics"]
        nstate = self.env.step(state, action)
        nstate.metrics["reward"] = nstate.reward
        steps = em["episode_steps"] + em["active_episodes"]
        ep = {k: em["episode_metrics"][k] + nstate.metrics[k] * em["active_episodes"] for k in em["episode_metrics"]}
        active = em["active_episodes"] * (1 - nstate.done)
        nstate.info["eval_metrics"] = dict(episode_metrics=ep, active_episodes=active, episode_steps=steps)
        return nstate


class FusedEpisodeAutoResetWrapper(Wrapper):
    """EpisodeWrapper + AutoResetWrapper of a HIP env with action_repeat 1: the same values, but the ~15 elementwise
    launches of the composed wrappers (step count, truncation, done, ten `where(done, first, current)` selects) are one
    launch of `rr_wrap_episode_autoreset` on the freshly produced state.  `tests/test_gpu_env.py` holds it to bitwise
    equality with the composition."""

    def __init__(self, env, episode_length: int):
        super().__init__(env)
        self.episode_length = episode_length
        self.action_repeat = 1

    def reset(self, rng):
        state = self.env.reset(rng)
        state.info["steps"] = torch.zeros_like(state.reward)
        state.info["truncation"] = torch.zeros_like(state.reward)
        state.info["first_pipeline_state"] = state.pipeline_state
        state.info["first_obs"] = state.obs
        return state

    def step(self, state, action):
        from .. import hip
        nstate = self.env.step(state, action)              # fresh tensors: safe to finish in place
        fps, ps = nstate.info["first_pipeline_state"], nstate.pipeline_state
        names = [f.name for f in dataclasses.fields(ps) if torch.is_tensor(getattr(ps, f.name))]
        first = [getattr(fps, n) for n in names] + [nstate.info["first_obs"]]
        cur = [getattr(ps, n) for n in names] + [nstate.obs]
        steps, trunc = torch.empty_like(nstate.done), torch.empty_like(nstate.done)
        hip.wrap_episode_autoreset(first, cur, state.done, state.info["steps"], nstate.done, steps, trunc,
                                   self.episode_length, 1)
        nstate.info["steps"] = steps
        nstate.info["truncation"] = trunc
        return nstate


    def unroll(self, state, actions):
        """`actions` [T, N, nu]: T wrapped steps in one launch (`Rodent.unroll_wrapped` / C ABI `rr_env_unroll`); the state after the
        last step, equal to T calls of `step` bit for bit."""
        base = self.env.unwrapped if hasattr(self.env, "unwrapped") else self.env
        return base.unroll_wrapped(state, actions, self.episode_length)


    def unroll_policy(self, state, actor, noise, traj, segment: int = 0):
        """`acting.generate_unroll` in one launch: see `Rodent.unroll_policy_wrapped` (C ABI `rr_env_unroll_policy`).
        Returns (state after the last step, actions taken [T, N, nu])."""
        base = self.env.unwrapped if hasattr(self.env, "unwrapped") else self.env
        return base.unroll_policy_wrapped(state, self.episode_length, actor, noise, traj, segment)


def wrap(env, episode_length: int = 1000, action_repeat: int = 1):
    """brax.envs.wrappers.training.wrap: Vmap -> Episode -> AutoReset (one fused wrapper for a HIP env with
    action_repeat 1; RR_FUSED_WRAPPERS=0 selects the composition)."""
    import os
    base = env.unwrapped if hasattr(env, "unwrapped") else env
    if action_repeat == 1 and hasattr(base, "_batch") and os.environ.get("RR_FUSED_WRAPPERS", "1") == "1":
        return FusedEpisodeAutoResetWrapper(VmapWrapper(env), episode_length)
    env = VmapWrapper(env)
    env = EpisodeWrapper(env, episode_length, action_repeat)
    return AutoResetWrapper(env)
