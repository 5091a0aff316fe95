"""`Rodent`: the reference task env [REF Rodent_Env_Brax.py:19-162] on the HIP backend.

Same constructor arguments, `reset(rng)` / `step(state, action)` surface, reward / done / obs
arithmetic and `info['cur_frame']` bookkeeping as the reference; tensors are torch (leading env
axis N) instead of vmapped jax arrays.  `step` runs ONE fused kernel launch: n_frames physics
substeps + reward / done / observation epilogue (C ABI `rr_env_step`).
"""
from __future__ import annotations

import os
import warnings
from typing import Optional

import numpy as np
import torch

from .. import assets, jax_random
from .base import PipelineEnv, PipelineState, State, System

_XML_PATH = "./models/rodent_new.xml"   # [REF Rodent_Env_Brax.py:16]


class Rodent(PipelineEnv):

    def __init__(
        self,
        track_pos,
        forward_reward_weight=10,
        ctrl_cost_weight=0.1,
        healthy_reward=1.0,
        terminate_when_unhealthy=True,
        healthy_z_range=(0.03, 0.5),
        reset_noise_scale=1e-2,
        solver="cg",
        iterations: int = 6,
        ls_iterations: int = 6,
        vision=False,
        num_envs: int = 1,
        xml_path: str = _XML_PATH,
        device=None,
        **kwargs,
    ):
        if solver.lower() not in ("cg", "newton"):       # [REF Rodent_Env_Brax.py:42-45]
            raise ValueError(f"solver must be 'cg' or 'newton', got {solver!r}")
        if vision:
            raise NotImplementedError("vision observations are not part of the reference obs either")
        if iterations < 2:
            # measured on the float64 oracle as on the GPU: with ONE solver iteration per substep the rollout under
            # random actions reaches |qvel| > 1e4 and non-finite states within two env steps (Newton 1/4), within a few (CG 1/4)
            warnings.warn(f"Rodent(solver={solver!r}, iterations={iterations}): a single solver iteration per substep diverges "
                          "under random actions (non-finite states within a few env steps, on the CPU oracle too); use iterations >= 2",
                          RuntimeWarning, stacklevel=2)
        sys = System(assets.resolve_model(xml_path), iterations, ls_iterations, solver)
        physics_steps_per_control_step = 10   # [REF Rodent_Env_Brax.py:53-57]
        kwargs["n_frames"] = kwargs.get("n_frames", physics_steps_per_control_step)
        kwargs["backend"] = "hip"
        super().__init__(sys, num_envs=num_envs, device=device, **kwargs)
        self._track_pos = torch.as_tensor(np.asarray(track_pos), dtype=torch.float32).to(self.device).contiguous()
        self._forward_reward_weight = forward_reward_weight
        self._ctrl_cost_weight = ctrl_cost_weight
        self._healthy_reward = healthy_reward
        self._terminate_when_unhealthy = terminate_when_unhealthy
        self._healthy_z_range = healthy_z_range
        self._reset_noise_scale = reset_noise_scale
        self._vision = vision
        self._ctor = dict(track_pos=track_pos, forward_reward_weight=forward_reward_weight, ctrl_cost_weight=ctrl_cost_weight,
                          healthy_reward=healthy_reward, terminate_when_unhealthy=terminate_when_unhealthy,
                          healthy_z_range=healthy_z_range, reset_noise_scale=reset_noise_scale, solver=solver,
                          iterations=iterations, ls_iterations=ls_iterations, vision=vision, xml_path=xml_path,
                          n_frames=kwargs["n_frames"], pipeline_outputs=kwargs.get("pipeline_outputs", False),
                          contact_outputs=kwargs.get("contact_outputs", False), balance=kwargs.get("balance"),
                          rebalance_every=kwargs.get("rebalance_every", 4))

    def with_num_envs(self, num_envs: int, device=None):
        """A sibling env with another batch size (ppo.train builds its per-rank and eval envs this way)."""
        return Rodent(num_envs=num_envs, device=device or self.device, **self._ctor)

    def _env_io(self, cur_frame, obs, reward=None, done=None, metrics=None):
        return dict(track_pos=self._track_pos, cur_frame=cur_frame, obs=obs, reward=reward, done=done, metrics=metrics,
                    healthy_reward=self._healthy_reward, ctrl_cost_weight=self._ctrl_cost_weight,
                    healthy_z_range=self._healthy_z_range, terminate_when_unhealthy=self._terminate_when_unhealthy)

    def reset(self, rng) -> State:
        """Resets the environment to an initial state.  `rng`: uint32 keys [N, 2] (one jax-style
        PRNG key per env, as `jax.vmap(env.reset)(split(key, N))` passes) or an int seed."""
        N, dev, s = self.num_envs, self.device, self.sys
        if isinstance(rng, (int, np.integer)):
            rng = jax_random.split(jax_random.PRNGKey(int(rng)), N)
        keys = np.asarray(rng, dtype=np.uint32).reshape(N, 2)
        ks = jax_random.split(keys, 4)                       # rng, rng1, rng2, rng_pos
        start_frame = jax_random.randint(ks[:, 0], 0, 100)   # [N] int32
        low, hi = -self._reset_noise_scale, self._reset_noise_scale
        track = self._track_pos.cpu().numpy()
        qpos = np.tile(np.asarray(s.qpos0, dtype=np.float32), (N, 1))
        qpos[:, :3] = track[np.clip(start_frame, 0, len(track) - 1)]
        qpos = qpos + jax_random.uniform(ks[:, 1], s.nq, low, hi)
        qvel = jax_random.uniform(ks[:, 2], s.nv, low, hi)

        st = dict(qpos=torch.from_numpy(qpos).to(dev), qvel=torch.from_numpy(qvel).to(dev),
                  act=torch.zeros(N, s.na, device=dev), qacc_warmstart=torch.zeros(N, s.nv, device=dev))
        out = self._alloc_outputs(full=False)
        cur_frame = torch.from_numpy(start_frame.astype(np.int32)).to(dev)
        obs = torch.empty(N, s.obs_dim, device=dev)
        self._batch.env_reset(st, self._env_io(cur_frame, obs), out)
        zero = torch.zeros(N, device=dev)
        metrics = {"pos_reward": zero, "reward_quadctrl": zero.clone(), "reward_alive": zero.clone()}
        return State(PipelineState(**st, **out), obs, zero.clone(), zero.clone(), metrics, {"cur_frame": cur_frame})

    def unroll_wrapped(self, state: State, actions: torch.Tensor, episode_length: float) -> State:
        """`actions.shape[0]` steps with `EpisodeWrapper(episode_length)` + `AutoResetWrapper` (action_repeat 1) in ONE launch
        (`rr_env_unroll`): what `lax.scan` over the wrapped `step` is to the reference.  `state` is a state of the wrapped env
        (info carries steps, truncation, first_pipeline_state, first_obs); returns the state after the last step, bit for bit what
        the per-step calls give.  Intermediate observations / rewards are not returned (a random-action rollout needs none)."""
        if self._pipeline_outputs or self._contact_outputs:
            raise ValueError("a multi-step rollout returns no pipeline / contact outputs: build the env without them")
        N, dev, s = self.num_envs, self.device, self.sys
        ps, info = state.pipeline_state, state.info
        fps = info["first_pipeline_state"]
        st_in = dict(qpos=ps.qpos, qvel=ps.qvel, act=ps.act, qacc_warmstart=ps.qacc_warmstart)
        first = dict(qpos=fps.qpos, qvel=fps.qvel, act=fps.act, qacc_warmstart=fps.qacc_warmstart)
        st = {k: torch.empty_like(v) for k, v in st_in.items()}
        cur_frame = torch.empty_like(info["cur_frame"])
        obs = torch.empty(N, s.obs_dim, device=dev)
        reward, done, steps, trunc = (torch.empty(N, device=dev) for _ in range(4))
        metrics = torch.empty(N, 3, device=dev)
        actions = actions.to(dev, torch.float32).contiguous()
        self._rebalance()
        self._batch.env_unroll(st_in, st, actions, self._n_frames, self._env_io(cur_frame, obs, reward, done, metrics), info["cur_frame"],
                               first, info["first_obs"], state.done, info["steps"], steps, trunc, episode_length)
        ninfo = dict(info)
        ninfo.update(cur_frame=cur_frame, steps=steps, truncation=trunc)
        m = dict(state.metrics)
        m.update(pos_reward=metrics[:, 0], reward_quadctrl=metrics[:, 1], reward_alive=metrics[:, 2])
        return state.replace(pipeline_state=PipelineState(**st), obs=obs, reward=reward, done=done, metrics=m, info=ninfo)

    def unroll_policy_wrapped(self, state: State, episode_length: float, actor: dict, noise: torch.Tensor, traj: dict, segment: int = 0) -> State:
        """`generate_unroll` in one launch (`rr_env_unroll_policy`): T = noise.shape[0] x [policy(obs) -> tanh-normal sample -> step ->
        EpisodeWrapper + AutoResetWrapper], the transitions written into `traj` (views of the learner's buffers, batch-major).
        `actor`: the policy's parameters as the kernel takes them (`acting.actor_params`).  `segment` = L: the T steps are recorded as
        T / L trajectories ([U, N, L(+1), ...] buffers) -- a whole rollout phase in one launch.  Returns the state after the last step."""
        if self._pipeline_outputs or self._contact_outputs:
            raise ValueError("a multi-step rollout returns no pipeline / contact outputs: build the env without them")
        N, dev, T = self.num_envs, self.device, noise.shape[0]
        ps, info = state.pipeline_state, state.info
        fps = info["first_pipeline_state"]
        st_in = dict(qpos=ps.qpos, qvel=ps.qvel, act=ps.act, qacc_warmstart=ps.qacc_warmstart)
        first = dict(qpos=fps.qpos, qvel=fps.qvel, act=fps.act, qacc_warmstart=fps.qacc_warmstart)
        st = {k: torch.empty_like(v) for k, v in st_in.items()}
        cur_frame = torch.empty_like(info["cur_frame"])
        reward, done, steps, trunc = (torch.empty(N, device=dev) for _ in range(4))
        metrics = torch.empty(N, 3, device=dev)
        actions = torch.empty(T, N, self.action_size, device=dev)
        obs_in = state.obs.contiguous()            # (also fills the env io's obs slot, which this entry point does not write)
        self._batch.env_unroll_policy(st_in, st, T, self._n_frames, self._env_io(cur_frame, obs_in, reward, done, metrics), info["cur_frame"],
                                      first, info["first_obs"], state.done, info["steps"], steps, trunc, episode_length, actor, noise, actions,
                                      traj, obs_in, segment)
        ninfo = dict(info)
        ninfo.update(cur_frame=cur_frame, steps=steps, truncation=trunc)
        m = dict(state.metrics)
        m.update(pos_reward=metrics[:, 0], reward_quadctrl=metrics[:, 1], reward_alive=metrics[:, 2])
        L_ = segment or T
        obs = traj["obs"].reshape(T // L_, N, L_ + 1, -1)[-1, :, L_].contiguous()
        return state.replace(pipeline_state=PipelineState(**st), obs=obs, reward=reward, done=done, metrics=m, info=ninfo), actions

    def step(self, state: State, action: torch.Tensor) -> State:
        """Runs one timestep of the environment's dynamics."""
        N, dev, s = self.num_envs, self.device, self.sys
        ps = state.pipeline_state
        st_in = dict(qpos=ps.qpos, qvel=ps.qvel, act=ps.act, qacc_warmstart=ps.qacc_warmstart)
        st = {k: torch.empty_like(v) for k, v in st_in.items()}      # the previous state is left untouched (no copies)
        out = self._alloc_outputs(full=False)
        cur_frame = torch.empty_like(state.info["cur_frame"])
        obs = torch.empty(N, s.obs_dim, device=dev)
        reward, done = torch.empty(N, device=dev), torch.empty(N, device=dev)
        metrics = torch.empty(N, 3, device=dev)
        action = action.to(dev, torch.float32).contiguous()
        self._rebalance()
        self._batch.env_step_to(st_in, st, action, self._n_frames, self._env_io(cur_frame, obs, reward, done, metrics),
                                state.info["cur_frame"], out)
        info = dict(state.info)
        info["cur_frame"] = cur_frame
        m = dict(state.metrics)
        m.update(pos_reward=metrics[:, 0], reward_quadctrl=metrics[:, 1], reward_alive=metrics[:, 2])
        return state.replace(pipeline_state=PipelineState(**st, **out), obs=obs, reward=reward, done=done, metrics=m, info=info)
