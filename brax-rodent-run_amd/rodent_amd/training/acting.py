"""Rollouts: `actor_step` / `generate_unroll` / `Evaluator` with the data flow of `brax.training.acting`
(SURVEY.md Appendix E), storing each observation ONCE per unroll ([T+1] instead of obs + next_obs:
the upstream Transition keeps both, 13.6 GB per training step at 2048 envs)."""
from __future__ import annotations

import time
from typing import Callable, Dict

import torch

from ..envs import wrappers


class UnrollBuffer:
    """Trajectory storage of one training step: [U unrolls, N envs, T(+1) steps, ...], a trajectory
    (u, n) is contiguous so minibatch gathers copy whole rows."""

    def __init__(self, U, N, T, obs_dim, act_dim, device):
        self.U, self.N, self.T = U, N, T
        self.obs = torch.empty(U, N, T + 1, obs_dim, device=device)
        self.raw_action = torch.empty(U, N, T, act_dim, device=device)
        self.log_prob = torch.empty(U, N, T, device=device)
        self.reward = torch.empty(U, N, T, device=device)
        self.discount = torch.empty(U, N, T, device=device)
        self.truncation = torch.empty(U, N, T, device=device)

    def flat(self):
        f = lambda x: x.reshape((self.U * self.N,) + x.shape[2:])
        return dict(obs=f(self.obs), raw_action=f(self.raw_action), log_prob=f(self.log_prob), reward=f(self.reward),
                    discount=f(self.discount), truncation=f(self.truncation))


@torch.no_grad()
def generate_unroll(env, state, policy: Callable, buf: UnrollBuffer, u: int, generator=None):
    """Collect `buf.T` steps with `policy`; transition t holds (obs_t, action_t, reward_{t+1}, 1-done_{t+1},
    truncation_{t+1}) and obs[T] is the bootstrap observation [UP acting.generate_unroll]."""
    for t in range(buf.T):
        buf.obs[u, :, t] = state.obs
        action, extras = policy(state.obs, generator)
        nstate = env.step(state, action)
        buf.raw_action[u, :, t] = extras["raw_action"]
        buf.log_prob[u, :, t] = extras["log_prob"]
        buf.reward[u, :, t] = nstate.reward
        buf.discount[u, :, t] = 1 - nstate.done
        buf.truncation[u, :, t] = nstate.info["truncation"]
        state = nstate
    buf.obs[u, :, buf.T] = state.obs
    return state


class Evaluator:
    """`brax.training.acting.Evaluator`: episode_length steps of a freshly reset eval env, episode metric sums."""

    def __init__(self, eval_env, eval_policy_fn: Callable, num_eval_envs: int, episode_length: int, action_repeat: int, key):
        self._key = key
        self._eval_walltime = 0.0
        self._env = wrappers.EvalWrapper(eval_env)
        self._policy_fn = eval_policy_fn
        self._steps_per_unroll = episode_length * num_eval_envs
        self._unroll_length = episode_length // action_repeat

    @torch.no_grad()
    def run_evaluation(self, policy_params, training_metrics: Dict, aggregate_episodes: bool = True) -> Dict:
        from .. import jax_random
        self._key, unroll_key = jax_random.split(self._key)
        t = time.time()
        policy = self._policy_fn(policy_params)
        state = self._env.reset(jax_random.split(unroll_key, self._env.num_envs))
        for _ in range(self._unroll_length):
            action, _ = policy(state.obs, None)
            state = self._env.step(state, action)
        if state.obs.is_cuda:
            torch.cuda.synchronize(state.obs.device)
        em = state.info["eval_metrics"]
        epoch_eval_time = time.time() - t
        metrics = {}
        for name, value in em["episode_metrics"].items():
            metrics[f"eval/episode_{name}"] = float(value.mean()) if aggregate_episodes else value.cpu().numpy()
        metrics["eval/avg_episode_length"] = float(em["episode_steps"].mean())
        metrics["eval/epoch_eval_time"] = epoch_eval_time
        metrics["eval/sps"] = self._steps_per_unroll / epoch_eval_time
        self._eval_walltime += epoch_eval_time
        metrics = {"eval/walltime": self._eval_walltime, **training_metrics, **metrics}
        return metrics
