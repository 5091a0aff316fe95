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


def actor_params(policy_net, normalizer_params, min_std: float) -> dict:
    """The policy's parameters in the layout of the in-kernel actor (`rr_env_unroll_policy`): first layer as torch holds it, hidden
    weights transposed, head transposed and zero-padded to 64 outputs.  Copies: call again after the parameters change."""
    layers = list(policy_net.layers)
    dev = layers[0].weight.device
    P = layers[-1].out_features
    head_wt = torch.zeros(32, 64, device=dev)
    head_wt[:, :P] = layers[-1].weight.detach().t()
    head_b = torch.zeros(64, device=dev)
    head_b[:P] = layers[-1].bias.detach()
    out = dict(w0=layers[0].weight.detach().contiguous(), b0=layers[0].bias.detach().contiguous(),
               hidden_wt=[l.weight.detach().t().contiguous() for l in layers[1:-1]], hidden_b=[l.bias.detach().contiguous() for l in layers[1:-1]],
               head_wt=head_wt, head_b=head_b, min_std=min_std, mean=None, std=None)
    if normalizer_params is not None:
        out.update(mean=normalizer_params.mean.contiguous(), std=normalizer_params.std.contiguous())
    return out


def fused_unroll_supported(wenv, policy_net, dist) -> bool:
    """The one-launch unroll needs the fused wrapper of a HIP rodent env (rollout configuration) and the default policy shape."""
    from .. import hip
    from . import fused_mlp
    base = wenv.unwrapped if hasattr(wenv, "unwrapped") else wenv
    return (isinstance(wenv, wrappers.FusedEpisodeAutoResetWrapper) and hasattr(base, "unroll_policy_wrapped") and base.device.type == "cuda"
            and not base._pipeline_outputs and not base._contact_outputs and getattr(getattr(base, "sys", None), "solver", "cg") == "cg"
            and fused_mlp.fusable(policy_net, fused_mlp.POLICY_HIDDEN, 64) and 2 <= len(policy_net.layers) <= 5
            and policy_net.layers[-1].out_features == 2 * dist.event_size and dist.event_size == base.action_size <= 32
            and base.observation_size <= 1280 and base._batch.unroll_supported(with_actor=True))


@torch.no_grad()
def generate_unroll_fused(wenv, state, actor: dict, buf: UnrollBuffer, u: int, generator=None):
    """`generate_unroll` as ONE launch (policy, sampling, env steps, wrappers and the recording of the transitions all inside the
    multi-step kernel): the envs of the batch never wait for each other between the T steps."""
    N, T = buf.N, buf.T
    noise = torch.randn(T, N, buf.raw_action.shape[-1], device=buf.obs.device, generator=generator)
    traj = dict(obs=buf.obs[u], raw_action=buf.raw_action[u], log_prob=buf.log_prob[u], reward=buf.reward[u], discount=buf.discount[u],
                truncation=buf.truncation[u])
    state, _ = wenv.unroll_policy(state, actor, noise, traj)
    return state


@torch.no_grad()
def generate_unrolls_fused(wenv, state, actor: dict, buf: UnrollBuffer, generator=None):
    """ALL `buf.U` unrolls of a training step as one launch: the policy does not change during the rollout phase, so its U x T env
    steps are one multi-step rollout recorded as U trajectories (`segment` = T).  Over U x T = 640 unsynchronised steps the envs'
    costs average out (a synchronised step lasts as long as its slowest env)."""
    U, N, T = buf.U, buf.N, buf.T
    noise = torch.randn(U * T, N, buf.raw_action.shape[-1], device=buf.obs.device, generator=generator)
    traj = dict(obs=buf.obs, raw_action=buf.raw_action, log_prob=buf.log_prob, reward=buf.reward, discount=buf.discount, truncation=buf.truncation)
    state, _ = wenv.unroll_policy(state, actor, noise, traj, segment=T)
    return state


class SubBatchRollout:
    """The rank's N envs collected as S sub-batches of N / S envs, each with its own env batch, HIP stream, wrapper state and
    sampling generator; one unroll of a sub-batch (T x [policy forward, sampling, fused env step, wrapper kernel, buffer
    writes]) is captured once as a HIP graph and replayed (`envs.graphed.GraphedSteps`).

    Why: a 2048-env step kernel lasts as long as its slowest env, and between two step kernels the GPU runs the policy's small
    launches; with two sub-batches in flight one's tail and policy launches overlap the other's step kernel (config 2 measured
    1.49 -> 1.43 ms per 2048-env step), and replay removes the ~0.7 ms of host work per env step that would otherwise serialise
    the two streams.  Every env still steps exactly once per rollout step: only the order of work on the GPU changes
    (`tests/test_gpu_env.py` holds sub-batches and replay to the one-launch, host-issued trajectories bit for bit).
    """

    def __init__(self, environment, num_sub: int, device, episode_length: int, action_repeat: int, T: int, seed: int, use_graph: bool = True):
        from ..envs import graphed
        self._graphed = graphed
        self.N = environment.num_envs
        self.S, self.n, self.T, self.device, self.use_graph = num_sub, environment.num_envs // num_sub, T, device, use_graph
        self.subs = []
        for si in range(num_sub):
            st = torch.cuda.Stream(device)
            with torch.cuda.stream(st):                                  # the batch binds the stream it is created on
                env = environment.with_num_envs(self.n, device)
                wenv = wrappers.wrap(env, episode_length=episode_length, action_repeat=action_repeat)
                gen = torch.Generator(device=device)
                gen.manual_seed(seed * 1009 + si)
                stage = UnrollBuffer(1, self.n, T, env.observation_size, env.action_size, device)
            self.subs.append(dict(stream=st, env=env, wenv=wenv, gen=gen, stage=stage, state=None, graph=None))

    def reset(self, keys):
        for si, sub in enumerate(self.subs):
            with torch.cuda.stream(sub["stream"]):
                sub["state"] = sub["wenv"].reset(keys[si * self.n:(si + 1) * self.n])
            sub["graph"] = None                                          # recaptured after the next host-issued unroll

    def unroll(self, policy: Callable, buf: UnrollBuffer, u: int):
        """One unroll of every sub-batch into buf[u] (asynchronous: `join()` before reading `buf`).  `policy` must be the SAME
        callable at every call (it is part of the captured graph): parameters and normaliser are read from fixed buffers."""
        main = torch.cuda.current_stream(self.device)
        for si, sub in enumerate(self.subs):
            sub["stream"].wait_stream(main)                              # parameter / normaliser updates issued on the caller's stream
            with torch.cuda.stream(sub["stream"]):
                if sub["graph"] is not None:
                    sub["graph"].replay()
                else:
                    step_fn = lambda st_, sub=sub: generate_unroll(sub["wenv"], st_, policy, sub["stage"], 0, sub["gen"])
                    sub["state"] = step_fn(sub["state"])
                    if self.use_graph and sub.get("warm"):               # second host-issued unroll done: capture from a step OUTPUT
                        sub["graph"] = self._graphed.GraphedSteps(step_fn, sub["state"], 1, sub["stream"], [sub["gen"]])
                        sub["state"] = sub["graph"].state
                    sub["warm"] = True
                lo, hi = si * self.n, (si + 1) * self.n
                for name in ("obs", "raw_action", "log_prob", "reward", "discount", "truncation"):
                    getattr(buf, name)[u, lo:hi].copy_(getattr(sub["stage"], name)[0])

    def join(self):
        main = torch.cuda.current_stream(self.device)
        for sub in self.subs:
            main.wait_stream(sub["stream"])


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
