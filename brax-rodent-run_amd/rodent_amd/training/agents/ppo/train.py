"""Proximal policy optimization training: the `brax.training.agents.ppo.train.train` entry point the
reference launcher calls [REF brax_rodent_run_ppo.py:97-114,200-202], with the same keyword
arguments, schedule arithmetic, callbacks and return value -- on PyTorch-ROCm.

Data parallelism: one process per GPU (launch with torch.distributed.run); `num_envs` / `batch_size`
are GLOBAL as upstream, each rank steps `num_envs // world_size` environments; the only collectives
are the per-minibatch gradient all-reduce(mean) on one flat buffer and the normaliser all-reduce(sum)
once per training step (RCCL over xGMI; gloo in the CPU tests).
"""
from __future__ import annotations

import copy
import math
import logging
import os
import time
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from .... import jax_random
from ....envs import wrappers
from ... import acting, distributed as D, fused_mlp, networks as ppo_networks_mod, running_statistics
from ...types import PPONetworkParams, TrainingState
from . import losses as ppo_losses


def _local_env(environment, local_num_envs: int, device):
    if getattr(environment, "num_envs", None) == local_num_envs:
        return environment
    if hasattr(environment, "with_num_envs"):
        return environment.with_num_envs(local_num_envs, device)
    raise ValueError(f"environment has {getattr(environment, 'num_envs', '?')} envs, need {local_num_envs} per rank")


def train(
    environment,
    num_timesteps: int,
    episode_length: int,
    action_repeat: int = 1,
    num_envs: int = 1,
    max_devices_per_host: Optional[int] = None,
    num_eval_envs: int = 128,
    learning_rate: float = 1e-4,
    entropy_cost: float = 1e-4,
    discounting: float = 0.9,
    seed: int = 0,
    unroll_length: int = 10,
    batch_size: int = 32,
    num_minibatches: int = 16,
    num_updates_per_batch: int = 2,
    num_evals: int = 1,
    num_resets_per_eval: int = 0,
    normalize_observations: bool = False,
    reward_scaling: float = 1.0,
    clipping_epsilon: float = 0.3,
    gae_lambda: float = 0.95,
    deterministic_eval: bool = False,
    network_factory: Callable = ppo_networks_mod.make_ppo_networks,
    progress_fn: Callable = lambda *args: None,
    normalize_advantage: bool = True,
    global_advantage_normalization: bool = False,
    eval_env=None,
    policy_params_fn: Callable = lambda *args: None,
    randomization_fn=None,
    max_training_steps: Optional[int] = None,
    timing_fn: Optional[Callable] = None,
    return_training_state: bool = False,
):
    """PPO training.  Returns (make_policy, params=(normalizer_params, policy_params), metrics).

    `max_training_steps` (extension) stops after that many training steps (benchmarks / tests).
    `return_training_state` (extension): also return the final `TrainingState` (optimizer, params, normaliser, env_steps).
    `global_advantage_normalization` (extension, default off = the reference's behaviour: advantages are normalised over
    the rank-local minibatch): normalise with the mean / variance of the GLOBAL minibatch instead, one
    3-float all-reduce {n, sum a, sum a^2} per minibatch (the north star's "advantage-normalisation all-reduce").
    """
    assert batch_size * num_minibatches % num_envs == 0
    if randomization_fn is not None:
        raise NotImplementedError("domain randomisation is not used by the reference launcher")
    xt = time.time()
    process_count, process_id = D.world_size(), D.rank()
    device_count = process_count                         # one device per process
    if num_envs % device_count:
        raise ValueError("num_envs must be divisible by the number of ranks")
    local_num_envs = num_envs // device_count
    env_step_per_training_step = batch_size * unroll_length * num_minibatches * action_repeat
    num_evals_after_init = max(num_evals - 1, 1)
    num_training_steps_per_epoch = int(np.ceil(
        num_timesteps / (num_evals_after_init * env_step_per_training_step * max(num_resets_per_eval, 1))))

    env = _local_env(environment, local_num_envs, getattr(environment, "device", None))
    device = env.device
    wenv = wrappers.wrap(env, episode_length=episode_length, action_repeat=action_repeat)

    # ---- keys [UP ppo.train]: reset keys follow the jax key tree so initial states match the reference
    key = jax_random.PRNGKey(seed)
    global_key, local_key = jax_random.split(key)
    local_key = jax_random.fold_in(local_key, process_id)
    local_key, key_env, eval_key = jax_random.split(local_key, 3)
    key_envs = jax_random.split(key_env, local_num_envs)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(local_key[0]) * 2 ** 32 + int(local_key[1]))
    torch.manual_seed(int(global_key[0]) * 2 ** 32 + int(global_key[1]))     # identical parameter init on every rank
    perm_gen = torch.Generator(device="cpu")
    perm_gen.manual_seed(int(local_key[1]))

    # Optional (RR_ROLLOUT_SUBSTREAMS=2): the rollouts as S sub-batches on S streams, each unroll replayed from a HIP graph
    # (acting.SubBatchRollout).  Off by default: it pays for the bare env step (bench config 2: 1.49 -> 1.43 ms) but not here --
    # measured 1.22 s against 1.05 s of rollout per training step, because the 256-VGPR policy forward of one sub-batch cannot
    # be placed while the other sub-batch's step kernel fills the register files, so the two streams end up taking turns.
    n_sub_streams = int(os.environ.get("RR_ROLLOUT_SUBSTREAMS", "1"))
    sub_rollout = None
    if (device.type == "cuda" and n_sub_streams > 1 and hasattr(env, "with_num_envs") and local_num_envs % n_sub_streams == 0
            and local_num_envs // n_sub_streams >= int(os.environ.get("RR_ROLLOUT_SUBSTREAMS_MIN_ENVS", "512"))):
        sub_rollout = acting.SubBatchRollout(env, n_sub_streams, device, episode_length, action_repeat, unroll_length,
                                             seed=int(local_key[1]) % (2 ** 31), use_graph=os.environ.get("RR_ROLLOUT_GRAPH", "1") == "1")
        sub_rollout.reset(key_envs)
        env_state = None
    else:
        env_state = wenv.reset(key_envs)

    ppo_network = network_factory(env.observation_size, env.action_size, device=device)
    dist = ppo_network.parametric_action_distribution
    policy_net, value_net = ppo_network.policy_network, ppo_network.value_network
    params = list(policy_net.parameters()) + list(value_net.parameters())
    if process_count > 1:                                # replicate (rank 0's init), like device_put_replicated
        for p in params:
            torch.distributed.broadcast(p.data, src=0)
    flat = D.FlatGrads(params)
    # optax.adam(lr): b1=.9, b2=.999, eps=1e-8, eps outside the sqrt -- torch.optim.Adam's update; one fused
    # multi-tensor launch per step on the GPU
    # One process on a GPU: the minibatch update (gather, normalise, both MLPs forward + backward, GAE kernel, fused Adam) is
    # captured once in a HIP graph and replayed -- ~70 small launches per update otherwise leave the GPU idle between them
    # (half of the learner's wall time at the launcher's sizes).  RR_PPO_GRAPH=0 keeps the eager path.
    # Several ranks: the same capture split in two around the gradient all-reduce -- graph A (gather .. backward), the RCCL
    # all-reduce of the flat gradient buffer issued eagerly, graph B (fused Adam) -- so a multi-GPU run keeps the replayed
    # learner instead of falling back to ~70 eager launches per minibatch.  (The optional global advantage normalisation
    # puts a collective inside the loss: that mode runs eagerly.)
    use_graph = (device.type == "cuda" and os.environ.get("RR_PPO_GRAPH", "1") == "1"
                 and not (global_advantage_normalization and process_count > 1))
    optimizer = torch.optim.Adam(params, lr=learning_rate, betas=(0.9, 0.999), eps=1e-8, fused=(device.type == "cuda"),
                                 capturable=use_graph)
    normalizer_params = running_statistics.init_state(env.observation_size, device)
    normalize = running_statistics.normalize if normalize_observations else (lambda x, y: x)
    make_policy = ppo_networks_mod.make_inference_fn(ppo_network)
    training_state = TrainingState(optimizer_state=optimizer, params=PPONetworkParams(policy=policy_net, value=value_net),
                                   normalizer_params=normalizer_params, env_steps=torch.zeros((), dtype=torch.int64))

    def current_params():
        return (normalizer_params if normalize_observations else None, policy_net)

    if eval_env is None and num_evals > 0 and num_eval_envs > 0:
        eval_env = _local_env(environment, num_eval_envs, device) if hasattr(environment, "with_num_envs") or \
            getattr(environment, "num_envs", None) == num_eval_envs else None
    evaluator = None
    if eval_env is not None:
        weval = wrappers.wrap(eval_env, episode_length=episode_length, action_repeat=action_repeat)
        evaluator = acting.Evaluator(weval, lambda p: make_policy(p, deterministic=deterministic_eval), num_eval_envs,
                                     episode_length, action_repeat, eval_key)

    U = batch_size * num_minibatches // num_envs
    N, T = local_num_envs, unroll_length
    buf = acting.UnrollBuffer(U, N, T, env.observation_size, env.action_size, device)
    local_batch = U * N // num_minibatches               # trajectories per rank per minibatch

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    rollout_policy = {"fn": None, "norm": None}
    # rollouts as one launch per unroll (rr_env_unroll_policy) where the env / policy shapes allow; RR_FUSED_ROLLOUT=0 keeps per-step launches
    fused_rollout = (sub_rollout is None and action_repeat == 1 and os.environ.get("RR_FUSED_ROLLOUT", "1") == "1"
                     and acting.fused_unroll_supported(wenv, policy_net, dist))
    gstate = {"graph": None, "graph_b": None, "calls": 0, "idx": None, "norm": None, "metrics": None, "failed": False}

    def adv_stats(adv):
        """mean / std of the advantages over the GLOBAL minibatch (all ranks): one all-reduce of {n, sum, sum of squares}."""
        st = torch.stack([torch.tensor(float(adv.numel()), device=adv.device), adv.sum(), (adv * adv).sum()])
        D.all_reduce_sum_(st)
        mean = st[1] / st[0]
        return mean, torch.sqrt(torch.clamp(st[2] / st[0] - mean * mean, min=0.0))

    # forward of both networks on the hand-written f32-MFMA kernel (one launch, observation tile read once for both nets,
    # normalisation fused) with an explicit backward; nn.Linear path for other shapes / CPU (RR_FUSED_MLP=0 forces it)
    use_fused = (device.type == "cuda" and os.environ.get("RR_FUSED_MLP", "1") == "1"
                 and fused_mlp.fusable(policy_net, fused_mlp.POLICY_HIDDEN, 64) and fused_mlp.fusable(value_net, fused_mlp.VALUE_HIDDEN, 1))

    # ... and the loss half + backward without an autograd graph (`fused_update`: rr_ppo_loss, gradients written straight into the
    # flat buffer).  RR_FUSED_LOSS=0 keeps compute_ppo_loss + loss.backward() on the fused forward.
    fused_update_fn = None
    if use_fused and os.environ.get("RR_FUSED_LOSS", "1") == "1" and not (global_advantage_normalization and process_count > 1):
        from . import fused_update
        fused_update_fn = fused_update.FusedUpdate(policy_net, value_net, dist, T, entropy_cost=entropy_cost, discounting=discounting,
                                                   reward_scaling=reward_scaling, gae_lambda=gae_lambda, clipping_epsilon=clipping_epsilon,
                                                   normalize_advantage=normalize_advantage)

    def fwd_bwd(data, idx, nparams):
        if fused_update_fn is not None:
            mean, std = (nparams.mean, nparams.std) if normalize_observations else (None, None)
            return fused_update_fn(data, idx, mean, std, gen)
        mbd = {k: data[k][idx].transpose(0, 1) for k in ("raw_action", "log_prob", "reward", "discount", "truncation")}
        if use_fused:
            raw = data["obs"][idx].transpose(0, 1)                         # [T+1, B, obs], time-major gather
            B_ = raw.shape[1]
            mean, std = (nparams.mean, nparams.std) if normalize_observations else (None, None)
            logits_all, values_all = fused_mlp.actor_critic(raw.reshape((T + 1) * B_, -1), mean, std, policy_net, value_net)
            policy_logits = logits_all[:T * B_].reshape(T, B_, -1)        # the bootstrap row's logits carry no gradient
            values = values_all.reshape(T + 1, B_)
        else:
            obs = normalize(data["obs"][idx].transpose(0, 1), nparams)     # [T+1, B, obs]
            policy_logits = policy_net(obs[:T])
            values = value_net(obs).squeeze(-1)
        loss, m = ppo_losses.compute_ppo_loss(
            policy_logits, values[:T], values[T], mbd, dist, entropy_cost=entropy_cost, discounting=discounting,
            reward_scaling=reward_scaling, gae_lambda=gae_lambda, clipping_epsilon=clipping_epsilon,
            normalize_advantage=normalize_advantage, generator=gen,
            advantage_stats_fn=adv_stats if (global_advantage_normalization and process_count > 1) else None)
        flat.zero_()
        loss.backward()
        return m

    def eager_update(data, idx, nparams):
        m = fwd_bwd(data, idx, nparams)
        flat.pmean_()                            # jax.lax.pmean(grads, 'i')
        optimizer.step()
        return m

    def minibatch_update(data, idx):
        if not use_graph or gstate["failed"]:
            return eager_update(data, idx, normalizer_params)
        gstate["calls"] += 1
        if gstate["graph"] is None:
            if gstate["calls"] <= 3:             # first updates run eagerly: they create the Adam state and the GEMM workspaces
                return eager_update(data, idx, normalizer_params)
            try:                                 # 4th update: capture it, then run it by replaying the capture
                gstate["idx"] = idx.clone()
                gstate["norm"] = normalizer_params.clone() if normalize_observations else None
                g = torch.cuda.CUDAGraph()
                if hasattr(g, "register_generator_state"):
                    g.register_generator_state(gen)
                torch.cuda.synchronize(device)
                npar = gstate["norm"] if normalize_observations else normalizer_params
                if process_count == 1:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        gstate["metrics"] = eager_update(data, gstate["idx"], npar)
                else:                            # capture records, it does not run: both halves are replayed below
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        gstate["metrics"] = fwd_bwd(data, gstate["idx"], npar)
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                        optimizer.step()
                    gstate["graph_b"] = gb
                gstate["graph"] = g
            except Exception as e:               # keep training on the eager path
                gstate["failed"] = True
                gstate["graph"] = gstate["graph_b"] = None
                logging.warning("PPO update graph capture failed (%s); continuing eagerly", e)
                return eager_update(data, idx, normalizer_params)
        gstate["idx"].copy_(idx)
        gstate["graph"].replay()
        if gstate["graph_b"] is not None:
            flat.pmean_()
            gstate["graph_b"].replay()
        return gstate["metrics"]

    D.time_collectives(timing_fn is not None)

    def training_step():
        nonlocal env_state, normalizer_params
        t0 = time.time()
        if sub_rollout is not None:
            if rollout_policy["fn"] is None:             # ONE policy object for all training steps: the captured graphs hold it;
                rollout_policy["norm"] = normalizer_params.clone() if normalize_observations else None     # it reads these buffers
                rollout_policy["fn"] = make_policy((rollout_policy["norm"], policy_net))
            elif normalize_observations:
                for f in ("count", "mean", "summed_variance", "std"):
                    getattr(rollout_policy["norm"], f).copy_(getattr(normalizer_params, f))
            for u in range(U):
                sub_rollout.unroll(rollout_policy["fn"], buf, u)
            sub_rollout.join()
        elif fused_rollout:                              # generate_unroll as ONE launch per unroll: the actor runs inside the kernel
            actor = acting.actor_params(policy_net, normalizer_params if normalize_observations else None, dist.min_std)
            if os.environ.get("RR_FUSED_ROLLOUT_PHASE", "1") == "1":     # the whole rollout phase (U unrolls, same policy) as ONE launch
                env_state = acting.generate_unrolls_fused(wenv, env_state, actor, buf, gen)
            else:
                for u in range(U):
                    env_state = acting.generate_unroll_fused(wenv, env_state, actor, buf, u, gen)
        else:
            policy = make_policy(current_params())
            for u in range(U):
                env_state = acting.generate_unroll(wenv, env_state, policy, buf, u, gen)
        sync()
        t1 = time.time()
        data = buf.flat()
        if normalize_observations:                       # update on `observation` (not next_observation), all ranks
            normalizer_params = running_statistics.update_from_unroll_buffer(normalizer_params, buf.obs, T)
        metrics = {}
        if normalize_observations and gstate["norm"] is not None:     # the graph reads the normaliser from fixed buffers
            for f in ("count", "mean", "summed_variance", "std"):
                getattr(gstate["norm"], f).copy_(getattr(normalizer_params, f))
        for _ in range(num_updates_per_batch):
            perm = torch.randperm(U * N, generator=perm_gen).to(device)
            for mb in range(num_minibatches):
                idx = perm[mb * local_batch:(mb + 1) * local_batch]
                metrics = minibatch_update(data, idx)
        sync()
        t2 = time.time()
        training_state.normalizer_params = normalizer_params
        training_state.env_steps += env_step_per_training_step          # int64 (upstream: int32, which wraps past 2.1e9 steps)
        if timing_fn is not None:
            # allreduce_s: the part of learner_s spent in the gradient / normaliser all-reduces as the learner's stream sees them
            # (0 on one rank); SURVEY.md 8(d): rollout / learner / all-reduce
            timing_fn({"rollout_s": t1 - t0, "learner_s": t2 - t1, "allreduce_s": D.pop_collective_seconds(),
                       "env_steps": env_step_per_training_step})
        return {f"training/{k}": float(v) for k, v in metrics.items()}

    metrics = {}
    if process_id == 0 and evaluator is not None and num_evals > 1:
        metrics = evaluator.run_evaluation(current_params(), training_metrics={})
        progress_fn(0, metrics)

    training_metrics = {}
    training_walltime = 0.0
    current_step = 0
    steps_done = 0
    stop = False
    for it in range(num_evals_after_init):
        for _ in range(max(num_resets_per_eval, 1)):
            t = time.time()
            for _ in range(num_training_steps_per_epoch):
                training_metrics = training_step()
                steps_done += 1
                if max_training_steps is not None and steps_done >= max_training_steps:
                    stop = True
                    break
            epoch_training_time = time.time() - t
            training_walltime += epoch_training_time
            done_steps = steps_done if stop else num_training_steps_per_epoch
            sps = (min(done_steps, num_training_steps_per_epoch) * env_step_per_training_step) / epoch_training_time
            training_metrics = {"training/sps": sps, "training/walltime": training_walltime, **training_metrics}
            current_step = steps_done * env_step_per_training_step
            if num_resets_per_eval > 0 and not stop:
                local_key, key_env = jax_random.split(local_key)
                if sub_rollout is not None:
                    sub_rollout.reset(jax_random.split(key_env, local_num_envs))
                else:
                    env_state = wenv.reset(jax_random.split(key_env, local_num_envs))
            if stop:
                break
        if process_id == 0:
            if evaluator is not None:
                metrics = evaluator.run_evaluation(current_params(), training_metrics)
            else:
                metrics = dict(training_metrics)
            progress_fn(current_step, metrics)
            # callbacks get a SNAPSHOT (the live network keeps training)
            policy_params_fn(current_step, make_policy, params_tuple(normalizer_params.clone(), copy.deepcopy(policy_net).requires_grad_(False),
                                                                     normalize_observations))
        if stop:
            break

    total_steps = current_step
    if max_training_steps is None:
        assert total_steps >= num_timesteps
    if process_count > 1:                                # pmap.assert_is_replicated / synchronize_hosts
        chk = torch.stack([p.detach().float().sum() for p in params]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert torch.allclose(lo, hi, rtol=1e-5, atol=1e-6), "parameters diverged across ranks"
    out = (make_policy, params_tuple(normalizer_params, policy_net, normalize_observations), metrics)
    return out + (training_state,) if return_training_state else out


def params_tuple(normalizer_params, policy_net, normalize_observations: bool):
    """(normalizer_params, policy_params) as the reference's callbacks receive them."""
    return (normalizer_params if normalize_observations else None, policy_net)
