"""GPU: one short PPO run on the real Rodent env through `ppo.train` (the launcher's entry point)."""
import math

import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def test_ppo_train_on_rodent_env():
    from rodent_amd import envs
    from rodent_amd.training.agents.ppo import train as ppo
    env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=256, xml_path="rodent_optimized.xml",
                               iterations=8, ls_iterations=8, device="cuda:0")
    log, timing = [], []
    make_policy, params, metrics = ppo.train(
        environment=env, num_timesteps=256 * 5 * 4, episode_length=150, num_envs=256, batch_size=256, num_minibatches=4,
        unroll_length=5, num_updates_per_batch=2, num_evals=2, num_eval_envs=32, learning_rate=5e-5, entropy_cost=1e-3,
        discounting=0.97, normalize_observations=True, seed=0, progress_fn=lambda n, m: log.append((n, m)),
        timing_fn=timing.append)
    assert log and log[-1][0] == 256 * 5 * 4
    m = log[-1][1]
    for k in ("eval/episode_reward", "eval/episode_pos_reward", "eval/episode_reward_alive", "training/sps",
              "training/total_loss", "eval/avg_episode_length"):
        assert k in m and math.isfinite(float(m[k])), k
    pol = make_policy(params, deterministic=True)
    act, _ = pol(torch.zeros(2, env.observation_size, device="cuda:0"))
    assert act.shape == (2, env.action_size) and torch.isfinite(act).all()
    assert timing and timing[0]["rollout_s"] > 0


def test_graph_captured_update_equals_eager(monkeypatch):
    """The HIP-graph replay of the minibatch update (default on one GPU) runs the same kernels as the eager update: same
    parameters after training, bit for bit."""
    from rodent_amd import envs
    from rodent_amd.training.agents.ppo import train as ppo
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RR_PPO_GRAPH", mode)
        env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=128, xml_path="rodent_optimized.xml",
                                   iterations=8, ls_iterations=8, device="cuda:0")
        _, params, _ = ppo.train(environment=env, num_timesteps=128 * 4 * 4 * 2, episode_length=150, num_envs=128, batch_size=128,
                                 num_minibatches=4, unroll_length=4, num_updates_per_batch=3, num_evals=1, num_eval_envs=0,
                                 learning_rate=5e-5, entropy_cost=1e-3, discounting=0.97, normalize_observations=True, seed=3)
        out[mode] = [p.detach().clone() for p in params[1].parameters()]
    assert len(out["1"]) == len(out["0"]) > 0
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a, b)


def test_gae_kernel_matches_torch_scan():
    from rodent_amd.training.agents.ppo import losses
    g = torch.Generator().manual_seed(0)
    T, B = 10, 3001
    r, v = torch.randn(T, B, generator=g), torch.randn(T, B, generator=g)
    boot = torch.randn(B, generator=g)
    trunc = (torch.rand(T, B, generator=g) < 0.1).float()
    term = (torch.rand(T, B, generator=g) < 0.2).float() * (1 - trunc)
    want_vs, want_adv = losses.compute_gae(trunc, term, r, v, boot, lambda_=0.95, discount=0.97)          # CPU: torch scan
    got_vs, got_adv = losses.compute_gae(*(x.cuda() for x in (trunc, term, r, v, boot)), lambda_=0.95, discount=0.97)
    torch.testing.assert_close(got_vs.cpu(), want_vs, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got_adv.cpu(), want_adv, rtol=1e-5, atol=1e-5)


def test_config3_full_size_training_step(monkeypatch):
    """BASELINE config 3 at its size: ONE training step of the launcher's PPO configuration [REF brax_rodent_run_ppo.py:97-114]
    at num_envs = batch_size = 2048 (64 unrolls x 10 steps = 1 310 720 env-steps, 512 minibatch updates of [11 x 2048 x 1263]
    through the fused f32-MFMA forward): finite losses, and the HIP-graph replay gives the eager run's parameters bit for bit."""
    from rodent_amd import envs
    from rodent_amd.training.agents.ppo import train as ppo
    out, losses = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RR_PPO_GRAPH", mode)
        env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=2048, xml_path="rodent_optimized.xml",
                                   terminate_when_unhealthy=True, solver="cg", iterations=8, ls_iterations=8, device="cuda:0")
        log, timing = [], []
        _, params, metrics = ppo.train(environment=env, num_timesteps=500_000_000, num_evals=100, reward_scaling=1, episode_length=150,
                                       normalize_observations=True, action_repeat=1, unroll_length=10, num_minibatches=64,
                                       num_updates_per_batch=8, discounting=0.97, learning_rate=5e-5, entropy_cost=1e-3, num_envs=2048,
                                       batch_size=2048, seed=0, num_eval_envs=0, max_training_steps=1, timing_fn=timing.append,
                                       progress_fn=lambda n, m: log.append((n, m)))
        assert timing[0]["env_steps"] == 1_310_720
        m = log[-1][1]
        for k in ("training/total_loss", "training/policy_loss", "training/v_loss", "training/entropy_loss"):
            assert math.isfinite(float(m[k])), (k, m[k])
        assert float(params[0].count) == 1_310_720            # the normaliser saw every transition of the step
        out[mode] = [p.detach().clone() for p in params[1].parameters()]
        losses[mode] = float(m["training/total_loss"])
        print(f"config 3, graph={mode}: rollout {timing[0]['rollout_s']:.2f} s, learner {timing[0]['learner_s']:.2f} s, "
              f"{timing[0]['env_steps'] / (timing[0]['rollout_s'] + timing[0]['learner_s']):.0f} env-steps/s")
    for a, b in zip(out["1"], out["0"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def _two_rank_worker(rank, world, port, out, graph):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RR_PPO_GRAPH=graph)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rodent_amd import envs
        from rodent_amd.training.agents.ppo import train as ppo
        torch.cuda.set_device(0)
        env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=64, xml_path="rodent_optimized.xml",
                                   iterations=8, ls_iterations=8, device="cuda:0")
        _, params, _ = ppo.train(environment=env, num_timesteps=128 * 4 * 4 * 3, episode_length=150, num_envs=128, batch_size=128,
                                 num_minibatches=4, unroll_length=4, num_updates_per_batch=3, num_evals=1, num_eval_envs=0,
                                 learning_rate=5e-5, entropy_cost=1e-3, discounting=0.97, normalize_observations=True, seed=3)
        vec = torch.cat([p.detach().reshape(-1) for p in params[1].parameters()]).cpu()
        out[rank] = (vec.numpy().copy(), params[0].mean.cpu().numpy().copy(), float(params[0].count))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("graph", ["1", "0"])
def test_two_ranks_share_the_gpu_through_the_hip_env(graph):
    """Rehearsal of the multi-GPU learner on one GPU: two ranks (gloo; RCCL refuses two ranks on one device) each step 64 real HIP
    envs; gradients and normaliser statistics are all-reduced, so both end with IDENTICAL parameters.  graph = 1: the
    two-graph learner (capture A: gather .. backward | all-reduce | capture B: Adam) that multi-rank runs use."""
    import os
    import numpy as np
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() + int(graph)) % 2000
    mp.spawn(_two_rank_worker, args=(2, port, out, graph), nprocs=2, join=True)
    (p0, m0, c0), (p1, m1, c1) = out[0], out[1]
    assert np.isfinite(p0).all()
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(m0, m1)
    assert c0 == c1 == 128 * 4 * 4 * 3


def test_sub_batch_rollouts_replayed_equal_host_issued(monkeypatch):
    """ppo.train with the rollouts as two sub-batches on two streams (acting.SubBatchRollout): replaying each unroll from a HIP
    graph gives the parameters of the host-issued run, bit for bit."""
    from rodent_amd import envs
    from rodent_amd.training.agents.ppo import train as ppo
    monkeypatch.setenv("RR_ROLLOUT_SUBSTREAMS", "2")
    monkeypatch.setenv("RR_ROLLOUT_SUBSTREAMS_MIN_ENVS", "32")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RR_ROLLOUT_GRAPH", mode)
        env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=128, xml_path="rodent_optimized.xml",
                                   iterations=8, ls_iterations=8, device="cuda:0")
        log = []
        _, params, _ = ppo.train(environment=env, num_timesteps=128 * 4 * 4 * 3, episode_length=150, num_envs=128, batch_size=128,
                                 num_minibatches=4, unroll_length=4, num_updates_per_batch=2, num_evals=1, num_eval_envs=0,
                                 learning_rate=5e-5, entropy_cost=1e-3, discounting=0.97, normalize_observations=True, seed=3,
                                 progress_fn=lambda n, m: log.append(m))
        assert math.isfinite(float(log[-1]["training/total_loss"]))
        out[mode] = ([p.detach().clone() for p in params[1].parameters()], params[0].mean.clone(), float(params[0].count))
    assert out["1"][2] == out["0"][2] == 128 * 4 * 4 * 3
    assert torch.equal(out["1"][1], out["0"][1])                                     # same observations seen
    for a, b in zip(out["1"][0], out["0"][0]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_autograd_free_learner_trains_like_the_autograd_path(monkeypatch):
    """End to end: ONE training step of ppo.train (one rollout, 8 minibatch updates through graph capture, flat gradients, fused
    Adam) with the hand-written update (rr_ppo_loss + explicit backward, default) and with compute_ppo_loss + loss.backward()
    (RR_FUSED_LOSS=0), same seeds, same rollout.  Gradients agree to float32 rounding (tests/test_gpu_ppo_loss.py) but Adam turns
    even a rounding-sized gradient into a step of ~lr, so parameters are compared as DISPLACEMENTS from the common initial point
    (a third run with lr = 0): same direction (cosine), bounded difference, same losses on the last minibatch."""
    from rodent_amd import envs
    from rodent_amd.training.agents.ppo import train as ppo
    lr, out, loss = 5e-5, {}, {}
    for mode, rate in (("init", 0.0), ("1", lr), ("0", lr)):
        monkeypatch.setenv("RR_FUSED_LOSS", "1" if mode == "init" else mode)
        env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=128, xml_path="rodent_optimized.xml",
                                   iterations=8, ls_iterations=8, device="cuda:0")
        log = []
        _, params, _ = ppo.train(environment=env, num_timesteps=10 ** 9, episode_length=150, num_envs=128, batch_size=128,
                                 num_minibatches=4, unroll_length=4, num_updates_per_batch=2, num_evals=2, num_eval_envs=0,
                                 learning_rate=rate, entropy_cost=1e-3, discounting=0.97, normalize_observations=True, seed=5,
                                 max_training_steps=1, progress_fn=lambda n, m: log.append(m))
        out[mode] = torch.cat([p.detach().reshape(-1) for p in params[1].parameters()]).double()
        loss[mode] = {k: float(v) for k, v in log[-1].items() if k.startswith("training/") and k.endswith("loss")}
    k = 4 * 2                                                # minibatches x epochs
    d1, d0 = out["1"] - out["init"], out["0"] - out["init"]
    cos = float((d1 * d0).sum() / (d1.norm() * d0.norm()))
    diff = (out["1"] - out["0"]).abs()
    print(f"displacement cosine {cos:.4f}; |displacement| {float(d1.norm()):.3e} / {float(d0.norm()):.3e}; max |param diff| "
          f"{float(diff.max()) / lr:.2f} lr; losses {loss['1']} vs {loss['0']}")
    assert float(d0.norm()) > 0 and cos > 0.9
    assert float(diff.max()) <= 2 * k * lr
    for name in loss["0"]:
        assert abs(loss["1"][name] - loss["0"][name]) <= 5e-3 * max(abs(loss["0"][name]), 1e-2), name


def test_one_launch_unroll_with_the_actor_inside():
    """rr_env_unroll_policy (policy MLP, sampling, T wrapped env steps and the recording of the transitions in one launch):
    (1) replaying the recorded actions through the per-step path reproduces every recorded observation, discount, truncation and
    the final state bit for bit (the reward to one ulp); (2) the recorded raw actions / log-probs are those of the two-launch actor
    (rr_policy_act) on the recorded observations with the same noise, to float32 rounding."""
    from rodent_amd import envs, hip, jax_random
    from rodent_amd.envs import graphed, wrappers
    from rodent_amd.training import acting, fused_mlp, networks, running_statistics
    dev = torch.device("cuda:0")
    N, T = 64, 9
    torch.manual_seed(3)

    def make():
        env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=N, xml_path="rodent_optimized.xml", iterations=8,
                                   ls_iterations=8, device=dev, healthy_z_range=(0.045, 0.5))
        wenv = wrappers.wrap(env, episode_length=5, action_repeat=1)
        return env, wenv, wenv.reset(jax_random.split(jax_random.PRNGKey(2), N))
    env, wenv, st0 = make()
    nets = networks.make_ppo_networks(env.observation_size, env.action_size, device=dev)
    net, dist = nets.policy_network, nets.parametric_action_distribution
    for l in net.layers:
        l.bias.data.uniform_(-0.3, 0.3)
    norm = running_statistics.init_state(env.observation_size, dev)
    norm.mean.copy_(torch.randn(env.observation_size, device=dev) * 0.05)
    norm.std.copy_(torch.rand(env.observation_size, device=dev) + 0.7)
    assert acting.fused_unroll_supported(wenv, net, dist)
    buf = acting.UnrollBuffer(2, N, T, env.observation_size, env.action_size, dev)
    actor = acting.actor_params(net, norm, dist.min_std)
    noise = torch.randn(T, N, env.action_size, device=dev)
    traj = dict(obs=buf.obs[1], raw_action=buf.raw_action[1], log_prob=buf.log_prob[1], reward=buf.reward[1], discount=buf.discount[1],
                truncation=buf.truncation[1])
    got, actions = wenv.unroll_policy(st0, actor, noise, traj)
    torch.cuda.synchronize()
    assert torch.isfinite(buf.obs[1]).all() and torch.isfinite(buf.log_prob[1]).all()
    # the same T steps recorded as three trajectories of three steps (a whole rollout phase in one launch): same transitions
    env3, wenv3, st3 = make()
    buf3 = acting.UnrollBuffer(3, N, 3, env.observation_size, env.action_size, dev)
    traj3 = dict(obs=buf3.obs, raw_action=buf3.raw_action, log_prob=buf3.log_prob, reward=buf3.reward, discount=buf3.discount, truncation=buf3.truncation)
    got3, actions3 = wenv3.unroll_policy(st3, actor, noise, traj3, segment=3)
    torch.cuda.synchronize()
    assert torch.equal(actions3, actions) and torch.equal(got3.obs, got.obs) and torch.equal(got3.pipeline_state.qpos, got.pipeline_state.qpos)
    for u in range(3):
        assert torch.equal(buf3.obs[u], buf.obs[1, :, 3 * u:3 * u + 4]) and torch.equal(buf3.raw_action[u], buf.raw_action[1, :, 3 * u:3 * u + 3])
        for name in ("log_prob", "reward", "discount", "truncation"):
            assert torch.equal(getattr(buf3, name)[u], getattr(buf, name)[1, :, 3 * u:3 * u + 3]), name
    # (1) physics + wrappers: replay the recorded actions step by step
    env2, wenv2, st = make()
    for t in range(T):
        assert torch.equal(buf.obs[1, :, t], st.obs), t
        st = wenv2.step(st, actions[t])
        # the reward's exp() is expanded differently inside the actor instance: the last bit of pos_reward may differ
        assert (buf.reward[1, :, t] - st.reward).abs().max() <= 2.5e-7 and torch.equal(buf.discount[1, :, t], 1 - st.done), t
        assert torch.equal(buf.truncation[1, :, t], st.info["truncation"]), t
    assert torch.equal(buf.obs[1, :, T], st.obs)
    assert float(buf.truncation[1].sum()) > 0                          # episodes of 5 steps: the reset path ran
    la, lb = graphed.tree_leaves(got), graphed.tree_leaves(st)
    assert len(la) == len(lb)
    nexact = 0
    for x, y in zip(la, lb):
        assert x.shape == y.shape
        if not torch.equal(x, y):                # reward and its pos_reward metric: one ulp (see above); everything else exact
            assert x.dim() == 1 and (x - y).abs().max() <= 2.5e-7
            nexact += 1
    assert nexact <= 2
    assert torch.equal(got.pipeline_state.qpos, st.pipeline_state.qpos) and torch.equal(got.obs, st.obs) and torch.equal(got.done, st.done)
    # (2) the actor: same noise through the two-launch actor on the recorded observations
    obs_t = buf.obs[1, :, :T].transpose(0, 1).reshape(T * N, -1).contiguous()
    act, raw, lp, _ = hip.policy_act(obs_t, norm.mean, norm.std, fused_mlp.net_params(net), noise.reshape(T * N, -1).contiguous(), dist.min_std)
    raw_k = buf.raw_action[1].transpose(0, 1).reshape(T * N, -1)
    lp_k = buf.log_prob[1].transpose(0, 1).reshape(-1)
    assert (raw_k - raw).abs().max() <= 2e-5 * max(1.0, float(raw.abs().max()))
    assert (actions.reshape(T * N, -1) - act).abs().max() <= 2e-5
    assert (lp_k - lp).abs().max() <= 2e-3


def test_fused_rollout_on_a_model_without_the_fixed_dimension_instance():
    """ppo.train's one-launch unrolls on rodent_new.xml (other contact / dof counts than rodent_optimized: the generic-dimension
    instance of the multi-step kernel) -- the configuration examples/rodent_run_ppo.py runs."""
    from rodent_amd import envs
    from rodent_amd.envs import wrappers
    from rodent_amd.training import acting
    from rodent_amd.training.agents.ppo import train as ppo
    env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=64, xml_path="rodent_new.xml", iterations=8,
                               ls_iterations=8, device="cuda:0")
    assert env._batch.unroll_supported(with_actor=True) and env._batch.unroll_supported(with_actor=False)
    log = []
    _, params, _ = ppo.train(environment=env, num_timesteps=10 ** 9, episode_length=150, num_envs=64, batch_size=64, num_minibatches=4,
                             unroll_length=5, num_updates_per_batch=2, num_evals=2, num_eval_envs=0, learning_rate=5e-5, entropy_cost=1e-3,
                             discounting=0.97, normalize_observations=True, seed=1, max_training_steps=2, progress_fn=lambda n, m: log.append(m))
    assert math.isfinite(float(log[-1]["training/total_loss"])) and float(params[0].count) == 64 * 4 * 5 * 2
