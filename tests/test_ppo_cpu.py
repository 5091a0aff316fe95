"""CPU tests of the PPO host logic (losses, normaliser, wrappers, train loop, 2-rank gloo DP)."""
import math
import os

import numpy as np
import pytest
import torch

from rodent_amd.envs import wrappers
from rodent_amd.training import networks, running_statistics
from rodent_amd.training.agents.ppo import losses
from rodent_amd.training.agents.ppo import train as ppo
from tests.fake_env import PointEnv


def test_gae_matches_loop_restatement():
    """Independent numpy restatement of the upstream GAE recurrence."""
    rng = np.random.default_rng(0)
    T, B = 10, 7
    r, v = rng.normal(size=(T, B)), rng.normal(size=(T, B))
    boot = rng.normal(size=B)
    trunc = (rng.random((T, B)) < 0.15).astype(np.float64)
    done = (rng.random((T, B)) < 0.2).astype(np.float64)
    term = done * (1 - trunc)
    g, lam = 0.97, 0.95
    vs_w = np.zeros((T, B)); adv_w = np.zeros((T, B))
    for b in range(B):
        acc = 0.0
        vs_next = boot[b]
        for t in range(T - 1, -1, -1):
            v_next = boot[b] if t == T - 1 else v[t + 1, b]
            delta = (r[t, b] + g * (1 - term[t, b]) * v_next - v[t, b]) * (1 - trunc[t, b])
            acc = delta + g * (1 - term[t, b]) * (1 - trunc[t, b]) * lam * acc
            vs_w[t, b] = acc + v[t, b]
        for t in range(T):
            vs_next = boot[b] if t == T - 1 else vs_w[t + 1, b]
            adv_w[t, b] = (r[t, b] + g * (1 - term[t, b]) * vs_next - v[t, b]) * (1 - trunc[t, b])
    tt = lambda a: torch.tensor(a, dtype=torch.float64)
    vs, adv = losses.compute_gae(tt(trunc), tt(term), tt(r), tt(v), tt(boot), lambda_=lam, discount=g)
    np.testing.assert_allclose(vs.numpy(), vs_w, rtol=1e-12)
    np.testing.assert_allclose(adv.numpy(), adv_w, rtol=1e-12)


def test_tanh_normal_log_prob_and_entropy():
    d = networks.NormalTanhDistribution(3)
    logits = torch.randn(5, 6, dtype=torch.float64)
    raw = torch.randn(5, 3, dtype=torch.float64)
    loc, s = logits[:, :3], torch.nn.functional.softplus(logits[:, 3:]) + 1e-3
    base = torch.distributions.Normal(loc, s).log_prob(raw)
    want = (base - torch.log(1 - torch.tanh(raw) ** 2)).sum(-1)           # change of variables a = tanh(raw)
    torch.testing.assert_close(d.log_prob(logits, raw), want, rtol=1e-8, atol=1e-8)
    assert torch.all(d.mode(logits).abs() <= 1)


def test_running_statistics_matches_numpy():
    st = running_statistics.init_state(4, "cpu")
    rng = np.random.default_rng(1)
    chunks = [rng.normal(2.0, 3.0, size=(50, 3, 4)) for _ in range(4)]
    for c in chunks:
        st = running_statistics.update(st, torch.tensor(c, dtype=torch.float32))
    allx = np.concatenate([c.reshape(-1, 4) for c in chunks])
    np.testing.assert_allclose(st.mean.numpy(), allx.mean(0), rtol=1e-4)
    np.testing.assert_allclose(st.std.numpy(), allx.std(0), rtol=1e-3)
    assert float(st.count) == allx.shape[0]


def test_episode_and_autoreset_wrappers():
    env = wrappers.wrap(PointEnv(4), episode_length=5, action_repeat=1)
    keys = np.arange(8, dtype=np.uint32).reshape(4, 2)
    s = env.reset(keys)
    first_obs = s.obs.clone()
    a = torch.zeros(4, 2)
    for t in range(5):
        s = env.step(s, a)
    assert torch.all(s.done == 1) and torch.all(s.info["truncation"] == 1)        # time limit, not termination
    assert torch.equal(s.obs, first_obs)                                           # auto-reset selects the first state
    assert torch.all(s.info["cur_frame"] == 5)                                     # info is NOT restored (App. D-7)
    s = env.step(s, a)
    assert torch.all(s.info["steps"] == 1) and torch.all(s.done == 0)


def test_train_runs_and_learns_on_point_env():
    rewards = []
    make_policy, params, metrics = ppo.train(
        environment=PointEnv(64), num_timesteps=64 * 10 * 4 * 6, episode_length=50, num_envs=64, batch_size=64,
        num_minibatches=4, unroll_length=10, num_updates_per_batch=2, num_evals=4, num_eval_envs=16, learning_rate=3e-3,
        normalize_observations=True, seed=0, progress_fn=lambda n, m: rewards.append(m.get("eval/episode_reward")))
    assert len(rewards) == 4 and all(math.isfinite(r) for r in rewards)
    assert "training/sps" in metrics and "eval/episode_dist" in metrics
    pol = make_policy(params, deterministic=True)
    act, _ = pol(torch.zeros(3, 4))
    assert act.shape == (3, 2)


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, params, _ = ppo.train(environment=PointEnv(16), num_timesteps=32 * 5 * 2 * 2, episode_length=20, num_envs=32,
                                 batch_size=32, num_minibatches=2, unroll_length=5, num_updates_per_batch=2, num_evals=2,
                                 num_eval_envs=8, normalize_observations=True, seed=3)
        vec = torch.cat([p.detach().reshape(-1) for p in params[1].parameters()])
        out[rank] = (vec.numpy().copy(), params[0].mean.numpy().copy(), float(params[0].count))
    finally:
        torch.distributed.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    """world_size 2 over gloo: gradients and normaliser statistics are all-reduced, so both ranks end with
    identical parameters although their env shards (and reset keys, via fold_in(rank)) differ."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    (p0, m0, c0), (p1, m1, c1) = out[0], out[1]
    np.testing.assert_allclose(p0, p1, rtol=0, atol=1e-7)
    np.testing.assert_allclose(m0, m1, rtol=0, atol=1e-7)
    assert c0 == c1 == 32 * 5 * 2 * 2                     # count sums the transitions of BOTH ranks


def test_evaluator_matches_an_independent_restatement():
    """`acting.Evaluator` [UP brax.training.acting.Evaluator; SURVEY.md a27] against a by-hand evaluation loop: same keys, same
    (deterministic) policy, episode sums accumulate only while the episode is active, avg_episode_length counts active steps."""
    from rodent_amd import jax_random
    from rodent_amd.training import acting
    torch.manual_seed(0)
    nets = networks.make_ppo_networks(4, 2)
    make_policy = networks.make_inference_fn(nets)
    n, ep = 6, 30
    key = jax_random.PRNGKey(11)
    ev = acting.Evaluator(wrappers.wrap(PointEnv(n), episode_length=ep, action_repeat=1), lambda p: make_policy(p, deterministic=True), n, ep, 1, key)
    got = ev.run_evaluation((None, nets.policy_network), training_metrics={"training/x": 1.0})
    # restatement: plain env + explicit bookkeeping
    _, unroll_key = jax_random.split(key)
    env = PointEnv(n)
    s = env.reset(jax_random.split(unroll_key, n))
    pol = make_policy((None, nets.policy_network), deterministic=True)
    active = torch.ones(n); rew = torch.zeros(n); dist = torch.zeros(n); steps = torch.zeros(n)
    first = s
    for t in range(ep):
        a, _ = pol(s.obs, None)
        s = env.step(s, a)
        done = torch.maximum(s.done, torch.tensor(float(t + 1 >= ep)))
        rew += s.reward * active; dist += s.metrics["dist"] * active; steps += active
        active = active * (1 - done)
    assert abs(got["eval/episode_reward"] - float(rew.mean())) < 1e-5
    assert abs(got["eval/episode_dist"] - float(dist.mean())) < 1e-5
    assert abs(got["eval/avg_episode_length"] - float(steps.mean())) < 1e-6
    assert got["training/x"] == 1.0 and got["eval/sps"] > 0 and "eval/walltime" in got


def test_training_state_is_carried():
    out = ppo.train(environment=PointEnv(16), num_timesteps=16 * 5 * 2 * 3, episode_length=20, num_envs=16, batch_size=16,
                    num_minibatches=2, unroll_length=5, num_updates_per_batch=2, num_evals=1, num_eval_envs=0,
                    normalize_observations=True, seed=1, return_training_state=True)
    ts = out[3]
    assert ts.env_steps.dtype == torch.int64 and int(ts.env_steps) == 16 * 5 * 2 * 3
    assert float(ts.normalizer_params.count) == 16 * 5 * 2 * 3
    assert ts.params.policy is out[1][1] and len(ts.optimizer_state.state) > 0


def test_state_pytree_helpers_of_the_graph_replay():
    """envs.graphed.tree_leaves / tree_map (used to keep a state pytree in fixed buffers across HIP-graph replays): order, aliases,
    None leaves, nested containers."""
    import dataclasses
    from rodent_amd.envs import graphed

    @dataclasses.dataclass
    class S:
        a: torch.Tensor
        b: dict
        c: object = None

    x = torch.arange(3.0)
    s = S(a=x, b={"k": [x, torch.ones(2)], "n": None}, c=(torch.zeros(1),))
    leaves = graphed.tree_leaves(s)
    assert [tuple(t.shape) for t in leaves] == [(3,), (3,), (2,), (1,)] and leaves[0] is leaves[1]
    t = graphed.tree_map(lambda v: v + 1, s)
    assert isinstance(t, S) and t.b["n"] is None and isinstance(t.c, tuple) and torch.equal(t.b["k"][1], torch.full((2,), 2.0))
    assert torch.equal(graphed.tree_leaves(t)[0], x + 1)


def test_actor_params_layout_of_the_in_kernel_policy():
    """acting.actor_params: first layer as torch holds it, hidden weights transposed, head transposed and zero-padded to 64 outputs --
    the plain-tensor evaluation of that layout reproduces the policy network."""
    from rodent_amd.training import acting, networks, running_statistics
    torch.manual_seed(0)
    K, A = 57, 30
    nets = networks.make_ppo_networks(K, A)
    net = nets.policy_network
    for l in net.layers:
        l.bias.data.uniform_(-0.2, 0.2)
    norm = running_statistics.init_state(K, torch.device("cpu"))
    norm.mean.copy_(torch.randn(K) * 0.1); norm.std.copy_(torch.rand(K) + 0.5)
    a = acting.actor_params(net, norm, 0.001)
    assert a["w0"].shape == (32, K) and a["head_wt"].shape == (32, 64) and a["head_b"].shape == (64,) and len(a["hidden_wt"]) == 3
    x = torch.randn(5, K)
    h = torch.nn.functional.silu(((x - a["mean"]) / a["std"]) @ a["w0"].t() + a["b0"])
    for wt, b in zip(a["hidden_wt"], a["hidden_b"]):
        h = torch.nn.functional.silu(h @ wt + b)
    logits = (h @ a["head_wt"] + a["head_b"])[:, :2 * A]
    want = net((x - norm.mean) / norm.std)
    assert torch.allclose(logits, want, atol=1e-5) and float((h @ a["head_wt"] + a["head_b"])[:, 2 * A:].abs().max()) == 0.0


def test_running_statistics_from_sums_equals_the_upstream_form():
    """`update_from_sums` (S1, S2 of the batch about the OLD mean; what the one-pass GPU kernel rr_obs_moments delivers) against
    `update` (the upstream formulation: sum (x - mean_old)(x - mean_new)), over several successive updates."""
    rng = np.random.default_rng(3)
    a = running_statistics.init_state(5, "cpu")
    b = running_statistics.init_state(5, "cpu")
    for i in range(4):
        x = torch.tensor(rng.normal(1.5, 2.0, size=(7, 11, 5)), dtype=torch.float32)
        a = running_statistics.update(a, x)
        d = x.reshape(-1, 5).double() - b.mean.double()
        b = running_statistics.update_from_sums(b, d.shape[0], torch.stack([d.sum(0), (d * d).sum(0)]))
        for f in ("count", "mean", "summed_variance", "std"):
            torch.testing.assert_close(getattr(a, f), getattr(b, f), rtol=2e-5, atol=1e-6)
    # CPU path of the unroll-buffer entry point = the generic update on the first T rows of every sequence
    buf = torch.tensor(rng.normal(size=(3, 4, 6, 5)), dtype=torch.float32)
    c = running_statistics.update_from_unroll_buffer(a, buf, 5)
    d_ = running_statistics.update(a, buf[:, :, :5])
    torch.testing.assert_close(c.mean, d_.mean); torch.testing.assert_close(c.std, d_.std)


def test_checkpoint_callback_round_trip(tmp_path):
    """The launcher's `policy_params_fn` contract [REF brax_rodent_run_ppo.py:135-139, 204-205]: called as
    (num_steps, make_policy, params) after every evaluation with params = (normalizer_params, policy_params); `model.save_params` /
    `load_params` carry exactly what the deterministic policy needs -- the reloaded parameters reproduce the actions bit for bit --
    and the snapshot a callback received is not changed by later training steps."""
    from rodent_amd.io import model
    calls = []

    def policy_params_fn(num_steps, make_policy, params):
        path = str(tmp_path / f"{num_steps}")
        model.save_params(path, params)
        obs = torch.linspace(-1, 1, 5 * 4).reshape(5, 4)
        act, _ = make_policy(params, deterministic=True)(obs, None)
        calls.append((num_steps, path, obs, act.clone(), params))

    make_policy, params, _ = ppo.train(environment=PointEnv(16), num_timesteps=16 * 5 * 2 * 4, episode_length=20, num_envs=16, batch_size=16,
                                       num_minibatches=2, unroll_length=5, num_updates_per_batch=2, num_evals=3, num_eval_envs=8,
                                       normalize_observations=True, seed=2, policy_params_fn=policy_params_fn)
    assert len(calls) >= 2 and [c[0] for c in calls] == sorted(c[0] for c in calls) and calls[-1][0] >= 16 * 5 * 2 * 4
    for num_steps, path, obs, act, snap in calls:
        norm, sd = model.load_params(path)
        net = networks.make_ppo_networks(4, 2).policy_network
        net.load_state_dict(sd)
        got, _ = make_policy((norm, net), deterministic=True)(obs, None)
        assert torch.equal(got, act)                                       # the checkpoint reproduces the policy
        again, _ = make_policy(snap, deterministic=True)(obs, None)
        assert torch.equal(again, act)                                     # the callback's snapshot did not move with training
        assert float(norm.count) == float(snap[0].count)
    assert not torch.equal(calls[0][3], calls[-1][3])                      # training did change the policy in between
