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
