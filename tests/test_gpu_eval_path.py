"""The evaluation side of the launcher on the HIP path, held to the oracle env (tests/oracle_env.py, parity.OracleEnvImpl) by teacher
forcing: every recorded (state, action) of the HIP rollout is re-stepped by the float64 oracle and compared by the criteria of
tests/parity.py (C1 integers exact; C3 error quantiles <= 3x the scalar float32 oracle's).

  * `rollout.eval_rollout` = the reference's `policy_params_fn` rollout [REF brax_rodent_run_ppo.py:135-151]: jit_reset / jit_step of a
    single env under the deterministic policy;
  * `acting.Evaluator` = brax.training.acting.Evaluator [UP; SURVEY.md a27]: its eval/episode_* values against the same bookkeeping
    (EvalWrapper over Episode + AutoReset) restated in numpy over the ORACLE's per-step rewards / metrics / done.
"""
import numpy as np
import pytest
import torch

from tests import parity, util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _state_of(s):
    return {k: _np(getattr(s.pipeline_state, k)) for k in parity.STATE}


def _nets(env, seed=0):
    from rodent_amd.training import networks
    torch.manual_seed(seed)
    nets = networks.make_ppo_networks(env.observation_size, env.action_size, device=DEV)
    return nets, networks.make_inference_fn(nets)


def test_eval_rollout_teacher_forced_against_the_oracle(oracle_built):
    from rodent_amd import envs, jax_random, rollout
    from tests.oracle_env import OracleRodent
    track = util.synthetic_track()
    env = envs.get_environment("rodent", track_pos=track, num_envs=1, xml_path="rodent_optimized.xml", iterations=8, ls_iterations=8, device=DEV)
    nets, make_policy = _nets(env)
    rec = []
    T = 60
    qposes = rollout.eval_rollout(env, make_policy, (None, nets.policy_network), steps=T, seed=3, record=rec)
    assert qposes.shape == (T + 1, 74) and len(rec) == T
    # the launcher's key chain: key = PRNGKey(seed); _, key = split(key); reset_rng, act_rng = split(key); env.reset(reset_rng)
    key = jax_random.split(jax_random.PRNGKey(3))[1]
    reset_rng = jax_random.split(key)[0]
    O = OracleRodent("rodent_optimized", 1, "f64", (8, 8), track)
    O.reset(reset_rng[None])
    np.testing.assert_array_equal(qposes[0], O.state()["qpos"][0].astype(np.float32))
    assert int(rec[0][0].info["cur_frame"][0]) == int(O.cur_frame[0])
    # the deterministic policy: action = tanh(loc) of the policy network on the recorded observation (float64 restatement)
    pol64 = [(l.weight.detach().double().cpu().numpy(), l.bias.detach().double().cpu().numpy()) for l in nets.policy_network.layers]
    A = parity.OracleEnvImpl("rodent_optimized", 1, "f64", (8, 8), track)
    B = parity.OracleEnvImpl("rodent_optimized", 1, "f32", (8, 8), track)
    tab = util_tables()
    seg = parity.obs_segments(tab)
    names = ["qpos", "qvel", "reward"] + ["obs_" + k for k in ("cinert", "cvel", "qfrc_actuator", "track_local")]
    err, gap = {k: [] for k in names}, {k: [] for k in names}
    act_err = 0.0
    for prev, ctrl, nxt in rec:
        x = _np(prev.obs)
        for i, (W, b) in enumerate(pol64):
            x = x @ W.T + b
            if i < len(pol64) - 1:
                x = x / (1 + np.exp(-x))
        act_err = max(act_err, float(np.abs(np.tanh(x[:, :30]) - _np(ctrl)).max()))
        st, a, cf = _state_of(prev), _np(ctrl), prev.info["cur_frame"].cpu().numpy()
        want, gp = A.env_step(st, a, cf), B.env_step(st, a, cf)
        assert np.array_equal(nxt.info["cur_frame"].cpu().numpy(), want["cur_frame"])
        z = want["qpos"][:, 2]
        if not ((abs(z - 0.03) < 1e-3) | (abs(z - 0.5) < 1e-3)).any():
            assert np.array_equal(_np(nxt.done), want["done"])
        got = dict(_state_of(nxt), obs=_np(nxt.obs), reward=_np(nxt.reward))
        for k in names:
            if k.startswith("obs_"):
                s = seg[k[4:]]
                sc = np.maximum(np.abs(want["obs"][:, s]).max(1), 1e-3) if k == "obs_cinert" else 1.0
                err[k].append(np.abs(got["obs"][:, s] - want["obs"][:, s]).max(1) / sc); gap[k].append(np.abs(gp["obs"][:, s] - want["obs"][:, s]).max(1) / sc)
            elif k == "reward":
                err[k].append(np.abs(got[k] - want[k])); gap[k].append(np.abs(gp[k] - want[k]))
            else:
                err[k].append(np.abs(got[k] - want[k]).max(1)); gap[k].append(np.abs(gp[k] - want[k]).max(1))
    assert act_err < 2e-5, act_err            # float32 MLP (1263-wide first layer) against float64
    rows = []
    for k in names:
        rows += parity.quantile_rows(k, np.concatenate(err[k]), np.concatenate(gap[k]), qs=(0.5, 0.9))
    print(rows)
    parity.check_quantiles(rows, parity.ENV_FLOORS)


def util_tables():
    from rodent_amd import assets, mjcf
    return mjcf.load_blob(assets.asset_path("rodent_optimized"))


class _Recorder:
    """Stands between EvalWrapper and the wrapped env: records what goes into and comes out of every step."""

    def __init__(self, env):
        self.env, self.log = env, []

    def __getattr__(self, k):
        return getattr(self.env, k)

    def reset(self, rng):
        return self.env.reset(rng)

    def step(self, state, action):
        ns = self.env.step(state, action)
        self.log.append((state, action, ns))
        return ns


def test_evaluator_metrics_against_the_oracle(oracle_built):
    from rodent_amd import envs, jax_random
    from rodent_amd.envs import wrappers
    from rodent_amd.training import acting
    N, EP = 16, 25
    track = util.synthetic_track()
    env = envs.get_environment("rodent", track_pos=track, num_envs=N, xml_path="rodent_optimized.xml", iterations=8, ls_iterations=8, device=DEV,
                               healthy_z_range=(0.05, 0.5))          # a tight floor so that some episodes end before EP steps
    nets, make_policy = _nets(env, seed=1)
    rec = _Recorder(wrappers.wrap(env, episode_length=EP, action_repeat=1))
    ev = acting.Evaluator(rec, lambda p: make_policy(p, deterministic=False), N, EP, 1, jax_random.PRNGKey(4))
    got = ev.run_evaluation((None, nets.policy_network), training_metrics={"training/sps": 1.0})
    for k in ("eval/episode_reward", "eval/episode_pos_reward", "eval/episode_reward_quadctrl", "eval/episode_reward_alive", "eval/avg_episode_length",
              "eval/epoch_eval_time", "eval/sps", "eval/walltime", "training/sps"):
        assert k in got, k
    assert len(rec.log) == EP
    # restatement over the ORACLE's per-step outputs (teacher-forced from the recorded states): EvalWrapper sums metric * active,
    # active *= 1 - done, where done is the Episode wrapper's (env done, or steps >= EP)
    A = parity.OracleEnvImpl("rodent_optimized", N, "f64", (8, 8), track, z_range=(0.05, 0.5))
    B = parity.OracleEnvImpl("rodent_optimized", N, "f32", (8, 8), track, z_range=(0.05, 0.5))      # the yardstick of criterion C3
    active = np.ones(N)
    sums = dict(reward=np.zeros(N), pos_reward=np.zeros(N), reward_quadctrl=np.zeros(N), reward_alive=np.zeros(N))
    sums32 = {k: np.zeros(N) for k in sums}
    steps = np.zeros(N)
    ended_early = 0
    for state, action, ns in rec.log:
        w = A.env_step(_state_of(state), _np(action), state.info["cur_frame"].cpu().numpy())
        w32 = B.env_step(_state_of(state), _np(action), state.info["cur_frame"].cpu().numpy())
        ep_steps = np.where(_np(state.done) != 0, 0.0, _np(state.info["steps"])) + 1
        done = np.where(ep_steps >= EP, 1.0, w["done"])
        np.testing.assert_array_equal(_np(ns.info["steps"]), ep_steps)
        z = w["qpos"][:, 2]
        near = (np.abs(z - 0.05) < 1e-3) | (np.abs(z - 0.5) < 1e-3)
        assert not ((_np(ns.done) != done) & ~near).any()
        done = _np(ns.done)                          # at the threshold follow the HIP decision (float32 vs float64 side of it)
        steps += active
        sums["reward"] += w["reward"] * active
        sums32["reward"] += w32["reward"] * active
        for i, k in enumerate(("pos_reward", "reward_quadctrl", "reward_alive")):
            sums[k] += w["metrics"][:, i] * active
            sums32[k] += w32["metrics"][:, i] * active
        ended_early += int(((done != 0) & (ep_steps < EP) & (active != 0)).sum())
        active = active * (1 - done)
    assert ended_early >= 1 and ended_early < N                     # both kinds of episode are in the sample
    assert abs(got["eval/avg_episode_length"] - steps.mean()) < 1e-6
    # pos_reward = exp(-100 |dx|) magnifies a position error a hundredfold, so the bound is C3's: within 3x what the scalar float32 oracle
    # loses on the same (teacher-forced) steps, plus a floor for the single sample a mean is
    for k, v in sums.items():
        gap = abs(sums32[k].mean() - v.mean())
        assert abs(got["eval/episode_" + k] - v.mean()) <= 3 * gap + 3e-4 * max(1.0, abs(v.mean())), (k, got["eval/episode_" + k], v.mean(), gap)
