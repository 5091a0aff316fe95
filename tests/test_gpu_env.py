"""GPU: env-level behaviour of the HIP path against the oracle (reset / step / obs / reward / cur_frame)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _mk_env(N, model="rodent_optimized"):
    from rodent_amd import envs
    return envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=N, xml_path=f"{model}.xml",
                                iterations=8, ls_iterations=8, device="cuda:0")


def test_n_frames_fused_equals_repeated_launches():
    """10 substeps inside one launch == 10 launches of one substep, bit for bit (no state is lost between frames)."""
    N = 32
    env = _mk_env(N)
    state = env.reset(7)
    ctrl = torch.rand(N, env.action_size, device="cuda:0") * 2 - 1
    ps = state.pipeline_state
    a = dict(qpos=ps.qpos.clone(), qvel=ps.qvel.clone(), act=ps.act.clone(), qacc_warmstart=ps.qacc_warmstart.clone())
    b = {k: v.clone() for k, v in a.items()}
    env._batch.pipeline_step(a, ctrl, 10)
    for _ in range(10):
        env._batch.pipeline_step(b, ctrl, 1)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_step_is_deterministic_across_launches_and_batch_positions():
    """The same state stepped twice, and the same env placed at different batch positions, give bit-identical results:
    one wavefront owns an env, nothing is shared between envs and no reduction order depends on scheduling."""
    N = 2048
    env = _mk_env(N)
    state = env.reset(3)
    g = torch.Generator(device="cuda:0"); g.manual_seed(11)
    for _ in range(3):
        state = env.step(state, torch.rand(N, env.action_size, device="cuda:0", generator=g) * 2 - 1)
    ps = state.pipeline_state
    ctrl = torch.rand(N, env.action_size, device="cuda:0", generator=g) * 2 - 1
    base = dict(qpos=ps.qpos, qvel=ps.qvel, act=ps.act, qacc_warmstart=ps.qacc_warmstart)
    runs = []
    for rep in range(3):
        st = {k: v.clone() for k, v in base.items()}
        env._batch.pipeline_step(st, ctrl, 10)
        runs.append(st)
    perm = torch.randperm(N, device="cuda:0", generator=g)
    st = {k: v[perm].clone() for k, v in base.items()}
    env._batch.pipeline_step(st, ctrl[perm].contiguous(), 10)
    torch.cuda.synchronize()
    for k in base:
        assert torch.isfinite(runs[0][k]).all(), k
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
        assert torch.equal(runs[0][k][perm], st[k]), k


def test_fused_training_wrapper_equals_the_composition():
    """FusedEpisodeAutoResetWrapper (one launch) == AutoResetWrapper(EpisodeWrapper(...)) (tensor ops), bit for bit, through
    episode ends (episode_length 7), terminations and the restored first states."""
    import dataclasses
    from rodent_amd.envs import wrappers as W
    N = 64
    env = _mk_env(N)
    a_env = W.AutoResetWrapper(W.EpisodeWrapper(W.VmapWrapper(env), 7, 1))
    b_env = W.FusedEpisodeAutoResetWrapper(W.VmapWrapper(env), 7)
    assert isinstance(W.wrap(env, episode_length=7, action_repeat=1), W.FusedEpisodeAutoResetWrapper)
    sa, sb = a_env.reset(4), b_env.reset(4)
    g = torch.Generator(device="cuda:0"); g.manual_seed(9)
    seen_done = 0
    for t in range(40):
        act = torch.rand(N, env.action_size, device="cuda:0", generator=g) * 4 - 2      # wild actions: some envs fall over
        sa, sb = a_env.step(sa, act), b_env.step(sb, act)
        seen_done += int(sa.done.sum())
        for f in dataclasses.fields(sa.pipeline_state):
            x, y = getattr(sa.pipeline_state, f.name), getattr(sb.pipeline_state, f.name)
            if torch.is_tensor(x):
                assert torch.equal(x, y), (t, f.name)
        for name in ("obs", "reward", "done"):
            assert torch.equal(getattr(sa, name), getattr(sb, name)), (t, name)
        for k in ("steps", "truncation", "cur_frame"):
            assert torch.equal(sa.info[k], sb.info[k]), (t, k)
        for k in sa.metrics:
            assert torch.equal(sa.metrics[k], sb.metrics[k]), (t, k)
    assert seen_done >= 5 * N          # every env went through several episode ends


def test_env_step_matches_oracle(oracle_built):
    from rodent_amd import assets
    ref = oracle_built
    N = 16
    env = _mk_env(N)
    track = util.synthetic_track()
    state = env.reset(0)
    action = torch.rand(N, env.action_size, device="cuda:0") * 2 - 1
    nstate = env.step(state, action)
    torch.cuda.synchronize()
    M64 = ref.RefModel(assets.asset_path("rodent_optimized"), "f64"); M64.set_iterations(8, 8)
    M32 = ref.RefModel(assets.asset_path("rodent_optimized"), "f32"); M32.set_iterations(8, 8)
    errs, gaps = [], []
    for e in range(N):
        res = {}
        for tag, M in (("f64", M64), ("f32", M32)):
            d = ref.RefData(M)
            d.init(state.pipeline_state.qpos[e].cpu().numpy(), state.pipeline_state.qvel[e].cpu().numpy())
            res[tag] = d.env_step(action[e].cpu().numpy().astype(np.float64), track, int(state.info["cur_frame"][e]), n_frames=10) + (d.get("qpos"),)
        obs, rew, done, cf, met, qpos = res["f64"]
        got_q = nstate.pipeline_state.qpos[e].cpu().numpy()
        errs.append(np.abs(got_q - qpos).max())
        gaps.append(np.abs(res["f32"][5] - qpos).max())
        assert int(nstate.info["cur_frame"][e]) == cf            # integer bookkeeping is bit-exact
        assert float(nstate.done[e]) == done
        assert abs(float(nstate.reward[e]) - rew) < 1e-3 + 20 * errs[-1]
        # obs layout: first nq entries are qpos, last 3 the local tracking vector
        np.testing.assert_allclose(nstate.obs[e, :env.sys.nq].cpu().numpy(), got_q, rtol=0, atol=0)
        scale = np.maximum(np.abs(obs), 1e-2)
        assert np.median(np.abs(nstate.obs[e].cpu().numpy() - obs) / scale) < 1e-4
    print("max |qpos - oracle f64| per env after 1 env-step:", np.array2string(np.array(errs), precision=2))
    print("oracle f32 vs f64 gap per env:                 ", np.array2string(np.array(gaps), precision=2))
    # HIP float32 must sit as close to the float64 truth as the scalar float32 oracle does (tests/util.py assert_f32_class)
    print("geometric mean of err / gap:", util.assert_f32_class(errs, gaps))
