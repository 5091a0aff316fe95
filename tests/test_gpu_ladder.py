"""GPU parity ladder (tests/parity.py explains the design): the HIP kernel through the C ABI against the float64 oracle,

  1. 1000 teacher-forced SUBSTEPS at the launcher's CG 8/8 (debug-dump instance): contact / limit activity bits and limit
     rows against the oracle, iteration counts, error quantiles of qpos / qvel / qacc against the scalar float32 oracle's;
  2. the same inputs through the PRODUCTION instance (what bench.py times);
  3. 1000 teacher-forced ENV steps through `Rodent.step` (production instance): cur_frame / done exact, every observation
     segment, reward;
  4. a free-running 1000-step rollout with a CONVERGED solver (50/50) through Episode(150)+AutoReset against the oracle
     env: integer bookkeeping, restored first states, and the divergence curve (reported; asserted over the first 10 steps);
  5. the np_ref fixtures (tests/golden/step_*.npz): numbers that never passed through oracle/rodent_ref.c;
  6. `Rodent.reset`: observation segments and qacc_warmstart against the oracle's init.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import parity, util
from tests.hip_impl import HipEnvImpl, HipImpl, NoDiscrete

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _report(name, obj):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f"ladder_{name}.json"), "w") as f:
        json.dump(obj, f, indent=1, default=float)
    print(name, json.dumps(obj, default=float))


@pytest.fixture(scope="module")
def substep_inputs(oracle_built):
    return parity.rollout_inputs("rodent_optimized", 16, 1000, (8, 8), seed=31)


def test_teacher_forced_1000_substeps_debug_instance(substep_inputs):
    seq, A, tab = substep_inputs
    out = parity.substep_ladder(HipImpl("rodent_optimized", 16, (8, 8), True), seq, A, parity.OracleImpl("rodent_optimized", 16, "f32", (8, 8)))
    _report("substeps_cg8_debug", out)
    parity.assert_substep_criteria(out)


def test_teacher_forced_1000_substeps_production_instance(substep_inputs):
    seq, A, tab = substep_inputs
    out = parity.substep_ladder(NoDiscrete(HipImpl("rodent_optimized", 16, (8, 8), False), A), seq, A,
                                parity.OracleImpl("rodent_optimized", 16, "f32", (8, 8)))
    _report("substeps_cg8_production", out)
    parity.check_quantiles(out["quantiles"], parity.SUBSTEP_FLOORS)


def test_teacher_forced_substeps_rodent_pair(oracle_built):
    # the free bodies of the replicas sit at the origin (the torsos are offset inside them): no height test for the restarts
    seq, A, tab = parity.rollout_inputs("rodent_pair", 8, 200, (8, 8), seed=33, z_range=(-1.0, 1.0))
    assert len(np.unique(np.stack([st["qpos"][0] for st, _ in seq]).round(9), axis=0)) > 150
    out = parity.substep_ladder(HipImpl("rodent_pair", 8, (8, 8), True), seq, A, parity.OracleImpl("rodent_pair", 8, "f32", (8, 8)))
    _report("substeps_cg8_pair", out)
    parity.assert_substep_criteria(out)


def test_rodent_pair_two_wave_instance(oracle_built):
    """The production path of a two-tree model: one wavefront per replica, coupled only through the solver's scalar sums
    (rr_kernel.h PAIR; SURVEY.md config 5 [REF models/rodent_pair.xml:163-500]).  (1) teacher-forced substeps against the float64 oracle by
    the state criteria; (2) against the generic one-wave instance (RR_PAIR_WAVES=0) on the same inputs: two float32 evaluations of one
    map, equal to float32 round-off amplified by the solver -- the same bound class as (1); (3) pipeline_init and a 10-substep launch."""
    from rodent_amd import assets, hip
    N = 8
    seq, A, tab = parity.rollout_inputs("rodent_pair", N, 200, (8, 8), seed=33, z_range=(-1.0, 1.0))
    impl = HipImpl("rodent_pair", N, (8, 8), False)
    out = parity.substep_ladder(NoDiscrete(impl, A), seq, A, parity.OracleImpl("rodent_pair", N, "f32", (8, 8)))
    _report("substeps_cg8_pair_two_wave", out)
    parity.check_quantiles(out["quantiles"], parity.SUBSTEP_FLOORS)
    os.environ["RR_PAIR_WAVES"] = "0"
    try:
        generic = HipImpl("rodent_pair", N, (8, 8), False)
    finally:
        del os.environ["RR_PAIR_WAVES"]
    err = {k: [] for k in ("qpos", "qvel")}
    gap = {k: [] for k in ("qpos", "qvel")}
    for st, ctrl in seq[:60]:
        a, b, w = impl.substep(st, ctrl), generic.substep(st, ctrl), A.substep(st, ctrl)
        for k in err:
            err[k].append(np.abs(a[k] - b[k]).max(1)); gap[k].append(np.abs(b[k] - w[k]).max(1))
    rows = []
    for k in err:
        rows += parity.quantile_rows(k, np.concatenate(err[k]), np.concatenate(gap[k]))
    _report("pair_two_wave_vs_generic", rows)
    parity.check_quantiles(rows, parity.SUBSTEP_FLOORS)
    # n_frames = 10 and the forward-only launch go through the same instance
    ds = impl._dev(seq[0][0])
    impl.batch.pipeline_init(ds)
    impl.batch.pipeline_step(ds, torch.tensor(seq[0][1], dtype=torch.float32, device=DEV), 10)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in ds.values())


def test_teacher_forced_substeps_newton(oracle_built):
    """solver = Newton (the Hessian M + J'DJ factored per iteration through the level schedules of M)."""
    N, it = 16, (4, 8)
    seq, A, tab = parity.rollout_inputs("rodent_optimized", N, 400, it, seed=41, solver="newton")
    out = parity.substep_ladder(HipImpl("rodent_optimized", N, it, True, solver="newton"), seq, A,
                                parity.OracleImpl("rodent_optimized", N, "f32", it, solver="newton"))
    _report("substeps_newton4", out)
    parity.assert_substep_criteria(out, newton=True)
    out = parity.substep_ladder(NoDiscrete(HipImpl("rodent_optimized", N, it, False, solver="newton"), A), seq[:200], A,
                                parity.OracleImpl("rodent_optimized", N, "f32", it, solver="newton"))
    _report("substeps_newton4_production", out)
    parity.check_quantiles(out["quantiles"], parity.SUBSTEP_FLOORS)


def test_teacher_forced_env_steps_newton(oracle_built):
    """The reference's default env configuration [REF Rodent_Env_Brax.py:42-45, solver='cg' | 'newton'] with solver='newton'."""
    N, T, it = 16, 200, (4, 8)
    track = util.synthetic_track()
    seq, A0, tab = parity.rollout_inputs("rodent_optimized", N, T, it, seed=45, n_frames=10, reset_every=150, solver="newton")
    rng = np.random.default_rng(46)
    seq = [(st, ctrl, rng.integers(0, 260, N).astype(np.int32)) for st, ctrl in seq]
    A = parity.OracleEnvImpl("rodent_optimized", N, "f64", it, track, solver="newton")
    out = parity.envstep_ladder(HipEnvImpl(N, it, track, solver="newton"), seq, A,
                                parity.OracleEnvImpl("rodent_optimized", N, "f32", it, track, solver="newton"), tab)
    _report("envsteps_newton4", out)
    parity.check_quantiles(out["quantiles"], parity.ENV_FLOORS)


def test_teacher_forced_substeps_short_solver_settings(oracle_built):
    """The notebook's setting CG 2/4 [NB /root/reference/mjcf.ipynb:444] and Newton 1/4 -- the setting whose FREE-RUNNING rollout went
    non-finite on the GPU in round 2 (gpurun_out/nb_a_newton_1_4.err).  The oracle diverges in the same way (float64 and float32:
    tests/test_abi_and_oracle.py::test_one_newton_iteration_diverges_on_the_oracle_too), so it is the configuration, not the kernel;
    what CAN be held is the one-step map, started from states of a stable trajectory (Newton 4/8 / CG 8/8)."""
    N = 16
    seq, _, tab = parity.rollout_inputs("rodent_optimized", N, 150, (4, 8), seed=51, solver="newton")
    A = parity.OracleImpl("rodent_optimized", N, "f64", (1, 4), "newton")
    out = parity.substep_ladder(HipImpl("rodent_optimized", N, (1, 4), True, solver="newton"), seq, A,
                                parity.OracleImpl("rodent_optimized", N, "f32", (1, 4), solver="newton"))
    _report("substeps_newton_1_4", out)
    parity.assert_substep_criteria(out, newton=True)
    seq, _, tab = parity.rollout_inputs("rodent_optimized", N, 300, (8, 8), seed=53)
    A = parity.OracleImpl("rodent_optimized", N, "f64", (2, 4))
    out = parity.substep_ladder(HipImpl("rodent_optimized", N, (2, 4), True), seq, A, parity.OracleImpl("rodent_optimized", N, "f32", (2, 4)))
    _report("substeps_cg_2_4", out)
    parity.assert_substep_criteria(out)


def test_newton_is_refused_for_models_without_an_instance():
    from rodent_amd import assets, hip
    with pytest.raises(RuntimeError, match="Newton"):
        hip.Model(assets.asset_path("rodent_pair"), 4, 8, solver="newton")


def test_teacher_forced_1000_env_steps(oracle_built):
    N, T = 16, 1000
    track = util.synthetic_track()
    seq, A0, tab = parity.rollout_inputs("rodent_optimized", N, T, (8, 8), seed=35, n_frames=10, reset_every=150)
    rng = np.random.default_rng(36)
    seq = [(st, ctrl, rng.integers(0, 260, N).astype(np.int32)) for st, ctrl in seq]      # incl. frames beyond the clip (saturation)
    A = parity.OracleEnvImpl("rodent_optimized", N, "f64", (8, 8), track)
    out = parity.envstep_ladder(HipEnvImpl(N, (8, 8), track), seq, A, parity.OracleEnvImpl("rodent_optimized", N, "f32", (8, 8), track), tab)
    _report("envsteps_cg8", out)
    parity.check_quantiles(out["quantiles"], parity.ENV_FLOORS)


def test_free_running_converged_solver_1000_steps(oracle_built):
    """HIP rollout through wrappers.wrap(episode_length=150) vs the oracle env (float64, and float32 for the gap), solver
    50/50, same reset keys and actions.  Asserted: integer bookkeeping (cur_frame, steps, truncation) exact for all 1000
    steps; `done` equal while an env's trajectory has not separated; the state restored at an episode end is bit-identical
    to the first state; error quantiles over the first 10 steps within 3x the float32 oracle's.  Reported: the divergence
    curve to 1000 steps (profiles/r02_parity_ladder.json)."""
    from rodent_amd import envs, jax_random
    from rodent_amd.envs import wrappers
    from tests.oracle_env import OracleRodent
    N, T, EP = 32, 1000, 150
    track = util.synthetic_track()
    env = envs.get_environment("rodent", track_pos=track, num_envs=N, xml_path="rodent_optimized.xml", iterations=50, ls_iterations=50, device=DEV)
    wenv = wrappers.wrap(env, episode_length=EP, action_repeat=1)
    keys = jax_random.split(jax_random.PRNGKey(5), N)
    hs = wenv.reset(keys)
    A = OracleRodent("rodent_optimized", N, "f64", (50, 50), track, EP)
    B = OracleRodent("rodent_optimized", N, "f32", (50, 50), track, EP)
    A.reset(keys); B.reset(keys)
    first_qpos = hs.pipeline_state.qpos.clone()
    np.testing.assert_array_equal(hs.info["cur_frame"].cpu().numpy(), A.cur_frame)
    rng = np.random.default_rng(6)
    insync = np.ones(N, bool)
    curve, early = [], {"err": [], "gap": []}
    marks = {1, 2, 5, 10, 20, 50, 100, 149, 150, 151, 300, 500, 1000}
    for t in range(1, T + 1):
        a = parity.f32r(rng.uniform(-1, 1, (N, 30)))
        hs = wenv.step(hs, torch.tensor(a, dtype=torch.float32, device=DEV))
        A.step(a); B.step(a)
        hq = hs.pipeline_state.qpos.cpu().numpy().astype(np.float64)
        assert np.isfinite(hq).all() and torch.isfinite(hs.obs).all()
        np.testing.assert_array_equal(hs.info["cur_frame"].cpu().numpy(), A.cur_frame)          # never restored, saturating index
        # an env whose `done` history equals the oracle's has the same step counter and truncation flag, exactly
        np.testing.assert_array_equal(hs.info["steps"].cpu().numpy()[insync], A.steps[insync])
        hd = hs.done.cpu().numpy()
        insync &= hd == A.done
        np.testing.assert_array_equal(hs.info["truncation"].cpu().numpy()[insync], A.truncation[insync])
        ended = np.nonzero(hd)[0]
        if len(ended):                                                                             # AutoReset: bit-identical first state
            assert torch.equal(hs.pipeline_state.qpos[ended], first_qpos[ended])
        err = np.abs(hq - A.state()["qpos"]).max(1)
        gap = np.abs(B.state()["qpos"] - A.state()["qpos"]).max(1)
        if t <= 10:
            early["err"].append(err); early["gap"].append(gap)
        if t in marks:
            curve.append(dict(step=t, insync=int(insync.sum()), err_median=float(np.median(err)), err_max=float(err.max()),
                              f32_oracle_median=float(np.median(gap)), f32_oracle_max=float(gap.max())))
    rows = []
    for i in range(10):
        rows += parity.quantile_rows(f"qpos@{i + 1}", early["err"][i], early["gap"][i], qs=(0.5, 0.9))
    _report("free_running_cg50", dict(curve=curve, early=rows))
    parity.check_quantiles(rows, {f"qpos@{i + 1}": 2e-6 for i in range(10)})


@pytest.mark.parametrize("model_name", ["rodent_optimized", "rodent_pair"])
def test_np_ref_fixtures(model_name, oracle_built):
    """HIP against tests/golden/step_*.npz, generated by oracle/np_ref.py alone (tools/make_step_golden.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"step_{model_name}.npz"))
    n = g["in_qpos"].shape[0]
    st = {k: g[f"in_{k}"] for k in parity.STATE}
    impl = HipImpl(model_name, n, (8, 8), True)
    ds = impl._dev(st)
    impl.batch.pipeline_step(ds, torch.tensor(g["in_ctrl"], dtype=torch.float32, device=DEV), 1, out=dict(debug=impl.dbg))
    dbg = impl.dbg.cpu().numpy().astype(np.float64)
    f = lambda name: dbg[:, impl.lay[name][0]:impl.lay[name][0] + impl.lay[name][1]]
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    worst = {}
    for name, tol in (("xpos", 2e-6), ("xmat", 2e-6), ("cinert", 5e-6), ("cvel", 2e-5), ("qfrc_bias", 5e-5), ("qfrc_passive", 2e-6),
                      ("qfrc_actuator", 2e-6), ("qfrc_smooth", 5e-5), ("qacc_smooth", 1e-4), ("con_dist", 2e-5), ("con_pos", 2e-6),
                      ("con_frame", 2e-6)):
        worst[name] = (rel(f(name), g[f"fwd_{name}"]), tol)
    nl = len(impl.lim_dof)
    active = g["fwd_efc_pos"][:, nl:].reshape(n, -1, 4)[:, :, 0] < 0
    worst["con_D"] = (rel(f("con_D")[active], g["fwd_efc_D"][:, nl:].reshape(n, -1, 4)[:, :, 0][active]), 5e-5)
    worst["con_aref"] = (rel(f("con_aref").reshape(n, -1, 4)[active], g["fwd_efc_aref"][:, nl:].reshape(n, -1, 4)[active]), 5e-5)
    np.testing.assert_array_equal(f("con_dist") < 0, g["fwd_con_dist"] < 0)
    np.testing.assert_array_equal(f("niter_cost")[:, 0].astype(int), g["fwd_cg8_niter"][:, 0].astype(int))
    worst["qacc"] = (rel(f("qacc"), g["fwd_cg8_qacc"]), 2e-4)
    # one substep.  FIXED bounds, nothing of oracle/rodent_ref.c evaluated here (round 2 took 3x that oracle's float32 loss on these
    # states, which tied the "never passed through rodent_ref.c" fixture to it again): the worst of the n <= 8 fixture states must stay
    # within 3x the 99th percentile of the one-substep error over the 16 000 samples of the teacher-forced ladder
    # (profiles/r02_parity_ladder.json substeps_cg8_*: qpos q99 1.7e-5 / 1.8e-5, qvel q99 8.6e-3 / 8.9e-3; the error is heavy-tailed --
    # a contact switching inside the substep -- so the median, 1e-7 / 4e-5, is no bound for a maximum)
    worst["qpos_substep"] = (float(np.abs(ds["qpos"].cpu().numpy() - g["sub_cg8_qpos"]).max()), 5e-5)
    worst["qvel_substep"] = (float(np.abs(ds["qvel"].cpu().numpy() - g["sub_cg8_qvel"]).max()), 2.5e-2)
    print(model_name, {k: f"{v[0]:.2e}" for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v[0] <= v[1]}
    assert not bad, bad


def test_reset_matches_oracle_init(oracle_built):
    """`Rodent.reset`: every observation segment and qacc_warmstart against the oracle's init + get_obs (a2)."""
    from rodent_amd import envs, jax_random
    from tests.oracle_env import OracleRodent
    N = 16
    track = util.synthetic_track()
    env = envs.get_environment("rodent", track_pos=track, num_envs=N, xml_path="rodent_optimized.xml", iterations=8, ls_iterations=8, device=DEV)
    keys = jax_random.split(jax_random.PRNGKey(9), N)
    hs = env.reset(keys)
    A = OracleRodent("rodent_optimized", N, "f64", (8, 8), track)
    B = OracleRodent("rodent_optimized", N, "f32", (8, 8), track)
    obs = A.reset(keys); obs32 = B.reset(keys)
    np.testing.assert_array_equal(hs.info["cur_frame"].cpu().numpy(), A.cur_frame)
    np.testing.assert_array_equal(hs.pipeline_state.qpos.cpu().numpy(), A.state()["qpos"].astype(np.float32))   # host RNG: bit-exact
    np.testing.assert_array_equal(hs.pipeline_state.qvel.cpu().numpy(), A.state()["qvel"].astype(np.float32))
    got = hs.obs.cpu().numpy().astype(np.float64)
    seg = parity.obs_segments(A.tables)
    rows = []
    for k, s in seg.items():
        sc = np.maximum(np.abs(obs[:, s]).max(1), 1e-3) if k == "cinert" else 1.0
        rows += parity.quantile_rows("obs_" + k, np.abs(got[:, s] - obs[:, s]).max(1) / sc, np.abs(obs32[:, s] - obs[:, s]).max(1) / sc, qs=(0.5, 1.0))
    w = hs.pipeline_state.qacc_warmstart.cpu().numpy().astype(np.float64)
    sc = np.maximum(np.abs(A.state()["qacc_warmstart"]).max(1), 1.0)
    rows += parity.quantile_rows("qacc_warmstart", np.abs(w - A.state()["qacc_warmstart"]).max(1) / sc,
                                 np.abs(B.state()["qacc_warmstart"] - A.state()["qacc_warmstart"]).max(1) / sc, qs=(0.5, 1.0))
    _report("reset", dict(rows=rows))
    floors = dict(obs_qpos=0.0, obs_qvel=0.0, obs_cinert=2e-7, obs_cvel=2e-6, obs_qfrc_actuator=2e-7, obs_track_local=2e-7, qacc_warmstart=2e-6)
    parity.check_quantiles(rows, floors)


def test_timed_instance_at_timed_size_against_the_oracle(oracle_built):
    """The kernel instance and batch size bench.py times -- `rr_env_unroll` (multi-step instance) on 2048 envs -- held to the float64
    oracle DIRECTLY (round 2 tied it to the oracle only transitively: UNROLL == per-step calls bitwise at N = 96, per-step vs oracle at
    N = 16): from a rollout state 12 steps in (contacts made), ONE T = 1 launch of all 2048 envs; 512 sampled envs are re-stepped by the
    oracle from the same state and action (512, not 32: between its median and its 90th percentile the one-step error of the truncated
    CG 8/8 map spans three decades -- 9e-4 to 2.8 in qvel on the first run -- so the median of 27 samples scatters by a factor of 3).  C1: cur_frame exact, done exact away from the height thresholds, wrapper steps exact;
    C3: every observation segment, qpos, qvel, reward within 3x the scalar float32 oracle's own distance (tests/parity.py)."""
    from rodent_amd import envs, jax_random
    from rodent_amd.envs import wrappers
    N, EP = 2048, 150
    track = util.synthetic_track()
    env = envs.get_environment("rodent", track_pos=track, num_envs=N, xml_path="rodent_optimized.xml", iterations=8, ls_iterations=8, device=DEV)
    wenv = wrappers.wrap(env, episode_length=EP, action_repeat=1)
    g = torch.Generator(device=DEV).manual_seed(77)
    st = wenv.reset(jax_random.split(jax_random.PRNGKey(21), N))
    st = wenv.unroll(st, torch.rand(12, N, 30, device=DEV, generator=g) * 2 - 1)
    a = torch.rand(1, N, 30, device=DEV, generator=g) * 2 - 1
    before = {k: getattr(st.pipeline_state, k).cpu().numpy().astype(np.float64) for k in parity.STATE}
    cf0 = st.info["cur_frame"].cpu().numpy().copy()
    steps0, done0 = st.info["steps"].cpu().numpy().copy(), st.done.cpu().numpy().copy()
    ns = wenv.unroll(st, a)
    torch.cuda.synchronize()
    pick = np.random.default_rng(5).choice(N, 512, replace=False)
    sub = {k: v[pick] for k, v in before.items()}
    act = a[0].cpu().numpy().astype(np.float64)[pick]
    A = parity.OracleEnvImpl("rodent_optimized", 512, "f64", (8, 8), track)
    B = parity.OracleEnvImpl("rodent_optimized", 512, "f32", (8, 8), track)
    want, gap = A.env_step(sub, act, cf0[pick]), B.env_step(sub, act, cf0[pick])
    np.testing.assert_array_equal(ns.info["cur_frame"].cpu().numpy()[pick], want["cur_frame"])
    np.testing.assert_array_equal(ns.info["steps"].cpu().numpy()[pick], np.where(done0[pick] != 0, 0.0, steps0[pick]) + 1)
    z = want["qpos"][:, 2]
    near = (np.abs(z - 0.03) < 1e-3) | (np.abs(z - 0.5) < 1e-3)
    over = ns.info["steps"].cpu().numpy()[pick] >= EP
    hd = ns.done.cpu().numpy()[pick]
    assert not ((hd != np.where(over, 1.0, want["done"])) & ~near).any()
    keep = hd == 0                                            # envs that were auto-reset carry their first state, not the stepped one
    assert keep.sum() >= 400
    got_obs = ns.obs.cpu().numpy().astype(np.float64)[pick][keep]
    seg = parity.obs_segments(mjcf_tables("rodent_optimized"))
    rows = []
    for k, s in seg.items():
        sc = np.maximum(np.abs(want["obs"][keep][:, s]).max(1), 1e-3) if k == "cinert" else 1.0
        rows += parity.quantile_rows("obs_" + k, np.abs(got_obs[:, s] - want["obs"][keep][:, s]).max(1) / sc,
                                     np.abs(gap["obs"][keep][:, s] - want["obs"][keep][:, s]).max(1) / sc, qs=(0.5, 0.9))
    for k in ("qpos", "qvel"):
        gv = getattr(ns.pipeline_state, k).cpu().numpy().astype(np.float64)[pick][keep]
        rows += parity.quantile_rows(k, np.abs(gv - want[k][keep]).max(1), np.abs(gap[k][keep] - want[k][keep]).max(1), qs=(0.5, 0.9))
    rows += parity.quantile_rows("reward", np.abs(ns.reward.cpu().numpy()[pick][keep] - want["reward"][keep]), np.abs(gap["reward"][keep] - want["reward"][keep]), qs=(0.5, 0.9))
    _report("timed_instance_2048", rows)
    floors = dict(parity.ENV_FLOORS, obs_qpos=parity.ENV_FLOORS["qpos"], obs_qvel=parity.ENV_FLOORS["qvel"])
    parity.check_quantiles(rows, floors)


def mjcf_tables(name):
    from rodent_amd import assets, mjcf
    return mjcf.load_blob(assets.asset_path(name))
