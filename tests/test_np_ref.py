"""The C oracle (oracle/rodent_ref.c, float64 build) held to the INDEPENDENT NumPy-float64 restatement oracle/np_ref.py
(per-body Jacobian mass matrix, classical Newton-Euler bias forces, rotation-matrix kinematics, dense Cholesky), and the
model compiler's mj_setConst constants held to np_ref's own.  Two formulations agreeing to double round-off rules out an
error of derivation shared by oracle and kernel; it does NOT pin either to MJX (parity unpinned vs the reference)."""
import os

import numpy as np
import pytest

from oracle import np_ref
from tests import util

G = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return np.abs(np.asarray(a).ravel() - np.asarray(b).ravel()).max() / max(np.abs(b).max(), 1e-30)


def _np_data(m, st, e, ctrl):
    d = np_ref.Data(m)
    d.qpos, d.qvel, d.act, d.qacc_warmstart = (st[k][e].copy() for k in ("qpos", "qvel", "act", "qacc_warmstart"))
    d.ctrl = np.asarray(ctrl, np.float64).copy()
    return d


@pytest.mark.parametrize("model_name,n", [("rodent_optimized", 6), ("rodent_new", 2), ("rodent_pair", 2)])
def test_forward_stages_agree_to_double_roundoff(oracle_built, model_name, n):
    ref = oracle_built
    st, M, tab = util.settled_states(ref, model_name, n, seed=3, iterations=(8, 8))
    m = np_ref.Model(tab, 8, 8)
    rng = np.random.default_rng(0)
    worst = {}
    nact = 0
    for e in range(n):
        ctrl = rng.uniform(-1, 1, M.nu)
        c = util.oracle_forward(ref, M, st, e, ctrl)
        d = _np_data(m, st, e, ctrl)
        np_ref.forward(m, d)
        pairs = dict(xpos=d.xpos, xmat=d.xmat, xipos=d.xipos, cinert=d.cinert, cvel=d.cvel, qfrc_bias=d.qfrc_bias,
                     qfrc_passive=d.qfrc_passive, qfrc_actuator=d.qfrc_actuator, qfrc_smooth=d.qfrc_smooth, qacc_smooth=d.qacc_smooth,
                     con_dist=d.con_dist, con_pos=d.con_pos, con_frame=d.con_frame, efc_J=d.efc_J, efc_D=d.efc_D, efc_aref=d.efc_aref,
                     efc_pos=d.efc_pos)
        for k, v in pairs.items():
            worst[k] = max(worst.get(k, 0.0), _rel(v, c.get(k)))
        worst["qM"] = max(worst.get("qM", 0.0), _rel(d.M, np_ref.dense_from_sparse(tab, c.get("qM"))))
        # the solver is branchy: identical iteration counts and a result equal to ~1e-9 (8 truncated CG iterations)
        assert d.solver_niter == int(c.get("solver_niter")[0])
        worst["qacc"] = max(worst.get("qacc", 0.0), _rel(d.qacc, c.get("qacc")))
        worst["qfrc_constraint"] = max(worst.get("qfrc_constraint", 0.0), _rel(d.qfrc_constraint, c.get("qfrc_constraint")))
        nact += int((d.con_dist < 0).sum())
    print({k: f"{v:.1e}" for k, v in worst.items()})
    assert nact >= n
    tol = dict(qacc=1e-9, qfrc_constraint=1e-9, qacc_smooth=1e-10)
    bad = {k: v for k, v in worst.items() if not v <= tol.get(k, 1e-10)}
    assert not bad, bad


@pytest.mark.parametrize("model_name", ["rodent_optimized", "rodent_new", "rodent_pair", "rodent_0"])
def test_set_const_of_the_model_compiler(model_name):
    """dof_invweight0 / body_invweight0 / stat.meaninertia in the blob (rodent_amd/mjcf.py _set_const, float32 on disk)
    against np_ref's own Jacobian-based computation at qpos0."""
    from rodent_amd import assets, mjcf
    tab = mjcf.load_blob(assets.asset_path(model_name))
    m = np_ref.Model(tab)
    dw, bw, mi = np_ref.set_const(m)
    assert _rel(dw, tab["dof_invweight0"]) < 3e-7          # float32 storage: 6e-8
    assert _rel(bw, tab["body_invweight0"]) < 3e-7
    assert abs(mi - float(tab["stat_meaninertia"])) / mi < 3e-7
    # contact invweight = invweight0 of the two bodies (world = 0), translational component
    np.testing.assert_allclose(tab["con_invweight"], bw[tab["con_body2"], 0] + bw[tab["con_body1"], 0], rtol=3e-7)


def test_converged_trajectory_100_substeps(oracle_built):
    """100 substeps with a converged solver (50/50): the two formulations stay together to 1e-6 (measured 3e-9); with the
    truncated 8/8 solver a double round-off difference is amplified to 1e-2 over the same horizon (branch flips in the
    line search), which is why long-horizon parity is asserted on the converged map only."""
    ref = oracle_built
    st, M, tab = util.settled_states(ref, "rodent_optimized", 2, seed=4, iterations=(50, 50))
    m = np_ref.Model(tab, 50, 50)
    rng = np.random.default_rng(0)
    for e in range(2):
        c = ref.RefData(M)
        for k in ("qpos", "qvel", "act", "qacc_warmstart"):
            c.set(k, st[k][e])
        d = _np_data(m, st, e, np.zeros(M.nu))
        for s in range(10):
            ctrl = rng.uniform(-1, 1, M.nu)
            c.step(ctrl, 10)
            np_ref.step(m, d, ctrl, 10)
        assert np.abs(d.qpos - c.get("qpos")).max() < 1e-6
        assert np.abs(d.qvel - c.get("qvel")).max() < 1e-4
        assert np.abs(d.act - c.get("act")).max() < 1e-12


@pytest.mark.parametrize("model_name", ["rodent_optimized", "rodent_pair"])
def test_c_oracle_reproduces_the_np_ref_fixtures(oracle_built, model_name):
    """tests/golden/step_*.npz (tools/make_step_golden.py, np_ref only) replayed through the C oracle: forward stages,
    one substep, one env step incl. obs / reward / done / cur_frame."""
    ref = oracle_built
    from rodent_amd import assets
    g = np.load(os.path.join(G, f"step_{model_name}.npz"))
    n = g["in_qpos"].shape[0]
    st = {k: g[f"in_{k}"] for k in ("qpos", "qvel", "act", "qacc_warmstart")}
    for its, tag in (((8, 8), "cg8"), ((50, 50), "cg50")):
        M = ref.RefModel(assets.asset_path(model_name), "f64")
        M.set_iterations(*its)
        for e in range(n):
            c = util.oracle_forward(ref, M, st, e, g["in_ctrl"][e])
            if tag == "cg8":
                for s in ("xpos", "xmat", "cinert", "cvel", "qfrc_bias", "qfrc_smooth", "qacc_smooth", "con_dist", "con_pos",
                          "con_frame", "efc_D", "efc_aref", "efc_pos"):
                    assert _rel(c.get(s), g[f"fwd_{s}"][e]) < 1e-10, s
            assert int(c.get("solver_niter")[0]) == int(g[f"fwd_{tag}_niter"][e, 0])
            assert _rel(c.get("qacc"), g[f"fwd_{tag}_qacc"][e]) < 1e-8
            c = ref.RefData(M)
            for k in st:
                c.set(k, st[k][e])
            c.step(g["in_ctrl"][e], 1)
            assert np.abs(c.get("qpos") - g[f"sub_{tag}_qpos"][e]).max() < 1e-11
            assert np.abs(c.get("qvel") - g[f"sub_{tag}_qvel"][e]).max() < 1e-8
            if tag == "cg50" and model_name == "rodent_optimized":       # 10 substeps: only the converged map is smooth enough
                c = ref.RefData(M)
                for k in st:
                    c.set(k, st[k][e])
                obs, rew, done, cf, met = c.env_step(g["in_ctrl"][e], g["track"], int(g["in_cur_frame"][e]))
                assert cf == int(g[f"env_{tag}_cur_frame"][e, 0]) and done == g[f"env_{tag}_done"][e, 0]
                assert np.abs(c.get("qpos") - g[f"env_{tag}_qpos"][e]).max() < 1e-8
                assert _rel(obs, g[f"env_{tag}_obs"][e]) < 1e-7
                assert abs(rew - g[f"env_{tag}_reward"][e, 0]) < 1e-8


def test_newton_solver_in_both_formulations(oracle_built):
    """`solver='newton'` [REF Rodent_Env_Brax.py:42-45; UP mjx solver, SolverType.NEWTON: H = M + J' diag(D active) J, Cholesky,
    search = -H^-1 grad]: the C oracle (dense Cholesky by hand) and np_ref (scipy) agree to double round-off with identical
    iteration counts, and Newton reaches the optimum a long CG run converges to."""
    ref = oracle_built
    from rodent_amd import assets
    st, M, tab = util.settled_states(ref, "rodent_optimized", 4, seed=3, iterations=(8, 8))
    M.set_solver("newton")
    m = np_ref.Model(tab, 8, 8)
    m.solver = "newton"
    rng = np.random.default_rng(0)
    for e in range(4):
        ctrl = rng.uniform(-1, 1, M.nu)
        c = util.oracle_forward(ref, M, st, e, ctrl)
        d = _np_data(m, st, e, ctrl)
        np_ref.forward(m, d)
        assert d.solver_niter == int(c.get("solver_niter")[0]) <= 8
        assert _rel(d.qacc, c.get("qacc")) < 1e-10
    Mcg = ref.RefModel(assets.asset_path("rodent_optimized"), "f64")
    Mcg.set_iterations(300, 50)
    M.set_iterations(20, 50)
    for e in range(2):
        a, b = util.oracle_forward(ref, M, st, e, np.zeros(M.nu)), util.oracle_forward(ref, Mcg, st, e, np.zeros(M.nu))
        assert abs(a.get("solver_cost")[0] - b.get("solver_cost")[0]) <= 1e-8 * abs(b.get("solver_cost")[0])
        assert _rel(a.get("qacc"), b.get("qacc")) < 1e-4 and a.get("solver_niter")[0] < 10 < b.get("solver_niter")[0]
