"""SURVEY.md 8(f)-4, self-collision primitives on the CPU oracles: sphere-sphere, sphere-capsule, capsule-capsule between two MOVING
bodies [UP mjx collision_primitive / math.closest_segment_*; restated from memory of the 3.1-era sources -- the packages are absent,
so parity against MJX itself stays unpinned], with J = jac(body2) - jac(body1) and condim-1 (frictionless, one row) next to condim-3
contacts -- what rodent_cpu.xml (BASELINE config 1 [REF models/rodent_cpu.xml:14-22, :477-504 excludes]) needs.

Checks that do not depend on either oracle's code: the primitives against a brute-force minimisation over points of the two
segments; the contact rows against finite differences of the signed distance (J n-row = d dist / d qpos); the two independent
formulations (C: cdof chains of both bodies; np_ref: point Jacobians) against each other to double round-off on rodent_cpu states
that do collide."""
import numpy as np
import pytest

from oracle import np_ref
from rodent_amd import assets, mjcf


def _brute_segments(a0, a1, b0, b1, n=400):
    s = np.linspace(0, 1, n)
    A = a0[None] + s[:, None] * (a1 - a0)[None]
    B = b0[None] + s[:, None] * (b1 - b0)[None]
    d = np.linalg.norm(A[:, None, :] - B[None, :, :], axis=2)
    return d.min()


def test_segment_primitives_against_brute_force():
    rng = np.random.default_rng(0)
    worst = 0.0
    for it in range(300):
        a0, a1, b0, b1 = (rng.normal(size=3) * 0.05 for _ in range(4))
        if it % 5 == 0:                       # nearly parallel segments: the 1e-6 in the denominator matters there
            b1 = b0 + (a1 - a0) * rng.uniform(0.5, 2) + rng.normal(size=3) * 1e-4
        pa, pb = np_ref._segment_segment(a0, a1, b0, b1)
        got, want = np.linalg.norm(pa - pb), _brute_segments(a0, a1, b0, b1)
        # the points lie on their segments
        for p, (x0, x1) in ((pa, (a0, a1)), (pb, (b0, b1))):
            t = (p - x0) @ (x1 - x0) / ((x1 - x0) @ (x1 - x0))
            assert -1e-9 <= t <= 1 + 1e-9 and np.linalg.norm(x0 + t * (x1 - x0) - p) < 1e-12
        assert got >= want - 5e-5                            # never closer than the true minimum (up to the grid's resolution)
        worst = max(worst, got - want)
    assert worst < 3e-3 * 0.05, worst                        # [UP]'s regularised line-line solve is not the exact minimiser; it is close
    p = np_ref._segment_point(np.zeros(3), np.array([1.0, 0, 0]), np.array([2.0, 1, 0]))
    np.testing.assert_allclose(p, [1, 0, 0], atol=1e-6)
    dist, pos, fr = np_ref._two_spheres(np.zeros(3), 0.1, np.array([0.0, 0, 0.15]), 0.1)
    assert abs(dist + 0.05) < 1e-15 and np.allclose(pos, [0, 0, 0.075]) and np.allclose(fr[0], [0, 0, 1]) and abs(np.linalg.det(fr) - 1) < 1e-12


@pytest.fixture(scope="module")
def colliding_states(oracle_built):
    """rodent_cpu states in which sphere / capsule pairs penetrate: random joint angles inside the limits."""
    ref = oracle_built
    path = assets.asset_path("rodent_cpu")
    tab = mjcf.load_blob(path)
    M = ref.RefModel(path, "f64")
    M.set_iterations(8, 8)
    rng = np.random.default_rng(2)
    lo, hi = tab["jnt_range"][:, 0], tab["jnt_range"][:, 1]
    out = []
    for e in range(600):
        q = tab["qpos0"] + rng.uniform(0.3, 1.0) * rng.uniform(lo, hi)
        d = ref.RefData(M)
        d.init(q.astype(np.float64), rng.uniform(-0.5, 0.5, M.nv))
        dist = d.get("con_dist")
        if (dist < 0).sum() >= 2:
            out.append((q.astype(np.float64), d.get("qvel"), d))
        if len(out) == 4:
            break
    assert len(out) >= 3
    return ref, M, tab, out


def test_compiled_contact_tables():
    tab = mjcf.load_blob(assets.asset_path("rodent_cpu"))
    kinds, dims = tab["con_kind"], tab["con_dim"]
    assert int(tab["ncon"]) == 2243 and int(tab["ndropped_pairs"]) == 2028            # sphere / capsule pairs kept; ellipsoid / box pairs dropped
    assert set(kinds.tolist()) == {4, 5, 6} and set(dims.tolist()) == {1, 3}
    assert int(tab["nefc"]) == int(tab["nlimit"]) + int((dims == 1).sum()) + 4 * int((dims == 3).sum())
    gt = tab["geom_type"]
    # MJX order: grouped by (type1, type2) ascending, geom ids ascending inside a group
    key = [(int(gt[a]), int(gt[b]), int(a), int(b)) for a, b in zip(tab["con_geom1"], tab["con_geom2"])]
    assert key == sorted(key) and all(k[0] <= k[1] for k in key)
    assert np.all(tab["con_body1"] > 0) and np.all(tab["con_body1"] != tab["con_body2"])
    assert int(tab["hip_supported"]) == 1 and int(tab["k_dyn"]) == 1


def test_two_formulations_agree_on_colliding_states(colliding_states):
    ref, M, tab, states = colliding_states
    m = np_ref.Model(tab, 8, 8)
    nact = 0
    for q, v, c in states:
        d = np_ref.Data(m)
        np_ref.init(m, d, q.copy(), v.copy())
        act = c.get("con_dist") < 0
        nact += int(act.sum())
        for k in ("con_dist", "con_pos", "con_frame", "efc_D", "efc_aref", "efc_pos"):
            a, b = np.asarray(getattr(d, k)).ravel(), c.get(k)
            assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-30), k
        J = c.get("efc_J").reshape(M.nefc, M.nv)
        assert np.abs(d.efc_J - J).max() <= 1e-11 * np.abs(J).max()
        assert int(d.solver_niter) == int(c.get("solver_niter")[0])
        assert np.abs(d.qacc - c.get("qacc")).max() <= 1e-8 * np.abs(c.get("qacc")).max()
    assert nact >= 6


def test_normal_rows_are_the_gradient_of_the_distance(colliding_states):
    """For an active contact the (first) row of J is d dist / d q: central differences of con_dist over the hinge angles."""
    ref, M, tab, states = colliding_states
    q, v, c = states[0]
    J = c.get("efc_J").reshape(M.nefc, M.nv)
    dist = c.get("con_dist")
    dims = tab["con_dim"]
    row_of = M.nlimit + np.concatenate([[0], np.cumsum(np.where(dims == 1, 1, 4))[:-1]])
    act = np.nonzero(dist < 0)[0][:6]
    mu = tab["con_friction"][:, 0]
    h = 1e-6
    for ci in act:
        r = int(row_of[ci])
        n_row = J[r] if dims[ci] == 1 else 0.5 * (J[r] + J[r + 1])          # (jn + mu j1 + jn - mu j1) / 2
        g = np.zeros(M.nv)
        for dof in range(M.nv):
            qp, qm = q.copy(), q.copy()
            qp[dof] += h; qm[dof] -= h
            dp, dm = ref.RefData(M), ref.RefData(M)
            dp.init(qp, v); dm.init(qm, v)
            g[dof] = (dp.get("con_dist")[ci] - dm.get("con_dist")[ci]) / (2 * h)
        np.testing.assert_allclose(n_row, g, atol=2e-6 * max(1.0, np.abs(g).max()))


def test_config1_four_envs_step_with_self_collisions(colliding_states):
    """BASELINE config 1 on the CPU oracle WITH the sphere / capsule self-collisions (2243 pairs): 4 envs, the first two started in
    colliding poses.  Shapes, finiteness, determinism; the contacts act: the penetrating pairs are pushed apart."""
    from tests import util
    from tests.oracle_env import OracleRodent
    ref, M, tab, states = colliding_states
    runs = []
    for rep in range(2):
        env = OracleRodent("rodent_cpu", 4, "f64", (6, 6), util.synthetic_track(), episode_length=150)
        env.reset(0)
        st = env.state()
        for e in range(2):
            st["qpos"][e] = states[e][0]
            st["qvel"][e] = 0.0
        env.b.set_state(st)
        for d_ in env.b.d:
            d_.forward()
        d0 = env.b.get("con_dist")
        act0 = d0 < 0
        assert act0[:2].sum() >= 2
        frc = env.b.get("qfrc_constraint")
        assert np.abs(frc[:2]).max() > 0
        rng = np.random.default_rng(0)
        for t in range(6):
            obs = env.step(np.zeros((4, 38)) if t < 3 else rng.uniform(-1, 1, (4, 38)))
            assert obs.shape == (4, 1244) and np.isfinite(obs).all()
        d1 = env.b.get("con_dist")
        assert (d1[act0] > d0[act0]).mean() > 0.7               # penetrations shrink
        runs.append(env.state()["qpos"].copy())
    assert np.array_equal(runs[0], runs[1])
