"""CPU: the C-ABI library loads and exports every symbol include/rodent_rr.h declares (no compute calls);
the model compiler's derived constants; oracle invariants (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

from rodent_amd import assets, hip, mjcf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rodent_rr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rr_[a-z_]+)\s*\(", hdr))
    assert declared == set(hip.EXPORTS)
    assert os.path.exists(hip.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(hip.LIB_PATH)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_kernarg_layout_matches_the_code_object():
    """The step kernel re-reads its I/O block through the kernarg segment pointer at offsetof(RRKArgs, io): that offset (and
    the block's size) must be what the DEVICE compiler assigned to the third explicit argument of every rr_step_kernel
    instance (code-object metadata, tools/kernel_meta.py).  Round 1 recorded a nil-address GPU fault while this mechanism
    was being introduced (gpurun_out/dbg_abort.log); this is the build-time guard."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_meta
    off, size, total = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    hip.lib().rr_kernarg_layout(ctypes.byref(off), ctypes.byref(size), ctypes.byref(total))
    ks = [k for k in kernel_meta.kernels(hip.LIB_PATH) if "rr_step_kernel" in k["name"]]
    assert len(ks) >= 8
    for k in ks:
        explicit = [a for a in k["args"] if a[2] == "by_value"]
        assert len(explicit) == 5, k["name"]
        assert explicit[2][:2] == (off.value, size.value), (k["name"], explicit, off.value, size.value)
        assert explicit[4][0] + explicit[4][1] == total.value
        assert k["lds"] == 0                      # the level schedules address LDS from byte 0: no static LDS


def test_model_tables_through_the_abi():
    """The integer ids the north star calls bit-exact, read back from the LOADED model through rr_model_table:
    Contact.geom1 / geom2 of rodent_optimized = plane-capsule pairs ascending (2 points each), then the plane-ellipsoid
    pairs [SURVEY.md Appendix B; ordering rule confirmed by NB mjcf.ipynb:917-921], and brax's link_idx = geom_bodyid - 1."""
    m = hip.Model(assets.asset_path("rodent_optimized"))
    caps = [11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 31, 42, 86, 87, 88, 90, 91, 92, 94, 95, 96, 98, 99, 100]
    want = [g for c in caps for g in (c, c)] + [1, 62, 75, 89, 97]
    assert m.table("con_geom2").tolist() == want
    assert m.table("con_geom1").tolist() == [0] * 59
    gb = m.table("geom_bodyid")
    assert gb.dtype == np.int32 and len(gb) == 101
    assert (gb[m.table("con_geom2")] == m.table("con_body2")).all()
    blob = mjcf.load_blob(assets.asset_path("rodent_optimized"))
    for name in ("dof_bodyid", "dof_parentid", "body_parentid", "jnt_qposadr", "jnt_dofadr", "actuator_dofadr", "con_kind"):
        assert m.table(name).tolist() == blob[name].ravel().tolist(), name
    assert m.table("body_mass").dtype == np.float32
    with pytest.raises(RuntimeError, match="no table"):
        m.table("no_such_table")


def test_model_load_dims_and_errors_without_gpu():
    m = hip.Model(assets.asset_path("rodent_optimized"), iterations=8, ls_iterations=8)
    d = m.dims
    assert (d.nq, d.nv, d.nu, d.nbody, d.ncon, d.nefc, d.obs_dim) == (74, 73, 30, 66, 59, 303, 1263)
    assert (d.iterations, d.ls_iterations) == (8, 8) and abs(d.timestep - 0.002) < 1e-9
    assert 0 < d.lds_bytes <= 160 * 1024 // 4            # four environments per CU
    # the benchmark models run the instance compiled for their dimensions (incl. the schedule-table parameters ktables.py derives)
    assert d.fixed_instance == 1 and hip.Model(assets.asset_path("rodent_new")).dims.fixed_instance == 1
    assert hip.Model(assets.asset_path("rodent_pair")).dims.fixed_instance == 0
    with pytest.raises(RuntimeError, match="cannot open"):
        hip.Model("/nonexistent.rrm")
    bad = os.path.join(ROOT, "tests", "golden", "env_step_reset.json")
    with pytest.raises(RuntimeError, match="RRM1"):
        hip.Model(bad)


def test_blob_of_an_older_table_format_is_refused(tmp_path):
    """round-1/2 blobs carry level schedules with flags and no alias-cell count: the row-program executor must not run them"""
    tab = mjcf.load_blob(assets.asset_path("rodent_optimized"))
    assert int(tab["k_nalias"]) > 0 and int(tab["k_factor3_rows"]) < int(tab["k_factor3p_rows"])
    old = {k: v for k, v in tab.items() if k not in ("k_nalias", "k_factor3p", "k_factor3p_rows")}
    path = str(tmp_path / "old.rrm")
    mjcf.save_blob(old, path)
    with pytest.raises(RuntimeError, match="lacks 'k_"):
        hip.Model(path)


@pytest.mark.parametrize("name,dims", [("rodent_optimized", (66, 74, 73, 30, 59, 303, 1119, 1263)),
                                       ("rodent_new", (67, 74, 73, 30, 57, 295, 1119, 1279)),
                                       ("rodent_pair", (133, 148, 146, 60, 114, 590, 2238, 2555)),
                                       ("rodent_0", (66, 74, 73, 30, 34, 203, 1119, 1263))])
def test_compiled_model_dims(name, dims):
    """SURVEY.md section 8 table ([DERIVED] from the XML); nefc 203 of rodent_0 is the notebook datum."""
    m = mjcf.load_blob(assets.asset_path(name))
    got = tuple(int(m[k]) for k in ("nbody", "nq", "nv", "nu", "ncon", "nefc", "nM", "obs_dim"))
    assert got == dims


def test_compiler_matches_shipped_blob_when_reference_present():
    xml = "/root/reference/models/rodent_optimized.xml"
    if not os.path.exists(xml):
        pytest.skip("reference MJCF not present")
    m = mjcf.compile_mjcf(xml)
    blob = mjcf.load_blob(assets.asset_path("rodent_optimized"))
    for k, v in blob.items():
        if k.startswith("opt_"):
            continue
        np.testing.assert_allclose(v, np.asarray(m[k]), rtol=1e-6, atol=1e-30, err_msg=k)
    # primitive inertia known answers: 1 cm sphere at density 500 (value printed in the reference notebook)
    vol, _ = mjcf._geom_volume_inertia(mjcf.SPHERE, [0.01, 0, 0])
    assert abs(vol * 500 - 0.002094395) < 1e-9
    assert abs(float(m["body_mass"].sum()) - 0.25623) < 1e-4                     # a 256 g rat


def _dense(m, qM):
    nv = int(m["nv"])
    M = np.zeros((nv, nv))
    for e, ij in enumerate(m["k_M_ij"]):
        M[ij & 0xFFFF, ij >> 16] = M[ij >> 16, ij & 0xFFFF] = qM[e]
    return M


def test_oracle_invariants(oracle_built):
    ref = oracle_built
    path = assets.asset_path("rodent_optimized")
    m = mjcf.load_blob(path)
    M = ref.RefModel(path, "f64")
    M.set_iterations(50, 50)
    d = ref.RefData(M)
    d.init(m["qpos0"].astype(np.float64), np.zeros(M.nv))
    for s in range(300):
        d.step(np.zeros(M.nu), 1)
    Md = _dense(m, d.get("qM"))
    assert np.linalg.eigvalsh(Md).min() > 0                      # mass matrix symmetric positive definite
    q = d.get("qpos")
    assert abs(np.linalg.norm(q[3:7]) - 1) < 1e-12 and np.all(np.isfinite(q))
    # standing on the floor: contact normal forces carry the weight (0.256 kg * 9.81)
    f = d.get("efc_force")[M.nlimit:].reshape(-1, 4).sum()
    assert abs(f - 0.25623 * 9.81) < 0.2 * 0.25623 * 9.81
    assert (d.get("con_dist") < 0).sum() >= 4
    # converged CG satisfies the optimality condition  M qacc - qfrc_smooth - J'f = 0
    grad = Md @ d.get("qacc") - d.get("qfrc_smooth") - d.get("qfrc_constraint")
    assert np.abs(grad).max() < 1e-4 * max(1.0, np.abs(d.get("qfrc_smooth")).max())


def test_oracle_energy_conservation_without_dissipation(oracle_built, tmp_path):
    """Contacts, damping, springs, actuation and gravity off: kinetic energy is conserved (first-order drift only)."""
    ref = oracle_built
    m = mjcf.load_blob(assets.asset_path("rodent_optimized"))
    m2 = dict(m)
    for k in ("dof_damping", "jnt_stiffness", "actuator_gainprm0", "actuator_biasprm", "opt_gravity"):
        m2[k] = m[k] * 0
    m2["jnt_range"] = m["jnt_range"] * 100
    p = str(tmp_path / "free.rrm")
    mjcf.save_blob(m2, p)
    M = ref.RefModel(p, "f64")
    d = ref.RefData(M)
    rng = np.random.default_rng(0)
    q = m["qpos0"].astype(np.float64); q[2] = 2.0; q[7:] += rng.uniform(-.2, .2, M.nq - 7)
    v = rng.uniform(-1, 1, M.nv) * np.r_[np.ones(6), 3 * np.ones(M.nv - 6)]
    d.init(q, v)
    ke = lambda: 0.5 * d.get("qvel") @ _dense(m, d.get("qM")) @ d.get("qvel")
    e0 = ke()
    for _ in range(200):
        d.step(np.zeros(M.nu), 1)
    d.forward()
    assert abs(ke() - e0) / e0 < 5e-3


def test_float32_divergence_is_inherent(oracle_built):
    """The chaos floor: the SAME scalar code in float32 vs float64 drifts apart over env-steps, so
    '1e-5 after 1000 steps' cannot hold for ANY float32 implementation (parity section)."""
    ref = oracle_built
    path = assets.asset_path("rodent_optimized")
    m = mjcf.load_blob(path)
    res = {}
    for prec in ("f64", "f32"):
        M = ref.RefModel(path, prec)
        M.set_iterations(8, 8)
        d = ref.RefData(M)
        d.init(m["qpos0"].astype(np.float64), np.zeros(M.nv))
        rng = np.random.default_rng(1)
        traj = []
        for s in range(60):
            d.step(rng.uniform(-1, 1, M.nu), 10)
            traj.append(d.get("qpos"))
        res[prec] = np.array(traj)
    gap = np.abs(res["f64"] - res["f32"]).max(axis=1)
    print("float32 vs float64 oracle, max |dqpos| after 1/10/30/60 env-steps:", gap[[0, 9, 29, 59]])
    assert gap[0] < 2e-3                     # one env-step (10 substeps): round-off, already amplified by joint limits
    assert np.all(np.isfinite(res["f32"]))
    assert gap[-1] > 1e-5                    # contact dynamics amplify it well past 1e-5 within 60 env-steps


def test_one_newton_iteration_diverges_on_the_oracle_too(oracle_built):
    """Round 2's record gpurun_out/nb_a_newton_1_4.err: `bench.py --solver newton --iterations 1 --ls-iterations 4` ended in
    "non-finite state in the rollout" on the GPU.  The same rollout (bench.py's reset keys, fresh U(-1,1) actions, Episode(150) +
    AutoReset) on the float64 ORACLE: within two env steps joint velocities pass 1e4 rad/s and states go non-finite -- one Newton step
    with a 4-iteration line search leaves an iterate that the next substeps amplify; it is the configuration that diverges, not the
    kernel's Hessian factorisation.  Newton 4/8 from the same states stays bounded.  `Rodent(iterations < 2)` warns (envs/rodent.py)."""
    from rodent_amd import jax_random as jr
    from tests import util
    from tests.oracle_env import OracleRodent
    N = 128
    keys = jr.split(jr.fold_in(jr.PRNGKey(0), 0), N)
    worst = {}
    for it in ((1, 4), (4, 8)):
        E = OracleRodent("rodent_optimized", N, "f64", it, util.synthetic_track(), episode_length=150)
        E.M.set_solver("newton")
        E.reset(keys)
        rng = np.random.default_rng(1234)
        w = 0.0
        for t in range(3):
            E.step(rng.uniform(-1, 1, (N, 30)))
            v = E.state()["qvel"]
            w = max(w, float(np.abs(np.where(np.isfinite(v), v, np.inf)).max()))
        worst[it] = w
    print("max |qvel| over 3 env steps of %d envs, float64 oracle:" % N, worst)
    assert worst[(1, 4)] > 1e4 and worst[(4, 8)] < 2e3
