"""BASELINE config 1: `Rodent_Env_Brax.step()` with num_envs = 4 on rodent_cpu.xml, CPU (plumbing, no GPU).

rodent_cpu.xml [REF models/rodent_cpu.xml] has no free joint and no floor, 8 fixed tendons driving 8 of its 38 actuators, and
4271 self-collision pairs.  Round 3: the 2243 sphere / capsule pairs are compiled and collide on the CPU oracles (sphere-sphere,
sphere-capsule, capsule-capsule, condim 1 and 3; tests/test_self_collision_cpu.py); the 2028 pairs that involve an ellipsoid or a box
stay dropped (count kept in the blob).  What is exercised here: the MJCF compiler on that file (dims, tendon
transmission tables), the env reset / step arithmetic on the CPU oracle for 4 envs (shapes, finite outputs, determinism, joint limits
active), the tendon transmission in BOTH CPU formulations (C oracle: sparse entries; np_ref: dense moment matrix), and that the HIP
library loads this model for its candidate-pair (DYN) instance (the GPU side: tests/test_gpu_self_collision.py).  The env's `qpos[:3] = track_pos` and `q[2]`
health test act on hinge angles here, exactly as the reference code would."""
import os

import numpy as np
import pytest

from oracle import np_ref
from rodent_amd import assets, hip, mjcf
from tests import util
from tests.oracle_env import OracleRodent


def test_compiled_dims_and_tendon_tables():
    m = mjcf.load_blob(assets.asset_path("rodent_cpu"))
    got = {k: int(m[k]) for k in ("nbody", "nq", "nv", "nu", "ngeom", "nM", "ntendon", "ncon", "nlimit", "hip_supported")}
    assert got == dict(nbody=66, nq=67, nv=67, nu=38, ngeom=100, nM=696, ntendon=8, ncon=2243, nlimit=67, hip_supported=1)
    assert int(m["ncon"]) + int(m["ndropped_pairs"]) == 4271
    adr = m["actuator_momentadr"]
    assert (np.diff(adr)[:8] == [2, 2, 2, 3, 2, 2, 12, 12]).all() and (np.diff(adr)[8:] == 1).all()      # [REF models/rodent_cpu.xml:505-560]
    for u in range(6):          # lumbar / cervical tendons: coefficients sum to 1
        assert abs(m["actuator_moment_coef"][adr[u]:adr[u + 1]].sum() - 1) < 1e-6
    if os.path.exists("/root/reference/models/rodent_cpu.xml"):
        c = mjcf.compile_mjcf("/root/reference/models/rodent_cpu.xml", contacts="supported_only")
        assert c["_names"]["tendon"][0] == "lumbar_extend" and c["_names"]["actuator"][8] == "hip_L_supinate"
        with pytest.raises(ValueError, match="not supported"):
            mjcf.compile_mjcf("/root/reference/models/rodent_cpu.xml")                                   # strict mode names the gap


def test_env_step_four_envs_on_the_cpu(oracle_built):
    track = util.synthetic_track()
    runs = []
    for rep in range(2):
        env = OracleRodent("rodent_cpu", 4, "f32", (6, 6), track, episode_length=150)      # the env class' default solver setting
        obs = env.reset(0)
        assert obs.shape == (4, 1244) and np.isfinite(obs).all()
        rng = np.random.default_rng(0)
        for t in range(20):
            obs = env.step(rng.uniform(-1, 1, (4, 38)))
            assert obs.shape == (4, 1244) and np.isfinite(obs).all() and np.isfinite(env.reward).all()
        assert (env.cur_frame == env.cur_frame[0] * 0 + env.cur_frame).all() and env.done.shape == (4,)
        runs.append((obs.copy(), env.state()["qpos"].copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])           # deterministic
    assert np.abs(runs[0][1]).max() < 3.5                                                                # joint limits hold the pose


def test_tendon_transmission_in_both_formulations(oracle_built):
    ref = oracle_built
    st, M, tab = util.settled_states(ref, "rodent_cpu", 3, seed=5, iterations=(8, 8))
    m = np_ref.Model(tab, 8, 8)
    rng = np.random.default_rng(1)
    for e in range(3):
        ctrl = rng.uniform(-1, 1, M.nu)
        c = util.oracle_forward(ref, M, st, e, ctrl)
        d = np_ref.Data(m)
        d.qpos, d.qvel, d.act, d.qacc_warmstart = (st[k][e].copy() for k in ("qpos", "qvel", "act", "qacc_warmstart"))
        d.ctrl = ctrl.copy()
        np_ref.forward(m, d)
        for k in ("qfrc_actuator", "qfrc_bias", "qacc_smooth", "efc_D", "efc_aref", "qacc"):
            a, b = getattr(d, k), c.get(k)
            assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-30), k
        assert np.count_nonzero(c.get("qfrc_actuator")) > 38          # tendon actuators spread over several dofs


def test_hip_library_loads_the_model_as_a_candidate_pair_model():
    """Round 3: the blob carries kernel tables for the DYN instance (candidate pairs of two moving geoms, tendon transmissions); the
    host half of the C ABI loads it without a GPU.  The solvers / diagnostics that instance does not have are refused loudly."""
    m = hip.Model(assets.asset_path("rodent_cpu"), 6, 6)
    d = m.dims
    assert (d.nq, d.nv, d.nu, d.nbody, d.ncon, d.obs_dim) == (67, 67, 38, 66, 2243, 1244) and d.lds_bytes <= 20480
    tab = mjcf.load_blob(assets.asset_path("rodent_cpu"))
    assert int(tab["k_dyn"]) == 1 and tab["k_con_i"].shape == (2243, 8) and tab["k_con_f"].shape == (2243, 32)
    # signed chains: every id is a dof (bit 7 = body1's side) or the padding id nv; never a dof on both sides
    rows = tab["k_con_chain_rows"].view(np.uint32).reshape(-1, 10)[:-1]
    ids = np.stack([(rows[:, j // 4] >> (8 * (j % 4))) & 255 for j in range(40)], axis=1)
    nsig = tab["k_con_i"][:, 4]
    for c in (0, 500, 2242):
        used = ids[c, :nsig[c]]
        assert np.all((used & 127) < 67) and len(set((used & 127).tolist())) == nsig[c] and np.all(ids[c, nsig[c]:] == 67)
    with pytest.raises(RuntimeError, match="Newton"):
        hip.Model(assets.asset_path("rodent_cpu"), 6, 6, solver="newton")
