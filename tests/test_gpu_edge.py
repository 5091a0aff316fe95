"""GPU edge cases of the HIP path against the oracle: no contact / every contact active / many limits active,
clamped and saturating inputs, odd batch sizes, n_frames variants, argument errors."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(N, model="rodent_optimized", it=(8, 8)):
    from rodent_amd import assets, hip
    return hip.Batch(hip.Model(assets.asset_path(model), *it), N, torch.device(DEV))


def _run_substep(ref, M, qpos, qvel, act, warm, ctrl):
    d = ref.RefData(M)
    d.set("qpos", qpos); d.set("qvel", qvel); d.set("act", act); d.set("qacc_warmstart", warm)
    d.step(ctrl, 1)
    return d


@pytest.mark.parametrize("scenario", ["airborne", "pressed_into_floor", "limits_violated"])
def test_contact_and_limit_extremes(scenario, oracle_built):
    from rodent_amd import assets, mjcf
    ref = oracle_built
    path = assets.asset_path("rodent_optimized")
    m = mjcf.load_blob(path)
    M = ref.RefModel(path, "f64"); M.set_iterations(8, 8)
    N = 5                                            # odd batch size on purpose
    rng = np.random.default_rng(3)
    q = np.tile(m["qpos0"].astype(np.float64), (N, 1))
    if scenario == "airborne":
        q[:, 2] = 1.0                                # no contact, solver sees only (possibly) limit rows
    elif scenario == "pressed_into_floor":
        q[:, 2] = 0.012                              # belly on the floor: (almost) every contact active
        q[:, 7:] += rng.uniform(-0.02, 0.02, (N, M.nq - 7))
    else:
        q[:, 7:] += rng.uniform(-1.5, 1.5, (N, M.nq - 7))   # far outside most joint ranges
        q[:, 2] = 0.5
    v = rng.uniform(-0.5, 0.5, (N, M.nv))
    act = rng.uniform(-1, 1, (N, M.nu)); warm = np.zeros((N, M.nv))
    ctrl = rng.uniform(-3, 3, (N, M.nu))             # outside ctrlrange: clamped to [-1, 1] by the actuator model
    b = _batch(N)
    st = {k: torch.tensor(x, dtype=torch.float32, device=DEV) for k, x in dict(qpos=q, qvel=v, act=act, qacc_warmstart=warm).items()}
    dbg = torch.zeros(N, b.dims.dbg_floats, device=DEV)
    b.pipeline_step(st, torch.tensor(ctrl, dtype=torch.float32, device=DEV), 1, out=dict(debug=dbg))
    torch.cuda.synchronize()
    lay = b.debug_layout()
    dbg = dbg.cpu().numpy().astype(np.float64)
    nact, nlim = [], []
    for e in range(N):
        d = _run_substep(ref, M, q[e], v[e], act[e], warm[e], ctrl[e])
        active = d.get("con_dist") < 0
        nact.append(int(active.sum())); nlim.append(int((d.get("efc_pos")[:M.nlimit] < 0).sum()))
        o, n = lay["con_dist"]
        assert np.array_equal(dbg[e, o:o + n] < 0, active)                       # same active set (integer-exact)
        for name, tol in (("qacc_smooth", 5e-3), ("qfrc_actuator", 1e-5), ("qM", 1e-4)):
            o, n = lay[name]
            want = d.get(name)
            assert np.abs(dbg[e, o:o + n] - want).max() <= tol * max(np.abs(want).max(), 1e-6), (scenario, name)
        qv = d.get("qvel")
        dv = np.abs(qv - v[e]).max()
        assert np.abs(st["qvel"][e].cpu().numpy() - qv).max() <= 5e-3 * max(dv, 1e-2), scenario
        assert np.abs(st["act"][e].cpu().numpy() - d.get("act")).max() < 1e-6
    if scenario == "airborne":
        assert max(nact) == 0
    if scenario == "pressed_into_floor":
        assert min(nact) >= 30
    if scenario == "limits_violated":
        assert min(nlim) >= 20
    assert torch.isfinite(st["qpos"]).all() and torch.isfinite(st["qvel"]).all()


def test_cur_frame_saturates_like_a_clamped_gather(oracle_built):
    """info['cur_frame'] is never reset by AutoReset, so it runs past the clip: JAX clamps the gather index."""
    from rodent_amd import envs
    track = util.synthetic_track(T=20)
    env = envs.get_environment("rodent", track_pos=track, num_envs=3, xml_path="rodent_optimized.xml", iterations=8,
                               ls_iterations=8, device=DEV, pipeline_outputs=True)
    s = env.reset(1)
    s.info["cur_frame"] = torch.tensor([18, 19, 400], dtype=torch.int32, device=DEV)
    n = env.step(s, torch.zeros(3, env.action_size, device=DEV))
    assert n.info["cur_frame"].tolist() == [19, 20, 401]
    xq = n.pipeline_state
    # local tracking vector uses track_pos[min(frame + 1, T - 1)] = the last row for all three envs
    want = torch.tensor(track[-1], dtype=torch.float32, device=DEV) - xq.qpos[:, :3]
    xm1 = xq.xmat[:, 1].reshape(3, 3, 3)
    torch.testing.assert_close(n.obs[:, -3:], torch.einsum("nij,nj->ni", xm1, want), rtol=1e-5, atol=1e-6)
    assert torch.isfinite(n.reward).all()


@pytest.mark.parametrize("n_frames", [1, 3, 10])
def test_n_frames_and_single_env(n_frames, oracle_built):
    from rodent_amd import assets, mjcf
    ref = oracle_built
    path = assets.asset_path("rodent_new")
    m = mjcf.load_blob(path)
    M = ref.RefModel(path, "f64"); M.set_iterations(6, 6)        # the env class' default solver setting
    q = m["qpos0"].astype(np.float64); q[2] = 0.05
    b = _batch(1, "rodent_new", (6, 6))
    st = dict(qpos=torch.tensor(q[None], dtype=torch.float32, device=DEV), qvel=torch.zeros(1, M.nv, device=DEV),
              act=torch.zeros(1, M.nu, device=DEV), qacc_warmstart=torch.zeros(1, M.nv, device=DEV))
    ctrl = np.linspace(-1, 1, M.nu)
    b.pipeline_step(st, torch.tensor(ctrl[None], dtype=torch.float32, device=DEV), n_frames)
    d = ref.RefData(M)
    d.set("qpos", q)
    d.step(ctrl, n_frames)
    assert np.abs(st["qpos"][0].cpu().numpy() - d.get("qpos")).max() < 1e-4


def test_argument_errors_are_reported_not_crashed():
    b = _batch(4)
    st = b.zeros_state()
    with pytest.raises(ValueError):
        b.pipeline_step(st, torch.zeros(3, b.dims.nu, device=DEV), 1)             # wrong batch size
    with pytest.raises(ValueError):
        b.pipeline_step(st, torch.zeros(4, b.dims.nu), 1)                          # host tensor
    with pytest.raises(RuntimeError, match="n_frames"):
        b.pipeline_step(st, torch.zeros(4, b.dims.nu, device=DEV), 0)
    bad = dict(st); bad["qpos"] = st["qpos"].double()
    with pytest.raises(ValueError):
        b.pipeline_init(bad)


def test_optional_outputs_all_null_and_all_set(oracle_built):
    """The debug-dump instance with every optional rr_outputs pointer NULL except the dump, and with all of them set: the
    kernel's re-read of its I/O block through the kernarg segment must equal the real parameter (dump field `kernarg_ok`),
    the states must be bit-identical, and the contact geometry outputs / pose outputs must equal the dump's fields."""
    from rodent_amd import assets
    ref = oracle_built
    N = 8
    b = _batch(N)
    st, M, m = util.settled_states(ref, "rodent_optimized", N, seed=8, iterations=(8, 8))
    ctrl = torch.tensor(np.random.default_rng(1).uniform(-1, 1, (N, M.nu)), dtype=torch.float32, device=DEV)
    mk = lambda: {k: torch.tensor(v, dtype=torch.float32, device=DEV).contiguous() for k, v in st.items()}
    d = b.dims
    dbg1 = torch.zeros(N, d.dbg_floats, device=DEV)
    s1 = mk()
    b.pipeline_step(s1, ctrl, 1, out=dict(debug=dbg1))
    full = dict(cinert=torch.zeros(N, d.nbody * 10, device=DEV), cvel=torch.zeros(N, d.nbody * 6, device=DEV),
                qfrc_actuator=torch.zeros(N, d.nv, device=DEV), xpos=torch.zeros(N, d.nbody * 3, device=DEV),
                xmat=torch.zeros(N, d.nbody * 9, device=DEV), subtree_com=torch.zeros(N, 3, device=DEV),
                debug=torch.zeros(N, d.dbg_floats, device=DEV), contact_dist=torch.zeros(N, d.ncon, device=DEV),
                contact_pos=torch.zeros(N, d.ncon * 3, device=DEV), contact_frame=torch.zeros(N, d.ncon * 9, device=DEV))
    s2 = mk()
    b.pipeline_step(s2, ctrl, 1, out=full)
    s3 = mk()
    b.pipeline_step(s3, ctrl, 1, out=dict(contact_dist=torch.zeros(N, d.ncon, device=DEV)))      # debug instance, no dump buffer
    s4 = mk()
    b.pipeline_step(s4, ctrl, 1)                                                                   # production instance, out = NULL
    torch.cuda.synchronize()
    lay = b.debug_layout()
    o, n = lay["kernarg_ok"]
    assert (dbg1[:, o] == 1).all() and (full["debug"][:, o] == 1).all()
    for k in s1:
        assert torch.equal(s1[k], s2[k]) and torch.equal(s1[k], s3[k]), k
        assert torch.isfinite(s4[k]).all()
    f = lambda name: full["debug"][:, lay[name][0]:lay[name][0] + lay[name][1]]
    assert torch.equal(full["contact_dist"], f("con_dist")) and torch.equal(full["contact_pos"], f("con_pos"))
    assert torch.equal(full["contact_frame"], f("con_frame"))
    assert torch.equal(full["xpos"], f("xpos")) and torch.equal(full["xmat"], f("xmat")) and torch.equal(full["cinert"], f("cinert"))
    assert torch.equal(full["cvel"], f("cvel")) and torch.equal(full["qfrc_actuator"], f("qfrc_actuator"))
    # contact geometry against the oracle (brax State.contact: dist, pos, frame)
    for e in range(N):
        dd = util.oracle_forward(ref, M, st, e, ctrl[e].cpu().numpy().astype(np.float64))
        assert np.abs(full["contact_dist"][e].cpu().numpy() - dd.get("con_dist")).max() < 2e-6
        assert np.abs(full["contact_pos"][e].cpu().numpy() - dd.get("con_pos")).max() < 2e-6
        assert np.abs(full["contact_frame"][e].cpu().numpy() - dd.get("con_frame")).max() < 2e-6


def test_env_contact_and_pipeline_outputs():
    """`Rodent(..., pipeline_outputs=True, contact_outputs=True)`: brax-style State.contact fields and `sys` id tables."""
    from rodent_amd import envs
    env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=4, xml_path="rodent_optimized.xml", iterations=8,
                               ls_iterations=8, device=DEV, pipeline_outputs=True, contact_outputs=True)
    r = env.reset(2)
    s = env.step(r, torch.zeros(4, env.action_size, device=DEV))
    ps = s.pipeline_state
    assert ps.contact_dist.shape == (4, 59) and ps.contact_pos.shape == (4, 59, 3) and ps.contact_frame.shape == (4, 59, 3, 3)
    assert torch.isfinite(ps.contact_dist).all() and ps.xpos.shape == (4, 66, 3)
    assert torch.allclose(ps.contact_frame[:, :, 0], torch.tensor([0.0, 0.0, 1.0], device=DEV).expand(4, 59, 3))     # floor normal
    assert env.sys.contact_geom2[-5:].tolist() == [1, 62, 75, 89, 97] and env.sys.contact_link_idx[0].tolist() == [-1] * 59
    c = env.contact(ps)                                   # brax State.contact, fields as in [NB mjcf.ipynb:917-921]
    assert c.geom2.dtype == torch.int32 and c.geom2[-5:].tolist() == [1, 62, 75, 89, 97] and c.geom1.tolist() == [0] * 59
    assert c.link_idx[1].tolist() == (env.sys.geom_bodyid[env.sys.contact_geom2] - 1).tolist()
    assert c.friction.shape == (59, 5) and torch.allclose(c.friction[0], torch.tensor([1.5, 1.5, 0.005, 1e-4, 1e-4], device=DEV))
    assert torch.allclose(c.solimp[0], torch.tensor([0.9, 0.95, 0.001, 0.5, 2.0], device=DEV)) and float(c.elasticity.abs().max()) == 0
    assert c.dist is ps.contact_dist
    lean = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=4, xml_path="rodent_optimized.xml", iterations=8,
                                ls_iterations=8, device=DEV)
    r2 = lean.reset(2)
    s2 = lean.step(r2, torch.zeros(4, env.action_size, device=DEV))
    assert s2.pipeline_state.cinert is None and s2.pipeline_state.contact_dist is None and torch.isfinite(s2.obs).all()
    # contact outputs are served by the debug-dump instance, the lean env by the production one: equal to rounding, which one
    # forward pass (reset) shows; ten substeps amplify it (tests/parity.py)
    assert torch.equal(r2.pipeline_state.qpos, r.pipeline_state.qpos) and torch.allclose(r2.obs, r.obs, atol=1e-5, rtol=1e-5)
