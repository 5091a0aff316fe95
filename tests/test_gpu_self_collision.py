"""BASELINE config 1 on the PRODUCT path: rodent_cpu.xml [REF models/rodent_cpu.xml] through librodent_hip.so's DYN
instance -- 2243 candidate pairs of two moving geoms (sphere-sphere / sphere-capsule / capsule-capsule, condim 1 and 3) scanned per
substep, the pairs in penetration compacted into the wave's contact slots, J = jac(body2) - jac(body1) through signed dof chains, fixed
tendon transmissions, no free joint -- against the float64 oracle (tests/parity.py criteria; the oracle's primitives themselves are
checked in tests/test_self_collision_cpu.py)."""
import numpy as np
import pytest
import torch

from rodent_amd import assets, mjcf
from tests import parity, util
from tests.hip_impl import HipImpl, NoDiscrete

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _colliding_inputs(ref, n_envs, steps, seed):
    """(state, ctrl) pairs on rodent_cpu: random poses inside the joint limits, about half of them with sphere / capsule pairs in penetration."""
    path = assets.asset_path("rodent_cpu")
    tab = mjcf.load_blob(path)
    M = ref.RefModel(path, "f64")
    rng = np.random.default_rng(seed)
    lo, hi = tab["jnt_range"][:, 0], tab["jnt_range"][:, 1]
    pen, free = [], []
    d = ref.RefData(M)
    while len(pen) < n_envs * steps // 2 or len(free) < n_envs * steps // 2:
        q = (tab["qpos0"] + rng.uniform(0.2, 1.0) * rng.uniform(lo, hi)).astype(np.float64)
        d.init(q, np.zeros(M.nv))
        (pen if (d.get("con_dist") < 0).any() else free).append(q)
    seq, npen = [], 0
    for t in range(steps):
        qs = []
        for e in range(n_envs):
            src = pen if (e + t) % 2 == 0 and pen else free
            qs.append(src.pop())
        st = dict(qpos=np.asarray(qs), qvel=rng.uniform(-0.5, 0.5, (n_envs, M.nv)), act=rng.uniform(-0.5, 0.5, (n_envs, M.nu)),
                  qacc_warmstart=np.zeros((n_envs, M.nv)))
        seq.append(({k: parity.f32r(v) for k, v in st.items()}, parity.f32r(rng.uniform(-1, 1, (n_envs, M.nu)))))
    return seq, tab


def test_teacher_forced_substeps_with_self_collisions(oracle_built):
    N = 8
    seq, tab = _colliding_inputs(oracle_built, N, 40, seed=11)
    A = parity.OracleImpl("rodent_cpu", N, "f64", (8, 8))
    B = parity.OracleImpl("rodent_cpu", N, "f32", (8, 8))
    nact = sum(int((A.substep(st, c)["con_dist"] < 0).sum()) for st, c in seq[:10])
    assert nact >= 20                                       # the sample does exercise the contacts
    impl = HipImpl("rodent_cpu", N, (8, 8), False)
    out = parity.substep_ladder(NoDiscrete(impl, A), seq, A, B)
    print(out["quantiles"])
    parity.check_quantiles(out["quantiles"], parity.SUBSTEP_FLOORS)
    assert impl.batch.contact_overflow() == 0                # never more than 64 pairs in penetration: nothing was dropped


def test_contact_slot_overflow_is_counted(oracle_built):
    """A pose with every joint at the same end of its range folds the animal into itself: if more than 64 pairs penetrate, the surplus is
    dropped for that substep and the batch's overflow counter says so (the state stays finite either way)."""
    N = 4
    tab = mjcf.load_blob(assets.asset_path("rodent_cpu"))
    impl = HipImpl("rodent_cpu", N, (8, 8), False)
    M = oracle_built.RefModel(assets.asset_path("rodent_cpu"), "f64")
    d = oracle_built.RefData(M)
    worst, q_worst = 0, None
    for sgn in (0, 1):
        q = np.where(np.arange(67) % 2 == sgn, tab["jnt_range"][:, 0], tab["jnt_range"][:, 1]).astype(np.float64) * 0.98
        d.init(q, np.zeros(67))
        n = int((d.get("con_dist") < 0).sum())
        if n > worst:
            worst, q_worst = n, q
    st = dict(qpos=np.tile(q_worst, (N, 1)), qvel=np.zeros((N, 67)), act=np.zeros((N, 38)), qacc_warmstart=np.zeros((N, 67)))
    ds = impl._dev(st)
    impl.batch.pipeline_step(ds, torch.zeros(N, 38, device=DEV), 1)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in ds.values())
    assert (impl.batch.contact_overflow() > 0) == (worst > 64), (worst, impl.batch.contact_overflow())


def test_contacts_act_and_results_are_deterministic(oracle_built):
    """Started in colliding poses at rest, the penetrations shrink over a few substeps (the contact forces act with the right sign on BOTH
    bodies); two launches from the same state are bit-identical; pipeline_init + a 10-substep launch stay finite."""
    from rodent_amd import hip
    N = 8
    seq, tab = _colliding_inputs(oracle_built, N, 2, seed=12)
    impl = HipImpl("rodent_cpu", N, (8, 8), False)
    st = {k: v.copy() for k, v in seq[0][0].items()}
    st["qvel"][:] = 0
    A = parity.OracleImpl("rodent_cpu", N, "f64", (8, 8))
    A.b.set_state(st)
    for d_ in A.b.d:
        d_.forward()
    d0 = A.b.get("con_dist")
    ds = impl._dev(st)
    impl.batch.pipeline_init(ds)
    ctrl = torch.zeros(N, impl.batch.dims.nu, device=DEV)
    ds2 = {k: v.clone() for k, v in ds.items()}
    impl.batch.pipeline_step(ds, ctrl, 10)
    impl.batch.pipeline_step(ds2, ctrl, 10)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in ds.values()) and all(torch.equal(ds[k], ds2[k]) for k in ds)
    after = {k: v.cpu().numpy().astype(np.float64) for k, v in ds.items()}
    A.b.set_state(after)
    for d_ in A.b.d:
        d_.forward()
    d1 = A.b.get("con_dist")
    act = d0 < 0
    assert act.sum() >= 4 and (d1[act] > d0[act]).mean() > 0.7


def test_config1_env_step_four_envs(oracle_built):
    """`Rodent.step` with num_envs = 4 on rodent_cpu.xml through the C ABI: shapes, finiteness, and one env step against the oracle env."""
    from rodent_amd import envs
    track = util.synthetic_track()
    env = envs.get_environment("rodent", track_pos=track, num_envs=4, xml_path="rodent_cpu.xml", iterations=6, ls_iterations=6, device=DEV)
    state = env.reset(0)
    assert state.obs.shape == (4, 1244) and torch.isfinite(state.obs).all()
    A = parity.OracleEnvImpl("rodent_cpu", 4, "f64", (6, 6), track)
    B = parity.OracleEnvImpl("rodent_cpu", 4, "f32", (6, 6), track)
    g = torch.Generator(device=DEV).manual_seed(3)
    err, gap = [], []
    for t in range(8):
        a = torch.rand(4, 38, device=DEV, generator=g) * 2 - 1
        st = {k: getattr(state.pipeline_state, k).cpu().numpy().astype(np.float64) for k in parity.STATE}
        cf = state.info["cur_frame"].cpu().numpy()
        nstate = env.step(state, a)
        want, gp = A.env_step(st, a.cpu().numpy().astype(np.float64), cf), B.env_step(st, a.cpu().numpy().astype(np.float64), cf)
        assert nstate.obs.shape == (4, 1244) and torch.isfinite(nstate.obs).all() and torch.isfinite(nstate.reward).all()
        assert np.array_equal(nstate.info["cur_frame"].cpu().numpy(), want["cur_frame"])
        err.append(np.abs(nstate.pipeline_state.qpos.cpu().numpy() - want["qpos"]).max(1)); gap.append(np.abs(gp["qpos"] - want["qpos"]).max(1))
        state = nstate
    rows = parity.quantile_rows("qpos", np.concatenate(err), np.concatenate(gap), qs=(0.5,))
    print(rows)
    parity.check_quantiles(rows, parity.ENV_FLOORS)
