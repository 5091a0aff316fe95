"""Shared helpers for the parity tests: seeded states and oracle drivers."""
import numpy as np

from rodent_amd import assets, mjcf


def settled_states(ref, model_name, n, seed=0, settle_steps=30, noise=0.02, vel=0.3, iterations=None):
    """n random but physically plausible states: let the oracle rodent drop onto the floor, then perturb.

    Returns dict of float64 arrays (qpos, qvel, act, qacc_warmstart) [n, .] and the oracle model.
    """
    path = assets.asset_path(model_name)
    m = mjcf.load_blob(path)
    M = ref.RefModel(path, "f64")
    if iterations:
        M.set_iterations(*iterations)
    rng = np.random.default_rng(seed)
    d = ref.RefData(M)
    d.init(m["qpos0"].astype(np.float64), np.zeros(M.nv))
    base = []
    for s in range(settle_steps):
        d.step(rng.uniform(-0.3, 0.3, M.nu), 10)
        base.append((d.get("qpos"), d.get("qvel"), d.get("act"), d.get("qacc_warmstart")))
    out = dict(qpos=[], qvel=[], act=[], qacc_warmstart=[])
    for e in range(n):
        q, v, a, w = base[rng.integers(len(base) // 2, len(base))]
        q = q.copy()
        hinge = np.ones(q.size, bool)
        for r0 in range(0, q.size, 74):           # every replica starts with its 7 free-joint coordinates
            hinge[r0:r0 + 7] = False
            q[r0 + 2] += rng.uniform(-0.004, 0.01)
        q[hinge] += rng.uniform(-noise, noise, int(hinge.sum()))
        out["qpos"].append(q)
        out["qvel"].append(v + rng.uniform(-vel, vel, v.size))
        out["act"].append(np.clip(a + rng.uniform(-0.2, 0.2, a.size), -1, 1))
        out["qacc_warmstart"].append(w * rng.uniform(0.5, 1.5))
    return {k: np.asarray(v) for k, v in out.items()}, M, m


def oracle_forward(ref, M, st, e, ctrl):
    d = ref.RefData(M)
    d.set("qpos", st["qpos"][e]); d.set("qvel", st["qvel"][e]); d.set("act", st["act"][e])
    d.set("qacc_warmstart", st["qacc_warmstart"][e]); d.set("ctrl", ctrl)
    d.forward()
    return d


def synthetic_track(T=250):
    """straight line x = 0.2 m/s * t * 0.02 s, y = 0, z = torso rest height."""
    t = np.arange(T, dtype=np.float64)
    return np.stack([0.004 * t, np.zeros(T), np.full(T, 0.0681)], axis=1)


def f32_class_stat(errs, gaps):
    """REPORTED statistic only (round 1 used it as the gate and loosened it twice after GPU failures -- DESIGN.md section 2;
    the gates are now the criteria of tests/parity.py): geometric mean of (err + 1e-7) / (gap + 1e-7)."""
    errs, gaps = np.asarray(errs, np.float64), np.asarray(gaps, np.float64)
    return float(np.exp(np.mean(np.log((errs + 1e-7) / (gaps + 1e-7)))))
