"""The oracles' forward pass against the reference's OWN stored `mjx.forward` output (tests/reference_pin.py: the 1260-value
observation of [NB Env_step.ipynb cell 8], qpos | qvel | cinert | cvel | qfrc_actuator after reset(PRNGKey(0))).

This is the only reference-held numeric pin of kinematics (A-1), com_pos / cinert (A-2), com_vel (A-6) and actuation (a17):
  * qvel is bit-exact threefry output (split(PRNGKey(0), 3)[2]) and qpos the notebook's reset_qpos  -> the state is the reference's;
  * cvel angular, all 65 bodies: the orientation chain, joint axes and dof order;
  * cvel linear, all 65 bodies, after ONE 3-vector (the COM shift of the notebook's XML variant, read from mass columns): the position
    chain, joint anchors, the about-the-root-subtree-COM convention;
  * cinert rows of the 35 bodies whose geoms the notebook's XML variant shares with rodent_optimized.xml or rodent_0.xml:
    inertia in world axes about that COM, m * off, m, packed [xx yy zz xy xz yz | m off | m];
  * qfrc_actuator: affine-bias actuator forces on the dofs.
The reference computed in float32 and printed ~8 digits, so agreement is asserted at 2e-6 of each segment's scale.
What this vector cannot pin: the mass matrix, bias forces, contacts, the solver and the integrator (a step() output is held nowhere)."""
import json
import os

import numpy as np
import pytest

from oracle import np_ref
from rodent_amd import assets, jax_random as jr, mjcf
from tests import reference_pin as rp

TOL = 2e-6


def _obs_from_fields(qpos, qvel, cinert, cvel, qfrc_actuator):
    return np.concatenate([qpos, qvel, np.asarray(cinert).reshape(-1, 10)[1:].ravel(), np.asarray(cvel).reshape(-1, 6)[1:].ravel(), qfrc_actuator])


def c_oracle_obs(ref, model_name, precision, qpos, qvel):
    M = ref.RefModel(assets.asset_path(model_name), precision)
    d = ref.RefData(M)
    d.init(qpos, qvel)
    return _obs_from_fields(qpos, qvel, d.get("cinert"), d.get("cvel"), d.get("qfrc_actuator"))


def np_ref_obs(model_name, qpos, qvel):
    m = np_ref.Model(mjcf.load_blob(assets.asset_path(model_name)))
    d = np_ref.Data(m)
    np_ref.init(m, d, qpos.copy(), qvel.copy())
    return _obs_from_fields(qpos, qvel, d.cinert, d.cvel, d.qfrc_actuator)


def check(results, names=None, tol=TOL, min_pinned=35):
    text, best = rp.table(results, names)
    print("\n" + text)
    for m, r in results.items():
        assert r["zero_pattern_qfrc_actuator"], m
        assert r["qfrc_actuator"] < tol and r["cvel_angular"] < tol and r["cvel_linear"] < tol, (m, r)
        assert r["delta_spread_m"] < 1e-7 and r["n_same_mass"] >= 20, (m, r["delta_spread_m"], r["n_same_mass"])
        rows = r["cinert_rows"][r["comparable"]]
        assert rows.size >= 26 and rows.max() < tol, (m, rows.max())
    pinned = ~np.isnan(best)
    assert pinned.sum() >= min_pinned and np.nanmax(best) < tol
    return text


def test_state_of_the_vector_is_the_reference_reset():
    """qvel = uniform(split(PRNGKey(0), 3)[2], (73,), -.01, .01) to the bit; qpos = the stored reset_qpos."""
    obs = rp.notebook_obs()
    u = jr.uniform(jr.split(jr.PRNGKey(0), 3)[2], 73, -0.01, 0.01)
    assert np.array_equal(u.astype(np.float32), obs[rp.SEG["qvel"]].astype(np.float32))
    q = np.asarray(json.load(open(os.path.join(rp.G, "env_step_reset.json")))["reset_qpos"])
    np.testing.assert_allclose(obs[rp.SEG["qpos"]], q, rtol=2e-7, atol=1e-9)      # two printouts of one float32 vector
    # the notebook data is self-consistent in the layout we read it in: sum_b m_b off_b = 0 about the subtree COM of the root
    cn, _, _ = rp.split(obs)
    assert np.abs(cn[:, 6:9].sum(0)).max() < 1e-8 * cn[:, 9].sum() / 1e-2


@pytest.mark.parametrize("impl", ["c_f64", "c_f32", "np_ref"])
def test_forward_pass_matches_the_notebook_vector(oracle_built, impl):
    obs = rp.notebook_obs()
    qpos, qvel = obs[rp.SEG["qpos"]], obs[rp.SEG["qvel"]]
    res = {}
    for name in rp.MODELS:
        mine = (np_ref_obs(name, qpos, qvel) if impl == "np_ref" else c_oracle_obs(oracle_built, name, impl[2:], qpos, qvel))
        res[name] = rp.compare(mine, obs)
    names = json.load(open(os.path.join(rp.G, "mjcf_contact_struct.json")))["link_names"]
    text = check(res, names, tol=TOL if impl != "c_f32" else 4e-6)
    if impl == "c_f64":
        # bodies the vector cannot pin, and why: stated, not hidden
        best = rp.table(res, names)[1]
        unpinned = [names[b] for b in range(65) if np.isnan(best[b])]
        assert unpinned == ["torso", "pelvis"] + ["vertebra_C%d" % i for i in range(2, 31) if i not in (9, 20)] + ["skull"]
