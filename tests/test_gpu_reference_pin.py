"""`rr_env_reset` (= pipeline_init + the obs epilogue [REF Rodent_Env_Brax.py:87-89,138-158]) on the MI355X against the reference's
own stored observation (tests/reference_pin.py; [NB Env_step.ipynb cell 8]): the HIP kernel's kinematics, subtree COM, cinert, cvel and
actuation pinned to the reference's mjx.forward output, through the C ABI, by the same comparison that pins the oracles on the CPU
(tests/test_reference_pin.py).  float32 kernel against float32 reference printed to ~8 digits: 4e-6 of each segment's scale."""
import json
import os

import numpy as np
import pytest
import torch

from tests import reference_pin as rp, util
from tests.test_reference_pin import check

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_reset_obs(model_name, qpos, qvel, n=4):
    from rodent_amd import assets, hip
    model = hip.Model(assets.asset_path(model_name), 8, 8)
    batch = hip.Batch(model, n, torch.device(DEV))
    st = batch.zeros_state()
    st["qpos"][:] = torch.tensor(qpos, dtype=torch.float32, device=DEV)
    st["qvel"][:] = torch.tensor(qvel, dtype=torch.float32, device=DEV)
    env = dict(track_pos=torch.tensor(util.synthetic_track(), dtype=torch.float32, device=DEV), cur_frame=torch.zeros(n, dtype=torch.int32, device=DEV),
               obs=torch.zeros(n, batch.dims.obs_dim, device=DEV))
    batch.env_reset(st, env)
    torch.cuda.synchronize()
    obs = env["obs"].cpu().numpy()
    assert np.isfinite(obs).all() and all(np.array_equal(obs[0], obs[i]) for i in range(1, n))
    return obs[0].astype(np.float64)


def test_env_reset_obs_matches_the_notebook_vector():
    nb = rp.notebook_obs()
    qpos, qvel = nb[rp.SEG["qpos"]], nb[rp.SEG["qvel"]]
    res = {}
    for name in rp.MODELS:
        obs = hip_reset_obs(name, qpos, qvel)
        # the first 147 entries ARE the state that went in (float32 of the notebook's numbers)
        assert np.array_equal(obs[:74].astype(np.float32), qpos.astype(np.float32)) and np.array_equal(obs[74:147].astype(np.float32), qvel.astype(np.float32))
        res[name] = rp.compare(obs[:rp.NB], nb)
    names = json.load(open(os.path.join(rp.G, "mjcf_contact_struct.json")))["link_names"]
    text = check(res, names, tol=4e-6)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, "reference_pin_hip.txt"), "w").write(text + "\n")


def test_contact_geometry_matches_the_notebook_contact_struct():
    """`rr_pipeline_init` at the state of [NB mjcf.ipynb cell 20]'s `Contact` struct (= the reset state of tests/golden/env_step_reset.json,
    see tests/test_known_answers.py::test_contact_geometry_pinned_to_the_notebook_struct): the HIP kernel's contact distances, points and
    frames of the 40 plane-capsule contacts against the reference's stored collision output, through the C ABI's contact outputs."""
    from rodent_amd import assets, hip, mjcf
    from tests.test_known_answers import check_contact_geometry_against_the_notebook
    G = os.path.join(os.path.dirname(__file__), "golden")
    qpos = np.asarray(json.load(open(os.path.join(G, "env_step_reset.json")))["reset_qpos"])
    golden = json.load(open(os.path.join(G, "mjcf_contact_struct.json")))
    tab = mjcf.load_blob(assets.asset_path("rodent_optimized"))
    batch = hip.Batch(hip.Model(assets.asset_path("rodent_optimized"), 8, 8), 2, torch.device(DEV))
    st = batch.zeros_state()
    st["qpos"][:] = torch.tensor(qpos, dtype=torch.float32, device=DEV)
    nc = batch.dims.ncon
    out = dict(contact_dist=torch.zeros(2, nc, device=DEV), contact_pos=torch.zeros(2, 3 * nc, device=DEV), contact_frame=torch.zeros(2, 9 * nc, device=DEV))
    batch.pipeline_init(st, out)
    torch.cuda.synchronize()
    got = check_contact_geometry_against_the_notebook(out["contact_dist"][0].cpu().numpy().astype(np.float64),
                                                      out["contact_pos"][0].cpu().numpy().astype(np.float64).reshape(-1, 3),
                                                      out["contact_frame"][0].cpu().numpy().astype(np.float64).reshape(-1, 3, 3), golden, tab, tol=5e-7, ftol=1e-6)
    print("HIP vs notebook Contact: max |ddist| %.1e, |dpos| %.1e, |dframe| %.1e" % got)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    open(os.path.join(out_dir, "reference_pin_contacts_hip.txt"), "w").write("HIP rr_pipeline_init vs [NB mjcf.ipynb cell 20] Contact, 40 plane-capsule contacts: max |ddist| %.2e m, |dpos| %.2e m, |dframe| %.2e\n" % got)
