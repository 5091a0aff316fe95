"""The one full numeric vector of the forward pass the reference holds, and how an implementation is compared with it.

`tests/golden/env_step_obs.json` = the 1260 numbers [NB /root/reference/Env_step.ipynb cell 8] prints for `rodent_state.obs` after
`reset(PRNGKey(0))`: qpos(74) | qvel(73) | cinert[1:](65 x 10) | cvel[1:](65 x 6) | qfrc_actuator(73) -- the reference's OWN
`mjx.forward` output (float32) at a state whose qpos and qvel are in the same vector [REF Rodent_Env_Brax.py:87-89,138-158].

The notebook's XML (`params/params.yaml: XML_PATH`, not in the reference tree) is a VARIANT of the models that are: identical
kinematic tree, joints and actuators (qpos0, cvel's angular part and qfrc_actuator agree to float32 print precision), but other
geoms on some bodies (torso 0.0841 kg against 0.0612 in rodent_optimized.xml / 0.0306 in rodent_0.xml; tail ellipsoids C2..C30;
pelvis; skull), and the six lumbar spheres at density 500 instead of 1000.  Masses enter the vector in two ways, and both are
removed without fitting anything to the quantity under test:

  * cinert and the linear half of cvel are taken about the subtree COM of the root, which moves with the masses.  The COM shift
    `delta = c_notebook - c_model` is READ from the `m * off` columns of the bodies whose mass is the same in both (off = xipos - c, xipos
    being mass-independent): `delta = off_model - off_notebook`, one 3-vector, the same for every such body (their spread is the
    first thing checked).  cvel_lin(notebook) = cvel_lin(model) + omega x delta for ALL 65 bodies; a cinert row is moved by the
    parallel-axis rule.
  * a body whose geoms are the same at another density has its whole cinert row scaled by the mass ratio (1 or 0.5 here).

A body is COMPARABLE in a model iff its mass ratio notebook / model is 1 or 0.5 to 1e-6 (decided on the mass column alone); the
other bodies' cinert rows describe geoms this tree does not have and are listed, not compared.  Everything else in the vector
(all of cvel, qfrc_actuator) is compared for every body / dof.
"""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")
NB, NBODY, NV, NQ = 1260, 66, 73, 74
SEG = dict(qpos=slice(0, 74), qvel=slice(74, 147), cinert=slice(147, 797), cvel=slice(797, 1187), qfrc_actuator=slice(1187, 1260))
MODELS = ("rodent_optimized", "rodent_0")      # the 66-body single-rodent models under /root/reference/models


def notebook_obs():
    obs = np.asarray(json.load(open(os.path.join(G, "env_step_obs.json")))["obs"], np.float64)
    assert obs.size == NB
    return obs


def split(obs):
    obs = np.asarray(obs, np.float64)
    return obs[SEG["cinert"]].reshape(65, 10), obs[SEG["cvel"]].reshape(65, 6), obs[SEG["qfrc_actuator"]]


def _par(o):
    x, y, z = o[:, 0], o[:, 1], o[:, 2]
    return np.stack([y * y + z * z, x * x + z * z, x * x + y * y, -x * y, -x * z, -y * z], 1)


def shift_cinert(c, delta):
    """cinert rows [Ixx Iyy Izz Ixy Ixz Iyz | m*off | m] taken about a point p -> about p + delta (parallel axes)."""
    out = c.copy()
    m = c[:, 9:10]
    off = c[:, 6:9] / m
    off2 = off - delta
    out[:, :6] = c[:, :6] - m * _par(off) + m * _par(off2)
    out[:, 6:9] = m * off2
    return out


def compare(mine_obs, nb_obs=None):
    """mine_obs: the first 1260 observation entries of an implementation's forward pass at the notebook's (qpos, qvel) on ONE
    model.  Returns a dict of per-segment errors, each RELATIVE to the segment's (row's) scale in the notebook vector."""
    nb_obs = notebook_obs() if nb_obs is None else nb_obs
    cn, vn, an = split(nb_obs)
    c, v, a = split(mine_obs)
    out = {}
    out["qfrc_actuator"] = float(np.abs(a - an).max() / np.abs(an).max())
    out["zero_pattern_qfrc_actuator"] = bool(np.array_equal(a == 0, an == 0))
    out["cvel_angular"] = float(np.abs(v[:, :3] - vn[:, :3]).max() / np.abs(vn[:, :3]).max())
    ratio = cn[:, 9] / c[:, 9]
    same = np.abs(ratio - 1) < 1e-6
    half = np.abs(ratio - 0.5) < 1e-6
    deltas = c[same, 6:9] / c[same, 9:10] - cn[same, 6:9] / cn[same, 9:10]
    delta = deltas.mean(0)
    out["n_same_mass"] = int(same.sum())
    out["delta"] = delta
    out["delta_spread_m"] = float(np.abs(deltas - delta).max())
    pred = v[:, 3:] + np.cross(v[:, :3], delta)
    out["cvel_linear"] = float(np.abs(pred - vn[:, 3:]).max() / np.abs(vn[:, 3:]).max())
    comparable = same | half
    cs = shift_cinert(c, delta) * ratio[:, None]
    err = np.abs(cs - cn)
    row = np.maximum(err[:, :6].max(1) / np.abs(cn[:, :6]).max(1), err[:, 6:9].max(1) / np.abs(cn[:, 6:9]).max(1))
    out["cinert_rows"] = np.where(comparable, row, np.nan)
    out["comparable"] = comparable
    out["mass_ratio"] = ratio
    return out


def table(results, names=None):
    """results: {model: compare(...)}.  Text table: per segment and per body the error and the model that pins it."""
    lines = []
    for k in ("qfrc_actuator", "cvel_angular", "cvel_linear", "delta_spread_m"):
        lines.append("%-16s " % k + "  ".join("%s %.2e" % (m, r[k]) for m, r in results.items()))
    lines.append("COM shift notebook - model [m]: " + "  ".join("%s (%.3e %.3e %.3e)" % ((m,) + tuple(r["delta"])) for m, r in results.items()))
    best = np.full(65, np.nan)
    who = [""] * 65
    for m, r in results.items():
        for b in range(65):
            e = r["cinert_rows"][b]
            if not np.isnan(e) and not e >= best[b]:
                best[b], who[b] = e, "%s (mass ratio %.4f)" % (m, r["mass_ratio"][b])
    for b in range(65):
        nm = names[b] if names else "body %d" % (b + 1)
        if np.isnan(best[b]):
            lines.append("cinert %2d %-22s not comparable: notebook mass / model mass = " % (b + 1, nm)
                         + ", ".join("%.4f (%s)" % (r["mass_ratio"][b], m) for m, r in results.items()))
        else:
            lines.append("cinert %2d %-22s %.2e  %s" % (b + 1, nm, best[b], who[b]))
    return "\n".join(lines), best
