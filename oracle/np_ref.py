"""np_ref.py -- SECOND, INDEPENDENT CPU restatement of the rodent physics step (test infrastructure only).

Purpose: pin `oracle/rodent_ref.c` (and the model compiler's `mj_setConst` constants) by a different
formulation of the same physics, so that an error shared by the C oracle and the HIP kernel -- both
follow MuJoCo's composite-rigid-body / spatial-vector-about-the-subtree-COM organisation -- cannot hide.
Pinned to the reference only for the forward pass up to the observation (kinematics, COM, cinert, cvel, actuation: the stored
`mjx.forward` vector of [NB Env_step.ipynb cell 8], tests/test_reference_pin.py); for everything past it PARITY STAYS UNPINNED
AGAINST THE REFERENCE ITSELF: mujoco / mujoco-mjx / brax are absent; this file pins the oracle against textbook
rigid-body dynamics, not against MJX.

What is formulated differently here (float64 numpy, dense everywhere, as MJX does with
`opt.jacobian = 0` [REF Rodent_Env_Brax.py:49]):
  * kinematics with rotation matrices and Rodrigues' formula (no quaternion algebra on the tree);
  * the mass matrix from per-body Jacobians,  M = sum_b m_b Jp_b' Jp_b + Jr_b' I_b Jr_b + diag(armature)
    (no composite inertias, no cdof about the subtree COM);
  * the bias force from the classical recursive Newton-Euler equations in world coordinates (angular
    velocity / acceleration of each body, classical acceleration of its COM) projected with the same
    Jacobians (no spatial cross products);
  * dense Cholesky solves (scipy), dense constraint Jacobian from contact-point Jacobians;
  * cinert / cvel (only needed for the observation) from their definitions;
  * its own `mj_setConst`: dof_invweight0, body_invweight0, stat.meaninertia from M(qpos0)^-1.
The contact geometry, the impedance formulas (A-5) and the CG solver (A-7) are
specifications rather than derivations; they are restated from the appendix.

Inputs are the MuJoCo-named tables of a compiled model blob (`rodent_amd.mjcf.load_blob`; float32 on
disk, used as float64 here exactly like the C oracle does); none of the kernel tables (`k_*`) is read.
Only tests/ and tools/make_step_golden.py import this module.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cho_factor, cho_solve

FREE, HINGE = 0, 3
MINVAL, MINIMP, MAXIMP = 1e-15, 1e-4, 0.9999


def _quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _rodrigues(u, ang):
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


class Model:
    def __init__(self, tables, iterations=None, ls_iterations=None):
        t = {k: (np.asarray(v, np.float64) if np.asarray(v).dtype.kind == "f" else np.asarray(v)) for k, v in tables.items()
             if not k.startswith("k_") and not k.startswith("_")}
        self.t = t
        for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "ncon", "nlimit", "nefc", "obs_dim"):
            setattr(self, k, int(t[k]))
        self.dt = float(t["opt_timestep"])
        self.gravity = t["opt_gravity"].copy()
        self.tolerance = float(t["opt_tolerance"]); self.ls_tolerance = float(t["opt_ls_tolerance"])
        self.impratio = float(t["opt_impratio"]); self.meaninertia = float(t["stat_meaninertia"])
        self.iterations = int(t["opt_iterations"]) if iterations is None else iterations
        self.ls_iterations = int(t["opt_ls_iterations"]) if ls_iterations is None else ls_iterations
        self.solver = "cg"          # or "newton" [REF Rodent_Env_Brax.py:42-45]
        nb, nv = self.nbody, self.nv
        # dof d moves body b  <=>  body(d) is b or an ancestor of b
        par = t["body_parentid"]
        self.anc = np.zeros((nb, nv), bool)
        for b in range(1, nb):
            a = b
            while a > 0:
                da, dn = t["body_dofadr"][a], t["body_dofnum"][a]
                if dn > 0:
                    self.anc[b, da:da + dn] = True
                a = par[a]
        self.bodyR0 = np.array([_quat_to_mat(q / np.linalg.norm(q)) for q in t["body_quat"]])
        self.bodyRi = np.array([_quat_to_mat(q / np.linalg.norm(q)) for q in t["body_iquat"]])
        self.geomR0 = np.array([_quat_to_mat(q / np.linalg.norm(q)) for q in t["geom_quat"]])
        n = np.linalg.norm(t["jnt_axis"], axis=1, keepdims=True)
        t["jnt_axis"] = t["jnt_axis"] / np.where(n > MINVAL, n, 1.0)     # unit to double precision, as the C oracle does at load


class Data:
    def __init__(self, m: Model):
        self.qpos = m.t["qpos0"].copy()
        self.qvel = np.zeros(m.nv)
        self.act = np.zeros(m.na)
        self.ctrl = np.zeros(m.nu)
        self.qacc_warmstart = np.zeros(m.nv)


# ----------------------------------------------------------------------------------------- position stage
def kinematics(m: Model, d: Data):
    """World poses by matrix recursion; per dof: world axis u, a point c on the axis (hinge) and its kind."""
    t, nb, nv = m.t, m.nbody, m.nv
    xpos = np.zeros((nb, 3)); xmat = np.zeros((nb, 3, 3)); xmat[0] = np.eye(3)
    d.dof_axis = np.zeros((nv, 3)); d.dof_pt = np.zeros((nv, 3)); d.dof_kind = np.zeros(nv, int)   # 0 translation, 1 rotation
    for b in range(1, nb):
        p = t["body_parentid"][b]
        pos = xpos[p] + xmat[p] @ t["body_pos"][b]
        R = xmat[p] @ m.bodyR0[b]
        for k in range(t["body_jntnum"][b]):
            j = t["body_jntadr"][b] + k
            qa, da = t["jnt_qposadr"][j], t["jnt_dofadr"][j]
            if t["jnt_type"][j] == FREE:
                pos = d.qpos[qa:qa + 3].copy()
                q = d.qpos[qa + 3:qa + 7]
                R = _quat_to_mat(q / np.linalg.norm(q))
                for i in range(3):
                    d.dof_axis[da + i] = np.eye(3)[i]; d.dof_kind[da + i] = 0
                    d.dof_axis[da + 3 + i] = R[:, i]; d.dof_pt[da + 3 + i] = pos; d.dof_kind[da + 3 + i] = 1
            else:
                c = pos + R @ t["jnt_pos"][j]
                u = R @ t["jnt_axis"][j]
                d.dof_axis[da] = u; d.dof_pt[da] = c; d.dof_kind[da] = 1
                R = R @ _rodrigues(t["jnt_axis"][j], d.qpos[qa] - t["qpos0"][qa])
                pos = c - R @ t["jnt_pos"][j]
        xpos[b], xmat[b] = pos, R
    d.xpos, d.xmat = xpos, xmat
    d.xipos = xpos + np.einsum("bij,bj->bi", xmat, t["body_ipos"])
    d.ximat = np.einsum("bij,bjk->bik", xmat, m.bodyRi)
    gb = t["geom_bodyid"]
    d.geom_xpos = xpos[gb] + np.einsum("gij,gj->gi", xmat[gb], t["geom_pos"])
    d.geom_xmat = np.einsum("gij,gjk->gik", xmat[gb], m.geomR0)


def point_jac(m: Model, d: Data, body: int, p):
    """3 x nv translational Jacobian of the point p (world) moving with `body`."""
    mask = m.anc[body]
    rot = np.cross(d.dof_axis, p[None, :] - d.dof_pt)                      # u x (p - c)
    J = np.where(d.dof_kind[:, None] == 1, rot, d.dof_axis)
    return (J * mask[:, None]).T


def body_jacobians(m: Model, d: Data):
    """Jp[b] (COM translation), Jr[b] (rotation): [nb, 3, nv]."""
    nb, nv = m.nbody, m.nv
    rot = np.cross(d.dof_axis[None, :, :], d.xipos[:, None, :] - d.dof_pt[None, :, :])      # [nb, nv, 3]
    isrot = (d.dof_kind == 1)[None, :, None]
    Jp = np.where(isrot, rot, d.dof_axis[None, :, :]) * m.anc[:, :, None]
    Jr = np.where(isrot, d.dof_axis[None, :, :], 0.0) * m.anc[:, :, None]
    return Jp.transpose(0, 2, 1), Jr.transpose(0, 2, 1)


def mass_matrix(m: Model, d: Data):
    t = m.t
    d.Jp, d.Jr = body_jacobians(m, d)
    d.Iw = np.einsum("bij,bj,bkj->bik", d.ximat, t["body_inertia"], d.ximat)                  # world-frame body inertias
    M = np.einsum("b,bid,bie->de", t["body_mass"], d.Jp, d.Jp) + np.einsum("bid,bij,bje->de", d.Jr, d.Iw, d.Jr)
    d.M = M + np.diag(t["dof_armature"])
    d.Mchol = cho_factor(d.M)


def com_terms(m: Model, d: Data):
    """subtree COM of every kinematic tree root; cinert (MuJoCo 10-vector) about it."""
    t = m.t
    root = t["body_rootid"]
    mass = t["body_mass"]
    d.com_root = np.zeros((m.nbody, 3))
    for r in np.unique(root[1:]):
        sel = (root == r) & (np.arange(m.nbody) > 0)
        d.com_root[r] = (mass[sel, None] * d.xipos[sel]).sum(0) / mass[sel].sum()
    off = d.xipos - d.com_root[root]
    I = d.Iw + mass[:, None, None] * (np.einsum("bi,bi->b", off, off)[:, None, None] * np.eye(3) - np.einsum("bi,bj->bij", off, off))
    d.cinert = np.concatenate([np.stack([I[:, 0, 0], I[:, 1, 1], I[:, 2, 2], I[:, 0, 1], I[:, 0, 2], I[:, 1, 2]], 1),
                               mass[:, None] * off, mass[:, None]], axis=1)
    d.cinert[0] = 0


# ----------------------------------------------------------------------------------------- velocity stage
def newton_euler_bias(m: Model, d: Data):
    """Classical recursive Newton-Euler with qacc = 0: angular velocity / acceleration of every body and the classical
    acceleration of its frame origin, carried joint by joint (a joint's anchor is a point fixed in the frames on both of
    its sides); gravity enters as a force.  Returns qfrc_bias; leaves omega, v (frame origin) for cvel."""
    t, nb = m.t, m.nbody
    om = np.zeros((nb, 3)); al = np.zeros((nb, 3)); v = np.zeros((nb, 3)); a = np.zeros((nb, 3))
    for b in range(1, nb):
        p = t["body_parentid"][b]
        # start: a point fixed in the parent frame, at the parent's origin
        w, dw, pt, vp, ap = om[p].copy(), al[p].copy(), d.xpos[p].copy(), v[p].copy(), a[p].copy()

        def move(to):          # same rigid frame, another point
            nonlocal pt, vp, ap
            r = to - pt
            vp = vp + np.cross(w, r)
            ap = ap + np.cross(dw, r) + np.cross(w, np.cross(w, r))
            pt = to.copy()

        for k in range(t["body_jntnum"][b]):
            j = t["body_jntadr"][b] + k
            da = t["jnt_dofadr"][j]
            if t["jnt_type"][j] == FREE:
                # world-frame translation coordinates, body-frame angular velocity: no bias acceleration of either kind
                pt = d.xpos[b].copy(); vp = d.qvel[da:da + 3].copy(); ap = np.zeros(3)
                w = d.xmat[b] @ d.qvel[da + 3:da + 6]; dw = np.zeros(3)
            else:
                u, c = d.dof_axis[da], d.dof_pt[da]
                move(c)                                     # the anchor, as a point of the frame before the joint
                dw = dw + np.cross(w, u) * d.qvel[da]      # d/dt (u qdot) with u carried by the frame before the joint
                w = w + u * d.qvel[da]                      # ... and now pt = c is a point of the frame after the joint
        move(d.xpos[b])
        om[b], al[b], v[b], a[b] = w, dw, vp, ap
    d.omega, d.vorigin = om, v
    # COM accelerations and the Newton / Euler equations, projected on the generalised coordinates
    rc = d.xipos - d.xpos
    ac = a + np.cross(al, rc) + np.cross(om, np.cross(om, rc))
    mass = t["body_mass"]
    F = mass[:, None] * (ac - m.gravity[None, :])
    Iom = np.einsum("bij,bj->bi", d.Iw, om)
    N = np.einsum("bij,bj->bi", d.Iw, al) + np.cross(om, Iom)
    return np.einsum("bid,bi->d", d.Jp, F) + np.einsum("bid,bi->d", d.Jr, N)


def com_vel(m: Model, d: Data):
    """cvel [omega; velocity of the body-fixed point that coincides with the tree's COM] (needs newton_euler_bias)."""
    com = d.com_root[m.t["body_rootid"]]
    d.cvel = np.concatenate([d.omega, d.vorigin + np.cross(d.omega, com - d.xpos)], axis=1)
    d.cvel[0] = 0


def smooth_forces(m: Model, d: Data):
    t = m.t
    d.qfrc_bias = newton_euler_bias(m, d)
    com_vel(m, d)
    passive = -t["dof_damping"] * d.qvel
    for j in range(m.njnt):
        if t["jnt_type"][j] == HINGE:
            qa, da = t["jnt_qposadr"][j], t["jnt_dofadr"][j]
            passive[da] -= t["jnt_stiffness"][j] * (d.qpos[qa] - t["qpos_spring"][qa])
    d.qfrc_passive = passive
    c = np.clip(d.ctrl, t["actuator_ctrlrange"][:, 0], t["actuator_ctrlrange"][:, 1])
    d.act_dot = (c - d.act) / np.maximum(t["actuator_dynprm0"], MINVAL)
    # transmission: dense moment matrix [nu, nv] built from the sparse entries (joint: gear 1; fixed tendon: its coefficients)
    moment = np.zeros((m.nu, m.nv)); moment_q = np.zeros((m.nu, m.nq))
    for u in range(m.nu):
        for e in range(t["actuator_momentadr"][u], t["actuator_momentadr"][u + 1]):
            moment[u, t["actuator_moment_dofadr"][e]] += t["actuator_moment_coef"][e]
            moment_q[u, t["actuator_moment_qposadr"][e]] += t["actuator_moment_coef"][e]
    length, vel = moment_q @ d.qpos, moment @ d.qvel
    force = t["actuator_gainprm0"] * d.act + t["actuator_biasprm"][:, 0] + t["actuator_biasprm"][:, 1] * length \
        + t["actuator_biasprm"][:, 2] * vel
    d.qfrc_actuator = moment.T @ force
    d.qfrc_smooth = d.qfrc_passive - d.qfrc_bias + d.qfrc_actuator
    d.qacc_smooth = cho_solve(d.Mchol, d.qfrc_smooth)


# ----------------------------------------------------------------------------------------- constraints
def _make_frame_default(n):
    y = np.array([0.0, 1.0, 0.0]) if -0.5 < n[1] < 0.5 else np.array([0.0, 0.0, 1.0])
    y = y - n * (n @ y)
    return y / np.linalg.norm(y)


def _make_frame(n):
    b = _make_frame_default(n)
    return np.stack([n, b, np.cross(n, b)])


def _segment_point(a, b, pt):
    """[UP mjx math.closest_segment_point]"""
    ab = b - a
    return a + np.clip((pt - a) @ ab / (ab @ ab + 1e-6), 0.0, 1.0) * ab


def _segment_segment(a0, a1, b0, b1):
    """[UP mjx math.closest_segment_to_segment_points] -- restated with the segments as (mid-point, unit direction, half length)"""
    la, lb = np.linalg.norm(a1 - a0), np.linalg.norm(b1 - b0)
    ua, ub = (a1 - a0) / la, (b1 - b0) / lb
    am, bm = 0.5 * (a0 + a1), 0.5 * (b0 + b1)
    tr = am - bm
    c = ua @ ub
    ta = (-(ua @ tr) + c * (ub @ tr)) / (1 - c * c + 1e-6)
    tb = ub @ tr + ta * c
    pa = am + ua * np.clip(ta, -0.5 * la, 0.5 * la)
    pb = bm + ub * np.clip(tb, -0.5 * lb, 0.5 * lb)
    na, nb = _segment_point(a0, a1, pb), _segment_point(b0, b1, pa)
    if (na - pb) @ (na - pb) < (nb - pa) @ (nb - pa):
        return na, pb
    return pa, nb


def _two_spheres(p1, r1, p2, r2):
    """[UP mjx collision_primitive._sphere_sphere]"""
    v = p2 - p1
    ln = np.linalg.norm(v)
    n = v / ln if ln > 0 else np.array([1.0, 0.0, 0.0])
    dist = ln - (r1 + r2)
    return dist, p1 + n * (r1 + 0.5 * dist), _make_frame(n)


def collision(m: Model, d: Data):
    """plane-sphere / plane-capsule end caps / plane-ellipsoid, and (8(f)-4) sphere-sphere / sphere-capsule /
    capsule-capsule between two moving bodies; kinds as in the blob's con_kind."""
    t, nc = m.t, m.ncon
    d.con_dist = np.zeros(nc); d.con_pos = np.zeros((nc, 3)); d.con_frame = np.zeros((nc, 3, 3))
    for c in range(nc):
        g1, g2, kind = t["con_geom1"][c], t["con_geom2"][c], t["con_kind"][c]
        if kind >= 4:
            P1, P2, s1, s2 = d.geom_xpos[g1], d.geom_xpos[g2], t["geom_size"][g1], t["geom_size"][g2]
            ax1, ax2 = d.geom_xmat[g1][:, 2], d.geom_xmat[g2][:, 2]
            a, b = P1, P2
            if kind == 5:
                b = _segment_point(P2 - ax2 * s2[1], P2 + ax2 * s2[1], P1)
            elif kind == 6:
                a, b = _segment_segment(P1 - ax1 * s1[1], P1 + ax1 * s1[1], P2 - ax2 * s2[1], P2 + ax2 * s2[1])
            d.con_dist[c], d.con_pos[c], d.con_frame[c] = _two_spheres(a, s1[0], b, s2[0])
            continue
        n = d.geom_xmat[g1][:, 2]
        pp = d.geom_xpos[g1]
        G, gp, size = d.geom_xmat[g2], d.geom_xpos[g2], t["geom_size"][g2]
        if kind == 3:
            s = (G.T @ n) * size
            s = -s / np.linalg.norm(s)
            pt = gp + G @ (s * size)
            dist = n @ (pt - pp)
            pos = pt - n * dist * 0.5
            b = _make_frame_default(n)
        else:
            ctr = gp.copy()
            if kind == 0:
                b = _make_frame_default(n)
            else:
                ax = G[:, 2]
                b = ax - n * (n @ ax)
                bn = np.linalg.norm(b)
                if bn < 0.5:       # capsule (nearly) along the normal: the raw y / z axis, not re-orthogonalised (A-4)
                    b = np.array([0.0, 1.0, 0.0]) if -0.5 < n[1] < 0.5 else np.array([0.0, 0.0, 1.0])
                else:
                    b = b / bn
                ctr = ctr + (1.0 if kind == 1 else -1.0) * ax * size[1]
            dist = (ctr - pp) @ n - size[0]
            pos = ctr - n * (size[0] + 0.5 * dist)
        d.con_dist[c], d.con_pos[c] = dist, pos
        d.con_frame[c] = np.stack([n, b, np.cross(n, b)])


def _kbi(m: Model, solref, solimp, pos):
    tc = max(solref[0], 2 * m.dt)
    dmin, dmax = np.clip(solimp[0], MINIMP, MAXIMP), np.clip(solimp[1], MINIMP, MAXIMP)
    width = max(MINVAL, solimp[2]); mid = np.clip(solimp[3], MINIMP, MAXIMP); power = max(1.0, solimp[4])
    k = 1 / (dmax * dmax * tc * tc * solref[1] * solref[1])
    b = 2 / (dmax * tc)
    if solref[0] <= 0:
        k = -solref[0] / (dmax * dmax)
    if solref[1] <= 0:
        b = -solref[1] / dmax
    x = abs(pos) / width
    y = x ** power / mid ** (power - 1) if x < mid else 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
    imp = np.clip(dmin + y * (dmax - dmin), dmin, dmax)
    if x > 1:
        imp = dmax
    return k, b, imp


def make_constraint(m: Model, d: Data):
    t, nv = m.t, m.nv
    J, D, aref, pos_ = [], [], [], []
    for l in range(m.nlimit):
        j = t["limit_jnt"][l]
        qa, da = t["jnt_qposadr"][j], t["jnt_dofadr"][j]
        q = d.qpos[qa]
        lo, hi = q - t["jnt_range"][j, 0], t["jnt_range"][j, 1] - q
        pos = min(lo, hi)
        row = np.zeros(nv)
        if pos < 0:
            row[da] = 1.0 if lo < hi else -1.0
        k, b, imp = _kbi(m, t["jnt_solref"][j], t["jnt_solimp"][j], pos)
        r = max(t["dof_invweight0"][da] * (1 - imp) / imp, MINVAL)
        J.append(row); D.append(1 / r); aref.append(-b * (row @ d.qvel) - k * imp * pos); pos_.append(pos)
    for c in range(m.ncon):
        body, dist = t["con_body2"][c], d.con_dist[c]
        mu = t["con_friction"][c, 0]
        Jc = np.zeros((3, nv))
        if dist < 0:       # J = jac(body2, pos) - jac(body1, pos); body1 = world for the floor contacts
            Jc = d.con_frame[c] @ point_jac(m, d, body, d.con_pos[c])
            if t["con_body1"][c] > 0:
                Jc = Jc - d.con_frame[c] @ point_jac(m, d, t["con_body1"][c], d.con_pos[c])
        k, b, imp = _kbi(m, t["con_solref"][c], t["con_solimp"][c], dist)
        frictionless = int(t["con_dim"][c]) == 1 if "con_dim" in t else False         # condim 1: one row along the normal
        invw = t["con_invweight"][c] if frictionless else (t["con_invweight"][c] + mu * mu * t["con_invweight"][c]) * 2 * mu * mu / m.impratio
        r = max(invw * (1 - imp) / imp, MINVAL)
        for row in ((Jc[0],) if frictionless else (Jc[0] + mu * Jc[1], Jc[0] - mu * Jc[1], Jc[0] + mu * Jc[2], Jc[0] - mu * Jc[2])):
            J.append(row); D.append(1 / r); aref.append(-b * (row @ d.qvel) - k * imp * dist); pos_.append(dist)
    d.efc_J, d.efc_D, d.efc_aref, d.efc_pos = np.array(J), np.array(D), np.array(aref), np.array(pos_)


# ----------------------------------------------------------------------------------------- solver (App. A-7)
def solve(m: Model, d: Data):
    nv = m.nv
    J, Dv, aref, M = d.efc_J, d.efc_D, d.efc_aref, d.M
    scale = 1 / (m.meaninertia * max(1, nv))

    class Ctx:
        pass

    def update_constraint(c):
        active = c.Jaref < 0
        c.force = np.where(active, -Dv * c.Jaref, 0.0)
        c.qfrc_constraint = J.T @ c.force
        c.gauss = 0.5 * (c.Ma - d.qfrc_smooth) @ (c.qacc - d.qacc_smooth)
        c.prev_cost = c.cost
        c.cost = 0.5 * np.sum(Dv * c.Jaref ** 2 * active) + c.gauss

    def update_gradient(c):
        c.grad = c.Ma - d.qfrc_smooth - c.qfrc_constraint
        if m.solver == "newton":        # H = M + J' diag(D active) J  [UP mjx solver._update_gradient]
            act = c.Jaref < 0
            H = M + (J[act].T * Dv[act]) @ J[act]
            c.Mgrad = cho_solve(cho_factor(H), c.grad)
        else:
            c.Mgrad = cho_solve(d.Mchol, c.grad)

    def create(qacc, grad=True):
        c = Ctx()
        c.qacc = qacc.copy(); c.Jaref = J @ qacc - aref; c.Ma = M @ qacc
        c.cost, c.prev_cost = np.inf, 0.0
        update_constraint(c)
        if grad:
            update_gradient(c)
            c.search = -c.Mgrad
        return c

    def linesearch(c):
        smag = np.linalg.norm(c.search) * m.meaninertia * max(1, nv)
        gtol = m.tolerance * m.ls_tolerance * smag
        mv = M @ c.search
        jv = J @ c.search
        qg = np.array([c.gauss, c.search @ c.Ma - c.search @ d.qfrc_smooth, 0.5 * c.search @ mv])
        quad = np.stack([0.5 * c.Jaref ** 2 * Dv, jv * c.Jaref * Dv, 0.5 * jv ** 2 * Dv], 1)

        def point(alpha):
            q = qg + quad[(c.Jaref + alpha * jv) < 0].sum(0)
            return (alpha, alpha * alpha * q[2] + alpha * q[1] + q[0], 2 * alpha * q[2] + q[1], 2 * q[2] + (MINVAL if q[2] == 0 else 0.0))

        p0 = point(0.0)
        lo = point(p0[0] - p0[2] / p0[3])
        if lo[2] < p0[2]:
            hi = p0
        else:
            hi, lo = lo, p0
        swap, it = True, 0
        while True:
            done = it >= m.ls_iterations or not swap or (lo[2] < 0 and lo[2] > -gtol) or (hi[2] > 0 and hi[2] < gtol)
            if done:
                break
            lo_next, hi_next, mid = point(lo[0] - lo[2] / lo[3]), point(hi[0] - hi[2] / hi[3]), point(0.5 * (lo[0] + hi[0]))
            s1 = lo[2] > 0 or lo[2] < lo_next[2]
            if s1:
                lo = lo_next
            s2 = mid[2] < 0 and lo[2] < mid[2]
            if s2:
                lo = mid
            s3 = hi[2] < 0 or hi[2] > hi_next[2]
            if s3:
                hi = hi_next
            s4 = mid[2] > 0 and hi[2] > mid[2]
            if s4:
                hi = mid
            swap = s1 or s2 or s3 or s4
            it += 1
        improved = lo[1] < p0[1] or hi[1] < p0[1]
        alpha = lo[0] if lo[1] < hi[1] else hi[0]
        if improved:
            c.qacc = c.qacc + c.search * alpha; c.Ma = c.Ma + mv * alpha; c.Jaref = c.Jaref + jv * alpha

    cs = create(d.qacc_smooth, grad=False)
    cw = create(d.qacc_warmstart, grad=False)
    c = create(d.qacc_warmstart if cw.cost < cs.cost else d.qacc_smooth)
    niter = 0
    while True:
        improvement = (c.prev_cost - c.cost) * scale
        gradient = np.linalg.norm(c.grad) * scale
        if niter >= m.iterations or improvement < m.tolerance or gradient < m.tolerance:
            break
        linesearch(c)
        pg, pM = c.grad.copy(), c.Mgrad.copy()
        update_constraint(c)
        update_gradient(c)
        beta = 0.0 if m.solver == "newton" else max(0.0, c.grad @ (c.Mgrad - pM) / max(MINVAL, pg @ pM))
        c.search = -c.Mgrad + beta * c.search
        niter += 1
    d.qacc, d.qfrc_constraint, d.efc_force, d.solver_niter = c.qacc, c.qfrc_constraint, c.force, niter
    d.efc_active = c.Jaref < 0
    d.qacc_warmstart = c.qacc.copy()


# ----------------------------------------------------------------------------------------- forward / step
def forward(m: Model, d: Data):
    kinematics(m, d)
    mass_matrix(m, d)
    com_terms(m, d)
    collision(m, d)
    make_constraint(m, d)
    smooth_forces(m, d)
    solve(m, d)


def euler(m: Model, d: Data):
    t, dt = m.t, m.dt
    qacc = np.linalg.solve(d.M + dt * np.diag(t["dof_damping"]), d.qfrc_smooth + d.qfrc_constraint)   # eulerdamp
    d.act = d.act + dt * d.act_dot
    d.qvel = d.qvel + dt * qacc
    for j in range(m.njnt):
        qa, da = t["jnt_qposadr"][j], t["jnt_dofadr"][j]
        if t["jnt_type"][j] == FREE:
            d.qpos[qa:qa + 3] += dt * d.qvel[da:da + 3]
            w = d.qvel[da + 3:da + 6]
            n = np.linalg.norm(w)
            ax = w / n if n > MINVAL else np.zeros(3)
            h = 0.5 * dt * n
            qr = np.concatenate([[np.cos(h)], ax * np.sin(h)])
            q = d.qpos[qa + 3:qa + 7]
            qn = np.array([q[0] * qr[0] - q[1:] @ qr[1:], *(q[0] * qr[1:] + qr[0] * q[1:] + np.cross(q[1:], qr[1:]))])
            d.qpos[qa + 3:qa + 7] = qn / np.linalg.norm(qn)
        else:
            d.qpos[qa] += dt * d.qvel[da]


def step(m: Model, d: Data, ctrl, n_frames=1):
    d.ctrl = np.asarray(ctrl, np.float64).copy()
    for _ in range(n_frames):
        forward(m, d)
        euler(m, d)


def init(m: Model, d: Data, qpos, qvel):
    d.qpos = np.asarray(qpos, np.float64).copy(); d.qvel = np.asarray(qvel, np.float64).copy()
    d.act = np.zeros(m.na); d.ctrl = np.zeros(m.nu); d.qacc_warmstart = np.zeros(m.nv)
    forward(m, d)


# ----------------------------------------------------------------------------------------- env layer [REF Rodent_Env_Brax.py:98-162]
def get_obs(m: Model, d: Data, track_pos, cur_frame):
    fi = int(np.clip(cur_frame + 1, 0, len(track_pos) - 1))
    local = d.xmat[1] @ (np.asarray(track_pos[fi], np.float64) - d.qpos[:3])
    return np.concatenate([d.qpos, d.qvel, d.cinert[1:].ravel(), d.cvel[1:].ravel(), d.qfrc_actuator, local])


def env_step(m: Model, d: Data, action, track_pos, cur_frame, n_frames=10, healthy_reward=1.0, ctrl_cost_weight=0.1,
             healthy_z_range=(0.03, 0.5), terminate_when_unhealthy=True):
    step(m, d, action, n_frames)
    fi = int(np.clip(cur_frame, 0, len(track_pos) - 1))
    pos_reward = np.exp(-100 * np.linalg.norm(d.qpos[:3] - np.asarray(track_pos[fi], np.float64)))
    z = d.qpos[2]
    healthy = 0.0 if (z < healthy_z_range[0] or z > healthy_z_range[1]) else 1.0
    hr = healthy_reward if terminate_when_unhealthy else healthy_reward * healthy
    cc = ctrl_cost_weight * float(np.sum(np.square(action)))
    obs = get_obs(m, d, track_pos, cur_frame + 1)
    return obs, pos_reward + hr - cc, (1 - healthy) if terminate_when_unhealthy else 0.0, cur_frame + 1, np.array([pos_reward, -cc, hr])


# ----------------------------------------------------------------------------------------- mj_setConst
def set_const(m: Model):
    """dof_invweight0, body_invweight0 [nb, 2], stat.meaninertia recomputed at qpos0 from the Jacobian formulation."""
    d = Data(m)
    kinematics(m, d)
    mass_matrix(m, d)
    Minv = np.linalg.inv(d.M)
    t = m.t
    dw = np.diag(Minv).copy()
    for j in range(m.njnt):
        if t["jnt_type"][j] == FREE:
            a = t["jnt_dofadr"][j]
            dw[a:a + 3] = dw[a:a + 3].mean(); dw[a + 3:a + 6] = dw[a + 3:a + 6].mean()
    bw = np.zeros((m.nbody, 2))
    for b in range(1, m.nbody):
        if not m.anc[b].any():
            continue
        bw[b, 0] = max(np.trace(d.Jp[b] @ Minv @ d.Jp[b].T) / 3, MINVAL)
        bw[b, 1] = max(np.trace(d.Jr[b] @ Minv @ d.Jr[b].T) / 3, MINVAL)
    return dw, bw, np.trace(d.M) / max(m.nv, 1)


def dense_from_sparse(tables, qM):
    """MuJoCo's sparse row layout (dof_Madr, ancestors along dof_parentid) -> dense symmetric matrix (for comparisons)."""
    nv = int(tables["nv"])
    M = np.zeros((nv, nv))
    for i in range(nv):
        a, j = int(tables["dof_Madr"][i]), i
        while j >= 0:
            M[i, j] = M[j, i] = qM[a]
            a += 1; j = int(tables["dof_parentid"][j])
    return M
