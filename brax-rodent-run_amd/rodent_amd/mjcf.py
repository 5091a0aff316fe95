"""MJCF-subset model compiler: XML -> flat model tables (host side, one-time).

Replaces the reference's `mujoco.MjModel.from_xml_path` + `brax.io.mjcf.load_model`
call pair [REF Rodent_Env_Brax.py:41-51] for exactly the MJCF subset the rodent models
use.  Everything is computed in float64 with numpy and cast to
float32 when the model blob is written, like MuJoCo (double) -> MJX (float32).

What is covered: `compiler angle=radian`; nested `<default class>` for joint / geom /
general; `freejoint` + hinge joints; plane / sphere / capsule / ellipsoid / box /
cylinder geoms with `pos`, `quat`, `euler`; inertia inferred from geoms; `<general>`
actuators on joints (filter dynamics, affine bias); `<replicate>`; `<exclude>`;
and the `mj_setConst` constants (`dof_invweight0`, `body_invweight0`, `stat.meaninertia`).

The MuJoCo behaviour restated here is from the public MuJoCo documentation
(XML reference, "Computation" chapter); the MuJoCo sources are not in the
reference tree, so this is pinned only by the known-answer data in the reference's
notebooks (tests/test_known_answers.py, tests/test_ktables.py).
"""
from __future__ import annotations

import copy
import math
import os
import struct
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

# geom type enum (MuJoCo mjtGeom)
PLANE, HFIELD, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX, MESH = range(8)
_GEOM_TYPES = {"plane": PLANE, "hfield": HFIELD, "sphere": SPHERE, "capsule": CAPSULE,
               "ellipsoid": ELLIPSOID, "cylinder": CYLINDER, "box": BOX, "mesh": MESH}
# joint type enum (MuJoCo mjtJoint)
FREE, BALL, SLIDE, HINGE = range(4)
_JNT_TYPES = {"free": FREE, "ball": BALL, "slide": SLIDE, "hinge": HINGE}

MJ_MINVAL = 1e-15


# ----------------------------------------------------------------------------- quaternions
def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat_to_quat(m):
    """Rotation matrix -> unit quaternion (w >= 0)."""
    t = np.trace(m)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    q = q / np.linalg.norm(q)
    return -q if q[0] < 0 else q


def axis_angle_quat(axis, angle):
    s = math.sin(angle * 0.5)
    return np.array([math.cos(angle * 0.5), axis[0] * s, axis[1] * s, axis[2] * s])


def euler_to_quat(e, seq="xyz"):
    """MuJoCo `euler` attribute: lower-case = intrinsic rotations, applied left to right."""
    q = np.array([1.0, 0, 0, 0])
    for ang, ax in zip(e, seq):
        a = {"x": (1, 0, 0), "y": (0, 1, 0), "z": (0, 0, 1)}[ax.lower()]
        qi = axis_angle_quat(a, ang)
        q = quat_mul(q, qi) if ax.islower() else quat_mul(qi, q)
    return q / np.linalg.norm(q)


def rot(q, v):
    return quat_to_mat(q) @ np.asarray(v, dtype=np.float64)


# ----------------------------------------------------------------------------- defaults
_BUILTIN = {
    "joint": dict(type="hinge", pos=[0, 0, 0], axis=[0, 0, 1], limited=None, stiffness=0.0,
                  range=[0, 0], margin=0.0, ref=0.0, springref=0.0, armature=0.0, damping=0.0,
                  frictionloss=0.0, solreflimit=[0.02, 1], solimplimit=[0.9, 0.95, 0.001, 0.5, 2]),
    "geom": dict(type="sphere", contype=1, conaffinity=1, condim=3, group=0, priority=0,
                 size=[0, 0, 0], friction=[1, 0.005, 0.0001], solmix=1.0, solref=[0.02, 1],
                 solimp=[0.9, 0.95, 0.001, 0.5, 2], margin=0.0, gap=0.0, density=1000.0, mass=None),
    "general": dict(ctrllimited=None, forcelimited=None, actlimited=None, ctrlrange=[0, 0],
                    forcerange=[0, 0], gear=[1, 0, 0, 0, 0, 0], dyntype="none", gaintype="fixed",
                    biastype="none", dynprm=[1, 0, 0], gainprm=[1, 0, 0], biasprm=[0, 0, 0]),
}
_VEC_ATTRS = {"pos", "axis", "range", "solreflimit", "solimplimit", "size", "friction", "solref",
              "solimp", "ctrlrange", "forcerange", "gear", "dynprm", "gainprm", "biasprm",
              "quat", "euler"}
_INT_ATTRS = {"contype", "conaffinity", "condim", "group", "priority"}
_STR_ATTRS = {"type", "dyntype", "gaintype", "biastype", "name", "class", "joint", "material",
              "tendon", "site", "rgba", "mesh"}
_BOOL_ATTRS = {"limited", "ctrllimited", "forcelimited", "actlimited"}


def _apply_attrs(dst: dict, elem: ET.Element):
    """Overlay XML attributes on a defaults dict; short vectors keep the trailing defaults."""
    for k, v in elem.attrib.items():
        if k in ("name", "class", "rgba", "material"):
            continue
        if k in _VEC_ATTRS:
            vals = [float(x) for x in v.split()]
            if k in dst and isinstance(dst[k], list) and len(vals) < len(dst[k]) and k not in ("quat", "euler"):
                vals = vals + list(dst[k][len(vals):])
            dst[k] = vals
        elif k in _INT_ATTRS:
            dst[k] = int(v)
        elif k in _BOOL_ATTRS:
            dst[k] = {"true": True, "false": False, "auto": None}[v]
        elif k in _STR_ATTRS:
            dst[k] = v
        else:
            try:
                dst[k] = float(v)
            except ValueError:
                dst[k] = v


class _Defaults:
    def __init__(self):
        self.classes: Dict[str, Dict[str, dict]] = {"main": copy.deepcopy(_BUILTIN)}
        self.replica_suffixes: List[str] = []

    def load(self, elem: ET.Element, parent: str = "main", top: bool = True):
        name = "main" if top else elem.attrib["class"]
        if not top:
            self.classes[name] = copy.deepcopy(self.classes[parent])
        cur = self.classes[name]
        for child in elem:
            if child.tag == "default":
                continue
            if child.tag in cur:
                _apply_attrs(cur[child.tag], child)
        for child in elem:
            if child.tag == "default":
                self.load(child, name, top=False)

    def get(self, tag: str, cls: Optional[str]) -> dict:
        return copy.deepcopy(self.classes[cls or "main"][tag])


# ----------------------------------------------------------------------------- geom inertia
def _geom_volume_inertia(gtype, size):
    """Volume and unit-density principal inertia of a primitive (MuJoCo geom frame)."""
    if gtype == SPHERE:
        r = size[0]
        vol = 4.0 / 3.0 * math.pi * r ** 3
        i = 0.4 * vol * r * r
        return vol, np.array([i, i, i])
    if gtype == CAPSULE:
        r, height = size[0], 2 * size[1]
        vol = math.pi * (r * r * height + 4.0 / 3.0 * r ** 3)
        ms = 4.0 / 3.0 * math.pi * r ** 3           # two hemispheres = one sphere
        mc = math.pi * r * r * height
        ixy = mc * (3 * r * r + height * height) / 12.0
        iz = mc * r * r / 2.0
        si = 0.4 * ms * r * r
        ixy += si + ms * height * (3 * r + 2 * height) / 8.0
        iz += si
        return vol, np.array([ixy, ixy, iz])
    if gtype == ELLIPSOID:
        a, b, c = size
        vol = 4.0 / 3.0 * math.pi * a * b * c
        return vol, vol / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == CYLINDER:
        r, height = size[0], 2 * size[1]
        vol = math.pi * r * r * height
        ixy = vol * (3 * r * r + height * height) / 12.0
        return vol, np.array([ixy, ixy, vol * r * r / 2.0])
    if gtype == BOX:
        a, b, c = size
        vol = 8 * a * b * c
        return vol, vol / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    return 0.0, np.zeros(3)          # plane / hfield: massless


# ----------------------------------------------------------------------------- compiler
class _Body:
    def __init__(self, name, parent, pos, quat):
        self.name, self.parent, self.pos, self.quat = name, parent, pos, quat
        self.joints: List[dict] = []
        self.geoms: List[dict] = []
        self.children: List["_Body"] = []
        self.id = -1


def _orientation(attrs: dict):
    if "quat" in attrs:
        q = np.array(attrs["quat"], dtype=np.float64)
        return q / np.linalg.norm(q)
    if "euler" in attrs:
        return euler_to_quat(attrs["euler"])
    return np.array([1.0, 0, 0, 0])


def _parse_body(elem, parent: Optional[_Body], defaults: _Defaults, suffix: str,
                frame_pos=None, frame_quat=None) -> _Body:
    a = {}
    _apply_attrs(a, elem)
    pos = np.array(a.get("pos", [0, 0, 0]), dtype=np.float64)
    quat = _orientation(a)
    if frame_quat is not None:                  # <replicate> frame applied to its direct children
        pos = frame_pos + rot(frame_quat, pos)
        quat = quat_mul(frame_quat, quat)
    b = _Body(elem.attrib.get("name", "") + suffix, parent, pos, quat)
    for child in elem:
        tag = child.tag
        if tag == "freejoint":
            b.joints.append(dict(type="free", name=child.attrib.get("name", "") + suffix))
        elif tag == "joint":
            j = defaults.get("joint", child.attrib.get("class"))
            _apply_attrs(j, child)
            j["name"] = child.attrib.get("name", "") + suffix
            b.joints.append(j)
        elif tag == "geom":
            g = defaults.get("geom", child.attrib.get("class"))
            _apply_attrs(g, child)
            g["name"] = child.attrib.get("name", "") + suffix
            b.geoms.append(g)
        elif tag == "body":
            b.children.append(_parse_body(child, b, defaults, suffix))
        elif tag == "replicate":
            b.children.extend(_parse_replicate(child, b, defaults, suffix))
        # site / camera / light / inertial(not used by the rodent models): ignored
    return b


def _parse_replicate(elem, parent, defaults, suffix):
    """`<replicate count euler sep>`: child i is rotated by i*euler (cumulative), named `name<sep>i`."""
    count = int(elem.attrib["count"])
    sep = elem.attrib.get("sep", "")
    e = [float(x) for x in elem.attrib.get("euler", "0 0 0").split()]
    off = np.array([float(x) for x in elem.attrib.get("offset", "0 0 0").split()])
    dq = euler_to_quat(e)
    out = []
    fq = np.array([1.0, 0, 0, 0])
    fp = np.zeros(3)
    for i in range(count):
        defaults.replica_suffixes.append(f"{suffix}{sep}{i}")
        for child in elem:
            if child.tag == "body":
                out.append(_parse_body(child, parent, defaults, f"{suffix}{sep}{i}", fp.copy(), fq.copy()))
        fp = fp + rot(fq, off)
        fq = quat_mul(fq, dq)
    return out


def compile_mjcf(xml_path: str, *, iterations: int = 6, ls_iterations: int = 6,
                 solver: str = "cg", contacts: str = "strict") -> Dict[str, np.ndarray]:
    """Compile an MJCF file into a dict of numpy tables (MuJoCo field names + engine tables).

    `iterations` / `ls_iterations` / `solver` mirror the `opt` overrides the reference applies
    after loading [REF Rodent_Env_Brax.py:42-49].  `contacts`: "strict" raises on a geom pair whose collision primitive is
    not implemented (everything but plane - sphere / capsule / ellipsoid); "supported_only" drops such pairs and records
    their number in `ndropped_pairs` (rodent_cpu.xml: ~4.3 k capsule-capsule / ellipsoid self-collision pairs -- the
    config-1 plumbing case runs with contacts off).
    """
    root = ET.parse(xml_path).getroot()
    comp = {}
    for c in root.findall("compiler"):
        comp.update(c.attrib)
    if comp.get("angle", "degree") != "radian":
        raise ValueError("only <compiler angle='radian'> is supported")
    defaults = _Defaults()
    for d in root.findall("default"):
        defaults.load(d)

    wb = root.find("worldbody")
    world = _parse_body(wb, None, defaults, "")
    world.name = "world"

    # depth-first body order (MuJoCo ids)
    bodies: List[_Body] = []

    def visit(b):
        b.id = len(bodies)
        bodies.append(b)
        for c in b.children:
            visit(c)

    visit(world)
    nbody = len(bodies)

    m: Dict[str, np.ndarray] = {}
    body_parentid = np.array([0 if b.parent is None else b.parent.id for b in bodies], dtype=np.int32)
    body_pos = np.array([b.pos for b in bodies])
    body_quat = np.array([b.quat for b in bodies])

    # ---- joints / dofs
    jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid = [], [], [], []
    jnt_pos, jnt_axis, jnt_stiffness, jnt_range, jnt_limited = [], [], [], [], []
    jnt_solref, jnt_solimp, jnt_margin, jnt_names = [], [], [], []
    dof_bodyid, dof_jntid, dof_parentid, dof_armature, dof_damping = [], [], [], [], []
    qpos0, qpos_spring = [], []
    body_jntadr = np.full(nbody, -1, np.int32)
    body_jntnum = np.zeros(nbody, np.int32)
    body_dofadr = np.full(nbody, -1, np.int32)
    body_dofnum = np.zeros(nbody, np.int32)
    body_lastdof = np.full(nbody, -1, np.int32)   # last dof of the nearest ancestor-or-self with dofs
    for b in bodies:
        pid = body_parentid[b.id]
        last = body_lastdof[pid] if b.id > 0 else -1
        if b.joints:
            body_jntadr[b.id] = len(jnt_type)
            body_jntnum[b.id] = len(b.joints)
            body_dofadr[b.id] = len(dof_bodyid)
        for j in b.joints:
            jt = _JNT_TYPES[j["type"]]
            jid = len(jnt_type)
            jnt_type.append(jt)
            jnt_qposadr.append(len(qpos0))
            jnt_dofadr.append(len(dof_bodyid))
            jnt_bodyid.append(b.id)
            jnt_names.append(j["name"])
            if jt == FREE:
                if b.parent is None or b.parent.id != 0:
                    raise ValueError("free joint must be on a child of the world")
                jnt_pos.append([0, 0, 0]); jnt_axis.append([0, 0, 1]); jnt_stiffness.append(0.0)
                jnt_range.append([0, 0]); jnt_limited.append(0)
                jnt_solref.append(_BUILTIN["joint"]["solreflimit"]); jnt_solimp.append(_BUILTIN["joint"]["solimplimit"])
                jnt_margin.append(0.0)
                qpos0 += list(b.pos) + list(b.quat)
                qpos_spring += list(b.pos) + list(b.quat)
                nd = 6
                arm, damp = 0.0, 0.0
            elif jt == HINGE:
                ax = np.array(j["axis"], dtype=np.float64)
                ax = ax / np.linalg.norm(ax)
                jnt_pos.append(j["pos"]); jnt_axis.append(list(ax)); jnt_stiffness.append(j["stiffness"])
                rng = j["range"]
                lim = j["limited"]
                if lim is None:                          # autolimits
                    lim = not (rng[0] == 0 and rng[1] == 0)
                jnt_range.append(rng); jnt_limited.append(int(lim))
                jnt_solref.append(j["solreflimit"]); jnt_solimp.append(j["solimplimit"])
                jnt_margin.append(j["margin"])
                qpos0.append(j["ref"]); qpos_spring.append(j["springref"])
                nd = 1
                arm, damp = j["armature"], j["damping"]
            else:
                raise ValueError(f"joint type {j['type']} not supported")
            for _ in range(nd):
                dof_bodyid.append(b.id); dof_jntid.append(jid)
                dof_parentid.append(last)
                last = len(dof_bodyid) - 1
                dof_armature.append(arm); dof_damping.append(damp)
        body_dofnum[b.id] = (len(dof_bodyid) - body_dofadr[b.id]) if b.joints else 0
        body_lastdof[b.id] = last
    nq, nv, njnt = len(qpos0), len(dof_bodyid), len(jnt_type)
    dof_parentid = np.array(dof_parentid, np.int32)

    # body_weldid / rootid
    body_weldid = np.zeros(nbody, np.int32)
    body_rootid = np.zeros(nbody, np.int32)
    for b in bodies[1:]:
        pid = body_parentid[b.id]
        body_weldid[b.id] = b.id if b.joints else body_weldid[pid]
        body_rootid[b.id] = b.id if pid == 0 else body_rootid[pid]

    # sparse M addressing: row k = [diag, parent, grandparent, ...]
    dof_Madr = np.zeros(nv, np.int32)
    dof_depth = np.zeros(nv, np.int32)
    adr = 0
    for k in range(nv):
        dof_Madr[k] = adr
        p, d = dof_parentid[k], 0
        while p >= 0:
            d += 1
            p = dof_parentid[p]
        dof_depth[k] = d
        adr += d + 1
    nM = adr

    # ---- geoms
    geom_type, geom_bodyid, geom_pos, geom_quat, geom_size = [], [], [], [], []
    geom_contype, geom_conaffinity, geom_condim, geom_priority = [], [], [], []
    geom_friction, geom_solref, geom_solimp, geom_solmix, geom_margin, geom_gap = [], [], [], [], [], []
    geom_names = []
    body_geomadr = np.full(nbody, -1, np.int32)
    body_geomnum = np.zeros(nbody, np.int32)
    body_mass = np.zeros(nbody)
    body_ipos = np.zeros((nbody, 3))
    body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    body_inertia = np.zeros((nbody, 3))
    for b in bodies:
        if b.geoms:
            body_geomadr[b.id] = len(geom_type)
            body_geomnum[b.id] = len(b.geoms)
        masses, coms, inerts = [], [], []
        for g in b.geoms:
            gt = _GEOM_TYPES[g["type"]]
            if gt in (MESH, HFIELD):
                raise ValueError("mesh / hfield geoms are not supported")
            if "fromto" in g:
                raise ValueError("geom fromto is not supported")
            size = list(g["size"]) + [0, 0, 0]
            size = size[:3]
            if gt == SPHERE:
                size = [size[0], 0, 0]
            elif gt in (CAPSULE, CYLINDER):
                size = [size[0], size[1], 0]
            gq = _orientation(g)
            gp = np.array(g.get("pos", [0, 0, 0]), dtype=np.float64)
            geom_type.append(gt); geom_bodyid.append(b.id); geom_pos.append(gp); geom_quat.append(gq)
            geom_size.append(size)
            geom_contype.append(g["contype"]); geom_conaffinity.append(g["conaffinity"])
            geom_condim.append(g["condim"]); geom_priority.append(g["priority"])
            geom_friction.append(g["friction"]); geom_solref.append(g["solref"]); geom_solimp.append(g["solimp"])
            geom_solmix.append(g["solmix"]); geom_margin.append(g["margin"]); geom_gap.append(g["gap"])
            geom_names.append(g["name"])
            vol, i_unit = _geom_volume_inertia(gt, size)
            if g.get("mass") is not None:
                mass = float(g["mass"])
                dens = mass / vol if vol > 0 else 0.0
            else:
                dens = g["density"]
                mass = dens * vol
            if mass > 0:
                masses.append(mass); coms.append(gp)
                R = quat_to_mat(gq)
                inerts.append(R @ np.diag(dens * i_unit) @ R.T)
        if b.id > 0 and masses:
            M = sum(masses)
            com = sum(mm * c for mm, c in zip(masses, coms)) / M
            I = np.zeros((3, 3))
            for mm, c, Ig in zip(masses, coms, inerts):
                d = c - com
                I += Ig + mm * (d @ d * np.eye(3) - np.outer(d, d))
            w, V = np.linalg.eigh(I)
            order = np.argsort(-w)                    # MuJoCo: eigenvalues in decreasing order
            w, V = w[order], V[:, order]
            if np.linalg.det(V) < 0:
                V[:, 2] = -V[:, 2]
            body_mass[b.id] = M
            body_ipos[b.id] = com
            body_iquat[b.id] = mat_to_quat(V)
            body_inertia[b.id] = w
    ngeom = len(geom_type)
    for b in bodies[1:]:
        if b.joints and body_mass[b.id] < MJ_MINVAL:
            # MuJoCo accepts a massless moving body when a welded descendant carries mass
            if not any(body_weldid[c.id] == b.id and body_mass[c.id] > MJ_MINVAL for c in bodies[1:]):
                raise ValueError(f"moving body {b.name} has no mass")

    # ---- fixed tendons [REF models/rodent_cpu.xml:505-560]: length = sum coef_i * qpos[joint_i] (MuJoCo <tendon><fixed>)
    name2jnt = {n: i for i, n in enumerate(jnt_names)}
    ten_names, ten_adr, ten_num, wrap_jnt, wrap_coef = [], [], [], [], []
    ten = root.find("tendon")
    if ten is not None:
        for t in ten:
            if t.tag != "fixed":
                raise ValueError(f"tendon <{t.tag}> not supported (fixed tendons only)")
            ten_names.append(t.attrib["name"]); ten_adr.append(len(wrap_jnt)); ten_num.append(0)
            for w in t:
                if w.tag != "joint":
                    raise ValueError("fixed tendons wrap joints only")
                if jnt_type[name2jnt[w.attrib["joint"]]] != HINGE:
                    raise ValueError("fixed tendon over a non-hinge joint not supported")
                wrap_jnt.append(name2jnt[w.attrib["joint"]]); wrap_coef.append(float(w.attrib["coef"])); ten_num[-1] += 1
    name2ten = {n: i for i, n in enumerate(ten_names)}

    # ---- actuators
    act = root.find("actuator")
    trn_type, trn_id, gain0, bias, tau, ctrlrange, ctrllimited, act_names = [], [], [], [], [], [], [], []
    specs = []
    if act is not None:
        for a in act:
            if a.tag != "general":
                raise ValueError(f"actuator <{a.tag}> not supported")
            g = defaults.get("general", a.attrib.get("class"))
            _apply_attrs(g, a)
            if g["dyntype"] != "filter" or g["gaintype"] != "fixed" or g["biastype"] != "affine":
                raise ValueError("only filter/fixed/affine general actuators are supported")
            if g["forcelimited"]:
                raise ValueError("forcelimited actuators not supported")
            cl = g["ctrllimited"]
            if cl is None:
                cl = not (g["ctrlrange"][0] == 0 and g["ctrlrange"][1] == 0)
            if "tendon" in a.attrib:
                specs.append(("tendon", a.attrib["tendon"], a.attrib.get("name", a.attrib["tendon"]), g, int(cl)))
            else:
                specs.append(("joint", a.attrib["joint"], a.attrib.get("name", a.attrib["joint"]), g, int(cl)))
        # <replicate> suffixes joint names but the rodent_pair actuator block sits outside it and
        # names the un-suffixed joints [REF models/rodent_pair.xml:545-576]: drive every replica,
        # ordered [all actuators of replica 0][all of replica 1]....
        sfxs = [""] if all((n in name2jnt) if k == "joint" else (n in name2ten) for k, n, _, _, _ in specs) else list(defaults.replica_suffixes)
        for sfx in sfxs:
            for kind, tn, nm, g, cl in specs:
                table = name2jnt if kind == "joint" else name2ten
                if tn + sfx not in table:
                    raise ValueError(f"actuator {kind} {tn + sfx} not found")
                trn_type.append(0 if kind == "joint" else 3); trn_id.append(table[tn + sfx])      # mjtTrn: 0 joint, 3 tendon
                gain0.append(g["gainprm"][0]); bias.append(g["biasprm"][:3])
                tau.append(g["dynprm"][0]); ctrlrange.append(g["ctrlrange"]); ctrllimited.append(cl)
                act_names.append(nm + sfx)
    nu = len(trn_id)
    # sparse transmission: actuator u acts on the dofs mom_dof[madr[u]:madr[u+1]] with the coefficients mom_coef
    # (`actuator_moment` rows; a joint transmission is one entry of gear 1)
    madr, mom_jnt, mom_coef = [0], [], []
    for tt, ti in zip(trn_type, trn_id):
        if tt == 0:
            mom_jnt.append(ti); mom_coef.append(1.0)
        else:
            mom_jnt += wrap_jnt[ten_adr[ti]:ten_adr[ti] + ten_num[ti]]; mom_coef += wrap_coef[ten_adr[ti]:ten_adr[ti] + ten_num[ti]]
        madr.append(len(mom_jnt))

    # ---- excludes
    excl = set()
    con = root.find("contact")
    if con is not None:
        name2body = {b.name: b.id for b in bodies}
        for e in con.findall("exclude"):
            b1, b2 = name2body[e.attrib["body1"]], name2body[e.attrib["body2"]]
            excl.add((min(b1, b2), max(b1, b2)))

    f64 = lambda x, shape=None: np.asarray(x, dtype=np.float64).reshape(shape) if shape else np.asarray(x, dtype=np.float64)
    i32 = lambda x: np.asarray(x, dtype=np.int32)
    m.update(
        nq=i32(nq), nv=i32(nv), nu=i32(nu), na=i32(nu), nbody=i32(nbody), njnt=i32(njnt), ngeom=i32(ngeom), nM=i32(nM),
        body_parentid=body_parentid, body_rootid=body_rootid, body_weldid=body_weldid,
        body_jntadr=body_jntadr, body_jntnum=body_jntnum, body_dofadr=body_dofadr, body_dofnum=body_dofnum,
        body_geomadr=body_geomadr, body_geomnum=body_geomnum, body_lastdof=body_lastdof,
        body_pos=body_pos, body_quat=body_quat, body_ipos=body_ipos, body_iquat=body_iquat,
        body_mass=body_mass, body_inertia=body_inertia,
        jnt_type=i32(jnt_type), jnt_qposadr=i32(jnt_qposadr), jnt_dofadr=i32(jnt_dofadr), jnt_bodyid=i32(jnt_bodyid),
        jnt_pos=f64(jnt_pos, (njnt, 3)), jnt_axis=f64(jnt_axis, (njnt, 3)), jnt_stiffness=f64(jnt_stiffness),
        jnt_range=f64(jnt_range, (njnt, 2)), jnt_limited=i32(jnt_limited), jnt_solref=f64(jnt_solref, (njnt, 2)),
        jnt_solimp=f64(jnt_solimp, (njnt, 5)), jnt_margin=f64(jnt_margin),
        dof_bodyid=i32(dof_bodyid), dof_jntid=i32(dof_jntid), dof_parentid=dof_parentid, dof_Madr=dof_Madr,
        dof_depth=dof_depth, dof_armature=f64(dof_armature), dof_damping=f64(dof_damping),
        qpos0=f64(qpos0), qpos_spring=f64(qpos_spring),
        geom_type=i32(geom_type), geom_bodyid=i32(geom_bodyid), geom_pos=f64(geom_pos, (ngeom, 3)),
        geom_quat=f64(geom_quat, (ngeom, 4)), geom_size=f64(geom_size, (ngeom, 3)),
        geom_contype=i32(geom_contype), geom_conaffinity=i32(geom_conaffinity), geom_condim=i32(geom_condim),
        geom_priority=i32(geom_priority), geom_friction=f64(geom_friction, (ngeom, 3)),
        geom_solref=f64(geom_solref, (ngeom, 2)), geom_solimp=f64(geom_solimp, (ngeom, 5)),
        geom_solmix=f64(geom_solmix), geom_margin=f64(geom_margin), geom_gap=f64(geom_gap),
        actuator_trnid=i32(trn_id), actuator_trntype=i32(trn_type), actuator_momentadr=i32(madr),
        actuator_moment_jnt=i32(mom_jnt), actuator_moment_coef=f64(mom_coef), ntendon=i32(len(ten_names)),
        actuator_gainprm0=f64(gain0), actuator_biasprm=f64(bias, (nu, 3)),
        actuator_dynprm0=f64(tau), actuator_ctrlrange=f64(ctrlrange, (nu, 2)), actuator_ctrllimited=i32(ctrllimited),
        # options: MuJoCo defaults (the rodent XMLs have no <option>) + the reference's overrides
        opt_timestep=f64(0.002), opt_gravity=f64([0, 0, -9.81]), opt_tolerance=f64(1e-8),
        opt_ls_tolerance=f64(0.01), opt_impratio=f64(1.0), opt_iterations=i32(iterations),
        opt_ls_iterations=i32(ls_iterations), opt_solver=i32({"cg": 1, "newton": 2}[solver.lower()]),
    )
    m["_names"] = dict(body=[b.name for b in bodies], joint=jnt_names, geom=geom_names, actuator=act_names, tendon=ten_names)
    m["_exclude"] = excl
    _set_const(m)
    _collision_tables(m, contacts)
    _engine_tables(m)
    # the fused HIP kernel serves the floor-contact, joint-actuated, free-floating rodents; other models (rodent_cpu.xml:
    # tendon transmissions, welded root) compile for the CPU path only and their blob carries no kernel tables, so
    # rr_model_load refuses them
    # floor-contact rodents: the fused static-slot instances; models with contacts between moving bodies / condim 1 / tendon transmissions
    # (rodent_cpu.xml): the DYN instance, sphere / capsule pairs only (ktables.build_kernel_tables decides and checks)
    floor_model = (not any(trn_type)) and any(t == FREE for t in jnt_type) and not np.any(m["con_body1"] != 0) and not np.any(m["con_dim"] != 3)
    dyn_model = int(m["ncon"]) > 0 and bool(np.all(np.isin(m["con_kind"], (4, 5, 6))))
    hip_ok = int(m["ncon"]) > 0 and (floor_model or dyn_model)
    m["hip_supported"] = np.int32(hip_ok)
    if hip_ok:
        from .ktables import build_kernel_tables, replica_model
        m.update(build_kernel_tables(m))
        # two identical trees (rodent_pair.xml): the kernel tables of ONE replica as `h_*` (h_k_dof_i, ...) + its dims `h_dims`; the
        # library then steps such a model with one wavefront per replica (csrc/rr_kernel.h, PAIR instances)
        try:
            half = replica_model(m)
        except ValueError:                 # two trees that are not identical replicas: the generic one-wave instance steps the model
            half = None
        if half is not None:
            hk = build_kernel_tables(half)
            m.update({"h_" + k: v for k, v in hk.items()})
            m["h_dims"] = np.asarray([half[k] for k in ("nq", "nv", "nu", "nbody", "njnt", "nM", "ncon")] + [int(half["dof_depth"].max())], np.int32)
            m["h_con_of_replica"] = half["_contacts_of_replica"]
    return m


# ----------------------------------------------------------------------------- mj_setConst
def _kinematics0(m):
    """FK / COM / cdof / composite inertia at qpos0 (float64), for the setConst constants."""
    nb, nv = int(m["nbody"]), int(m["nv"])
    xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4)); xquat[0, 0] = 1
    xmat = np.zeros((nb, 3, 3)); xmat[0] = np.eye(3)
    xanchor = np.zeros((int(m["njnt"]), 3)); xaxis = np.zeros((int(m["njnt"]), 3))
    q0 = m["qpos0"]
    for b in range(1, nb):
        p = m["body_parentid"][b]
        pos = xpos[p] + xmat[p] @ m["body_pos"][b]
        quat = quat_mul(xquat[p], m["body_quat"][b])
        for k in range(m["body_jntnum"][b]):
            j = m["body_jntadr"][b] + k
            if m["jnt_type"][j] == FREE:
                a = m["jnt_qposadr"][j]
                pos = q0[a:a + 3].copy(); quat = q0[a + 3:a + 7] / np.linalg.norm(q0[a + 3:a + 7])
                xanchor[j] = pos; xaxis[j] = [0, 0, 1]
            else:
                xanchor[j] = pos + rot(quat, m["jnt_pos"][j])
                xaxis[j] = rot(quat, m["jnt_axis"][j])
                # qpos0 - qpos0 = 0 -> no rotation
        xpos[b], xquat[b], xmat[b] = pos, quat / np.linalg.norm(quat), quat_to_mat(quat)
    xipos = np.array([xpos[b] + xmat[b] @ m["body_ipos"][b] for b in range(nb)])
    ximat = np.array([quat_to_mat(quat_mul(xquat[b], m["body_iquat"][b])) for b in range(nb)])
    return xpos, xquat, xmat, xipos, ximat, xanchor, xaxis


def _dense_mass_matrix(m, xmat, xipos, ximat, xanchor, xaxis):
    nb, nv = int(m["nbody"]), int(m["nv"])
    mass = m["body_mass"]
    root = m["body_rootid"]
    # subtree com per root
    com = np.zeros((nb, 3))
    for r in set(root[1:]):
        sel = [b for b in range(1, nb) if root[b] == r]
        com[r] = sum(mass[b] * xipos[b] for b in sel) / sum(mass[b] for b in sel)
    # 6D jacobian columns (angular; linear) of each dof about the root COM
    cdof = np.zeros((nv, 6))
    for d in range(nv):
        j = m["dof_jntid"][d]; b = m["dof_bodyid"][d]
        off = com[root[b]] - xanchor[j]
        if m["jnt_type"][j] == FREE:
            k = d - m["jnt_dofadr"][j]
            if k < 3:
                cdof[d, 3 + k] = 1
            else:
                ax = xmat[b][:, k - 3]
                cdof[d, :3] = ax; cdof[d, 3:] = np.cross(ax, off)
        else:
            cdof[d, :3] = xaxis[j]; cdof[d, 3:] = np.cross(xaxis[j], off)
    # spatial inertia of each body about its root COM (6x6), accumulated to composites
    I6 = np.zeros((nb, 6, 6))
    for b in range(1, nb):
        R = ximat[b]
        Ic = R @ np.diag(m["body_inertia"][b]) @ R.T
        d = xipos[b] - com[root[b]]
        dx = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
        I6[b, :3, :3] = Ic - mass[b] * dx @ dx
        I6[b, :3, 3:] = mass[b] * dx
        I6[b, 3:, :3] = -mass[b] * dx
        I6[b, 3:, 3:] = mass[b] * np.eye(3)
    crb = I6.copy()
    for b in range(nb - 1, 0, -1):
        p = m["body_parentid"][b]
        if p > 0:
            crb[p] += crb[b]
    M = np.zeros((nv, nv))
    for i in range(nv):
        f = crb[m["dof_bodyid"][i]] @ cdof[i]
        j = i
        while j >= 0:
            M[i, j] = M[j, i] = cdof[j] @ f
            j = m["dof_parentid"][j]
        M[i, i] += m["dof_armature"][i]
    return M, cdof, com


def _set_const(m):
    nb, nv = int(m["nbody"]), int(m["nv"])
    xpos, xquat, xmat, xipos, ximat, xanchor, xaxis = _kinematics0(m)
    M, cdof, com = _dense_mass_matrix(m, xmat, xipos, ximat, xanchor, xaxis)
    Minv = np.linalg.inv(M)
    inv = np.diag(Minv).copy()
    for j in range(int(m["njnt"])):
        if m["jnt_type"][j] == FREE:
            a = m["jnt_dofadr"][j]
            inv[a:a + 3] = inv[a:a + 3].mean()
            inv[a + 3:a + 6] = inv[a + 3:a + 6].mean()
    m["dof_invweight0"] = inv
    m["stat_meaninertia"] = np.float64(np.trace(M) / max(nv, 1))
    # body_invweight0: mean diagonal of J M^-1 J' for the 6D jacobian at the body COM
    biw = np.zeros((nb, 2))
    for b in range(1, nb):
        if m["body_weldid"][b] == 0:
            continue
        J = np.zeros((6, nv))
        d = m["body_lastdof"][b]
        off = xipos[b] - com[m["body_rootid"][b]]
        while d >= 0:
            J[:3, d] = cdof[d, 3:] + np.cross(cdof[d, :3], off)
            J[3:, d] = cdof[d, :3]
            d = m["dof_parentid"][d]
        A = J @ Minv @ J.T
        biw[b, 0] = max((A[0, 0] + A[1, 1] + A[2, 2]) / 3, MJ_MINVAL)
        biw[b, 1] = max((A[3, 3] + A[4, 4] + A[5, 5]) / 3, MJ_MINVAL)
    m["body_invweight0"] = biw
    # subtree mass
    st = m["body_mass"].copy()
    for b in range(nb - 1, 0, -1):
        st[m["body_parentid"][b]] += st[b]
    m["body_subtreemass"] = st


# ----------------------------------------------------------------------------- collision tables
def _collision_tables(m, contacts="strict"):
    """Static geom-pair list + per-contact mixed parameters, as MJX builds at trace time.

    Pair filter, type ordering and parameter mixing restate `mjx collision_driver` (SURVEY A-4):
    pairs ordered by (type1, type2) then geom ids; plane-capsule emits 2 contacts, others 1.
    """
    ng = int(m["ngeom"])
    gb = m["geom_bodyid"]
    weld, par = m["body_weldid"], m["body_parentid"]
    pairs = []
    for g1 in range(ng):
        for g2 in range(g1 + 1, ng):
            b1, b2 = gb[g1], gb[g2]
            if (min(b1, b2), max(b1, b2)) in m["_exclude"]:
                continue
            w1, w2 = weld[b1], weld[b2]
            if w1 == w2:
                continue
            w1p, w2p = weld[par[w1]], weld[par[w2]]
            if w1 != 0 and w2 != 0 and (w1 == w2p or w2 == w1p):
                continue
            a, b = g1, g2
            if m["geom_type"][a] > m["geom_type"][b]:
                a, b = b, a
            t1, t2 = m["geom_type"][a], m["geom_type"][b]
            if t1 == PLANE and t2 == PLANE:
                continue
            mask = (m["geom_contype"][a] & m["geom_conaffinity"][b]) | (m["geom_contype"][b] & m["geom_conaffinity"][a])
            if not mask:
                continue
            pairs.append((int(t1), int(t2), a, b))
    pairs.sort()
    # supported primitives [UP mjx collision_primitive]: plane - sphere / capsule / ellipsoid (the floor contacts of the rodent models) and,
    # SURVEY.md 8(f)-4, the sphere / capsule pairs of self-colliding models (rodent_cpu.xml): sphere-sphere, sphere-capsule, capsule-capsule
    ok = [p for p in pairs if (p[0] == PLANE and p[1] in (SPHERE, CAPSULE, ELLIPSOID)) or (p[0] in (SPHERE, CAPSULE) and p[1] in (SPHERE, CAPSULE))]
    m["ndropped_pairs"] = np.int32(len(pairs) - len(ok))
    if len(ok) != len(pairs):
        if contacts != "supported_only":
            t1, t2 = next((p[0], p[1]) for p in pairs if p not in ok)
            raise ValueError(f"collision type pair ({t1},{t2}) not supported (plane-sphere/capsule/ellipsoid only)")
        pairs = ok
    # kind: 0 plane-sphere, 1 / 2 plane-capsule +axis / -axis end, 3 plane-ellipsoid, 4 sphere-sphere, 5 sphere-capsule, 6 capsule-capsule
    con_geom1, con_geom2, con_kind, con_dim = [], [], [], []
    fr, sr, si = [], [], []
    for t1, t2, g1, g2 in pairs:
        p1, p2 = m["geom_priority"][g1], m["geom_priority"][g2]
        if p1 == p2:
            mix = m["geom_solmix"][g1] / (m["geom_solmix"][g1] + m["geom_solmix"][g2])
            f = np.maximum(m["geom_friction"][g1], m["geom_friction"][g2])
            s_ref = mix * m["geom_solref"][g1] + (1 - mix) * m["geom_solref"][g2]
            s_imp = mix * m["geom_solimp"][g1] + (1 - mix) * m["geom_solimp"][g2]
            cd = max(m["geom_condim"][g1], m["geom_condim"][g2])
        else:
            gp = g1 if p1 > p2 else g2
            f, s_ref, s_imp, cd = m["geom_friction"][gp], m["geom_solref"][gp], m["geom_solimp"][gp], m["geom_condim"][gp]
        if cd not in (1, 3):
            raise ValueError("only condim 1 (frictionless) and condim 3 contacts are supported")
        if max(m["geom_margin"][g1], m["geom_margin"][g2]) != 0 or max(m["geom_gap"][g1], m["geom_gap"][g2]) != 0:
            raise ValueError("margin / gap not supported")
        kinds = {SPHERE: [0], CAPSULE: [1, 2], ELLIPSOID: [3]}[t2] if t1 == PLANE else [{(SPHERE, SPHERE): 4, (SPHERE, CAPSULE): 5, (CAPSULE, CAPSULE): 6}[(t1, t2)]]
        for k in kinds:
            con_geom1.append(g1); con_geom2.append(g2); con_kind.append(k); con_dim.append(int(cd))
            fr.append([f[0], f[0], f[1], f[2], f[2]]); sr.append(s_ref); si.append(s_imp)
    ncon = len(con_geom1)
    m["ncon"] = np.int32(ncon)
    m["con_geom1"] = np.asarray(con_geom1, np.int32).reshape(ncon)
    m["con_geom2"] = np.asarray(con_geom2, np.int32).reshape(ncon)
    m["con_kind"] = np.asarray(con_kind, np.int32).reshape(ncon)
    m["con_dim"] = np.asarray(con_dim, np.int32).reshape(ncon)      # condim: 3 = pyramid of 4 rows, 1 = frictionless, one normal row
    m["con_friction"] = np.asarray(fr, np.float64).reshape(ncon, 5)
    m["con_solref"] = np.asarray(sr, np.float64).reshape(ncon, 2)
    m["con_solimp"] = np.asarray(si, np.float64).reshape(ncon, 5)
    b1 = gb[m["con_geom1"]] if ncon else np.zeros(0, np.int32)
    b2 = gb[m["con_geom2"]] if ncon else np.zeros(0, np.int32)
    m["con_body1"], m["con_body2"] = b1.astype(np.int32), b2.astype(np.int32)
    # contacts between two moving bodies (kinds 4-6): J = J(body2) - J(body1); the CPU oracles take them, the HIP kernel's
    # J-free products walk ONE ancestor chain per contact (floor contacts), so such models get no kernel tables (hip_supported = 0)
    m["con_invweight"] = (m["body_invweight0"][b1, 0] + m["body_invweight0"][b2, 0]) if ncon else np.zeros(0)
    nlim = int(np.sum(m["jnt_limited"]))
    m["nlimit"] = np.int32(nlim)
    m["nefc"] = np.int32(nlim + sum(4 if c == 3 else 1 for c in con_dim))


# ----------------------------------------------------------------------------- engine tables
def _engine_tables(m):
    """Index tables the step engines (oracle C and HIP) share: tree levels, ancestor lists, J layout."""
    nb, nv = int(m["nbody"]), int(m["nv"])
    par = m["body_parentid"]
    depth = np.zeros(nb, np.int32)
    for b in range(1, nb):
        depth[b] = depth[par[b]] + 1
    m["body_depth"] = depth
    # ancestor dof chain of each dof, root first, self last (length depth+1)
    anc_adr = np.zeros(nv + 1, np.int32)
    anc = []
    for d in range(nv):
        chain = []
        k = d
        while k >= 0:
            chain.append(k)
            k = m["dof_parentid"][k]
        chain.reverse()
        anc_adr[d] = len(anc)
        anc += chain
    anc_adr[nv] = len(anc)
    m["dof_ancadr"] = anc_adr
    m["dof_anc"] = np.asarray(anc, np.int32)
    # contact Jacobian layout: contact c stores 3 base rows x nanc(c) ancestor dofs of its body
    ncon = int(m["ncon"])
    jadr = np.zeros(ncon + 1, np.int32)
    for c in range(ncon):
        d = m["body_lastdof"][m["con_body2"][c]]
        jadr[c + 1] = jadr[c] + 3 * (m["dof_depth"][d] + 1 if d >= 0 else 0)
    m["con_jadr"] = jadr
    m["con_lastdof"] = np.asarray([m["body_lastdof"][b] for b in m["con_body2"]], np.int32).reshape(ncon)
    # limited joints (hinge) in joint order
    lim = [j for j in range(int(m["njnt"])) if m["jnt_limited"][j]]
    m["limit_jnt"] = np.asarray(lim, np.int32)
    # actuator -> qpos / dof addresses of its transmission entries (joint transmission: one entry)
    m["actuator_moment_qposadr"] = m["jnt_qposadr"][m["actuator_moment_jnt"]].astype(np.int32)
    m["actuator_moment_dofadr"] = m["jnt_dofadr"][m["actuator_moment_jnt"]].astype(np.int32)
    first = m["actuator_momentadr"][:-1]
    m["actuator_qposadr"] = m["actuator_moment_qposadr"][first].astype(np.int32)      # (first entry: the joint of a joint transmission)
    m["actuator_dofadr"] = m["actuator_moment_dofadr"][first].astype(np.int32)
    # obs size of the reference env [REF Rodent_Env_Brax.py:147-158]
    m["obs_dim"] = np.int32(int(m["nq"]) + nv + 16 * (nb - 1) + nv + 3)


# ----------------------------------------------------------------------------- blob I/O
_MAGIC = b"RRM1"
_DT = {np.dtype(np.float32): 0, np.dtype(np.int32): 1}


def save_blob(m: Dict[str, np.ndarray], path: str):
    """Write the model tables as a flat binary blob (float64 -> float32, ints -> int32).

    Layout: magic 'RRM1', u32 n_entries, then per entry: name[32], u32 dtype (0=f32,1=i32),
    u32 ndim, u32 dims[4], u64 byte offset from file start, u64 nbytes; then 16-byte aligned data.
    """
    entries = []
    for k in sorted(m):
        if k.startswith("_"):
            continue
        v = np.asarray(m[k])
        v = v.astype(np.int32) if np.issubdtype(v.dtype, np.integer) else v.astype(np.float32)
        if v.ndim > 4:
            raise ValueError(k)
        entries.append((k, np.require(v, requirements='C')))
    header = 8 + len(entries) * (32 + 4 + 4 + 16 + 8 + 8)
    off = (header + 15) // 16 * 16
    recs, data = [], []
    for k, v in entries:
        nb_ = v.nbytes
        dims = list(v.shape) + [1] * (4 - v.ndim)
        recs.append(struct.pack("<32sII4IQQ", k.encode(), _DT[v.dtype], v.ndim, *dims, off, nb_))
        data.append((off, v.tobytes()))
        off = (off + nb_ + 15) // 16 * 16
    with open(path, "wb") as f:
        f.write(_MAGIC + struct.pack("<I", len(entries)))
        for r in recs:
            f.write(r)
        for o, b in data:
            f.seek(o)
            f.write(b)
        f.seek(off - 1) if off > 0 else None
        f.write(b"\0")


def load_blob(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != _MAGIC:
        raise ValueError(f"{path}: not an RRM1 model blob")
    n = struct.unpack_from("<I", raw, 4)[0]
    out = {}
    p = 8
    for _ in range(n):
        name, dt, nd, d0, d1, d2, d3, off, nbytes = struct.unpack_from("<32sII4IQQ", raw, p)
        p += 72
        dtype = np.float32 if dt == 0 else np.int32
        arr = np.frombuffer(raw, dtype=dtype, count=nbytes // 4, offset=off).reshape([d0, d1, d2, d3][:nd])
        out[name.rstrip(b"\0").decode()] = arr.copy()
    return out
