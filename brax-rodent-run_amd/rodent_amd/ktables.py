"""Kernel schedule tables: lane-major index tables for the one-wave-per-env HIP step kernel.

The fused step kernel (csrc/rr_kernel.h) runs one 64-lane wavefront per environment.  All of
its irregular access patterns (kinematic-tree level sweeps, sparse L'DL factor/solve, sparse
M*x, contact-Jacobian transpose products) are driven by static index tables computed here once
per model and stored in the model blob as `k_*` arrays.  Tables indexed by a loop iteration `t`
and a lane are laid out `[t][slot*64 + lane]` so one wave-load is a single coalesced 256-byte
read.  `slot` handles per-env element counts above 64 (dof d lives at lane d%64, slot d//64).

Packed entry format (int32): low 8 bits = element index, upper bits = LDS address offset,
-1 = no entry.
"""
from __future__ import annotations

import numpy as np

LANES = 64


def _slots(n):
    return max(1, (int(n) + LANES - 1) // LANES)


def replica_model(m):
    """A model of two identical kinematic trees (`<replicate count="2">`, rodent_pair.xml [REF models/rodent_pair.xml:163-500]) seen as
    ONE tree: the MuJoCo-named tables of replica 0 with ids re-based (world body 0 kept), or None when the model is not of that
    shape.  The two trees share no constraint (M block-diagonal, floor contacts only), so the step kernel can give each replica its own
    wavefront running on these tables and couple the two only through the solver's scalar sums (csrc/rr_kernel.h, PAIR instances).
    Every table that enters the dynamics must be identical for the two replicas (checked); what differs -- the root's qpos0 -- does
    not enter them (a free joint takes its pose from qpos)."""
    nb, nv, njnt, nq, nu, ncon = (int(m[k]) for k in ("nbody", "nv", "njnt", "nq", "nu", "ncon"))
    roots = sorted(set(int(r) for r in m["body_rootid"][1:]))
    if len(roots) != 2 or (nb - 1) % 2 or nv % 2 or njnt % 2 or nq % 2 or nu % 2 or ncon % 2 or int(m["na"]) != nu:
        return None
    hb, hv, hj, hq, hu = (nb - 1) // 2, nv // 2, njnt // 2, nq // 2, nu // 2
    if roots != [1, 1 + hb]:
        return None
    FREE = 0
    if m["jnt_type"][0] != FREE or m["jnt_type"][hj] != FREE or m["body_parentid"][1] != 0 or m["body_parentid"][1 + hb] != 0:
        return None
    h = {}

    def same(a, b, what):
        a, b = np.asarray(a), np.asarray(b)
        ok = a.shape == b.shape and (np.allclose(a, b, rtol=1e-7, atol=1e-12) if a.dtype.kind == "f" else np.array_equal(a, b))     # float32 on disk
        if not ok:
            raise ValueError(f"replicas differ in {what}: the two-wave instance needs identical trees")

    def body(name, rebase=None, root_free=False):
        a = np.asarray(m[name])
        r0, r1 = a[1:1 + hb].copy(), a[1 + hb:1 + 2 * hb].copy()
        if rebase is not None:
            r0, r1 = rebase(r0, 0), rebase(r1, 1)
        if root_free:             # the pose of a free-joint body comes from qpos
            r1[0] = r0[0]
        same(r0, r1, name)
        h[name] = np.concatenate([a[:1], r0])

    def rows(name, n, rebase=None):
        a = np.asarray(m[name])
        r0, r1 = a[:n].copy(), a[n:2 * n].copy()
        if rebase is not None:
            r0, r1 = rebase(r0, 0), rebase(r1, 1)
        same(r0, r1, name)
        h[name] = r0

    reb = lambda step, keep=None: (lambda x, r: np.where(x == keep, x, x - r * step) if keep is not None else x - r * step)
    body("body_parentid", lambda x, r: np.where(x > 0, x - r * hb, 0))
    body("body_rootid", reb(hb))
    body("body_depth"); body("body_subtreemass"); body("body_jntnum"); body("body_dofnum"); body("body_mass"); body("body_inertia")
    body("body_jntadr", reb(hj, -1)); body("body_dofadr", reb(hv, -1)); body("body_lastdof", reb(hv, -1))
    body("body_pos", root_free=True); body("body_quat", root_free=True); body("body_ipos"); body("body_iquat"); body("body_invweight0")
    rows("jnt_type", hj); rows("jnt_pos", hj); rows("jnt_axis", hj); rows("jnt_limited", hj); rows("jnt_stiffness", hj)
    rows("jnt_range", hj); rows("jnt_solref", hj); rows("jnt_solimp", hj)
    rows("jnt_qposadr", hj, reb(hq)); rows("jnt_dofadr", hj, reb(hv)); rows("jnt_bodyid", hj, reb(hb))
    rows("dof_parentid", hv, reb(hv, -1)); rows("dof_depth", hv); rows("dof_jntid", hv, reb(hj)); rows("dof_bodyid", hv, reb(hb))
    rows("dof_armature", hv); rows("dof_damping", hv); rows("dof_invweight0", hv)
    hM = int(m["nM"]) // 2
    rows("dof_Madr", hv, reb(hM))
    rows("actuator_dofadr", hu, reb(hv)); rows("actuator_gainprm0", hu); rows("actuator_biasprm", hu); rows("actuator_dynprm0", hu)
    rows("actuator_ctrlrange", hu)
    madr = np.asarray(m["actuator_momentadr"])
    if np.any(np.diff(madr) != 1):
        return None                       # joint transmissions only
    rows("actuator_moment_dofadr", hu, reb(hv)); rows("actuator_moment_qposadr", hu, reb(hq)); rows("actuator_moment_coef", hu)
    h["actuator_momentadr"] = np.arange(hu + 1, dtype=np.int32)
    # qpos0 / qpos_spring: hinge entries identical, the free joint's 7 are not used by the dynamics
    q0, q1 = np.asarray(m["qpos0"])[:hq].copy(), np.asarray(m["qpos0"])[hq:].copy()
    q1[:7] = q0[:7]
    same(q0, q1, "qpos0 (hinges)")
    h["qpos0"] = q0
    s0, s1 = np.asarray(m["qpos_spring"])[:hq].copy(), np.asarray(m["qpos_spring"])[hq:].copy()
    s1[:7] = s0[:7]
    same(s0, s1, "qpos_spring (hinges)")
    h["qpos_spring"] = s0
    adr = np.asarray(m["dof_ancadr"])
    a0 = np.asarray(m["dof_anc"])[adr[0]:adr[hv]]
    a1 = np.asarray(m["dof_anc"])[adr[hv]:adr[2 * hv]] - hv
    same(a0, a1, "dof_anc")
    same(adr[:hv + 1], adr[hv:] - adr[hv], "dof_ancadr")
    h["dof_anc"], h["dof_ancadr"] = a0.copy(), adr[:hv + 1].copy()
    # contacts of each replica (the pair's list is grouped by geom type first, so a replica's contacts are not contiguous)
    cb = np.asarray(m["con_body2"])
    if np.any(np.asarray(m["con_body1"]) != 0):
        return None
    sel = [np.nonzero((cb >= 1 + r * hb) & (cb < 1 + (r + 1) * hb))[0] for r in (0, 1)]
    if len(sel[0]) != ncon // 2 or len(sel[1]) != ncon // 2:
        return None
    for name in ("con_kind", "con_dim", "con_friction", "con_invweight", "con_solref", "con_solimp"):
        same(np.asarray(m[name])[sel[0]], np.asarray(m[name])[sel[1]], name)
        h[name] = np.asarray(m[name])[sel[0]].copy()
    same(cb[sel[0]], cb[sel[1]] - hb, "con_body2")
    same(np.asarray(m["con_lastdof"])[sel[0]], np.asarray(m["con_lastdof"])[sel[1]] - hv, "con_lastdof")
    g2 = np.asarray(m["con_geom2"])
    for name in ("geom_pos", "geom_quat", "geom_size"):
        same(np.asarray(m[name])[g2[sel[0]]], np.asarray(m[name])[g2[sel[1]]], name + " of the colliding geoms")
    same(np.asarray(m["con_geom1"])[sel[0]], np.asarray(m["con_geom1"])[sel[1]], "con_geom1")
    h["con_body1"] = np.zeros(ncon // 2, np.int32)
    h["con_body2"], h["con_lastdof"] = cb[sel[0]].copy(), np.asarray(m["con_lastdof"])[sel[0]].copy()
    h["con_geom1"], h["con_geom2"] = np.asarray(m["con_geom1"])[sel[0]].copy(), g2[sel[0]].copy()
    jadr = np.zeros(ncon // 2 + 1, np.int32)
    for c, d in enumerate(h["con_lastdof"]):
        jadr[c + 1] = jadr[c] + 3 * (h["dof_depth"][d] + 1 if d >= 0 else 0)
    h["con_jadr"] = jadr
    for name in ("geom_pos", "geom_quat", "geom_size", "opt_impratio"):      # indexed by the (unchanged) geom ids
        h[name] = m[name]
    h.update(nbody=np.int32(1 + hb), nv=np.int32(hv), njnt=np.int32(hj), nq=np.int32(hq), nu=np.int32(hu), na=np.int32(hu),
             nM=np.int32(hM), ncon=np.int32(ncon // 2))
    h["_contacts_of_replica"] = np.stack(sel).astype(np.int32)
    return h


def build_kernel_tables(m):
    nb, nv, njnt, nM = int(m["nbody"]), int(m["nv"]), int(m["njnt"]), int(m["nM"])
    ncon = int(m["ncon"])
    par = m["body_parentid"]
    dpar = m["dof_parentid"]
    Madr = m["dof_Madr"]
    ddepth = m["dof_depth"]
    # DYN models (rodent_cpu.xml): contacts between two moving bodies (sphere / capsule pairs), condim 1 rows,
    # tendon transmissions.  The contact list is then a list of CANDIDATE pairs: the kernel scans it every substep and keeps the pairs
    # in penetration in the 64 contact slots of the wave (`rr_step_kernel<..., DYN>`).
    dyn = bool(ncon and (np.any(np.asarray(m["con_body1"]) != 0) or ("con_dim" in m and np.any(np.asarray(m["con_dim"]) != 3)))) or \
        bool(np.any(np.diff(np.asarray(m["actuator_momentadr"])) != 1))
    if dyn and ncon and not np.all(np.isin(np.asarray(m["con_kind"]), (4, 5, 6))):
        raise ValueError("a model with contacts between moving bodies must have only sphere / capsule pairs (no floor contacts) on the HIP path")
    NBS, NVS, NCS = _slots(nb), _slots(nv), (1 if dyn else _slots(ncon))
    k = {}
    k["k_slots"] = np.array([NBS, NVS, NCS], np.int32)
    k["k_dyn"] = np.int32(dyn)

    depth = m["body_depth"]
    nlevel = int(depth.max())
    # roots (kinematic trees) and their total mass
    roots = sorted(set(int(r) for r in m["body_rootid"][1:]))
    k["k_root"] = np.asarray(roots, np.int32)
    k["k_root_mass"] = np.asarray([m["body_subtreemass"][r] for r in roots], np.float64)
    root_index = {r: i for i, r in enumerate(roots)}
    k["k_body_root"] = np.asarray([root_index.get(int(r), 0) for r in m["body_rootid"]], np.int32)

    # ---- packed per-body / per-joint / per-dof parameter rows
    sib = np.zeros(nb, np.int32)                      # rank among siblings, largest body id first: the
    for b in range(nb):                               # backward sweeps add children into the parent in that
        kids = [c for c in range(1, nb) if par[c] == b]   # order (= the reference's `for b = nb-1..1` loop)
        for r, c in enumerate(sorted(kids, reverse=True)):
            sib[c] = r
    if int(m["body_jntnum"].max()) > 3:
        raise ValueError("more than 3 joints on one body is not supported by the kernel")
    blast = np.arange(nb, dtype=np.int32)                 # last body of each subtree (bodies are in DFS order)
    for b in range(nb - 1, 0, -1):
        blast[par[b]] = max(blast[par[b]], blast[b])
    for b in range(1, nb):
        assert par[b] < b and all(par[c] >= b or c > blast[b] for c in range(b + 1, nb) if c > blast[b] or True)
    k["k_body_i"] = np.stack([par, m["body_jntadr"], m["body_jntnum"], m["body_dofadr"], m["body_dofnum"],
                              k["k_body_root"], np.zeros(nb, np.int32), np.zeros(nb, np.int32), depth, sib,
                              blast, np.zeros(nb, np.int32)], axis=1).astype(np.int32)   # 12 ints
    # pointer-doubling ancestor table: byte k of (anc[2b], anc[2b+1]) = 2^k-th ancestor of body b (0 = none / world)
    nround = max(1, int(np.ceil(np.log2(max(2, nlevel)))))
    if nb > 255 or nround > 8:
        raise ValueError("body count / tree depth above the packed ancestor table")
    anck = np.zeros((nround, nb), np.int64)
    anck[0] = par
    for r in range(1, nround):
        anck[r] = anck[r - 1][anck[r - 1]]
    packed_anc = np.zeros((nb, 2), np.int64)
    for r in range(nround):
        packed_anc[:, r // 4] |= anck[r] << (8 * (r % 4))
    k["k_body_anc"] = packed_anc.astype(np.uint32).view(np.int32)
    k["k_nround"] = np.int32(nround)
    k["k_body_f"] = np.concatenate([m["body_pos"], m["body_quat"], m["body_ipos"], m["body_iquat"],
                                    m["body_mass"][:, None], m["body_inertia"]], axis=1)  # 18 floats
    qa = m["jnt_qposadr"]
    k["k_jnt_i"] = np.stack([m["jnt_type"], qa, m["jnt_dofadr"], m["jnt_bodyid"]], axis=1).astype(np.int32)
    k["k_jnt_f"] = np.concatenate([m["jnt_pos"], m["jnt_axis"], m["qpos0"][qa][:, None],
                                   np.zeros((njnt, 1))], axis=1)  # 8 floats
    # per dof: kind 0-2 free translation x/y/z, 3-5 free rotation x/y/z, 6 hinge
    kind = np.zeros(nv, np.int32)
    dof_qposadr = np.zeros(nv, np.int32)
    act_of_dof = np.full(nv, -1, np.int32)
    for d in range(nv):
        j = m["dof_jntid"][d]
        if m["jnt_type"][j] == 0:
            kind[d] = d - m["jnt_dofadr"][j]
            dof_qposadr[d] = m["jnt_qposadr"][j] + kind[d]     # translation: qpos index; rotation: unused
        else:
            kind[d] = 6
            dof_qposadr[d] = m["jnt_qposadr"][j]
    # transmission [UP mjx smooth.transmission]: actuator u acts on the dofs of its moment entries with coefficients (a joint: one entry,
    # coefficient 1; a fixed tendon: its joints).  Per dof: the one actuator that drives it and its coefficient.
    act_coef = np.zeros(nv)
    madr = np.asarray(m["actuator_momentadr"])
    for u in range(int(m["nu"])):
        for e in range(int(madr[u]), int(madr[u + 1])):
            d = int(m["actuator_moment_dofadr"][e])
            if act_of_dof[d] >= 0:
                raise ValueError("two actuators on one dof are not supported")
            act_of_dof[d] = u
            act_coef[d] = float(m["actuator_moment_coef"][e])
    k["k_act_i"] = np.stack([madr[:-1], np.diff(madr)], axis=1).astype(np.int32)                      # first moment entry, count
    k["k_act_m_i"] = np.stack([m["actuator_moment_dofadr"], m["actuator_moment_qposadr"]], axis=1).astype(np.int32)
    k["k_act_m_f"] = np.asarray(m["actuator_moment_coef"], np.float64)
    if int(np.diff(madr).max()) > 16:
        raise ValueError("more than 16 joints in one tendon transmission")
    lim = np.zeros(nv, np.int32)
    for j in range(njnt):
        if m["jnt_limited"][j]:
            lim[m["jnt_dofadr"][j]] = 1
    jid = m["dof_jntid"]
    last_desc = np.arange(nv, dtype=np.int32)
    for d in range(nv - 1, -1, -1):
        if dpar[d] >= 0:
            last_desc[dpar[d]] = max(last_desc[dpar[d]], last_desc[d])
    for d in range(nv):                                   # dofs must be in DFS preorder (MuJoCo order)
        p = dpar[d]
        while p >= 0:
            assert p < d <= last_desc[p]
            p = dpar[p]
    k["k_dof_i"] = np.stack([m["dof_bodyid"], jid, kind, ddepth, Madr, dpar, dof_qposadr, act_of_dof, lim,
                             k["k_body_root"][m["dof_bodyid"]], last_desc, np.zeros(nv, np.int32)],
                            axis=1).astype(np.int32)                                           # 12 ints
    hinge = (kind == 6)
    k["k_dof_f"] = np.stack([
        m["dof_armature"], m["dof_damping"], np.where(hinge, m["jnt_stiffness"][jid], 0.0),
        np.where(hinge, m["qpos_spring"][dof_qposadr], 0.0),
        m["jnt_range"][jid, 0], m["jnt_range"][jid, 1], m["jnt_solref"][jid, 0], m["jnt_solref"][jid, 1],
        m["jnt_solimp"][jid, 0], m["jnt_solimp"][jid, 1], m["jnt_solimp"][jid, 2], m["jnt_solimp"][jid, 3],
        m["jnt_solimp"][jid, 4], m["dof_invweight0"], act_coef, np.zeros(nv)], axis=1)         # 16 floats
    nu = int(m["nu"])
    k["k_act_f"] = np.stack([m["actuator_gainprm0"], m["actuator_biasprm"][:, 0], m["actuator_biasprm"][:, 1],
                             m["actuator_biasprm"][:, 2], m["actuator_dynprm0"], m["actuator_ctrlrange"][:, 0],
                             m["actuator_ctrlrange"][:, 1], np.zeros(nu)], axis=1)                   # 8 floats

    # ---- sparse M: element -> (row dof i, col dof j), and row address of the column's own row
    anc_adr, anc = m["dof_ancadr"], m["dof_anc"]       # chain root..self
    M_ij = np.zeros(nM, np.int32)
    for i in range(nv):
        chain = anc[anc_adr[i]:anc_adr[i + 1]][::-1]   # self, parent, ..., root
        for p, j in enumerate(chain):
            M_ij[Madr[i] + p] = i | (int(j) << 16)
    # kernel layout: row | col << 8 per entry, padded with -1 to whole 64-lane rows plus one row of slack (prefetch)
    nrow = max((nM + LANES - 1) // LANES + 1, {1: 10, 2: 18, 3: 35}.get(NVS, 35))   # the kernel reads Wave::NME whole rows
    mk = np.full(nrow * LANES, -1, np.int32)
    mk[:nM] = (M_ij & 0xFFFF) | ((M_ij >> 16) << 8)
    k["k_M_ij"] = M_ij
    k["k_M_ij_k"] = mk
    dmax = int(ddepth.max())
    Wd = NVS * LANES
    by_level = [[i for i in range(nv) if ddepth[i] == l] for l in range(dmax + 1)]
    if 8 * (nM + 20) >= 65536 or nv >= 255:
        raise ValueError("nM above the 12-bit address field of the solve job tables")

    # ---- atomic-free factorisation and inversion schedules (levelsched.py): table rows of 64 independent QUAD operations on the
    # sparse-matrix array, four targets d0..d3 -= src_a * src_(b0 + j) [/ piv], j = 0..3 -- the targets of one source row are
    # consecutive entries of an ancestor row, so one shared operand and one run of four feed four multiply-adds.  The operations
    # of each schedule are list-scheduled as ONE dependency graph (a row sees what earlier rows wrote: the LDS instructions of a
    # wavefront execute in order), the side branches of a fork accumulating into private alias copies of their ancestors' rows.
    # Operation = 4 ints:  x = a | b0 << 16,  y = d0 | d1 << 16,  z = d2 | d3 << 16,  w = q  (element indices; piv = a + 1 - q).
    # Cells behind the nM entries: ZERO = nM (0.0), ONE = nM + 1 (1.0), TRASH = nM + 2, MINUS_ONE = nM + 3 (-1.0), alias cells from
    # nM + 4 (k_nalias of them; the host maps them to the pose cells, dead during the factorisation): an empty operation is
    # a = b0 = ZERO, q = 0 (piv = ONE), d* = TRASH, a short run sends its unused targets to TRASH, so the kernel needs no predicates.
    RING = 8                  # rows in flight (RR_RING): that many empty rows follow each schedule
    from . import levelsched
    # where the host puts the two regions (csrc/rr_kernel.h rr_layout, float offsets; used for the lane assignment's bank model only)
    up4 = lambda x: (x + 3) // 4 * 4
    o_xpos = up4(int(m["nq"])) + up4(nv + 1) + 2 * up4(int(m["nu"]))
    o_qLD = o_xpos + up4(max(7 * nb + 4, 6 * nv)) + up4(10 * nb) + up4(6 * nv + 6) + up4(6 * nb)
    k.update(levelsched.build(ddepth, Madr, dpar, last_desc, nM, alias_cells=max(7 * nb + 4, 6 * nv) // 2, ring=RING,
                              matrix_slot=o_qLD // 2, alias_slot=o_xpos // 2))

    # ---- balanced jobs of the two sparse products of the solve (Wave::ldl_solve).  Column product (U' b): column j sums
    # over its descendants i = j+1 .. last_desc[j]; row product (U y): row i sums over its ancestors p = 1 .. depth[i].  Both
    # total nM - nv multiply-adds but the longest column / row is ~nv / ~dmax long, so each is cut into pieces of at most
    # LMAX entries, one piece per lane-slot (NVS * 64 of them); the owner of a column / row adds up its pieces' partial sums.
    nslot = (NVS + 1 if NVS >= 3 else NVS) * LANES          # Wave::NJS job slots per lane
    lmax = 1
    while True:
        ncj = sum(-(-int(last_desc[j] - j) // lmax) for j in range(nv))
        nrj = sum(-(-int(ddepth[i]) // lmax) for i in range(nv))
        if ncj <= nslot and nrj <= nslot:
            break
        lmax += 1
    if lmax > 16:
        raise ValueError("solve jobs longer than 16 entries")
    # Predicate-free job descriptors (the kernel runs every job for lmax steps).  The matrix array holds PAIRS (entry of
    # M's factor, entry of the eulerdamp matrix' factor), 8 bytes per entry: a column job lists the byte offsets
    # 8 * (base[i] - depth[j]) of its matrix entries, 2 per int, padded with the ZERO cell 8 * nM, and walks the vector from
    # i0; a row job lists its ancestor dof ids, 4 per int, padded with nv (a vector cell that always holds 0), and walks
    # the matrix row from byte offset 8 * (Madr[i] + p0).
    coljob = np.zeros((9, nslot), np.int64)     # [0..7]: byte offsets (16 bits each), [8]: i0
    rowjob = np.zeros((5, nslot), np.int64)     # [0..3]: ancestor ids (8 bits each), [4]: byte offset of the first entry
    for t in range(nslot):
        for u in range(16):
            coljob[u >> 1, t] |= (8 * nM) << (16 * (u & 1))
            rowjob[u >> 2, t] |= nv << (8 * (u & 3))
    own = np.zeros(nv, np.int64)                # first column job | count << 8 | first row job << 16 | count << 24
    t = 0
    for j in range(nv):
        n = int(last_desc[j] - j)
        c = -(-n // lmax)
        own[j] |= t | (c << 8)
        for r in range(c):
            i0 = j + 1 + r * n // c
            i1 = j + 1 + (r + 1) * n // c
            coljob[8, t] = i0
            for u, i in enumerate(range(i0, i1)):
                coljob[u >> 1, t] &= ~(0xFFFF << (16 * (u & 1)))
                coljob[u >> 1, t] |= (8 * int(Madr[i] + ddepth[i] - ddepth[j])) << (16 * (u & 1))
            t += 1
    t = 0
    for i in range(nv):
        n = int(ddepth[i])
        c = -(-n // lmax)
        own[i] |= (t << 16) | (c << 24)
        chain = anc[anc_adr[i]:anc_adr[i + 1]][::-1]     # self, parent, ..., root
        for r in range(c):
            p0 = 1 + r * n // c
            p1 = 1 + (r + 1) * n // c
            rowjob[4, t] = 8 * int(Madr[i] + p0)
            for u, pp in enumerate(range(p0, p1)):
                rowjob[u >> 2, t] &= ~(0xFF << (8 * (u & 3)))
                rowjob[u >> 2, t] |= int(chain[pp]) << (8 * (u & 3))
            t += 1
    k["k_solve_lmax"] = np.int32(lmax)
    k["k_solve_cmax"] = np.int32(max(int((o >> 8) & 255) for o in own))     # most pieces of one column / row
    k["k_solve_rmax"] = np.int32(max(int(o >> 24) for o in own))
    k["k_coljob"] = (coljob & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    k["k_rowjob"] = (rowjob & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    k["k_jobown"] = (own & 0xFFFFFFFF).astype(np.uint32).view(np.int32)

    # ---- contacts
    WC = NCS * LANES
    if dyn:
        _pair_tables(m, k, anc, anc_adr, ddepth, last_desc)
        return k
    g2 = m["con_geom2"]
    g1 = m["con_geom1"]
    body = m["con_body2"]
    lastdof = m["con_lastdof"]
    nanc = np.asarray([(ddepth[d] + 1) if d >= 0 else 0 for d in lastdof], np.int32).reshape(ncon)
    k["k_con_i"] = np.stack([m["con_kind"], body, k["k_body_root"][body] if ncon else np.zeros(0, np.int32),
                             lastdof, nanc, m["con_jadr"][:-1], g1, g2], axis=1).astype(np.int32).reshape(ncon, 8)
    # plane (geom1) is on the world body: world pose = local pose
    from .mjcf import quat_to_mat
    pn = np.array([quat_to_mat(m["geom_quat"][g])[:, 2] for g in g1]).reshape(ncon, 3)
    mu = m["con_friction"][:, 0] if ncon else np.zeros(0)
    invw = (m["con_invweight"] + mu * mu * m["con_invweight"]) * 2 * mu * mu / float(m["opt_impratio"])
    k["k_con_f"] = np.concatenate([
        m["geom_pos"][g2].reshape(ncon, 3), m["geom_quat"][g2].reshape(ncon, 4), m["geom_size"][g2].reshape(ncon, 3),
        pn, m["geom_pos"][g1].reshape(ncon, 3), mu[:, None], invw[:, None],
        m["con_solref"].reshape(ncon, 2), m["con_solimp"].reshape(ncon, 5), np.zeros((ncon, 1))], axis=1)  # 26 floats
    # chain of ancestor dofs per contact, [p][lane], leaf first so shallow chains end early
    maxc = int(nanc.max()) if ncon else 1
    chain_tab = np.full((max(1, maxc), WC), -1, np.int32)
    for c in range(ncon):
        d = int(lastdof[c])
        if d < 0:
            continue
        chain = anc[anc_adr[d]:anc_adr[d + 1]][::-1]   # leaf..root
        for p, dd in enumerate(chain):
            chain_tab[p, c] = int(dd)
    if maxc > 36:
        raise ValueError("contact ancestor chains longer than 36 dofs are not supported by the kernel")
    packed = np.zeros((9, WC), np.int64)
    for c in range(ncon):
        for p in range(int(nanc[c])):
            packed[p // 4, c] |= int(chain_tab[p, c]) << (8 * (p % 4))
    k["k_con_chain_packed"] = packed.astype(np.uint32).view(np.int32)
    # the same chains contact-major: 9 ints (36 dof ids, leaf first) per contact, + one row of slack; the J*x jobs of the
    # kernel (Wave::contact_jobs) read aligned pieces of 12 / 20 / 36 ids from it
    rows9 = np.zeros((ncon + 1, 9), np.int64)
    for c in range(ncon):
        for p in range(int(nanc[c])):
            rows9[c, p // 4] |= int(chain_tab[p, c]) << (8 * (p % 4))
    k["k_con_chain_rows"] = (rows9 & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(-1)
    return k


def _pair_tables(m, k, anc, anc_adr, ddepth, last_desc):
    """Candidate-pair tables of a DYN model (two moving geoms per contact): per pair 8 ints (kind, body1, body2, tree, signed-chain length,
    last dof of body1, last dof of body2, rows: 1 = frictionless, 4 = pyramid), 32 floats (geom1 local pos / quat / size, geom2 the same,
    mu, effective invweight, solref, solimp), and the SIGNED dof chain of J = jac(body2) - jac(body1): the dofs on exactly one of the
    two ancestor chains, id | 128 for body1's (they enter with a minus sign), padded with nv (a zero motion vector), 40 ids = 10 ints."""
    ncon, nv = int(m["ncon"]), int(m["nv"])
    if nv >= 128:
        raise ValueError("signed dof ids need nv < 128")
    ld = np.asarray(m["body_lastdof"])
    g1, g2 = np.asarray(m["con_geom1"]), np.asarray(m["con_geom2"])
    b1, b2 = np.asarray(m["con_body1"]), np.asarray(m["con_body2"])
    mu = np.asarray(m["con_friction"])[:, 0]
    dim = np.asarray(m["con_dim"])
    invw = np.where(dim == 1, m["con_invweight"], (m["con_invweight"] + mu * mu * m["con_invweight"]) * 2 * mu * mu / float(m["opt_impratio"]))
    pi = np.zeros((ncon, 8), np.int32)
    rows10 = np.full((ncon + 1, 40), nv, np.int64)
    root = k["k_body_root"]
    for c in range(ncon):
        def chain(b):
            d = int(ld[b])
            return [int(x) for x in anc[anc_adr[d]:anc_adr[d + 1]]] if d >= 0 else []
        c1, c2 = chain(b1[c]), chain(b2[c])
        s1, s2 = set(c1), set(c2)
        sig = [d for d in c2 if d not in s1] + [d | 128 for d in c1 if d not in s2]
        if len(sig) > 40:
            raise ValueError("contact chains longer than 40 dofs are not supported by the kernel")
        rows10[c, :len(sig)] = sig
        pi[c] = (m["con_kind"][c], b1[c], b2[c], root[b2[c]], len(sig), ld[b1[c]], ld[b2[c]], 1 if dim[c] == 1 else 4)
        if root[b1[c]] != root[b2[c]]:
            raise ValueError("contacts between two kinematic trees are not supported")
    k["k_con_i"] = pi
    k["k_con_f"] = np.concatenate([
        m["geom_pos"][g1], m["geom_quat"][g1], m["geom_size"][g1], m["geom_pos"][g2], m["geom_quat"][g2], m["geom_size"][g2],
        mu[:, None], invw[:, None], m["con_solref"].reshape(ncon, 2), m["con_solimp"].reshape(ncon, 5), np.zeros((ncon, 3))], axis=1)   # 32 floats
    packed = np.zeros((ncon + 1, 10), np.int64)
    for j in range(40):
        packed[:, j // 4] |= rows10[:, j] << (8 * (j % 4))
    k["k_con_chain_rows"] = (packed & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(-1)
    k["k_con_chain_packed"] = np.zeros((9, LANES), np.int32)        # (static-slot table of the floor-contact instances: unused)
