"""Host logic: the kernel schedule tables (rodent_amd/ktables.py) executed by a numpy restatement of
the kernel's loops must reproduce the oracle's dense results (factor, solve, M*x)."""
import numpy as np
import pytest

from rodent_amd import assets, mjcf

LANES = 64


def dense_from_sparse(m, qM):
    nv = int(m["nv"])
    M = np.zeros((nv, nv))
    for e, ij in enumerate(m["k_M_ij"]):
        i, j = ij & 0xFFFF, ij >> 16
        M[i, j] = M[j, i] = qM[e]
    return M


@pytest.fixture(scope="module", params=["rodent_optimized", "rodent_pair"])
def model_and_state(request, oracle_built):
    ref = oracle_built
    path = assets.asset_path(request.param)
    m = mjcf.load_blob(path)
    M = ref.RefModel(path, "f64")
    d = ref.RefData(M)
    rng = np.random.default_rng(0)
    q = m["qpos0"].astype(np.float64).copy()
    q += rng.uniform(-0.05, 0.05, q.size)
    d.init(q, rng.uniform(-0.5, 0.5, M.nv))
    return m, M, d


def run_jobs(m, mat, vec, sign):
    """numpy restatement of the job loops shared by Wave::ldl_solve and Wave::mul_m: every lane-slot runs its column job and
    its row job for lmax steps without predicates (padding = the matrix array's ZERO cell / the vector's zero cell);
    returns, per dof, the sums of its column pieces and of its row pieces.  `mat` carries nM entries + zero cell(s)."""
    nv, nM = int(m["nv"]), int(m["nM"])
    lmax = int(m["k_solve_lmax"])
    cj, rj, own = m["k_coljob"].view(np.uint32).astype(np.int64), m["k_rowjob"].view(np.uint32).astype(np.int64), m["k_jobown"].view(np.uint32).astype(np.int64)
    assert mat[nM] == 0.0
    x = np.concatenate([vec, [0.0]])                       # vector cell nv always holds 0
    xpad = np.concatenate([vec, np.full(32, 1e30)])        # what a padded column step may read: anything finite
    matpad = np.concatenate([mat, np.full(32, 1e30)])
    nslot = cj.shape[1]
    pc, pr = np.zeros(nslot), np.zeros(nslot)
    for t in range(nslot):
        i0, adr0 = int(cj[8, t]), int(rj[4, t])
        assert adr0 % 8 == 0
        for u in range(lmax):
            off = int(cj[u >> 1, t] >> (16 * (u & 1))) & 0xFFFF
            assert off % 8 == 0 and off // 8 <= nM
            pc[t] += mat[off // 8] * xpad[i0 + u]
            a = int(rj[u >> 2, t] >> (8 * (u & 3))) & 255
            assert a <= nv
            pr[t] += matpad[adr0 // 8 + u] * x[a]
        for u in range(lmax, 16):                            # beyond lmax the descriptors are padding
            assert (int(cj[u >> 1, t] >> (16 * (u & 1))) & 0xFFFF) == 8 * nM and (int(rj[u >> 2, t] >> (8 * (u & 3))) & 255) == nv
    col = np.array([pc[int(o & 255):int(o & 255) + int((o >> 8) & 255)].sum() for o in own])
    row = np.array([pr[int((o >> 16) & 255):int((o >> 16) & 255) + int(o >> 24)].sum() for o in own])
    return sign * col, sign * row


def test_mulm_by_solve_jobs(model_and_state):
    """numpy restatement of Wave::mul_m: y = M x from the balanced column / row jobs of the solve, on qM."""
    m, M, d = model_and_state
    nv = M.nv
    qM = d.get("qM")
    Md = dense_from_sparse(m, qM)
    x = np.random.default_rng(1).normal(size=nv)
    Madr = m["k_dof_i"][:, 4]
    col, row = run_jobs(m, np.concatenate([qM, [0.0]]), x, 1.0)
    np.testing.assert_allclose(qM[Madr] * x + col + row, Md @ x, rtol=1e-12, atol=1e-14)


RING = 8        # rows in flight (RR_RING)


def _cells(nM, nalias):
    """[ZERO, ONE, TRASH, MINUS_ONE] + zeroed alias cells + 4 spare cells"""
    return np.concatenate([[0.0, 1.0, 0.0, -1.0], np.zeros(nalias + 4)])


def run_rows(L, nM, tab, nrows, div):
    assert L[nM] == 0.0 and L[nM + 1] == 1.0 and L[nM + 3] == -1.0
    assert tab.shape[0] == nrows + RING                     # RING empty rows follow: the executor prefetches that far
    u32 = tab.view(np.uint32).astype(np.int64)
    used = 0
    for r in range(nrows + RING):
        e = u32[r]
        a, b_ = e[:, 0] & 0xFFFF, e[:, 0] >> 16
        d = np.stack([e[:, 1] & 0xFFFF, e[:, 1] >> 16, e[:, 2] & 0xFFFF, e[:, 2] >> 16], axis=1)
        q = e[:, 3]
        assert np.all(q < 256)                               # no flags: rows carry no hand-offs
        real = a != nM
        if r >= nrows:
            assert not real.any()
            continue
        used += int(real.sum())
        assert np.all(b_[~real] == nM) and np.all(q[~real] == 0) and np.all(d[~real] == nM + 2)
        valid = d != nM + 2                                  # targets of short runs and of empty operations go to TRASH
        assert np.all(valid[real, 0]) and np.all(valid[:, :-1] >= valid[:, 1:])      # the valid targets of a quad come first
        assert np.all(b_[real] + 3 < len(L)) and np.all(d < len(L) - 4)
        snap = L.copy()                                      # what the row's reads see
        if div:
            piv = a + 1 - q
            assert np.all(piv[~real] == nM + 1)
            t = snap[a] / snap[piv]
        else:
            assert np.all(q == 0)
            t = snap[a]
        tg = d[valid]
        assert len(set(tg.tolist())) == len(tg)              # plain read-modify-writes: no two lanes / slots share a target
        for j in range(4):
            v = valid[:, j]
            L[d[v, j]] = snap[d[v, j]] - snap[b_[v] + j] * t[v]
        L[nM:nM + 4] = [0.0, 1.0, 0.0, -1.0]                 # whatever landed in TRASH is never used
    return L, used


def kernel_factor(m, qM, plain=False):
    """numpy restatement of Wave::factor: the row schedule k_factor3 (with alias copies; k_factor3p without), then the row scaling by 1/D."""
    nM = len(qM)
    name, nalias = ("k_factor3p", 0) if plain else ("k_factor3", int(m["k_nalias"]))
    L, used = run_rows(np.concatenate([qM, _cells(nM, nalias)]), nM, m[name], int(m[name + "_rows"]), True)
    L = L[:nM]
    Madr = m["k_dof_i"][:, 4]
    dinv = 1.0 / L[Madr]
    ij = m["k_M_ij"]
    i, j = ij & 0xFFFF, ij >> 16
    off = i != j
    L[off] *= dinv[i[off]]
    return L, dinv


def kernel_invert(m, L):
    """numpy restatement of Wave::invert: W = I - L^-1 in place (k_linv)."""
    nM = len(L)
    return run_rows(np.concatenate([L, _cells(nM, 0)]), nM, m["k_linv"], int(m["k_linv_rows"]), False)[0][:nM]


def kernel_solve(m, W, dinv, b):
    """numpy restatement of Wave::ldl_solve with the explicit inverse: x = U D^-1 U' b, U = I - W, both products cut into
    balanced per-lane-slot jobs whose partial sums the owner of the column / row adds up."""
    Wz = np.concatenate([W, [0.0]])
    col, _ = run_jobs(m, Wz, b, -1.0)
    y = (b + col) * dinv
    _, row = run_jobs(m, Wz, y, -1.0)
    return y + row


def test_factor_and_solve_tables(model_and_state):
    m, M, d = model_and_state
    qM, qLD = d.get("qM"), d.get("qLD")
    for plain in (True, False):        # the alias-free schedule (Newton instances) and the one with alias copies of the hot rows
        L, dinv = kernel_factor(m, qM, plain)
        np.testing.assert_allclose(L, qLD, rtol=1e-9, atol=1e-16)
        np.testing.assert_allclose(dinv, d.get("qLDiagInv"), rtol=1e-9)
    U = kernel_invert(m, L)
    Ld = np.eye(M.nv)
    ij = m["k_M_ij"]
    for e, v in enumerate(ij):
        if (v & 0xFFFF) != (v >> 16):
            Ld[v & 0xFFFF, v >> 16] = L[e]
    Wd = np.eye(M.nv) - np.linalg.inv(Ld)
    for e, v in enumerate(ij):
        if (v & 0xFFFF) != (v >> 16):
            assert abs(U[e] - Wd[v & 0xFFFF, v >> 16]) < 1e-9 * (1 + abs(U[e]))
    Md = dense_from_sparse(m, qM)
    b = np.random.default_rng(2).normal(size=M.nv)
    x = kernel_solve(m, U, dinv, b)
    np.testing.assert_allclose(Md @ x, b, rtol=1e-7, atol=1e-9)


def test_contact_chain_is_a_dfs_interval(model_and_state):
    """Wave::update_constraint applies a contact's force to dof d iff d <= leaf <= last_desc(d) (leaf = first byte of the
    packed chain): that set must be exactly the ancestor chain of the contact's body, which jac_mul walks byte by byte."""
    m, M, d = model_and_state
    nv, ncon = M.nv, M.ncon
    last = m["k_dof_i"][:, 10]
    packed = m["k_con_chain_packed"].view(np.uint32)
    anc_adr, anc = m["dof_ancadr"], m["dof_anc"]
    for c in range(ncon):
        nanc = int(m["k_con_i"][c, 4])
        chain = [(int(packed[p >> 2, c]) >> (8 * (p & 3))) & 255 for p in range(nanc)]
        leaf = int(m["k_con_i"][c, 3])
        if leaf < 0:
            assert nanc == 0
            continue
        assert chain[0] == leaf and chain == [int(x) for x in anc[anc_adr[leaf]:anc_adr[leaf + 1]][::-1]]
        assert sorted(chain) == [dd for dd in range(nv) if dd <= leaf <= last[dd]]
        rows = m["k_con_chain_rows"].view(np.uint32).reshape(-1, 9)            # contact-major copy read by the J*x jobs
        assert [(int(rows[c, p >> 2]) >> (8 * (p & 3))) & 255 for p in range(nanc)] == chain
    assert m["k_con_chain_rows"].size == 9 * (ncon + 1)
