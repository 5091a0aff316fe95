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


BLK = 1         # rows per block of the level schedules (RR_BLK)
RING = 8        # rows in flight (RR_RING)


def run_levels(L, tab, nrows, div):
    """numpy restatement of Wave::run_levels: blocks of BLK rows of quad operations; all reads of a block (sources and the
    old target values) are taken before its writes; a level's sources are never written inside the level.  L carries the
    extra cells ZERO, ONE, TRASH (+ pad) behind its nM entries."""
    nM = len(L) - 4
    assert L[nM] == 0.0 and L[nM + 1] == 1.0
    acc = np.zeros((LANES, 4))
    written, read = set(), set()
    assert nrows % RING == 0 and tab.shape[0] == nrows + RING
    u32 = tab.view(np.uint32).astype(np.int64)
    for b0 in range(0, nrows, BLK):
        snap = L.copy()                                   # what the block's batched reads see
        for u in range(BLK):
            e = u32[b0 + u]
            fl = int(e[0, 3] >> 8)
            assert np.all((e[:, 3] >> 8) == fl) and 0 <= fl < 4
            a, b_ = e[:, 0] & 0xFFFF, e[:, 0] >> 16
            d = np.stack([e[:, 1] & 0xFFFF, e[:, 1] >> 16, e[:, 2] & 0xFFFF, e[:, 2] >> 16], axis=1)
            q = e[:, 3] & 0xFF
            real = a != nM
            assert np.all(b_[~real] == nM) and np.all(q[~real] == 0)
            valid = d < nM                                # targets of short runs and of empty operations go to TRASH
            assert np.all(d[~valid] == nM + 2) and np.all(valid[real, 0])
            assert np.all(valid[:, :-1] >= valid[:, 1:])  # the valid targets of a quad come first
            read.update(a[real].tolist())
            for j in range(4):
                read.update((b_ + j)[valid[:, j] & real].tolist())
            if div:
                piv = a + 1 - q
                assert np.all(piv[~real] == nM + 1) and np.all(piv[real] < nM)
                read.update(piv[real].tolist())
                t = snap[a] / snap[piv]
            else:
                assert np.all(q == 0)
                t = snap[a]
            assert np.all(b_[real] + 3 < len(L))
            for j in range(4):
                acc[:, j] += snap[np.minimum(b_ + j, len(L) - 1)] * t
            if fl & 1 or not div:
                tg = d[valid]
                assert len(set(tg.tolist())) == len(tg)   # plain RMW: no two lanes / slots share a target
                assert not (set(tg.tolist()) & written)   # one write per target and level
                written.update(tg.tolist())
                L[tg] = snap[tg] - acc[valid]
                acc[:] = 0.0
                L[nM:] = [0.0, 1.0, 0.0, 0.0]             # whatever landed in TRASH is never used
            assert not (fl & 2) or u == BLK - 1           # levels end at block ends
            if fl & 2:
                assert not (written & read)               # reads of a level never see its writes
                assert np.all(acc == 0.0)
                written, read = set(), set()
    return L


def kernel_factor(m, qM):
    """numpy restatement of Wave::factor: gather rows by target entry (k_factor3), then the row scaling by 1/D."""
    L = run_levels(np.concatenate([qM, [0.0, 1.0, 0.0, 0.0]]), m["k_factor3"], int(m["k_factor3_rows"]), True)[:len(qM)]
    Madr = m["k_dof_i"][:, 4]
    dinv = 1.0 / L[Madr]
    ij = m["k_M_ij"]
    i, j = ij & 0xFFFF, ij >> 16
    off = i != j
    L[off] *= dinv[i[off]]
    return L, dinv


def kernel_invert(m, L):
    """numpy restatement of Wave::invert: W = I - L^-1 in place (k_linv)."""
    return run_levels(np.concatenate([L, [0.0, 1.0, 0.0, 0.0]]), m["k_linv"], int(m["k_linv_rows"]), False)[:len(L)]


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
    L, dinv = kernel_factor(m, qM)
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
