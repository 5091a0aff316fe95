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


def test_mulm_table(model_and_state):
    m, M, d = model_and_state
    nv = M.nv
    qM = d.get("qM")
    Md = dense_from_sparse(m, qM)
    x = np.random.default_rng(1).normal(size=nv)
    tab = m["k_mulm"]
    y = np.zeros(tab.shape[1])
    for t in range(tab.shape[0]):
        e = tab[t]
        ok = e >= 0
        y[ok] += qM[e[ok] >> 8] * x[e[ok] & 255]
    np.testing.assert_allclose(y[:nv], Md @ x, rtol=1e-12, atol=1e-14)


def kernel_factor(m, qM):
    """numpy restatement of Wave::factor: per level, gather rows by target entry (k_factor3), plain read-modify-write on
    the rows flagged 1, level hand-off on the rows flagged 2; then the row scaling by 1/D."""
    nv = int(m["nv"])
    L = qM.copy()
    tab, flag, R = m["k_factor3"], m["k_factor3_flag"], int(m["k_factor3_rows"])
    acc = np.zeros(LANES)
    dst = np.full(LANES, -1)
    written_in_level, read_in_level = set(), set()
    for r in range(R):
        e = tab[r]
        ok = e[:, 0] >= 0
        a, bq, piv = e[ok, 0] & 4095, e[ok, 0] >> 12, e[ok, 1] >> 12
        read_in_level.update(a.tolist()); read_in_level.update(bq.tolist()); read_in_level.update(piv.tolist())
        acc[ok] += L[a] / L[piv] * L[bq]
        has = e[:, 1] >= 0
        assert np.all(dst[has] == -1) or np.all(dst[has] == (e[has, 1] & 4095))   # one target per lane within a group
        dst[has] = e[has, 1] & 4095
        if flag[r] & 1:
            t = dst[dst >= 0]
            assert len(set(t.tolist())) == len(t)                                # plain RMW: no two lanes share a target
            assert not (set(t.tolist()) & written_in_level)                      # one RMW per target and level
            written_in_level.update(t.tolist())
            L[t] -= acc[dst >= 0]
            acc[:] = 0.0
            dst[:] = -1
        if flag[r] & 2:
            assert not (written_in_level & read_in_level)                        # reads of a level never see its writes
            written_in_level, read_in_level = set(), set()
    assert np.all(acc == 0.0) and np.all(tab[R:] == -1)
    Madr = m["k_dof_i"][:, 4]
    dinv = 1.0 / L[Madr]
    ij = m["k_M_ij"]
    i, j = ij & 0xFFFF, ij >> 16
    off = i != j
    L[off] *= dinv[i[off]]
    return L, dinv


def kernel_invert(m, L):
    """numpy restatement of Wave::invert: U = L^-1 in place, one k_linv row per depth level (read phase, then write)."""
    U = L.copy()
    rowadr = m["k_M_rowadr"]
    tab = m["k_linv"]
    dmax = int(m["k_dof_i"][:, 3].max())
    for l in range(dmax):
        e = tab[l][tab[l] >= 0]
        adr, p = e & 4095, e >> 12
        val = np.zeros(len(e))
        for n in range(len(e)):
            mi = adr[n] - p[n]
            sm = U[adr[n]]
            for q in range(1, p[n]):
                sm += U[mi + q] * U[rowadr[mi + q] + p[n] - q]
            val[n] = -sm
        U[adr] = val
    assert np.all(tab[dmax:] == -1)
    return U


def kernel_solve(m, U, dinv, b):
    """numpy restatement of Wave::ldl_solve with the explicit inverse: x = U D^-1 U' b (unit diagonal implied).
    U' b gathers over the DFS range of descendants (entry of (i, j) at base[i] - depth[j]); U y walks the ancestor chain."""
    nv = int(m["nv"])
    di = m["k_dof_i"]
    depth, Madr, last = di[:, 3], di[:, 4], di[:, 10]
    base, chain = m["k_dof_base"], m["k_dof_chain"]
    y = b.copy()
    for j in range(nv):
        for i in range(j + 1, last[j] + 1):
            y[j] += U[base[i] - depth[j]] * b[i]
    y *= dinv
    x = y.copy()
    for i in range(nv):
        for p in range(1, depth[i] + 1):
            a = (int(chain[(p - 1) >> 2, i]) >> (8 * ((p - 1) & 3))) & 255
            x[i] += U[Madr[i] + p] * y[a]
    return x


def test_factor_and_solve_tables(model_and_state):
    m, M, d = model_and_state
    qM, qLD = d.get("qM"), d.get("qLD")
    L, dinv = kernel_factor(m, qM)
    np.testing.assert_allclose(L, qLD, rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(dinv, d.get("qLDiagInv"), rtol=1e-9)
    U = kernel_invert(m, L)
    Md = dense_from_sparse(m, qM)
    b = np.random.default_rng(2).normal(size=M.nv)
    x = kernel_solve(m, U, dinv, b)
    np.testing.assert_allclose(Md @ x, b, rtol=1e-7, atol=1e-9)


def test_jtf_and_chain_tables(model_and_state):
    m, M, d = model_and_state
    nv, ncon = M.nv, M.ncon
    chain = m["k_con_chain"]
    jadr = m["con_jadr"]
    # a random "J" in the kernel's layout and random base forces: J^T f by gather lists == by chains
    rng = np.random.default_rng(3)
    J = rng.normal(size=jadr[-1])
    f = rng.normal(size=(ncon, 3))
    want = np.zeros(nv)
    for c in range(ncon):
        nanc = m["k_con_i"][c, 4]
        for p in range(nanc):
            dd = chain[p, c]
            assert dd >= 0
            want[dd] += J[jadr[c] + 3 * p:jadr[c] + 3 * p + 3] @ f[c]
        assert nanc == chain.shape[0] or chain[nanc, c] == -1
    tab = m["k_jtf"]
    got = np.zeros(tab.shape[1])
    for t in range(tab.shape[0]):
        e = tab[t]
        ok = np.nonzero(e >= 0)[0]
        c, a = e[ok] & 255, e[ok] >> 8
        got[ok] += J[a] * f[c, 0] + J[a + 1] * f[c, 1] + J[a + 2] * f[c, 2]
    np.testing.assert_allclose(got[:nv], want, rtol=1e-12, atol=1e-13)


def test_level_tables(model_and_state):
    m, M, d = model_and_state
    adr, order = m["k_lvl_adr"], m["k_lvl_body"]
    seen = set()
    for L in range(1, len(adr) - 1):
        bodies = order[adr[L]:adr[L + 1]]
        assert 0 < len(bodies) <= LANES
        for b in bodies:
            assert m["body_depth"][b] == L
            assert m["body_parentid"][b] == 0 or m["body_parentid"][b] in seen
        seen.update(int(b) for b in bodies)
    assert seen == set(range(1, M.nbody))


def kernel_factor2(m, qM):
    """numpy restatement of the level-parallel Wave::factor (k_factor2 rows, atomics = np.add.at)."""
    L = qM.copy()
    tab, first = m["k_factor2"], m["k_factor2_first"]
    for r in range(tab.shape[0]):
        e = tab[r]
        ok = e[:, 0] >= 0
        a, bq = e[ok, 0] & 4095, e[ok, 0] >> 12
        dst, piv = e[ok, 1] & 4095, e[ok, 1] >> 12
        np.add.at(L, dst, -(L[bq] * (L[a] / L[piv])))
    Madr = m["k_dof_i"][:, 4]
    dinv = 1.0 / L[Madr]
    for e_, ij in enumerate(m["k_M_ij"]):
        i, j = ij & 0xFFFF, ij >> 16
        if i != j:
            L[e_] *= dinv[i]
    return L, dinv


def kernel_solve2(m, L, dinv, x):
    """numpy restatement of the level-parallel Wave::ldl_solve (k_solve2 rows)."""
    nv = int(m["nv"])
    tab = m["k_solve2"]
    sx = x.copy()
    for l in range(tab.shape[0] - 1, -1, -1):
        e = tab[l][tab[l] >= 0]
        np.add.at(sx, e >> 20, -(L[e & 4095] * sx[(e >> 12) & 255]))
    sx *= dinv
    for l in range(tab.shape[0]):
        e = tab[l][tab[l] >= 0]
        np.add.at(sx, (e >> 12) & 255, -(L[e & 4095] * sx[e >> 20]))
    return sx


def test_level_parallel_factor_and_solve_tables(model_and_state):
    m, M, d = model_and_state
    qM, qLD = d.get("qM"), d.get("qLD")
    L, dinv = kernel_factor2(m, qM)
    np.testing.assert_allclose(L, qLD, rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(dinv, d.get("qLDiagInv"), rtol=1e-9)
    Md = dense_from_sparse(m, qM)
    b = np.random.default_rng(4).normal(size=M.nv)
    x = kernel_solve2(m, L, dinv, b)
    np.testing.assert_allclose(Md @ x, b, rtol=1e-7, atol=1e-9)
    # a level's rows never read an entry that the same level writes (reads: rows of that level; writes: ancestor rows)
    tab, first = m["k_factor2"], m["k_factor2_first"]
    r = 0
    while r < tab.shape[0]:
        r1 = r + 1
        while r1 < tab.shape[0] and not first[r1]:
            r1 += 1
        e = tab[r:r1].reshape(-1, 2)
        e = e[e[:, 0] >= 0]
        reads = set((e[:, 0] & 4095).tolist()) | set((e[:, 0] >> 12).tolist()) | set((e[:, 1] >> 12).tolist())
        writes = set((e[:, 1] & 4095).tolist())
        assert not (reads & writes)
        r = r1
