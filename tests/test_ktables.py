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
    """numpy restatement of Wave::factor (pair table)."""
    nv = int(m["nv"])
    L = qM.copy()
    depth, Madr = m["k_dof_i"][:, 3], m["k_dof_i"][:, 4]
    tri, rowadr = m["k_tri"], m["k_M_rowadr"]
    for k in range(nv - 1, -1, -1):
        dk = depth[k]
        if dk == 0:
            continue
        Mkk = Madr[k]
        dkk = L[Mkk]
        npairs = dk * (dk + 1) // 2
        pq = tri[:npairs]
        p, q = pq & 255, pq >> 8
        tmp = L[Mkk + p] / dkk
        adr = rowadr[Mkk + p] + (q - p)
        assert len(set(adr.tolist())) == npairs          # no two lanes update one address
        L[adr] -= L[Mkk + q] * tmp
        L[Mkk + 1:Mkk + dk + 1] /= dkk
    return L, 1.0 / L[Madr]


def kernel_solve(m, L, dinv, x):
    """numpy restatement of Wave::ldl_solve (level-synchronous)."""
    nv = int(m["nv"])
    depth = m["k_dof_i"][:, 3]
    dmax = int(depth.max())
    W = m["k_solve_fwd"].shape[1]
    xr = np.zeros(W)
    xr[:nv] = x
    dep = np.full(W, -1)
    dep[:nv] = depth
    sx = np.zeros(W)
    bwd, adr = m["k_solve_bwd"], m["k_solve_bwd_adr"]
    for li in range(dmax):
        level = dmax - li
        sel = dep == level
        sx[sel] = xr[sel]
        for r in range(adr[li], adr[li + 1]):
            e = bwd[r]
            ok = e >= 0
            xr[ok] -= L[e[ok] >> 8] * sx[e[ok] & 255]
    xr[:nv] *= dinv
    fwd = m["k_solve_fwd"]
    for l in range(dmax):
        sel = dep == l
        sx[sel] = xr[sel]
        e = fwd[l]
        ok = e >= 0
        xr[ok] -= L[e[ok] >> 8] * sx[e[ok] & 255]
    return xr[:nv]


def test_factor_and_solve_tables(model_and_state):
    m, M, d = model_and_state
    qM, qLD = d.get("qM"), d.get("qLD")
    L, dinv = kernel_factor(m, qM)
    np.testing.assert_allclose(L, qLD, rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(dinv, d.get("qLDiagInv"), rtol=1e-9)
    Md = dense_from_sparse(m, qM)
    b = np.random.default_rng(2).normal(size=M.nv)
    x = kernel_solve(m, L, dinv, b)
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
