"""Host logic: the list-scheduled row programs of the factorisation / inversion (rodent_amd/levelsched.py) on RANDOM kinematic trees.
A numpy restatement of the executor (tests/test_ktables.py run_rows: every row reads before it writes, rows in order, nothing else) runs
the packed tables on a random SPD matrix of the tree's sparsity and must reproduce the dense L'DL factor and its inverse; the schedule
itself is checked against its three rules, and the lane assignment must be a pure permutation of each row."""
import numpy as np
import pytest

from rodent_amd import levelsched as ls
from tests.test_ktables import run_rows, _cells, RING


def random_tree(rng, nv, branch=0.25, roots=1):
    """dofs in DFS preorder (MuJoCo order): parent ids, depths, row addresses, last descendants"""
    par = [-1] * nv
    for d in range(roots, nv):
        # attach below the previous dof (a chain) or, with probability `branch`, below one of its ancestors (a fork): keeps DFS order
        p = d - 1
        while rng.random() < branch and par[p] >= 0:
            p = par[p]
        par[d] = p
    for r in range(1, roots):
        par[r] = -1
    depth = [0] * nv
    for d in range(nv):
        depth[d] = 0 if par[d] < 0 else depth[par[d]] + 1
    Madr = np.concatenate([[0], np.cumsum([x + 1 for x in depth])]).astype(int)
    last = list(range(nv))
    for d in range(nv - 1, -1, -1):
        if par[d] >= 0:
            last[par[d]] = max(last[par[d]], last[d])
    return par, depth, Madr[:-1], last, int(Madr[-1])


def spd_on_tree(rng, par, depth, Madr, nM):
    """dense SPD matrix whose non-zeros are the (dof, ancestor) pairs + the packed sparse array the kernel holds (row i: self, parent, ..., root)"""
    nv = len(par)
    M = np.zeros((nv, nv))
    # M = sum over dofs of v v^T with v supported on the dof's ancestor chain: tree sparsity, positive definite with the diagonal added
    for i in range(nv):
        chain = [i]
        while par[chain[-1]] >= 0:
            chain.append(par[chain[-1]])
        v = rng.normal(size=len(chain))
        M[np.ix_(chain, chain)] += np.outer(v, v)
    M += np.diag(rng.uniform(0.5, 1.5, nv))
    q = np.zeros(nM)
    for i in range(nv):
        a, p = i, 0
        while a >= 0:
            q[Madr[i] + p] = M[i, a]
            a, p = par[a], p + 1
    return M, q


def dense_ldl(M, par):
    """L'DL with L unit LOWER triangular in MuJoCo's convention: M = L^T D L, eliminating the deepest dofs first"""
    nv = M.shape[0]
    A = M.copy()
    L = np.eye(nv)
    D = np.zeros(nv)
    for k in range(nv - 1, -1, -1):
        D[k] = A[k, k]
        L[k, :k] = A[k, :k] / D[k]
        A[:k, :k] -= np.outer(L[k, :k], L[k, :k]) * D[k]
    return L, D


@pytest.mark.parametrize("seed,nv,branch,roots", [(0, 12, 0.3, 1), (1, 40, 0.2, 1), (2, 73, 0.1, 1), (3, 48, 0.4, 2), (4, 25, 0.0, 1), (5, 45, 0.25, 3)])
def test_row_programs_on_random_trees(seed, nv, branch, roots):
    rng = np.random.default_rng(seed)
    par, depth, Madr, last, nM = random_tree(rng, nv, branch, roots)
    k = ls.build(depth, Madr, par, last, nM, alias_cells=200, ring=RING, matrix_slot=int(rng.integers(0, 64)), alias_slot=int(rng.integers(0, 64)))
    M, q = spd_on_tree(rng, par, depth, Madr, nM)
    Ld, Dd = dense_ldl(M, par)
    for name, nalias in (("k_factor3", int(k["k_nalias"])), ("k_factor3p", 0)):
        arr, used = run_rows(np.concatenate([q, _cells(nM, nalias)]), nM, k[name], int(k[name + "_rows"]), True)
        arr = arr[:nM].copy()
        dinv = 1.0 / arr[Madr]
        for i in range(nv):                                      # the kernel's row scaling by 1/D
            arr[Madr[i] + 1:Madr[i] + depth[i] + 1] *= dinv[i]
        for i in range(nv):
            a, p = i, 0
            while a >= 0:
                want = Dd[i] if p == 0 else Ld[i, a]
                assert abs(arr[Madr[i] + p] - want) < 1e-9 * (1 + abs(want)), (name, i, p)
                a, p = par[a], p + 1
    # inversion on the factor: W = I - L^-1 (strictly lower part)
    W, _ = run_rows(np.concatenate([arr, _cells(nM, 0)]), nM, k["k_linv"], int(k["k_linv_rows"]), False)
    Wd = np.eye(nv) - np.linalg.inv(Ld)
    for i in range(nv):
        a, p = par[i], 1
        while a >= 0:
            assert abs(W[Madr[i] + p] - Wd[i, a]) < 1e-8 * (1 + abs(Wd[i, a])), (i, p)
            a, p = par[a], p + 1
    assert int(k["k_factor3_rows"]) <= int(k["k_factor3p_rows"])      # alias copies are only kept when they shorten the program


def test_schedule_rules_and_bounds():
    """the three rules a row program must satisfy (levelsched.check re-derives the hazards), its row count against the packing bound and
    the longest dependency chain, and what a violated rule looks like"""
    rng = np.random.default_rng(7)
    par, depth, Madr, last, nM = random_tree(rng, 50, 0.25)
    tr = ls.Tree(depth, Madr, par, last, nM)
    ops, nalias = ls.factor_ops(tr, ls.alias_candidates(tr)[:2])
    rows = ls.schedule(ops)
    ls.check(ops, rows)
    assert -(-len(ops) // ls.LANES) <= len(rows)
    g = ls.Graph(ops)
    assert len(rows) >= max(g.tail_length()) + 1                 # no shorter than the longest chain of strict dependencies
    # moving an operation in front of something it reads breaks the read-after-write rule
    t = next(t for t in range(len(ops)) if g.raw[t])
    bad = [list(r) for r in rows]
    for r in bad:
        if t in r:
            r.remove(t)
    bad[0].append(t)
    if len(bad[0]) <= ls.LANES:
        with pytest.raises(AssertionError):
            ls.check(ops, bad)
    # two writers of one cell in one row break the one-write rule
    w = {}
    for t, op in enumerate(ops):
        w.setdefault(op.writes, []).append(t)
    pair = next(v for v in w.values() if len(v) >= 2)
    bad = [[x for x in r if x not in pair[:2]] for r in rows] + [pair[:2]]
    with pytest.raises(AssertionError):
        ls.check(ops, bad)


def test_lane_assignment_is_a_permutation_and_spreads_the_write_classes():
    rng = np.random.default_rng(11)
    par, depth, Madr, last, nM = random_tree(rng, 73, 0.15)
    tr = ls.Tree(depth, Madr, par, last, nM)
    ops = ls.inverse_ops(tr)
    rows = ls.schedule(ops, weights=(1.0,), rounds=1)
    bk = ls.Banks(tr, 5, 9)
    before = after = 0
    for row in rows:
        lanes = ls.assign_lanes(ops, row, bk, False)
        assert len(lanes) == ls.LANES and sorted(t for t in lanes if t is not None) == sorted(row)
        def write_groups(order):
            c = 0
            for q in range(4):
                cls = [bk.slot(ops[t].word[2][0]) & 15 for t in order[16 * q:16 * q + 16] if t is not None]
                c += max([cls.count(x) for x in set(cls)] or [1])
            return c
        a, b = write_groups(list(row) + [None] * (ls.LANES - len(row))), write_groups(lanes)
        before += a
        after += b
    assert after < 0.8 * before          # write-group cycles of the whole program (4 per row when conflict-free)
