"""Row schedules of the quad executor that factorises and inverts the sparse mass matrix (csrc/rr_kernel.h Wave::run_levels).

The executor runs table ROWS: 64 independent quad operations  d[0..3] -= src_a * src_b[0..3] [/ piv]  on the sparse-matrix array in
LDS, all sources read before any target of the row is written.  The LDS instructions of one wavefront execute in program order, so a
row sees everything earlier rows wrote and no row needs a wait of its own: WHICH operations share a row is free as long as

  * an operation comes in a later row than every operation that writes something it reads          (read after write),
  * no earlier than (same row allowed) every operation that reads something it overwrites            (write after read),
  * and no two operations of a row write the same cell (plain read-modify-writes, no atomics).

Rounds 1-2 packed the operations level by level (all dofs of one tree depth, a hand-off after each level): 97 + 63 rows for the rodent,
57 % of the slots filled, because the 35 levels of the trunk are nearly empty at the shallow end while the limbs' levels collide on the
root's rows.  Here the operations are list-scheduled as ONE dependency graph per schedule: 57 + 44 rows.

The hottest targets -- the 8 quads of the free joint's rows receive one contribution from EVERY other dof -- would serialise the schedule
(one write per cell and row), so the side branches of a fork accumulate into private ALIAS copies of their ancestors' rows (cells that are
dead during the factorisation; zeroed by the kernel) and one merge operation per quad folds a copy into the row before the row's own dof is
eliminated: alias -= a * b / piv gives -S, the merge  main -= (-1) * alias / 1  subtracts S.

Element indices of the tables: matrix entries 0 .. nM-1, then the constant cells ZERO (0.0), ONE (1.0), TRASH, MINUS_ONE (-1.0), then the
alias cells nM + 4 .. (the host maps those to their own LDS region at upload).
"""
from __future__ import annotations

import numpy as np

LANES = 64


class Op:
    __slots__ = ("reads", "writes", "word")

    def __init__(self, reads, writes, word):
        self.reads, self.writes, self.word = frozenset(reads), tuple(writes), word


def _quads(n):
    """runs of n consecutive (source, target) pairs cut into quads: (first offset, count)"""
    return [(j0, min(4, n - j0)) for j0 in range(0, n, 4)]


class Tree:
    def __init__(self, ddepth, Madr, dpar, last_desc, nM):
        self.nv = len(ddepth)
        self.depth = [int(x) for x in ddepth]
        self.Madr = [int(x) for x in Madr]
        self.par = [int(x) for x in dpar]
        self.last = [int(x) for x in last_desc]
        self.nM = int(nM)
        self.children = [[] for _ in range(self.nv)]
        for d in range(self.nv):
            if self.par[d] >= 0:
                self.children[self.par[d]].append(d)
        self.ZERO, self.ONE, self.TRASH, self.MINUS_ONE, self.ALIAS0 = self.nM, self.nM + 1, self.nM + 2, self.nM + 3, self.nM + 4

    def chain(self, k):
        """k, parent, ..., root"""
        out = [k]
        while self.par[out[-1]] >= 0:
            out.append(self.par[out[-1]])
        return out

    def subtree(self, r):
        return range(r, self.last[r] + 1)


def alias_candidates(tr: Tree, min_size: int = 3):
    """(contributing subtree root c, rows = the ancestors of c) for every child of a fork except the largest, shallow forks first"""
    out = []
    forks = sorted((d for d in range(tr.nv) if len(tr.children[d]) > 1), key=lambda d: tr.depth[d])
    for f in forks:
        kids = sorted(tr.children[f], key=lambda c: -(tr.last[c] - c + 1))
        for c in kids[1:]:
            if tr.last[c] - c + 1 >= min_size:
                out.append((c, tuple(reversed(tr.chain(f)))))
    return out


def factor_ops(tr: Tree, groups=()):
    """Operations of the L'DL factorisation [MuJoCo mj_factorM] in a valid sequential order (deep dofs first).  Dof k at depth l
    updates the entries (anc_p, anc_q), q = p..l, of each ancestor row p = 1..l:  M[anc_p][anc_q] -= M[k][anc_p] * M[k][anc_q] / M[k][k]
    -- consecutive entries of row anc_p, one shared operand.  `groups`: alias copies (see the module docstring)."""
    cell, base = {}, tr.ALIAS0
    gsets = []
    for gi, (c, rows) in enumerate(groups):
        gsets.append((set(tr.subtree(c)), set(rows)))
        for a in rows:
            for p in range(tr.depth[a] + 1):
                cell[(gi, tr.Madr[a] + p)] = base
                base += 1
    nalias = base - tr.ALIAS0
    ops, merged = [], set()

    def merges_for(a):
        for gi, (_, rows) in enumerate(gsets):
            if a in rows and (gi, a) not in merged:
                merged.add((gi, a))
                for j0, m_ in _quads(tr.depth[a] + 1):
                    src = cell[(gi, tr.Madr[a] + j0)]
                    dst = tuple(tr.Madr[a] + j0 + j if j < m_ else tr.TRASH for j in range(4))
                    ops.append(Op({src + j for j in range(m_)}, dst[:m_], (tr.MINUS_ONE, src, dst, 3)))

    by_depth = sorted(range(tr.nv), key=lambda k: (-tr.depth[k], k))
    for kk in by_depth:
        l = tr.depth[kk]
        if l == 0:
            continue
        merges_for(kk)
        ch = tr.chain(kk)
        for p in range(1, l + 1):
            a = ch[p]
            g = next((gi for gi, (sub, rows) in enumerate(gsets) if kk in sub and a in rows), None)
            src = tr.Madr[kk] + p
            for j0, m_ in _quads(l - p + 1):
                tgt = [tr.Madr[a] + j0 + j for j in range(m_)]
                if g is not None:
                    tgt = [cell[(g, e)] for e in tgt]
                d = tuple(tgt[j] if j < m_ else tr.TRASH for j in range(4))
                ops.append(Op({src, tr.Madr[kk]} | {src + j0 + j for j in range(m_)}, tgt, (src, src + j0, d, p + 1)))
    for a in range(tr.nv):
        merges_for(a)
    return ops, nalias


def inverse_ops(tr: Tree):
    """Operations of W = I - L^-1 in place (Gauss-Jordan, shallow dofs first): for dof k at depth l and every descendant row i,
    W[i][a] -= W[i][k] * W[k][a] over the strict ancestors a of k (consecutive entries of rows i and k); W[i][k] still holds L[i][k]."""
    ops = []
    for kk in sorted(range(tr.nv), key=lambda k: (tr.depth[k], k)):
        l = tr.depth[kk]
        if l == 0:
            continue
        for i in range(kk + 1, tr.last[kk] + 1):
            mk = tr.Madr[i] + tr.depth[i] - l
            for j0, m_ in _quads(l):
                tgt = [mk + 1 + j0 + j for j in range(m_)]
                d = tuple(tgt[j] if j < m_ else tr.TRASH for j in range(4))
                b0 = tr.Madr[kk] + 1 + j0
                ops.append(Op({mk} | {b0 + j for j in range(m_)}, tgt, (mk, b0, d, 0)))
    return ops


class Graph:
    """Hazards of the operations taken in their (valid) sequential order.  The read-modify-writes of one cell commute (sums), so
    writers of a cell are ordered only against its readers; among themselves they must merely sit in different rows."""

    def __init__(self, ops):
        n = self.n = len(ops)
        self.ops = ops
        writers, readers = {}, {}
        self.raw = [set() for _ in range(n)]     # strictly earlier rows
        self.war = [set() for _ in range(n)]     # same or earlier row
        for t, op in enumerate(ops):
            for e in op.reads:
                self.raw[t].update(writers.get(e, ()))
                readers.setdefault(e, []).append(t)
            for e in op.writes:
                self.war[t].update(r_ for r_ in readers.get(e, ()) if r_ != t)
                writers.setdefault(e, []).append(t)
        for t in range(n):
            self.war[t] -= self.raw[t]
        self.raw_succ = [[] for _ in range(n)]
        self.war_succ = [[] for _ in range(n)]
        for t in range(n):
            for d in self.raw[t]:
                self.raw_succ[d].append(t)
            for d in self.war[t]:
                self.war_succ[d].append(t)

    def reversed(self):
        g = object.__new__(Graph)
        g.n, g.ops = self.n, self.ops
        g.raw = [set(s) for s in self.raw_succ]
        g.war = [set(s) for s in self.war_succ]
        g.raw_succ = [list(s) for s in self.raw]
        g.war_succ = [list(s) for s in self.war]
        return g

    def tail_length(self):
        """rows that must follow each operation (longest chain of strict dependencies below it)"""
        order = self.topological()
        tl = [0] * self.n
        for t in reversed(order):
            for s in self.raw_succ[t]:
                tl[t] = max(tl[t], tl[s] + 1)
            for s in self.war_succ[t]:
                tl[t] = max(tl[t], tl[s])
        return tl

    def topological(self):
        indeg = [len(self.raw[t]) + len(self.war[t]) for t in range(self.n)]
        stack = [t for t in range(self.n) if indeg[t] == 0]
        out = []
        while stack:
            t = stack.pop()
            out.append(t)
            for s in self.raw_succ[t] + self.war_succ[t]:
                indeg[s] -= 1
                if indeg[s] == 0:
                    stack.append(s)
        assert len(out) == self.n, "cyclic hazards"
        return out


def list_schedule(g: Graph, key, width: int = LANES):
    """Greedy rows: the most urgent ready operations first (key(t, remaining writers of t's target): smaller = more urgent)."""
    n = g.n
    nraw = [len(p) for p in g.raw]
    remaining = {}
    for op in g.ops:
        remaining[op.writes] = remaining.get(op.writes, 0) + 1
    ready = [t for t in range(n) if nraw[t] == 0]
    row_of = [-1] * n
    rows, done = [], 0
    while done < n:
        ready.sort(key=lambda t: (key(t, remaining[g.ops[t].writes]), t))
        row, used, cand, progress = [], set(), ready, True
        while progress and len(row) < width:
            progress, rest = False, []
            for t in cand:
                if len(row) >= width or any(row_of[d] < 0 for d in g.war[t]) or any(e in used for e in g.ops[t].writes):
                    rest.append(t)
                    continue
                row_of[t] = len(rows)
                row.append(t)
                used.update(g.ops[t].writes)
                remaining[g.ops[t].writes] -= 1
                progress = True
            cand = rest
        assert row, "list_schedule: no operation could be placed"
        ready = cand
        for t in row:
            for s in g.raw_succ[t]:
                nraw[s] -= 1
                if nraw[s] == 0:
                    ready.append(s)
        rows.append(row)
        done += len(row)
    return rows, row_of


def schedule(ops, width: int = LANES, weights=(0.75, 1.0, 2.0), rounds: int = 2):
    """Best of: forward list schedules whose urgency is (rows that must follow) + w * (writes still due on the same cell -- a cell
    takes one write per row), each improved by backward / forward passes that use the previous pass' rows as deadlines."""
    g = Graph(ops)
    gr = g.reversed()
    tail, head = g.tail_length(), gr.tail_length()
    best = None
    for w in weights:
        rows, row_of = list_schedule(g, lambda t, rem: -(tail[t] + w * (rem - 1)), width)
        cands = [rows]
        for _ in range(rounds):
            rrows, rrow_of = list_schedule(gr, lambda t, rem: (-row_of[t], -head[t]), width)
            cands.append(rrows[::-1])
            R = len(rrows)
            rows, row_of = list_schedule(g, lambda t, rem: (R - 1 - rrow_of[t], -tail[t]), width)
            cands.append(rows)
        for c in cands:
            if best is None or len(c) < len(best):
                best = c
    check(ops, best, width)
    return best


def check(ops, rows, width: int = LANES):
    g = Graph(ops)
    row_of = {}
    for r, row in enumerate(rows):
        assert len(row) <= width
        seen = set()
        for t in row:
            row_of[t] = r
            for e in ops[t].writes:
                assert e not in seen, "two writes of one cell in a row"
                seen.add(e)
    assert len(row_of) == len(ops)
    for t in range(len(ops)):
        assert all(row_of[d] < row_of[t] for d in g.raw[t]) and all(row_of[d] <= row_of[t] for d in g.war[t])


def _split_wide(rows, width):
    """EXPERIMENT (RR_SCHED_WIDTH=128): a row of up to 128 operations becomes two consecutive table rows the executor treats as one"""
    if width <= LANES:
        return rows
    out = []
    for row in rows:
        out.append(row[:LANES]); out.append(row[LANES:])
    return out


class Banks:
    """8-byte LDS slot of an element index, up to a common constant: matrix cells from `matrix_slot`, alias cells from `alias_slot`
    (the host's layout, csrc/rr_kernel.h rr_layout; only their difference mod 32 matters, and only for timing)."""

    def __init__(self, tr: Tree, matrix_slot: int = 0, alias_slot: int = 0):
        self.a0, self.m0, self.a_s = tr.ALIAS0, matrix_slot, alias_slot

    def slot(self, e):
        return self.m0 + e if e < self.a0 else self.a_s + e - self.a0


def _half_read_cycles(words, bk: Banks, div: bool):
    """LDS-array cycles of a row's reads for one 32-lane half [MI355X guide: ds_read_b64 is served per half, bank = 8-byte slot mod 32,
    equal addresses broadcast, every further address on a busy bank costs a cycle]"""
    streams = [[bk.slot(w[0]) for w in words]]
    if div:
        streams.append([bk.slot(w[0] + 1 - w[3]) for w in words])
    for j in range(4):
        streams.append([bk.slot(w[1] + j) for w in words])
        streams.append([bk.slot(w[2][j]) for w in words])
    c = 0
    for st in streams:
        cnt = {}
        for x in set(st):
            cnt[x & 31] = cnt.get(x & 31, 0) + 1
        c += max(cnt.values()) if cnt else 0
    return c


def assign_lanes(ops, row, bk: Banks, div: bool, sweeps: int = 2):
    """Which lane runs which operation of a row -- free for correctness, not for the LDS: `ds_write_b64` is served in groups of 16
    consecutive lanes with bank = 8-byte slot mod 16, so a quarter of the wave writes conflict-free when the first targets of its
    16 quads lie in 16 different slot classes (the other three targets follow).  The operations of each class are spread over the four
    quarters; operations of one class then swap quarters while that lowers the modelled read cycles of the two halves.
    -> 64 entries: operation index or None."""
    cls = {}
    for t in row:
        cls.setdefault(bk.slot(ops[t].word[2][0]) & 15, []).append(t)
    Q = [[] for _ in range(4)]
    over = []
    for c, ts in sorted(cls.items(), key=lambda kv: (-len(kv[1]), kv[0])):
        qs = sorted(range(4), key=lambda q: (len(Q[q]), q))
        for i, t in enumerate(ts):
            if i < 4 and len(Q[qs[i]]) < 16:
                Q[qs[i]].append(t)
            else:
                over.append(t)
    for t in over:
        Q[min(range(4), key=lambda q: (len(Q[q]), q))].append(t)
    assert all(len(q) <= 16 for q in Q)

    def cost():
        return (_half_read_cycles([ops[t].word for t in Q[0] + Q[1]], bk, div) +
                _half_read_cycles([ops[t].word for t in Q[2] + Q[3]], bk, div))
    best = cost()
    for _ in range(sweeps):
        for c in sorted(cls):
            pos = [(q, i) for q in range(4) for i, t in enumerate(Q[q]) if bk.slot(ops[t].word[2][0]) & 15 == c]
            for x in range(len(pos)):
                for y in range(x + 1, len(pos)):
                    (qa, ia), (qb, ib) = pos[x], pos[y]
                    if qa >> 1 == qb >> 1:
                        continue                      # same half: the reads do not change
                    Q[qa][ia], Q[qb][ib] = Q[qb][ib], Q[qa][ia]
                    v = cost()
                    if v < best:
                        best = v
                    else:
                        Q[qa][ia], Q[qb][ib] = Q[qb][ib], Q[qa][ia]
    lanes = []
    for q in Q:
        lanes += q + [None] * (16 - len(q))
    return lanes


def pack(tr: Tree, ops, rows, ring: int, bk: Banks = None, div: bool = True):
    """rows of 64 x 4 ints:  x = a | b0 << 16,  y = d0 | d1 << 16,  z = d2 | d3 << 16,  w = q  (piv = a + 1 - q);  an empty slot is
    a = b0 = ZERO, piv = ONE, targets TRASH.  `ring` empty rows follow (the executor prefetches that far ahead)."""
    Z, T = tr.ZERO, tr.TRASH
    empty = (Z | (Z << 16), T | (T << 16), T | (T << 16), 0)
    out = np.empty((len(rows) + ring, LANES, 4), np.int64)
    out[:] = empty
    for r, row in enumerate(rows):
        lanes = assign_lanes(ops, row, bk, div) if bk is not None else row
        for ln, t in enumerate(lanes):
            if t is None:
                continue
            a, b0, d, q = ops[t].word
            assert 0 <= q < 256 and max(a, b0 + 3, *d) < 65536
            out[r, ln] = (a | (b0 << 16), d[0] | (d[1] << 16), d[2] | (d[3] << 16), q)
    return (out & 0xFFFFFFFF).astype(np.uint32).view(np.int32), np.int32(len(rows))


def build(ddepth, Madr, dpar, last_desc, nM, alias_cells: int, ring: int, matrix_slot: int = 0, alias_slot: int = 0):
    """-> dict of tables: k_factor3 (with alias copies, k_nalias cells), k_factor3p (plain: Newton's Hessian reuses the schedule on a
    second array, which has no alias region), k_linv, and their row counts."""
    tr = Tree(ddepth, Madr, dpar, last_desc, nM)
    bk = Banks(tr, matrix_slot, alias_slot)
    cand = alias_candidates(tr)
    plain_ops, _ = factor_ops(tr)
    import os
    W = int(os.environ.get("RR_SCHED_WIDTH", LANES))
    plain = schedule(plain_ops, W)
    best = (len(plain), 0, plain_ops, plain)
    tried = []
    shallow = [c for c in cand if cand and len(c[1]) == len(cand[0][1])]
    for groups in (shallow, cand):
        if not groups or groups in tried:
            continue
        tried.append(groups)
        ops, nalias = factor_ops(tr, groups)
        if nalias + 4 > alias_cells:
            continue
        rows = schedule(ops, W)
        if (len(rows), nalias) < best[:2]:
            best = (len(rows), nalias, ops, rows)
    k = {}
    k["k_factor3"], k["k_factor3_rows"] = pack(tr, best[2], _split_wide(best[3], W), ring, bk, True)
    k["k_nalias"] = np.int32(best[1])
    k["k_factor3p"], k["k_factor3p_rows"] = pack(tr, plain_ops, _split_wide(plain, W), ring, bk, True)
    inv = inverse_ops(tr)
    k["k_linv"], k["k_linv_rows"] = pack(tr, inv, _split_wide(schedule(inv, W, weights=(1.0, 2.0)), W), ring, bk, False)
    return k
