"""Set-up of the substructured direct mu solve for mid-size meshes (`tdgl_poisson_set_substructure`).

The reference factorises ``L_mu`` once with a sparse LU (operators.py:305-308) and back-substitutes
every step (solver.py:516).  A triangular solve is a long chain of dependent launches on a GPU, an
explicit inverse of the whole matrix (`tdgl_poisson_set_dense_inverse`) streams n^2 doubles per step and
stops paying at ~16k sites.  In between sits one level of nested dissection with every factor made
explicit, so that a solve is four launches of dense row products:

* the sites are cut into P compact parts (recursive coordinate bisection) and a vertex separator S (the
  sites of a part with a neighbour in a higher part); interiors first, part by part, then S -- this IS
  the library's internal site order (`substructure_order` is passed as the context's permutation);
* per part: ``G_p = A_pp^-1`` (dense, n_p ~ 450) and ``E_p = G_p A_pS`` (dense, n_p x s_p with s_p the
  ~100 separator sites the part touches); the Schur complement ``S_c = A_SS - sum_p A_Sp E_p`` (dense,
  singular like A: its null space is the constants);
* a solve:  ``y_p = G_p b_p``,  ``r_S = b_S - sum_p E_p^T b_p``  (one launch: every output is a sum of
  dense row segments),  ``x_S = pinv(S_c) r_S``  (the symmetric dense kernel pair of the small-mesh
  solve),  ``x_p = y_p - E_p x_S``  and the removal of the mean, which is known before the last launch:
  ``sum x = sum_p (G_p 1)^T b_p + u^T x_S`` with ``u = -sum_p E_p^T 1``.

Bytes per solve at 59k sites with 128 parts: 220 MB of G, 2 x 36 MB of E, 117 MB of the Schur pseudo-
inverse -- 0.4 GB, ~75 us, against nine PCG iterations of ~30 us.  The crossover with AMG-PCG is where the
separator's dense inverse stops fitting the time budget (~150k sites).

`solve_host` restates the device algorithm in NumPy (CPU tests; not used by the product path).
"""

from dataclasses import dataclass
from typing import List

import os

import numpy as np
import scipy.sparse as sp

from .partition import rcb_partition


def substructure_order(sites: np.ndarray, edges: np.ndarray, target_block: int = 448, rank_hint=None):
    """Permutation ``perm`` (internal -> reference site) and ``part_ptr`` ([P + 1]: interior ranges of the
    parts in internal numbering; the separator is ``[part_ptr[P], n)``).  ``rank_hint[i]`` orders the
    sites inside a part (e.g. their reverse Cuthill-McKee rank: keeps the stencil's gathers local)."""
    n = len(sites)
    nparts = max(2, int(round(n / float(target_block))))
    part = rcb_partition(np.asarray(sites, dtype=float), nparts)
    i, j = edges[:, 0], edges[:, 1]
    is_sep = np.zeros(n, dtype=bool)
    # a vertex cover of the cut edges: the endpoint in the lower part
    lo_i = part[i] < part[j]
    lo_j = part[j] < part[i]
    is_sep[i[lo_i]] = True
    is_sep[j[lo_j]] = True
    key = np.arange(n) if rank_hint is None else np.asarray(rank_hint)
    group = np.where(is_sep, nparts, part)
    perm = np.lexsort((key, group)).astype(np.int32)
    counts = np.bincount(group, minlength=nparts + 1)
    part_ptr = np.concatenate([[0], np.cumsum(counts[:nparts])]).astype(np.int32)
    return perm, part_ptr


@dataclass
class Substructure:
    n: int
    part_ptr: np.ndarray          # [P + 1]
    G: List[np.ndarray]           # per part [n_p, n_p]
    E: List[np.ndarray]           # per part [n_p, s_p]
    sep_idx: List[np.ndarray]     # per part: separator-local indices of the s_p separator sites it touches
    schur: np.ndarray             # [n_S, n_S], symmetric, singular (null space: constants); None: see `C`
    g: np.ndarray                 # [n_I]  G_p 1 per part, concatenated
    u: np.ndarray                 # [n_S]  -sum_p E_p^T 1

    @property
    def n_parts(self):
        return len(self.G)

    @property
    def n_interior(self):
        return int(self.part_ptr[-1])

    @property
    def n_sep(self):
        return self.n - self.n_interior

    def bytes_per_solve(self):
        sym = lambda m: 8 * ((m + 127) // 128) * (((m + 127) // 128) + 1) // 2 * 128 * 128
        return (sum(8 * g.size for g in self.G) + 2 * sum(8 * e.size for e in self.E)
                + (sym(self.n_sep) if self.schur is not None else 0))


def build_substructure(A: sp.spmatrix, part_ptr: np.ndarray, weights: np.ndarray = None, with_schur: bool = True) -> Substructure:
    """``A`` = the level-0 Poisson matrix in the internal order `substructure_order` produced.

    ``weights`` (two-level form, `build_substructure2`): the functional whose value on the solution the factors
    must deliver -- ``g = G_p w_p`` and ``u = w_S - sum_p E_p^T w_p`` so that ``w . x = g . b_I + u . x_S``
    (default: ``w = 1`` with the ``1_S . x_S`` term dropped, x_S being the zero-sum solution of the Schur system).
    ``with_schur=False``: the Schur complement is not formed densely (`Substructure.schur` is None); the blocks
    ``A_Sp E_p`` are kept in `Substructure.C` instead."""
    A = A.tocsr()
    n = A.shape[0]
    P = len(part_ptr) - 1
    nI = int(part_ptr[-1])
    nS = n - nI
    if nS < 2:
        raise ValueError("substructure: no separator (a single part?)")
    schur = A[nI:, nI:].toarray() if with_schur else None
    AIS = A[:nI, nI:].tocsr()
    AII = A[:nI, :nI].tocsr()
    g = np.empty(nI)
    u = np.zeros(nS) if weights is None else np.array(weights[nI:], dtype=float)
    w_int = np.ones(nI) if weights is None else np.asarray(weights[:nI], dtype=float)

    def one_part(p):
        a, b = int(part_ptr[p]), int(part_ptr[p + 1])
        App = AII[a:b, a:b].toarray()
        # (cross-part couplings between interiors do not exist: the separator covers every cut edge)
        Gp = np.linalg.inv(App)  # (positive definite: a part's interior is a proper piece of a connected graph)
        if not np.all(np.isfinite(Gp)) or np.any(np.diag(Gp) <= 0.0):
            raise np.linalg.LinAlgError(f"part {p}: interior block not positive definite")
        Gp = 0.5 * (Gp + Gp.T)
        ApS = AIS[a:b]
        cols = np.unique(ApS.indices)
        B = ApS[:, cols].toarray()
        Ep = Gp @ B
        return Gp, np.ascontiguousarray(Ep), cols.astype(np.int32), B.T @ Ep

    # the parts are independent: a thread each (LAPACK / BLAS release the GIL; one BLAS thread per call)
    from concurrent.futures import ThreadPoolExecutor

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        from contextlib import nullcontext as threadpool_limits
    workers = max(1, min(32, (os.cpu_count() or 1)))
    with threadpool_limits(limits=1):
        with ThreadPoolExecutor(workers) as pool:
            results = list(pool.map(one_part, range(P)))
    G, E, sidx, Cs = [], [], [], []
    for p, (Gp, Ep, cols, C) in enumerate(results):
        a, b = int(part_ptr[p]), int(part_ptr[p + 1])
        if with_schur:
            schur[np.ix_(cols, cols)] -= C
        else:
            Cs.append(C)
        G.append(Gp)
        E.append(Ep)
        sidx.append(cols)
        g[a:b] = Gp @ w_int[a:b]
        u[cols] -= Ep.T @ w_int[a:b]
    # interiors of different parts must not be coupled
    off = AII.tocoo()
    pr = np.searchsorted(part_ptr, off.row, side="right")
    pc = np.searchsorted(part_ptr, off.col, side="right")
    if np.any(pr != pc):
        raise ValueError("substructure: the separator does not cover every cut edge")
    if with_schur:
        schur = 0.5 * (schur + schur.T)
    out = Substructure(n=n, part_ptr=np.asarray(part_ptr, dtype=np.int32), G=G, E=E, sep_idx=sidx, schur=schur, g=g, u=u)
    out.C = Cs
    return out


def substructure_order2(sites: np.ndarray, edges: np.ndarray, target_block: int = 128, target_super: int = 4096, rank_hint=None):
    """Two levels of nested dissection.  The sites are cut into Q compact super-blocks (recursive coordinate
    bisection) with a top separator T covering every edge between two of them; inside a super-block, T removed,
    the sites are cut into parts of ~``target_block`` sites with a fine separator S'_Q covering the edges between
    two parts.  Internal order: part interiors (part by part, the parts of a super-block together), then S'_0,
    S'_1, ... , then T.  Returns ``perm`` (internal -> reference site), ``part_ptr`` ([P + 1] interior ranges)
    and ``super_ptr`` ([Q + 1]: the range of S'_Q is ``[super_ptr[Q], super_ptr[Q + 1])``, T is
    ``[super_ptr[Q], n)``).  Seen from the first level the separator is everything behind ``part_ptr[-1]``; seen
    from the second, the S'_Q are the "parts" of the first level's Schur complement (decoupled from each other:
    every path between two super-blocks passes T) and T is its separator."""
    sites = np.asarray(sites, dtype=float)
    n = len(sites)
    nsuper = max(2, int(round(n / float(target_super))))
    sup = rcb_partition(sites, nsuper)
    i, j = edges[:, 0], edges[:, 1]
    is_T = np.zeros(n, dtype=bool)
    is_T[i[sup[i] < sup[j]]] = True
    is_T[j[sup[j] < sup[i]]] = True
    part = np.full(n, -1, dtype=np.int64)
    nparts = 0
    for q in range(nsuper):
        idx = np.flatnonzero((sup == q) & ~is_T)
        if len(idx) == 0:
            continue
        k = max(1, int(round(len(idx) / float(target_block))))
        part[idx] = nparts + (rcb_partition(sites[idx], k) if k > 1 else 0)
        nparts += k
    free = ~is_T[i] & ~is_T[j]
    is_S = np.zeros(n, dtype=bool)
    is_S[i[free & (part[i] < part[j])]] = True
    is_S[j[free & (part[j] < part[i])]] = True
    key = np.arange(n) if rank_hint is None else np.asarray(rank_hint)
    group = np.where(is_T, nparts + nsuper, np.where(is_S, nparts + sup, part))
    perm = np.lexsort((key, group)).astype(np.int32)
    counts = np.bincount(group, minlength=nparts + nsuper + 1)
    part_ptr = np.concatenate([[0], np.cumsum(counts[:nparts])]).astype(np.int32)
    super_ptr = (part_ptr[-1] + np.concatenate([[0], np.cumsum(counts[nparts:nparts + nsuper])])).astype(np.int32)
    return perm, part_ptr, super_ptr


@dataclass
class Substructure2:
    """Two-level factors: `outer` eliminates the part interiors (no dense Schur complement), `inner` is the
    same construction applied to the outer level's Schur complement S1 on [S'_0 .. S'_{Q-1} | T]."""
    outer: Substructure
    inner: Substructure

    def bytes_per_solve(self):
        return self.outer.bytes_per_solve() + self.inner.bytes_per_solve()


def build_substructure2(A: sp.spmatrix, part_ptr: np.ndarray, super_ptr: np.ndarray) -> Substructure2:
    """``A`` in the internal order of `substructure_order2`."""
    A = A.tocsr()
    n = A.shape[0]
    nI = int(part_ptr[-1])
    nS = n - nI
    outer = build_substructure(A, part_ptr, with_schur=False)
    # S1 = A_SS - sum_p A_Sp E_p as a sparse matrix: the blocks of different super-blocks do not overlap
    # outside T, so it has ~ (|S'_Q| + |T_Q|)^2 entries per super-block
    rows, cols, vals = [], [], []
    for idx, C in zip(outer.sep_idx, outer.C):
        k = len(idx)
        rows.append(np.repeat(idx, k))
        cols.append(np.tile(idx, k))
        vals.append(-C.ravel())
    ASS = A[nI:, nI:].tocoo()
    rows.append(ASS.row)
    cols.append(ASS.col)
    vals.append(ASS.data)
    S1 = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nS, nS)).tocsr()
    S1 = (0.5 * (S1 + S1.T)).tocsr()
    outer.C = None
    # the functional sum x = sum_p g_p . b_p + v . x_S with v = 1_S - sum_p E_p^T 1 is carried down a level
    v = 1.0 + outer.u
    inner = build_substructure(S1, np.asarray(super_ptr, dtype=np.int64) - nI, weights=v)
    # the sparse coupling blocks (separator rows x interior columns) of both levels: r_S = b_S - A_SI y_I
    outer.coupling = A[nI:, :nI].tocsr()
    nI2 = int(super_ptr[-1]) - nI
    inner.coupling = S1[nI2:, :nI2].tocsr()
    return Substructure2(outer=outer, inner=inner)


def solve_host2(sub2: Substructure2, b: np.ndarray, spinv: np.ndarray = None, remove_mean: bool = True,
                sparse_sep: bool = False) -> np.ndarray:
    """The two-level device sequence in NumPy (six launches: down, down, dense pair, up, up; ``sparse_sep``: the
    separator right-hand sides through the sparse coupling blocks, two more launches, a fifth of the bytes less)."""
    o, q = sub2.outer, sub2.inner
    nI, P = o.n_interior, o.n_parts
    if remove_mean:
        b = b - b.mean()
    y = np.empty(nI)
    r = b[nI:].copy()
    total = 0.0
    for p in range(P):
        a, e = int(o.part_ptr[p]), int(o.part_ptr[p + 1])
        y[a:e] = o.G[p] @ b[a:e]
        if not sparse_sep:
            r[o.sep_idx[p]] -= o.E[p].T @ b[a:e]
        total += o.g[a:e] @ b[a:e]
    if sparse_sep:
        r -= o.coupling @ y
    # the Schur system S1 x_S = r on the second level; no gauge: v . x_S is what the first level needs
    nI2, Q = q.n_interior, q.n_parts
    if spinv is None:
        spinv = schur_pinv(q.schur)
    y2 = np.empty(nI2)
    rT = r[nI2:].copy()
    for k in range(Q):
        a, e = int(q.part_ptr[k]), int(q.part_ptr[k + 1])
        y2[a:e] = q.G[k] @ r[a:e]
        if not sparse_sep:
            rT[q.sep_idx[k]] -= q.E[k].T @ r[a:e]
        total += q.g[a:e] @ r[a:e]
    if sparse_sep:
        rT -= q.coupling @ y2
    xT = spinv @ rT
    total += q.u @ xT
    xs = np.empty(o.n_sep)
    for k in range(Q):
        a, e = int(q.part_ptr[k]), int(q.part_ptr[k + 1])
        xs[a:e] = y2[a:e] - q.E[k] @ xT[q.sep_idx[k]]
    xs[nI2:] = xT
    mean = total / o.n
    x = np.empty(o.n)
    for p in range(P):
        a, e = int(o.part_ptr[p]), int(o.part_ptr[p + 1])
        x[a:e] = y[a:e] - o.E[p] @ xs[o.sep_idx[p]] - mean
    x[nI:] = xs - mean
    return x


def substructure_order3(sites: np.ndarray, edges: np.ndarray, target_block: int = 160, target_super: int = 4096,
                        target_big: int = 32768, rank_hint=None):
    """Three levels: like `substructure_order2` with one more cut above it -- super-super-blocks of ``target_big`` sites
    with a top-top separator TT, super-blocks inside them with separators T'_R, parts inside those with fine separators
    S'_Q.  Order: part interiors | S'_0 .. | T'_0 .. | TT.  Returns ``perm`` and the three pointer arrays (part interiors;
    the S'_Q; the T'_R: each [count + 1], in internal numbering, each starting where the previous one ends)."""
    sites = np.asarray(sites, dtype=float)
    n = len(sites)
    i, j = edges[:, 0], edges[:, 1]

    def cover(label, active):  # the endpoint with the lower label of every edge between two labels among the active sites
        m = active[i] & active[j]
        out = np.zeros(n, dtype=bool)
        out[i[m & (label[i] < label[j])]] = True
        out[j[m & (label[j] < label[i])]] = True
        return out

    def cut_inside(label_of_groups, n_groups, excluded, target):
        lab = np.full(n, -1, dtype=np.int64)
        count = 0
        for g in range(n_groups):
            idx = np.flatnonzero((label_of_groups == g) & ~excluded)
            if len(idx) == 0:
                continue
            k = max(1, int(round(len(idx) / float(target))))
            lab[idx] = count + (rcb_partition(sites[idx], k) if k > 1 else 0)
            count += k
        return lab, count

    n_big = max(2, int(round(n / float(target_big))))
    big = rcb_partition(sites, n_big).astype(np.int64)
    is_TT = cover(big, np.ones(n, dtype=bool))
    sup, n_sup = cut_inside(big, n_big, is_TT, target_super)
    is_T = cover(sup, ~is_TT)
    part, n_part = cut_inside(sup, n_sup, is_TT | is_T, target_block)
    is_S = cover(part, ~is_TT & ~is_T)
    key = np.arange(n) if rank_hint is None else np.asarray(rank_hint)
    group = np.where(is_TT, n_part + n_sup + n_big, np.where(is_T, n_part + n_sup + big, np.where(is_S, n_part + sup, part)))
    perm = np.lexsort((key, group)).astype(np.int32)
    counts = np.bincount(group, minlength=n_part + n_sup + n_big + 1)
    ptr1 = np.concatenate([[0], np.cumsum(counts[:n_part])]).astype(np.int32)
    ptr2 = (ptr1[-1] + np.concatenate([[0], np.cumsum(counts[n_part:n_part + n_sup])])).astype(np.int32)
    ptr3 = (ptr2[-1] + np.concatenate([[0], np.cumsum(counts[n_part + n_sup:n_part + n_sup + n_big])])).astype(np.int32)
    return perm, ptr1, ptr2, ptr3


def _schur_sparse(level: Substructure, A_level: sp.spmatrix) -> sp.csr_matrix:
    """A level's Schur complement ``A_SS - sum_p A_Sp E_p`` as a sparse matrix (its blocks do not overlap outside the
    next level's separator); consumes ``level.C``."""
    nI = level.n_interior
    rows, cols, vals = [], [], []
    for idx, C in zip(level.sep_idx, level.C):
        k = len(idx)
        rows.append(np.repeat(idx, k))
        cols.append(np.tile(idx, k))
        vals.append(-C.ravel())
    ASS = A_level[nI:, nI:].tocoo()
    rows.append(ASS.row)
    cols.append(ASS.col)
    vals.append(ASS.data)
    nS = A_level.shape[0] - nI
    S = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nS, nS)).tocsr()
    level.C = None
    return (0.5 * (S + S.T)).tocsr()


def build_substructure_levels(A: sp.spmatrix, ptrs, gauge: bool = True) -> List[Substructure]:
    """Factors of a dissection with ``len(ptrs)`` levels (pointer arrays as `substructure_order3` returns them): level
    k is `build_substructure` applied to level k - 1's Schur complement, the gauge functional is handed down as weights,
    only the last level forms its Schur complement densely; every level carries its sparse ``coupling`` block.
    ``gauge=False``: ``A`` is positive definite (a rank's interior block, `schur_dd`): no functional is carried (all
    weights zero, so the device's mean comes out as zero) and the last level's complement is inverted as it is."""
    A_level = A.tocsr()
    levels, weights, offset = [], (None if gauge else np.zeros(A_level.shape[0])), 0
    for k, ptr in enumerate(ptrs):
        last = k == len(ptrs) - 1
        lv = build_substructure(A_level, np.asarray(ptr, dtype=np.int64) - offset, weights=weights, with_schur=last)
        nI = lv.n_interior
        lv.coupling = A_level[nI:, :nI].tocsr()
        levels.append(lv)
        if not last:
            weights = (1.0 + lv.u) if (k == 0 and gauge) else lv.u  # (first level: v = 1_S - sum E^T 1; further down: u itself is the functional)
            A_level = _schur_sparse(lv, A_level)
            offset += nI
    return levels


def solve_host_levels(levels: List[Substructure], b: np.ndarray, sparse_sep: bool = True) -> np.ndarray:
    """The multi-level device sequence in NumPy: ways down, the dense top separator, ways up, the mean removed."""
    n = levels[0].n
    vec = b - b.mean()
    ys, total = [], 0.0
    for lv in levels:
        nI = lv.n_interior
        y = np.empty(nI)
        r = vec[nI:].copy()
        for p in range(lv.n_parts):
            a, e = int(lv.part_ptr[p]), int(lv.part_ptr[p + 1])
            y[a:e] = lv.G[p] @ vec[a:e]
            if not sparse_sep:
                r[lv.sep_idx[p]] -= lv.E[p].T @ vec[a:e]
            total += lv.g[a:e] @ vec[a:e]
        if sparse_sep:
            r -= lv.coupling @ y
        ys.append(y)
        vec = r
    x = schur_pinv(levels[-1].schur) @ vec
    total += levels[-1].u @ x
    for lv, y in zip(reversed(levels), reversed(ys)):
        full = np.empty(lv.n)
        for p in range(lv.n_parts):
            a, e = int(lv.part_ptr[p]), int(lv.part_ptr[p + 1])
            full[a:e] = y[a:e] - lv.E[p] @ x[lv.sep_idx[p]]
        full[lv.n_interior:] = x
        x = full
    return x - total / n


def schur_pinv(schur: np.ndarray) -> np.ndarray:
    """pinv of the singular Schur complement (null space = constants), as `amg.dense_pseudo_inverse` does."""
    m = schur.shape[0]
    s = float(np.diag(schur).mean())
    inv = np.linalg.inv(schur + s / m)
    return 0.5 * (inv + inv.T) - 1.0 / (s * m)


def solve_host(sub: Substructure, b: np.ndarray, spinv: np.ndarray = None, remove_mean: bool = True) -> np.ndarray:
    """The device algorithm in NumPy: ``pinv(A) b``, the zero-mean solution of ``A x = b - mean(b)``.

    The four launches themselves assume a right-hand side orthogonal to the constants: the pseudo-inverse
    of the Schur complement projects the SEPARATOR residual only, so for ``sum(b) != 0`` the bare sequence
    answers ``A x = b - (sum(b) / n_S) e_S`` instead (an error ~25x ``mean(b)`` on a 1.5k-site mesh).  In the
    time loop ``sum(b) = 0`` holds to round-off (the divergence of an edge field sums to zero site-area
    weighted, and `validate_terminal_currents` makes the terminal currents add up to zero), and the
    library's one-off entry point `tdgl_poisson_solve` removes the mean first, as ``remove_mean`` does here;
    ``remove_mean=False`` reproduces the bare sequence (tests)."""
    nI, P = sub.n_interior, sub.n_parts
    if remove_mean:
        b = b - b.mean()
    if spinv is None:
        spinv = schur_pinv(sub.schur)
    y = np.empty(nI)
    r = b[nI:].copy()
    gd = np.empty(P)
    for p in range(P):
        a, e = int(sub.part_ptr[p]), int(sub.part_ptr[p + 1])
        y[a:e] = sub.G[p] @ b[a:e]
        r[sub.sep_idx[p]] -= sub.E[p].T @ b[a:e]
        gd[p] = sub.g[a:e] @ b[a:e]
    xs = spinv @ r
    mean = (gd.sum() + sub.u @ xs) / sub.n
    x = np.empty(sub.n)
    for p in range(P):
        a, e = int(sub.part_ptr[p]), int(sub.part_ptr[p + 1])
        x[a:e] = y[a:e] - sub.E[p] @ xs[sub.sep_idx[p]] - mean
    x[nI:] = xs - mean
    return x


def pack_for_device(sub: Substructure, sparse_sep: bool = False):
    """``sparse_sep``: the separator rows of the way down keep their identity segment only -- the library forms
    ``r_S = b_S - A_SI y_I`` with the sparse coupling block instead (`tdgl_poisson_set_substructure_coupling`), which
    needs ``y_I`` complete, i.e. a launch of its own, but not the -E^T rows (a fifth of a two-level solve's bytes).

    Flat arrays of `tdgl_substructure` (include/tdgl_hip.h): the way down as rows of dense segments
    over one value pool -- rows [0, n_I): ``G_p`` rows; rows [n_I, n): ``b_S`` itself (a segment of one
    entry with value 1) minus the ``E_p^T`` rows of the parts that touch the site; rows [n, n + P):
    ``(G_p 1)^T`` -- and the way up as the ``E_p`` blocks with their separator index lists."""
    nI, nS, P, n = sub.n_interior, sub.n_sep, sub.n_parts, sub.n
    pp = sub.part_ptr
    sizes = np.diff(pp).astype(np.int64)
    # value pool: [1.0 | G blocks | -E^T blocks | g]
    g_off = 1 + np.concatenate([[0], np.cumsum(sizes * sizes)])
    s_cnt = np.array([len(s) for s in sub.sep_idx], dtype=np.int64)
    et_off = g_off[-1] + np.concatenate([[0], np.cumsum(sizes * s_cnt)])
    gvec_off = et_off[-1]
    vals = np.empty(gvec_off + nI)
    vals[0] = 1.0
    for p in range(P):
        vals[g_off[p]:g_off[p + 1]] = sub.G[p].ravel()
        vals[et_off[p]:et_off[p + 1]] = (-sub.E[p].T).ravel()
    vals[gvec_off:] = sub.g
    # segments
    seg_val, seg_x, seg_len, seg_row = [], [], [], []
    # interior rows
    rows_I = np.arange(nI, dtype=np.int64)
    part_of = np.searchsorted(pp, rows_I, side="right") - 1
    seg_val.append(g_off[part_of] + (rows_I - pp[part_of]) * sizes[part_of])
    seg_x.append(pp[part_of].astype(np.int64))
    seg_len.append(sizes[part_of])
    seg_row.append(rows_I)
    # separator rows: identity + one segment per touching part
    seg_val.append(np.zeros(nS, dtype=np.int64))
    seg_x.append(nI + np.arange(nS, dtype=np.int64))
    seg_len.append(np.ones(nS, dtype=np.int64))
    seg_row.append(nI + np.arange(nS, dtype=np.int64))
    for p in range(0 if sparse_sep else P):
        k = len(sub.sep_idx[p])
        seg_val.append(et_off[p] + np.arange(k, dtype=np.int64) * sizes[p])
        seg_x.append(np.full(k, pp[p], dtype=np.int64))
        seg_len.append(np.full(k, sizes[p], dtype=np.int64))
        seg_row.append(nI + sub.sep_idx[p].astype(np.int64))
    # g rows
    seg_val.append(gvec_off + pp[:-1].astype(np.int64))
    seg_x.append(pp[:-1].astype(np.int64))
    seg_len.append(sizes)
    seg_row.append(n + np.arange(P, dtype=np.int64))
    seg_val, seg_x, seg_len, seg_row = (np.concatenate(a) for a in (seg_val, seg_x, seg_len, seg_row))
    order = np.argsort(seg_row, kind="stable")
    seg_val, seg_x, seg_len, seg_row = seg_val[order], seg_x[order], seg_len[order], seg_row[order]
    seg_ptr = np.concatenate([[0], np.cumsum(np.bincount(seg_row, minlength=n + P))]).astype(np.int32)
    e_off = np.concatenate([[0], np.cumsum(sizes * s_cnt)]).astype(np.int64)
    e_vals = np.concatenate([e.ravel() for e in sub.E]) if P else np.zeros(0)
    sep_ptr = np.concatenate([[0], np.cumsum(s_cnt)]).astype(np.int32)
    sep_idx = np.concatenate(sub.sep_idx).astype(np.int32)
    return dict(
        n_interior=nI, n_sep=nS, n_parts=P, part_ptr=pp.astype(np.int32),
        seg_ptr=seg_ptr, seg_val=seg_val.astype(np.int64), seg_x=seg_x.astype(np.int32), seg_len=seg_len.astype(np.int32),
        vals=np.ascontiguousarray(vals), sep_ptr=sep_ptr, sep_idx=sep_idx, e_off=e_off[:-1].copy(),
        e_vals=np.ascontiguousarray(e_vals), u=np.ascontiguousarray(sub.u),
        schur=None if sub.schur is None else np.ascontiguousarray(sub.schur),
    )


def down_host(pk, b):
    """`k_sub_down` restated on the packed arrays (tests: the packing itself)."""
    nrows = len(pk["seg_ptr"]) - 1
    out = np.zeros(nrows)
    for r in range(nrows):
        for k in range(pk["seg_ptr"][r], pk["seg_ptr"][r + 1]):
            v0, x0, ln = int(pk["seg_val"][k]), int(pk["seg_x"][k]), int(pk["seg_len"][k])
            out[r] += pk["vals"][v0:v0 + ln] @ b[x0:x0 + ln]
    return out


def plan_for_device(A: sp.spmatrix, part_ptr: np.ndarray):
    """Index arrays of `tdgl_substructure_plan` (include/tdgl_hip.h): everything `tdgl_poisson_build_substructure`
    needs to form the factors ON THE DEVICE -- which separator sites each part touches, the entries of
    ``A_pS`` grouped by (part, separator site), the (part, site) pairs grouped by site, ``A_SS`` as CSR, and
    the segment description of the way down with the offsets of the value pool the library fills
    (``[1.0 | G_p blocks | -E_p^T blocks | G_p 1]``, as `pack_for_device` lays it out).  No floating point
    work happens here."""
    A = A.tocsr()
    n = A.shape[0]
    pp = np.asarray(part_ptr, dtype=np.int64)
    P = len(pp) - 1
    nI = int(pp[-1])
    nS = n - nI
    if nS < 2 or P < 1:
        raise ValueError("substructure: no separator (a single part?)")
    AII = A[:nI, :nI].tocoo()
    pr = np.searchsorted(pp, AII.row, side="right")
    pc = np.searchsorted(pp, AII.col, side="right")
    if np.any(pr != pc):
        raise ValueError("substructure: the separator does not cover every cut edge")
    AIS = A[:nI, nI:].tocoo()
    part_of = np.searchsorted(pp, AIS.row, side="right") - 1
    key = part_of.astype(np.int64) * nS + AIS.col
    order = np.argsort(key, kind="stable")
    key, rows, vals = key[order], AIS.row[order], AIS.data[order]
    pairs, start, counts = np.unique(key, return_index=True, return_counts=True)
    pair_part = (pairs // nS).astype(np.int64)
    sep_idx = (pairs % nS).astype(np.int32)
    s_cnt = np.bincount(pair_part, minlength=P).astype(np.int64)
    sep_ptr = np.concatenate([[0], np.cumsum(s_cnt)]).astype(np.int32)
    ent_ptr = np.concatenate([start, [len(key)]]).astype(np.int32)
    node_order = np.argsort(sep_idx, kind="stable")  # pairs grouped by separator site, ascending pair id
    node_ptr = np.concatenate([[0], np.cumsum(np.bincount(sep_idx, minlength=nS))]).astype(np.int32)
    ASS = A[nI:, nI:].tocsr()
    ASS.sort_indices()
    sizes = np.diff(pp)
    g_off = 1 + np.concatenate([[0], np.cumsum(sizes * sizes)])
    et_off = g_off[-1] + np.concatenate([[0], np.cumsum(sizes * s_cnt)])
    gvec_off = int(et_off[-1])
    e_off = np.concatenate([[0], np.cumsum(sizes * s_cnt)])
    # segments of the way down (pack_for_device without the values)
    rows_I = np.arange(nI, dtype=np.int64)
    pof = np.searchsorted(pp, rows_I, side="right") - 1
    seg_val = [g_off[pof] + (rows_I - pp[pof]) * sizes[pof], np.zeros(nS, dtype=np.int64)]
    seg_x = [pp[pof], nI + np.arange(nS, dtype=np.int64)]
    seg_len = [sizes[pof], np.ones(nS, dtype=np.int64)]
    seg_row = [rows_I, nI + np.arange(nS, dtype=np.int64)]
    jloc = np.arange(len(pairs), dtype=np.int64) - sep_ptr[pair_part]
    seg_val.append(et_off[pair_part] + jloc * sizes[pair_part])
    seg_x.append(pp[pair_part])
    seg_len.append(sizes[pair_part])
    seg_row.append(nI + sep_idx.astype(np.int64))
    seg_val.append(gvec_off + pp[:-1])
    seg_x.append(pp[:-1])
    seg_len.append(sizes)
    seg_row.append(n + np.arange(P, dtype=np.int64))
    seg_val, seg_x, seg_len, seg_row = (np.concatenate(a) for a in (seg_val, seg_x, seg_len, seg_row))
    o2 = np.argsort(seg_row, kind="stable")
    seg_ptr = np.concatenate([[0], np.cumsum(np.bincount(seg_row, minlength=n + P))]).astype(np.int32)
    c = np.ascontiguousarray
    return dict(
        n_interior=nI, n_sep=nS, n_parts=P, part_ptr=c(pp.astype(np.int32)), sep_ptr=c(sep_ptr), sep_idx=c(sep_idx),
        ent_ptr=c(ent_ptr), ent_row=c(rows.astype(np.int32)), ent_val=c(vals.astype(np.float64)),
        node_ptr=c(node_ptr), node_pair=c(node_order.astype(np.int32)),
        ass_indptr=c(ASS.indptr.astype(np.int32)), ass_indices=c(ASS.indices.astype(np.int32)), ass_data=c(ASS.data.astype(np.float64)),
        seg_ptr=c(seg_ptr), seg_val=c(seg_val[o2].astype(np.int64)), seg_x=c(seg_x[o2].astype(np.int32)),
        seg_len=c(seg_len[o2].astype(np.int32)), g_off=c(g_off[:-1].astype(np.int64)), et_off=c(et_off[:-1].astype(np.int64)),
        gvec_off=gvec_off, n_vals=gvec_off + nI, e_off=c(e_off[:-1].astype(np.int64)), n_e=int(e_off[-1]),
        max_sep=int(s_cnt.max()), parts=P, separator=nS,
        bytes_per_solve=int(8 * ((sizes * sizes).sum() + 2 * (sizes * s_cnt).sum())
                            + 8 * ((nS + 127) // 128) * (((nS + 127) // 128) + 1) // 2 * 128 * 128),
    )
