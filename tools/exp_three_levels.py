"""Experiment (host only): what a THIRD level of nested dissection would stream per solve.  The two-level direct solve keeps
the pseudo-inverse of the top separator's Schur complement as one dense matrix (224 MB of 0.76 GB at 250k sites, 392 MB of
1.7 GB at 450k); here that system is cut once more -- super-super-blocks of `m3` sites with a top-top separator -- with the
product's own construction (`substructure.build_substructure` applied to the level's Schur complement), the solution is
checked against the matrix, and the bytes of every level are printed.

    python tools/exp_three_levels.py [L=465] [m1=160] [m2=4096] [m3=32768]"""
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from helpers import synthetic_mesh  # noqa: E402
from tdgl_amd.hipcore import poisson_matrix  # noqa: E402
from tdgl_amd.partition import rcb_partition  # noqa: E402
from tdgl_amd.substructure import build_substructure, schur_pinv  # noqa: E402

L, m1, m2, m3 = (int(a) for a in (sys.argv[1:5] + ["465", "160", "4096", "32768"][len(sys.argv) - 1:]))
mesh = synthetic_mesh(L)
sites, edges = np.asarray(mesh.sites), mesh.edge_mesh.edges
n = len(sites)
i, j = edges[:, 0], edges[:, 1]


def cover(label, active):
    """vertex cover of the edges between different labels among the active sites: the endpoint with the lower label"""
    m = active[i] & active[j]
    out = np.zeros(n, dtype=bool)
    out[i[m & (label[i] < label[j])]] = True
    out[j[m & (label[j] < label[i])]] = True
    return out


t0 = time.time()
big = rcb_partition(sites, max(2, round(n / m3)))
is_TT = cover(big, np.ones(n, dtype=bool))
sup = np.full(n, -1, dtype=np.int64)
n_sup = 0
for b in range(big.max() + 1):
    idx = np.flatnonzero((big == b) & ~is_TT)
    k = max(1, round(len(idx) / m2))
    sup[idx] = n_sup + (rcb_partition(sites[idx], k) if k > 1 else 0)
    n_sup += k
is_T = cover(sup, ~is_TT)
part = np.full(n, -1, dtype=np.int64)
n_part = 0
for q in range(n_sup):
    idx = np.flatnonzero((sup == q) & ~is_T)
    k = max(1, round(len(idx) / m1))
    part[idx] = n_part + (rcb_partition(sites[idx], k) if k > 1 else 0)
    n_part += k
is_S = cover(part, ~is_TT & ~is_T)
n_big = int(big.max()) + 1
group = np.where(is_TT, n_part + n_sup + n_big, np.where(is_T, n_part + n_sup + big, np.where(is_S, n_part + sup, part)))
perm = np.lexsort((np.arange(n), group))
counts = np.bincount(group, minlength=n_part + n_sup + n_big + 1)
ptr1 = np.concatenate([[0], np.cumsum(counts[:n_part])])
ptr2 = ptr1[-1] + np.concatenate([[0], np.cumsum(counts[n_part:n_part + n_sup])])
ptr3 = ptr2[-1] + np.concatenate([[0], np.cumsum(counts[n_part + n_sup:n_part + n_sup + n_big])])
iperm = np.empty(n, dtype=np.int64)
iperm[perm] = np.arange(n)
A = poisson_matrix(edges.astype(np.int64), mesh.edge_mesh.dual_edge_lengths / mesh.edge_mesh.edge_lengths, n, iperm)
print(f"{n} sites: {n_part} parts, {n_sup} super-blocks, {n_big} super-super-blocks; separators {ptr2[0] and n - ptr1[-1]} / "
      f"{n - ptr2[-1]} / {n - ptr3[-1]}; ordering {time.time() - t0:.1f} s")


def schur_sparse(lv, Acur, nI):
    """the level's Schur complement as a sparse matrix (the C blocks do not overlap outside the next separator)"""
    rows, cols, vals = [], [], []
    for idx, C in zip(lv.sep_idx, lv.C):
        k = len(idx)
        rows.append(np.repeat(idx, k)); cols.append(np.tile(idx, k)); vals.append(-C.ravel())
    ASS = Acur[nI:, nI:].tocoo()
    rows.append(ASS.row); cols.append(ASS.col); vals.append(ASS.data)
    nS = Acur.shape[0] - nI
    S = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nS, nS)).tocsr()
    return (0.5 * (S + S.T)).tocsr()


t0 = time.time()
lv1 = build_substructure(A, ptr1, with_schur=False)
S1 = schur_sparse(lv1, A, int(ptr1[-1]))
lv2 = build_substructure(S1, ptr2 - ptr1[-1], weights=1.0 + lv1.u, with_schur=False)
S2 = schur_sparse(lv2, S1, int(ptr2[-1] - ptr1[-1]))
lv3 = build_substructure(S2, ptr3 - ptr2[-1], weights=lv2.u)
print(f"factors in {time.time() - t0:.1f} s")
sym = lambda m: 8 * ((m + 127) // 128) * (((m + 127) // 128) + 1) // 2 * 128 * 128
mb = lambda lv: (sum(8 * g.size for g in lv.G) / 1e6, sum(8 * e.size for e in lv.E) / 1e6)
(g1, e1), (g2, e2), (g3, e3) = mb(lv1), mb(lv2), mb(lv3)
two = g1 + e1 + g2 + e2 + sym(lv2.n_sep) / 1e6
three = g1 + e1 + g2 + e2 + g3 + e3 + sym(lv3.n_sep) / 1e6
print(f"MB per solve (sparse separator right-hand sides: E once): level 1 G {g1:.0f} E {e1:.0f} | level 2 G {g2:.0f} E {e2:.0f} | "
      f"two levels: top separator {lv2.n_sep} sites, {sym(lv2.n_sep) / 1e6:.0f} -> {two:.0f} | three levels: G {g3:.0f} E {e3:.0f}, "
      f"top separator {lv3.n_sep} sites, {sym(lv3.n_sep) / 1e6:.0f} -> {three:.0f}")

# the three-level sequence on the host, against the matrix
b = np.random.default_rng(0).standard_normal(n)
b -= b.mean()


def down(lv, vec, coupling):
    nI = lv.n_interior
    y = np.empty(nI)
    tot = 0.0
    for p in range(lv.n_parts):
        a, e = int(lv.part_ptr[p]), int(lv.part_ptr[p + 1])
        y[a:e] = lv.G[p] @ vec[a:e]
        tot += lv.g[a:e] @ vec[a:e]
    return y, vec[nI:] - coupling @ y, tot


def up(lv, y, xs):
    x = np.empty(lv.n)
    for p in range(lv.n_parts):
        a, e = int(lv.part_ptr[p]), int(lv.part_ptr[p + 1])
        x[a:e] = y[a:e] - lv.E[p] @ xs[lv.sep_idx[p]]
    x[lv.n_interior:] = xs
    return x


y1, r1, t1 = down(lv1, b, A[ptr1[-1]:, :ptr1[-1]])
y2, r2, t2 = down(lv2, r1, S1[lv2.n_interior:, :lv2.n_interior])
y3, r3, t3 = down(lv3, r2, S2[lv3.n_interior:, :lv3.n_interior])
x_top = schur_pinv(lv3.schur) @ r3
mean = (t1 + t2 + t3 + lv3.u @ x_top) / n
x = up(lv1, y1, up(lv2, y2, up(lv3, y3, x_top))) - mean
print(f"three-level solve: ||b - A x|| / ||b|| = {np.linalg.norm(b - A @ x) / np.linalg.norm(b):.1e}, mean {x.mean():.1e}")
