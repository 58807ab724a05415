"""Rank-level nested dissection: the CG's second preconditioner in one-process-per-GPU mode.

The reference solves ``L_mu mu = rhs`` with one sparse LU on one core (tdgl/solver/solver.py:516, factorised at
tdgl/finite_volume/operators.py:305-308).  On one GPU this package applies three levels of nested dissection with
explicit factors as a preconditioner of the CG (`tdgl_poisson_set_substructure_precond`); cut across ranks, the SAME
construction gets one more level on top -- the cut between the ranks:

* Gamma = a vertex cover of the edges between ranks (the endpoint on the lower rank); every rank's remaining sites
  I_r are coupled to other ranks through Gamma only, so ``A_II`` is block diagonal over the ranks and every block is
  positive definite (Dirichlet data on Gamma);
* every rank factorises ITS block by itself (one to three local levels, `substructure.build_substructure_levels` with
  zero weights: no gauge to carry, the block is not singular) -- no communication in the set-up except the dense
  interface complement ``S = A_GG - sum_r A_GI_r A_II_r^-1 A_I_rG``, summed once over the bootstrap group and
  pseudo-inverted on every rank (|Gamma| ~ 2 sqrt(n N): 5k sites for 1M sites on 8 ranks);
* one application ``z = M r``:  ``y_I = A_II^-1 r_I`` (local),  ``t = r_G - sum_r A_GI_r y_I`` (ONE all-reduce of
  |Gamma| doubles),  ``x_G = S^+ t`` (replicated dense product),  ``x_I = y_I - A_II^-1 (A_IG x_G)`` (a second local
  solve; the explicit ``A_II^-1 A_IG`` would be n_I x |Gamma_r| dense).

With exact local solves ``M = A^+`` exactly (nested dissection is a direct method); with the fp32-stored factors it
contracts the residual by ~1e-6 per application, so the CG needs ONE iteration from the projection guess: per step
one sum of |Gamma| doubles, one exchange of z's first ghost layer and the CG's one sum of 3 x 1024 partials, where the
two-level distributed AMG cycle needs ~9 iterations of (exchange + two sums).

This module holds the host side: the cover, each rank's index sets and coupling blocks, the local dissection order.
`distributed.DistributedTDGL` drives the device (`tdgl_poisson_set_schur_*`), `tests/dist_model.py` restates the
application in NumPy for the CPU tests.
"""

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


def interface_cover(edges: np.ndarray, part: np.ndarray) -> np.ndarray:
    """``is_gamma[n]``: the endpoint on the LOWER rank of every edge between two ranks."""
    i, j = edges[:, 0], edges[:, 1]
    g = np.zeros(len(part), dtype=bool)
    g[i[part[i] < part[j]]] = True
    g[j[part[j] < part[i]]] = True
    return g


@dataclass
class SchurPiece:
    """One rank's share of the rank-level dissection (local ids are `LocalProblem` ids: owned sites first)."""

    n_gamma: int                 # |Gamma|, global
    interior: np.ndarray         # [n_I] local ids of the owned sites outside Gamma, in the LOCAL DISSECTION order
    ptrs: list                   # pointer arrays of the local dissection (one per level), positions in `interior`
    gamma_owned_local: np.ndarray  # local ids of the Gamma sites this rank owns
    gamma_owned_gid: np.ndarray    # ... and their positions in Gamma
    A_II: sp.csr_matrix          # [n_I, n_I] in dissection order
    A_IG: sp.csr_matrix          # [n_I, n_gamma]
    A_GI: sp.csr_matrix          # [n_gamma, n_I] = A_IG^T
    A_GG_owned: sp.csr_matrix    # [n_gamma, n_gamma]: the rows of A_GG this rank owns (zero elsewhere)

    @property
    def n_interior(self):
        return len(self.interior)


def local_dissection(sites: np.ndarray, edges: np.ndarray, blocks=None):
    """Dissection order of a rank's interior: ``(perm, ptrs)`` with one, two or three levels by size (``blocks`` =
    (part, super-block, super-super-block) sizes; the product's by default)."""
    from .substructure import substructure_order, substructure_order2, substructure_order3

    n = len(sites)
    if blocks is None:
        blocks = (144, 3072, 24576)  # (`hipcore.TDGLContext.PD_BLOCKS`: the sizes measured best for the factors as preconditioner)
    b1, b2, b3 = blocks
    if n >= 6 * b3 // 2 and n >= 4 * b2:  # three levels need a handful of super-super-blocks
        perm, p1, p2, p3 = substructure_order3(sites, edges, b1, b2, b3)
        return perm, [p1, p2, p3]
    if n >= 3 * b2:
        perm, p1, p2 = substructure_order2(sites, edges, b1, b2)
        return perm, [p1, p2]
    perm, p1 = substructure_order(sites, edges, b1)
    return perm, [p1]


def build_piece(lp, is_gamma_local: np.ndarray, gamma_gid_local: np.ndarray, n_gamma: int, blocks=None) -> SchurPiece:
    """``is_gamma_local`` / ``gamma_gid_local``: over the rank's LOCAL sites (owned, then ghosts): in Gamma? / position
    in Gamma (-1 outside).  The rows of A of every owned site are complete locally (all its edges are local)."""
    em = lp.mesh.edge_mesh
    e0, e1 = em.edges[:, 0].astype(np.int64), em.edges[:, 1].astype(np.int64)
    w = em.dual_edge_lengths / em.edge_lengths
    n_loc, n_own = lp.n_loc, lp.n_own
    owned = np.zeros(n_loc, dtype=bool)
    owned[:n_own] = True
    interior_mask = owned & ~is_gamma_local
    interior0 = np.flatnonzero(interior_mask)
    # every neighbour of an interior site is interior (same rank) or in Gamma
    other = np.concatenate([e1[interior_mask[e0]], e0[interior_mask[e1]]])
    if np.any(~interior_mask[other] & ~is_gamma_local[other]):
        raise ValueError("schur_dd: the interface does not cover every edge between ranks")
    # local dissection of the interior sub-graph
    pos0 = np.full(n_loc, -1, dtype=np.int64)
    pos0[interior0] = np.arange(len(interior0))
    both = interior_mask[e0] & interior_mask[e1]
    sub_edges = np.column_stack([pos0[e0[both]], pos0[e1[both]]])
    perm, ptrs = local_dissection(lp.mesh.sites[interior0], sub_edges, blocks)
    interior = interior0[perm]
    pos = np.full(n_loc, -1, dtype=np.int64)
    pos[interior] = np.arange(len(interior))
    nI = len(interior)
    # A_II
    i, j, ww = pos[e0[both]], pos[e1[both]], w[both]
    diag = np.zeros(nI)
    # (the diagonal of an interior row sums ALL its edges, also those into Gamma)
    for a, b in ((e0, e1), (e1, e0)):
        m = interior_mask[a]
        np.add.at(diag, pos[a[m]], w[m])
    A_II = sp.coo_matrix((np.concatenate([-ww, -ww, diag]), (np.concatenate([i, j, np.arange(nI)]), np.concatenate([j, i, np.arange(nI)]))),
                         shape=(nI, nI)).tocsr()
    A_II.sum_duplicates()
    A_II.sort_indices()
    # A_IG: interior -- Gamma edges (either orientation)
    rows, cols, vals = [], [], []
    for a, b in ((e0, e1), (e1, e0)):
        m = interior_mask[a] & is_gamma_local[b]
        rows.append(pos[a[m]])
        cols.append(gamma_gid_local[b[m]])
        vals.append(-w[m])
    A_IG = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nI, n_gamma)).tocsr()
    A_IG.sum_duplicates()
    A_IG.sort_indices()
    # owned rows of A_GG: diagonal (all edges of the site) and Gamma -- Gamma edges
    g_owned_local = np.flatnonzero(owned & is_gamma_local)
    g_owned_gid = gamma_gid_local[g_owned_local]
    own_g = owned & is_gamma_local
    rows, cols, vals = [g_owned_gid], [g_owned_gid], [np.zeros(len(g_owned_gid))]
    dg = np.zeros(n_loc)
    for a, b in ((e0, e1), (e1, e0)):
        m = own_g[a]
        np.add.at(dg, a[m], w[m])
        m2 = m & is_gamma_local[b]
        rows.append(gamma_gid_local[a[m2]])
        cols.append(gamma_gid_local[b[m2]])
        vals.append(-w[m2])
    vals[0] = dg[g_owned_local]
    A_GG = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n_gamma, n_gamma)).tocsr()
    A_GG.sum_duplicates()
    A_GG.sort_indices()
    A_GI = A_IG.T.tocsr()
    A_GI.sort_indices()
    return SchurPiece(n_gamma=int(n_gamma), interior=interior, ptrs=[np.asarray(p, dtype=np.int32) for p in ptrs],
                      gamma_owned_local=g_owned_local, gamma_owned_gid=g_owned_gid, A_II=A_II, A_IG=A_IG, A_GI=A_GI, A_GG_owned=A_GG)


def gamma_numbering(is_gamma: np.ndarray):
    """Global positions in Gamma: ascending site id.  Returns ``(gid[n] with -1 outside, n_gamma)``."""
    gid = np.full(len(is_gamma), -1, dtype=np.int64)
    idx = np.flatnonzero(is_gamma)
    gid[idx] = np.arange(len(idx))
    return gid, len(idx)


def interface_pinv(S: np.ndarray) -> np.ndarray:
    """Pseudo-inverse of the interface complement (singular like A: its null space is the constants)."""
    m = S.shape[0]
    S = 0.5 * (S + S.T)
    s = float(np.diag(S).mean())
    inv = np.linalg.inv(S + s / m)
    return 0.5 * (inv + inv.T) - 1.0 / (s * m)
