"""Domain decomposition of the mesh for one-process-per-GPU runs.

The reference is single-process (no collectives exist).  The stencil is nearest
neighbour on the site graph, so the mesh is cut into ``world`` compact pieces; each rank owns its
sites, keeps ghost copies of the neighbouring sites it couples to and every edge with at least
one owned endpoint (cut edges are duplicated, so J on a cut edge is computed identically on both
sides).  METIS is not available on the target image; the partitioner is recursive coordinate
bisection (balanced to one site, compact pieces, cut ~ 2 sqrt(n/P) edges per interface on a
square film -- what a k-way graph partitioner would also find for this geometry).

Exchange pattern per step: ghost values of psi (once), of the PCG direction and
the level-0 smoothing iterates (per PCG iteration), of mu (once); sums over ranks for the dot
products and for the restricted residual of the replicated coarse AMG levels.

Everything here is deterministic and computed redundantly by every rank from the global mesh, so
send and receive lists agree without any negotiation.
"""

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee


def rcb_partition(sites: np.ndarray, nparts: int) -> np.ndarray:
    """Recursive coordinate bisection: ``part[i]`` in ``0..nparts-1``.  Splits are made along
    the longer extent of the current piece, proportionally when ``nparts`` is not a power of 2."""
    part = np.zeros(len(sites), dtype=np.int32)

    def split(idx, first, count):
        if count == 1:
            part[idx] = first
            return
        left = count // 2
        pts = sites[idx]
        axis = int(np.argmax(np.ptp(pts, axis=0)))
        # stable order: coordinate, then the other coordinate, then index -> deterministic
        order = np.lexsort((idx, pts[:, 1 - axis], pts[:, axis]))
        k = int(round(len(idx) * left / count))
        split(idx[order[:k]], first, left)
        split(idx[order[k:]], first + left, count - left)

    split(np.arange(len(sites)), 0, int(nparts))
    return part


@dataclass
class LocalProblem:
    """What one rank needs: its sub-mesh in local numbering and the halo plan."""

    rank: int
    world: int
    n_global: int
    n_own: int
    local_to_global: np.ndarray          # [n_loc] owned first, then ghosts grouped by owner
    edge_local_to_global: np.ndarray     # [m_loc]
    owned_edge_mask: np.ndarray          # [m_loc] True where this rank reports the edge
    mesh: object                         # Mesh-like with local arrays (tdgl_amd.finite_volume)
    fixed_sites: np.ndarray              # local ids (owned or ghost)
    neighbors: List[int] = field(default_factory=list)
    send_idx: Dict[int, np.ndarray] = field(default_factory=dict)   # local owned ids per neighbour
    recv_range: Dict[int, tuple] = field(default_factory=dict)      # (start, stop) into local ids
    boundary_positions: np.ndarray = None  # positions (in the GLOBAL boundary list) of local boundary edges
    n_interior: int = 0                    # leading owned sites with no neighbour on another rank

    @property
    def n_loc(self):
        return len(self.local_to_global)

    @property
    def n_ghost(self):
        return self.n_loc - self.n_own


def build_local_problem(mesh, part: np.ndarray, rank: int, fixed_sites=None) -> LocalProblem:
    """Cut ``mesh`` (global, tdgl_amd.finite_volume.Mesh) for ``rank``."""
    from .finite_volume import EdgeMesh, Mesh

    em = mesh.edge_mesh
    n = len(mesh.sites)
    world = int(part.max()) + 1
    e0, e1 = em.edges[:, 0], em.edges[:, 1]
    own_mask = part == rank
    touch = own_mask[e0] | own_mask[e1]
    edge_ids = np.flatnonzero(touch)
    # owned sites in reverse Cuthill-McKee order of the owned sub-graph (gather locality)
    owned = np.flatnonzero(own_mask)
    both = own_mask[e0] & own_mask[e1]
    g2o = np.full(n, -1, dtype=np.int64)
    g2o[owned] = np.arange(len(owned))
    sub = sp.csr_matrix(
        (np.ones(2 * int(both.sum()), dtype=np.int8),
         (np.concatenate([g2o[e0[both]], g2o[e1[both]]]), np.concatenate([g2o[e1[both]], g2o[e0[both]]]))),
        shape=(len(owned), len(owned)),
    )
    owned = owned[np.asarray(reverse_cuthill_mckee(sub, symmetric_mode=True))]
    # interior sites (no neighbour on another rank) first, each group in RCM order: the stencil
    # kernels process that prefix while a halo exchange is in flight (csrc/comm.inc)
    on_cut = np.zeros(n, dtype=bool)
    cut_edges = touch & ~both
    on_cut[e0[cut_edges]] = True
    on_cut[e1[cut_edges]] = True
    owned = np.concatenate([owned[~on_cut[owned]], owned[on_cut[owned]]])
    n_interior = int((~on_cut[owned]).sum())
    # ghosts: the other endpoint of cut edges, grouped by owner, ascending global id inside a group
    ends = np.concatenate([e0[edge_ids], e1[edge_ids]])
    ghosts = np.unique(ends[~own_mask[ends]])
    ghosts = ghosts[np.lexsort((ghosts, part[ghosts]))]
    l2g = np.concatenate([owned, ghosts]).astype(np.int64)
    g2l = np.full(n, -1, dtype=np.int64)
    g2l[l2g] = np.arange(len(l2g))
    n_own = len(owned)

    neighbors = sorted(set(part[ghosts].tolist()))
    recv_range, send_idx = {}, {}
    for nb in neighbors:
        sel = np.flatnonzero(part[ghosts] == nb)
        recv_range[nb] = (n_own + int(sel[0]), n_own + int(sel[-1]) + 1)
        # what `nb` needs from us = our owned sites adjacent to its owned sites, ascending global id
        nb_mask = part == nb
        cut = (own_mask[e0] & nb_mask[e1]) | (own_mask[e1] & nb_mask[e0])
        mine = np.where(own_mask[e0[cut]], e0[cut], e1[cut])
        send_idx[nb] = g2l[np.unique(mine)]

    local_edges = np.column_stack([g2l[e0[edge_ids]], g2l[e1[edge_ids]]])
    # an edge is reported by the owner of its first endpoint (cut edges exist on two ranks)
    owned_edge = own_mask[e0[edge_ids]]
    is_b = np.zeros(len(em.edges), dtype=bool)
    is_b[em.boundary_edge_indices] = True
    pos_of_edge = np.full(len(em.edges), -1, dtype=np.int64)
    pos_of_edge[em.boundary_edge_indices] = np.arange(len(em.boundary_edge_indices))
    local_b = np.flatnonzero(is_b[edge_ids])
    lem = EdgeMesh(
        em.centers[edge_ids], local_edges, local_b, em.directions[edge_ids],
        em.edge_lengths[edge_ids], em.dual_edge_lengths[edge_ids],
    )
    lmesh = Mesh(mesh.sites[l2g], np.zeros((0, 3), dtype=np.int64), np.array([], dtype=np.int64),
                 areas=mesh.areas[l2g], edge_mesh=lem)
    fixed = np.array([], dtype=np.int64) if fixed_sites is None else np.asarray(fixed_sites, dtype=np.int64)
    fixed_local = g2l[fixed]
    fixed_local = fixed_local[fixed_local >= 0]
    return LocalProblem(
        rank=rank, world=world, n_global=n, n_own=n_own, n_interior=n_interior, local_to_global=l2g,
        edge_local_to_global=edge_ids, owned_edge_mask=owned_edge, mesh=lmesh, fixed_sites=fixed_local,
        neighbors=neighbors, send_idx=send_idx, recv_range=recv_range,
        boundary_positions=pos_of_edge[edge_ids[local_b]],
    )


def local_hierarchy_level0(h, lp: LocalProblem):
    """Slice level 0 of the GLOBAL AMG hierarchy ``h`` (built in global numbering) for one rank:

    * ``A``  rows = owned sites, columns = local sites (owned + ghosts);
    * ``P``  rows = all local sites (ghost rows too: the coarse correction of a ghost value is
             then computed locally, saving one exchange per cycle), columns = global coarse;
    * ``R``  = ``P[owned].T``: the restriction restricted to owned fine columns -- its product
             with the owned residual is this rank's PARTIAL coarse right-hand side; the partials
             are summed over ranks.
    Coarser levels are replicated on every rank unchanged.
    """
    lv0 = h.levels[0]
    l2g = lp.local_to_global
    own = l2g[: lp.n_own]
    g2l = np.full(lv0.A.shape[0], -1, dtype=np.int64)
    g2l[l2g] = np.arange(len(l2g))
    A = lv0.A.tocsr()[own]
    A = sp.csr_matrix((A.data, g2l[A.indices], A.indptr), shape=(lp.n_own, len(l2g)))
    assert A.indices.min() >= 0, "halo does not cover the stencil"
    A.sort_indices()
    P = lv0.P.tocsr()[l2g]
    P.sort_indices()
    R = P[: lp.n_own].T.tocsr()
    R.sort_indices()
    # dinv for ALL local sites: the fused level-0 residual gathers dinv * r at ghost columns
    return dict(A=A, dinv=lv0.dinv[l2g], rho=lv0.rho, P=P, R=R)


# ---------------------------------------------------------------------------------------
# Two distributed AMG levels, ONE vector exchange per PCG iteration.
#
# `local_hierarchy_level0` above keeps every level below 0 replicated: each iteration then sums a level-1-sized
# partial right-hand side over all ranks (1.5 MB at 4M sites) and repeats the level-1 work on every rank.  Here
#
#   * the aggregates of level 0 never straddle ranks (`amg.build_hierarchy(part=...)`), so every level-1 row has an
#     owner; level 1 is distributed like level 0, the replicated part starts at level 2 (20k rows at 4M sites);
#   * level 1 is used through its explicit operators (`amg.collapsed_operators`: b2 = M1 b1 on the way down,
#     x1 = W1 b1 + V1 x2 on the way up), sliced per rank;
#   * every rank computes redundantly what it would otherwise have to receive: x1 on the level-1 rows its
#     prolongation reads, b1 on the columns W1 reads there, z on its first ghost layer -- all from ONE exchange
#     per iteration, the residual r on a deep ghost zone (the closure of those stencils: 5-9 % of the owned rows at
#     500k rows per rank), instead of three exchanges (r, z / p, the smoothing iterate) of the first ghost layer.
#
# Per iteration a rank then communicates three times: the deep exchange of r, the sum of the partial level-2
# right-hand sides (n2 values), the CG's dot products.
@dataclass
class DeepPlan:
    """A rank's piece of the two-level decomposition.  Fine numbering: [owned | ghost layer 1 (the `LocalProblem`'s
    ghosts, same order) | ghost layer 2 | deeper ghosts], the two outer groups by owner, ascending global id; level-1
    numbering: [owned | further rows the prolongation reads | further columns W1 reads]."""

    n_own: int
    n1: int                       # owned + ghost layer 1 (= LocalProblem.n_loc): rows of A, rows where z is formed
    n2: int                       # + ghost layer 2: rows of P (where x is formed)
    n_ext: int                    # + deeper ghosts: where r is needed
    ext_to_global: np.ndarray     # [n_ext]
    neighbors: List[int]          # owners of the ghosts, ascending
    send_idx: Dict[int, np.ndarray]   # per neighbour: local OWNED ids whose values it needs, ascending global id
    recv_idx: Dict[int, np.ndarray]   # per neighbour: local ghost ids receiving them, same order
    l1_own: int
    l1_x: int                     # rows of W / V (level-1 values the prolongation reads)
    l1_loc: int                   # columns of W = rows of F
    l1_to_global: np.ndarray      # [l1_loc]
    A: sp.csr_matrix              # [n1, n2]    level-0 operator rows
    dinv: np.ndarray              # [n_ext]
    P: sp.csr_matrix              # [n2, l1_x]
    F: sp.csr_matrix              # [l1_loc, n_ext]   R0 (I - c A0 D0^-1) rows: b1 = F r
    M: sp.csr_matrix              # [n_level2, l1_own]   partial b2 = M b1[owned]
    W: sp.csr_matrix              # [l1_x, l1_loc]
    V: sp.csr_matrix              # [l1_x, n_level2]
    rho: float = 0.0
    c: float = 0.0                # the level-0 smoothing coefficient F was built for
    l1_interior: int = 0          # leading owned level-1 rows whose row of F has owned columns only


def deep_plan_applicable(hierarchy, plan) -> bool:
    """Level 1 must be an intermediate level of the collapsed chain with explicit operators both ways, and the
    hierarchy must have been built with per-rank aggregates."""
    return (plan is not None and plan.get("tail", 0) >= 2 and 1 in plan.get("mid", {}) and 1 in plan.get("up", {})
            and getattr(hierarchy.levels[0], "owner", None) is not None and len(hierarchy.levels) >= 3
            and getattr(hierarchy.levels[1], "owner", None) is not None)


def _cols_of_rows(M: sp.csr_matrix, rows: np.ndarray) -> np.ndarray:
    ip = M.indptr
    if len(rows) == 0:
        return np.empty(0, dtype=np.int64)
    counts = ip[rows + 1] - ip[rows]
    starts = np.repeat(ip[rows] - np.concatenate([[0], np.cumsum(counts)[:-1]]), counts)
    return np.unique(M.indices[starts + np.arange(int(counts.sum()))])


def _rows_remapped(M: sp.csr_matrix, rows: np.ndarray, col_map: np.ndarray, n_cols: int) -> sp.csr_matrix:
    """``M[rows]`` with global column ids replaced through ``col_map`` (every column must be present)."""
    ip = M.indptr
    counts = (ip[rows + 1] - ip[rows]).astype(np.int64)
    total = int(counts.sum())
    pos = np.repeat(ip[rows].astype(np.int64) - np.concatenate([[0], np.cumsum(counts)[:-1]]), counts) + np.arange(total)
    cols = col_map[M.indices[pos]]
    assert total == 0 or cols.min() >= 0, "the ghost zone does not cover a stencil"
    out = sp.csr_matrix((M.data[pos], cols, np.concatenate([[0], np.cumsum(counts)])), shape=(len(rows), n_cols))
    out.sort_indices()
    return out


class DeepPlanner:
    """Holds the global operators of the two distributed levels (built once by the root) and cuts a rank's piece."""

    def __init__(self, hierarchy, plan, part: np.ndarray, cheb_lo: float = 0.1, smoother: str = "chebyshev"):
        from .amg import fused_restriction, smoother_coefficients

        if not deep_plan_applicable(hierarchy, plan):
            raise ValueError("the hierarchy / collapsed plan does not allow two distributed levels")
        lv0 = hierarchy.levels[0]
        self.part = np.asarray(part)
        self.owner1 = np.asarray(hierarchy.levels[1].owner)
        self.A0 = lv0.A.tocsr()
        self.P0 = lv0.P.tocsr()
        self.dinv0, self.rho0 = lv0.dinv, lv0.rho
        self.c = float(smoother_coefficients(lv0.rho, 1, smoother, cheb_lo)[1][0])
        self.F = fused_restriction(hierarchy, self.c).tocsr()
        self.F_pattern = sp.csr_matrix((np.ones(self.F.nnz), self.F.indices, self.F.indptr), shape=self.F.shape)
        self.M1 = plan["mid"][1].tocsc()
        W, V = plan["up"][1]
        self.W1, self.V1 = W.tocsr(), V.tocsr()
        self.n_level2 = hierarchy.levels[2].A.shape[0]
        for M in (self.A0, self.P0, self.F, self.W1, self.V1):
            M.sort_indices()

    def cut(self, lp: "LocalProblem") -> DeepPlan:
        part, n = self.part, len(self.part)
        r = lp.rank
        l2g1 = lp.local_to_global            # owned + ghost layer 1, the mesh kernels' numbering
        own = l2g1[: lp.n_own]
        seen = np.zeros(n, dtype=bool)
        seen[l2g1] = True

        def by_owner(ids):
            return ids[np.lexsort((ids, part[ids]))]

        # z is formed on owned + layer 1 -> x (rows of P) on one layer more
        g2 = _cols_of_rows(self.A0, l2g1)
        g2 = by_owner(g2[~seen[g2]])
        seen[g2] = True
        x_rows = np.concatenate([l2g1, g2])
        # level 1: the rows the prolongation reads, the columns W reads there
        own1 = np.flatnonzero(self.owner1 == r)
        # (owned level-1 rows in the order of their first fine member: gather locality follows the fine RCM order)
        first = np.full(len(self.owner1), np.iinfo(np.int64).max)
        agg_cols = self.P0[own]  # rows in local order
        rows_rep = np.repeat(np.arange(lp.n_own), np.diff(agg_cols.indptr))
        np.minimum.at(first, agg_cols.indices, rows_rep)
        own1 = own1[np.argsort(first[own1], kind="stable")]
        # ... those whose restriction reads no ghost value first: formed while the exchange of r is in flight
        touches_ghost = (self.F_pattern[own1] @ (part != r).astype(np.float64)) > 0
        own1 = np.concatenate([own1[~touches_ghost], own1[touches_ghost]])
        l1_interior = int((~touches_ghost).sum())
        seen1 = np.zeros(len(self.owner1), dtype=bool)
        seen1[own1] = True
        xr = _cols_of_rows(self.P0, x_rows)
        xr = xr[~seen1[xr]]
        xr = xr[np.lexsort((xr, self.owner1[xr]))]
        seen1[xr] = True
        l1_x = np.concatenate([own1, xr])
        bc = _cols_of_rows(self.W1, l1_x)
        bc = bc[~seen1[bc]]
        bc = bc[np.lexsort((bc, self.owner1[bc]))]
        l1_loc = np.concatenate([l1_x, bc])
        # r wherever F reads it for those level-1 rows
        deep = _cols_of_rows(self.F, l1_loc)
        deep = by_owner(deep[~seen[deep]])
        ext = np.concatenate([x_rows, deep])
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[ext] = np.arange(len(ext))
        g2l1 = np.full(len(self.owner1), -1, dtype=np.int64)
        g2l1[l1_loc] = np.arange(len(l1_loc))
        ghosts = ext[lp.n_own:]
        neighbors = sorted(set(part[ghosts].tolist()))
        recv_idx = {}
        for nb in neighbors:
            mine = ghosts[part[ghosts] == nb]
            mine = np.sort(mine)  # ascending global id: the order the owner packs in
            recv_idx[nb] = g2l[mine]
        M_loc = self.M1[:, own1].tocsr()
        M_loc.sort_indices()
        return DeepPlan(
            n_own=lp.n_own, n1=len(l2g1), n2=len(x_rows), n_ext=len(ext), ext_to_global=ext, neighbors=neighbors,
            send_idx={}, recv_idx=recv_idx, l1_own=len(own1), l1_x=len(l1_x), l1_loc=len(l1_loc), l1_to_global=l1_loc,
            A=_rows_remapped(self.A0, l2g1, g2l, len(x_rows)), dinv=self.dinv0[ext],
            P=_rows_remapped(self.P0, x_rows, g2l1, len(l1_x)), F=_rows_remapped(self.F, l1_loc, g2l, len(ext)),
            M=M_loc, W=_rows_remapped(self.W1, l1_x, g2l1, len(l1_loc)),
            V=sp.csr_matrix(self.V1[l1_x]), rho=float(self.rho0), c=self.c, l1_interior=l1_interior,
        )


def link_deep_plans(plans: Dict[int, DeepPlan], lps: Dict[int, "LocalProblem"]) -> None:
    """Fill every plan's send lists from the others' receive lists (what rank q receives from p, p sends: its local
    ids of those sites, ascending global id).  Needs the plans of all ranks: the root builds them together."""
    for p, plan in plans.items():
        g2l = {}
        own = lps[p].local_to_global[: lps[p].n_own]
        lookup = np.full(int(own.max()) + 1 if len(own) else 1, -1, dtype=np.int64)
        lookup[own] = np.arange(len(own))
        plan.send_idx = {}
        for q, other in plans.items():
            if q == p or p not in other.recv_idx:
                continue
            wanted = other.ext_to_global[other.recv_idx[p]]  # ascending global id by construction
            loc = lookup[wanted]
            assert (loc >= 0).all()
            plan.send_idx[q] = loc
        # (a rank we send to is not necessarily one we receive from: the two lists are kept separately)
