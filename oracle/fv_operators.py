"""Oracle: finite-volume operators as SciPy sparse matrices (TEST INFRASTRUCTURE).

Restates `tdgl/finite_volume/operators.py:59-394` of the reference.  Notation: an edge
``e = (i, j)`` with ``i < j`` has length ``l_e``, Voronoi dual length ``s_e``, direction
``d_e = r_j - r_i`` (un-normalised) and link variable ``U_e = exp(-1j * A_e . d_e)``;
site ``i`` has Voronoi area ``a_i``.

The mesh argument only needs ``.sites``, ``.areas`` and ``.edge_mesh`` with ``edges``,
``boundary_edge_indices``, ``directions``, ``edge_lengths``, ``dual_edge_lengths``.
"""

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def link_variables(link_exponents, directions):
    """``U_e = exp(-i A_e . d_e)`` (operators.py:109-111, 153-155)."""
    if link_exponents is None:
        return np.ones(len(directions))
    return np.exp(-1j * np.einsum("ij, ij -> i", link_exponents, directions))


def divergence_matrix(mesh):
    """Edge -> site divergence, CSR ``n x m`` (operators.py:59-84).

    Row ``i`` has ``+s_e / a_i`` for edges leaving ``i`` (``i = e[0]``) and
    ``-s_e / a_i`` for edges arriving at ``i`` (``i = e[1]``).
    """
    em = mesh.edge_mesh
    m, n = len(em.edges), len(mesh.sites)
    tail, head = em.edges[:, 0], em.edges[:, 1]
    eid = np.arange(m)
    s = em.dual_edge_lengths
    data = np.concatenate([s / mesh.areas[tail], -s / mesh.areas[head]])
    return sp.csr_array(
        (data, (np.concatenate([tail, head]), np.concatenate([eid, eid]))), shape=(n, m)
    )


def gradient_matrix(mesh, link_exponents=None):
    """Site -> edge (covariant) gradient, CSR ``m x n`` (operators.py:87-117).

    ``(grad f)_e = (U_e f_j - f_i) / l_e``.
    """
    em = mesh.edge_mesh
    m, n = len(em.edges), len(mesh.sites)
    inv_len = 1 / em.edge_lengths
    U = link_variables(link_exponents, em.directions)
    eid = np.arange(m)
    data = np.concatenate([U * inv_len, -inv_len])
    cols = np.concatenate([em.edges[:, 1], em.edges[:, 0]])
    return sp.csr_array((data, (np.concatenate([eid, eid]), cols)), shape=(m, n))


def laplacian_matrix(mesh, link_exponents=None, fixed_sites=None, free_rows=None):
    """(Covariant) Laplacian, CSC ``n x n`` (operators.py:120-185).

    Row ``i``: ``(1/a_i) sum_j (s_ij / l_ij) (U_ij f_j - f_i)``.  Rows listed in
    ``fixed_sites`` are replaced by identity rows (eigenvalue 1); the mask is by ROW only,
    so neighbours of a fixed site still couple to it (operators.py:170-181).
    Returns ``(matrix, free_rows_mask)`` like the reference.
    """
    if fixed_sites is None:
        fixed_sites = np.array([], dtype=int)
    em = mesh.edge_mesh
    n = len(mesh.sites)
    w = em.dual_edge_lengths / em.edge_lengths
    U = link_variables(link_exponents, em.directions)
    i, j = em.edges[:, 0], em.edges[:, 1]
    ai, aj = mesh.areas[i], mesh.areas[j]
    rows = np.concatenate([i, j, i, j])
    cols = np.concatenate([j, i, i, j])
    data = np.concatenate([w * U / ai, w * U.conjugate() / aj, -w / ai, -w / aj])
    if free_rows is None:
        free_rows = np.isin(rows, fixed_sites, invert=True)
    rows = np.concatenate([rows[free_rows], fixed_sites])
    cols = np.concatenate([cols[free_rows], fixed_sites])
    data = np.concatenate([data[free_rows], np.ones(len(fixed_sites))])
    return sp.csc_array((data, (rows, cols)), shape=(n, n)), free_rows


def neumann_boundary_matrix(mesh):
    """Boundary-flux matrix ``n x n_b`` (operators.py:188-230, called without fixed
    sites at operators.py:286): ``B[i, k] = l_k / (2 a_i)`` for both end sites of
    boundary edge ``k`` (``k`` = position in ``boundary_edge_indices``)."""
    em = mesh.edge_mesh
    bidx = em.boundary_edge_indices
    k = np.arange(len(bidx))
    be = em.edges[bidx]
    bl = em.edge_lengths[bidx]
    rows = np.concatenate([be[:, 0], be[:, 1]])
    data = np.concatenate(
        [bl / (2 * mesh.areas[be[:, 0]]), bl / (2 * mesh.areas[be[:, 1]])]
    )
    mat = sp.csr_array(
        (data, (rows, np.concatenate([k, k]))), shape=(len(mesh.sites), len(bidx))
    )
    return mat.tocsr()


class FVOperators:
    """Oracle counterpart of the reference's ``MeshOperators`` (operators.py:233-394)."""

    def __init__(self, mesh, fixed_sites=None, fix_psi=True):
        self.mesh = mesh
        self.edges = mesh.edge_mesh.edges
        self.fixed_sites = (
            np.array([], dtype=np.int64) if fixed_sites is None else np.asarray(fixed_sites)
        )
        self.fix_psi = fix_psi
        self.free_rows = None
        self.psi_gradient = None
        self.psi_laplacian = None
        self.link_exponents = None

    def build_operators(self):
        """A-independent operators + SuperLU factorisation (operators.py:282-308)."""
        mesh = self.mesh
        self.mu_laplacian, _ = laplacian_matrix(mesh)
        self.mu_boundary_laplacian = neumann_boundary_matrix(mesh)
        self.mu_gradient = gradient_matrix(mesh)
        self.divergence = divergence_matrix(mesh)
        spla.use_solver(useUmfpack=False)
        self.mu_laplacian_lu = spla.factorized(self.mu_laplacian)

    def set_link_exponents(self, link_exponents):
        """(Re)build the covariant gradient / Laplacian for psi (operators.py:310-383).

        The reference updates matrix values in place when called a second time; the values
        it writes are the ones a rebuild produces, so the oracle always rebuilds.
        """
        self.link_exponents = np.asarray(link_exponents)
        self.psi_gradient = gradient_matrix(self.mesh, self.link_exponents)
        if self.fix_psi:
            fixed, free = self.fixed_sites, self.free_rows
        else:
            fixed = free = None
        self.psi_laplacian, self.free_rows = laplacian_matrix(
            self.mesh, self.link_exponents, fixed_sites=fixed, free_rows=free
        )

    def get_supercurrent(self, psi):
        """``J_s,e = Im(conj(psi_i) (grad psi)_e)`` with ``i = e[0]`` (operators.py:385-394)."""
        return (psi.conjugate()[self.edges[:, 0]] * (self.psi_gradient @ psi)).imag
