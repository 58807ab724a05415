"""Smoothed-aggregation AMG hierarchy SETUP for the mu Poisson problem (host side).

The reference factorises the (singular, pure-Neumann) mu Laplacian once with SuperLU and
back-substitutes every step (`tdgl/finite_volume/operators.py:305-308`,
`tdgl/solver/solver.py:516`).  A sparse direct solve does not map to the GPU; the HIP path
solves the symmetrised system

    A mu = b,   A = -diag(a) L_mu  (symmetric positive semi-definite, null space = constants),
                b = -a * rhs

with conjugate gradients preconditioned by one AMG V-cycle.  This module is the analogue of
the factorisation: it runs once per mesh on the host and produces the level matrices the
HIP kernels consume (`csrc/poisson.hip`).  The cycle itself never runs on the host in the
product path.

Algorithm (Vanek/Mandel/Brezina smoothed aggregation with the constant as the only
near-null-space vector):

* strength graph: all off-diagonal couplings with ``|a_ij| >= theta*sqrt(a_ii a_jj)``;
* aggregation: roots = a distance-2 maximal independent set (parallel Luby rounds with
  hashed priorities, deterministic); every other node joins the root reached through its
  strongest connection (distance 1 first, then distance 2);
* tentative prolongator T = aggregate indicator (unnormalised, so T 1_c = 1 and the coarse
  null space is again the constant vector);
* P = (I - omega/rho(D^-1 A) * D^-1 A) T,  A_c = P^T A P;
* recursion until n_c <= max_coarse; the coarsest operator gets a dense pseudo-inverse.
"""

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.sparse as sp


def _mm(A, B):
    """Sparse ``A @ B``: `tdgl_host_spgemm` (threaded, include/tdgl_host_amg.h) for the large products of the set-up,
    SciPy for small ones.  Both give the same entries (SciPy's come out unsorted; they are sorted here as well)."""
    if sp.issparse(A) and sp.issparse(B) and A.nnz + B.nnz > 200_000 and max(A.nnz, B.nnz) < 2**31:
        from . import _mesh_lib

        return _mesh_lib.spgemm(A, B)
    C = (A @ B).tocsr()
    C.sort_indices()
    return C


def _rowmax(indptr, vals, empty=-np.inf):
    """Row-wise max of CSR-ordered ``vals`` (rows may be empty)."""
    n = len(indptr) - 1
    out = np.full(n, empty, dtype=vals.dtype)
    nonempty = indptr[1:] > indptr[:-1]
    if len(vals):
        red = np.maximum.reduceat(vals, indptr[:-1][nonempty])
        out[nonempty] = red
    return out


def _hash_priority(n, level_seed):
    """Deterministic pseudo-random distinct priorities (a permutation of 1..n)."""
    rng = np.random.default_rng(12345 + level_seed)
    return (rng.permutation(n) + 1).astype(np.int64)


def mis2_aggregate(S: sp.csr_matrix, seed: int = 0, backend: str = "native"):
    """Aggregate the nodes of the symmetric strength graph ``S`` (CSR, no diagonal).

    Returns ``(agg[n] int64, n_agg)``.  ``backend="native"``: `tdgl_host_mis2_aggregate` (include/tdgl_host_amg.h,
    threaded C++), ``"numpy"``: the construction it replaced -- the same aggregates, node for node
    (`tests/test_host_logic.py`).
    """
    if backend == "native":
        from . import _mesh_lib

        return _mesh_lib.mis2_aggregate(S, _hash_priority(S.shape[0], seed))
    if backend != "numpy":
        raise ValueError(f"unknown backend {backend!r}")
    n = S.shape[0]
    indptr, indices = S.indptr, S.indices
    prio = _hash_priority(n, seed)
    # state: 0 undecided, 1 root, -1 excluded (within distance 2 of a root)
    state = np.zeros(n, dtype=np.int8)
    isolated = indptr[1:] == indptr[:-1]
    state[isolated] = 1  # isolated nodes are their own aggregates
    for _ in range(200):
        und = state == 0
        if not und.any():
            break
        p = np.where(und, prio, 0)
        m1 = np.maximum(p, _rowmax(indptr, p[indices], empty=0))
        m2 = np.maximum(m1, _rowmax(indptr, m1[indices], empty=0))
        new_root = und & (p == m2)
        state[new_root] = 1
        r = new_root.astype(np.int8)
        d1 = np.maximum(r, _rowmax(indptr, r[indices], empty=0))
        d2 = np.maximum(d1, _rowmax(indptr, d1[indices], empty=0))
        state[(d2 > 0) & (state == 0)] = -1
    else:  # pragma: no cover
        raise RuntimeError("MIS(2) did not terminate")
    roots = np.flatnonzero(state == 1)
    n_agg = len(roots)
    agg = np.full(n, -1, dtype=np.int64)
    agg[roots] = np.arange(n_agg)
    w = np.abs(S.data)
    # distance-1: join the adjacent root with the strongest coupling
    for _pass in range(2):
        nbr_agg = agg[indices]
        score = np.where(nbr_agg >= 0, w, -1.0)
        best = _rowmax(indptr, score, empty=-1.0)
        todo = (agg < 0) & (best >= 0)
        if not todo.any():
            break
        # position of the best entry in each row
        row_of = np.repeat(np.arange(n), np.diff(indptr))
        is_best = (score == best[row_of]) & todo[row_of]
        pos = np.flatnonzero(is_best)
        # first best per row
        first = np.ones(len(pos), dtype=bool)
        first[1:] = row_of[pos][1:] != row_of[pos][:-1]
        pos = pos[first]
        new_agg = agg.copy()
        new_agg[row_of[pos]] = nbr_agg[pos]
        agg = new_agg
    left = np.flatnonzero(agg < 0)
    if len(left):  # cannot happen for a maximal MIS(2); keep robust
        agg[left] = n_agg + np.arange(len(left))
        n_agg += len(left)
    return agg, n_agg


def estimate_rho_DinvA(A: sp.csr_matrix, dinv: np.ndarray, iters: int = 24, seed: int = 0, backend: str = "native"):
    """Estimate (from above, in practice) of the largest eigenvalue of D^-1 A: ``iters`` Lanczos steps
    on the symmetrised operator D^-1/2 A D^-1/2, largest Ritz value plus the norm of its residual,
    capped by the Gershgorin bound.  Not a guaranteed bound: theta + r only guarantees an eigenvalue
    within r of theta, and without re-orthogonalisation a ghost Ritz value can shrink r; the build adds
    5 % (`build_hierarchy`) and `tests/test_host_logic.py` holds estimate x 1.05 >= the true value
    (ARPACK) on the graded and the quasi-uniform test meshes for several start vectors (measured: the
    raw estimate is 0.02-0.4 % above).

    The Chebyshev smoother diverges on eigenvalues above its upper bound, so a plain power iteration
    (which converges from below, slowly) is not safe; a fixed, small number of Lanczos steps with the
    residual added is (the classical recipe of AMG codes), and costs ``iters`` matrix-vector products
    instead of the ~200 a converged ARPACK run takes (the largest single item of the set-up at 1M
    sites).

    ``backend="native"`` runs the Lanczos recurrence in `tdgl_host_lanczos` (include/tdgl_host_amg.h; threaded, dot
    products summed block-wise in a fixed order: the same value on any number of threads), ``"numpy"`` the loop it
    replaced (BLAS dot products: the two agree to round-off, `tests/test_host_logic.py`)."""
    n = A.shape[0]
    rng = np.random.default_rng(seed)
    sq = np.sqrt(dinv)
    if n < 50:
        gersh = float((abs(A) @ np.ones(n) * dinv).max())
        S = (A.toarray() * sq[:, None]) * sq[None, :]
        return min(gersh, float(np.linalg.eigvalsh(S)[-1]))
    m = int(min(iters, n - 1))
    v = rng.standard_normal(n)
    v /= np.linalg.norm(v)
    if backend == "native":
        from . import _mesh_lib

        A = sp.csr_matrix(A)
        alpha, beta, gersh = _mesh_lib.lanczos(A, dinv, v, m)
        steps = len(alpha)
        T = np.diag(alpha) + np.diag(beta[:steps - 1], 1) + np.diag(beta[:steps - 1], -1)
        theta, Y = np.linalg.eigh(T)
        return min(gersh, float(theta[-1] + abs(beta[steps - 1] * Y[-1, -1])))
    if backend != "numpy":
        raise ValueError(f"unknown backend {backend!r}")
    # Gershgorin: never exceeded, and tight (= 2) for the fine-level M-matrix
    gersh = float((abs(A) @ np.ones(n) * dinv).max())
    v_prev = np.zeros(n)
    alpha, beta = np.zeros(m), np.zeros(m)
    b_prev = 0.0
    steps = m
    for j in range(m):
        w = sq * (A @ (sq * v)) - b_prev * v_prev
        alpha[j] = float(w @ v)
        w -= alpha[j] * v
        beta[j] = float(np.linalg.norm(w))
        if beta[j] <= 1e-12 * max(abs(alpha[j]), 1.0):  # invariant subspace: the Ritz values are exact
            steps = j + 1
            beta[j] = 0.0
            break
        v_prev, v, b_prev = v, w / beta[j], beta[j]
    T = np.diag(alpha[:steps]) + np.diag(beta[:steps - 1], 1) + np.diag(beta[:steps - 1], -1)
    theta, Y = np.linalg.eigh(T)
    lam = float(theta[-1] + abs(beta[steps - 1] * Y[-1, -1]))  # Ritz value + residual norm of its pair
    return min(gersh, lam)


@dataclass
class Level:
    A: sp.csr_matrix
    dinv: np.ndarray
    rho: float
    P: Optional[sp.csr_matrix] = None  # n_l x n_{l+1}
    R: Optional[sp.csr_matrix] = None  # n_{l+1} x n_l  (= P^T, stored CSR)
    agg: Optional[np.ndarray] = None
    owner: Optional[np.ndarray] = None  # rank that owns each row (build_hierarchy(part=...): the distributed levels)


@dataclass
class Hierarchy:
    levels: List[Level] = field(default_factory=list)
    coarse_pinv: Optional[np.ndarray] = None  # dense pseudo-inverse of the last level

    @property
    def sizes(self):
        return [lv.A.shape[0] for lv in self.levels]

    @property
    def operator_complexity(self):
        return sum(lv.A.nnz for lv in self.levels) / self.levels[0].A.nnz


def build_hierarchy(
    A: sp.spmatrix,
    max_coarse: int = 600,
    max_levels: int = 12,
    theta: float = 0.0,
    omega: float = 1.45,
    part: Optional[np.ndarray] = None,
    part_levels: int = 1,
    seed: int = 0,
) -> Hierarchy:
    """Build the SA-AMG hierarchy for a symmetric positive semi-definite ``A`` whose null
    space is the constant vector.

    ``part`` (one-process-per-GPU runs): the rank that owns every fine site.  On the first ``part_levels``
    levels the aggregation then ignores couplings between sites of different ranks, so every aggregate -- every
    row of the next level -- lies inside one rank and has one owner (`Level.owner`); the prolongator smoothing
    still uses the whole matrix, so the hierarchy remains a plain smoothed-aggregation one of the GLOBAL operator.

    ``omega``: the prolongator smoothing's damping, ``P = (I - omega / rho D^-1 A) T``.  The textbook value is 4/3
    with the exact spectral radius; ``rho`` here is an estimate with 5 % of safety on top, and on the square films of
    250k and 1M sites and the 500k-site strip the PCG's contraction per iteration is smallest between 1.45 and 1.5
    (0.3012 -> 0.2953 at 1M sites; same sparsity pattern: the iteration costs the same) -- tools/exp_precond_sweep.py.

    ``seed``: of the hashed priorities that decide the MIS(2) roots.  The aggregates they give differ in quality by
    luck: at 1M sites the PCG's convergence factor moves between 0.286 and 0.308 over six seeds, 7.50 - 7.79 iterations
    per step in the time loop (`TDGLContext.build_poisson` builds a few candidates and keeps the best)."""
    A = sp.csr_matrix(A, dtype=float)
    A.sum_duplicates()
    A.sort_indices()
    h = Hierarchy()
    for lvl in range(max_levels):
        n = A.shape[0]
        diag = A.diagonal()
        dinv = np.where(diag > 0, 1.0 / np.where(diag > 0, diag, 1.0), 0.0)
        rho = 1.05 * estimate_rho_DinvA(A, dinv, seed=lvl)
        level = Level(A=A, dinv=dinv, rho=rho)
        if part is not None:
            level.owner = np.asarray(part, dtype=np.int32)
        h.levels.append(level)
        if n <= max_coarse or lvl == max_levels - 1:
            break
        # strength graph
        C = A.tocoo()
        off = C.row != C.col
        keep = off & (np.abs(C.data) >= theta * np.sqrt(np.abs(diag[C.row] * diag[C.col])))
        keep &= C.data != 0
        if part is not None and lvl < part_levels:
            keep &= part[C.row] == part[C.col]
        S = sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=A.shape)
        S.sort_indices()
        agg, n_agg = mis2_aggregate(S, seed=lvl + 100 * int(seed))
        if n_agg >= n:  # no coarsening possible
            break
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, n_agg))
        if part is not None:
            if lvl < part_levels:  # every member of an aggregate has the same owner
                nxt = np.zeros(n_agg, dtype=np.int32)
                nxt[agg] = part
                assert (nxt[agg] == part).all()
                part = nxt
            else:
                part = None
        DinvA = sp.diags(dinv) @ A
        P = (T - (omega / rho) * _mm(DinvA, T)).tocsr()
        P.sort_indices()
        R = P.T.tocsr()
        R.sort_indices()
        level.P, level.R, level.agg = P, R, agg
        A = _mm(R, _mm(A, P))
        A.sum_duplicates()
        A.sort_indices()
        # symmetrise round-off
        A = ((A + A.T) * 0.5).tocsr()
        A.sort_indices()
    last = h.levels[-1].A.toarray()
    nc = last.shape[0]
    # pseudo-inverse on the complement of the constant vector
    J = np.full((nc, nc), 1.0 / nc)
    h.coarse_pinv = np.linalg.inv(last + J) - J
    return h


# ---------------------------------------------------------------------------------------
# Host-side reference application of the cycle.  Used by the CPU tests to validate the
# hierarchy and to cross-check the HIP V-cycle; the product path never calls it.
def smoother_coefficients(rho, nu=2, smoother="chebyshev", cheb_lo=0.1):
    """(c1[k], c2[k]) of the smoothing recurrence d <- c1 d + c2 D^-1 (b - A x), x <- x + d
    (same numbers as `smoother_coef` in csrc/poisson.inc)."""
    if smoother == "jacobi":
        return [0.0] * nu, [(4.0 / 3.0) / rho] * nu
    hi, lo = rho, cheb_lo * rho
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta
    rho_k = 1.0 / sigma
    c1, c2 = [0.0], [1.0 / theta]
    for _ in range(1, nu):
        rho_new = 1.0 / (2.0 * sigma - rho_k)
        c1.append(rho_new * rho_k)
        c2.append(2.0 * rho_new / delta)
        rho_k = rho_new
    return c1, c2


def fused_restriction(h: Hierarchy, c: float) -> sp.csr_matrix:
    """``R_0 (I - c A_0 D_0^-1)``: restriction of the residual left by ONE smoothing step
    ``x = c D^-1 b`` from a zero guess, as a single operator on ``b``."""
    lv = h.levels[0]
    return fused_restriction_from(lv.A, lv.R, lv.dinv, c)


def fused_restriction_from(A, R, dinv, c: float) -> sp.csr_matrix:
    """The same from explicit pieces.  ``A`` may be a rank's slice [n_own x n_loc] (owned rows,
    owned + ghost columns) with ``R`` [n_c x n_own] the restriction to its owned fine columns and
    ``dinv`` over all local sites: the result [n_c x n_loc] then gives this rank's PARTIAL coarse
    right-hand side, gathering the residual at ghost columns too."""
    A, R = sp.csr_matrix(A), sp.csr_matrix(R)
    n_own, n_loc = A.shape
    Rp = sp.hstack([R, sp.csr_matrix((R.shape[0], n_loc - n_own))]).tocsr() if n_loc > n_own else R
    M = (Rp - _mm(R, A) @ sp.diags(c * np.asarray(dinv))).tocsr()
    M.sort_indices()
    return M


def fused_level_operators(level):
    """``(R A, A P, P on the pattern of A P)`` of a coarse level as CSR matrices / value array
    (`tdgl_poisson_set_fused_level`)."""
    A, P, R = level.A.tocsr(), level.P.tocsr(), level.R.tocsr()
    RA = _mm(R, A)
    RA.sort_indices()
    AP = _mm(A, P)
    # union pattern (P's pattern is contained in A P's whenever diag(A) != 0; do not rely on it),
    # then both value sets laid out on it (SciPy's sparse sum drops explicit zeros, so by hand)
    def keys(M):
        M = M.tocsr()
        M.sort_indices()
        rows = np.repeat(np.arange(M.shape[0], dtype=np.int64), np.diff(M.indptr))
        return rows * M.shape[1] + M.indices, M.data

    k_ap, v_ap = keys(AP)
    k_p, v_p = keys(P)
    k_u = np.union1d(k_ap, k_p)
    ap_vals = np.zeros(len(k_u))
    p_vals = np.zeros(len(k_u))
    ap_vals[np.searchsorted(k_u, k_ap)] = v_ap
    p_vals[np.searchsorted(k_u, k_p)] = v_p
    rows_u, cols_u = k_u // AP.shape[1], k_u % AP.shape[1]
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows_u, minlength=AP.shape[0]))])
    AP_u = sp.csr_matrix((ap_vals, cols_u, indptr), shape=AP.shape)
    return RA, AP_u, p_vals


def vcycle_host(h: Hierarchy, b: np.ndarray, nu: int = 2, smoother: str = "chebyshev",
                cheb_lo: float = 0.1, lvl: int = 0, nu_fine: int = 0) -> np.ndarray:
    """``nu_fine`` > 0 overrides the smoother degree on level 0 (the library's default is
    degree 1 on level 0, degree 2 below)."""
    level = h.levels[lvl]
    if lvl == len(h.levels) - 1:
        return h.coarse_pinv @ b
    A, dinv = level.A, level.dinv
    nu_all = nu
    if lvl == 0 and nu_fine > 0:
        nu = nu_fine
    c1, c2 = smoother_coefficients(level.rho, nu, smoother, cheb_lo)
    d = c2[0] * dinv * b
    x = d.copy()
    for k in range(1, nu):
        d = c1[k] * d + c2[k] * dinv * (b - A @ x)
        x = x + d
    r = b - A @ x
    xc = vcycle_host(h, level.R @ r, nu_all, smoother, cheb_lo, lvl + 1)
    x = x + level.P @ xc
    for k in range(nu):
        d = c1[k] * d + c2[k] * dinv * (b - A @ x)
        x = x + d
    return x


def pcg_host(A, b, h: Hierarchy, x0=None, rtol=1e-10, maxiter=200, nu=2, smoother="chebyshev", nu_fine=1):
    """Preconditioned CG on the semi-definite system (b is projected onto range(A))."""
    n = len(b)
    b = b - b.mean()
    x = np.zeros(n) if x0 is None else x0 - np.mean(x0)
    r = b - A @ x
    bnorm = np.linalg.norm(b)
    if bnorm == 0:
        return x, 0, 0.0
    z = vcycle_host(h, r, nu, smoother, nu_fine=nu_fine)
    p = z.copy()
    rz = r @ z
    res = np.linalg.norm(r) / bnorm
    it = 0
    while res > rtol and it < maxiter:
        q = A @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        res = np.linalg.norm(r) / bnorm
        it += 1
        if res <= rtol:
            break
        z = vcycle_host(h, r, nu, smoother, nu_fine=nu_fine)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x - x.mean(), it, res


# ---------------------------------------------------------------------------------------
# Collapsed coarse levels.  The levels below level 0 move a few MB per kernel and are bound by
# kernel boundaries and dependent memory round trips, not by bandwidth.
# Everything here is the SAME V-cycle re-associated into fewer, denser operators, built once on
# the host (the counterpart of the reference's one-off LU factorisation, operators.py:305-308):
#
#   * an intermediate level k:  x_k = S_k b_k is the fused two-step pre-smoothing (unchanged), but
#     the next right-hand side is b_{k+1} = M_k b_k with M_k = R_k (I - A_k S_k) explicit, so the
#     restriction no longer waits for x_k: one launch computes both;
#   * the tail: the first level t with at most `tail_rows` rows.  The whole cycle below it is the
#     dense matrix B_{t+1} (pseudo-inverse of the coarsest operator, or an explicitly formed small
#     cycle).  Then  e_{t+1} = G b_t,  G = B_{t+1} M_t  (dense [n_{t+1}, n_t]), and the post-smoothed
#     result is  e_t = W [b_t ; e_{t+1}] + V e_{t+1}[:g1]  with a sparse W over the concatenated
#     vector (plain cycle: W = [T_x S_t + T_b | T_x P_t], no dense part) and, with two folded cycles,
#     a dense block V for the part that is not sparse.  y = T_x x' + T_b b is the two-step
#     post-smoothing as one operator: two launches for everything from level t down.  If level t itself is small (<= dense_rows) its cycle is formed
#     densely, B_t, and applied in one launch.
def smoothing_operators(A, dinv, rho, nu=2, smoother="chebyshev", cheb_lo=0.1):
    """``(S, T_x, T_b)``: pre-smoothing from a zero guess ``x = S b`` and post-smoothing
    ``y = T_x x' + T_b b`` of `vcycle_host` as explicit sparse operators."""
    c1, c2 = smoother_coefficients(rho, nu, smoother, cheb_lo)
    n = A.shape[0]
    D = sp.diags(dinv)
    DA = (D @ A).tocsr()
    eye = sp.identity(n, format="csr")
    # pre: d = c2[0] D b, x = d; then d = c1[k] d + c2[k] D (b - A x), x += d   (as operators on b)
    Dop, Xop = c2[0] * D, c2[0] * D
    for k in range(1, nu):
        Dop = c1[k] * Dop + c2[k] * (D - _mm(DA, Xop))
        Xop = Xop + Dop
    S = sp.csr_matrix(Xop)
    # post: the polynomial restarts from x' with d = 0 (c1[0] = 0)
    Dx, Db = sp.csr_matrix((n, n)), sp.csr_matrix((n, n))
    Xx, Xb = eye, sp.csr_matrix((n, n))
    for k in range(nu):
        Dx = c1[k] * Dx - c2[k] * _mm(DA, Xx)
        Db = c1[k] * Db + c2[k] * (D - _mm(DA, Xb))
        Xx, Xb = Xx + Dx, Xb + Db
    Tx, Tb = sp.csr_matrix(Xx), sp.csr_matrix(Xb)
    for M in (S, Tx, Tb):
        M.sort_indices()
    return S, Tx, Tb


def dense_cycle(h: Hierarchy, lvl: int, nu=2, smoother="chebyshev", cheb_lo=0.1) -> np.ndarray:
    """The V-cycle from level ``lvl`` down as a dense matrix (small levels only)."""
    if lvl == len(h.levels) - 1:
        return np.asarray(h.coarse_pinv)
    lv = h.levels[lvl]
    S, Tx, Tb = smoothing_operators(lv.A, lv.dinv, lv.rho, nu, smoother, cheb_lo)
    S, Tx, Tb = S.toarray(), Tx.toarray(), Tb.toarray()
    A, P, R = lv.A.toarray(), lv.P.toarray(), lv.R.toarray()
    Bc = dense_cycle(h, lvl + 1, nu, smoother, cheb_lo)
    n = A.shape[0]
    return Tx @ (S + P @ (Bc @ (R @ (np.eye(n) - A @ S)))) + Tb


def exact_pinv(A) -> np.ndarray:
    """Dense pseudo-inverse of a symmetric positive semi-definite operator whose null space is the
    constant vector (the same construction as `Hierarchy.coarse_pinv`)."""
    M = A.toarray() if sp.issparse(A) else np.asarray(A)
    n = M.shape[0]
    J = np.full((n, n), 1.0 / n)
    return np.linalg.inv(M + J) - J


def dense_pseudo_inverse(A: sp.spmatrix, check_rtol: float = 1e-11, seed: int = 0):
    """``pinv(A)`` of the level-0 Poisson matrix as a dense symmetric array, for the direct solve
    of small meshes (`tdgl_poisson_set_dense_inverse`): the counterpart of the reference's LU
    factorisation (operators.py:305-308) in a form whose application is one dense matrix-vector
    product.  ``A`` is symmetric positive semi-definite with the constants as its null space, so
    ``A + (s/n) 1 1^T`` is positive definite (s = mean diagonal: the added eigenvalue sits inside the
    spectrum), its Cholesky inverse is ``pinv(A) + 1 1^T / (s n)``, and the rank-one term is removed
    again.  Returns ``None`` when the factorisation fails (a mesh of several disconnected pieces has
    a larger null space) or when ``||A G b - b|| > check_rtol ||b||`` for a random zero-mean ``b``
    (ill conditioned: stay with the iterative solver, which controls its residual)."""
    from scipy.linalg import lapack

    n = A.shape[0]
    s = float(A.diagonal().mean())
    if not (s > 0.0) or not np.isfinite(s):
        return None
    M = A.toarray(order="F")
    M += s / n
    c, info = lapack.dpotrf(M, lower=1, overwrite_a=1)
    if info != 0:
        return None
    G, info = lapack.dpotri(c, lower=1, overwrite_c=1)
    if info != 0:
        return None
    # dpotri fills the lower triangle only: mirror it (the result is exactly symmetric)
    il = np.tril_indices(n, -1)
    G.T[il] = G[il]
    G -= 1.0 / (s * n)
    G = np.ascontiguousarray(G)
    b = np.random.default_rng(seed).standard_normal(n)
    b -= b.mean()
    if not np.linalg.norm(A @ (G @ b) - b) <= check_rtol * np.linalg.norm(b):
        return None
    return G


def _drop_small_symmetric(W: sp.csr_matrix, tol: float) -> sp.csr_matrix:
    """``W`` without the off-diagonal entries below ``tol * sqrt(|w_ii w_jj|)`` (symmetric criterion)."""
    C = W.tocoo()
    dg = np.abs(W.diagonal())
    keep = (np.abs(C.data) >= tol * np.sqrt(dg[C.row] * dg[C.col])) | (C.row == C.col)
    out = sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=W.shape)
    out.sort_indices()
    return out


def collapsed_operators(h: Hierarchy, nu=2, smoother="chebyshev", cheb_lo=0.1, tail_rows=8192, dense_rows=1536,
                        tail_cycles=2, drop_tol=1e-3, mid_up=True, mid_drop_tol=3e-3):
    """Plan of the collapsed coarse chain: ``dict(mid={k: M_k}, tail=t, mode="dense"|"gwv", ...)`` or
    ``None`` when the hierarchy has no intermediate level to collapse.

    ``tail_cycles``: how accurately the tail level is solved.  Applying explicit operators costs the
    same two launches whatever they contain, so the tail can afford more than one V-cycle:
    1 = the plain cycle (bit-for-bit the re-association of `vcycle_host`); 2 (default) = two cycles,
    ``B' = B (2 I - A B)``, folded into operators of the same shape (dense mode: the exact
    pseudo-inverse).  With the 10-18x coarsening per level used here the plain V-cycle's inexact
    coarse solves cost ~20 % more PCG iterations than a near-exact tail.

    ``drop_tol``: the folded ``W' = 2W - WAW`` reaches 9 hops, but most of its entries are tiny.
    Entries with ``|w_ij| < drop_tol * sqrt(|w_ii w_jj|)`` are dropped -- a symmetric criterion on a
    symmetric matrix, so the preconditioner stays symmetric; at 1M sites 1e-3 keeps 28 % of the
    entries, leaves the spectrum of ``B'`` unchanged to three digits (smallest eigenvalue 0.0889 ->
    0.0889) and the PCG iteration count unchanged.  Not applied with ``tail_cycles = 1`` (which is
    the plain cycle bit for bit)."""
    L = len(h.levels)
    if L < 3:
        return None
    sizes = h.sizes
    t = None
    for k in range(1, L - 1):
        if sizes[k] <= dense_rows or (sizes[k] <= tail_rows and sizes[k + 1] <= dense_rows):
            t = k
            break
    if t is None:
        return None
    plan = dict(tail=t, mid={}, up={}, coef=(nu, smoother, cheb_lo), tail_cycles=int(tail_cycles))
    for k in range(1, t):
        lv = h.levels[k]
        S, Tx, Tb = smoothing_operators(lv.A, lv.dinv, lv.rho, nu, smoother, cheb_lo)
        M = _mm(lv.R, (sp.identity(sizes[k], format="csr") - _mm(lv.A, S)).tocsr())
        plan["mid"][k] = M
        if mid_up:
            # the way up of this level as explicit operators: e = (T_x S + T_b) b + (T_x P) e_next.  W is
            # a degree-(2 nu) polynomial in A (77 entries per row on level 1 at 1M sites); like the
            # tail's W' it sheds its small entries when the plan is not asked to reproduce the plain
            # cycle bit for bit (tail_cycles >= 2): 3e-3 keeps 46 per row at the same PCG iteration count
            # and convergence factor (0.303), 1e-2 38 per row at 0.304.  V = M^T is kept whole.
            W = (_mm(Tx, S) + Tb).tocsr()
            W = ((W + W.T) * 0.5).tocsr()  # (round-off)
            if tail_cycles >= 2 and mid_drop_tol > 0:
                W = _drop_small_symmetric(W, mid_drop_tol)
            W.sort_indices()
            V = _mm(Tx, lv.P)
            plan["up"][k] = (W, V)
    lv = h.levels[t]
    if sizes[t] <= dense_rows:
        plan["mode"] = "dense"
        plan["B"] = np.ascontiguousarray(
            dense_cycle(h, t, nu, smoother, cheb_lo) if tail_cycles <= 1 else exact_pinv(lv.A))
        return plan
    S, Tx, Tb = smoothing_operators(lv.A, lv.dinv, lv.rho, nu, smoother, cheb_lo)
    M = _mm(lv.R, (sp.identity(sizes[t], format="csr") - _mm(lv.A, S)).tocsr())
    Bc = dense_cycle(h, t + 1, nu, smoother, cheb_lo)
    plan["mode"] = "gwv"
    G = Bc @ M.toarray()                                     # [n_{t+1}, n_t]
    W = (_mm(Tx, S) + Tb).tocsr()
    Vs = _mm(Tx, lv.P)                                 # [n_t, n_{t+1}], a few entries per row
    V = Vs.toarray()
    nt = sizes[t]
    if tail_cycles < 2:
        # e = W b + V (G b): both operators sparse -> one sparse operator on [b ; G b], no dense block
        Wc = sp.hstack([W, Vs]).tocsr()
        Wc.sort_indices()
        plan["G"] = np.ascontiguousarray(G)
        plan["W"], plan["V"] = Wc, np.zeros((nt, 0))
        return plan
    if tail_cycles >= 2:
        # B = W + V G;  B' = 2 B - B A B = W' + [V1 | -V] [G ; H]  with
        #   W' = 2 W - W A W,  K = G A V,  H = G A W,  V1 = 2 V - W A V - V K
        A = lv.A.tocsr()
        AW = _mm(A, W)
        H = np.asarray(G @ AW.toarray()) if AW.shape[0] <= 2048 else np.asarray((AW.T @ G.T).T)
        AV = A @ V
        K = G @ AV
        V1 = 2.0 * V - W @ AV - V @ K
        W = (2.0 * W - _mm(W, AW)).tocsr()
        W = ((W + W.T) * 0.5).tocsr()  # (round-off)
        if drop_tol > 0:
            C = W.tocoo()
            dg = np.abs(W.diagonal())
            keep = (np.abs(C.data) >= drop_tol * np.sqrt(dg[C.row] * dg[C.col])) | (C.row == C.col)
            W = sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=W.shape)
        # e = W' b + V1 (G b) - V (H b):  y = [G ; H] b,  the sparse V rides in the sparse operator
        # (columns n_t + g1 ...), only V1 is dense
        g1 = G.shape[0]
        G = np.vstack([G, H])
        W = sp.hstack([W, sp.csr_matrix((nt, g1)), -Vs]).tocsr()
        V = V1
    W.sort_indices()
    plan["G"] = np.ascontiguousarray(G)
    plan["W"], plan["V"] = W, np.ascontiguousarray(V)
    return plan


def vcycle_collapsed_host(h: Hierarchy, plan, b: np.ndarray, nu=2, smoother="chebyshev", cheb_lo=0.1,
                          nu_fine=0) -> np.ndarray:
    """`vcycle_host` evaluated through the collapsed operators (test helper: must agree with the
    plain cycle to round-off)."""
    def level(k, bk):
        lv = h.levels[k]
        if k == plan["tail"]:
            if plan["mode"] == "dense":
                return plan["B"] @ bk
            y = plan["G"] @ bk
            return plan["W"] @ np.concatenate([bk, y]) + plan["V"] @ y[: plan["V"].shape[1]]
        n_here = nu_fine if (k == 0 and nu_fine > 0) else nu
        c1, c2 = smoother_coefficients(lv.rho, n_here, smoother, cheb_lo)
        d = c2[0] * lv.dinv * bk
        x = d.copy()
        for s in range(1, n_here):
            d = c1[s] * d + c2[s] * lv.dinv * (bk - lv.A @ x)
            x = x + d
        if k in plan.get("up", {}):  # two explicit operators: no smoothing steps of its own
            W, V = plan["up"][k]
            return W @ bk + V @ level(k + 1, plan["mid"][k] @ bk)
        bc = plan["mid"][k] @ bk if k in plan["mid"] else lv.R @ (bk - lv.A @ x)
        x = x + lv.P @ level(k + 1, bc)
        for s in range(n_here):
            d = c1[s] * d + c2[s] * lv.dinv * (bk - lv.A @ x)
            x = x + d
        return x

    return level(0, b)
