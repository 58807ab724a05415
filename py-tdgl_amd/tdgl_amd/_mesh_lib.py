"""ctypes binding of libtdgl_mesh.so (include/tdgl_host_mesh.h, include/tdgl_host_amg.h): the host-side set-up
helpers -- Delaunay triangulation, the Voronoi dual mesh, and the two loops of the AMG set-up (Lanczos estimate of
rho(D^-1 A), MIS(2) aggregation) -- in plain C++ (no HIP, no GPU needed).

It is built by ``__graft_entry__.build()`` (g++ only).  `delaunay` / `dual_mesh` raise if the library is
missing; their callers `meshgen.triangulate` and `Mesh.from_triangulation` ask `available()` first and, on a host
where it was never built, warn once and use the constructions it replaced (SciPy's Qhull, the NumPy dual mesh --
the latter bit-identical).  Nothing of the time loop lives here: the HIP library has no substitute.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDGL_MESH_LIB") or os.path.join(_HERE, "lib", "libtdgl_mesh.so")

OK, ERR_ARG, ERR_DEGENERATE, ERR_SKIPPED, ERR_INDEX, ERR_RESOURCES = 0, -1, -2, -3, -4, -5

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)

SIGNATURES = {
    "tdgl_host_delaunay": (C.c_int, [C.c_int64, _f64p, _i64p, _i64p]),
    "tdgl_host_is_delaunay": (C.c_int, [C.c_int64, _f64p, C.c_int64, _i64p]),
    # include/tdgl_host_amg.h
    "tdgl_host_lanczos": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_int, _f64p, C.c_int, _f64p, _f64p,
                                    C.POINTER(C.c_int), _f64p]),
    "tdgl_host_mis2_aggregate": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _i64p, C.c_int, _i64p, _i64p]),
    "tdgl_host_spgemm": (C.c_void_p, [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, C.c_int, _i64p]),
    "tdgl_host_spgemm_take": (C.c_int, [C.c_void_p, _i64p, _i32p, _f64p]),
    "tdgl_host_dual_mesh": (C.c_int, [C.c_int64, _f64p, C.c_int64, _i64p, _i64p, _i64p, _u8p, _i64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _u8p]),
}

_lib = None


class MeshLibraryError(RuntimeError):
    pass


_warned = False


def available() -> bool:
    """Whether libtdgl_mesh.so can be loaded; warns once when it cannot."""
    global _warned
    if _lib is not None or os.path.exists(LIB_PATH):
        return True
    if not _warned:
        import warnings

        warnings.warn(f"{LIB_PATH} not found (build it with `python -c 'import __graft_entry__ as g; g.build()'`): "
                      "mesh set-up falls back to SciPy Qhull / NumPy, several times slower", RuntimeWarning,
                      stacklevel=3)
        _warned = True
    return False


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MeshLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = lib
    return _lib


def _xy(points):
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 2:
        raise ValueError(f"points must have shape (n, 2), got {pts.shape!r}")
    return pts


def delaunay(points):
    """``(status, triangles[t, 3] int64)``, counter-clockwise.  status: OK, ERR_SKIPPED (coinciding points
    were left out; the triangulation of the distinct points is returned), ERR_DEGENERATE (all collinear)."""
    pts = _xy(points)
    n = len(pts)
    if n < 3:
        return ERR_DEGENERATE, np.empty((0, 3), dtype=np.int64)
    out = np.empty(3 * 2 * n, dtype=np.int64)
    nt = C.c_int64(0)
    rc = load().tdgl_host_delaunay(n, pts.ctypes.data_as(_f64p), out.ctypes.data_as(_i64p), C.byref(nt))
    if rc == ERR_ARG:
        raise ValueError("tdgl_host_delaunay: non-finite coordinates")
    if rc == ERR_RESOURCES:
        raise MemoryError("tdgl_host_delaunay: out of memory")
    return rc, out[: 3 * nt.value].reshape(-1, 3).copy()


def is_delaunay(points, triangles) -> bool:
    pts = _xy(points)
    tri = np.ascontiguousarray(triangles, dtype=np.int64)
    rc = load().tdgl_host_is_delaunay(len(pts), pts.ctypes.data_as(_f64p), len(tri), tri.ctypes.data_as(_i64p))
    if rc < 0:
        raise ValueError(f"tdgl_host_is_delaunay: status {rc}")
    return bool(rc)


def dual_mesh(points, triangles):
    """dict(edges[m, 2], is_boundary[m], tri_edge[t, 3], centers[m, 2], directions[m, 2], edge_lengths[m],
    circumcenters[t, 2], dual_lengths[m], areas[n], suspicious[n]) -- see include/tdgl_host_mesh.h."""
    pts = _xy(points)
    tri = np.ascontiguousarray(triangles, dtype=np.int64)
    n, t = len(pts), len(tri)
    edges = np.empty((3 * t, 2), dtype=np.int64)
    is_boundary = np.empty(3 * t, dtype=np.uint8)
    tri_edge = np.empty((t, 3), dtype=np.int64)
    cc = np.empty((t, 2))
    centers = np.empty((3 * t, 2))
    directions = np.empty((3 * t, 2))
    edge_lengths = np.empty(3 * t)
    dual = np.empty(3 * t)
    areas = np.empty(n)
    suspicious = np.empty(n, dtype=np.uint8)
    m = C.c_int64(0)
    rc = load().tdgl_host_dual_mesh(
        n, pts.ctypes.data_as(_f64p), t, tri.ctypes.data_as(_i64p), C.byref(m), edges.ctypes.data_as(_i64p),
        is_boundary.ctypes.data_as(_u8p), tri_edge.ctypes.data_as(_i64p), centers.ctypes.data_as(_f64p),
        directions.ctypes.data_as(_f64p), edge_lengths.ctypes.data_as(_f64p), cc.ctypes.data_as(_f64p),
        dual.ctypes.data_as(_f64p), areas.ctypes.data_as(_f64p), suspicious.ctypes.data_as(_u8p))
    if rc == ERR_INDEX:
        raise IndexError("a triangle refers to a site that does not exist")
    if rc == ERR_RESOURCES:
        raise MemoryError("tdgl_host_dual_mesh: out of memory")
    if rc != OK:
        raise ValueError(f"tdgl_host_dual_mesh: status {rc}")
    m = m.value
    return dict(edges=edges[:m].copy(), is_boundary=is_boundary[:m].astype(bool), tri_edge=tri_edge,
                centers=centers[:m].copy(), directions=directions[:m].copy(), edge_lengths=edge_lengths[:m].copy(),
                circumcenters=cc, dual_lengths=dual[:m].copy(), areas=areas, suspicious=suspicious.astype(bool))


def _csr32(A):
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    return indptr, indices


def lanczos(A, dinv, v0, iters, threads=0):
    """``(alpha[steps], beta[steps], gershgorin)`` of `tdgl_host_lanczos` for the CSR matrix ``A``
    (include/tdgl_host_amg.h)."""
    n = A.shape[0]
    if A.nnz >= 2**31:
        raise ValueError("matrix too large for 32-bit indices")
    indptr, indices = _csr32(A)
    data = np.ascontiguousarray(A.data, dtype=np.float64)
    dinv = np.ascontiguousarray(dinv, dtype=np.float64)
    v0 = np.ascontiguousarray(v0, dtype=np.float64)
    alpha, beta = np.zeros(iters), np.zeros(iters)
    steps, gersh = C.c_int(0), C.c_double(0.0)
    rc = load().tdgl_host_lanczos(n, indptr.ctypes.data_as(_i32p), indices.ctypes.data_as(_i32p), data.ctypes.data_as(_f64p),
                                  dinv.ctypes.data_as(_f64p), int(iters), v0.ctypes.data_as(_f64p), int(threads),
                                  alpha.ctypes.data_as(_f64p), beta.ctypes.data_as(_f64p), C.byref(steps), C.byref(gersh))
    if rc == ERR_RESOURCES:
        raise MemoryError("tdgl_host_lanczos: out of memory or threads")
    if rc != 0:
        raise ValueError(f"tdgl_host_lanczos: status {rc}")
    return alpha[: steps.value], beta[: steps.value], gersh.value


def mis2_aggregate(S, priority, threads=0):
    """``(agg[n] int64, n_agg)`` of `tdgl_host_mis2_aggregate` for the strength graph ``S`` (CSR, no diagonal)."""
    n = S.shape[0]
    if S.nnz >= 2**31:
        raise ValueError("graph too large for 32-bit indices")
    indptr, indices = _csr32(S)
    w = np.ascontiguousarray(np.abs(S.data), dtype=np.float64)
    prio = np.ascontiguousarray(priority, dtype=np.int64)
    agg = np.empty(n, dtype=np.int64)
    n_agg = C.c_int64(0)
    rc = load().tdgl_host_mis2_aggregate(n, indptr.ctypes.data_as(_i32p), indices.ctypes.data_as(_i32p), w.ctypes.data_as(_f64p),
                                         prio.ctypes.data_as(_i64p), int(threads), agg.ctypes.data_as(_i64p), C.byref(n_agg))
    if rc == -2:
        raise RuntimeError("MIS(2) did not terminate")
    if rc == ERR_RESOURCES:
        raise MemoryError("tdgl_host_mis2_aggregate: out of memory or threads")
    if rc != 0:
        raise ValueError(f"tdgl_host_mis2_aggregate: status {rc}")
    return agg, int(n_agg.value)


def spgemm(A, B, threads=0):
    """``A @ B`` for SciPy CSR matrices through `tdgl_host_spgemm` (include/tdgl_host_amg.h): sorted indices, exact
    zeros not stored, the same result on any number of threads."""
    import scipy.sparse as sp

    A = sp.csr_matrix(A)
    B = sp.csr_matrix(B)
    if A.shape[1] != B.shape[0]:
        raise ValueError(f"dimension mismatch: {A.shape} @ {B.shape}")
    if A.nnz >= 2**31 or B.nnz >= 2**31:
        raise ValueError("matrix too large for 32-bit indices")
    rows, cols = A.shape[0], B.shape[1]
    if rows == 0 or cols == 0:
        return sp.csr_matrix((rows, cols))
    ap, ai = _csr32(A)
    bp, bi = _csr32(B)
    ad = np.ascontiguousarray(A.data, dtype=np.float64)
    bd = np.ascontiguousarray(B.data, dtype=np.float64)
    nnz = C.c_int64(0)
    lib = load()
    h = lib.tdgl_host_spgemm(rows, cols, ap.ctypes.data_as(_i32p), ai.ctypes.data_as(_i32p), ad.ctypes.data_as(_f64p),
                             bp.ctypes.data_as(_i32p), bi.ctypes.data_as(_i32p), bd.ctypes.data_as(_f64p), int(threads),
                             C.byref(nnz))
    if not h:
        raise MemoryError("tdgl_host_spgemm: bad arguments, or out of memory / threads")
    indptr = np.empty(rows + 1, dtype=np.int64)
    indices = np.empty(nnz.value, dtype=np.int32)
    data = np.empty(nnz.value, dtype=np.float64)
    lib.tdgl_host_spgemm_take(h, indptr.ctypes.data_as(_i64p), indices.ctypes.data_as(_i32p), data.ctypes.data_as(_f64p))
    if nnz.value < 2**31:
        indptr = indptr.astype(np.int32)
    out = sp.csr_matrix((data, indices, indptr), shape=(rows, cols))
    out.has_sorted_indices = True
    return out
