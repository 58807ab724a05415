"""Finite-volume mesh containers (Delaunay primal mesh + Voronoi dual), vectorised.

Same attribute surface as the reference's ``tdgl.finite_volume.Mesh`` /
``EdgeMesh`` (`tdgl/finite_volume/mesh.py:23-151`, `tdgl/finite_volume/edge_mesh.py:9-92`),
because these arrays are the input layout of the HIP kernels:

    sites[n,2], elements[t,3], boundary_indices,
    areas[n]                      Voronoi cell areas
    edge_mesh.edges[m,2]          lexicographically sorted (i<j) unique site pairs
    edge_mesh.boundary_edge_indices, .centers, .directions (r_j - r_i, un-normalised),
    .normalized_directions, .edge_lengths, .dual_edge_lengths

The reference builds the dual mesh with per-edge and per-site Python loops
(`tdgl/finite_volume/util.py:59-97,169-255`; 133 s at 1M sites).  Here everything is
array arithmetic (~seconds at 1M sites).  Definitions reproduced:

* edges: `util.py:15-28` (sorted pairs, `np.unique(axis=0)`, boundary = seen once);
* circumcentres: `util.py:100-124`;
* dual edge length: distance between the two adjacent circumcentres, or circumcentre to
  edge midpoint on the boundary (`util.py:88-96`);
* Voronoi areas: area of the polygon of adjacent circumcentres; boundary cells are closed
  by the two boundary-edge midpoints and the site itself (`util.py:204-253`).  For a
  Delaunay mesh this equals the sum of the signed "kites" (site, edge midpoint,
  circumcentre) used below; `tests/test_mesh.py` checks it against reference output.
"""

from typing import Optional

import numpy as np


def unique_edges(elements: np.ndarray, num_sites: int):
    """Sorted unique edges of a triangulation.

    Returns ``(edges[m,2], is_boundary[m], tri_edge[t,3])`` where ``tri_edge[t,k]`` is the
    edge index of local edge ``k`` (vertex pairs (0,1), (1,2), (2,0)) of triangle ``t``.
    """
    elements = np.asarray(elements, dtype=np.int64)
    a = np.concatenate([elements[:, 0], elements[:, 1], elements[:, 2]])
    b = np.concatenate([elements[:, 1], elements[:, 2], elements[:, 0]])
    lo = np.minimum(a, b)
    hi = np.maximum(a, b)
    key = lo * np.int64(num_sites) + hi
    ukey, inverse, counts = np.unique(key, return_inverse=True, return_counts=True)
    edges = np.column_stack([ukey // num_sites, ukey % num_sites]).astype(np.int64)
    tri_edge = inverse.reshape(3, -1).T.copy()
    return edges, counts == 1, tri_edge


def circumcenters(sites: np.ndarray, elements: np.ndarray) -> np.ndarray:
    """Circumcentre of every triangle (Voronoi vertices of a Delaunay mesh)."""
    p0 = sites[elements[:, 0]]
    u = sites[elements[:, 1]] - p0
    v = sites[elements[:, 2]] - p0
    uu = (u**2).sum(axis=1)
    vv = (v**2).sum(axis=1)
    det = 2 * u[:, 0] * v[:, 1] - 2 * u[:, 1] * v[:, 0]
    cx = (v[:, 1] * uu - u[:, 1] * vv) / det
    cy = (u[:, 0] * vv - v[:, 0] * uu) / det
    return np.column_stack([cx, cy]) + p0


def _reference_hull_areas(areas, which, sites, elements, cc, edges, boundary_edge_indices, boundary_indices):
    """Cell areas the way the reference builds them (`tdgl/finite_volume/util.py:169-277`), for the
    sites in ``which`` (those with a circumcentre on the far side of an incident edge -- obtuse
    boundary triangles, smoothed / non-Delaunay connectivity), overwriting ``areas`` there.

    Interior site: area of the convex hull of the incident triangles' circumcentres (the reference
    refuses a non-convex cell).  Boundary site: hull of circumcentres + the two adjacent
    boundary-edge midpoints + the site itself; if those points are not in convex position, the
    triangle (midpoint, midpoint, site) is subtracted.  On such cells this is NOT the Voronoi area
    (the cells overlap), but it is what the reference's operators are built from.
    """
    from scipy.spatial import ConvexHull, QhullError

    def hull_area(pts):
        try:
            hull = ConvexHull(pts)
        except QhullError:  # collinear
            return 0.0, True
        return hull.volume, len(hull.vertices) == len(pts)

    is_boundary = np.zeros(len(sites), dtype=bool)
    is_boundary[boundary_indices] = True
    bedges = edges[boundary_edge_indices]
    which_set = np.zeros(len(sites), dtype=bool)
    which_set[which] = True
    # incident triangles of the requested sites (triangle ids grouped by site, ascending like the
    # reference's scan over the elements)
    hit = which_set[elements]                      # (t, 3)
    t_idx, _ = np.nonzero(hit)
    v_idx = elements[hit]
    order = np.argsort(v_idx, kind="stable")
    v_sorted, t_sorted = v_idx[order], t_idx[order]
    starts = np.searchsorted(v_sorted, which)
    stops = np.searchsorted(v_sorted, which, side="right")
    tri_of = {int(i): t_sorted[a:b] for i, a, b in zip(which, starts, stops)}
    for i in which:
        poly = cc[tri_of[int(i)]]
        if not is_boundary[i]:
            area, convex = hull_area(poly)
            if not convex:
                raise ValueError(
                    f"Malformed Voronoi cell surrounding site {int(i)}: all interior Voronoi cells must be convex."
                )
            areas[i] = area
            continue
        mids = sites[bedges[(bedges == i).any(axis=1)]].mean(axis=1)
        pts = np.concatenate([poly, mids, sites[i][None, :]], axis=0)
        area, convex = hull_area(pts)
        if not convex:
            area -= hull_area(np.concatenate([mids, sites[i][None, :]], axis=0))[0]
        areas[i] = area


def _dual_mesh_numpy(sites: np.ndarray, elements: np.ndarray) -> dict:
    """The Voronoi dual in NumPy -- the construction `tdgl_host_dual_mesh` replaced and is held to, bit for bit
    (`tests/test_host_logic.py`); `Mesh.from_triangulation(..., backend="numpy")`."""
    n = len(sites)
    edges, is_boundary, tri_edge = unique_edges(elements, n)
    cc = circumcenters(sites, elements)
    ends = sites[edges]  # (m, 2, 2)
    centers = ends.mean(axis=1)
    directions = ends[:, 1] - ends[:, 0]
    edge_lengths = np.linalg.norm(directions, axis=1)

    # --- dual edge lengths -------------------------------------------------------
    m = len(edges)
    flat_edge = tri_edge.ravel()  # (3t,) edge id of each (triangle, local edge)
    flat_tri = np.repeat(np.arange(len(elements)), 3)
    order = np.argsort(flat_edge, kind="stable")
    se, st = flat_edge[order], flat_tri[order]
    first = np.ones(len(se), dtype=bool)
    first[1:] = se[1:] != se[:-1]
    t_a = np.full(m, -1, dtype=np.int64)
    t_b = np.full(m, -1, dtype=np.int64)
    t_a[se[first]] = st[first]
    t_b[se[~first]] = st[~first]
    interior = t_b >= 0
    dual = np.empty(m, dtype=float)
    dual[interior] = np.linalg.norm(cc[t_a[interior]] - cc[t_b[interior]], axis=1)
    dual[~interior] = np.linalg.norm(cc[t_a[~interior]] - centers[~interior], axis=1)

    # --- Voronoi areas: sum of signed kites (site, edge midpoint, circumcentre) ------
    # For local edge (p, q) of triangle t with opposite vertex r, the signed height of
    # the circumcentre above the edge (positive towards r) times |pq|/4 goes to both
    # p and q.
    areas = np.zeros(n, dtype=float)
    suspicious = np.zeros(n, dtype=bool)
    for k, (ip, iq, ir) in enumerate([(0, 1, 2), (1, 2, 0), (2, 0, 1)]):
        p = sites[elements[:, ip]]
        q = sites[elements[:, iq]]
        r = sites[elements[:, ir]]
        d = q - p
        length = np.linalg.norm(d, axis=1)
        mid = 0.5 * (p + q)
        # unit normal pointing to the side of r
        nrm = np.column_stack([-d[:, 1], d[:, 0]]) / length[:, None]
        side = np.sign(((r - p) * nrm).sum(axis=1))
        h = ((cc - mid) * nrm).sum(axis=1) * side
        contrib = 0.25 * length * h
        # (np.add.at, not bincount + add: the summation order decides the last bit of an area, and the
        # reference's LU of the singular Neumann matrix -- the oracle's too -- lives on those bits on
        # tiny meshes, tests/test_hip_parity.py::test_very_small_meshes_match_oracle)
        np.add.at(areas, elements[:, ip], contrib)
        np.add.at(areas, elements[:, iq], contrib)
        # a circumcentre on the far side of its edge: the cell of both end sites needs the
        # reference's hull-based construction (`_reference_hull_areas`)
        neg = h < -1e-14 * length
        suspicious[elements[neg, ip]] = True
        suspicious[elements[neg, iq]] = True
    return dict(edges=edges, is_boundary=is_boundary, tri_edge=tri_edge, centers=centers, directions=directions,
                edge_lengths=edge_lengths, circumcenters=cc, dual_lengths=dual, areas=areas, suspicious=suspicious)


class EdgeMesh:
    """Edge-centred quantities of a triangular mesh."""

    def __init__(
        self,
        centers,
        edges,
        boundary_edge_indices,
        directions,
        edge_lengths,
        dual_edge_lengths,
    ):
        self.centers = np.asarray(centers)
        self.edges = np.asarray(edges)
        self.boundary_edge_indices = np.asarray(boundary_edge_indices, dtype=np.int64)
        self.directions = np.asarray(directions)
        self.normalized_directions = (
            self.directions / np.linalg.norm(self.directions, axis=1)[:, np.newaxis]
        )
        self.edge_lengths = np.asarray(edge_lengths)
        self.dual_edge_lengths = np.asarray(dual_edge_lengths)

    @property
    def x(self):
        return self.centers[:, 0]

    @property
    def y(self):
        return self.centers[:, 1]


class Mesh:
    """A triangular mesh plus its Voronoi dual."""

    def __init__(
        self,
        sites,
        elements,
        boundary_indices,
        areas=None,
        dual_sites=None,
        edge_mesh: Optional[EdgeMesh] = None,
    ):
        self.sites = np.asarray(sites)
        self.elements = np.asarray(elements, dtype=np.int64)
        self.boundary_indices = np.asarray(boundary_indices, dtype=np.int64)
        self.areas = None if areas is None else np.asarray(areas)
        self.dual_sites = None if dual_sites is None else np.asarray(dual_sites)
        self.edge_mesh = edge_mesh

    @property
    def x(self):
        return self.sites[:, 0]

    @property
    def y(self):
        return self.sites[:, 1]

    def closest_site(self, xy) -> int:
        """Index of the site nearest to ``xy`` (`tdgl/finite_volume/mesh.py:92-101`)."""
        return int(np.argmin(np.linalg.norm(self.sites - np.atleast_2d(xy), axis=1)))

    @staticmethod
    def from_triangulation(sites, elements, create_submesh: bool = True, backend: str = "native") -> "Mesh":
        sites = np.asarray(sites, dtype=float)
        elements = np.asarray(elements, dtype=np.int64)
        if sites.ndim != 2 or sites.shape[1] != 2:
            raise ValueError(
                f"The site coordinates must have shape (n, 2), got {sites.shape!r}"
            )
        if elements.ndim != 2 or elements.shape[1] != 3:
            raise ValueError(
                f"The elements must have shape (m, 3), got {elements.shape!r}."
            )
        n = len(sites)
        if not create_submesh:
            edges, is_boundary, _ = unique_edges(elements, n)
            return Mesh(sites, elements, np.unique(edges[is_boundary].ravel()))
        if backend == "native":  # tdgl_host_dual_mesh (include/tdgl_host_mesh.h): the same numbers, bit for bit, 5x sooner
            from . import _mesh_lib

            if _mesh_lib.available():
                d = _mesh_lib.dual_mesh(sites, elements)
            else:  # host-only set-up code: the NumPy construction gives the same numbers, bit for bit
                d = _dual_mesh_numpy(sites, elements)
        elif backend == "numpy":
            d = _dual_mesh_numpy(sites, elements)
        else:
            raise ValueError(f"unknown backend {backend!r}")
        edges, is_boundary, cc, dual = d["edges"], d["is_boundary"], d["circumcenters"], d["dual_lengths"]
        areas, suspicious = d["areas"], d["suspicious"]
        boundary_edge_indices = np.flatnonzero(is_boundary)
        boundary_indices = np.unique(edges[is_boundary].ravel())
        centers, directions, edge_lengths = d["centers"], d["directions"], d["edge_lengths"]
        if suspicious.any():
            _reference_hull_areas(areas, np.flatnonzero(suspicious), sites, elements, cc, edges,
                                  boundary_edge_indices, boundary_indices)

        edge_mesh = EdgeMesh(
            centers, edges, boundary_edge_indices, directions, edge_lengths, dual
        )
        return Mesh(
            sites,
            elements,
            boundary_indices,
            areas=areas,
            dual_sites=cc,
            edge_mesh=edge_mesh,
        )

    def smooth(self, iterations: int, create_submesh: bool = True) -> "Mesh":
        """Laplacian smoothing (`tdgl/finite_volume/mesh.py:245-283`): every interior vertex moves
        to the mean of its neighbours, ``iterations`` times, at fixed connectivity; boundary
        vertices stay.  Returns a new mesh."""
        edges, _, _ = unique_edges(self.elements, len(self.sites))
        n = len(self.sites)
        degree = np.bincount(edges.ravel(), minlength=n)
        boundary = self.boundary_indices
        sites = np.asarray(self.sites, dtype=float)
        for _ in range(int(iterations)):
            new = np.zeros((n, 2))
            for k in range(2):
                new[:, k] = np.bincount(edges[:, 0], sites[edges[:, 1], k], minlength=n)
                new[:, k] += np.bincount(edges[:, 1], sites[edges[:, 0], k], minlength=n)
            new /= degree[:, None]
            new[boundary] = sites[boundary]
            sites = new
        return Mesh.from_triangulation(sites, self.elements, create_submesh=create_submesh)

    def get_quantity_on_site(self, quantity_on_edge, vector: bool = True):
        """Edge -> site averaging (`tdgl/finite_volume/mesh.py:203-243`): mean over the
        incident edges of ``F_e * e_hat`` (or ``F_e``), divided by 2."""
        nd = self.edge_mesh.normalized_directions
        edges = self.edge_mesh.edges
        n = len(self.sites)
        if vector:
            fx = quantity_on_edge * nd[:, 0]
            fy = quantity_on_edge * nd[:, 1]
        else:
            fx = fy = quantity_on_edge
        verts = np.concatenate([edges[:, 0], edges[:, 1]])
        counts = np.bincount(verts, minlength=n)
        gx = np.bincount(verts, weights=np.concatenate([fx, fx]), minlength=n) / counts
        gy = np.bincount(verts, weights=np.concatenate([fy, fy]), minlength=n) / counts
        out = np.column_stack([gx, gy]) / 2
        return out if vector else out[:, 0]
