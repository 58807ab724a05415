"""Synthetic mesh point generators for rectangular films.

The reference meshes arbitrary polygons with meshpy/Triangle
(`tdgl/device/meshing.py:15-123`), which is not available on the target image.  The
benchmark configurations (BASELINE.json) are all rectangles, for which a jittered
triangular ("hex") lattice gives an all-acute Delaunay triangulation, i.e. strictly
positive Voronoi dual-edge lengths.  The recipe is the one SURVEY.md §8(d) used for every
reference timing, so site counts match (L=70 -> 5,791; L=465 -> 250,510; L=930 -> 1,000,431).
"""

from typing import Tuple

import numpy as np
from scipy.spatial import Delaunay, cKDTree


def hex_jitter_points(
    width: float,
    height: float = None,
    pitch: float = 1.0,
    jitter: float = 0.05,
    seed: int = 0,
    center: Tuple[float, float] = (0.0, 0.0),
) -> np.ndarray:
    """Sites of a jittered triangular lattice filling a ``width x height`` rectangle.

    Rows are spaced ``~pitch*sqrt(3)/2`` apart; odd rows are shifted by ``pitch/2`` and get
    one extra point on each of the left/right edges so that the rectangle boundary is
    exact.  Interior sites are displaced by ``jitter*pitch*(U(0,1)^2 - 1/2)``.
    """
    if height is None:
        height = width
    lx, ly, h = float(width), float(height), float(pitch)
    ny = int(ly / (h * np.sqrt(3.0) / 2.0)) + 1
    nx = int(lx / h) + 1
    ys = -ly / 2 + np.arange(ny) * (ly / (ny - 1))
    even = np.linspace(-lx / 2, lx / 2, nx)
    odd = np.concatenate(
        [[-lx / 2], np.linspace(-lx / 2 + h / 2, lx / 2 - h / 2, nx - 1), [lx / 2]]
    )
    rows = []
    for j in range(ny):
        xs = even if j % 2 == 0 else odd
        rows.append(np.column_stack([xs, np.full(len(xs), ys[j])]))
    pts = np.concatenate(rows, axis=0)
    interior = (np.abs(pts[:, 0]) < lx / 2) & (np.abs(pts[:, 1]) < ly / 2)
    rng = np.random.default_rng(seed)
    pts[interior] += jitter * h * (rng.random((int(interior.sum()), 2)) - 0.5)
    pts += np.asarray(center, dtype=float)[None, :]
    return pts


def triangulate(points: np.ndarray, backend: str = "native") -> np.ndarray:
    """Delaunay triangles (``(t, 3)`` int64) of a point cloud (its convex hull is covered).

    ``backend="native"``: `tdgl_host_delaunay` (include/tdgl_host_mesh.h; sweep-hull insertion with exact
    predicates, counter-clockwise triangles) -- 0.9 s per million points where Qhull takes 7.7 s, the same
    set of triangles for points in general position.  ``"qhull"``: `scipy.spatial.Delaunay`.  A point cloud
    with coinciding points (the native code leaves the repeats out and says so), one it calls degenerate
    (collinear, or coordinates so large / small that squared distances leave the fp64 range) goes to Qhull;
    ``ValueError`` if Qhull cannot triangulate it either."""
    if backend == "native":
        from . import _mesh_lib

        if _mesh_lib.available():
            status, tri = _mesh_lib.delaunay(points)
            if status == _mesh_lib.ERR_DEGENERATE and len(points) >= 3:
                # all collinear -- or squared distances that overflow / underflow, which the native predicates
                # report the same way: once more in the unit box (a shift and a power-of-two scale)
                pts = np.asarray(points, dtype=float)
                lo = pts.min(axis=0)
                extent = float((pts.max(axis=0) - lo).max())
                if np.isfinite(extent) and extent > 0:
                    points = (pts - lo) * 2.0 ** -np.ceil(np.log2(extent))
                    status, tri = _mesh_lib.delaunay(points)
            if status == _mesh_lib.OK:
                return tri
        # ERR_SKIPPED (coinciding points), still ERR_DEGENERATE, or no library: Qhull decides
    elif backend != "qhull":
        raise ValueError(f"unknown backend {backend!r}")
    try:
        tri = Delaunay(points).simplices
    except Exception as exc:  # QhullError: flat input
        raise ValueError(f"triangulate: the points cannot be triangulated ({str(exc).splitlines()[0]})") from exc
    return np.asarray(tri, dtype=np.int64)


def rectangle_mesh_points(
    width: float, height: float, max_edge_length: float, seed: int = 0
) -> Tuple[np.ndarray, np.ndarray]:
    """Points and triangles for a rectangle with edges no longer than ``max_edge_length``."""
    # A jittered equilateral lattice of pitch h has edges up to ~h*(1 + jitter*sqrt(2)),
    # and the boundary rows are stretched to fit; 0.9 keeps the longest edge under the cap.
    pitch = 0.9 * max_edge_length
    pts = hex_jitter_points(width, height, pitch=pitch, seed=seed)
    return pts, triangulate(pts)


# ---------------------------------------------------------------------------------------
# General polygons (with holes).  The reference calls meshpy/Triangle
# (`tdgl/device/meshing.py:15-123`); here: boundary points at spacing <= h on every polygon
# edge, a jittered triangular lattice in the interior kept >= 0.6 h away from the boundary, a
# Delaunay triangulation, triangles outside the domain removed, and boundary segments split
# until (i) every segment is a triangle edge and (ii) no point lies inside a segment's diametral
# circle (Gabriel condition).  (ii) puts the circumcentre of every boundary triangle on the
# domain side of its boundary edge, so all Voronoi dual lengths / cell areas are the plain
# finite-volume ones (positive), exactly what `Mesh.from_triangulation` computes.
def _points_in_poly(poly, pts):
    from matplotlib import path as mpath

    poly = np.asarray(poly, dtype=float)
    if not np.allclose(poly[0], poly[-1]):  # Path(closed=True) ignores the last vertex
        poly = np.concatenate([poly, poly[:1]])
    return mpath.Path(poly, closed=True).contains_points(pts)


def _segment_distance(pts, a, b, within=None):
    """Distance of every point to the segments a[k] -> b[k]; returns the minimum over segments.

    ``within``: the caller only compares the result with this threshold, so points farther than that from a
    segment need not be measured (they keep ``inf``): a k-d tree hands every segment the points inside the
    circle around its midpoint that contains the whole ``within``-neighbourhood, and the distance formula
    runs on those alone -- the same numbers for every point that can fall below the threshold."""
    out = np.full(len(pts), np.inf)
    if within is None or len(pts) < 2000:
        for p0, p1 in zip(a, b):
            d = p1 - p0
            t = np.clip(((pts - p0) @ d) / (d @ d), 0.0, 1.0)
            out = np.minimum(out, np.linalg.norm(pts - (p0 + t[:, None] * d), axis=1))
        return out
    from scipy.spatial import cKDTree

    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    tree = cKDTree(pts)
    reach = 0.5 * np.linalg.norm(b - a, axis=1) + float(within)
    near = tree.query_ball_point(0.5 * (a + b), reach * (1 + 1e-9) + 1e-300)
    for p0, p1, idx in zip(a, b, near):
        if not idx:
            continue
        idx = np.asarray(idx)
        q = pts[idx]
        d = p1 - p0
        t = np.clip(((q - p0) @ d) / (d @ d), 0.0, 1.0)
        out[idx] = np.minimum(out[idx], np.linalg.norm(q - (p0 + t[:, None] * d), axis=1))
    return out


MIN_BOUNDARY_ANGLE_DEG = 1.0  # interior (or exterior) angle between consecutive boundary segments, see _polygon_mesh_at_pitch


def polygon_mesh(film, holes=(), max_edge_length=1.0, seed=0, max_rounds=12, backend="native"):
    """Boundary-conforming Delaunay mesh of ``film`` minus ``holes`` (closed or open ``(k, 2)``
    vertex arrays) with no edge longer than ``max_edge_length``.  Returns ``(points, triangles)``.
    ``backend``: the triangulator (`triangulate`)."""
    pitch = 0.65 * float(max_edge_length)
    for _ in range(6):
        pts, tri = _polygon_mesh_at_pitch(film, holes, pitch, seed, max_rounds, backend)
        e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]])
        longest = np.linalg.norm(pts[e[:, 0]] - pts[e[:, 1]], axis=1).max()
        if longest <= max_edge_length:
            return pts, tri
        pitch *= 0.97 * max_edge_length / longest
        if pitch < 0.65 * float(max_edge_length) / 3.0:
            # (a few per cent of adjustment is normal; an edge many times too long means the boundary could not be
            # resolved -- a needle-shaped notch or spike narrower than the pitch -- and a lattice fine enough to hide
            # it would have 10x the points asked for)
            raise RuntimeError(f"polygon_mesh: an edge of length {longest:.3g} remains at max_edge_length = {max_edge_length:g} "
                               "(a notch or spike of the boundary much narrower than that?)")
    raise RuntimeError("polygon_mesh: could not satisfy max_edge_length")  # pragma: no cover


def _polygon_mesh_at_pitch(film, holes, h, seed, max_rounds, backend="native"):
    h = float(h)
    loops = []
    for poly in [film] + list(holes):
        poly = np.asarray(poly, dtype=float)
        if np.allclose(poly[0], poly[-1]):
            poly = poly[:-1]
        # repeated vertices (polygon generators repeat corner points) would be zero-length
        # boundary segments
        keep = np.linalg.norm(poly - np.roll(poly, 1, axis=0), axis=1) > 1e-12 * max(1.0, np.abs(poly).max())
        loops.append(poly[keep])

    # Two boundary segments that meet at a very small angle encroach on each other however often they are split
    # (Ruppert's small-angle problem; the concentric-shell remedy is not implemented): say so instead of refining
    # until the memory is gone.
    for lp in loops:
        if len(lp) < 3:
            raise ValueError("polygon_mesh: a boundary loop needs at least three distinct vertices")
        a = np.roll(lp, 1, axis=0) - lp
        b = np.roll(lp, -1, axis=0) - lp
        cosang = (a * b).sum(axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
        k = int(np.argmax(cosang))
        ang = float(np.degrees(np.arccos(np.clip(cosang[k], -1.0, 1.0))))
        if ang < MIN_BOUNDARY_ANGLE_DEG:
            raise ValueError(f"polygon_mesh: the boundary turns by all but {ang:.2f} degrees at vertex {k} "
                             f"({lp[k][0]:.6g}, {lp[k][1]:.6g}); corners sharper than {MIN_BOUNDARY_ANGLE_DEG:g} degrees "
                             "cannot be meshed with well-shaped cells -- round or cut the spike")

    def resample(loop):
        # (also: which of the points are input vertices where the boundary turns by more than 120 degrees -- the two
        # segments there encroach on each other's diametral circles unless they are cut at equal distances from the
        # vertex, see the split below)
        a = np.roll(loop, 1, axis=0) - loop
        b = np.roll(loop, -1, axis=0) - loop
        cosang = (a * b).sum(axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
        sharp = cosang > 0.5  # interior or exterior angle below 60 degrees
        pts, flags = [], []
        for j, (p0, p1) in enumerate(zip(loop, np.roll(loop, -1, axis=0))):
            k = max(1, int(np.ceil(np.linalg.norm(p1 - p0) / h)))
            pts.append(p0 + (p1 - p0) * (np.arange(k) / k)[:, None])
            f = np.zeros(k, dtype=bool)
            f[0] = sharp[j]
            flags.append(f)
        return np.concatenate(pts), np.concatenate(flags)

    resampled = [resample(lp) for lp in loops]
    bloops = [r[0] for r in resampled]
    bsharp = [r[1] for r in resampled]
    (x0, y0), (x1, y1) = loops[0].min(axis=0), loops[0].max(axis=0)
    lattice = hex_jitter_points(x1 - x0 + 2 * h, y1 - y0 + 2 * h, pitch=h, seed=seed,
                                center=(0.5 * (x0 + x1), 0.5 * (y0 + y1)))
    keep = _points_in_poly(loops[0], lattice)
    for hole in loops[1:]:
        keep &= ~_points_in_poly(hole, lattice)
    lattice = lattice[keep]
    seg_a = np.concatenate(loops)
    seg_b = np.concatenate([np.roll(lp, -1, axis=0) for lp in loops])
    lattice = lattice[_segment_distance(lattice, seg_a, seg_b, within=0.6 * h) >= 0.6 * h]

    for _ in range(max_rounds):
        nb = [len(b) for b in bloops]
        pts = np.concatenate(bloops + [lattice])
        tri = triangulate(pts, backend)
        cent = pts[tri].mean(axis=1)
        inside = _points_in_poly(loops[0], cent)
        for hole in loops[1:]:
            inside &= ~_points_in_poly(hole, cent)
        # Points resampled along an OBLIQUE straight side are collinear only up to rounding: the triangulation of their
        # convex hull then contains slivers of ~1e-15 height along that side whose centroids pass for inside.  They are
        # not cells of the mesh (their edges skip boundary points); drop them.
        pa, pb, pc = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
        area2 = np.abs((pb[:, 0] - pa[:, 0]) * (pc[:, 1] - pa[:, 1]) - (pb[:, 1] - pa[:, 1]) * (pc[:, 0] - pa[:, 0]))
        inside &= area2 > 1e-9 * h * h
        tri = tri[inside]
        edges = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
        edge_keys = np.unique(edges[:, 0] * np.int64(len(pts)) + edges[:, 1])
        tree = cKDTree(pts)
        split = False
        off = 0
        new_loops, new_sharp = [], []
        for b, n_b, shp in zip(bloops, nb, bsharp):
            idx = off + np.arange(n_b)
            nxt = off + (np.arange(n_b) + 1) % n_b
            mids = 0.5 * (pts[idx] + pts[nxt])
            rad = 0.5 * np.linalg.norm(pts[nxt] - pts[idx], axis=1)
            # (i) the segment is a triangle edge
            keys = np.minimum(idx, nxt) * np.int64(len(pts)) + np.maximum(idx, nxt)
            pos = np.searchsorted(edge_keys, keys)
            present = (pos < len(edge_keys)) & (edge_keys[np.minimum(pos, len(edge_keys) - 1)] == keys)
            # (ii) no other point strictly inside its diametral circle: the tree proposes, the distance decides
            near = tree.query_ball_point(mids, rad)
            out, out_sharp = [], []
            for k in range(n_b):
                cand = np.asarray([j for j in near[k] if j != idx[k] and j != nxt[k]], dtype=np.int64)
                encroached = bool(len(cand)) and bool(
                    (np.linalg.norm(pts[cand] - mids[k], axis=1) < rad[k] * (1 - 1e-12)).any())
                out.append(b[k])
                out_sharp.append(bool(shp[k]))
                if not present[k] or encroached:
                    k2 = (k + 1) % n_b
                    cut = mids[k]
                    if shp[k] != shp[k2]:
                        # one end is a sharp input vertex: cut on a concentric shell around it (a power of two times
                        # the pitch, the one nearest to the midpoint), so that the two segments that meet there end
                        # up with pieces of EQUAL length next to the vertex and stop encroaching on each other
                        v, w = (b[k], b[k2]) if shp[k] else (b[k2], b[k])
                        length = 2.0 * rad[k]
                        d = h * 2.0 ** np.round(np.log2(0.5 * length / h))
                        if 0.2 * length < d < 0.8 * length:
                            cut = v + (w - v) * (d / length)
                    out.append(cut)
                    out_sharp.append(False)
                    split = True
            new_loops.append(np.array(out))
            new_sharp.append(np.array(out_sharp, dtype=bool))
            off += n_b
        if not split:
            break
        bloops, bsharp = new_loops, new_sharp
        # lattice points too close to a refined boundary piece are dropped
        ba = np.concatenate(bloops)
        bb = np.concatenate([np.roll(b, -1, axis=0) for b in bloops])
        seg_len = np.linalg.norm(bb - ba, axis=1)
        short = seg_len < 0.5 * h
        if short.any():
            thr = 0.6 * seg_len[short].max()
            lattice = lattice[_segment_distance(lattice, ba[short], bb[short], within=thr) >= thr]
    else:  # pragma: no cover
        raise RuntimeError("polygon_mesh: boundary did not become conforming")
    used = np.unique(tri)
    remap = np.full(len(pts), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    return pts[used], remap[tri]
