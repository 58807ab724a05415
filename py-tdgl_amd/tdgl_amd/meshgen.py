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
from scipy.spatial import Delaunay


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


def triangulate(points: np.ndarray) -> np.ndarray:
    """Delaunay triangles (``(t, 3)`` int64) of a convex point cloud."""
    return np.asarray(Delaunay(points).simplices, dtype=np.int64)


def rectangle_mesh_points(
    width: float, height: float, max_edge_length: float, seed: int = 0
) -> Tuple[np.ndarray, np.ndarray]:
    """Points and triangles for a rectangle with edges no longer than ``max_edge_length``."""
    # A jittered equilateral lattice of pitch h has edges up to ~h*(1 + jitter*sqrt(2)),
    # and the boundary rows are stretched to fit; 0.9 keeps the longest edge under the cap.
    pitch = 0.9 * max_edge_length
    pts = hex_jitter_points(width, height, pitch=pitch, seed=seed)
    return pts, triangulate(pts)
