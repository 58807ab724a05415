"""Piecewise-linear interpolation on the mesh triangulation.

The reference evaluates fields between the sites with ``matplotlib.tri.LinearTriInterpolator``
on ``Device.triangulation`` (`tdgl/solution/solution.py:364-462`, `tdgl/device/device.py:209-219`).
This is the same interpolant without matplotlib: locate the triangle containing each query point,
weight its three corner values with the barycentric coordinates; points outside every triangle
(beyond the rim, inside a hole) come back as NaN, as a masked value's data does there.
"""

import numpy as np
from scipy.spatial import cKDTree


class TriLinearInterpolator:
    """``f = TriLinearInterpolator(sites, elements); f(values, points)``.

    Point location: the containing triangle is searched among the triangles whose centroids are
    nearest to the point; the few points this does not settle (strongly graded meshes) are tested
    against all triangles.
    """

    def __init__(self, sites: np.ndarray, elements: np.ndarray, candidates: int = 16):
        self.sites = np.asarray(sites, dtype=float)
        self.elements = np.asarray(elements, dtype=np.int64)
        tri = self.sites[self.elements]  # [t, 3, 2]
        self._a = tri[:, 0]
        # inverse of the edge matrix [b - a, c - a] per triangle
        e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
        det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
        self._inv = np.stack([e2[:, 1], -e2[:, 0], -e1[:, 1], e1[:, 0]], axis=1) / det[:, None]
        self._tree = cKDTree(tri.mean(axis=1))
        self._k = int(min(candidates, len(self.elements)))
        self._tol = 1e-12

    def _barycentric(self, t: np.ndarray, p: np.ndarray) -> np.ndarray:
        d = p - self._a[t]
        inv = self._inv[t]
        l1 = inv[..., 0] * d[..., 0] + inv[..., 1] * d[..., 1]
        l2 = inv[..., 2] * d[..., 0] + inv[..., 3] * d[..., 1]
        return np.stack([1.0 - l1 - l2, l1, l2], axis=-1)

    def locate(self, points: np.ndarray):
        """Triangle index per point (-1: in no triangle) and the barycentric weights [m, 3]."""
        p = np.atleast_2d(np.asarray(points, dtype=float))
        m = len(p)
        index = np.full(m, -1, dtype=np.int64)
        weights = np.zeros((m, 3))
        if m == 0:
            return index, weights
        _, cand = self._tree.query(p, k=self._k)
        cand = cand.reshape(m, -1)
        lam = self._barycentric(cand, p[:, None, :])          # [m, k, 3]
        inside = (lam >= -self._tol).all(axis=2)
        hit = inside.any(axis=1)
        first = inside.argmax(axis=1)
        rows = np.nonzero(hit)[0]
        index[rows] = cand[rows, first[rows]]
        weights[rows] = lam[rows, first[rows]]
        # leftovers: exhaustive test (chunked), settles both "far centroid" and "outside"
        rest = np.nonzero(~hit)[0]
        all_t = np.arange(len(self.elements))
        for lo in range(0, len(rest), 256):
            r = rest[lo:lo + 256]
            lam_all = self._barycentric(all_t[None, :], p[r][:, None, :])
            ok = (lam_all >= -self._tol).all(axis=2)
            found = ok.any(axis=1)
            f = ok.argmax(axis=1)
            index[r[found]] = f[found]
            weights[r[found]] = lam_all[np.nonzero(found)[0], f[found]]
        return index, weights

    def __call__(self, values: np.ndarray, points: np.ndarray) -> np.ndarray:
        values = np.asarray(values)
        index, w = self.locate(points)
        corner = values[self.elements[np.maximum(index, 0)]]   # [m, 3, ...]
        out = np.einsum("mk,mk...->m...", w, corner)
        if np.iscomplexobj(out):
            out = out.astype(complex)
            out[index < 0] = complex(np.nan, np.nan)
        else:
            out = out.astype(float)
            out[index < 0] = np.nan
        return out
