"""Shape helpers for film / terminal polygons (`tdgl/geometry.py:85-136` of the reference)."""

from typing import Optional, Tuple

import numpy as np


def rotate(coords: np.ndarray, angle_degrees: float) -> np.ndarray:
    """Rotate ``(n, 2)`` coordinates counter-clockwise about the origin."""
    a = np.radians(angle_degrees)
    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    return np.asarray(coords) @ rot.T


def box(width: float, height: Optional[float] = None, points: int = 101,
        center: Tuple[float, float] = (0, 0), angle: float = 0) -> np.ndarray:
    """Boundary points of a ``width x height`` rectangle, counter-clockwise from the
    lower-right corner, about ``points`` of them, centred on ``center``."""
    width = abs(width)
    height = width if height is None else abs(height)
    per = 2 * (width + height)
    nx = round(points * width / per)
    ny = round(points * height / per)
    hx, hy = width / 2, height / 2
    right = np.column_stack([np.full(ny, hx), np.linspace(-hy, hy, ny)])
    top = np.column_stack([np.linspace(hx, -hx, nx), np.full(nx, hy)])
    left = np.column_stack([np.full(ny, -hx), np.linspace(hy, -hy, ny)])
    bottom = np.column_stack([np.linspace(-hx, hx, nx), np.full(nx, -hy)])
    coords = np.concatenate([right, top, left, bottom]) + np.asarray(center, dtype=float)
    if angle:
        coords = rotate(coords, angle)
    return coords


def circle(radius: float, points: int = 100, center: Tuple[float, float] = (0, 0)) -> np.ndarray:
    """Boundary points of a circle."""
    return ellipse(radius, radius, points=points, center=center)


def ellipse(a: float, b: float, points: int = 100, center: Tuple[float, float] = (0, 0),
            angle: float = 0) -> np.ndarray:
    """Boundary points of an ellipse with semi-axes ``a`` (x) and ``b`` (y)."""
    t = np.linspace(0, 2 * np.pi, points, endpoint=False)
    xy = np.column_stack([a * np.cos(t), b * np.sin(t)])
    if angle:
        xy = rotate(xy, angle)
    return xy + np.asarray(center, dtype=float)


def ensure_unique(coords: np.ndarray) -> np.ndarray:
    """Drop repeated vertices, keeping first occurrences in order."""
    coords = np.asarray(coords)
    _, idx = np.unique(coords, axis=0, return_index=True)
    return coords[np.sort(idx)]


def close_curve(points: np.ndarray) -> np.ndarray:
    """Append the first point if the curve is open."""
    points = np.asarray(points)
    if not np.allclose(points[0], points[-1]):
        points = np.concatenate([points, points[:1]], axis=0)
    return points


def unit_vector(vector: np.ndarray) -> np.ndarray:
    """Rows of ``vector`` scaled to unit length (`tdgl/geometry.py:166-168`)."""
    vector = np.asarray(vector, dtype=float)
    return vector / np.linalg.norm(vector, axis=-1)[:, np.newaxis]


def path_vectors(path: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Segment lengths [n-1] and unit normals [n-1, 2] of a path given by its n points
    (`tdgl/geometry.py:171-185`): the normal of a segment d is d x z = (d_y, -d_x) / |d|, i.e. it
    points to the right of the direction of travel."""
    dr = np.diff(np.asarray(path, dtype=float), axis=0)
    normals = np.stack([dr[:, 1], -dr[:, 0]], axis=1)
    return np.linalg.norm(dr, axis=1), unit_vector(normals)
