"""Device description for the solver path: ``Layer``, ``Polygon``, ``Device``.

Kept from the reference (`tdgl/device/device.py:49-566`, `layer.py:6-57`, `polygon.py:29-140`):
constructor signatures, ``make_mesh``, ``terminal_info()``, probe points and the unit scales
the solver needs, ``to_hdf5`` / ``from_hdf5``.  Not kept: shapely polygon algebra (a native
boundary-conforming mesher stands in for meshpy, `tdgl_amd.meshgen`) and plotting.

Units.  The reference does unit conversion with ``pint`` (absent on the target image).  Only
four conversions reach the solver, so they are written out with CODATA-2018 constants, the
ones pint's default registry uses:

    Bc2     = Phi_0 / (2 pi xi^2)                         (device.py:152-155)
    K0      = 4 xi Bc2 / (mu_0 Lambda),  Lambda = lambda^2/d   (device.py:162-168)
    A_scale = [field_units] / (Bc2 * xi)                  (solver.py:176-180)
    J_scale = 4 [current_units]/[length_units] / K0       (solver.py:251-253)

Quantities returned here are plain floats in SI units (the reference returns pint
Quantities); this boundary is the one place where parity with the reference is not pinned
by a fixture.
"""

from operator import attrgetter
from typing import List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np

from .finite_volume import Mesh
from .geometry import close_curve
from .meshgen import polygon_mesh, rectangle_mesh_points

PHI_0 = 2.067833848e-15  # Wb, magnetic flux quantum h / 2e
MU_0 = 1.25663706212e-6  # N / A^2

LENGTH_UNITS = {"m": 1.0, "mm": 1e-3, "um": 1e-6, "µm": 1e-6, "nm": 1e-9}
FIELD_UNITS = {"T": 1.0, "mT": 1e-3, "uT": 1e-6, "µT": 1e-6, "nT": 1e-9, "G": 1e-4, "gauss": 1e-4}
CURRENT_UNITS = {"A": 1.0, "mA": 1e-3, "uA": 1e-6, "µA": 1e-6, "nA": 1e-9}


def _unit(table, name, kind):
    try:
        return table[name]
    except KeyError:
        raise ValueError(f"Unsupported {kind} unit {name!r}; supported: {sorted(table)}") from None


class Layer:
    """A superconducting thin film (`tdgl/device/layer.py:6-41`).  Lengths are in the
    device's ``length_units``."""

    def __init__(
        self,
        *,
        london_lambda: float,
        coherence_length: float,
        thickness: float,
        conductivity: Union[float, None] = None,
        u: float = 5.79,
        gamma: float = 10.0,
        z0: float = 0,
    ):
        self.london_lambda = london_lambda
        self.coherence_length = coherence_length
        self.thickness = thickness
        self.conductivity = conductivity
        self.u = u
        self.gamma = gamma
        self.z0 = z0

    @property
    def Lambda(self) -> float:
        return self.london_lambda**2 / self.thickness

    def copy(self) -> "Layer":
        return Layer(
            london_lambda=self.london_lambda,
            coherence_length=self.coherence_length,
            thickness=self.thickness,
            conductivity=self.conductivity,
            u=self.u,
            gamma=self.gamma,
            z0=self.z0,
        )

    def to_hdf5(self, h5_group) -> None:
        """`tdgl/device/layer.py:59-72`."""
        for k in ("london_lambda", "coherence_length", "thickness", "u", "gamma", "z0"):
            h5_group.attrs[k] = getattr(self, k)
        if self.conductivity is not None:
            h5_group.attrs["conductivity"] = self.conductivity

    @staticmethod
    def from_hdf5(h5_group) -> "Layer":
        """`tdgl/device/layer.py:74-98`."""
        get = lambda key: h5_group.attrs[key] if key in h5_group.attrs else None  # noqa: E731
        kwargs = {k: get(k) for k in ("london_lambda", "coherence_length", "thickness", "conductivity", "u", "gamma", "z0")}
        return Layer(**{k: (v if v is None else float(v)) for k, v in kwargs.items()})

    def __eq__(self, other):
        if not isinstance(other, Layer):
            return False
        keys = ("london_lambda", "coherence_length", "thickness", "conductivity", "u", "gamma", "z0")
        return all(getattr(self, k) == getattr(other, k) for k in keys)

    def __repr__(self):
        return (
            f"Layer(london_lambda={self.london_lambda}, coherence_length={self.coherence_length},"
            f" thickness={self.thickness}, u={self.u}, gamma={self.gamma}, z0={self.z0})"
        )


def _points_in_polygon(poly: np.ndarray, pts: np.ndarray, radius: float = 0.0) -> np.ndarray:
    """Even-odd test.  Uses matplotlib's path (what the reference calls,
    `tdgl/device/polygon.py:137`) when available, else a vectorised ray cast."""
    try:
        from matplotlib import path as mpath

        return mpath.Path(poly, closed=True).contains_points(pts, radius=radius)
    except ImportError:  # pragma: no cover
        x, y = pts[:, 0], pts[:, 1]
        inside = np.zeros(len(pts), dtype=bool)
        x0, y0 = poly[:-1, 0], poly[:-1, 1]
        x1, y1 = poly[1:, 0], poly[1:, 1]
        for a, b, c, d in zip(x0, y0, x1, y1):
            crosses = (b > y) != (d > y)
            with np.errstate(divide="ignore", invalid="ignore"):
                xi = a + (y - b) * (c - a) / (d - b)
            inside ^= crosses & (x < xi)
        return inside


class Polygon:
    """A simply connected polygon given by its vertices (`tdgl/device/polygon.py:29-140`)."""

    def __init__(self, name: Union[str, None] = None, *, points, mesh: bool = True):
        self.name = name
        self.mesh = mesh
        self.points = points

    @property
    def points(self) -> np.ndarray:
        return self._points

    @points.setter
    def points(self, points) -> None:
        if isinstance(points, Polygon):
            points = points.points
        pts = np.asarray(points, dtype=float)
        if pts.ndim != 2 or pts.shape[-1] != 2 or len(pts) < 3:
            raise ValueError(f"Expected shape (n, 2), but got {pts.shape}.")
        pts = close_curve(pts)
        # counter-clockwise orientation (signed area > 0)
        area2 = np.sum(pts[:-1, 0] * pts[1:, 1] - pts[1:, 0] * pts[:-1, 1])
        if area2 == 0:
            raise ValueError("The given points do not define a valid polygon (zero area).")
        if area2 < 0:
            pts = pts[::-1]
        self._points = pts

    def to_hdf5(self, h5_group) -> None:
        """`tdgl/device/polygon.py:581-586`."""
        if self.name is not None:
            h5_group.attrs["name"] = self.name
        h5_group.attrs["mesh"] = self.mesh
        h5_group["points"] = self.points

    @classmethod
    def from_hdf5(cls, h5_group) -> "Polygon":
        """`tdgl/device/polygon.py:588-598`."""
        name = h5_group.attrs["name"] if "name" in h5_group.attrs else None
        return cls(name=None if name is None else str(name), points=np.array(h5_group["points"]),
                   mesh=bool(h5_group.attrs["mesh"]))

    def __eq__(self, other) -> bool:
        if other is self:
            return True
        if not isinstance(other, Polygon):
            return False
        return self.name == other.name and self.points.shape == other.points.shape and np.allclose(
            self.points, other.points)

    __hash__ = None

    @property
    def area(self) -> float:
        p = self._points
        return 0.5 * abs(np.sum(p[:-1, 0] * p[1:, 1] - p[1:, 0] * p[:-1, 1]))

    @property
    def bbox(self):
        p = self._points
        return (p[:, 0].min(), p[:, 1].min()), (p[:, 0].max(), p[:, 1].max())

    @property
    def extents(self) -> Tuple[float, float]:
        (x0, y0), (x1, y1) = self.bbox
        return x1 - x0, y1 - y0

    @property
    def is_valid(self) -> bool:
        return self.name is not None and self.area > 0

    def is_rectangle(self, tol: float = 1e-12) -> bool:
        """All vertices on the bounding box and the area equal to the box area."""
        (x0, y0), (x1, y1) = self.bbox
        p = self._points
        scale = max(x1 - x0, y1 - y0)
        on_x = np.isclose(p[:, 0], x0, atol=tol * scale) | np.isclose(p[:, 0], x1, atol=tol * scale)
        on_y = np.isclose(p[:, 1], y0, atol=tol * scale) | np.isclose(p[:, 1], y1, atol=tol * scale)
        return bool(np.all(on_x | on_y)) and np.isclose(self.area, (x1 - x0) * (y1 - y0), rtol=1e-9)

    def contains_points(self, points, index: bool = False, radius: float = 0):
        mask = _points_in_polygon(self._points, np.atleast_2d(points), radius=radius)
        return np.where(mask)[0] if index else mask

    def copy(self) -> "Polygon":
        return Polygon(self.name, points=self._points.copy(), mesh=self.mesh)

    def translate(self, dx: float = 0.0, dy: float = 0.0, inplace: bool = False) -> "Polygon":
        poly = self if inplace else self.copy()
        poly._points = poly._points + np.array([[dx, dy]])
        return poly

    def __repr__(self):
        return f"Polygon(name={self.name!r}, {len(self._points)} points)"


class TerminalInfo(NamedTuple):
    """Same fields as the reference's ``TerminalInfo`` (`tdgl/device/device.py:29-46`)."""

    name: str
    site_indices: Sequence[int]
    edge_indices: Sequence[int]
    boundary_edge_indices: Sequence[int]
    length: float


class Device:
    """A thin-film device: one film polygon, optional current terminals and voltage probes.

    Holes are accepted only when a mesh is supplied explicitly
    (``Device.mesh_from_triangulation``); the built-in mesher handles rectangular films.
    """

    def __init__(
        self,
        name: str,
        *,
        layer: Layer,
        film: Polygon,
        holes: Union[List[Polygon], None] = None,
        terminals: Union[List[Polygon], None] = None,
        probe_points: Optional[Sequence[Tuple[float, float]]] = None,
        length_units: str = "um",
    ):
        self.name = name
        self.layer = layer
        self.film = film
        self.holes = holes or []
        self.terminals = tuple(terminals or [])
        seen = set()
        for term in self.terminals:
            term.mesh = False
            if term.name is None or term.name in seen:
                raise ValueError("All terminals must have a unique name")
            seen.add(term.name)
        for polygon in [self.film] + self.holes:
            if not polygon.is_valid:
                raise ValueError(f"Invalid Polygon: {polygon!r}.")
        if len(self.holes) != len(set(h.name for h in self.holes)):
            raise ValueError("All holes must have a unique name.")
        if probe_points is not None:
            probe_points = np.asarray(probe_points, dtype=float).squeeze()
            if probe_points.ndim != 2 or probe_points.shape[1] != 2:
                raise ValueError(f"Probe points must have shape (n, 2), got {probe_points.shape}.")
            if not self.contains_points(probe_points).all():
                raise ValueError("All probe points must lie within the film.")
        self.probe_points = probe_points
        _unit(LENGTH_UNITS, length_units, "length")
        self._length_units = length_units
        self.mesh: Optional[Mesh] = None

    def to_hdf5(self, h5_group, save_mesh: bool = True) -> None:
        """Serialise into an open HDF5 group in the reference's layout (`tdgl/device/device.py:772-809`)."""
        from .io import write_mesh

        h5_group.attrs["name"] = self.name
        h5_group.attrs["length_units"] = self.length_units
        self.layer.to_hdf5(h5_group.create_group("layer"))
        self.film.to_hdf5(h5_group.create_group("film"))
        if self.terminals:
            grp = h5_group.create_group("terminals")
            for terminal in self.terminals:
                terminal.to_hdf5(grp.create_group(terminal.name))
        if self.probe_points is not None:
            h5_group["probe_points"] = self.probe_points
        if self.holes:
            grp = h5_group.create_group("holes")
            for hole in sorted(self.holes, key=lambda h: h.name):
                hole.to_hdf5(grp.create_group(hole.name))
        if save_mesh and self.mesh is not None:
            write_mesh(h5_group.create_group("mesh"), self.mesh)

    @classmethod
    def from_hdf5(cls, path_or_group) -> "Device":
        """The inverse of :meth:`to_hdf5` (`tdgl/device/device.py:811-865`): a path (opened through
        `tdgl_amd.io.open_h5`, i.e. h5py) or an open group."""
        from .io import open_h5, read_mesh

        if isinstance(path_or_group, (str, bytes)) or hasattr(path_or_group, "__fspath__"):
            f = open_h5(path_or_group, "r")
            try:
                return cls.from_hdf5(f)
            finally:
                f.close()
        f = path_or_group
        if not hasattr(f, "attrs"):
            raise TypeError(f"Expected an h5py.File or h5py.Group, but got {type(f)}.")
        terminals = holes = probe_points = mesh = None
        if "terminals" in f:
            terminals = [Polygon.from_hdf5(f["terminals"][k]) for k in f["terminals"]]
        if "holes" in f:
            holes = [Polygon.from_hdf5(f["holes"][k]) for k in sorted(f["holes"])]
        if "probe_points" in f:
            probe_points = np.array(f["probe_points"])
        if "mesh" in f:
            mesh = read_mesh(f["mesh"])
        device = cls(str(f.attrs["name"]), layer=Layer.from_hdf5(f["layer"]), film=Polygon.from_hdf5(f["film"]),
                     holes=holes, terminals=terminals, probe_points=probe_points,
                     length_units=str(f.attrs["length_units"]))
        if mesh is not None:
            device.mesh = mesh
        return device

    def __eq__(self, other) -> bool:
        """Same description (`tdgl/device/device.py:885-915`): name, layer, film, holes, terminals,
        probe points and length units; the mesh is not compared."""
        if other is self:
            return True
        if not isinstance(other, Device):
            return False

        def same(seq1, seq2):
            key = attrgetter("name")
            return sorted(seq1, key=key) == sorted(seq2, key=key)

        if self.probe_points is None or other.probe_points is None:
            same_probes = self.probe_points is None and other.probe_points is None
        else:
            same_probes = self.probe_points.shape == other.probe_points.shape and np.allclose(
                self.probe_points, other.probe_points)
        return (self.name == other.name and self.layer == other.layer and self.film == other.film
                and same(self.holes, other.holes) and same(list(self.terminals), list(other.terminals))
                and same_probes and self.length_units == other.length_units)

    __hash__ = None

    # -- units and scales -----------------------------------------------------------------
    @property
    def length_units(self) -> str:
        return self._length_units

    @property
    def _len_m(self) -> float:
        return LENGTH_UNITS[self._length_units]

    @property
    def coherence_length(self) -> float:
        """xi in ``length_units``."""
        return self.layer.coherence_length

    @property
    def london_lambda(self) -> float:
        return self.layer.london_lambda

    @property
    def thickness(self) -> float:
        return self.layer.thickness

    @property
    def Lambda(self) -> float:
        """Effective penetration depth lambda^2/d in ``length_units``."""
        return self.layer.Lambda

    @property
    def kappa(self) -> float:
        return self.layer.london_lambda / self.layer.coherence_length

    @property
    def Bc2(self) -> float:
        """Upper critical field in tesla, Phi_0 / (2 pi xi^2)."""
        xi_m = self.layer.coherence_length * self._len_m
        return PHI_0 / (2 * np.pi * xi_m**2)

    @property
    def A0(self) -> float:
        """Vector potential scale xi * Bc2 in T m."""
        return self.Bc2 * self.layer.coherence_length * self._len_m

    @property
    def K0(self) -> float:
        """Sheet current density scale in A/m, 4 xi Bc2 / (mu_0 Lambda)."""
        xi_m = self.layer.coherence_length * self._len_m
        return 4 * xi_m * self.Bc2 / (MU_0 * self.layer.Lambda * self._len_m)

    def tau0(self, conductivity: Union[float, None] = None) -> float:
        """Time scale mu_0 sigma lambda^2 in seconds (sigma in S / length_units)."""
        sigma = self.layer.conductivity if conductivity is None else conductivity
        if sigma is None:
            raise ValueError(
                "The time scale tau0 requires the normal state conductivity to be defined."
            )
        sigma_si = sigma / self._len_m
        return MU_0 * sigma_si * (self.layer.london_lambda * self._len_m) ** 2

    def V0(self, conductivity: Union[float, None] = None) -> float:
        """Potential scale xi J0 / sigma in volts."""
        sigma = self.layer.conductivity if conductivity is None else conductivity
        if sigma is None:
            raise ValueError(
                "The electric potential scale V_0 requires the normal state"
                " conductivity to be defined."
            )
        sigma_si = sigma / self._len_m
        j0 = self.K0 / (self.layer.thickness * self._len_m)
        return self.layer.coherence_length * self._len_m * j0 / sigma_si

    def field_scale(self, field_units: str) -> float:
        """A_scale of solver.py:176-180: multiply A in [field_units * length_units] by this."""
        return _unit(FIELD_UNITS, field_units, "field") / (self.Bc2 * self.layer.coherence_length)

    def current_scale(self, current_units: str) -> float:
        """J_scale of solver.py:251-253."""
        return 4 * (_unit(CURRENT_UNITS, current_units, "current") / self._len_m) / self.K0

    # -- geometry ---------------------------------------------------------------------------
    @property
    def polygons(self):
        return (self.film,) + tuple(self.holes) + self.terminals

    def contains_points(self, points, index: bool = False, radius: float = 0):
        mask = self.film.contains_points(points, radius=radius)
        for hole in self.holes:
            mask = mask & ~hole.contains_points(points, radius=-radius)
        return np.where(mask)[0] if index else mask

    @property
    def points(self):
        return None if self.mesh is None else self.mesh.sites * self.layer.coherence_length

    @property
    def triangles(self):
        return None if self.mesh is None else self.mesh.elements

    @property
    def edges(self):
        return None if self.mesh is None else self.mesh.edge_mesh.edges

    @property
    def edge_lengths(self):
        if self.mesh is None:
            return None
        return self.mesh.edge_mesh.edge_lengths * self.layer.coherence_length

    @property
    def areas(self):
        return None if self.mesh is None else self.mesh.areas * self.layer.coherence_length**2

    @property
    def probe_point_indices(self):
        if self.mesh is None or self.probe_points is None:
            return None
        xi = self.layer.coherence_length
        return [self.mesh.closest_site(xy) for xy in self.probe_points / xi]

    # -- meshing ----------------------------------------------------------------------------
    def make_mesh(
        self,
        max_edge_length: Union[float, None] = None,
        min_points: Union[float, None] = None,
        smooth: int = 0,
        seed: int = 0,
        **unused,
    ) -> None:
        """Mesh the film (`Device.make_mesh`, device.py:520-566).

        Rectangular, hole-free films get a jittered triangular lattice; any other polygon
        (holes included) a boundary-conforming Delaunay mesh of such a lattice
        (`meshgen.polygon_mesh`).  Edges do not exceed ``max_edge_length`` (default: one
        coherence length, as in the reference).  ``smooth`` > 0 applies that many iterations of
        the reference's Laplacian smoothing (`Mesh.smooth`) before the dual mesh is built.  A
        mesh made elsewhere can be installed with :meth:`mesh_from_triangulation`.
        """
        if max_edge_length is None or max_edge_length <= 0:
            max_edge_length = 1.0 * self.layer.coherence_length
        if self.holes or not self.film.is_rectangle():
            # general polygon (with holes): boundary-conforming Delaunay mesh
            while True:
                pts, tri = polygon_mesh(self.film.points, [h.points for h in self.holes], max_edge_length, seed=seed)
                if min_points is None or len(pts) >= min_points:
                    break
                max_edge_length *= 0.9
            self.mesh_from_triangulation(pts, tri, smooth=smooth)
            return
        (x0, y0), (x1, y1) = self.film.bbox
        width, height = x1 - x0, y1 - y0
        while True:
            pts, tri = rectangle_mesh_points(width, height, max_edge_length, seed=seed)
            if min_points is None or len(pts) >= min_points:
                break
            max_edge_length *= 0.9
        pts = pts + np.array([[0.5 * (x0 + x1), 0.5 * (y0 + y1)]])
        self.mesh_from_triangulation(pts, tri, smooth=smooth)

    def mesh_from_triangulation(self, points, triangles, smooth: int = 0) -> None:
        """Install a mesh given in ``length_units`` (`_create_dimensionless_mesh`,
        device.py:568-583), optionally after ``smooth`` Laplacian smoothing iterations
        (device.py:552-556)."""
        points = np.asarray(points, dtype=float)
        triangles = np.asarray(triangles)
        if smooth:
            points = Mesh.from_triangulation(points, triangles, create_submesh=False).smooth(
                smooth, create_submesh=False).sites
        self.mesh = Mesh.from_triangulation(
            points / self.layer.coherence_length, triangles, create_submesh=True
        )

    # -- terminals --------------------------------------------------------------------------
    def terminal_info(self) -> Tuple[TerminalInfo, ...]:
        """One ``TerminalInfo`` per terminal, sorted by length (device.py:221-256): boundary
        sites inside the terminal polygon, boundary edges whose CENTRES are inside it, and the
        summed physical length of those edges."""
        if self.mesh is None:
            raise ValueError("The device has no mesh: call make_mesh() first.")
        xi = self.layer.coherence_length
        mesh = self.mesh
        sites = self.points
        edge_pos = xi * mesh.edge_mesh.centers
        ix_boundary = mesh.edge_mesh.boundary_edge_indices
        boundary_lengths = self.edge_lengths[ix_boundary]
        boundary_pos = edge_pos[ix_boundary]
        info = []
        for term in self.terminals:
            site_idx = np.intersect1d(term.contains_points(sites, index=True), mesh.boundary_indices)
            edge_idx = np.intersect1d(term.contains_points(edge_pos, index=True), ix_boundary)
            bpos = term.contains_points(boundary_pos, index=True)
            info.append(
                TerminalInfo(term.name, site_idx, edge_idx, bpos, boundary_lengths[bpos].sum())
            )
        return tuple(sorted(info, key=attrgetter("length")))

    def copy(self, with_mesh: bool = True) -> "Device":
        dev = Device(
            self.name,
            layer=self.layer.copy(),
            film=self.film.copy(),
            holes=[h.copy() for h in self.holes],
            terminals=[t.copy() for t in self.terminals],
            probe_points=None if self.probe_points is None else self.probe_points.copy(),
            length_units=self.length_units,
        )
        if with_mesh:
            dev.mesh = self.mesh
        return dev

    def mesh_stats_dict(self):
        el, ar = self.edge_lengths, self.areas
        return dict(
            num_sites=len(self.mesh.sites) if self.mesh else None,
            num_elements=len(self.mesh.elements) if self.mesh else None,
            min_edge_length=None if el is None else el.min(),
            max_edge_length=None if el is None else el.max(),
            mean_edge_length=None if el is None else el.mean(),
            min_area=None if ar is None else ar.min(),
            max_area=None if ar is None else ar.max(),
            mean_area=None if ar is None else ar.mean(),
            coherence_length=self.layer.coherence_length,
            length_units=self.length_units,
        )

    def __repr__(self):
        return f"Device({self.name!r}, film={self.film!r}, terminals={[t.name for t in self.terminals]})"
