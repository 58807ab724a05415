"""Result of a simulation and its post-processing.

The reference's ``Solution`` (`tdgl/solution/solution.py:59-1090`) wraps the HDF5 file its
``DataHandler`` wrote.  This class carries what the solver produced -- the fields at every saved step
(without ``SolverOptions.output_file``) or at the last saved step (with it: the others were streamed
to ``path`` in the reference's layout, `tdgl_amd.io.DataHandler`, and are read back on demand by
:meth:`Solution.load_tdgl_data` / :meth:`Solution.from_hdf5`), the per-step scalars (``dt``, probe
``mu``/``theta``) and the configuration -- and the part of the reference's post-processing that
SURVEY.md section 8(f) rank 3 names: loading a solve step (solution.py:161-196), sheet currents
between the sites and through a path (:364-429, :623-667), the vector potential of the currents and
the fluxoid of a polygon (:464-548, :768-872), the dipole moment (:259-289), the field of the
currents (:669-766).  Plotting and the pint unit registry are not part of it: quantities are plain
floats / arrays in the units the docstrings state; ``with_units=True`` wraps them in a small
``Quantity`` (``.magnitude``, ``.units``).
"""

import numbers
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, NamedTuple, Optional, Union

import numpy as np


class Quantity(np.ndarray):
    """A value with the name of its unit: the two things code written for the reference's pint
    quantities reads off a result (``.magnitude``, ``.units``).  No unit algebra."""

    def __new__(cls, value, units: str):
        obj = np.asarray(value).view(cls)
        obj.units = units
        return obj

    def __array_finalize__(self, obj):
        self.units = getattr(obj, "units", None)

    @property
    def magnitude(self):
        plain = np.asarray(self)
        return plain.item() if plain.ndim == 0 else plain


class Fluxoid(NamedTuple):
    """`tdgl/fluxoid.py`: the two parts of a fluxoid; ``sum(fluxoid)`` is the total."""

    flux_part: Any
    supercurrent_part: Any


class BiotSavartField(NamedTuple):
    supercurrent: Any
    normal_current: Any


def get_data_range(h5file):
    """Smallest and largest ``data/<k>`` in an output file (`tdgl/solution/data.py:14-17`)."""
    keys = np.asarray([int(key) for key in h5file["data"]])
    return int(keys.min()), int(keys.max())


@dataclass
class TDGLData:
    """Fields at one saved step (cf. `tdgl/solution/data.py:69-170`)."""

    step: int
    time: float
    dt: float
    psi: np.ndarray
    mu: np.ndarray
    supercurrent: np.ndarray
    normal_current: np.ndarray
    applied_vector_potential: Optional[np.ndarray] = None
    epsilon: Optional[np.ndarray] = None
    induced_vector_potential: Optional[np.ndarray] = None  # include_screening only
    state: Optional[Dict[str, Any]] = None  # the group's attributes (step, time, dt, timestamp)

    @staticmethod
    def from_hdf5(h5file, step: int) -> "TDGLData":
        """Group ``data/<step>`` of an output file; static fields (``applied_vector_potential``,
        ``epsilon``) sit at the top level (`tdgl/solution/data.py:95-125`).  ``step`` here is the
        solver iteration stored in the group's attributes (the reference keeps the group number)."""
        group = h5file["data"][str(step)]
        state = dict(group.attrs)

        def get(key):
            if key in h5file:
                return np.array(h5file[key])
            if key in group:
                return np.array(group[key])
            return None

        return TDGLData(
            step=int(state.get("step", step)), time=float(state.get("time", 0.0)), dt=float(state.get("dt", 0.0)),
            psi=get("psi"), mu=get("mu"), supercurrent=get("supercurrent"), normal_current=get("normal_current"),
            applied_vector_potential=get("applied_vector_potential"), epsilon=get("epsilon"),
            induced_vector_potential=get("induced_vector_potential"), state=state,
        )


@dataclass
class DynamicsData:
    """Per-step scalars of the saved stage (cf. `tdgl/solution/data.py:145-330`)."""

    dt: np.ndarray
    time: Optional[np.ndarray] = None
    mu: Optional[np.ndarray] = None      # [n_probe, n_steps]
    theta: Optional[np.ndarray] = None   # [n_probe, n_steps]
    pcg_iterations: Optional[np.ndarray] = None
    screening_iterations: Optional[np.ndarray] = None  # include_screening only

    def __post_init__(self):
        if self.time is None:  # data.py:167-168
            self.time = np.cumsum(self.dt)

    def time_slice(self, tmin: float = -np.inf, tmax: float = np.inf) -> np.ndarray:
        (indices,) = np.where((self.time >= tmin) & (self.time <= tmax))
        return indices

    def closest_time(self, time: float) -> int:
        return int(np.argmin(np.abs(self.time - time)))

    def voltage(self, i: int = 0, j: int = 1) -> np.ndarray:
        """mu_i - mu_j between two probe points, per step (data.py:195-210)."""
        if self.mu is None:
            raise ValueError("No voltage data available.")
        if self.mu.shape[0] == 1:
            raise ValueError("The solution has only one probe point.")
        return self.mu[i] - self.mu[j]

    def phase_difference(self, i: int = 0, j: int = 1) -> np.ndarray:
        if self.theta is None:
            raise ValueError("No phase data available.")
        if self.theta.shape[0] == 1:
            raise ValueError("The solution has only one probe point.")
        return self.theta[i] - self.theta[j]  # (not unwrapped, as in data.py:212-228)

    def mean_voltage(self, i: int = 0, j: int = 1, tmin: float = -np.inf, tmax: float = np.inf) -> float:
        """Time average of the voltage weighted with dt (data.py:230-253)."""
        if self.mu is None:
            raise ValueError("No voltage data available.")
        indices = self.time_slice(tmin, tmax)
        return float(np.average(self.voltage(i, j)[indices], weights=self.dt[indices]))

    def resample(self, num_points: Union[int, None] = None) -> "DynamicsData":
        """Linear interpolation onto a uniform time grid (data.py:255-273)."""
        time = self.time
        if num_points is None:
            num_points = len(time)
        ts = np.linspace(time.min(), time.max(), num_points)
        mu = None if self.mu is None else np.array([np.interp(ts, time, val) for val in self.mu])
        theta = None if self.theta is None else np.array([np.interp(ts, time, val) for val in self.theta])
        return DynamicsData(dt=(ts[1] - ts[0]) * np.ones_like(ts), mu=mu, theta=theta)

    @staticmethod
    def from_hdf5(h5file, step_min: Union[int, None] = None, step_max: Union[int, None] = None) -> "DynamicsData":
        """From ``DynamicsData.to_hdf5`` output, or from the ``running_state`` buffers of the saved
        steps of an output file: concatenated, the zero padding (dt = 0) removed (data.py:369-428)."""
        if "theta" in h5file or ("dt" in h5file and "data" not in h5file):
            get = lambda k: np.array(h5file[k]) if k in h5file else None  # noqa: E731
            return DynamicsData(dt=get("dt"), mu=get("mu"), theta=get("theta"),
                                screening_iterations=get("screening_iterations"))
        if step_min is None:
            step_min, step_max = get_data_range(h5file)
        cols: Dict[str, list] = {"dt": [], "mu": [], "theta": [], "screening_iterations": []}
        for i in range(step_min, step_max + 1):
            grp = h5file[f"data/{i}"]
            if "running_state" not in grp:
                continue
            grp = grp["running_state"]
            for name in cols:
                if name in grp:
                    cols[name].append(np.array(grp[name]))
        dt = np.concatenate([np.atleast_1d(d) for d in cols["dt"]])
        mask = dt > 0
        two_d = lambda parts: np.concatenate([np.atleast_2d(p) for p in parts], axis=1)[..., mask]  # noqa: E731
        return DynamicsData(
            dt=dt[mask],
            mu=two_d(cols["mu"]) if cols["mu"] else None,
            theta=two_d(cols["theta"]) if cols["theta"] else None,
            screening_iterations=(np.concatenate([np.atleast_1d(p) for p in cols["screening_iterations"]])[mask]
                                  if cols["screening_iterations"] else None),
        )

    def to_hdf5(self, h5group) -> None:
        h5group["dt"] = self.dt
        for name in ("mu", "theta", "screening_iterations"):
            if getattr(self, name) is not None:
                h5group[name] = getattr(self, name)


def _split_units(units: str, sep: str):
    parts = [p.strip() for p in units.replace("**", "^").split(sep)]
    return parts


@dataclass
class Solution:
    device: object
    options: object
    saved_steps: List[TDGLData] = field(default_factory=list)
    dynamics: Optional[DynamicsData] = None
    applied_vector_potential: object = None
    terminal_currents: object = None
    disorder_epsilon: object = None
    total_seconds: float = 0.0
    stats: Dict[str, float] = field(default_factory=dict)
    solve_step: int = -1
    dynamic_vector_potential: bool = False
    dynamic_epsilon: bool = False
    path: Optional[str] = None                 # the streamed HDF5 file (SolverOptions.output_file)
    saved_step_index: Optional[list] = None    # streaming: (step, time) of every group data/<k> on disk
    data_range: Optional[tuple] = None         # (first, last) group number on disk, once loaded
    _loaded: Optional[TDGLData] = field(default=None, repr=False)
    _interp: object = field(default=None, repr=False)

    # -- where the data lives ----------------------------------------------------------------------
    @property
    def saved_on_disk(self) -> bool:
        """solution.py:119-122."""
        return self.path is not None and os.path.exists(self.path)

    @property
    def field_units(self) -> str:
        return self.options.field_units

    @property
    def current_units(self) -> str:
        return self.options.current_units

    def to_hdf5(self, file) -> None:
        """Write the saved steps in the reference's DataHandler layout (`tdgl_amd.io`); ``file``
        is a path (needs h5py) or an open h5py-like group."""
        from .io import write_solution_h5

        write_solution_h5(self, file, self.dynamic_vector_potential, self.dynamic_epsilon)

    @property
    def tdgl_data(self) -> TDGLData:
        """The loaded solve step: by default the last saved one."""
        if self._loaded is not None:
            return self._loaded
        return self.saved_steps[self.solve_step]

    @property
    def times(self) -> np.ndarray:
        """Times of the saved steps (solution.py:137-148)."""
        if self.saved_step_index is not None:
            return np.array([t for _, t in self.saved_step_index])
        return np.array([s.time for s in self.saved_steps])

    def closest_solve_step(self, time: float) -> int:
        """Index of the saved step whose time is closest to ``time`` (solution.py:150-159)."""
        return int(np.argmin(np.abs(self.times - time)))

    def load_tdgl_data(self, solve_step: int = -1, h5file=None) -> None:
        """Make saved step ``solve_step`` the current one (solution.py:161-196): 0 is the first
        saved step, negative numbers count from the last.  Steps that were streamed to
        ``output_file`` are read back from it (or from the open ``h5file`` given)."""
        if h5file is None and not self.saved_on_disk:
            count = len(self.saved_steps)
            index = solve_step if solve_step >= 0 else count + solve_step
            if not 0 <= index < count:
                raise IndexError(f"solve_step {solve_step} out of range: {count} saved steps in memory.")
            self.solve_step, self._loaded = index, None
            return
        from .io import open_h5

        f = h5file if h5file is not None else open_h5(self.path, "r")
        try:
            self.data_range = step_min, step_max = get_data_range(f)
            if solve_step == 0:
                step = step_min
            elif solve_step < 0:
                step = step_max + 1 + solve_step
            else:
                step = solve_step
            self._loaded = TDGLData.from_hdf5(f, step)
            self.dynamics = DynamicsData.from_hdf5(f, step_min, step_max)
            if self.saved_step_index is None:
                self.saved_step_index = [
                    (int(f[f"data/{k}"].attrs["step"]), float(f[f"data/{k}"].attrs["time"]))
                    for k in range(step_min, step_max + 1)
                ]
            self.solve_step = step
        finally:
            if h5file is None:
                f.close()

    @staticmethod
    def from_hdf5(path, solve_step: int = -1) -> "Solution":
        """Load a solution from an output file (solution.py:957-999): the ``/solution`` group
        (options, units, inputs, device) + the requested solve step."""
        import cloudpickle

        from .device import Device
        from .io import open_h5
        from .options import SolverOptions

        f = path if hasattr(path, "create_group") else open_h5(path, "r")
        try:
            grp = f["solution"]

            def load(name):
                if name in grp.attrs:
                    value = grp.attrs[name]
                    return None if isinstance(value, str) and value == "None" else value
                if f"{name}.pickle" in grp:
                    return cloudpickle.loads(np.void(np.array(grp[f"{name}.pickle"])).tobytes())
                raise IOError(f"Unable to load {name}.")

            # (HDF5 attributes come back as NumPy scalars: numpy.bool_, numpy.int64, ...)
            options = SolverOptions(**{k: (v.item() if isinstance(v, np.generic) else v)
                                       for k, v in dict(grp["options"].attrs).items()})
            options.validate()
            solution = Solution(
                device=Device.from_hdf5(grp["device"]), options=options,
                applied_vector_potential=load("applied_vector_potential"), terminal_currents=load("terminal_currents"),
                disorder_epsilon=load("disorder_epsilon"), total_seconds=float(grp.attrs["total_seconds"]),
                path=None if hasattr(path, "create_group") else os.fspath(path),
            )
            solution.time_created = grp.attrs["time_created"]
            solution.load_tdgl_data(solve_step, h5file=f)
        finally:
            if f is not path:
                f.close()
        return solution

    def delete_hdf5(self) -> None:
        """solution.py:1001-1004."""
        if self.saved_on_disk:
            os.remove(self.path)
            self.path = None

    # -- sheet current densities on the sites, in current_units / length_units -------------------
    def _k0(self) -> float:
        """K0 in ``current_units / length_units`` (`tdgl/solution/solution.py:192`)."""
        from .device import CURRENT_UNITS, LENGTH_UNITS

        dev = self.device
        return dev.K0 * LENGTH_UNITS[dev.length_units] / CURRENT_UNITS[self.options.current_units]

    def _site_vector(self, quantity_on_edges) -> np.ndarray:
        # magnitude * unit direction of the edge->site average (`tdgl/solution/data.py:48-65`)
        return self.device.mesh.get_quantity_on_site(quantity_on_edges)

    @property
    def supercurrent_density(self) -> np.ndarray:
        """K_s on the sites, shape (n, 2) (`tdgl/solution/solution.py:186-194`)."""
        return self._k0() * self._site_vector(self.tdgl_data.supercurrent)

    @property
    def normal_current_density(self) -> np.ndarray:
        return self._k0() * self._site_vector(self.tdgl_data.normal_current)

    @property
    def current_density(self) -> np.ndarray:
        """Total sheet current density K = K_s + K_n (`tdgl/solution/solution.py:230-237`)."""
        return self.supercurrent_density + self.normal_current_density

    def current_through_cut(self, x0: float, physical: bool = False) -> float:
        """Total sheet current crossing the vertical line x = x0, summed over the mesh edges that
        cross it: ``sum_e (J_s + J_n)_e s_e sign`` with ``s_e`` the Voronoi dual length.  Discretely
        conserved.  By default dimensionless with ``x0`` in units of
        xi; ``physical=True``: ``x0`` in ``length_units``, result in ``options.current_units``."""
        if physical:
            xi = self.device.coherence_length
            scale = self.device.current_scale(self.options.current_units)
            return self.current_through_cut(x0 / xi) / scale * xi
        mesh = self.device.mesh
        em = mesh.edge_mesh
        d = self.tdgl_data
        xa = mesh.sites[em.edges[:, 0], 0]
        xb = mesh.sites[em.edges[:, 1], 0]
        crossing = (xa < x0) != (xb < x0)
        sign = np.where(xa < x0, 1.0, -1.0)
        j = (d.supercurrent + d.normal_current) * em.dual_edge_lengths * sign
        return float(j[crossing].sum())

    # -- between the sites --------------------------------------------------------------------------
    def _interpolator(self, method: str):
        valid_methods = ("linear", "cubic")
        if method not in valid_methods:
            raise ValueError(f"Interpolation method must be one of {valid_methods} (got {method}).")
        if method == "cubic":
            raise NotImplementedError(
                "Cubic interpolation (matplotlib's CubicTriInterpolator in the reference) is not provided; "
                "use method='linear'.")
        if self._interp is None:
            from .triinterp import TriLinearInterpolator

            mesh = self.device.mesh
            self._interp = TriLinearInterpolator(self.device.coherence_length * mesh.sites, mesh.elements)
        return self._interp

    def _density_factor(self, units: Union[str, None]) -> float:
        """Conversion from ``current_units / length_units`` to ``units`` ("<current> / <length>")."""
        from .device import CURRENT_UNITS, LENGTH_UNITS, _unit

        if units is None:
            return 1.0
        parts = _split_units(units, "/")
        if len(parts) != 2:
            raise ValueError(f"Expected current density units like 'uA / um' (got {units!r}).")
        have = CURRENT_UNITS[self.current_units] / LENGTH_UNITS[self.device.length_units]
        want = _unit(CURRENT_UNITS, parts[0], "current") / _unit(LENGTH_UNITS, parts[1], "length")
        return have / want

    def interp_current_density(self, positions: np.ndarray, *, dataset: Union[str, None] = None,
                               method: str = "linear", units: Union[str, None] = None,
                               with_units: bool = False) -> np.ndarray:
        """Sheet current density at arbitrary points (solution.py:364-429): piecewise-linear between
        the site values; zero outside the film and inside holes.  ``dataset``: ``None`` (total),
        ``"supercurrent"`` or ``"normal_current"``.  In ``units`` (default
        ``current_units / length_units``)."""
        if dataset is None:
            J = self.current_density
        elif dataset == "supercurrent":
            J = self.supercurrent_density
        elif dataset == "normal_current":
            J = self.normal_current_density
        else:
            raise ValueError(f"Unexpected dataset: {dataset}.")
        interp = self._interpolator(method)
        positions = np.atleast_2d(positions)
        J = interp(J * self._density_factor(units), positions)
        J[~np.isfinite(J).all(axis=1)] = 0
        J[~self.device.contains_points(positions)] = 0
        if with_units:
            return Quantity(J, units or f"{self.current_units} / {self.device.length_units}")
        return J

    def interp_order_parameter(self, positions: np.ndarray, method: str = "linear") -> np.ndarray:
        """psi at arbitrary points (solution.py:431-462); NaN outside the mesh."""
        interp = self._interpolator(method)
        return interp(self.tdgl_data.psi, np.atleast_2d(positions))

    def current_through_path(self, path_coords: np.ndarray, dataset: Union[str, None] = None,
                             method: str = "linear", units: Union[str, None] = None, with_units: bool = True):
        """Total current crossing a path (solution.py:623-667): the interpolated current density,
        averaged over each path segment, dotted into the segment's normal ``(dy, -dx)/|d|``, times the
        segment length, for the segments whose centres lie in the device, accumulated with the
        trapezoid rule as the reference does (``np.trapz`` of the per-segment currents).  In
        ``units`` (default ``current_units``)."""
        from .device import CURRENT_UNITS, _unit
        from .geometry import path_vectors

        path_coords = np.asarray(path_coords, dtype=float)
        J = self.interp_current_density(path_coords, dataset=dataset, method=method)
        centres = (path_coords[:-1] + path_coords[1:]) / 2
        J_edge = (J[:-1] + J[1:]) / 2
        lengths, normals = path_vectors(path_coords)
        J_dot_n = (J_edge * normals).sum(axis=1)
        inside = self.device.contains_points(centres)
        y = (J_dot_n * lengths)[inside]
        total = float(np.sum((y[1:] + y[:-1]) / 2)) if len(y) > 1 else 0.0
        if units is not None:
            total *= CURRENT_UNITS[self.current_units] / _unit(CURRENT_UNITS, units, "current")
        return Quantity(total, units or self.current_units) if with_units else total

    def magnetic_moment(self, units: Union[str, None] = None, with_units: bool = True):
        """z component of the dipole moment, (1/2) sum_i (r_i - r_cm) x K_i a_i (solution.py:259-289),
        in ``current_units * length_units**2``."""
        if units is not None:
            raise NotImplementedError("magnetic_moment: unit conversion is not provided; pass units=None.")
        mesh, xi = self.device.mesh, self.device.coherence_length
        com = (mesh.sites * mesh.areas[:, None]).sum(axis=0) / mesh.areas.sum()
        r = xi * (mesh.sites - com[None, :])
        K = self.current_density
        m = float(np.sum(0.5 * (r[:, 0] * K[:, 1] - r[:, 1] * K[:, 0]) * mesh.areas * xi**2))
        return Quantity(m, f"{self.current_units} * {self.device.length_units}**2") if with_units else m

    # -- fields of the currents ----------------------------------------------------------------------
    def _positions_and_heights(self, positions, zs):
        positions = np.atleast_2d(np.asarray(positions, dtype=float))
        if positions.shape[1] == 3:
            if zs is not None:
                raise ValueError("If positions has shape (m, 3) then zs cannot be specified.")
            zs, positions = positions[:, 2], positions[:, :2]
        elif isinstance(zs, numbers.Real):
            zs = zs * np.ones(len(positions))
        if not isinstance(zs, np.ndarray):
            raise ValueError(f"Expected zs to be an ndarray, but got {type(zs)}.")
        return positions, np.asarray(zs, dtype=float).reshape(-1)

    def _applied_vector_potential_at(self, positions: np.ndarray, zs: np.ndarray) -> np.ndarray:
        """A_applied at points, [m, 3], in ``field_units * length_units``: the user's parameter, or
        for a plain number the uniform-field potential the solver used (gauge centre = centre of
        the evaluation points' bounding box, `tdgl/sources/constant.py:7-21`)."""
        from .solver import uniform_field_vector_potential

        A = self.applied_vector_potential
        x, y = positions[:, 0], positions[:, 1]
        if callable(A):
            kwargs = {}
            if getattr(A, "time_dependent", False):
                kwargs["t"] = float(self.tdgl_data.time)
            out = np.asarray(A(x, y, zs, **kwargs), dtype=float)
        else:
            out = uniform_field_vector_potential(x, y, float(A or 0.0))
        if out.shape[1] == 2:
            out = np.concatenate([out, np.zeros_like(out[:, :1])], axis=1)
        return out

    def vector_potential_at_position(self, positions: np.ndarray, *, zs=None, units: Union[str, None] = None,
                                     with_units: bool = True, return_sum: bool = True):
        """Applied vector potential plus the one of the sheet currents,
        ``A(r) = mu_0 / (4 pi) sum_j K_j a_j / |r - r_j|`` (solution.py:768-872), shape [m, 3], in
        ``field_units * length_units``.  ``return_sum=False``: a dict with the parts ``applied``,
        ``supercurrent_density``, ``normal_current_density``."""
        from .device import CURRENT_UNITS, FIELD_UNITS, LENGTH_UNITS, MU_0

        dev = self.device
        default = f"{self.field_units} * {dev.length_units}"
        if units is not None and units.replace(" ", "") != default.replace(" ", ""):
            raise NotImplementedError(f"vector_potential_at_position: only units={default!r} is provided.")
        positions, zs = self._positions_and_heights(positions, zs)
        wrap = (lambda a: Quantity(a, default)) if with_units else (lambda a: a)
        parts = {"applied": wrap(self._applied_vector_potential_at(positions, zs))}
        points = dev.points
        areas = dev.mesh.areas * dev.coherence_length**2
        d = positions[:, None, :] - points[None, :, :]
        rho = np.sqrt((d**2).sum(axis=2) + (zs[:, None] - dev.layer.z0) ** 2)
        # mu_0/(4 pi) [current_units] -> [field_units * length_units]:  T m = (N/A^2) A
        to_units = MU_0 / (4 * np.pi) * CURRENT_UNITS[self.current_units] / (
            FIELD_UNITS[self.field_units] * LENGTH_UNITS[dev.length_units])
        for name in ("supercurrent_density", "normal_current_density"):
            J = getattr(self, name)
            with np.errstate(divide="ignore", invalid="ignore"):
                Axy = (J[None, :, :] / rho[:, :, None] * areas[None, :, None]).sum(axis=1)
            parts[name] = wrap(to_units * np.concatenate([Axy, np.zeros_like(Axy[:, :1])], axis=1))
        if return_sum:
            return sum(parts.values())
        return parts

    def field_at_position(self, positions: np.ndarray, *, zs=None, vector: bool = False,
                          units: Union[str, None] = None, with_units: bool = True, return_sum: bool = True):
        """Magnetic field of the sheet currents (Biot-Savart over the sites' Voronoi cells,
        solution.py:669-766, `tdgl/em.py:252-330`), in ``field_units``; the z component, or with
        ``vector=True`` all three.  Not defined in the plane of the film inside the film."""
        from .device import CURRENT_UNITS, FIELD_UNITS, LENGTH_UNITS, MU_0

        dev = self.device
        if units is not None and units != self.field_units:
            raise NotImplementedError(f"field_at_position: only units={self.field_units!r} is provided.")
        positions, zs = self._positions_and_heights(positions, zs)
        dz = zs - dev.layer.z0
        if np.all(dz == 0) and dev.film.contains_points(positions).any():
            raise ValueError("Cannot interpolate fields within a film.")
        points = dev.points
        areas = dev.mesh.areas * dev.coherence_length**2
        dx = positions[:, None, 0] - points[None, :, 0]
        dy = positions[:, None, 1] - points[None, :, 1]
        r3 = (dx**2 + dy**2 + dz[:, None] ** 2) ** 1.5
        # mu_0/(4 pi) [current/length * length^2 / length^2] -> tesla -> field_units
        to_units = MU_0 / (4 * np.pi) * (CURRENT_UNITS[self.current_units] / LENGTH_UNITS[dev.length_units]) / (
            FIELD_UNITS[self.field_units])
        fields = []
        for name in ("supercurrent_density", "normal_current_density"):
            J = getattr(self, name)
            jx, jy = (J[:, 0] * areas)[None, :], (J[:, 1] * areas)[None, :]
            Hz = ((jx * dy - jy * dx) / r3).sum(axis=1)
            if vector:
                Hx = (jy * dz[:, None] / r3).sum(axis=1)
                Hy = (-jx * dz[:, None] / r3).sum(axis=1)
                H = np.stack([Hx, Hy, Hz], axis=1)
            else:
                H = Hz
            H = to_units * H
            fields.append(Quantity(H, self.field_units) if with_units else H)
        fields = BiotSavartField(*fields)
        return sum(fields) if return_sum else fields

    def polygon_fluxoid(self, polygon_points, interp_method: str = "linear", units: Union[str, None] = "Phi_0",
                        with_units: bool = True) -> Fluxoid:
        """Fluxoid of a closed polygon inside the film (solution.py:464-548):
        ``oint A . dl  +  mu_0 oint Lambda / |psi|^2  K_s . dl`` with A the applied plus the
        currents' vector potential, both line integrals as the reference accumulates them
        (``dl`` = backward differences of the closed, counter-clockwise vertex list; trapezoid rule
        over the per-vertex terms).  In ``units``: ``"Phi_0"`` (default) or ``None`` =
        ``field_units * length_units**2``."""
        from .device import CURRENT_UNITS, FIELD_UNITS, LENGTH_UNITS, MU_0, PHI_0, Polygon

        dev = self.device
        points = Polygon(points=polygon_points).points
        if not dev.film.contains_points(points).all():
            raise ValueError("The polygon must lie completely within the superconducting film.")
        native = FIELD_UNITS[self.field_units] * LENGTH_UNITS[dev.length_units] ** 2  # Wb per native unit
        if units is None:
            scale, units = 1.0, f"{self.field_units} * {dev.length_units} ** 2"
        elif units == "Phi_0":
            scale = native / PHI_0
        elif units == "Wb":
            scale = native
        else:
            raise NotImplementedError(f"polygon_fluxoid: units {units!r} not provided (use 'Phi_0', 'Wb' or None).")
        J_poly = self.interp_current_density(points, dataset="supercurrent", method=interp_method)
        zs = dev.layer.z0 * np.ones(len(points))
        dl = np.diff(points, axis=0, prepend=points[:1])
        A_poly = self.vector_potential_at_position(points, zs=zs, with_units=False)[:, :2]
        trapz = lambda y: float(np.sum((y[1:] + y[:-1]) / 2))  # noqa: E731
        flux_part = trapz((A_poly * dl).sum(axis=1)) * scale
        ns = np.abs(self.interp_order_parameter(points, method=interp_method)) ** 2
        Lambda = dev.layer.Lambda / ns
        int_J = trapz((Lambda[:, None] * J_poly * dl).sum(axis=1))  # [current_units * length_units]
        # mu_0 [current * length] = Wb
        supercurrent_part = MU_0 * int_J * CURRENT_UNITS[self.current_units] * LENGTH_UNITS[dev.length_units] / native * scale
        if with_units:
            return Fluxoid(Quantity(flux_part, units), Quantity(supercurrent_part, units))
        return Fluxoid(flux_part, supercurrent_part)
