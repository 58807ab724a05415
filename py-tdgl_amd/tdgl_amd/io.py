"""HDF5 output in the reference's on-disk layout.

The reference streams every saved step into an HDF5 file through its ``DataHandler``
(`tdgl/solver/runner.py:24-183`) and its analysis / visualisation tools read that file back.
`DataHandler` below does the same for ``SolverOptions.output_file``: every ``save_every`` steps the
fields leave the GPU and are written at once (a run that dies later keeps everything saved so far),
next to the ``<file>.tmp`` latest-step file the reference's live monitor reads.  `write_solution_h5`
writes an in-memory `Solution` in the same layout:

    /mesh/{sites, elements, boundary_indices, areas, dual_sites}          mesh.py:345-368
    /mesh/edge_mesh/{centers, edges, boundary_edge_indices, directions,
                     edge_lengths, dual_edge_lengths}                     edge_mesh.py:94-105
    /applied_vector_potential, /epsilon      when static ("fixed values")  runner.py:151-156, solver.py:755-766
    /data/<k>  attrs: step, time, dt, timestamp                            runner.py:165-170
    /data/<k>/{psi, mu, supercurrent, normal_current,
               induced_vector_potential[, applied_vector_potential, epsilon]}
    /data/<k>/running_state/{dt[, mu, theta][, screening_iterations]}      runner.py:180-183

``running_state`` of save ``k`` holds the per-step scalars of the steps since save ``k - 1`` in a
zero-padded buffer of ``save_every`` columns, exactly as the reference's ``RunningState`` buffer
(runner.py:186-221) is exported (the first save carries an empty buffer).

h5py is not a dependency of the time-stepping core: it is imported here, on use.  The Voronoi
polygon lists of the reference's mesh group are not written; ``Mesh.from_hdf5`` then rebuilds
the mesh from ``sites`` / ``elements`` (mesh.py:395-399).
"""

from datetime import datetime

import numpy as np


def _require_h5py():
    try:
        import h5py
    except ImportError as exc:  # pragma: no cover - depends on the environment
        raise ImportError(
            "Writing HDF5 output (SolverOptions.output_file / Solution.to_hdf5) needs h5py, "
            "which is not installed; the results are available in memory as Solution.saved_steps."
        ) from exc
    return h5py


class RunningState:
    """The reference's buffer of per-step scalars between two saves (`tdgl/solver/runner.py:186-221`):
    ``values[name]`` is ``[size, buffer_size]``, column ``step`` receives the next step's values, the
    rest stays zero.  Filled here from the per-step arrays a batch of `tdgl_run` steps returns."""

    def __init__(self, names_and_sizes, buffer_size: int):
        self.names_and_sizes = dict(names_and_sizes)
        self.buffer_size = int(buffer_size)
        self.clear()

    def clear(self) -> None:
        self.step = 0
        self.values = {name: np.zeros((size, self.buffer_size)) for name, size in self.names_and_sizes.items()}

    def append(self, name: str, value) -> None:
        self.values[name][:, self.step] = value

    def extend(self, columns) -> None:
        """``columns[name]``: array ``[k]`` or ``[k, size]`` for the next k steps (k may be 0)."""
        k = len(next(iter(columns.values()))) if columns else 0
        for j in range(k):
            for name, arr in columns.items():
                self.values[name][:, self.step] = np.asarray(arr[j]).reshape(-1)
            self.step += 1

    def export(self):
        return {name: v.copy() for name, v in self.values.items()}


def _h5py_factory(path, mode, **kw):
    return _require_h5py().File(path, mode, **kw)


def open_h5(path, mode="r"):
    """Open an output file for reading through the same factory the writer uses (h5py by default)."""
    return _h5py_factory(path, mode)


def read_mesh(group):
    """`Mesh.from_hdf5` (mesh.py:371-400) for the layout `write_mesh` produces: the dual mesh is
    rebuilt from ``sites`` / ``elements``."""
    from .finite_volume import Mesh

    if not ("sites" in group and "elements" in group):
        raise IOError("Could not load mesh due to missing data.")
    return Mesh.from_triangulation(np.array(group["sites"]).squeeze(), np.array(group["elements"], dtype=np.int64))


class DataHandler:
    """Streams the saved steps to disk in the reference's layout (`tdgl/solver/runner.py:24-183`).

    ``file_factory(path, mode, **kw)`` opens a file object with the h5py ``File`` interface
    (``create_group``, item assignment, ``attrs``, ``close``); the default is ``h5py.File``.  As in
    the reference an existing output file is never overwritten: the name gets a serial number."""

    def __init__(self, output_file, logger=None, file_factory=None):
        import logging

        self.logger = logger if logger is not None else logging.getLogger(__name__)
        self._base_output_file = output_file
        self._factory = file_factory or _h5py_factory
        self.output_file = self.tmp_file = None
        self.output_path = self.tmp_path = None
        self.time_step_group = self.mesh_group = None
        self.save_number = 0
        self._swmr = False

    def _create_output_file(self, output):
        import os
        from pathlib import Path

        Path(output).parent.mkdir(parents=True, exist_ok=True)
        parts = str(output).split(".")
        name, suffix = ".".join(parts[:-1]), parts[-1]
        serial = None
        while True:
            file_name = f"{name}{'' if serial is None else f'-{serial}'}.{suffix}"
            path = os.path.join(os.getcwd(), file_name)
            tmp_path = path + ".tmp"
            try:
                if os.path.exists(path) or os.path.exists(tmp_path):
                    raise FileExistsError(path)
                f = self._factory(path, "x")
                tmp = self._factory(tmp_path, "x", libver="latest")
            except (OSError, FileExistsError):
                serial = 1 if serial is None else serial + 1
                if serial > 10000:
                    raise
                continue
            if serial is not None:
                self.logger.warning(f"Output file already exists. Renaming to {file_name}.")
            return f, path, tmp, tmp_path

    def __enter__(self):
        self.output_file, self.output_path, self.tmp_file, self.tmp_path = self._create_output_file(
            self._base_output_file)
        self.time_step_group = self.output_file.create_group("data", track_order=True)
        grp = self.tmp_file.create_group("data/-1")
        grp["step"] = np.array([0])
        grp["time"] = np.array([0.0])
        grp["dt"] = np.array([0.0])
        return self

    def __exit__(self, exc_type, exc_value, tb):
        self.close()

    def close(self) -> None:
        import os

        if self.output_file is not None:
            self.output_file.close()
            self.output_file = None
        if self.tmp_file is not None:
            self.tmp_file.flush()
            self.tmp_file.close()
            self.tmp_file = None
            if self.tmp_path is not None and os.path.exists(self.tmp_path):
                os.remove(self.tmp_path)

    def save_mesh(self, mesh) -> None:
        self.mesh_group = self.output_file.create_group("mesh")
        write_mesh(self.mesh_group, mesh)

    def save_fixed_values(self, fixed_data) -> None:
        for key, value in fixed_data.items():
            self.output_file[key] = value
            self.tmp_file[key] = value

    def save_time_step(self, state, data, running_state) -> None:
        """runner.py:155-183: group ``data/<k>`` with the state as attributes, one dataset per field,
        the running-state buffers squeezed; the same fields overwrite ``data/-1`` of the tmp file."""
        group = self.time_step_group.create_group(f"{self.save_number}")
        group.attrs["timestamp"] = datetime.now().isoformat()
        self.save_number += 1
        tmp_grp = self.tmp_file["data/-1"]
        for key, value in state.items():
            group.attrs[key] = value
        for key, value in data.items():
            value = np.asarray(value)
            group[key] = value
            if key in tmp_grp:
                tmp_grp[key][:] = value
            else:
                tmp_grp[key] = value
            tmp_grp[key].flush()
        for key in ("step", "time", "dt"):
            tmp_grp[key][:] = np.array([state[key]])
            tmp_grp[key].flush()
        if running_state is not None:
            running_grp = group.create_group("running_state")
            for key, value in running_state.items():
                running_grp[key] = np.squeeze(np.asarray(value))
        # runner.py:402-403: once the first step is saved the latest-step file goes into single-writer /
        # multiple-reader mode, which is what lets the reference's live monitor open it (swmr=True) while
        # the run is in progress.  (File objects without the attribute skip it.)
        if not self._swmr and hasattr(self.tmp_file, "swmr_mode"):
            self.tmp_file.swmr_mode = True
        self._swmr = True


def write_solution_group(file, solution) -> None:
    """The ``/solution`` group the reference's ``Solution.from_hdf5`` starts from
    (`tdgl/solution/solution.py:874-931`): options, units, the (pickled) callables, the device."""
    import dataclasses

    if "solution" in file:
        del file["solution"]
    group = file.create_group("solution")
    options_grp = group.create_group("options")
    for k, v in dataclasses.asdict(solution.options).items():
        if k == "sparse_solver":
            v = getattr(v, "value", v)
        if v is not None:
            options_grp.attrs[k] = v
    group.attrs["time_created"] = getattr(solution, "time_created", datetime.now()).isoformat()
    group.attrs["current_units"] = solution.options.current_units
    group.attrs["field_units"] = solution.options.field_units
    for name in ("applied_vector_potential", "terminal_currents", "disorder_epsilon"):
        func = getattr(solution, name)
        try:
            if func is None or isinstance(func, (int, float, str, np.ndarray)):
                group.attrs[name] = "None" if func is None else func
            else:
                raise TypeError
        except TypeError:
            import cloudpickle

            group[f"{name}.pickle"] = np.void(cloudpickle.dumps(func))
    group.attrs["total_seconds"] = float(solution.total_seconds)
    if hasattr(solution.device, "to_hdf5"):
        solution.device.to_hdf5(group.create_group("device"), save_mesh=True)


def write_mesh(group, mesh) -> None:
    group["sites"] = mesh.sites
    group["elements"] = mesh.elements
    group["boundary_indices"] = mesh.boundary_indices
    group["areas"] = mesh.areas
    em = group.create_group("edge_mesh")
    e = mesh.edge_mesh
    em["centers"] = e.centers
    em["edges"] = e.edges
    em["boundary_edge_indices"] = e.boundary_edge_indices
    em["directions"] = e.directions
    em["edge_lengths"] = e.edge_lengths
    em["dual_edge_lengths"] = e.dual_edge_lengths
    if mesh.dual_sites is not None:
        group["dual_sites"] = mesh.dual_sites


def _running_state_buffers(solution):
    """Per save: ``{name: array[size, save_every]}`` as `RunningState.export` would give."""
    dyn, saves = solution.dynamics, solution.saved_steps
    every = int(solution.options.save_every)
    n_dyn = 0 if dyn is None else len(dyn.dt)
    names = {"dt": None if dyn is None else np.asarray(dyn.dt)[None, :]}
    if dyn is not None and dyn.mu is not None:
        names["mu"], names["theta"] = np.asarray(dyn.mu), np.asarray(dyn.theta)
    if dyn is not None and dyn.screening_iterations is not None:
        names["screening_iterations"] = np.asarray(dyn.screening_iterations, dtype=float)[None, :]
    out = []
    prev_step = 0
    for s in saves:
        # a save at a multiple of save_every holds the steps before it; the final save after the
        # loop (runner.py:452-453) also holds the step that ended the loop
        final_partial = int(s.step) % every != 0
        lo, hi = prev_step, min(int(s.step) + (1 if final_partial else 0), n_dyn)
        bufs = {}
        for name, arr in names.items():
            size = 1 if arr is None else arr.shape[0]
            buf = np.zeros((size, every))
            if arr is not None and hi > lo:
                buf[:, : hi - lo] = arr[:, lo:hi]
            bufs[name] = buf
        out.append(bufs)
        prev_step = int(s.step)
    return out


def write_solution_h5(solution, file, dynamic_vector_potential=False, dynamic_epsilon=False) -> None:
    """Write ``solution`` into ``file``: a path (opened with h5py) or an open h5py-like group."""
    if isinstance(file, (str, bytes)) or hasattr(file, "__fspath__"):
        h5py = _require_h5py()
        with h5py.File(file, "x") as f:
            write_solution_h5(solution, f, dynamic_vector_potential, dynamic_epsilon)
        return
    write_mesh(file.create_group("mesh"), solution.device.mesh)
    last = solution.saved_steps[-1]
    if not dynamic_vector_potential and last.applied_vector_potential is not None:
        file["applied_vector_potential"] = last.applied_vector_potential
    if not dynamic_epsilon and last.epsilon is not None:
        file["epsilon"] = last.epsilon
    try:
        data = file.create_group("data", track_order=True)
    except TypeError:  # a plain dict-like stand-in
        data = file.create_group("data")
    m = len(solution.device.mesh.edge_mesh.edges)
    for k, (s, running) in enumerate(zip(solution.saved_steps, _running_state_buffers(solution))):
        g = data.create_group(str(k))
        g.attrs["timestamp"] = datetime.now().isoformat()
        g.attrs["step"], g.attrs["time"], g.attrs["dt"] = int(s.step), float(s.time), float(s.dt)
        g["psi"], g["mu"] = s.psi, s.mu
        g["supercurrent"], g["normal_current"] = s.supercurrent, s.normal_current
        g["induced_vector_potential"] = (
            np.zeros((m, 2)) if s.induced_vector_potential is None else s.induced_vector_potential
        )
        if dynamic_vector_potential:
            g["applied_vector_potential"] = s.applied_vector_potential
        if dynamic_epsilon:
            g["epsilon"] = s.epsilon
        rs = g.create_group("running_state")
        for name, buf in running.items():
            rs[name] = np.squeeze(buf)
    import dataclasses

    if dataclasses.is_dataclass(solution.options):
        write_solution_group(file, solution)
