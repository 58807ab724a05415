"""HDF5 output in the reference's on-disk layout.

The reference streams every saved step into an HDF5 file through its ``DataHandler``
(`tdgl/solver/runner.py:104-183`) and its analysis / visualisation tools read that file back.
Here the saved steps live in memory (`Solution.saved_steps`); `write_solution_h5` lays them out
the same way so those tools can open the result:

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
