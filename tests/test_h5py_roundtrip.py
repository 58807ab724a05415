"""SURVEY.md section 8(f) rank 3 against the real thing: the same writes the recorder tests check
(`tests/test_host_logic.py`, in-memory stand-in for h5py) go to an actual HDF5 file and come back.

h5py is on neither the build image nor the GPU box (no HDF5 library at all, no network), so this
module is SKIPPED there; it runs wherever `pip install h5py` has been done.  Until it has run somewhere
the HDF5 layout stays pinned only through the recorder -- README.md says so.  Where the reference
checkout is also present (/root/reference), the file is additionally opened with the REFERENCE's own
`Solution.from_hdf5` (tdgl/solution/solution.py:623-667).
"""

import os
import sys

import numpy as np
import pytest

h5py = pytest.importorskip("h5py")

from helpers import synthetic_mesh  # noqa: E402
from test_host_logic import _annulus_device  # noqa: E402


def _write_run(path, device, monkeypatch, tmp_path):
    import tdgl_amd as tdgl
    from tdgl_amd.io import DataHandler, RunningState, write_solution_group
    from tdgl_amd.solution import Solution

    monkeypatch.chdir(tmp_path)
    mesh = device.mesh
    n, m = len(mesh.sites), len(mesh.edge_mesh.edges)
    rng = np.random.default_rng(1)
    opts = tdgl.SolverOptions(solve_time=2.0, field_units="mT", current_units="uA", save_every=4, output_file=path)
    running = RunningState({"dt": 1, "mu": 2, "theta": 2}, 4)
    dts = 0.1 + 0.01 * np.arange(11)
    mu_p, th_p = rng.normal(size=(11, 2)), rng.normal(size=(11, 2))
    saved = {}
    with DataHandler(path) as h:
        assert isinstance(h.output_file, h5py.File) and h.tmp_file.libver == ("latest", "latest")
        h.save_mesh(mesh)
        h.save_fixed_values({"applied_vector_potential": np.ones((m, 2)), "epsilon": np.ones(n)})
        t = 0.0
        for i in range(11):
            if i % 4 == 0:
                saved[i // 4] = dict(psi=rng.normal(size=n) + 1j, mu=rng.normal(size=n), supercurrent=rng.normal(size=m),
                                     normal_current=rng.normal(size=m), induced_vector_potential=np.zeros((m, 2)))
                h.save_time_step(dict(step=i, time=t, dt=float(dts[i])), saved[i // 4], None if i == 0 else running.export())
                running.clear()
                # runner.py:402-403: the latest-step file is readable by a second process while the run goes on
                assert h.tmp_file.swmr_mode
                with h5py.File(h.tmp_path, "r", swmr=True, libver="latest") as live:
                    assert live["data/-1/step"][0] == i and np.array_equal(live["data/-1/psi"][:], saved[i // 4]["psi"])
            running.extend({"dt": dts[i:i + 1], "mu": mu_p[i:i + 1], "theta": th_p[i:i + 1]})
            t += dts[i]
        saved[3] = dict(saved[2], mu=saved[2]["mu"] + 1)
        h.save_time_step(dict(step=10, time=t, dt=float(dts[-1])), saved[3], running.export())
        stub = Solution(device=device, options=opts, applied_vector_potential=0.25,
                        terminal_currents={"a": 1.0, "b": -1.0}, disorder_epsilon=1.0, total_seconds=1.5)
        write_solution_group(h.output_file, stub)
        out_path, tmp = h.output_path, h.tmp_path
    assert os.path.exists(out_path) and not os.path.exists(tmp)
    return out_path, saved, dts, mu_p, th_p


def test_real_hdf5_file_has_the_reference_layout_and_reads_back(tmp_path, monkeypatch):
    """runner.py:104-183 (`/mesh`, `/data/<k>/...`, attrs, running_state buffers) written by
    `DataHandler` with h5py itself, inspected with h5py, and read back by `Solution.from_hdf5`."""
    from tdgl_amd.solution import Solution

    device = _annulus_device(pitch=0.25)
    mesh = device.mesh
    m = len(mesh.edge_mesh.edges)
    path, saved, dts, mu_p, th_p = _write_run("o.h5", device, monkeypatch, tmp_path)
    with h5py.File(path, "r") as f:
        assert set(f) == {"mesh", "data", "applied_vector_potential", "epsilon", "solution"}
        assert set(f["mesh"]) == {"sites", "elements", "boundary_indices", "areas", "edge_mesh", "dual_sites"}
        assert set(f["mesh/edge_mesh"]) == {"centers", "edges", "boundary_edge_indices", "directions", "edge_lengths",
                                            "dual_edge_lengths"}
        assert list(f["data"]) == ["0", "1", "2", "3"]  # creation order is tracked (runner.py:142)
        g2 = f["data/2"]
        assert set(g2) == {"psi", "mu", "supercurrent", "normal_current", "induced_vector_potential", "running_state"}
        assert set(g2.attrs) == {"timestamp", "step", "time", "dt"} and g2.attrs["step"] == 8
        assert g2["psi"].dtype == np.complex128 and g2["induced_vector_potential"].shape == (m, 2)
        assert np.array_equal(g2["running_state/dt"][:], dts[4:8]) and np.array_equal(g2["running_state/mu"][:], mu_p[4:8].T)
        assert np.array_equal(f["data/3/running_state/dt"][:], np.concatenate([dts[8:11], [0.0]]))
        assert "running_state" not in f["data/0"]
        assert f["solution"].attrs["field_units"] == "mT" and f["solution/options"].attrs["save_every"] == 4
    sol = Solution.from_hdf5(path)
    assert sol.data_range == (0, 3) and sol.options.save_every == 4 and type(sol.options.adaptive) is bool
    assert sol.options.pcg_precond_fp32 is True  # (numpy.bool_ attributes are cast back)
    assert sol.device.layer == device.layer and np.array_equal(sol.device.mesh.sites, mesh.sites)
    for k in (0, 1, 2, 3):
        sol.load_tdgl_data(k)
        assert np.array_equal(sol.tdgl_data.psi, saved[k]["psi"]) and np.array_equal(sol.tdgl_data.mu, saved[k]["mu"])
    assert np.array_equal(sol.dynamics.dt, dts) and np.array_equal(sol.dynamics.theta, th_p.T)
    # an existing file is never overwritten (runner.py:104-134)
    path2, *_ = _write_run("o.h5", device, monkeypatch, tmp_path)
    assert path2.endswith("o-1.h5")


@pytest.mark.skipif(not os.path.isdir("/root/reference/tdgl"), reason="reference checkout not present")
def test_reference_reads_a_file_written_here(tmp_path, monkeypatch):
    """The reference's own `Solution.from_hdf5` (tdgl/solution/solution.py:623-667) opens a file this
    code wrote: same device, same fields per saved step, same dynamics."""
    pytest.importorskip("pint")
    pytest.importorskip("shapely")
    device = _annulus_device(pitch=0.25)
    path, saved, dts, mu_p, th_p = _write_run("o.h5", device, monkeypatch, tmp_path)
    sys.path.insert(0, "/root/reference")
    try:
        import tdgl as ref
    finally:
        sys.path.remove("/root/reference")
    sol = ref.Solution.from_hdf5(path)
    assert np.array_equal(sol.device.mesh.sites, device.mesh.sites)
    for k in (0, 1, 2, 3):
        sol.load_tdgl_data(k)
        assert np.array_equal(sol.tdgl_data.psi, saved[k]["psi"]) and np.array_equal(sol.tdgl_data.mu, saved[k]["mu"])
    assert np.array_equal(sol.dynamics.dt, dts) and np.array_equal(sol.dynamics.mu, mu_p.T)


@pytest.mark.gpu
def test_solve_writes_a_real_file_that_reads_back(tmp_path, monkeypatch):
    """`tdgl.solve(..., SolverOptions(output_file=...))` end to end with h5py: every saved step on disk
    equals the step the run took, and the file's last step equals the returned solution's."""
    import tdgl_amd as tdgl
    from tdgl_amd.geometry import box
    from tdgl_amd.solution import Solution

    monkeypatch.chdir(tmp_path)
    layer = tdgl.Layer(coherence_length=0.5, london_lambda=2.0, thickness=0.1)
    device = tdgl.Device("film", layer=layer, film=tdgl.Polygon("film", points=box(8, 6)), probe_points=[(-2, 0), (2, 0)])
    device.make_mesh(max_edge_length=0.3)
    opts = tdgl.SolverOptions(solve_time=3.0, dt_init=1e-3, save_every=25, field_units="mT", output_file="run.h5")
    solution = tdgl.solve(device, opts, applied_vector_potential=0.4)
    assert solution.path.endswith("run.h5") and os.path.exists(solution.path) and not os.path.exists(solution.path + ".tmp")
    back = Solution.from_hdf5(solution.path)
    lo, hi = back.data_range
    assert lo == 0 and hi >= 2
    back.load_tdgl_data(hi)
    assert np.array_equal(back.tdgl_data.psi, solution.tdgl_data.psi) and np.array_equal(back.tdgl_data.mu, solution.tdgl_data.mu)
    assert np.array_equal(back.dynamics.dt, solution.dynamics.dt)
    assert np.allclose(back.dynamics.mu, solution.dynamics.mu)
