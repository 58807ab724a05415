"""GPU tests written against the public API the way the reference's own tests are
(`tdgl/test/test_solve.py`): Device / Layer / Polygon in physical units, `tdgl.solve`,
`SolverOptions`, terminal currents, physical assertions."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _transport_device():
    import tdgl_amd as tdgl
    from tdgl_amd.geometry import box

    layer = tdgl.Layer(coherence_length=0.5, london_lambda=2.0, thickness=0.1)
    film = tdgl.Polygon("film", points=box(30.0, 4.0))
    source = tdgl.Polygon("source", points=box(0.2, 4.0, center=(-15.0, 0)))
    drain = tdgl.Polygon("drain", points=box(0.2, 4.0, center=(15.0, 0)))
    device = tdgl.Device("strip", layer=layer, film=film, terminals=[source, drain],
                         probe_points=[(-10.0, 0.0), (10.0, 0.0)], length_units="um")
    device.make_mesh(max_edge_length=0.25)
    return device


@pytest.mark.parametrize("current", [5.0, lambda t: 10.0])
@pytest.mark.parametrize("field", [0.0, 1.0])
@pytest.mark.parametrize("terminal_psi", [0.0, None])
def test_source_drain_current(current, field, terminal_psi):
    """cf. test_source_drain_current, tdgl/test/test_solve.py:15-125."""
    import tdgl_amd as tdgl

    device = _transport_device()
    if callable(current):
        def terminal_currents(t):
            return dict(source=current(t), drain=-current(t))
        total = current(0)
    else:
        terminal_currents = dict(source=current, drain=-current)
        total = current
    with pytest.raises(tdgl.SolverOptionsError):
        tdgl.solve(device, tdgl.SolverOptions(solve_time=1, sparse_solver="bogus"), terminal_currents=terminal_currents)
    with pytest.raises(ValueError, match="epsilon must be <= 1"):
        tdgl.solve(device, tdgl.SolverOptions(solve_time=1), disorder_epsilon=2.0)
    with pytest.raises(ValueError, match="sum of all terminal currents must be 0"):
        tdgl.solve(device, tdgl.SolverOptions(solve_time=1), terminal_currents=dict(source=1.0, drain=0.0))
    options = tdgl.SolverOptions(solve_time=20, dt_init=1e-3, field_units="uT", current_units="uA",
                                 save_every=100, terminal_psi=terminal_psi)
    solution = tdgl.solve(device, options, applied_vector_potential=field, terminal_currents=terminal_currents)
    assert isinstance(solution, tdgl.Solution) and len(solution.saved_steps) >= 2
    # current through five cross-sections = the applied current (the reference asserts rtol=0.1
    # on an interpolated path; the finite-volume sum is conserved to round-off)
    j_scale = device.current_scale("uA")
    for x in np.linspace(-12, 12, 5) + 0.013:
        measured = solution.current_through_cut(x / device.coherence_length) / j_scale * device.coherence_length
        assert np.isclose(measured, total, rtol=1e-6), (x, measured, total)
    # ... and the reference's own measurement (test_solve.py:113-125): interpolated current density
    # integrated along five vertical paths that overshoot the film, rtol 0.1.  The path runs upwards,
    # its normal (dy, -dx) points along +x: current flowing from the source (left) to the drain (right).
    ys = np.linspace(-2.5, 2.5, 501)
    measured = np.array([
        solution.current_through_path(np.array([x0 * np.ones_like(ys), ys]).T, with_units=False)
        for x0 in (-12.0, -3.0, 0.0, 3.0, 12.0)
    ])
    assert np.allclose(measured, total, rtol=0.1), measured
    v = solution.dynamics.voltage()
    assert v.shape == solution.dynamics.dt.shape
    dtw = solution.dynamics.dt[len(v) // 2:]
    assert (v[len(v) // 2:] * dtw).sum() / dtw.sum() > 0  # time-averaged voltage follows the current


def test_reference_physical_pin_kmax():
    """The only place the reference pins its unit conversion numerically
    (tdgl/test/test_solve.py:128-176): a 2 x 1 um bar, xi = 0.1, lambda = 0.075, d = 0.05 um in
    0.1 mT carries K_max ~ 450 uA/um (rtol 5e-2 ON THE REFERENCE'S Triangle MESH) after 2 tau_0.
    K_max sits on boundary sites, where the reference's edge->site average (mean of F_e e_hat over
    the incident edges, halved) depends on how the mesh edges meet the boundary: on the jittered
    triangular lattice used here it converges to 411 uA/um (408 / 412 / 411 at pitch 0.045 / 0.03 /
    0.02 um, same number from the oracle).  So: within 12 % of the reference's figure -- a wrong
    Phi_0, mu_0, factor 2 pi or factor 4 in Bc2 / K0 / A_scale would be off by >= 2x -- and equal
    to the oracle run through the same unit conversion."""
    import tdgl_amd as tdgl
    from tdgl_amd.geometry import box

    layer = tdgl.Layer(coherence_length=0.1, london_lambda=0.075, thickness=0.05)
    film = tdgl.Polygon("film", points=box(2, 1, points=301))
    device = tdgl.Device("bar", layer=layer, film=film, length_units="um")
    device.make_mesh(max_edge_length=0.05)
    options = tdgl.SolverOptions(solve_time=2, field_units="mT", current_units="uA")
    solution = tdgl.solve(device, options, applied_vector_potential=0.1)
    K = solution.current_density
    k_max = np.sqrt(K[:, 0] ** 2 + K[:, 1] ** 2).max()
    assert np.isclose(k_max, 450, rtol=0.12), k_max
    from types import SimpleNamespace

    from helpers import uniform_field_A
    from oracle import OracleSolver, run_time_loop

    mesh = device.mesh
    o = SimpleNamespace(solve_time=2.0, skip_time=0.0, dt_init=1e-6, dt_max=0.1, adaptive=True, adaptive_window=10,
                        max_solve_retries=10, adaptive_time_step_multiplier=0.25, terminal_psi=0.0, save_every=100)
    b = 0.1e-3 / device.Bc2
    want = run_time_loop(OracleSolver(mesh, uniform_field_A(mesh, b), 1.0, 5.79, 10.0, o), o)
    k_ref = device.K0 * np.linalg.norm(
        mesh.get_quantity_on_site(want["supercurrent"] + want["normal_current"]), axis=1).max()
    assert np.isclose(k_max, k_ref, rtol=1e-6), (k_max, k_ref)


def test_meissner_state_matches_the_london_solution():
    """An independent pin of the whole unit chain (field units -> A_scale -> solver -> K0 -> uA/um),
    which the reference itself only holds to 5 % through a mesh-dependent boundary maximum: in a weak
    field (|psi|^2 = 1 - 2e-4) the sheet current is the London response, K = curl(g z) with
    lap g = B / (mu_0 Lambda), g = 0 on the rim -- the torsion problem of a rectangle, solved by its
    classical series.  Away from the rim (where the reference's edge->site average is biased, see
    above) the solver must reproduce it to the mesh's accuracy; K_max = 0.930 B w / (2 mu_0 Lambda)
    contains neither Phi_0 nor xi, so a slip in Bc2, K0 or A_scale does not cancel."""
    import tdgl_amd as tdgl
    from scipy import constants as sc
    from tdgl_amd.geometry import box

    xi, lam, d, a, b, B = 0.1, 0.075, 0.05, 2.0, 1.0, 0.1e-3
    layer = tdgl.Layer(coherence_length=xi, london_lambda=lam, thickness=d)
    device = tdgl.Device("bar", layer=layer, film=tdgl.Polygon("film", points=box(a, b, points=301)), length_units="um")
    device.make_mesh(max_edge_length=0.03)
    options = tdgl.SolverOptions(solve_time=20, field_units="mT", current_units="uA")
    solution = tdgl.solve(device, options, applied_vector_potential=B * 1e3)
    K = solution.current_density  # uA/um = A/m

    c = B / (sc.mu_0 * lam**2 / d * 1e-6)  # A/m^2
    x, y, am, bm = device.points[:, 0] * 1e-6, device.points[:, 1] * 1e-6, a * 1e-6, b * 1e-6
    g_x, g_y = np.zeros_like(x), c * y
    for n in range(1, 60, 2):
        cn, k = c * (4 * bm**2 / np.pi**3) * (-1) ** ((n - 1) // 2) / n**3, n * np.pi / bm
        decay = np.exp(k * (np.abs(x) - am / 2)) / (1 + np.exp(-k * am))  # cosh, sinh (k x) / cosh(k a / 2)
        g_y -= cn * k * np.sin(k * y) * decay * (1 + np.exp(-2 * k * np.abs(x)))
        g_x += cn * k * np.cos(k * y) * np.sign(x) * decay * (1 - np.exp(-2 * k * np.abs(x)))
    K_london = np.stack([g_y, -g_x], axis=1)
    k_max = np.linalg.norm(K_london, axis=1).max()
    assert np.isclose(k_max, 0.930 * B * bm / (2 * sc.mu_0 * lam**2 / d * 1e-6), rtol=2e-3)  # 328.9 uA/um
    inner = (np.abs(device.points[:, 0]) < a / 2 - 0.07) & (np.abs(device.points[:, 1]) < b / 2 - 0.07)
    assert inner.sum() > 0.6 * len(inner)
    err = np.abs(K - K_london)[inner].max() / k_max
    assert err < 0.03, err
    assert (np.abs(solution.tdgl_data.psi) ** 2).min() > 0.999


def test_screening_restores_fluxoid_quantisation():
    """The reference's screening test (tdgl/test/test_solve.py:152-196) with its device, field and
    curves: without screening the fluxoid of a region (applied flux + mu_0 Lambda oint K_s / |psi|^2)
    is far from zero relative to its flux part; with the self-field iterated to 1e-6 it vanishes to
    5 %.  The sheet-current maxima the reference quotes there (450 / 270 uA/um) sit on boundary sites
    and depend on the mesher (see test_reference_physical_pin_kmax); their ratio is checked loosely."""
    import tdgl_amd as tdgl
    from tdgl_amd.geometry import box, circle

    xi = 0.1
    layer = tdgl.Layer(coherence_length=xi, london_lambda=0.075, thickness=0.05)
    device = tdgl.Device("bar", layer=layer, film=tdgl.Polygon("film", points=box(2, 1, points=301)), length_units="um")
    # (the reference's mesher refines until no edge exceeds xi / 2 and ends well below it; the lattice
    # mesher here uses the given length as its pitch, so ask for xi / 3: the smallest curve has radius xi)
    device.make_mesh(max_edge_length=xi / 3, smooth=100)
    curves = [circle(0.25, center=(0, 0)), circle(0.1, center=(0.15, 0.25)), circle(0.3, center=(0.6, -0.1)),
              box(0.5, center=(-0.5, 0)), box(0.5, center=(-0.6, -0.2))]
    options = tdgl.SolverOptions(solve_time=2, field_units="mT", current_units="uA", include_screening=False)
    bare = tdgl.solve(device, options, applied_vector_potential=0.1)
    k_bare = np.linalg.norm(bare.current_density, axis=1).max()
    for curve in curves:
        fluxoid = bare.polygon_fluxoid(curve)
        assert abs(sum(fluxoid).magnitude / fluxoid.flux_part.magnitude) > 1
    options.include_screening = True
    options.screening_tolerance = 1e-6
    options.dt_max = 1e-3
    screened = tdgl.solve(device, options, applied_vector_potential=0.1)
    k_scr = np.linalg.norm(screened.current_density, axis=1).max()
    errors = []
    for curve in curves:
        fluxoid = screened.polygon_fluxoid(curve)
        errors.append(abs(sum(fluxoid).magnitude / fluxoid.flux_part.magnitude))
    assert max(errors) < 5e-2, errors
    assert np.isclose(k_scr / k_bare, 270 / 450, rtol=0.05), (k_bare, k_scr)


def test_time_dependent_field_through_the_public_api():
    """Field ramp written like the reference's docs: LinearRamp(...) * ConstantField(...)."""
    import tdgl_amd as tdgl
    from tdgl_amd.geometry import box

    layer = tdgl.Layer(coherence_length=0.5, london_lambda=2.0, thickness=0.1)
    device = tdgl.Device("sq", layer=layer, film=tdgl.Polygon("film", points=box(8.0)), length_units="um")
    device.make_mesh(max_edge_length=0.25)
    ramp = tdgl.LinearRamp(tmin=0, tmax=4) * tdgl.ConstantField(1.2, field_units="mT", length_units="um")
    assert ramp.time_dependent
    options = tdgl.SolverOptions(solve_time=6, dt_init=1e-3, field_units="mT", save_every=50)
    solution = tdgl.solve(device, options, applied_vector_potential=ramp)
    static = tdgl.solve(device, tdgl.SolverOptions(solve_time=6, dt_init=1e-3, field_units="mT", save_every=50),
                        applied_vector_potential=1.2)
    # after the ramp has ended both runs sit in the same field: screening currents agree in scale
    k_dyn = np.linalg.norm(solution.current_density, axis=1).max()
    k_sta = np.linalg.norm(static.current_density, axis=1).max()
    assert 0.5 < k_dyn / k_sta < 2.0
    # while the field changes, dA/dt drives a normal current: J_n != 0 early in the ramp
    early = solution.saved_steps[1]
    assert np.abs(early.normal_current).max() > 1e-6


def test_device_with_holes_and_terminals_runs_and_conserves_current():
    """The reference's transport_device shape (tdgl/test/conftest.py:7-49: 10x10 box united with a
    30x4 strip, two round holes, source/drain terminals), meshed by the built-in polygon mesher."""
    import tdgl_amd as tdgl
    from conftest import load_golden

    g = load_golden("mesh_polygon")
    layer = tdgl.Layer(coherence_length=0.5, london_lambda=2.0, thickness=0.1)
    device = tdgl.Device(
        "transport", layer=layer, film=tdgl.Polygon("film", points=g["film"]),
        holes=[tdgl.Polygon("h0", points=g["hole0"]), tdgl.Polygon("h1", points=g["hole1"])],
        terminals=[tdgl.Polygon("source", points=[(-15.1, -2.1), (-14.9, -2.1), (-14.9, 2.1), (-15.1, 2.1)]),
                   tdgl.Polygon("drain", points=[(14.9, -2.1), (15.1, -2.1), (15.1, 2.1), (14.9, 2.1)])],
        probe_points=[(-10, 0), (10, 0)], length_units="um",
    )
    device.make_mesh(max_edge_length=0.25)
    assert len(device.mesh.sites) > 5000
    options = tdgl.SolverOptions(solve_time=10, dt_init=1e-3, field_units="uT", current_units="uA", save_every=100)
    solution = tdgl.solve(device, options, applied_vector_potential=1.0, terminal_currents=dict(source=10.0, drain=-10.0))
    j_scale = device.current_scale("uA")
    xi = device.coherence_length
    for x in (-12.03, -7.51, 7.49, 11.97):  # cuts through the strip arms
        measured = solution.current_through_cut(x / xi) / j_scale * xi
        assert np.isclose(measured, 10.0, rtol=1e-6), (x, measured)
    # a cut through the box passes the holes: the current through the film is still I
    assert np.isclose(solution.current_through_cut(0.013 / xi) / j_scale * xi, 10.0, rtol=1e-6)
    assert np.abs(solution.tdgl_data.psi).max() < 1.2  # transient overshoot near phase slips is physical


def test_output_file_is_streamed_in_the_reference_layout(tmp_path, monkeypatch):
    """SolverOptions.output_file: every saved step is written when it is taken
    (DataHandler.save_time_step, tdgl/solver/runner.py:155-183), the latest step also into the
    `.tmp` file of the reference's monitor; only the last step stays in memory.  h5py is not on the
    image, so every h5py call lands in an in-memory recorder (tests/h5_recorder.py).  Compared with
    the same run held in memory; a run that dies keeps what it had saved."""
    import sys

    sys.path.insert(0, str(__import__("pathlib").Path(__file__).parent))
    import h5_recorder as rec
    import tdgl_amd as tdgl
    from tdgl_amd import io as tio

    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(tio, "_h5py_factory", rec.RecorderFile)
    rec.OPENED.clear()
    device = _transport_device()
    cur = dict(source=5.0, drain=-5.0)
    kw = dict(solve_time=3, skip_time=0.5, dt_init=1e-3, field_units="uT", current_units="uA", save_every=40)
    mem = tdgl.solve(device, tdgl.SolverOptions(**kw), applied_vector_potential=1.0, terminal_currents=cur)
    sol = tdgl.solve(device, tdgl.SolverOptions(**kw, output_file="run/out.h5"), applied_vector_potential=1.0,
                     terminal_currents=cur)
    assert sol.path.endswith("run/out.h5") and len(sol.saved_steps) == 1
    f = rec.OPENED[sol.path]
    tmp = rec.OPENED[sol.path + ".tmp"]
    assert f.closed and tmp.closed
    assert set(f) == {"mesh", "data", "applied_vector_potential", "epsilon", "solution"}
    assert list(f["data"]) == [str(k) for k in range(len(mem.saved_steps))]
    assert [(int(f[f"data/{k}"].attrs["step"])) for k in range(len(mem.saved_steps))] == [s.step for s in mem.saved_steps]
    every = kw["save_every"]
    for k, s in enumerate(mem.saved_steps):
        g = f[f"data/{k}"]
        assert g.attrs["time"] == s.time and g.attrs["dt"] == s.dt
        for name in ("psi", "mu", "supercurrent", "normal_current"):
            assert np.array_equal(g[name].value, getattr(s, name)), (k, name)
        assert g["induced_vector_potential"].shape == (len(s.supercurrent), 2)
        if k == 0:
            assert "running_state" not in g
            continue
        lo = mem.saved_steps[k - 1].step
        hi = min(s.step + (1 if s.step % every else 0), len(mem.dynamics.dt))
        want = np.zeros(every)
        want[: hi - lo] = mem.dynamics.dt[lo:hi]
        assert np.array_equal(g["running_state/dt"].value, want), k
        assert np.array_equal(g["running_state/mu"].value[:, : hi - lo], mem.dynamics.mu[:, lo:hi])
    # the solution group the reference's Solution.from_hdf5 starts from
    sg = f["solution"]
    assert sg["options"].attrs["save_every"] == every and sg.attrs["current_units"] == "uA"
    assert set(sg["device"]) >= {"layer", "film", "terminals", "probe_points", "mesh"}
    assert sg["device/layer"].attrs["coherence_length"] == 0.5 and "terminal_currents.pickle" in sg
    # latest-step file: one group, last saved state
    assert set(tmp["data"]) == {"-1"} and tmp["data/-1/step"][0] == mem.saved_steps[-1].step
    # a run that dies after some saves leaves them on disk, files closed
    solver = tdgl.TDGLSolver(device, tdgl.SolverOptions(**kw, output_file="run/dies.h5"),
                             applied_vector_potential=1.0, terminal_currents=cur)
    real_run, calls = solver.ctx.run, []

    def flaky(*a, **k):
        calls.append(1)
        written = [v for p, v in rec.OPENED.items() if p.endswith("run/dies.h5")]
        if written and len(list(written[0]["data"])) >= 1:  # something has been saved: now die
            raise RuntimeError("boom")
        return real_run(*a, **k)

    solver.ctx.run = flaky
    with pytest.raises(RuntimeError, match="boom"):
        solver.solve()
    died = [v for p, v in rec.OPENED.items() if p.endswith("run/dies.h5")][0]
    assert died.closed and len(list(died["data"])) >= 1 and "solution" not in died


def test_seed_solution_read_back_from_the_output_file_continues_the_run(tmp_path, monkeypatch):
    """`seed_solution` (solver.py:731-752): a run of 2N steps equals a run of N steps, streamed to an
    output file, read back with Solution.from_hdf5 and continued for N steps from that solution
    (fixed dt, so the two halves take exactly the steps of the whole; the loop's one step past the end,
    runner.py:429-433, is accounted for in the solve times).  The device that comes back from the file
    is equal to, not identical with, the simulated one."""
    import sys

    sys.path.insert(0, str(__import__("pathlib").Path(__file__).parent))
    import h5_recorder as rec
    import tdgl_amd as tdgl
    from tdgl_amd import io as tio

    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(tio, "_h5py_factory", rec.open_file)
    rec.OPENED.clear()
    device = _transport_device()
    cur = dict(source=5.0, drain=-5.0)
    dt, half = 2.0**-7, 1.0
    kw = dict(dt_init=dt, adaptive=False, field_units="uT", current_units="uA", save_every=10)
    whole = tdgl.solve(device, tdgl.SolverOptions(solve_time=2 * half + dt, **kw), applied_vector_potential=2.0,
                       terminal_currents=cur)
    first = tdgl.solve(device, tdgl.SolverOptions(solve_time=half, output_file="first.h5", **kw),
                       applied_vector_potential=2.0, terminal_currents=cur)
    n_half = len(first.dynamics.dt)
    assert len(whole.dynamics.dt) == 2 * n_half
    seed = tdgl.Solution.from_hdf5(first.path)
    assert seed.device == device and seed.device is not device
    assert np.array_equal(seed.tdgl_data.psi, first.tdgl_data.psi) and seed.tdgl_data.step == first.tdgl_data.step
    assert np.array_equal(seed.dynamics.dt, first.dynamics.dt) and np.array_equal(seed.dynamics.mu, first.dynamics.mu)
    second = tdgl.solve(device, tdgl.SolverOptions(solve_time=half, **kw), applied_vector_potential=2.0,
                        terminal_currents=cur, seed_solution=seed)
    a, b = whole.tdgl_data, second.tdgl_data
    assert np.abs(a.psi - b.psi).max() < 1e-8 * np.abs(a.psi).max()
    assert np.abs((a.mu - a.mu.mean()) - (b.mu - b.mu.mean())).max() < 1e-7 * np.abs(a.mu).max()
    assert np.abs(a.supercurrent - b.supercurrent).max() < 1e-8
    other = _transport_device()
    other.name = "another"
    with pytest.raises(ValueError, match="seed_solution.device must be equal"):
        tdgl.solve(other, tdgl.SolverOptions(solve_time=half, **kw), seed_solution=seed)


def test_hierarchy_candidates_are_probed_and_the_best_one_stays():
    """`TDGLContext.build_poisson(amg_candidates=3)`: the hashed priorities of the MIS(2) aggregation give hierarchies
    that differ by luck (at 1M sites 7.50 to 7.79 PCG iterations per step); three are built, each solves one fixed
    pseudo-random right-hand side on the device, the smallest contraction per iteration stays on the device."""
    from helpers import synthetic_mesh
    from tdgl_amd.hipcore import TDGLContext

    mesh = synthetic_mesh(120, 100)
    ctx = TDGLContext(mesh, direct_solve=False)
    h = ctx.build_poisson(rtol=1e-10, amg_candidates=3)
    scores = ctx.setup_times["amg_candidates"]
    assert [s["seed"] for s in scores] == [0, 1, 2] and len({tuple(s["sizes"]) for s in scores}) >= 2
    best = min(scores, key=lambda s: s["contraction"])
    assert h.sizes == best["sizes"] and ctx.hierarchy.sizes == best["sizes"]
    assert all(0.05 < s["contraction"] < 0.6 and s["relres"] <= 1e-10 for s in scores)
    rng = np.random.default_rng(1)
    rhs = rng.standard_normal(len(mesh.sites)) / mesh.areas
    rhs -= (rhs * mesh.areas).sum() / mesh.areas.sum()
    _, its, relres = ctx.poisson_solve(rhs)
    assert relres <= 1e-10 and abs(its - best["iterations"]) <= 2
    # one candidate: nothing is probed
    ctx2 = TDGLContext(mesh, direct_solve=False)
    ctx2.build_poisson(amg_candidates=1)
    assert "amg_candidates" not in ctx2.setup_times
    ctx.close(); ctx2.close()
