"""Generate the golden fixtures by running the REFERENCE itself (build container only).

    python tests/golden/generate_golden.py [--out DIR] [--quick]

imports py-tdgl v0.8.3 from `/root/reference` (see `_reference_shim.py`), drives its own
`Mesh.from_triangulation`, `MeshOperators`, `TDGLSolver.solve_for_psi_squared`,
`TDGLSolver.update` and `Runner` on small synthetic problems and stores inputs + outputs as
`tests/golden/*.npz`.  The fixtures are data only (no reference source text).

`TDGLSolver.__init__` needs pint/shapely/meshpy (absent), so the solver instance is made
with `object.__new__` and given exactly the attributes `update()` reads
(SURVEY.md §8(c)); the loop is the reference's `Runner` with an in-memory data handler.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "py-tdgl_amd"))

from _reference_shim import import_reference  # noqa: E402

tdgl = import_reference()
from tdgl.device.device import TerminalInfo  # noqa: E402
from tdgl.finite_volume import Mesh as RefMesh  # noqa: E402
from tdgl.finite_volume.operators import MeshOperators  # noqa: E402
from tdgl.solver.options import SolverOptions, SparseSolver  # noqa: E402
from tdgl.solver.runner import Runner  # noqa: E402
from tdgl.solver.solver import TDGLSolver  # noqa: E402

from scipy.spatial import Delaunay  # noqa: E402

from tdgl_amd.meshgen import hex_jitter_points  # noqa: E402

# Where the .npz files go: next to this script unless --out DIR is given (the regeneration test in
# tests/test_oracle_golden.py writes into a scratch directory and compares with the committed files).
OUT_DIR = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE


def triangulate(pts):
    """Fixture inputs must not depend on product code that can change: the triangulation every fixture mesh is
    built on is SciPy's Qhull Delaunay (what SURVEY.md Appendix B prescribes), never the product's own
    triangulator -- its element order and orientation differ, and with them the last bits of the reference's
    Voronoi areas and every chaotic trajectory."""
    return np.asarray(Delaunay(pts).simplices, dtype=np.int64)


U_DEFAULT, GAMMA_DEFAULT = 5.79, 10.0


def uniform_field_A(mesh, b):
    """Dimensionless symmetric-gauge A on edge centres, gauge centre = bbox centre of the
    edge centres (what tdgl/em.py:437-472 does after unit scaling)."""
    c = mesh.edge_mesh.centers
    xc = c[:, 0].min() + np.ptp(c[:, 0]) / 2
    yc = c[:, 1].min() + np.ptp(c[:, 1]) / 2
    return np.column_stack([-b * (c[:, 1] - yc) / 2, b * (c[:, 0] - xc) / 2])


def make_ref_mesh(lx, ly):
    pts = hex_jitter_points(lx, ly)
    tri = triangulate(pts)
    return RefMesh.from_triangulation(pts, tri)


def mesh_arrays(mesh, prefix="mesh_"):
    em = mesh.edge_mesh
    return {
        prefix + "sites": mesh.sites,
        prefix + "elements": mesh.elements,
        prefix + "boundary_indices": mesh.boundary_indices,
        prefix + "areas": mesh.areas,
        prefix + "dual_sites": mesh.dual_sites,
        prefix + "edges": em.edges,
        prefix + "boundary_edge_indices": em.boundary_edge_indices,
        prefix + "centers": em.centers,
        prefix + "directions": em.directions,
        prefix + "edge_lengths": em.edge_lengths,
        prefix + "dual_edge_lengths": em.dual_edge_lengths,
    }


def edge_terminal(mesh, name, x0):
    """A terminal covering the whole boundary segment x = x0 (what Device.terminal_info,
    tdgl/device/device.py:221-256, returns for a terminal polygon enclosing that side)."""
    em = mesh.edge_mesh
    site_idx = np.intersect1d(
        np.flatnonzero(np.isclose(mesh.sites[:, 0], x0)), mesh.boundary_indices
    )
    bidx = em.boundary_edge_indices
    on_side = np.isclose(em.centers[bidx, 0], x0)
    boundary_pos = np.flatnonzero(on_side)
    edge_idx = bidx[on_side]
    length = em.edge_lengths[bidx][boundary_pos].sum()
    return TerminalInfo(name, site_idx, edge_idx, boundary_pos, length)


def make_ref_solver(mesh, A, options, terminals=(), currents=None, probe_points=None,
                    epsilon=1.0, u=U_DEFAULT, gamma=GAMMA_DEFAULT):
    options.validate()
    terminals = tuple(sorted(terminals, key=lambda t: t.length))
    names = [t.name for t in terminals]
    fixed = (
        np.concatenate([t.site_indices for t in terminals], dtype=np.int64)
        if terminals
        else np.array([], dtype=np.int64)
    )
    ops = MeshOperators(
        mesh,
        SparseSolver.SUPERLU,
        use_cupy=False,
        fixed_sites=fixed,
        fix_psi=(options.terminal_psi is not None),
    )
    ops.build_operators()
    ops.set_link_exponents(A)
    s = object.__new__(TDGLSolver)
    s.options = options
    s.xp = np
    s.use_cupy = False
    s.u, s.gamma = u, gamma
    s.probe_points = probe_points
    s.terminal_info = terminals
    s.terminal_names = names
    s.terminal_current_densities = {name: 0 for name in names}
    if currents is None:
        currents = {name: 0 for name in names}
    s.current_func = currents if callable(currents) else (lambda t: currents)
    s.dynamic_vector_potential = False
    s.dynamic_epsilon = False
    s.operators = ops
    n = len(mesh.sites)
    s.epsilon = epsilon * np.ones(n)
    s.mu_boundary = np.zeros(len(mesh.edge_mesh.boundary_edge_indices))
    s.normalized_directions = mesh.edge_mesh.normalized_directions
    s.current_A_applied = A
    s.d_psi_sq_vals = []
    s.tentative_dt = options.dt_init
    s.dt_max = options.dt_max if options.adaptive else options.dt_init
    psi0 = np.ones(n, dtype=np.complex128)
    if options.terminal_psi is not None:
        psi0[fixed] = options.terminal_psi
    return s, psi0, fixed


class MemoryHandler:
    """Stands in for DataHandler (no HDF5): records every save_time_step call."""

    tmp_file = None
    output_path = None

    def __init__(self):
        self.saves = []

    def save_fixed_values(self, fixed):
        pass

    def save_time_step(self, state, data, running_state):
        self.saves.append(
            dict(
                step=state["step"],
                time=state["time"],
                dt=state["dt"],
                has_running=running_state is not None,
            )
        )


def run_reference(solver, psi0, options, snapshot_steps=()):
    """Run the reference Runner around solver.update; log every call."""
    m = len(solver.operators.edges)
    n = len(psi0)
    calls = []
    snaps = {}
    extra_names, extra_values = [], []
    if solver.dynamic_vector_potential:  # solver.py:756-757
        extra_names.append("applied_vector_potential")
        extra_values.append(solver.current_A_applied)
    if solver.dynamic_epsilon:  # solver.py:761-762
        extra_names.append("epsilon")
        extra_values.append(solver.epsilon)

    def logged_update(state, running_state, dt, **values):
        result = solver.update(state, running_state, dt, **values)
        rec = dict(stage_step=state["step"], time=state["time"], state_dt=state["dt"], dt=result[0])
        if solver.options.include_screening:
            rec["screening_iterations"] = int(running_state.values["screening_iterations"][0, running_state.step])
        if solver.probe_points is not None:
            rec["mu_probe"] = np.array(result[2][solver.probe_points])
            rec["theta_probe"] = np.angle(result[1][solver.probe_points])
        calls.append(rec)
        k = len(calls) - 1
        if k in snapshot_steps:
            snaps[k] = tuple(np.array(x) for x in result[1:5])
        return result

    sizes = {"dt": 1}
    if solver.probe_points is not None:
        sizes["mu"] = len(solver.probe_points)
        sizes["theta"] = len(solver.probe_points)
    if solver.options.include_screening:
        sizes["screening_iterations"] = 1
    handler = MemoryHandler()
    runner = Runner(
        function=logged_update,
        options=options,
        data_handler=handler,
        initial_values=[psi0, np.zeros(n), np.zeros(m), np.zeros(m), np.zeros((m, 2))] + extra_values,
        names=["psi", "mu", "supercurrent", "normal_current", "induced_vector_potential"] + extra_names,
        fixed_values=(),
        fixed_names=(),
        running_names_and_sizes=sizes,
    )
    ok = runner.run()
    assert ok
    psi, mu, js, jn = runner.values[:4]
    a_ind = runner.values[4]
    out = dict(
        final_A_induced=np.asarray(a_ind),
        final_psi=psi,
        final_mu=mu,
        final_supercurrent=js,
        final_normal_current=jn,
        call_stage_step=np.array([c["stage_step"] for c in calls]),
        call_time=np.array([c["time"] for c in calls], dtype=float),
        call_state_dt=np.array([c["state_dt"] for c in calls], dtype=float),
        call_dt=np.array([c["dt"] for c in calls], dtype=float),
        save_step=np.array([s["step"] for s in handler.saves]),
        save_time=np.array([s["time"] for s in handler.saves], dtype=float),
        save_dt=np.array([s["dt"] for s in handler.saves], dtype=float),
        save_has_running=np.array([s["has_running"] for s in handler.saves]),
        final_runner_time=float(runner.time),
        d_psi_sq_vals=np.array(solver.d_psi_sq_vals),
    )
    if solver.options.include_screening:
        out["call_screening_iterations"] = np.array([c["screening_iterations"] for c in calls])
    if solver.probe_points is not None:
        out["call_mu_probe"] = np.array([c["mu_probe"] for c in calls])
        out["call_theta_probe"] = np.array([c["theta_probe"] for c in calls])
    for k, (p, mu_k, js_k, jn_k) in snaps.items():
        out[f"snap{k}_psi"] = p
        out[f"snap{k}_mu"] = mu_k
        out[f"snap{k}_supercurrent"] = js_k
        out[f"snap{k}_normal_current"] = jn_k
    out["snapshot_steps"] = np.array(sorted(snaps), dtype=np.int64)
    return out


def options_arrays(o):
    return dict(
        opt_solve_time=o.solve_time,
        opt_skip_time=o.skip_time,
        opt_dt_init=o.dt_init,
        opt_dt_max=o.dt_max,
        opt_adaptive=o.adaptive,
        opt_adaptive_window=o.adaptive_window,
        opt_max_solve_retries=o.max_solve_retries,
        opt_adaptive_time_step_multiplier=o.adaptive_time_step_multiplier,
        opt_save_every=o.save_every,
        opt_terminal_psi=np.nan if o.terminal_psi is None else o.terminal_psi,
    )


def coo(mat, prefix):
    c = mat.tocoo()
    c.sum_duplicates()
    order = np.lexsort((c.col, c.row))
    return {
        prefix + "_row": c.row[order].astype(np.int64),
        prefix + "_col": c.col[order].astype(np.int64),
        prefix + "_val": c.data[order],
        prefix + "_shape": np.array(c.shape),
    }


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays")


def gen_dynamic_lag(small):
    """(4g') A(t) = ramp(t) * A_base moving by LESS than np.allclose's tolerance per step: the
    reference then updates dA/dt and current_A_applied but NOT the link variables
    (solver.py:636-637), which stay at their t = 0 values for the whole run."""
    o = SolverOptions(solve_time=2.0, dt_init=1e-3, dt_max=1e-3, adaptive=False, save_every=500)
    probes = [small.closest_site((-5, 0)), small.closest_site((5, 0))]
    A_base = uniform_field_A(small, 0.01)
    ramp = dict(tmin=0.0, tmax=5.0, initial=30.0, final=31.0)

    def scale(t):
        frac = min(max((t - ramp["tmin"]) / (ramp["tmax"] - ramp["tmin"]), 0.0), 1.0)
        return ramp["initial"] + (ramp["final"] - ramp["initial"]) * frac

    def ramp_A(x, y, z, *, t=0):
        a2 = scale(t) * A_base
        return np.column_stack([a2, np.zeros(len(a2))])

    s, psi0, _ = make_ref_solver(small, scale(0) * A_base, o, probe_points=probes)
    s.dynamic_vector_potential = True
    s.applied_vector_potential = ramp_A
    s.A_scale = 1.0
    s.edge_centers = small.edge_mesh.centers
    s.z0 = np.zeros(len(s.edge_centers))
    s.current_A_applied = ramp_A(None, None, None, t=0)[:, :2]
    s.operators.set_link_exponents(s.current_A_applied)
    out = run_reference(s, psi0, o)
    print("dynamic-lag calls:", len(out["call_dt"]), "max |J_s|", np.abs(out["final_supercurrent"]).max())
    save("traj_dynamic_lag", probe_points=np.array(probes), A_base=A_base,
         **{"ramp_" + k: v for k, v in ramp.items()}, **options_arrays(o), **out)


def gen_device_meshes():
    """Trajectories on real device meshes, run by the reference on meshes
    already committed as fixtures (so the meshes are not regenerated):

    * traj_transport_polygon: the reference's own test device shape (tdgl/test/conftest.py:7-49:
      10x10 box united with a 30x4 strip, two round holes, current terminals on the strip ends, probe
      points on the strip, weak field) on `mesh_polygon` (constrained Delaunay mesh of the polygon);
    * traj_irregular_smoothed: a Laplacian-smoothed, non-Delaunay mesh whose boundary cells take the
      reference's convex-hull area branch (tdgl/finite_volume/util.py:169-255), uniform field."""
    def load(name):
        with np.load(os.path.join(HERE, name + ".npz")) as f:
            return {k: f[k] for k in f.files}

    gp = load("mesh_polygon")
    poly = RefMesh.from_triangulation(gp["mesh_sites"], gp["mesh_elements"])
    assert np.array_equal(poly.edge_mesh.edges, gp["mesh_edges"]) and np.allclose(poly.areas, gp["mesh_areas"], rtol=0, atol=0)
    terms = [edge_terminal(poly, "source", -15.0), edge_terminal(poly, "drain", 15.0)]
    probes = [poly.closest_site((-10, 0)), poly.closest_site((10, 0))]
    o = SolverOptions(solve_time=12.0, skip_time=2.0, dt_init=1e-4, save_every=100)
    cur = {"source": 1.0, "drain": -1.0}
    s, psi0, fx = make_ref_solver(poly, uniform_field_A(poly, 0.15), o, terminals=terms, currents=cur,
                                  probe_points=probes)
    out = run_reference(s, psi0, o, snapshot_steps=(10, 150))
    print("transport_polygon calls:", len(out["call_dt"]), "V", out["call_mu_probe"][-1],
          "min|psi|^2", (np.abs(out["final_psi"]) ** 2).min())
    term_arrays = {}
    for t in terms:
        term_arrays[f"term_{t.name}_sites"] = t.site_indices
        term_arrays[f"term_{t.name}_edges"] = t.edge_indices
        term_arrays[f"term_{t.name}_boundary_pos"] = t.boundary_edge_indices
        term_arrays[f"term_{t.name}_length"] = t.length
    save("traj_transport_polygon", b=0.15, current=1.0, probe_points=np.array(probes), fixed_sites=fx,
         mu_boundary=s.mu_boundary, **term_arrays, **options_arrays(o), **out)

    # The adaptive run above is chaotic on this coarse mesh (the controller bounces dt between 0.005 and
    # 0.03: a 1e-14 perturbation of psi_0 changes the reference's OWN dt sequence by O(1),
    # tests/test_sensitivity.py), so it pins single steps, not the whole trajectory.  The same device
    # with a fixed time step is stable (1e-14 -> 5e-11): stronger field, flux enters the holes.
    o = SolverOptions(solve_time=4.0, skip_time=0.5, dt_init=5e-3, dt_max=5e-3, adaptive=False, save_every=100)
    s, psi0, fx = make_ref_solver(poly, uniform_field_A(poly, 0.3), o, terminals=terms, currents=cur,
                                  probe_points=probes)
    out = run_reference(s, psi0, o, snapshot_steps=(10, 400))
    print("transport_polygon_fixed_dt calls:", len(out["call_dt"]), "V", out["call_mu_probe"][-1],
          "max|Js|", np.abs(out["final_supercurrent"]).max())
    save("traj_transport_polygon_fixed_dt", b=0.3, current=1.0, probe_points=np.array(probes), fixed_sites=fx,
         mu_boundary=s.mu_boundary, **term_arrays, **options_arrays(o), **out)

    gs = load("mesh_irregular_smoothed")
    sm = RefMesh.from_triangulation(gs["mesh_sites"], gs["mesh_elements"])
    assert np.allclose(sm.areas, gs["mesh_areas"], rtol=0, atol=0)
    o = SolverOptions(solve_time=6.0, dt_init=1e-4, save_every=100)
    s, psi0, _ = make_ref_solver(sm, uniform_field_A(sm, 0.9), o)
    out = run_reference(s, psi0, o, snapshot_steps=(5, 120))
    print("irregular_smoothed calls:", len(out["call_dt"]), "min|psi|^2", (np.abs(out["final_psi"]) ** 2).min())
    save("traj_irregular_smoothed", b=0.9, **options_arrays(o), **out)


def main():
    if "--dynamic-lag-only" in sys.argv:
        gen_dynamic_lag(make_ref_mesh(20, 20))
        return
    if "--device-meshes-only" in sys.argv:
        gen_device_meshes()
        return
    # ---- (1) meshes ---------------------------------------------------------------
    small = make_ref_mesh(20, 20)  # 516 sites
    strip = make_ref_mesh(60, 15)  # 1107 sites
    save("mesh_small", **mesh_arrays(small))
    save("mesh_strip", **mesh_arrays(strip))
    # a mesh the synthetic generator does not produce: random interior points
    rng = np.random.default_rng(7)
    side = np.linspace(-5, 5, 21)
    ring = np.concatenate(
        [
            np.column_stack([side, -5 * np.ones(21)]),
            np.column_stack([side, 5 * np.ones(21)]),
            np.column_stack([-5 * np.ones(19), side[1:-1]]),
            np.column_stack([5 * np.ones(19), side[1:-1]]),
        ]
    )
    inner = hex_jitter_points(9.0, 9.0, pitch=0.5, jitter=0.5, seed=3)
    inner = inner[(np.abs(inner[:, 0]) < 4.4) & (np.abs(inner[:, 1]) < 4.4)]
    pts = np.concatenate([ring, inner])
    irregular = RefMesh.from_triangulation(pts, triangulate(pts))
    save("mesh_irregular", **mesh_arrays(irregular))
    # the reference's Laplacian smoothing (finite_volume/mesh.py:245-283): 3 iterations at fixed
    # connectivity, boundary vertices pinned
    save("mesh_irregular_smoothed", iterations=3, **mesh_arrays(irregular.smooth(3)))

    # a polygon with holes, meshed by the product's own mesher (the reference's shape of
    # tdgl/test/conftest.py:7-49: 10x10 box united with a 30x4 strip, two round holes); the
    # dual mesh is the REFERENCE's
    from tdgl_amd.geometry import circle
    from tdgl_amd.meshgen import polygon_mesh

    film = np.array([(-15, -2), (-5, -2), (-5, -5), (5, -5), (5, -2), (15, -2), (15, 2), (5, 2), (5, 5),
                     (-5, 5), (-5, 2), (-15, 2)], dtype=float)
    holes = [circle(1.5, center=(-2.5, 0), points=40), circle(1.0, center=(2.5, 1.0), points=30)]
    ppts, ptri = polygon_mesh(film, holes, max_edge_length=0.8, backend="qhull")  # see triangulate() above
    save("mesh_polygon", film=film, hole0=holes[0], hole1=holes[1],
         **mesh_arrays(RefMesh.from_triangulation(ppts, ptri)))
    if "--meshes-only" in sys.argv:
        return

    # ---- (2) operators --------------------------------------------------------------
    A = uniform_field_A(small, 0.3)
    fixed = np.flatnonzero(np.isclose(small.sites[:, 0], -10.0))
    arrays = dict(A=A, fixed_sites=fixed)
    for tag, fix_psi in [("fixed", True), ("free", False)]:
        ops = MeshOperators(small, SparseSolver.SUPERLU, fixed_sites=fixed, fix_psi=fix_psi)
        ops.build_operators()
        ops.set_link_exponents(A)
        arrays.update(coo(ops.psi_laplacian, f"{tag}_psi_laplacian"))
        if tag == "fixed":
            arrays.update(coo(ops.psi_gradient, "psi_gradient"))
            arrays.update(coo(ops.divergence, "divergence"))
            arrays.update(coo(ops.mu_laplacian, "mu_laplacian"))
            arrays.update(coo(ops.mu_boundary_laplacian, "mu_boundary_laplacian"))
            arrays.update(coo(ops.mu_gradient, "mu_gradient"))
            psi = np.exp(1j * rng.uniform(0, 2 * np.pi, len(small.sites))) * rng.uniform(
                0.2, 1.0, len(small.sites)
            )
            arrays["psi"] = psi
            arrays["supercurrent"] = ops.get_supercurrent(psi)
            # second call = in-place value update path (operators.py:346-383)
            A2 = uniform_field_A(small, 0.45)
            ops.set_link_exponents(A2)
            arrays["A2"] = A2
            arrays.update(coo(ops.psi_laplacian, "fixed_psi_laplacian_A2"))
            arrays.update(coo(ops.psi_gradient, "psi_gradient_A2"))
    save("operators_small", **arrays)

    # ---- (3) psi update single calls ---------------------------------------------------
    ops = MeshOperators(small, SparseSolver.SUPERLU, fixed_sites=fixed, fix_psi=True)
    ops.build_operators()
    ops.set_link_exponents(A)
    n = len(small.sites)
    psi = np.exp(1j * rng.uniform(0, 2 * np.pi, n)) * rng.uniform(0.0, 1.1, n)
    psi[fixed] = 0
    mu = rng.normal(0, 0.5, n)
    eps = rng.uniform(-0.5, 1.0, n)
    arrays = dict(A=A, fixed_sites=fixed, psi=psi, mu=mu, epsilon=eps, u=U_DEFAULT, gamma=GAMMA_DEFAULT)
    dts = np.array([1e-4, 1e-2, 0.1, 1.0, 30.0])
    ok = []
    for k, dt in enumerate(dts):
        res = TDGLSolver.solve_for_psi_squared(
            psi=psi, abs_sq_psi=np.abs(psi) ** 2, mu=mu, epsilon=eps,
            gamma=GAMMA_DEFAULT, u=U_DEFAULT, dt=dt, psi_laplacian=ops.psi_laplacian,
        )
        ok.append(res is not None)
        if res is not None:
            arrays[f"out{k}_psi"], arrays[f"out{k}_abs_sq"] = res
    arrays["dts"] = dts
    arrays["ok"] = np.array(ok)
    print("psi_update ok flags:", ok)
    save("psi_update_small", **arrays)
    if "--quick" in sys.argv:  # meshes + operators + single psi updates (tests/test_oracle_golden.py regenerates these)
        return

    # ---- (4) trajectories ------------------------------------------------------------
    # (4a) BASELINE config 1: ~5k sites, zero field, 500 steps (adaptive, reference defaults
    # except dt_init): psi=1 is stationary, so this is plumbing (dt ramps to dt_max).
    big = make_ref_mesh(70, 70)  # 5,791 sites
    o = SolverOptions(solve_time=45.0, dt_init=1e-4, save_every=100)
    s, psi0, _ = make_ref_solver(big, uniform_field_A(big, 0.0), o)
    out = run_reference(s, psi0, o)
    print("zero_field_5k calls:", len(out["call_dt"]))
    save("traj_zero_field_5k", sites=big.sites, elements=big.elements, b=0.0,
         **options_arrays(o), **out)

    # (4b) uniform field, vortices enter (small mesh)
    o = SolverOptions(solve_time=30.0, dt_init=1e-4, save_every=100)
    probes = [small.closest_site((-5, 0)), small.closest_site((5, 0))]
    s, psi0, _ = make_ref_solver(small, uniform_field_A(small, 0.5), o, probe_points=probes)
    out = run_reference(s, psi0, o, snapshot_steps=(0, 50, 200, 500))
    print("field_small calls:", len(out["call_dt"]), "min|psi|^2", (np.abs(out["final_psi"]) ** 2).min())
    save("traj_field_small", b=0.5, probe_points=np.array(probes), **options_arrays(o), **out)

    # (4c) same, fixed time step
    o = SolverOptions(solve_time=2.0, dt_init=5e-3, dt_max=5e-3, adaptive=False, save_every=150)
    s, psi0, _ = make_ref_solver(small, uniform_field_A(small, 0.5), o, probe_points=probes)
    out = run_reference(s, psi0, o, snapshot_steps=(0, 100))
    print("field_small_fixed_dt calls:", len(out["call_dt"]))
    save("traj_field_small_fixed_dt", b=0.5, probe_points=np.array(probes), **options_arrays(o), **out)

    # (4d) transport strip: two terminals, constant current, weak field, thermalisation
    terms = [edge_terminal(strip, "source", -30.0), edge_terminal(strip, "drain", 30.0)]
    probes = [strip.closest_site((-15, 0)), strip.closest_site((15, 0))]
    o = SolverOptions(solve_time=25.0, skip_time=5.0, dt_init=1e-4, save_every=100)
    cur = {"source": 6.0, "drain": -6.0}
    s, psi0, fx = make_ref_solver(strip, uniform_field_A(strip, 0.05), o, terminals=terms,
                                  currents=cur, probe_points=probes)
    out = run_reference(s, psi0, o, snapshot_steps=(10, 300))
    print("transport calls:", len(out["call_dt"]), "V", out["call_mu_probe"][-1])
    term_arrays = {}
    for t in terms:
        term_arrays[f"term_{t.name}_sites"] = t.site_indices
        term_arrays[f"term_{t.name}_edges"] = t.edge_indices
        term_arrays[f"term_{t.name}_boundary_pos"] = t.boundary_edge_indices
        term_arrays[f"term_{t.name}_length"] = t.length
    save("traj_transport_strip", b=0.05, current=6.0, probe_points=np.array(probes),
         fixed_sites=fx, mu_boundary=s.mu_boundary, **term_arrays, **options_arrays(o), **out)

    # (4e) transport with a time-dependent current (ramp) and terminal_psi=None
    o = SolverOptions(solve_time=10.0, dt_init=1e-4, save_every=100, terminal_psi=None)
    ramp = lambda t: {"source": 0.5 * min(t, 8.0), "drain": -0.5 * min(t, 8.0)}  # noqa: E731
    s, psi0, fx = make_ref_solver(strip, uniform_field_A(strip, 0.0), o, terminals=terms,
                                  currents=ramp, probe_points=probes)
    out = run_reference(s, psi0, o)
    print("transport_ramp calls:", len(out["call_dt"]))
    save("traj_transport_ramp", b=0.0, probe_points=np.array(probes), fixed_sites=fx,
         **term_arrays, **options_arrays(o), **out)

    # (4f) a run that needs dt retries: dt_init = dt_max = 2 at high field
    o = SolverOptions(solve_time=40.0, dt_init=2.0, dt_max=2.0, save_every=100)
    s, psi0, _ = make_ref_solver(small, uniform_field_A(small, 0.8), o, epsilon=1.0)
    out = run_reference(s, psi0, o)
    print("retry calls:", len(out["call_dt"]), "dt min/max", out["call_dt"].min(), out["call_dt"].max())
    save("traj_retry_small", b=0.8, **options_arrays(o), **out)

    # (4g) time-dependent applied field (ramp) AND time-dependent epsilon (moving hot spot):
    # exercises dA/dt in the Poisson right-hand side and J_n, and the per-step link update
    # (solver.py:626-648, operators.py:346-383)
    o = SolverOptions(solve_time=8.0, dt_init=1e-4, save_every=100)
    probes = [small.closest_site((-5, 0)), small.closest_site((5, 0))]
    s, psi0, _ = make_ref_solver(small, uniform_field_A(small, 0.0), o, probe_points=probes)
    A_full = uniform_field_A(small, 0.6)

    def ramp_A(x, y, z, *, t=0):
        a2 = min(t / 5.0, 1.0) * A_full
        return np.column_stack([a2, np.zeros(len(a2))])

    def hot_spot(r, *, t=0, vectorized=True):
        c = np.array([-6.0 + 1.5 * t, 1.0])
        return 1.0 - 0.6 * np.exp(-((r - c) ** 2).sum(axis=1) / 4.0)

    s.dynamic_vector_potential = True
    s.applied_vector_potential = ramp_A
    s.A_scale = 1.0
    s.edge_centers = small.edge_mesh.centers
    s.z0 = np.zeros(len(s.edge_centers))
    s.current_A_applied = ramp_A(None, None, None, t=0)[:, :2]
    s.operators.set_link_exponents(s.current_A_applied)
    s.dynamic_epsilon = True
    s.disorder_epsilon = hot_spot
    s.vectorized_epsilon = True
    s.sites = small.sites
    s.epsilon = hot_spot(small.sites, t=0)
    out = run_reference(s, psi0, o, snapshot_steps=(0, 20, 60))
    print("dynamic calls:", len(out["call_dt"]), "min|psi|^2", (np.abs(out["final_psi"]) ** 2).min())
    save("traj_dynamic_small", b_final=0.6, probe_points=np.array(probes), A_full=A_full,
         **options_arrays(o), **out)

    gen_dynamic_lag(small)
    gen_device_meshes()

    # (4h) screening (solver.py:522-578, 654-688; tdgl/solver/screening.py:12-42): the induced
    # vector potential is iterated to self-consistency inside every step.  Tiny mesh: without
    # numba the reference's 1/r double loop is pure Python.
    tiny = make_ref_mesh(12, 9)
    o = SolverOptions(solve_time=1.5, dt_init=1e-3, save_every=100, include_screening=True,
                      screening_tolerance=1e-3, max_iterations_per_step=400)
    probes = [tiny.closest_site((-3, 0)), tiny.closest_site((3, 0))]
    A_tiny = uniform_field_A(tiny, 0.4)
    s, psi0, _ = make_ref_solver(tiny, A_tiny, o, probe_points=probes)
    from types import SimpleNamespace

    s.device = SimpleNamespace(mesh=tiny)
    screening_scale = 0.05  # stands for (mu_0 / 4 pi) K0 / A0 * xi^2 (solver.py:307-309)
    s.areas = screening_scale * tiny.areas
    s.sites = tiny.sites
    s.edge_centers = tiny.edge_mesh.centers
    s.num_edges = len(tiny.edge_mesh.edges)
    s.new_A_induced = np.empty((s.num_edges, 2))
    out = run_reference(s, psi0, o, snapshot_steps=(0, 5))
    print("screening calls:", len(out["call_dt"]), "max |A_ind|", np.abs(out["final_A_induced"]).max())
    save("traj_screening_tiny", b=0.4, screening_scale=screening_scale, probe_points=np.array(probes),
         **mesh_arrays(tiny), **options_arrays(o),
         opt_screening_tolerance=o.screening_tolerance, opt_max_iterations_per_step=o.max_iterations_per_step,
         opt_screening_step_size=o.screening_step_size, opt_screening_step_drag=o.screening_step_drag, **out)

    # ---- (5) runner bookkeeping: thermalise + save_every not dividing the step count ----
    o = SolverOptions(solve_time=1.0, skip_time=0.5, dt_init=1e-3, save_every=7)
    s, psi0, _ = make_ref_solver(small, uniform_field_A(small, 0.3), o, probe_points=probes[:1])
    out = run_reference(s, psi0, o)
    print("bookkeeping calls:", len(out["call_dt"]), "saves:", out["save_step"])
    save("runner_bookkeeping", b=0.3, probe_points=np.array(probes[:1]), **options_arrays(o), **out)


if __name__ == "__main__":
    main()
