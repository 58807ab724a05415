"""A large polygon device through the user-level API (`Device.make_mesh` + `tdgl.solve`): where the wall time goes.

    python tools/diag_large_device.py [WIDTH=260] [HEIGHT=160] [MAX_EDGE=0.45] [SOLVE_TIME=3]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-tdgl_amd"))
import tdgl_amd as tdgl  # noqa: E402
from tdgl_amd.geometry import box, circle  # noqa: E402

w, hgt, h, solve_time = [float(a) for a in (sys.argv[1:5] + ["260", "160", "0.45", "3"][len(sys.argv) - 1:])]
layer = tdgl.Layer(coherence_length=1.0, london_lambda=2.0, thickness=0.1, gamma=10)
film = tdgl.Polygon("film", points=box(w, hgt))
holes = [tdgl.Polygon("hole0", points=circle(0.06 * hgt, center=(0.2 * w, 0.1 * hgt))),
         tdgl.Polygon("hole1", points=box(0.1 * w, 0.05 * hgt, center=(-0.25 * w, -0.2 * hgt)))]
source = tdgl.Polygon("source", points=box(0.02, 0.6 * hgt, center=(-w / 2, 0)))
drain = tdgl.Polygon("drain", points=box(0.02, 0.6 * hgt, center=(w / 2, 0)))
for smooth in (0, 2):
    device = tdgl.Device("plate", layer=layer, film=film, holes=holes, terminals=[source, drain],
                         probe_points=[(-0.3 * w, 0), (0.3 * w, 0)], length_units="um")
    t0 = time.perf_counter()
    device.make_mesh(max_edge_length=h, smooth=smooth)
    t_mesh = time.perf_counter() - t0
    mesh = device.mesh
    print(f"smooth={smooth}: {len(mesh.sites)} sites, {len(mesh.edge_mesh.edges)} edges, make_mesh {t_mesh:.2f} s; "
          f"min dual length {mesh.edge_mesh.dual_edge_lengths.min():.3g}, min area {mesh.areas.min():.3g}", flush=True)
    options = tdgl.SolverOptions(solve_time=solve_time, field_units="mT", current_units="uA", save_every=500)
    t0 = time.perf_counter()
    sol = tdgl.solve(device, options, applied_vector_potential=0.02, terminal_currents=dict(source=5.0, drain=-5.0))
    wall = time.perf_counter() - t0
    steps = sol.stats["steps_thermalizing"] + sol.stats["steps_simulating"]
    print(f"   tdgl.solve: {steps} steps, {sol.total_seconds:.2f} s in the time loop, {wall:.2f} s wall (set-up {wall - sol.total_seconds:.2f} s), "
          f"{sol.stats['mean_pcg_iterations']:.1f} PCG it/step, mu solver {sol.stats.get('mu_solver')}, "
          f"min|psi| {np.abs(sol.tdgl_data.psi).min():.3f}, I(x=0) = {sol.current_through_cut(0.0, physical=True):.4f} uA", flush=True)
