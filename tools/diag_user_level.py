"""The quick-start device through `tdgl.solve` at two resolutions (3.9k sites: dense inverse; 37k sites:
substructured solve) and, for comparison, with the iterative solve forced: physical read-outs must agree."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-tdgl_amd"))
import tdgl_amd as tdgl  # noqa: E402
from tdgl_amd.geometry import box, circle  # noqa: E402
from tdgl_amd.hipcore import TDGLContext  # noqa: E402

layer = tdgl.Layer(coherence_length=0.5, london_lambda=2.0, thickness=0.1, gamma=10)
film = tdgl.Polygon("film", points=box(6, 3))
hole = tdgl.Polygon("hole", points=circle(0.6, center=(0.5, 0.2)))
source = tdgl.Polygon("source", points=box(0.02, 3, center=(-3, 0)))
drain = tdgl.Polygon("drain", points=box(0.02, 3, center=(3, 0)))
product = (TDGLContext.DENSE_MAX_SITES, TDGLContext.SUB_MAX_SITES)
for h, solve_time in ((0.12, 40), (0.04, 10)):
    device = tdgl.Device("strip", layer=layer, film=film, holes=[hole], terminals=[source, drain],
                         probe_points=[(-2, 0), (2, 0)], length_units="um")
    device.make_mesh(max_edge_length=h, smooth=2)
    out = {}
    for mode, limits in (("product default", product), ("amg_pcg", (0, 0))):
        TDGLContext.DENSE_MAX_SITES, TDGLContext.SUB_MAX_SITES = limits
        options = tdgl.SolverOptions(solve_time=solve_time, skip_time=5, field_units="mT", current_units="uA", save_every=200,
                                     pcg_rtol=1e-12)
        t0 = time.perf_counter()
        sol = tdgl.solve(device, options, applied_vector_potential=0.4, terminal_currents=dict(source=12.0, drain=-12.0))
        wall = time.perf_counter() - t0
        steps = sol.stats["steps_thermalizing"] + sol.stats["steps_simulating"]
        out[mode] = sol
        print(f"{len(device.mesh.sites)} sites, {mode}: {steps} steps in {sol.total_seconds:.2f} s (wall {wall:.2f} s incl. set-up), "
              f"{sol.stats['mean_pcg_iterations']:.1f} PCG it/step, V = {sol.dynamics.voltage().mean():.6f}, "
              f"I(x=1.5) = {sol.current_through_cut(1.5, physical=True):.4f} uA, min|psi| = {np.abs(sol.tdgl_data.psi).min():.4f}", flush=True)
    a, b = out["product default"], out["amg_pcg"]
    n = min(len(a.dynamics.dt), len(b.dynamics.dt))
    print("   same number of steps:", len(a.dynamics.dt) == len(b.dynamics.dt), " max |dt difference| over the common steps / max dt:",
          float(np.abs(a.dynamics.dt[:n] - b.dynamics.dt[:n]).max() / b.dynamics.dt.max()),
          " max ||psi|^2 difference|:", float(np.abs(np.abs(a.tdgl_data.psi) ** 2 - np.abs(b.tdgl_data.psi) ** 2).max()))
TDGLContext.DENSE_MAX_SITES, TDGLContext.SUB_MAX_SITES = product
