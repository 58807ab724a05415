"""Quick start: the script a py-tdgl user would write, with `import tdgl_amd as tdgl`.

A 6 x 3 um strip (xi = 0.5 um, lambda = 2 um, d = 0.1 um) with a round hole, source/drain terminals
on the short edges and two voltage probes; 0.4 mT applied field, 12 uA transport current.
Run on an MI355X:  python examples/quickstart.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-tdgl_amd"))
import tdgl_amd as tdgl  # noqa: E402
from tdgl_amd.geometry import box, circle  # noqa: E402

layer = tdgl.Layer(coherence_length=0.5, london_lambda=2.0, thickness=0.1, gamma=10)
film = tdgl.Polygon("film", points=box(6, 3))
hole = tdgl.Polygon("hole", points=circle(0.6, center=(0.5, 0.2)))
source = tdgl.Polygon("source", points=box(0.02, 3, center=(-3, 0)))
drain = tdgl.Polygon("drain", points=box(0.02, 3, center=(3, 0)))
device = tdgl.Device("strip", layer=layer, film=film, holes=[hole], terminals=[source, drain],
                     probe_points=[(-2, 0), (2, 0)], length_units="um")
device.make_mesh(max_edge_length=0.12, smooth=2)
print(device)

options = tdgl.SolverOptions(solve_time=60, skip_time=20, field_units="mT", current_units="uA", save_every=200)
solution = tdgl.solve(device, options, applied_vector_potential=0.4,
                      terminal_currents=dict(source=12.0, drain=-12.0))

dyn = solution.dynamics
print(f"{solution.stats['steps_thermalizing']} + {solution.stats['steps_simulating']} steps in "
      f"{solution.total_seconds:.2f} s; mu solve: {solution.stats['mu_solver']}"
      + (f", {solution.stats['mean_pcg_iterations']:.1f} PCG iterations per step" if solution.stats['mu_solver'] == "amg_pcg" else ""))
print(f"saved steps: {len(solution.saved_steps)}; min |psi| = {np.abs(solution.tdgl_data.psi).min():.3f}")
print(f"mean voltage between the probes: {dyn.voltage().mean():.4f} V0")
print(f"current through the cut x = 1.5: {solution.current_through_cut(1.5, physical=True):.3f} uA (injected: 12)")
