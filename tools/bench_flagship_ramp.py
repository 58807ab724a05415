"""The reference's flagship workload shape through the public API (docs/notebooks/logo.ipynb cell 15: a ~4k-site
polygon device, A(t) = LinearRamp(0 -> 100 tau) * ConstantField, held to t = 800; 12 min 35 s there): steps/s while the
field ramps (one host synchronisation per step: the loop evaluates the ramp) and after it has ended (the run-ahead loop
takes over two steps after t_max, run.inc: ramp_settled), and the wall time of the whole run.
    python tools/bench_flagship_ramp.py [max_edge_length] > profiles/r05_flagship_ramp.json"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
import tdgl_amd as tdgl  # noqa: E402
from tdgl_amd.geometry import box, circle  # noqa: E402

h = float(sys.argv[1]) if len(sys.argv) > 1 else 0.42
layer = tdgl.Layer(coherence_length=0.5, london_lambda=2.0, thickness=0.1)
film = tdgl.Polygon("film", points=box(30, 20))
holes = [tdgl.Polygon("h1", points=circle(2.5, center=(-7, 2))), tdgl.Polygon("h2", points=circle(1.5, center=(6, -3)))]
device = tdgl.Device("flagship", layer=layer, film=film, holes=holes, length_units="um")
t0 = time.perf_counter()
device.make_mesh(max_edge_length=h)
mesh_s = time.perf_counter() - t0
n = len(device.mesh.sites)
ramp = tdgl.LinearRamp(tmin=0, tmax=100) * tdgl.ConstantField(0.6, field_units="mT", length_units="um")
out = dict(workload="polygon device with two holes, LinearRamp(0 -> 100 tau) * ConstantField(0.6 mT), solve_time 800 "
                    "(the shape of the reference's docs/notebooks/logo.ipynb cell 15)", sites=n, mesh_s=round(mesh_s, 2))
for label, solve_time in (("whole_run", 800.0),):
    opts = tdgl.SolverOptions(solve_time=solve_time, field_units="mT", save_every=1000)
    t0 = time.perf_counter()
    sol = tdgl.solve(device, opts, applied_vector_potential=ramp)
    wall = time.perf_counter() - t0
    st = dict(getattr(sol, "stats", None) or {})
    out[label] = dict(wall_s=round(wall, 2), stats={k: (float(v) if isinstance(v, (int, float, np.floating)) else str(v)) for k, v in st.items()})
# the two regimes separately, on the solver object: steps/s inside the ramp and after it
from tdgl_amd.solver import TDGLSolver  # noqa: E402

solver = TDGLSolver(device, tdgl.SolverOptions(solve_time=800.0, field_units="mT", save_every=10**9), applied_vector_potential=ramp)
ctx = solver.ctx
ctx.set_state(solver.psi_init, solver.mu_init)
ctx.begin_stage()
def rate(end_time):
    ctx.synchronize()
    s0, t0 = ctx.step_stats(), time.perf_counter()
    ctx.run(10**7, end_time=end_time)
    ctx.synchronize()
    el = time.perf_counter() - t0
    s1 = ctx.step_stats()
    k = s1["steps"] - s0["steps"]
    return dict(steps=int(k), wall_s=round(el, 3), steps_per_s=round(k / el, 1), host_syncs_per_step=round((s1["host_syncs"] - s0["host_syncs"]) / max(k, 1), 3),
                time=round(ctx.loop_state()["time"], 2))
out["inside_ramp_t_5_to_100"] = (rate(5.0), rate(100.0))[1]
out["after_ramp_t_100_to_800"] = rate(800.0)
out["link_scale_at_end"] = ctx.link_scale()
print(json.dumps(out))
