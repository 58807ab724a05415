"""Dev experiment: how long after an idle gap does the step rate need to recover?  (GPU clocks / host wake-up.)"""
import sys, time, json
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from helpers import synthetic_mesh, uniform_field_A
from tdgl_amd import SolverOptions, TDGLSolver

mesh = synthetic_mesh(930)
opts = SolverOptions(solve_time=1e9, dt_init=1e-4, dt_max=0.1, save_every=10**9)
s = TDGLSolver.from_dimensionless(mesh, opts, uniform_field_A(mesh, 0.1), 1.0)
ctx = s.ctx
ctx.set_state(s.psi_init, s.mu_init); ctx.begin_stage()
ctx.run(225); ctx.synchronize()
def chunks(label, k=12):
    out = []
    for _ in range(k):
        t0 = time.perf_counter(); ctx.run(10); ctx.synchronize(); out.append(round((time.perf_counter() - t0) * 100, 3))
    print(label, "ms/step per chunk of 10:", out, flush=True)
chunks("back to back")
for gap in (0.02, 0.1, 0.3, 1.0):
    time.sleep(gap); chunks(f"after {gap} s idle")
time.sleep(0.3); ctx.time_kernel(1, 400); chunks("0.3 s idle + 400 K1 launches")
time.sleep(0.3); ctx.time_kernel(1, 3000); chunks("0.3 s idle + 3000 K1 launches")
time.sleep(0.3)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    ctx.time_kernel(5, 50)
chunks("0.3 s idle + 0.3 s of V-cycles")
