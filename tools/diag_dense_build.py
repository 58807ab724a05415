"""Time the device-side build of the dense pseudo-inverse for several contexts in one process."""
import sys, time
sys.path[:0] = ["py-tdgl_amd", "tests"]
import numpy as np
from helpers import synthetic_mesh, uniform_field_A
from tdgl_amd import SolverOptions, TDGLSolver
from tdgl_amd.hipcore import TDGLContext

for side in (3, 22, 70, 88, 101, 117):
    mesh = synthetic_mesh(side)
    t0 = time.perf_counter()
    opts = SolverOptions(solve_time=1e9, dt_init=1e-4, save_every=1000)
    solver = TDGLSolver.from_dimensionless(mesh, opts, uniform_field_A(mesh, 0.1), 1.0)
    t1 = time.perf_counter()
    ctx = solver.ctx
    relres = ctx.poisson_solve(np.random.default_rng(1).standard_normal(ctx.n))[2]
    ctx.set_state(solver.psi_init, solver.mu_init)
    ctx.begin_stage()
    ctx.run(200)
    t2 = time.perf_counter()
    ctx.run(1000)
    t3 = time.perf_counter()
    print(len(mesh.sites), "direct", ctx.dense_direct, "relres %.2e" % relres, "setup %.3f s" % (t1 - t0), {k: round(v, 3) for k, v in ctx.setup_times.items()},
          "first 200 steps %.3f s, next 1000 steps %.3f s" % (t2 - t1, t3 - t2), flush=True)
    ctx.close()
