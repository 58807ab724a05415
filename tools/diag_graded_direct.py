"""Direct mu solves on the graded ring mesh (600 : 1 edge lengths): residuals and how fast the adaptive
trajectory separates from the AMG-PCG one."""
import sys
sys.path[:0] = ["py-tdgl_amd", "tests", "."]
import numpy as np
import test_hip_parity as P
from helpers import uniform_field_A, max_abs
from tdgl_amd import SolverOptions, TDGLSolver
from tdgl_amd.hipcore import TDGLContext

mesh = P._graded_ring_mesh()
n = len(mesh.sites)
out = {}
for m, lim in (("dense", (20000, 0)), ("substructured", (199, 10 ** 9)), ("pcg", (0, 0)), ("pcg_1e-13", (0, 0))):
    TDGLContext.DENSE_MAX_SITES, TDGLContext.SUB_MAX_SITES = lim
    solver = TDGLSolver.from_dimensionless(mesh, SolverOptions(solve_time=1e9, dt_init=1e-5, save_every=10**6,
                                                               pcg_rtol=1e-13 if m.endswith("13") else 1e-12), uniform_field_A(mesh, 0.05), 1.0)
    ctx = solver.ctx
    rhs = np.random.default_rng(4).standard_normal(n)
    rhs -= (rhs * mesh.areas).sum() / mesh.areas.sum()
    mu, iters, relres = ctx.poisson_solve(rhs)
    ctx.set_state(solver.psi_init, solver.mu_init); ctx.begin_stage()
    res = ctx.run(100)
    out[m] = (mu, relres, res["dt"], ctx.step_stats()["psi_retries"])
    print(m, "direct", ctx.dense_direct, "relres %.2e" % relres, "iters", iters, "retries", out[m][3], flush=True)
ref = out["pcg_1e-13"]
for m in ("dense", "substructured", "pcg"):
    mu, rel, dt, _ = out[m]
    d = np.abs(dt - ref[2]) / ref[2]
    first = int(np.argmax(d > 1e-6)) if np.any(d > 1e-6) else -1
    print(m, "mu vs pcg_1e-13: %.2e" % (max_abs(mu, ref[0]) / np.abs(ref[0]).max()), "first step with dt off by 1e-6:", first,
          "max rel dt diff first 10/20/100: %.1e %.1e %.1e" % (d[:10].max(), d[:20].max(), d.max()))
