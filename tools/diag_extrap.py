"""Dev diagnostic: PCG behaviour on the strip+terminals case for each extrapolation order."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
import numpy as np
from helpers import *
from tdgl_amd import SolverOptions, TDGLSolver
mesh = synthetic_mesh(60, 15)
terms = [edge_terminal(mesh, "source", -30.0), edge_terminal(mesh, "drain", 30.0)]
for ex in (2,):
    opts = SolverOptions(solve_time=1e9, dt_init=1e-4, save_every=1000, pcg_rtol=1e-11, pcg_max_iter=60)
    s = TDGLSolver.from_dimensionless(mesh, opts, uniform_field_A(mesh, 0.05), 1.0, terminal_info=terms,
                                      current_func={"source": 6.0, "drain": -6.0})
    s.ctx.set_poisson_options(rtol=1e-11, extrapolate=ex, max_iter=60)
    s.ctx.set_state(s.psi_init, s.mu_init); s.ctx.begin_stage(); s.update_mu_boundary(0.0)
    prev = None
    for k in range(30):
        try:
            r = s.ctx.run(1)
            st = s.ctx.get_state(supercurrent=False, normal_current=False)
            print(ex, "step", k, "dt", r["dt"][-1], "iters", r["pcg_iters"].tolist(), "|mu|max", np.abs(st["mu"]).max(), "mean", st["mu"].mean())
        except RuntimeError as e:
            st = s.ctx.get_state(supercurrent=False, normal_current=False)
            print(ex, "step", k, "FAILED:", e, "|mu|max", np.abs(st["mu"]).max(), "mean", st["mu"].mean())
            break
