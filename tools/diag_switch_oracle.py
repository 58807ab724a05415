"""When does the 9k-site strip of tests/test_hip_direct.py pause its direct solve?  (step / simulated time of the switch,
so that the oracle-checked switching test can place its current step behind it.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from helpers import GAMMA_DEFAULT, U_DEFAULT, edge_terminal, synthetic_mesh, uniform_field_A
from tdgl_amd import SolverOptions, TDGLSolver
from tdgl_amd.hipcore import TDGLContext

TDGLContext.DENSE_MAX_SITES = 199
TDGLContext.SUB_MAX_SITES = 199
TDGLContext.SUB2_MAX_SITES = 10 ** 9
TDGLContext.SUB2_BLOCK = 60
TDGLContext.SUB2_SUPER = 500
TDGLContext.SUB3_MIN_SITES = 10 ** 9
TDGLContext.DIRECT_SWITCH_MIN_SITES = 0
mesh = synthetic_mesh(160, 48)
terms = [edge_terminal(mesh, "source", -80.0), edge_terminal(mesh, "drain", 80.0)]
s = TDGLSolver.from_dimensionless(mesh, SolverOptions(solve_time=1e9, dt_init=1e-4, save_every=10**9), uniform_field_A(mesh, 0.0),
                                  1.0, U_DEFAULT, GAMMA_DEFAULT, terminal_info=terms, current_func={"source": 9.6, "drain": -9.6})
ctx = s.ctx
ctx.set_state(s.psi_init, s.mu_init)
ctx.begin_stage()
s.update_mu_boundary(0.0)
total = 0
for k in range(40):
    r = ctx.run(200)
    total += 200
    sw = ctx.direct_switching()
    print(total, round(ctx.loop_state()["time"], 3), sw, float(r["pcg_iters"].mean()), float(r["dt"][-1]), flush=True)
