import sys
sys.path[:0]=['py-tdgl_amd','tests','.']
import numpy as np
from types import SimpleNamespace
from helpers import GAMMA_DEFAULT, U_DEFAULT, synthetic_mesh, uniform_field_A, max_abs, remove_mean
from oracle import OracleSolver, run_time_loop
from tdgl_amd import SolverOptions, TDGLSolver
mesh = synthetic_mesh(226)
A = uniform_field_A(mesh, 0.1)
kw = dict(solve_time=1e9, dt_init=1e-3, save_every=10**6)
solver = TDGLSolver.from_dimensionless(mesh, SolverOptions(**kw), A, 1.0, U_DEFAULT, GAMMA_DEFAULT)
ctx = solver.ctx
ctx.set_state(solver.psi_init, solver.mu_init); ctx.begin_stage()
res = ctx.run(40); got = ctx.get_state()
o = SimpleNamespace(skip_time=0.0, dt_max=0.1, adaptive=True, adaptive_window=10, max_solve_retries=10, adaptive_time_step_multiplier=0.25, terminal_psi=0.0, **kw)
want = run_time_loop(OracleSolver(mesh, A, 1.0, U_DEFAULT, GAMMA_DEFAULT, o), o, max_steps=40)
print('dt', max_abs(res["dt"], want["log"].array("dt"))/res["dt"].max(), 'psi2', max_abs(np.abs(got["psi"])**2, np.abs(want["psi"])**2),
      'js', max_abs(got["supercurrent"], want["supercurrent"]), 'jn', max_abs(got["normal_current"], want["normal_current"]),
      'mu', max_abs(got["mu"], remove_mean(want["mu"])))
