"""Dev diagnostic: first step at which the run-ahead loop with the ramp evaluated on the device leaves the loop with one
synchronisation per step (TDGL_NO_RUN_AHEAD), and in which field."""
import os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from helpers import synthetic_mesh, uniform_field_A, U_DEFAULT, GAMMA_DEFAULT
from tdgl_amd import SolverOptions, TDGLSolver

mesh = synthetic_mesh(40, 30)
A_base = uniform_field_A(mesh, 0.6)
ramp = dict(tmin=0.0, tmax=3.0, initial=0.0, final=1.0)
def make(classic):
    if classic: os.environ["TDGL_NO_RUN_AHEAD"] = "1"
    else: os.environ.pop("TDGL_NO_RUN_AHEAD", None)
    opts = SolverOptions(solve_time=1e9, dt_init=1e-3, dt_max=0.1, save_every=10**6)
    s = TDGLSolver.from_dimensionless(mesh, opts, 0.0 * A_base, 1.0, U_DEFAULT, GAMMA_DEFAULT, vector_potential_ramp=(A_base, ramp))
    s.ctx.set_state(s.psi_init, s.mu_init); s.ctx.begin_stage()
    return s
a, b = make(False), make(True)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for k in range(0, 400, chunk):
    ra, rb = a.ctx.run(chunk), b.ctx.run(chunk)
    sa, sb = a.ctx.get_state(), b.ctx.get_state()
    diffs = {key: float(np.abs(sa[key] - sb[key]).max()) for key in sa}
    ddt = float(np.abs(ra["dt"] - rb["dt"]).max())
    if ddt or any(diffs.values()):
        print("first difference after step", k + chunk, "dt", ddt, diffs, "scale", a.ctx.link_scale(), b.ctx.link_scale(),
              "time", a.ctx.loop_state(), b.ctx.loop_state(), "retries", a.ctx.step_stats()["psi_retries"], b.ctx.step_stats()["psi_retries"])
        break
else:
    print("identical over 400 steps; time", a.ctx.loop_state()["time"], "retries", a.ctx.step_stats()["psi_retries"])
