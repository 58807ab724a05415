"""Which in-loop disturbance makes the 9k-site strip of tests/test_hip_direct.py resume its direct solve?
(scenarios: a step of the terminal current through a table; a pulse of the epsilon factor)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from helpers import GAMMA_DEFAULT, U_DEFAULT, edge_terminal, synthetic_mesh, uniform_field_A
from tdgl_amd import SolverOptions, TDGLSolver
from tdgl_amd.hipcore import TDGLContext
from tdgl_amd.parameter import PiecewiseLinear, SeparableEpsilon, TabulatedCurrents

TDGLContext.DENSE_MAX_SITES = 199
TDGLContext.SUB_MAX_SITES = 199
TDGLContext.SUB2_MAX_SITES = 10 ** 9
TDGLContext.SUB2_BLOCK = 60
TDGLContext.SUB2_SUPER = 500
TDGLContext.SUB3_MIN_SITES = 10 ** 9
TDGLContext.DIRECT_SWITCH_MIN_SITES = 0
mesh = synthetic_mesh(160, 48)
terms = [edge_terminal(mesh, "source", -80.0), edge_terminal(mesh, "drain", 80.0)]
I0, T1 = 9.6, 150.0
for name in sys.argv[1:] or ["current16", "eps07", "eps04"]:
    kw = {}
    cur = {"source": I0, "drain": -I0}
    if name.startswith("current"):
        f = float(name[7:]) / 10.0
        cur = TabulatedCurrents([0.0, T1, T1 + 5.0, 1e9], dict(source=[I0, I0, f * I0, f * I0], drain=[-I0, -I0, -f * I0, -f * I0]))
    else:
        f = float(name[3:]) / 10.0
        kw["epsilon"] = SeparableEpsilon(np.ones(len(mesh.sites)), PiecewiseLinear([0.0, T1, T1 + 2.0, T1 + 12.0, T1 + 14.0, 1e9], [1.0, 1.0, f, f, 1.0, 1.0]))
    s = TDGLSolver.from_dimensionless(mesh, SolverOptions(solve_time=1e9, dt_init=1e-4, save_every=10**9), uniform_field_A(mesh, 0.0),
                                      kw.get("epsilon", 1.0), U_DEFAULT, GAMMA_DEFAULT, terminal_info=terms, current_func=cur)
    ctx = s.ctx
    ctx.set_state(s.psi_init, s.mu_init)
    ctx.begin_stage()
    if not isinstance(cur, TabulatedCurrents):
        s.update_mu_boundary(0.0)
    total = 0
    print("==", name, flush=True)
    for k in range(36):
        r = ctx.run(100)
        total += 100
        t = ctx.loop_state()["time"]
        if t > T1 - 12:
            sw = ctx.direct_switching()
            a2 = np.abs(ctx.get_state(mu=False, supercurrent=False, normal_current=False)["psi"]) ** 2
            print(total, round(t, 2), sw, "its mean/max", round(float(r["pcg_iters"].mean()), 2), int(r["pcg_iters"].max()), "dt", round(float(r["dt"][-1]), 4),
                  "min|psi|^2 (free)", round(float(np.sort(a2)[200]), 4), "retries", ctx.step_stats()["psi_retries"], flush=True)
    ctx.close()
