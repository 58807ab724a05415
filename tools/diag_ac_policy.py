"""The time loop's solver choice under an AC transport current (a strip of ~251k sites, I(t) = I0 (1 + 0.3 sin(2 pi t / T))
through `tdgl_set_mu_boundary_table`): steps/s and switches with the choice on, against the two fixed choices.
    python tools/diag_ac_policy.py [period_in_tau ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from helpers import GAMMA_DEFAULT, U_DEFAULT, edge_terminal, synthetic_mesh, uniform_field_A  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402
from tdgl_amd.parameter import TabulatedCurrents  # noqa: E402

LX, LY = (float(os.environ.get("AC_LX", 920)), float(os.environ.get("AC_LY", 236)))
STEPS = int(os.environ.get("AC_STEPS", 6000))
mesh = synthetic_mesh(LX, LY)
terms = [edge_terminal(mesh, "source", -LX / 2), edge_terminal(mesh, "drain", LX / 2)]
I0 = 0.2 * LY
periods = [float(a) for a in sys.argv[1:]] or [1.6, 6.4, 25.6]
for T in periods:
    t = np.arange(0.0, 1200.0, T / 16.0)
    cur = I0 * (1.0 + 0.3 * np.sin(2 * np.pi * t / T) * np.minimum(1.0, t / 20.0))
    table = TabulatedCurrents(np.append(t, 1e9), dict(source=np.append(cur, cur[-1]), drain=np.append(-cur, -cur[-1])))
    for variant in ("choice", "direct", "amg_pcg"):
        opts = SolverOptions(solve_time=1e9, dt_init=1e-4, save_every=10**9, **(dict(sparse_solver="amg_pcg") if variant == "amg_pcg" else {}))
        s = TDGLSolver.from_dimensionless(mesh, opts, uniform_field_A(mesh, 0.0), 1.0, U_DEFAULT, GAMMA_DEFAULT, terminal_info=terms,
                                          current_func=table)
        ctx = s.ctx
        if variant == "direct":
            ctx.direct_switching(False)
        ctx.set_state(s.psi_init, s.mu_init)
        ctx.begin_stage()
        ctx.run(400)  # (the first steps: dt opens up)
        ctx.synchronize()
        t0 = time.perf_counter()
        its, sw_trace = [], []
        for k in range(STEPS // 500):
            r = ctx.run(500)
            its.append(float(r["pcg_iters"].mean()))
            sw_trace.append(ctx.direct_switching()["switches"] if ctx.dense_direct else 0)
        ctx.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps(dict(period=T, variant=variant, sites=len(mesh.sites), steps_per_s=round(STEPS // 500 * 500 / el, 1),
                              switches=sw_trace[-1], switch_trace=sw_trace, its=[round(x, 2) for x in its],
                              time=round(ctx.loop_state()["time"], 1), retries=ctx.step_stats()["psi_retries"])), flush=True)
        ctx.close()
