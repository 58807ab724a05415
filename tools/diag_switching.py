"""The loop's choice of the mu solve over an I-V staircase on a 251k-site strip: the transport current is raised in
steps (tabulated, evaluated on the device), the state relaxes to a stationary one on every plateau.  Per 1,000 steps:
steps/s, PCG iterations, the solver in charge, switches so far -- with the choice on (product default) and off.

    python tools/diag_switching.py [steps=16000]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import GAMMA_DEFAULT, U_DEFAULT, edge_terminal, synthetic_mesh, uniform_field_A  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402
from tdgl_amd.parameter import TabulatedCurrents  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
lx, ly = 920, 236
mesh = synthetic_mesh(lx, ly)
terms = [edge_terminal(mesh, "source", -lx / 2), edge_terminal(mesh, "drain", lx / 2)]
# plateaus of 300 tau at 0.10, 0.15, 0.20, 0.25 Ly, ramps of 20 tau between them
t, i = [0.0], [0.0]
for k, frac in enumerate((0.10, 0.15, 0.20, 0.25)):
    t += [t[-1] + 20.0, t[-1] + 320.0]
    i += [frac * ly, frac * ly]
t.append(1e9)
i.append(i[-1])
table = TabulatedCurrents(t, dict(source=i, drain=[-v for v in i]))
for choice in (True, False):
    opts = SolverOptions(solve_time=1e9, dt_init=1e-3, save_every=10**6)
    s = TDGLSolver.from_dimensionless(mesh, opts, uniform_field_A(mesh, 0.0), 1.0, U_DEFAULT, GAMMA_DEFAULT, terminal_info=terms,
                                      current_func=table)
    ctx = s.ctx
    ctx.direct_switching(choice)
    ctx.set_state(s.psi_init, s.mu_init)
    ctx.begin_stage()
    t_all = time.perf_counter()
    done = 0
    while done < steps:
        t0 = time.perf_counter()
        res = ctx.run(1000)
        ctx.synchronize()
        el = time.perf_counter() - t0
        done += 1000
        sw = ctx.direct_switching()
        print(json.dumps(dict(choice=choice, steps=done, time=round(ctx.loop_state()["time"], 1), steps_per_s=round(1000 / el),
                              pcg_mean=round(float(res["pcg_iters"].mean()), 2), dt_last=round(float(res["dt"][-1]), 4), **sw)), flush=True)
    print(json.dumps(dict(choice=choice, total_steps=done, wall_s=round(time.perf_counter() - t_all, 2),
                          steps_per_s=round(done / (time.perf_counter() - t_all)))), flush=True)
    ctx.close()
