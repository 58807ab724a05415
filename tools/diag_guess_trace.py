"""Per-step trace of the projection guess in a chosen regime of the headline workload: dt, PCG iterations,
vectors the guess kept, its relative residual, psi retries.

    python tools/diag_guess_trace.py [--workload 1M] [--skip 7500] [--steps 200] [--rtol 1e-9]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="1M")
ap.add_argument("--skip", type=int, default=7500)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--rtol", type=float, default=1e-9)
ap.add_argument("--guess-window", type=int, default=0)
ap.add_argument("--out", default=None)
a = ap.parse_args()
wl = bench.build_workload(a.workload)
opts = SolverOptions(**bench.OPT_KW, pcg_rtol=a.rtol, edge_currents_every_step=True)
solver = TDGLSolver.from_dimensionless(wl.mesh, opts, wl.A, 1.0, terminal_info=wl.terms, current_func=wl.currents)
solver.update_mu_boundary(0.0)
ctx = solver.ctx
ctx.set_poisson_options(rtol=a.rtol, guess_window=a.guess_window)
ctx.set_state(solver.psi_init, np.zeros(wl.n))
ctx.begin_stage()
done = 0
while done < a.skip:
    k = min(2500, a.skip - done)
    ctx.run(k)
    done += k
rows = []
retries0 = ctx.step_stats()["psi_retries"]
for s in range(a.steps):
    r = ctx.run(1)
    g = ctx.guess_stats()
    st = ctx.step_stats()
    rows.append(dict(step=done + s, dt=float(r["dt"][0]), iters=int(r["pcg_iters"][0]), vectors=g["vectors"],
                     relres=g["initial_relres"], retries=st["psi_retries"] - retries0))
    retries0 = st["psi_retries"]
for r in rows:
    print("%6d dt %.5f it %2d vec %2d relres %.2e%s" % (r["step"], r["dt"], r["iters"], r["vectors"], r["relres"], "  RETRY x%d" % r["retries"] if r["retries"] else ""))
it = np.array([r["iters"] for r in rows])
print("mean iterations %.2f; after a retry (next 3 steps) %.2f" % (it.mean(), np.mean([it[i] for i in range(len(it)) if any(rows[j]["retries"] for j in range(max(0, i - 3), i + 1))] or [0])))
if a.out:
    json.dump(rows, open(a.out, "w"))
