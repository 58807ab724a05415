"""Soak run: tens of thousands of steps of a bench workload in batches, watching the counters that should
stay quiet (fp64 restarts, psi retries, non-finite fields) and the device memory in use.

    python tools/soak.py [workload] [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "1M"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
wl = bench.build_workload(name)
opts = SolverOptions(**bench.OPT_KW, pcg_rtol=1e-10, edge_currents_every_step=True)
solver = TDGLSolver.from_dimensionless(wl.mesh, opts, wl.A, 1.0, terminal_info=wl.terms, current_func=wl.currents,
                                       probe_points=wl.probes)
solver.update_mu_boundary(0.0)
ctx = solver.ctx
ctx.set_state(solver.psi_init, np.zeros(wl.n))
ctx.begin_stage()
ctx.step_stats(reset=True)
t0, done = time.perf_counter(), 0
while done < total:
    res = ctx.run(2500)
    done += len(res["dt"])
    a2 = np.abs(ctx.get_state(mu=False, supercurrent=False, normal_current=False)["psi"]) ** 2
    st, ps, ls = ctx.step_stats(), ctx.poisson_stats(), ctx.loop_state()
    print(json.dumps(dict(steps=done, wall_s=round(time.perf_counter() - t0, 2), time=round(ls["time"], 2), dt_last=float(res["dt"][-1]),
                          dt_min=float(res["dt"].min()), pcg_mean=round(float(res["pcg_iters"].mean()), 2), pcg_max=int(res["pcg_iters"].max()),
                          psi_retries=st["psi_retries"], fp64_fallbacks=ps["fp64_fallbacks"], finite=bool(np.all(np.isfinite(a2))),
                          max_abs_sq_psi=float(a2.max()), sites_below_0p1=int((a2 < 0.1).sum()),
                          probes_finite=None if res["mu"] is None else bool(np.all(np.isfinite(res["mu"]))),
                          direct=ctx.direct_stats() if ctx.dense_direct else None,
                          preconditioner=ctx.precond_direct_stats() if getattr(ctx, "precond_direct", None) else None)), flush=True)
    assert np.all(np.isfinite(a2)) and a2.max() < 1.5  # (|psi|^2 overshoots 1 by a few 1e-3 when dt sits at dt_max: the scheme, not an error)
