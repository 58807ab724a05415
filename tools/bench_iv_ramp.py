"""An I-V ramp on small devices: terminal currents as piecewise-linear tables (TabulatedCurrents), evaluated inside
`tdgl_run`.  Compares the run-ahead loop (tables evaluated on the device at every attempt, one host synchronisation
per batch of up to 64 attempts) with the loop that synchronises once per step (TDGL_NO_RUN_AHEAD=1, tables evaluated
on the host) -- the same trajectory, bit for bit.

    python tools/bench_iv_ramp.py [steps=4000]      -> one JSON line per mesh size and loop
"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for lx, ly in ((60, 24), (120, 56), (300, 140)):
    mesh = synthetic_mesh(lx, ly)
    terms = [edge_terminal(mesh, "source", -lx / 2), edge_terminal(mesh, "drain", lx / 2)]
    probes = [mesh.closest_site((-lx / 4, 0)), mesh.closest_site((lx / 4, 0))]
    i_max = 0.3 * ly
    table = TabulatedCurrents([0.0, 200.0, 1e9], dict(source=[0.0, i_max, i_max], drain=[0.0, -i_max, -i_max]))
    ref = None
    for loop in ("run-ahead", "one synchronisation per step"):
        if loop == "run-ahead":
            os.environ.pop("TDGL_NO_RUN_AHEAD", None)
        else:
            os.environ["TDGL_NO_RUN_AHEAD"] = "1"
        opts = SolverOptions(solve_time=1e9, dt_init=1e-3, save_every=10**6)
        solver = TDGLSolver.from_dimensionless(mesh, opts, uniform_field_A(mesh, 0.02), 1.0, U_DEFAULT, GAMMA_DEFAULT,
                                               terminal_info=terms, current_func=table, probe_points=probes)
        ctx = solver.ctx
        ctx.set_state(solver.psi_init, solver.mu_init)
        ctx.begin_stage()
        ctx.run(200)
        ctx.synchronize()
        ctx.step_stats(reset=True)
        t0 = time.perf_counter()
        res = ctx.run(steps)
        ctx.synchronize()
        el = time.perf_counter() - t0
        st = ctx.step_stats()
        state = ctx.get_state()
        same = None if ref is None else bool(np.array_equal(ref[0], res["dt"]) and np.array_equal(ref[1], state["psi"]))
        ref = ref or (res["dt"], state["psi"])
        print(json.dumps(dict(sites=len(mesh.sites), loop=loop, steps=steps, steps_per_s=round(steps / el, 1),
                              host_syncs_per_step=round(st["host_syncs"] / steps, 3), mu_solver="direct" if ctx.dense_direct else "amg_pcg",
                              end_time=round(ctx.loop_state()["time"], 3), identical_to_run_ahead=same)), flush=True)
        ctx.close()
os.environ.pop("TDGL_NO_RUN_AHEAD", None)
