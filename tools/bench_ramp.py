"""Dev benchmark: a field ramp A(t) = LinearRamp(t) * A_base, three ways (per-step upload of A(t),
device-side scaling per step, ramp evaluated inside tdgl_run).  python tools/bench_ramp.py [L]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from helpers import synthetic_mesh, uniform_field_A, U_DEFAULT, GAMMA_DEFAULT  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 465
mesh = synthetic_mesh(L)
A_base = uniform_field_A(mesh, 0.1)
ramp = dict(tmin=0.0, tmax=50.0, initial=0.0, final=1.0)
scale = lambda t: min(max(t / 50.0, 0.0), 1.0)  # noqa: E731
for path in ("upload_per_step", "scale_per_step", "native_ramp"):
    opts = SolverOptions(solve_time=1e9, dt_init=1e-4, dt_max=1e-1, save_every=10**9)
    kw = dict(vector_potential_ramp=(A_base, ramp)) if path == "native_ramp" else dict(
        vector_potential_func=lambda t: scale(t) * A_base)
    s = TDGLSolver.from_dimensionless(mesh, opts, 0.0 * A_base, 1.0, U_DEFAULT, GAMMA_DEFAULT, **kw)
    if path == "scale_per_step":
        s._A_base, s._A_factor = A_base, scale
        s.ctx.set_link_exponents_base(A_base, 0.0)
    ctx = s.ctx
    ctx.set_state(s.psi_init, s.mu_init)
    ctx.begin_stage()

    def advance(k):
        if path == "native_ramp":
            ctx.run(k)
            return
        for _ in range(k):
            ls = ctx.loop_state()
            s.update_dynamic_inputs(ls["time"], ls["dt"])
            ctx.run(1)

    advance(60)
    ctx.synchronize()
    t0 = time.perf_counter()
    advance(200)
    ctx.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps(dict(path=path, sites=len(mesh.sites), steps_per_s=round(200 / el, 1), time=ctx.loop_state()["time"])), flush=True)
