"""Aggregate throughput of several INDEPENDENT small simulations sharing one MI355X: one host thread, one
tdgl_ctx (own HIP stream) per simulation -- the shape of a parameter sweep (I-V curves, field sweeps),
the reference's dominant use.  A 5.8k-site step is ~60 launches of kernels that occupy a few of the 256
CUs for 2-4 us each; concurrent contexts fill the rest.  ctypes releases the GIL during tdgl_run.

    python tools/bench_concurrent.py [side] [steps]          # default 70 (5,791 sites), 400 steps
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import synthetic_mesh, uniform_field_A  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402

side = float(sys.argv[1]) if len(sys.argv) > 1 else 70.0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
mesh = synthetic_mesh(side)
opts = SolverOptions(solve_time=1e12, dt_init=1e-4, save_every=10**9)


def make(b):
    s = TDGLSolver.from_dimensionless(mesh, opts, uniform_field_A(mesh, b), 1.0)
    s.ctx.set_state(s.psi_init, s.mu_init)
    s.ctx.begin_stage()
    s.ctx.run(250)  # past the opening transient
    return s


for n_sim in (() if (len(sys.argv) > 3 and sys.argv[3] == "procs") else (1, 2, 4, 8, 16, 32)):
    sims = [make(0.08 + 0.002 * k) for k in range(n_sim)]  # a small field sweep
    barrier = threading.Barrier(n_sim + 1)
    done = [0] * n_sim

    def work(k):
        barrier.wait()
        done[k] = len(sims[k].ctx.run(steps)["dt"])
        sims[k].ctx.synchronize()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(n_sim)]
    for t in threads:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    print(json.dumps(dict(sites=len(mesh.sites), simulations=n_sim, steps_each=steps, wall_s=round(wall, 4),
                          aggregate_steps_per_s=round(sum(done) / wall, 1), per_simulation_steps_per_s=round(steps / wall, 1))),
          flush=True)
    for s in sims:
        s.ctx.close()

# ---- the same with one PROCESS per simulation (no shared HIP runtime locks): python tools/bench_concurrent.py side steps procs
if len(sys.argv) > 3 and sys.argv[3] == "procs":
    import subprocess

    child = r"""
import os, sys, time, json
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import synthetic_mesh, uniform_field_A
from tdgl_amd import SolverOptions, TDGLSolver
side, steps, b, t_start = float(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
mesh = synthetic_mesh(side)
s = TDGLSolver.from_dimensionless(mesh, SolverOptions(solve_time=1e12, dt_init=1e-4, save_every=10**9), uniform_field_A(mesh, b), 1.0)
s.ctx.set_state(s.psi_init, s.mu_init); s.ctx.begin_stage(); s.ctx.run(250)
while time.time() < t_start: pass
t0 = time.time(); n = len(s.ctx.run(steps)["dt"]); s.ctx.synchronize(); t1 = time.time()
print(json.dumps(dict(t0=t0, t1=t1, n=n)))
"""
    for n_sim in (1, 2, 4, 8, 16):
        t_start = time.time() + 25.0 + 1.0 * n_sim  # every child is set up by then
        procs = [subprocess.Popen([sys.executable, "-c", child, ROOT, str(side), str(steps), str(0.08 + 0.002 * k), str(t_start)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(n_sim)]
        outs = [json.loads(pr.communicate()[0].strip().splitlines()[-1]) for pr in procs]
        wall = max(o["t1"] for o in outs) - min(o["t0"] for o in outs)
        print(json.dumps(dict(sites=len(mesh.sites), processes=n_sim, steps_each=steps, wall_s=round(wall, 4),
                              aggregate_steps_per_s=round(sum(o["n"] for o in outs) / wall, 1))), flush=True)
