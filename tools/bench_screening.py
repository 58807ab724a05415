"""Dev benchmark: the induced-vector-potential kernel (screening) alone, and whole screening steps.

    python tools/bench_screening.py [L ...]     # film side lengths in xi (L=100 -> 11.8k sites)

Reports pairs/s and fp64 TFLOP/s at 12 algorithmic flops per (edge, site) pair
(tdgl/solver/screening.py:35-42: 2 sub, 2 mul, 1 add, 1 sqrt, 2 mul, 2 div, 2 add)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from helpers import synthetic_mesh, uniform_field_A, U_DEFAULT, GAMMA_DEFAULT  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402

for L in [int(a) for a in sys.argv[1:]] or [100, 200]:
    mesh = synthetic_mesh(L)
    n, m = len(mesh.sites), len(mesh.edge_mesh.edges)
    # fixed dt: with the reference's loop every screening iteration advances psi by dt again, so
    # the adaptive controller's large steps stall the iteration (reference and oracle alike)
    opts = SolverOptions(solve_time=1e9, dt_init=1e-3, dt_max=1e-3, adaptive=False, save_every=10**9, include_screening=True,
                         screening_tolerance=1e-3, max_iterations_per_step=1000)
    s = TDGLSolver.from_dimensionless(
        mesh, opts, uniform_field_A(mesh, 0.2), 1.0, U_DEFAULT, GAMMA_DEFAULT,
        screening=dict(sites=mesh.sites, edge_centers=mesh.edge_mesh.centers, areas=0.004 * mesh.areas))
    ctx = s.ctx
    ctx.set_state(s.psi_init, s.mu_init)
    ctx.begin_stage()
    ms = ctx.time_kernel(7, reps=5)
    pairs = float(n) * m
    tf = 12 * pairs / (ms * 1e-3) / 1e12
    # (the kernel is bound by the fp64 VECTOR rate, 78.6 TFLOP/s on MI355X: a reciprocal square root per pair, not a contraction)
    roofline = dict(bound="fp64 vector", kernel="k_induced_vector_potential", achieved=round(tf, 2), peak=78.6, unit="TFLOP/s",
                    frac=round(tf / 78.6, 4), flops_per_pair=12, pairs=pairs, avg_launch_ms=ms)
    print(json.dumps(dict(L=L, sites=n, edges=m, kernel_ms=ms, pairs_per_s=pairs / (ms * 1e-3), tflops_fp64=tf,
                          roofline_screening=roofline)), flush=True)
    t0 = time.perf_counter()
    try:
        res = ctx.run(20)
    except RuntimeError as exc:
        print("steps failed:", exc)
        continue
    wall = time.perf_counter() - t0
    iters = int(res["screening_iterations"].sum())
    print(json.dumps(dict(L=L, sites=n, edges=m, kernel_ms=ms, pairs_per_s=pairs / (ms * 1e-3),
                          tflops_fp64=12 * pairs / (ms * 1e-3) / 1e12, steps=20, screening_iterations=iters,
                          ms_per_screening_iteration=1e3 * wall / max(iters, 1),
                          kernel_share=ms * iters / (1e3 * wall))))
