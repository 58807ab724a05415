"""Dev experiment (GPU): contraction per PCG iteration on a fixed pseudo-random right-hand side as a function of the
hierarchy's knobs (prolongator-smoothing omega, Chebyshev interval, smoother degree) -- the figure that predicts the
iterations per step of the time loop (profiles/EXPERIMENTS.md, round 5)."""
import sys, time, json
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from helpers import synthetic_mesh
from tdgl_amd.amg import build_hierarchy
from tdgl_amd.hipcore import TDGLContext, poisson_matrix

def sweep(mesh, label):
    ctx = TDGLContext(mesh, direct_solve=False)
    k = ctx._keep
    A = poisson_matrix(k["edges"].astype(np.int64), k["dl"] / k["el"], ctx.n, ctx.iperm)
    probe = np.random.default_rng(2024).standard_normal(ctx.n); probe -= probe.mean()
    rhs = probe / k["areas"]
    for seed in (0, 2):
        row = []
        for omega in (4.0 / 3.0, 1.4, 1.45, 1.5, 1.55, 1.6):
            h = build_hierarchy(A, max_coarse=600, seed=seed, omega=omega)
            ctx._shipped_plan = None
            ctx.set_hierarchy(h)
            ctx.set_poisson_options(rtol=1e-10)
            _, its, rel = ctx.poisson_solve(rhs)
            row.append(f"{omega:.3f}: {its} its {rel ** (1.0 / its):.4f}")
        print(label, "seed", seed, "|", " | ".join(row), flush=True)
    ctx.close()

sweep(synthetic_mesh(930), "1M square")
sweep(synthetic_mesh(465), "250k square")
sweep(synthetic_mesh(1300, 333), "500k strip")
