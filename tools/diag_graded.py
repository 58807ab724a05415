"""Iteration counts of the mu solve on the graded ring mesh (tests/test_hip_parity.py) for every storage of
the V-cycle's operators and both CG betas: python tools/diag_graded.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import uniform_field_A  # noqa: E402
from tdgl_amd import SolverOptions, TDGLSolver  # noqa: E402
from test_hip_parity import _graded_ring_mesh  # noqa: E402

mesh = _graded_ring_mesh()
solver = TDGLSolver.from_dimensionless(mesh, SolverOptions(solve_time=1.0), uniform_field_A(mesh, 0.3), 1.0)
ctx = solver.ctx
rng = np.random.default_rng(3)
for label, rhs in (("rough", rng.normal(size=ctx.n) / np.sqrt(mesh.areas)), ("smooth", np.sin(mesh.sites[:, 0]) * np.cos(0.5 * mesh.sites[:, 1]))):
    rhs = rhs - (rhs * mesh.areas).sum() / mesh.areas.sum()
    for name, kw in (("fp64", dict(precond_fp32=False)), ("fp32 flex", dict(precond_fp32=1)), ("fp32 FR", dict(precond_fp32=1, flexible_cg=False)),
                     ("f16 flex", dict(precond_fp32=True)), ("f16 FR", dict(precond_fp32=True, flexible_cg=False))):
        ctx.set_poisson_options(rtol=1e-11, max_iter=300, **kw)
        mu, iters, relres = ctx.poisson_solve(rhs)
        print(f"{label:7s} {name:10s} iterations {iters:3d} relres {relres:.2e} storage {ctx.precond_storage()}", flush=True)
