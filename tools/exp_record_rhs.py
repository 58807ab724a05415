"""Experiment (CPU, no GPU): record the Poisson right-hand sides and solutions of an oracle trajectory
so that initial-guess / deflation strategies for the mu solve can be evaluated offline.

    python tools/exp_record_rhs.py SIDE STEPS OUT.npz [start-every]

Writes b_n = -a * rhs_n (the symmetric form the HIP path solves, A = -diag(a) L_mu), the exact mean-free
mu_n (SuperLU), dt_n and the mesh's Poisson matrix.
"""
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "py-tdgl_amd")
from bench import OPT_KW, uniform_A  # noqa: E402
from oracle import OracleSolver  # noqa: E402
from tdgl_amd.finite_volume import Mesh  # noqa: E402
from tdgl_amd.meshgen import hex_jitter_points, triangulate  # noqa: E402

side, steps, out = float(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
keep_from = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pts = hex_jitter_points(side, side)
mesh = Mesh.from_triangulation(pts, triangulate(pts))
n = len(mesh.sites)
A = uniform_A(mesh, 0.1)
o = SimpleNamespace(skip_time=0.0, terminal_psi=0.0, **OPT_KW)
solver = OracleSolver(mesh, A, 1.0, 5.79, 10.0, o)
ops = solver.operators
lu = ops.mu_laplacian_lu
rec_b, rec_mu = [], []


def hooked(rhs):
    mu = lu(rhs)
    rec_b.append(-mesh.areas * rhs)
    rec_mu.append(mu - mu.mean())
    return mu


ops.mu_laplacian_lu = hooked
psi = np.ones(n, dtype=complex)
mu = np.zeros(n)
t, dt = 0.0, o.dt_init
dts = []
t0 = time.time()
for k in range(steps):
    new_dt, psi, mu, js, jn = solver.update({"step": k, "time": t, "dt": dt}, None, dt, psi=psi, mu=mu)
    dts.append(new_dt)
    dt = new_dt
    t += dt
    if k % 100 == 0:
        print(k, "t", t, "dt", dt, "min|psi|2", (abs(psi) ** 2).min(), time.time() - t0, flush=True)
sl = slice(keep_from, None)
np.savez(out, b=np.array(rec_b[sl]), mu=np.array(rec_mu[sl]), dt=np.array(dts[sl]), sites=mesh.sites,
         edges=mesh.edge_mesh.edges, w=mesh.edge_mesh.dual_edge_lengths / mesh.edge_mesh.edge_lengths, areas=mesh.areas)
print("saved", out, len(rec_b) - keep_from, "steps")
