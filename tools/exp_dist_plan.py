"""Dev experiment (CPU): what a per-rank aggregation costs in PCG iterations, and how large the ghost sets of the
one-exchange-per-iteration decomposition are.  python tools/exp_dist_plan.py L world"""
import sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from helpers import synthetic_mesh
from tdgl_amd.amg import build_hierarchy, collapsed_operators, pcg_host, vcycle_collapsed_host, smoother_coefficients, fused_restriction
from tdgl_amd.hipcore import poisson_matrix
from tdgl_amd.partition import rcb_partition

L = int(sys.argv[1]) if len(sys.argv) > 1 else 465
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
mesh = synthetic_mesh(L)
em = mesh.edge_mesh
n = len(mesh.sites)
A = poisson_matrix(em.edges, em.dual_edge_lengths / em.edge_lengths, n)
part = rcb_partition(mesh.sites, world)
print("mesh", n, "sites", round(time.time() - t0, 1), "s")

def pcg_collapsed(A, b, h, plan, rtol=1e-10, maxiter=100):
    b = b - b.mean(); x = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b)
    z = vcycle_collapsed_host(h, plan, r, nu_fine=1); p = z.copy(); rz = r @ z; it = 0
    while np.linalg.norm(r) > rtol * bn and it < maxiter:
        q = A @ p; al = rz / (p @ q); x += al * p; r -= al * q; it += 1
        z = vcycle_collapsed_host(h, plan, r, nu_fine=1); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, it, np.linalg.norm(r) / bn

rng = np.random.default_rng(0)
b = rng.standard_normal(n)
res = {}
for name, kw in (("global", {}), ("per-rank", dict(part=part))):
    t0 = time.time()
    h = build_hierarchy(A, max_coarse=600, **kw)
    plan = collapsed_operators(h, 2, "chebyshev", 0.1, tail_cycles=2)
    x, it, rr = pcg_collapsed(A, b, h, plan)
    print(name, "sizes", h.sizes, "iterations", it, "relres %.2e" % rr, round(time.time() - t0, 1), "s", flush=True)
    res[name] = (h, plan)

# ghost sets of the per-rank hierarchy, rank by rank
h, plan = res["per-rank"]
lv0, lv1 = h.levels[0], h.levels[1]
c = smoother_coefficients(lv0.rho, 1, "chebyshev", 0.1)[1][0]
F = fused_restriction(h, c).tocsr()
P0 = lv0.P.tocsr(); A0 = lv0.A.tocsr()
W1, V1 = plan["up"][1]; W1 = W1.tocsr(); M1 = plan["mid"][1].tocsr()
agg_owner = np.zeros(h.sizes[1], dtype=np.int64); agg_owner[lv0.agg] = part  # every member has the same owner
assert (agg_owner[lv0.agg] == part).all()
def cols_of(M, rows):
    sub = M[rows]
    return np.unique(sub.indices)
for r in range(world):
    own = np.flatnonzero(part == r)
    g1 = np.setdiff1d(cols_of(A0, own), own)
    ext1 = np.union1d(own, g1)
    g2 = np.setdiff1d(cols_of(A0, ext1), ext1)
    ext2 = np.union1d(ext1, g2)
    # z on own + g1 needs x on ext2, r on ext2
    x1_rows = cols_of(P0, ext2)
    b1_cols = np.union1d(cols_of(W1, x1_rows), np.flatnonzero(agg_owner == r))
    r_cols = np.union1d(cols_of(F, b1_cols), ext2)
    own1 = np.flatnonzero(agg_owner == r)
    nbrs = np.unique(part[np.setdiff1d(r_cols, own)])
    print(f"rank {r}: own {len(own)} g1 {len(g1)} g2 {len(g2)} deep-r ghosts {len(r_cols) - len(own)} "
          f"({100 * (len(r_cols) - len(own)) / len(own):.1f} %) | level 1: own {len(own1)} x-rows {len(x1_rows)} b-cols {len(b1_cols)} "
          f"| neighbours {len(nbrs)} | variant with a b1 exchange: r ghosts {len(np.union1d(cols_of(F, own1), ext2)) - len(own)}", flush=True)
    if r >= 2 and world > 4:
        break
