"""Where the host set-up time goes (no GPU needed): meshing, dual mesh, AMG hierarchy, collapsed operators.

    python tools/diag_setup.py [SIDE=1000] [--profile]
"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "py-tdgl_amd")
from tdgl_amd.amg import build_hierarchy, collapsed_operators, fused_level_operators, fused_restriction, smoother_coefficients  # noqa: E402
from tdgl_amd.finite_volume import Mesh  # noqa: E402
from tdgl_amd.hipcore import poisson_matrix  # noqa: E402
from tdgl_amd.meshgen import hex_jitter_points, triangulate  # noqa: E402

side = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1000.0
prof = cProfile.Profile() if "--profile" in sys.argv else None


def timed(label, f, *a, **k):
    t = time.perf_counter()
    out = f(*a, **k)
    print(f"{label:28s} {time.perf_counter() - t:7.2f} s", flush=True)
    return out


if prof:
    prof.enable()
pts = timed("points", hex_jitter_points, side, side)
tri = timed("Delaunay (Qhull)", triangulate, pts)
mesh = timed("dual mesh", Mesh.from_triangulation, pts, tri)
n = len(pts)
em = mesh.edge_mesh
A = timed("Poisson matrix", poisson_matrix, em.edges.astype(np.int64), em.dual_edge_lengths / em.edge_lengths, n, np.arange(n))
h = timed("AMG hierarchy", build_hierarchy, A, max_coarse=600)
print("   levels", h.sizes)
timed("collapsed coarse chain", collapsed_operators, h, 2, "chebyshev", 0.1, tail_cycles=2)
timed("pre-multiplied levels", lambda: [fused_level_operators(h.levels[k]) for k in range(1, len(h.levels) - 1)])
c = smoother_coefficients(h.levels[0].rho, 1, "chebyshev", 0.1)[1][0]
timed("pre-multiplied restriction", fused_restriction, h, c)
if prof:
    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(30)
