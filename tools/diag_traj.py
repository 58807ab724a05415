"""Dev diagnostic: growth of the HIP-vs-reference deviation along each golden trajectory."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
import numpy as np
from conftest import load_golden
from helpers import *
import test_hip_parity as T

def run(name):
    g = load_golden(name)
    b = float(g["b"])
    if "transport" in name:
        mesh = reference_mesh(load_golden("mesh_strip"))
        terms = [edge_terminal(mesh, "source", -30.0), edge_terminal(mesh, "drain", 30.0)]
        if "ramp" in name:
            cf = lambda t: {"source": 0.5 * min(t, 8.0), "drain": -0.5 * min(t, 8.0)}
        else:
            cur = float(g["current"]); cf = {"source": cur, "drain": -cur}
        sol = T._hip_solver(g, mesh, b, terminals=terms, current_func=cf).solve()
    elif "5k" in name:
        sol = T._hip_solver(g, synthetic_mesh(70), b).solve()
    else:
        sol = T._hip_solver(g, reference_mesh(load_golden("mesh_small")), b).solve()
    want = g["call_dt"][-len(sol.dynamics.dt):] if len(sol.dynamics.dt) <= len(g["call_dt"]) else g["call_dt"]
    n = min(len(want), len(sol.dynamics.dt))
    d = np.abs(sol.dynamics.dt[:n] - want[:n]) / want[:n]
    cum = np.maximum.accumulate(d)
    idx = [int(n * f) - 1 for f in (0.1, 0.25, 0.5, 0.75, 1.0)]
    last = sol.tdgl_data
    print(f"{name}: steps hip {len(sol.dynamics.dt)} ref {len(want)}; cum max rel dt dev at 10/25/50/75/100%:",
          " ".join(f"{cum[i]:.1e}" for i in idx))
    if len(sol.dynamics.dt) == len(want):
        print("    final: |psi|^2 %.1e  Js %.1e  Jn %.1e  mu %.1e   mean iters %.1f" % (
            max_abs(np.abs(last.psi)**2, np.abs(g["final_psi"])**2), max_abs(last.supercurrent, g["final_supercurrent"]),
            max_abs(last.normal_current, g["final_normal_current"]), max_abs(remove_mean(last.mu), remove_mean(g["final_mu"])),
            sol.dynamics.pcg_iterations.mean()))
    else:
        print("    first differing dt index:", int(np.flatnonzero(d > 1e-6)[0]) if (d > 1e-6).any() else None)

for name in ["traj_zero_field_5k", "traj_field_small", "traj_field_small_fixed_dt", "traj_transport_strip",
             "traj_transport_ramp", "traj_retry_small", "runner_bookkeeping"]:
    run(name)
