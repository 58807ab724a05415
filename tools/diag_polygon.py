"""First step at which the HIP trajectory on the polygon fixture leaves the oracle's, for several solver settings."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
import numpy as np
from conftest import load_golden
from helpers import edge_terminal, max_abs, options_from_golden, reference_mesh, remove_mean, uniform_field_A, U_DEFAULT, GAMMA_DEFAULT
from oracle import OracleSolver
from test_hip_parity import _hip_solver

g = load_golden("traj_transport_polygon")
mesh = reference_mesh(load_golden("mesh_polygon"))
terms = [edge_terminal(mesh, "source", -15.0), edge_terminal(mesh, "drain", 15.0)]
cf = {"source": float(g["current"]), "drain": -float(g["current"])}
nsteps = 160
o = options_from_golden(g)
ora = OracleSolver(mesh, uniform_field_A(mesh, float(g["b"])), 1.0, U_DEFAULT, GAMMA_DEFAULT, o, terminals=terms, current_func=lambda t: cf)
psi, mu, t, dt = ora.psi_init.copy(), ora.mu_init.copy(), 0.0, o.dt_init
ref = []
for k in range(nsteps):
    dt, psi, mu, js, jn = ora.update({"step": k, "time": t, "dt": dt}, None, dt, psi=psi, mu=mu)
    ref.append((dt, np.abs(psi) ** 2, remove_mean(mu)))
    t += dt
print("oracle dt vs golden", max_abs([r[0] for r in ref], g["call_dt"][:nsteps]))
for label, kw in (("default", {}), ("extrapolate=2", dict(extrapolate=2)), ("no collapse", dict(collapse=False)),
                  ("tail_cycles=1", dict(tail_cycles=1)), ("fp64 precond", dict(precond_fp32=False))):
    solver = _hip_solver(g, mesh, float(g["b"]), terminals=terms, current_func=cf)
    ctx = solver.ctx
    ctx.set_poisson_options(rtol=1e-11, **kw)
    ctx.set_state(solver.psi_init, solver.mu_init)
    ctx.begin_stage()
    solver.update_mu_boundary(0.0)
    first = None
    worst = 0.0
    for k in range(nsteps):
        res = ctx.run(1)
        st = ctx.get_state(supercurrent=False, normal_current=False)
        dev = max(abs(res["dt"][0] - ref[k][0]) / ref[k][0], max_abs(np.abs(st["psi"]) ** 2, ref[k][1]),
                  max_abs(remove_mean(st["mu"]), ref[k][2]))
        worst = max(worst, dev)
        if dev > 1e-7 and first is None:
            first = (k, dev, int(res["pcg_iters"][0]), ctx.guess_stats(), float(res["dt"][0]), ref[k][0])
    print(label, "levels", ctx.hierarchy.sizes, "worst", worst, "first deviation:", first)
    ctx.close()
