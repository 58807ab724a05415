"""Dev diagnostic: tiny screening case, HIP vs oracle, step by step."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
from conftest import load_golden
from helpers import reference_mesh, uniform_field_A, options_from_golden, U_DEFAULT, GAMMA_DEFAULT
from tdgl_amd import SolverOptions, TDGLSolver

g = load_golden("traj_screening_tiny")
mesh = reference_mesh(g)
print("sites", len(mesh.sites), "edges", len(mesh.edge_mesh.edges))
o = options_from_golden(g)
for scr in (False, True):
    kw = dict(solve_time=o.solve_time, dt_init=o.dt_init, dt_max=o.dt_max, save_every=o.save_every, pcg_rtol=1e-12)
    if scr:
        kw.update(include_screening=True, screening_tolerance=float(g["opt_screening_tolerance"]),
                  max_iterations_per_step=int(g["opt_max_iterations_per_step"]))
    s = TDGLSolver.from_dimensionless(
        mesh, SolverOptions(**kw), uniform_field_A(mesh, float(g["b"])), 1.0, U_DEFAULT, GAMMA_DEFAULT,
        screening=dict(sites=mesh.sites, edge_centers=mesh.edge_mesh.centers,
                       areas=float(g["screening_scale"]) * mesh.areas) if scr else None)
    ctx = s.ctx
    ctx.set_state(s.psi_init, s.mu_init)
    if scr:
        ctx.set_induced_vector_potential(np.zeros((s.num_edges, 2)))
    ctx.begin_stage()
    try:
        for k in range(3):
            r = ctx.run(1)
            st = ctx.get_state()
            print(scr, k, r["dt"], r["pcg_iters"], r["screening_iterations"], np.abs(st["psi"]).min(),
                  np.abs(st["mu"]).max(), np.abs(ctx.induced_vector_potential()).max())
    except Exception as e:
        print("ERR", e)
        st = ctx.get_state()
        print(np.isnan(st["psi"]).sum(), np.isnan(st["mu"]).sum(), np.isnan(ctx.induced_vector_potential()).sum())
print("want", g["call_dt"][:3], g["call_screening_iterations"][:3])
