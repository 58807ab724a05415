"""Shared helpers for the parity tests (fixture decoding, gauge-invariant comparisons)."""

from types import SimpleNamespace

import numpy as np

from tdgl_amd.finite_volume import Mesh
from tdgl_amd.meshgen import hex_jitter_points, triangulate

U_DEFAULT, GAMMA_DEFAULT = 5.79, 10.0


def synthetic_mesh(lx, ly=None, **kw):
    pts = hex_jitter_points(lx, ly, **kw)
    return Mesh.from_triangulation(pts, triangulate(pts))


def mesh_from_golden(g, prefix="mesh_"):
    return Mesh.from_triangulation(g[prefix + "sites"], g[prefix + "elements"])


def reference_mesh(g, prefix="mesh_"):
    """Mesh object holding the reference's own arrays verbatim (nothing recomputed), so the
    oracle can be pinned without the ~1e-15 area differences of the vectorised constructor
    (which vortex dynamics amplify by ~1e6 over a few hundred steps)."""
    from tdgl_amd.finite_volume import EdgeMesh

    em = EdgeMesh(
        g[prefix + "centers"],
        g[prefix + "edges"],
        g[prefix + "boundary_edge_indices"],
        g[prefix + "directions"],
        g[prefix + "edge_lengths"],
        g[prefix + "dual_edge_lengths"],
    )
    return Mesh(
        g[prefix + "sites"],
        g[prefix + "elements"],
        g[prefix + "boundary_indices"],
        areas=g[prefix + "areas"],
        dual_sites=g[prefix + "dual_sites"],
        edge_mesh=em,
    )


def uniform_field_A(mesh, b):
    """Dimensionless symmetric-gauge vector potential on edge centres
    (tdgl/em.py:437-472 after the A_scale of tdgl/solver/solver.py:176-185)."""
    c = mesh.edge_mesh.centers
    xc = c[:, 0].min() + np.ptp(c[:, 0]) / 2
    yc = c[:, 1].min() + np.ptp(c[:, 1]) / 2
    return np.column_stack([-b * (c[:, 1] - yc) / 2, b * (c[:, 0] - xc) / 2])


def options_from_golden(g, **override):
    tp = float(g["opt_terminal_psi"])
    d = dict(
        solve_time=float(g["opt_solve_time"]),
        skip_time=float(g["opt_skip_time"]),
        dt_init=float(g["opt_dt_init"]),
        dt_max=float(g["opt_dt_max"]),
        adaptive=bool(g["opt_adaptive"]),
        adaptive_window=int(g["opt_adaptive_window"]),
        max_solve_retries=int(g["opt_max_solve_retries"]),
        adaptive_time_step_multiplier=float(g["opt_adaptive_time_step_multiplier"]),
        save_every=int(g["opt_save_every"]),
        terminal_psi=None if np.isnan(tp) else tp,
    )
    d.update(override)
    return SimpleNamespace(**d)


def edge_terminal(mesh, name, x0):
    """Terminal covering the boundary segment x = x0 (cf. Device.terminal_info,
    tdgl/device/device.py:221-256)."""
    em = mesh.edge_mesh
    sites = np.intersect1d(
        np.flatnonzero(np.isclose(mesh.sites[:, 0], x0)), mesh.boundary_indices
    )
    bidx = em.boundary_edge_indices
    pos = np.flatnonzero(np.isclose(em.centers[bidx, 0], x0))
    return dict(
        name=name,
        site_indices=sites,
        edge_indices=bidx[pos],
        boundary_edge_indices=pos,
        length=em.edge_lengths[bidx][pos].sum(),
    )


# ---- gauge-invariant comparisons (mu is defined up to a constant, which
# multiplies psi by a global phase) -------------------------------------------------------
def remove_mean(mu, areas=None):
    w = np.ones_like(mu) if areas is None else areas
    return mu - (mu * w).sum() / w.sum()


def align_phase(psi, psi_ref):
    """psi rotated by the global phase that best matches psi_ref."""
    ov = np.vdot(psi, psi_ref)
    if abs(ov) == 0:
        return psi
    return psi * (ov / abs(ov))


def max_abs(a, b):
    a, b = np.atleast_1d(a), np.atleast_1d(b)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def coo_sorted(mat):
    c = mat.tocoo()
    c.sum_duplicates()
    order = np.lexsort((c.col, c.row))
    return c.row[order], c.col[order], c.data[order]
