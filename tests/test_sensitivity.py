"""Evidence for the trajectory tolerances used in the GPU parity tests (CPU only).

TDGL dynamics amplify perturbations (vortex nucleation is an instability; runs at the
stability edge of the time step are chaotic maps).  This test perturbs the ORACLE's initial
condition by 1e-14 relative -- one unit in the last place -- and records how far its own
trajectory moves.  The GPU tests accept the HIP path when it is as close to the reference as
the reference is to its perturbed self.
"""

import numpy as np

from conftest import load_golden
from helpers import (
    GAMMA_DEFAULT,
    U_DEFAULT,
    edge_terminal,
    max_abs,
    options_from_golden,
    reference_mesh,
    uniform_field_A,
)
from oracle import OracleSolver, run_time_loop


def _perturbed_run(name, eps, seed=0):
    g = load_golden(name)
    opts = options_from_golden(g)
    terms, cf = (), None
    if "transport" in name:
        mesh = reference_mesh(load_golden("mesh_strip"))
        terms = [edge_terminal(mesh, "source", -30.0), edge_terminal(mesh, "drain", 30.0)]
        cf = lambda t: {"source": 0.5 * min(t, 8.0), "drain": -0.5 * min(t, 8.0)}  # noqa: E731
    else:
        mesh = reference_mesh(load_golden("mesh_small"))
    s = OracleSolver(mesh, uniform_field_A(mesh, float(g["b"])), 1.0, U_DEFAULT, GAMMA_DEFAULT, opts,
                     terminals=terms, current_func=cf)
    rng = np.random.default_rng(seed)
    psi0 = s.psi_init * (1 + eps * rng.standard_normal(len(s.psi_init)))
    return g, run_time_loop(s, opts, psi=psi0)


def test_vortex_entry_amplifies_one_ulp_to_1e9():
    g, out = _perturbed_run("traj_field_small", 1e-14)
    dts = out["log"].array("dt")
    assert len(dts) == len(g["call_dt"])
    dev_dt = max_abs(dts, g["call_dt"]) / g["call_dt"].max()
    dev_psi = max_abs(np.abs(out["psi"]) ** 2, np.abs(g["final_psi"]) ** 2)
    assert 1e-11 < dev_dt < 5e-8 and 1e-11 < dev_psi < 5e-8  # measured 3e-9 / 3e-9


def test_stability_edge_run_is_chaotic():
    g, out = _perturbed_run("traj_transport_ramp", 1e-14)
    dts = out["log"].array("dt")
    assert len(dts) == len(g["call_dt"])
    rel = np.abs(dts - g["call_dt"]) / g["call_dt"]
    assert rel[:65].max() < 1e-6  # first half still tight (measured 8e-10)
    assert rel.max() > 1e-6  # by the end a 1e-14 perturbation is macroscopic (measured 6e-4)


def test_retry_decisions_flip_under_one_ulp():
    g, out = _perturbed_run("traj_retry_small", 1e-14)
    dts = out["log"].array("dt")
    n = min(len(dts), len(g["call_dt"]))
    assert np.array_equal(dts[:10], g["call_dt"][:10])
    assert not np.array_equal(dts[:n], g["call_dt"][:n])  # some later retry decision differs


def _perturbed_polygon_run(name, eps):
    g = load_golden(name)
    opts = options_from_golden(g)
    mesh = reference_mesh(load_golden("mesh_polygon"))
    terms = [edge_terminal(mesh, "source", -15.0), edge_terminal(mesh, "drain", 15.0)]
    cur = float(g["current"])
    s = OracleSolver(mesh, uniform_field_A(mesh, float(g["b"])), 1.0, U_DEFAULT, GAMMA_DEFAULT, opts,
                     terminals=terms, current_func=lambda t: {"source": cur, "drain": -cur})
    psi0 = s.psi_init * (1 + eps * np.cos(mesh.sites[:, 0]))
    return g, run_time_loop(s, opts, psi=psi0)


def test_adaptive_run_on_the_polygon_device_is_chaotic_and_its_fixed_dt_twin_is_not():
    """traj_transport_polygon (adaptive dt on a coarse device mesh): the reference's own dt sequence
    moves by O(1) under a 1e-14 perturbation, so the GPU test compares its early part and single
    steps only; with a fixed time step the same device is stable and compared as a whole."""
    g, out = _perturbed_polygon_run("traj_transport_polygon", 1e-14)
    n_sim = int((g["call_time"] == 0).nonzero()[0][-1])
    dts, want = out["log"].array("dt"), g["call_dt"][n_sim:]
    k = min(len(dts), len(want))
    rel = np.abs(dts[:k] - want[:k]) / want.max()
    assert rel.max() > 1e-2  # measured: O(1) -- already the thermalisation stage ends differently
    g, out = _perturbed_polygon_run("traj_transport_polygon_fixed_dt", 1e-14)
    assert max_abs(np.abs(out["psi"]) ** 2, np.abs(g["final_psi"]) ** 2) < 1e-9  # measured 6e-11
    assert max_abs(out["supercurrent"], g["final_supercurrent"]) < 1e-9
