"""The domain-decomposed path at size (BASELINE config 5: 4M-site film, 8-way decomposition).

A test box has ONE MI355X, so the ranks share it through the library's host-callback transport
(every kernel, pack/unpack, sliced hierarchy, halo plan and the step logic are those of a multi-GPU
run; only the RCCL calls are swapped for gloo).  What these tests add over test_hip_distributed.py
is SIZE: ~500k rows per rank (the shape of the 8-GPU scaling run), 16-bit column offsets falling
back to 32 bits on ghost columns, the replicated fp32 coarse chain of a 4M-site hierarchy, and --
in the 2-rank case -- partitions big enough (>= 750k ghost-free rows) for the automatic
halo-overlap mode to switch itself on.

The pieces are built once by the parent (`distributed.prepare_payloads`: the root-built set-up
path of `DistributedTDGL(root=0)`) and handed to the ranks as files.
"""

import gc
import os
import pickle
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _paths():
    for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _options():
    from tdgl_amd import SolverOptions

    return SolverOptions(solve_time=1e9, dt_init=1e-4, save_every=10**6, pcg_rtol=1e-11)


def _worker(rank, world, port, work_dir, steps):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tdgl_amd import _lib

    _lib.load()  # libtdgl_hip and its ROCm runtime before torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tdgl_amd.distributed import DistributedTDGL

        with open(os.path.join(work_dir, f"piece_{rank}.pkl"), "rb") as f:
            payload = pickle.load(f)
        run = DistributedTDGL(None, _options(), rank=rank, world=world, transport="gloo", device_id=0,
                              payload=payload, overlap="auto")
        del payload
        run.set_state(1.0, 0.0)
        run.begin_stage()
        res = run.run(steps)
        on, rows = run.ctx.comm_overlap()
        comm = run.ctx.comm_stats()
        fields = run.gather_state()
        if rank == 0:
            np.savez(os.path.join(work_dir, "dist.npz"), dt=res["dt"], iters=res["pcg_iters"], overlap=on,
                     interior_rows=rows, n_own=run.lp.n_own, n_ghost=run.lp.n_ghost, halos=comm["halos"],
                     allreduces=comm["allreduces"], **fields)
        run.close()
    finally:
        dist.destroy_process_group()


def _run_case(side, world, steps, tmp_path):
    _paths()
    from helpers import synthetic_mesh, uniform_field_A
    from tdgl_amd import TDGLSolver
    from tdgl_amd.distributed import prepare_payloads

    mesh = synthetic_mesh(side)
    A = uniform_field_A(mesh, 0.1)
    n = len(mesh.sites)
    pieces = prepare_payloads(mesh, world, A, 1.0)
    sizes = pieces[0]["coarse"]["sizes"]
    for r, pay in enumerate(pieces):
        with open(os.path.join(tmp_path, f"piece_{r}.pkl"), "wb") as f:
            pickle.dump(pay, f, protocol=4)
    own = [p["lp"].n_own for p in pieces]
    del pieces
    gc.collect()
    # single-GPU run of the same problem (its own RCM numbering and hierarchy)
    solver = TDGLSolver.from_dimensionless(mesh, _options(), A, 1.0)
    ctx = solver.ctx
    ctx.set_state(solver.psi_init, solver.mu_init)
    ctx.begin_stage()
    ref_res = ctx.run(steps)
    ref = ctx.get_state()
    ctx.close()
    del solver, ctx
    gc.collect()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), steps), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, "dist.npz"))
    return mesh, n, sizes, own, ref_res, ref, got


def _assert_same_run(ref_res, ref, got, steps):
    assert len(got["dt"]) == steps
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-9
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-9
    assert np.abs(got["normal_current"] - ref["normal_current"]).max() < 1e-9
    # a different (global-numbering) hierarchy and summation order, the same solver: similar counts
    assert abs(got["iters"][2:].mean() - ref_res["pcg_iters"][2:].mean()) < 3.0


def test_config5_four_million_sites_on_eight_ranks(tmp_path):
    """BASELINE config 5 through the decomposition: 4.0M sites cut 8 ways (~500k rows per rank, up
    to 5 neighbours), 12 steps, equal to the single-GPU run of the same film to 1e-9, and the
    gathered fields satisfy the defining equations (div(J_s + J_n) = 0 on interior sites)."""
    steps = 12
    mesh, n, sizes, own, ref_res, ref, got = _run_case(1860.0, 8, steps, tmp_path)
    assert n > 3_900_000 and sizes[0] == n and len(sizes) >= 4
    assert max(own) - min(own) <= 8 and min(own) > 480_000
    _assert_same_run(ref_res, ref, got, steps)
    assert not bool(got["overlap"])  # 500k-row partitions: below the automatic overlap threshold
    assert int(got["halos"]) > 0 and int(got["allreduces"]) > 0
    # current conservation of the gathered fields (operators.py:59-84 applied to J_s + J_n)
    em = mesh.edge_mesh
    k = got["supercurrent"] + got["normal_current"]
    w = em.dual_edge_lengths * k
    div = (np.bincount(em.edges[:, 0], weights=w, minlength=n) - np.bincount(em.edges[:, 1], weights=w, minlength=n))
    div /= mesh.areas
    scale = np.abs(got["supercurrent"]).max() + 1e-30
    assert np.abs(div).max() < 1e-6 * max(scale, 1e-3)


def test_overlap_switches_itself_on_for_large_partitions(tmp_path):
    """1.6M sites on 2 ranks: ~800k rows each with more than 750k ghost-free rows, so the automatic
    mode runs the stencil kernels split around the halo exchange on the second stream -- and the run
    still equals the single-GPU one."""
    steps = 10
    mesh, n, sizes, own, ref_res, ref, got = _run_case(1178.0, 2, steps, tmp_path)
    assert n > 1_560_000
    assert bool(got["overlap"]) and int(got["interior_rows"]) >= 750_000
    _assert_same_run(ref_res, ref, got, steps)
