"""GPU tests of the one-process-per-GPU path.

A test box has ONE MI355X and RCCL refuses two ranks on one device, so the multi-rank tests use
the library's host-callback transport (ghost values and all-reduces staged through pinned host
buffers and torch.distributed/gloo): every kernel, the pack/unpack, the sliced hierarchy and the
step logic are exactly those of a multi-GPU run; only the three RCCL calls are swapped.  The RCCL
transport itself is exercised with world_size 1 (communicator creation, all-reduce, the
no-neighbour halo).
"""

import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

N_STEPS = 60


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(width=60, height=15, dt_init=1e-4, dt_max=0.1, b=0.05):
    for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import edge_terminal, synthetic_mesh, uniform_field_A
    from tdgl_amd import SolverOptions

    mesh = synthetic_mesh(width, height)
    terms = [edge_terminal(mesh, "source", -width / 2), edge_terminal(mesh, "drain", width / 2)]
    A = uniform_field_A(mesh, b)
    em = mesh.edge_mesh
    mu_b = np.zeros(len(em.boundary_edge_indices))
    for t, sign in zip(terms, (1.0, -1.0)):
        mu_b[t["boundary_edge_indices"]] = sign * 6.0 / t["length"]
    opts = SolverOptions(solve_time=1e9, dt_init=dt_init, dt_max=dt_max, save_every=1000, pcg_rtol=1e-11)
    probes = [mesh.closest_site((-15, 0)), mesh.closest_site((15, 0))]
    fixed = np.concatenate([t["site_indices"] for t in terms])
    psi0 = np.ones(len(mesh.sites), dtype=complex)
    psi0[fixed] = 0
    return mesh, terms, A, mu_b, opts, probes, psi0


def _worker(rank, world, port, transport, out_dir, overlap=True, size=(60, 15), problem_kw=None, dist_kw=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mesh, terms, A, mu_b, opts, probes, psi0 = _problem(*size, **(problem_kw or {}))
    from tdgl_amd import _lib  # noqa: F401  (load libtdgl_hip and its ROCm runtime before torch)

    _lib.load()
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tdgl_amd.distributed import DistributedTDGL

        run = DistributedTDGL(mesh, opts, A, 1.0, rank=rank, world=world, terminal_info=terms, mu_boundary=mu_b,
                              probe_points=probes, transport=transport, device_id=0, overlap=overlap, **(dist_kw or {}))
        run.set_state(psi0, np.zeros(len(mesh.sites)))
        run.begin_stage()
        run.ctx.comm_stats(reset=True)
        res = run.run(N_STEPS)
        comm = run.ctx.comm_stats()
        fields = run.gather_state()
        if rank == 0:
            on, rows = run.ctx.comm_overlap()
            dp = run.deep
            np.savez(os.path.join(out_dir, f"dist_{transport}_{world}.npz"), dt=res["dt"], mu_probe=res["mu"],
                     deep=dp is not None, halos=comm["halos"], allreduce_bytes=comm["allreduce_bytes"],
                     deep_sizes=np.array([0] * 4 if dp is None else [dp.n_ext - dp.n_own, dp.l1_own, dp.l1_loc, dp.M.shape[0]]),
                     theta_probe=res["theta"], iters=res["pcg_iters"], overlap=on, interior_rows=rows,
                     n_own=run.lp.n_own, n_interior=run.lp.n_interior, gram=run.ctx.guess_gram(),
                     retries=run.ctx.step_stats()["psi_retries"], allreduces=comm["allreduces"],
                     schur=np.array([0, 0, 0, 0, 0, 0] if run.schur is None else
                                    [1, run.schur["interface"], run.schur["levels"], run.ctx.precond_direct_stats()["solves_factors"],
                                     run.ctx.precond_direct_stats()["iterations_factors"], run.ctx.precond_direct_stats()["solves_vcycle"]]),
                     **fields)
        run.close()
    finally:
        dist.destroy_process_group()


def _single_gpu_reference(size=(60, 15), problem_kw=None, want_ctx=False):
    mesh, terms, A, mu_b, opts, probes, psi0 = _problem(*size, **(problem_kw or {}))
    from tdgl_amd import TDGLSolver

    cur = {"source": 6.0, "drain": -6.0}
    solver = TDGLSolver.from_dimensionless(mesh, opts, A, 1.0, terminal_info=terms, current_func=cur, probe_points=probes)
    assert np.allclose(solver.mu_boundary, 0)  # set on the first step
    solver.update_mu_boundary(0.0)
    assert np.allclose(solver.mu_boundary, mu_b)
    ctx = solver.ctx
    ctx.set_state(psi0, np.zeros(len(mesh.sites)))
    ctx.begin_stage()
    res = ctx.run(N_STEPS)
    if want_ctx:
        return mesh, res, ctx.get_state(), ctx
    return mesh, res, ctx.get_state()


@pytest.mark.parametrize("world,transport", [(2, "gloo"), (3, "gloo"), (2, "ipc"), (3, "ipc")])
def test_multi_rank_run_matches_single_gpu(world, transport, tmp_path):
    """transport "ipc": the peer-mapped transport of csrc/ipc.inc -- the ranks' kernels store straight into each
    other's hipIpc-mapped inboxes and poll flags there, everything on one stream; here the processes share one GPU,
    on a node they sit on peers."""
    mesh, ref_res, ref = _single_gpu_reference()
    mp.spawn(_worker, args=(world, _free_port(), transport, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"dist_{transport}_{world}.npz"))
    # same algorithm, same tolerances; only the summation order of the dot products differs
    assert len(got["dt"]) == N_STEPS
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-9
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-9
    assert np.abs(got["normal_current"] - ref["normal_current"]).max() < 1e-9
    assert np.abs((got["mu_probe"][:, 0] - got["mu_probe"][:, 1]) - (ref_res["mu"][:, 0] - ref_res["mu"][:, 1])).max() < 1e-9
    # the decomposition must not change the iteration count materially
    assert abs(got["iters"].mean() - ref_res["pcg_iters"].mean()) < 2.0


@pytest.mark.parametrize("world,transport,choice", [(2, "gloo", 1), (3, "ipc", 1), (8, "ipc", 1), (3, "ipc", 0)])
def test_rank_level_dissection_as_the_preconditioner_matches_single_gpu(world, transport, choice, tmp_path, monkeypatch):
    """`tdgl_amd/schur_dd.py`, `csrc/schur.inc`: the cut between the ranks as the top level of a nested dissection --
    every rank holds the fp32-stored factors of its interior block, all hold the pseudo-inverse of the interface
    complement -- as the CG's second preconditioner in one-process-per-GPU mode.  12.7k sites on 2 / 3 / 8 ranks
    sharing one GPU, every solve through the factors (choice 1) or the cheaper preconditioner per solve (choice 0),
    against the single-GPU run: the same trajectory at 1e-9, ONE or two CG iterations per solve, and per step a handful
    of sums where the distributed AMG cycle needs two per iteration."""
    size, kw = (130, 95), dict(b=0.3)
    mesh, ref_res, ref = _single_gpu_reference(size, kw)
    if world == 3:  # (the local G blocks as symmetric tiles, which a rank's few parts here would not get by themselves)
        monkeypatch.setenv("TDGL_PD_SYM", "2")
    mp.spawn(_worker, args=(world, _free_port(), transport, str(tmp_path), True, size, kw,
                            dict(schur=True, schur_blocks=(60, 500, 3000), schur_choice=choice)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"dist_{transport}_{world}.npz"))
    on, n_gamma, levels, solves_f, iters_f, solves_v = (int(v) for v in got["schur"])
    assert on == 1 and levels >= 1 and 0.5 * np.sqrt(len(mesh.sites)) < n_gamma < 6 * world * np.sqrt(len(mesh.sites))
    assert len(got["dt"]) == N_STEPS
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-9
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-9
    assert np.abs(got["normal_current"] - ref["normal_current"]).max() < 1e-9
    if choice == 1:
        assert solves_v == 0 and solves_f == N_STEPS and iters_f <= 2 * N_STEPS and got["iters"].max() <= 2
        # per step: the Gram sum, the status MAX, per iteration the interface sum and the CG's sum, the gauge
        assert got["allreduces"] / N_STEPS < 8.0
    else:
        assert solves_f + solves_v == N_STEPS and solves_f >= 1


def test_peer_mapped_transport_with_fences_gives_the_same_bits(tmp_path, monkeypatch):
    """`TDGL_IPC_FENCES=1`: the peer-mapped transport with the documented system-scope release / acquire fences around
    every flag instead of its fence-free ordering (write-through payload stores drained before the flag, cache-bypassing
    loads behind the flag's branch) -- the same trajectory to the last bit on three ranks, which is what the fence-free form
    has to deliver; the switch exists so that a multi-GPU node can validate the ordering over xGMI the same way."""
    out = {}
    for fences in ("0", "1"):
        monkeypatch.setenv("TDGL_IPC_FENCES", fences)
        d = tmp_path / fences
        d.mkdir()
        mp.spawn(_worker, args=(3, _free_port(), "ipc", str(d)), nprocs=3, join=True)
        out[fences] = np.load(os.path.join(d, "dist_ipc_3.npz"))
    for key in ("dt", "psi", "mu", "supercurrent", "normal_current", "iters"):
        assert np.array_equal(out["0"][key], out["1"][key]), key


@pytest.mark.parametrize("transport", ["gloo", "ipc"])
def test_more_ranks_than_the_exact_gather_holds(transport, tmp_path, monkeypatch):
    """Beyond 16 ranks the guess's double-double totals are not gathered rank by rank: hi and lo parts are
    all-reduced separately (fp64 accuracy) and the pivot threshold of the small solve is raised to 1e-13.
    Forced here on 2 ranks: same trajectory, at most a few more iterations."""
    monkeypatch.setenv("TDGL_GUESS_NO_GATHER", "1")
    mesh, ref_res, ref = _single_gpu_reference()
    # (peer-mapped transport: these sums are 28k doubles long, more than one inbox slot holds -- they travel in pieces)
    mp.spawn(_worker, args=(2, _free_port(), transport, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, f"dist_{transport}_2.npz"))
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert got["iters"].mean() < ref_res["pcg_iters"].mean() + 4.0


def test_projection_gram_stays_global_through_psi_retries(tmp_path):
    """A psi update that fails abandons the mu solve after its first synchronisation and the step
    calls the solver again (solver.py:475-485).  The projection guess's dot products are gathered over
    the ranks at that synchronisation, and the Gram row of the window's newest vector arrives with them:
    it must be taken the first time (the second call for the same step computes no row).  Checked through
    the Gram matrix G_ij = y_i . y_j (y = A x) itself -- a global quantity -- of a run with forced retries
    (dt_init = dt_max = 2, b = 0.8: the reference's own retry fixture settings) on 2 ranks against
    the single-GPU run."""
    kw = dict(dt_init=2.0, dt_max=2.0, b=0.8)
    mesh, ref_res, ref, ctx = _single_gpu_reference(problem_kw=kw, want_ctx=True)
    assert ctx.step_stats()["psi_retries"] > 0
    G = ctx.guess_gram()
    mp.spawn(_worker, args=(2, _free_port(), "gloo", str(tmp_path), True, (60, 15), kw), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "dist_gloo_2.npz"))
    assert int(got["retries"]) == ctx.step_stats()["psi_retries"]
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert got["gram"].shape == G.shape and G.shape[0] >= 2
    scale = np.sqrt(np.outer(np.abs(np.diag(G)), np.abs(np.diag(G)))) + 1e-300
    filled = np.diag(G) != 0  # (every entry is filled: the newest row holds its stand-in y_j . b_new until the next solve)
    assert filled.sum() >= G.shape[0] - 1
    assert np.abs((got["gram"] - G) / scale)[np.ix_(filled, filled)].max() < 1e-6
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-8


@pytest.mark.parametrize("overlap", [True, False])
def test_halo_overlap_on_second_stream_matches_single_gpu(overlap, tmp_path):
    """16k-site film on 3 ranks: most 256-row tiles are ghost-free, so the stencil kernels run
    split (interior while the exchange is in flight on the communication stream, boundary rows
    after it); with the overlap disabled the same run uses the plain exchange-then-kernel order."""
    size = (120, 120)
    mesh, ref_res, ref = _single_gpu_reference(size)
    mp.spawn(_worker, args=(3, _free_port(), "gloo", str(tmp_path), overlap, size), nprocs=3, join=True)
    got = np.load(os.path.join(tmp_path, "dist_gloo_3.npz"))
    assert bool(got["overlap"]) == overlap
    assert int(got["interior_rows"]) >= 0.8 * int(got["n_interior"]) - 256 > 1000
    assert int(got["n_interior"]) < int(got["n_own"])
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-9
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-9
    assert np.abs(got["normal_current"] - ref["normal_current"]).max() < 1e-9


def test_peer_mapped_transport_overlaps_at_any_size(tmp_path):
    """With the peer-mapped transport an exchange is a sending and a receiving launch on the compute stream; the
    ghost-free tiles of the consumer are queued between them whatever the size of the partition (with RCCL the two
    cross-stream dependencies of an overlapped exchange cost 21 us: automatic mode waits for 750k ghost-free rows)."""
    size = (120, 120)
    mesh, ref_res, ref = _single_gpu_reference(size)
    mp.spawn(_worker, args=(3, _free_port(), "ipc", str(tmp_path), "auto", size), nprocs=3, join=True)
    got = np.load(os.path.join(tmp_path, "dist_ipc_3.npz"))
    assert bool(got["overlap"]) and not bool(got["deep"])
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-9
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-9


def _selftest_worker(rank, world, port, transport, out_dir, sabotage):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mesh, terms, A, mu_b, opts, probes, psi0 = _problem(120, 120)
    from tdgl_amd import _lib  # noqa: F401

    _lib.load()
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tdgl_amd.distributed import DistributedTDGL, prepare_payloads_for

        pay = prepare_payloads_for(mesh, world, [rank], A, 1.0, terminal_info=terms, mu_boundary=mu_b, **DEEP_KW)[0]
        if sabotage and rank == 1:  # a wrong send list on one rank: its neighbour must notice, nobody may hang
            dp = pay["deep"]
            nb = sorted(dp.send_idx)[0]
            dp.send_idx[nb] = dp.send_idx[nb][::-1].copy()
        run = DistributedTDGL(None, opts, rank=rank, world=world, transport=transport, device_id=0, payload=pay)
        rep = run.selftest()
        reports = [None] * world
        dist.all_gather_object(reports, rep)
        if rank == 0:
            import json

            with open(os.path.join(out_dir, f"selftest_{transport}_{int(sabotage)}.json"), "w") as f:
                json.dump(reports, f)
        run.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["ipc", "gloo"])
def test_transport_selftest_passes_and_names_a_broken_neighbour(transport, tmp_path):
    """`DistributedTDGL.selftest` (what `bench.py --gpus N` runs before anything is timed): one exchange per pattern
    and one sum per kind on functions of the global site id.  Intact plans: every check passes on every rank.  With
    one rank's deep send list reversed, the rank that receives from it reports which entry of which neighbour is
    wrong -- and nobody hangs."""
    import json

    for sabotage in (False, True):
        mp.spawn(_selftest_worker, args=(3, _free_port(), transport, str(tmp_path), sabotage), nprocs=3, join=True)
        with open(os.path.join(tmp_path, f"selftest_{transport}_{int(sabotage)}.json")) as f:
            reports = json.load(f)
        names = [c["name"] for c in reports[0]["checks"]]
        assert len(reports) == 3 and any("deep" in n for n in names) and any("fp32" in n for n in names)
        if not sabotage:
            assert all(r["ok"] for r in reports), reports
        else:
            bad = [r for r in reports if not r["ok"]]
            assert bad and all(r["rank"] != 1 or True for r in bad)
            failing = [c for r in bad for c in r["checks"] if not c["ok"]]
            assert failing and all("deep" in c["name"] for c in failing) and all(c["owner"] == 1 for c in failing)


def _dying_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mesh, terms, A, mu_b, opts, probes, psi0 = _problem()
    from tdgl_amd import _lib  # noqa: F401

    _lib.load()
    import time

    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tdgl_amd.distributed import DistributedTDGL

        run = DistributedTDGL(mesh, opts, A, 1.0, rank=rank, world=world, terminal_info=terms, mu_boundary=mu_b,
                              transport="ipc", device_id=0)
        run.set_state(psi0, np.zeros(len(mesh.sites)))
        run.begin_stage()
        # the LAST rank will leave; the first one is patient (60 s), the ones in between are not (8 s)
        run.ctx.comm_ipc_set_timeout(60.0 if rank == 0 and world > 2 else 8.0)
        run.run(3)  # all ranks alive: fine
        dist.barrier()
        if rank == world - 1:
            time.sleep(25.0 if world > 2 else 0.0)  # (stays connected to the bootstrap group while the others find out)
            return  # leaves the collective sequence: its partners' next wait can never be satisfied
        t0 = time.perf_counter()
        try:
            run.run(3, host_barrier=False)  # (the device side alone: the host meeting would notice a dead process first)
            outcome = "no error"
        except RuntimeError as exc:
            outcome = str(exc)
        with open(os.path.join(out_dir, f"dying_{rank}.txt"), "w") as f:
            f.write(f"{time.perf_counter() - t0:.1f}\n{outcome}\n")
    finally:
        dist.destroy_process_group()


def test_peer_mapped_transport_turns_a_dead_rank_into_an_error_not_a_hang(tmp_path):
    """Waits of the peer-mapped transport are bounded (8 s here): when a rank leaves, its neighbour's next exchange times
    out ONCE, every later wait of that context returns at once, and `run` raises at its end -- the box is never left
    with a kernel that spins for good."""
    mp.spawn(_dying_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    seconds, outcome = open(os.path.join(tmp_path, "dying_0.txt")).read().split("\n")[:2]
    assert "did not arrive within 8 s" in outcome
    assert 7.0 < float(seconds) < 20.0


def test_a_timed_out_rank_poisons_its_peers_instead_of_feeding_them_stale_ghosts(tmp_path):
    """Three ranks; rank 2 leaves.  Rank 1 waits 8 s, gives up, stops sending and stores the poison value into every
    flag it owns at its peers; rank 0 -- whose own bound is a minute -- fails WITH it, within seconds, instead of either
    waiting out its minute or (round 5) carrying on with whatever its inbox held and returning TDGL_OK."""
    mp.spawn(_dying_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    for rank, bound in ((1, "8 s"), (0, "60 s")):
        seconds, outcome = open(os.path.join(tmp_path, f"dying_{rank}.txt")).read().split("\n")[:2]
        assert f"did not arrive within {bound}" in outcome, (rank, outcome)
        assert 7.0 < float(seconds) < 22.0, (rank, seconds)


def _soak_worker(rank, world, port, transport, out_dir, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mesh, terms, A, mu_b, opts, probes, psi0 = _problem(120, 120, dt_init=1e-3, dt_max=0.5, b=0.6)
    from tdgl_amd import _lib  # noqa: F401

    _lib.load()
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tdgl_amd.distributed import DistributedTDGL

        run = DistributedTDGL(mesh, opts, A, 1.0, rank=rank, world=world, terminal_info=terms, mu_boundary=mu_b,
                              transport=transport, device_id=0, **DEEP_KW)
        run.set_state(psi0, np.zeros(len(mesh.sites)))
        run.begin_stage()
        res = run.run(steps)
        fields = run.gather_state()
        if rank == 0:
            np.savez(os.path.join(out_dir, f"soak_{transport}.npz"), dt=res["dt"], iters=res["pcg_iters"],
                     retries=run.ctx.step_stats()["psi_retries"], **fields)
        run.close()
    finally:
        dist.destroy_process_group()


def test_peer_mapped_transport_is_bit_identical_to_the_host_transport_over_a_long_run(tmp_path):
    """A race detector for the transport's ordering (payload and flags are write-through stores drained before the
    flag, no fences: csrc/ipc.inc).  On TWO ranks every sum over ranks is a + b, whatever the transport, so the
    peer-mapped run must reproduce the host-callback run BIT FOR BIT -- dt sequence, iteration counts, fields -- over
    600 steps with vortex entry and dt retries (~25,000 exchanges and sums); one stale value anywhere would show."""
    steps = 600
    for transport in ("gloo", "ipc"):
        mp.spawn(_soak_worker, args=(2, _free_port(), transport, str(tmp_path), steps), nprocs=2, join=True)
    a, b = (np.load(os.path.join(tmp_path, f"soak_{t}.npz")) for t in ("gloo", "ipc"))
    assert len(a["dt"]) == steps and int(a["retries"]) > 0 and a["iters"].sum() > 3000
    assert np.array_equal(a["dt"], b["dt"]) and np.array_equal(a["iters"], b["iters"]) and int(a["retries"]) == int(b["retries"])
    for key in ("psi", "mu", "supercurrent", "normal_current"):
        assert np.array_equal(a[key], b[key]), key
    assert (np.abs(a["psi"]) ** 2).min() < 0.5  # vortices are in


def test_eight_ranks_match_single_gpu(tmp_path):
    """The node size of the scaling runs: 8 ranks (here sharing one GPU through the callback
    transport), 2x4 RCB blocks with up to 5 neighbours per rank, corner-only contacts included."""
    size = (120, 60)
    mesh, ref_res, ref = _single_gpu_reference(size)
    mp.spawn(_worker, args=(8, _free_port(), "gloo", str(tmp_path), "auto", size), nprocs=8, join=True)
    got = np.load(os.path.join(tmp_path, "dist_gloo_8.npz"))
    assert len(got["dt"]) == N_STEPS
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-9
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-9
    assert np.abs(got["normal_current"] - ref["normal_current"]).max() < 1e-9
    assert not bool(got["overlap"])  # automatic mode: partitions this small do not overlap


DEEP_KW = dict(deep=True, max_coarse=50, plan_kw=dict(tail_rows=1200, dense_rows=400))


@pytest.mark.parametrize("world,transport", [(2, "gloo"), (3, "gloo"), (8, "gloo"), (1, "rccl"), (1, "ipc"), (2, "ipc"), (3, "ipc"), (8, "ipc")])
def test_two_distributed_levels_match_single_gpu(world, transport, tmp_path):
    """The decomposition the scaling run uses from ~30k sites per job on (forced here on a 16k-site film by making
    level 1 an intermediate level of the collapsed chain): per-rank aggregates, level 1 distributed through its
    explicit operators, ONE exchange of the residual on a deep ghost zone per PCG iteration instead of the ghosts of
    r and z, an all-reduce of level-2 size instead of level-1 size (`partition.DeepPlanner`,
    `tdgl_set_deep_halo_plan`).  Same trajectory as the single-GPU run to 1e-9, the iteration count within one, and the
    communication counted by the library itself: per step 3 exchanges (psi, the guess, mu) plus one per iteration."""
    size = (120, 120)
    mesh, ref_res, ref = _single_gpu_reference(size)
    mp.spawn(_worker, args=(world, _free_port(), transport, str(tmp_path), "auto", size, None, DEEP_KW), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"dist_{transport}_{world}.npz"))
    assert bool(got["deep"]) and len(got["dt"]) == N_STEPS
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-9 * ref_res["dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(ref["psi"]) ** 2).max() < 1e-9
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-9 * max(1.0, np.abs(ref["mu"]).max())
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-9
    assert np.abs(got["normal_current"] - ref["normal_current"]).max() < 1e-9
    assert abs(got["iters"].mean() - ref_res["pcg_iters"].mean()) < 1.5
    ghosts, l1_own, l1_loc, n_level2 = got["deep_sizes"]
    its = float(got["iters"].sum())
    if world > 1:
        assert ghosts > 0 and l1_loc > l1_own
        # (iterations queued beyond convergence freeze themselves but still exchange: the launched count is a little
        # above the counted one)
        assert its + 2 * N_STEPS <= int(got["halos"]) <= 1.15 * its + 3 * N_STEPS + int(got["retries"])
        # summed per iteration: the level-2 right-hand side in fp32 and 3 x 1024 partials; per step a few small arrays
        assert int(got["allreduce_bytes"]) <= 1.15 * its * (4 * n_level2 + 3 * 8192) + N_STEPS * 60000


def _screening_worker(rank, world, port, out_dir, transport="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from conftest import load_golden
    from helpers import options_from_golden, reference_mesh, uniform_field_A
    from tdgl_amd import SolverOptions, _lib

    _lib.load()
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tdgl_amd.distributed import DistributedTDGL

        g = load_golden("traj_screening_tiny")
        mesh = reference_mesh(g)
        o = options_from_golden(g)
        opts = SolverOptions(
            solve_time=o.solve_time, dt_init=o.dt_init, dt_max=o.dt_max, save_every=o.save_every, pcg_rtol=1e-12,
            include_screening=True, screening_tolerance=float(g["opt_screening_tolerance"]),
            max_iterations_per_step=int(g["opt_max_iterations_per_step"]),
            screening_step_size=float(g["opt_screening_step_size"]), screening_step_drag=float(g["opt_screening_step_drag"]),
        )
        run = DistributedTDGL(
            mesh, opts, uniform_field_A(mesh, float(g["b"])), 1.0, rank=rank, world=world, transport=transport, device_id=0,
            screening=dict(sites=mesh.sites, edge_centers=mesh.edge_mesh.centers,
                           areas=float(g["screening_scale"]) * mesh.areas),
        )
        run.set_state(np.ones(len(mesh.sites), dtype=complex), np.zeros(len(mesh.sites)))
        run.begin_stage()
        res = run.run(len(g["call_dt"]))
        fields = run.gather_state()
        if rank == 0:
            np.savez(os.path.join(out_dir, f"scr_{world}.npz"), dt=res["dt"], iters=res["screening_iterations"], **fields)
        run.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport", [(2, "gloo"), (3, "gloo"), (3, "ipc")])
def test_screening_on_several_ranks_matches_the_reference_fixture(world, transport, tmp_path):
    """include_screening in one-process-per-GPU mode: the site currents of all ranks are summed
    into a global array per screening iteration; against the REFERENCE's trajectory (fixture
    traj_screening_tiny): same screening iteration counts, same fields, same A_induced."""
    from conftest import load_golden

    g = load_golden("traj_screening_tiny")
    mp.spawn(_screening_worker, args=(world, _free_port(), str(tmp_path), transport), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, f"scr_{world}.npz"))
    assert np.array_equal(got["iters"], g["call_screening_iterations"])
    assert np.abs(got["dt"] - g["call_dt"]).max() <= 1e-9 * g["call_dt"].max()
    assert np.abs(np.abs(got["psi"]) ** 2 - np.abs(g["final_psi"]) ** 2).max() < 1e-7
    assert np.abs(got["supercurrent"] - g["final_supercurrent"]).max() < 1e-7
    assert np.abs(got["normal_current"] - g["final_normal_current"]).max() < 1e-7
    assert np.abs(got["induced_vector_potential"] - g["final_A_induced"]).max() < 1e-9


def test_rccl_transport_world_size_one(tmp_path):
    mesh, ref_res, ref = _single_gpu_reference()
    mp.spawn(_worker, args=(1, _free_port(), "rccl", str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(tmp_path, "dist_rccl_1.npz"))
    assert np.abs(got["dt"] - ref_res["dt"]).max() <= 1e-12 * ref_res["dt"].max()
    assert np.abs(got["mu"] - ref["mu"]).max() < 1e-10
    assert np.abs(got["supercurrent"] - ref["supercurrent"]).max() < 1e-10
