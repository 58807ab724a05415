#!/usr/bin/env python
"""Benchmark of the TDGL time-stepping hot path (BASELINE.json: "TDGL time-steps/sec on 1M-site
mesh; achieved HBM GB/s vs roofline").

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one call of the reference's ``TDGLSolver.update`` (psi update with retries, Poisson
solve for mu, supercurrent + normal current on every edge, probe read-out, adaptive-dt
controller) on a synthetic square film in a uniform field, fields resident in HBM.  Prints ONE
JSON line on rank 0.  See DESIGN.md "Measurement" for the byte accounting.
"""

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

# the host driver of this pool only supports dmabuf IPC: RCCL needs this before the runtime loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (side in xi, description)           site counts follow SURVEY.md §8(d)
    "5k": (70.0, "square film 70 xi, 5,791 sites"),
    "60k": (226.0, "square film 226 xi, 59,377 sites"),
    "250k": (465.0, "square film 465 xi, 250,510 sites"),
    "1M": (930.0, "square film 930 xi, 1,000,431 sites"),
    "4M": (1860.0, "square film 1860 xi, ~4.0M sites"),
    # BASELINE config 4: strip with two current terminals (short edges), I = 0.2 * Ly, zero field
    "strip500k": ((1300.0, 333.0), "strip 1300 x 333 xi, 500,955 sites, two current terminals, I = 0.2 Ly"),
}
B_FIELD = 0.1  # B / Bc2

# HBM bytes per launch of the fused psi-Laplacian kernel from rocprofv3 PMC counters
# ((2 * FETCH_SIZE + WRITE_SIZE) KiB, gfx950 correction calibrated on a copy kernel in the same
# run: profiles/r01h_pmc_hbm_traffic_1M.txt).  Counters cannot be read inside this process.
PMC_TRAFFIC_BYTES = {"1M": 175.7e6}
PMC_TRAFFIC_AXP_BYTES = {"1M": 98.0e6}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def uniform_A(mesh, b):
    c = mesh.edge_mesh.centers
    xc = c[:, 0].min() + np.ptp(c[:, 0]) / 2
    yc = c[:, 1].min() + np.ptp(c[:, 1]) / 2
    return np.column_stack([-b * (c[:, 1] - yc) / 2, b * (c[:, 0] - xc) / 2])


def algorithmic_bytes(n, m):
    """SURVEY.md §8(d) / BASELINE.md §4 (canonical CSR, int32 indices, fp64 / complex128)."""
    nnz = 2 * m + n
    return {
        "K1_psi_laplacian_spmv": 20 * nnz + 36 * n,
        "K2_psi_update": 80 * n,
        "K3_supercurrent": 40 * m + 16 * n,
        "K4_div_rhs": 32 * m + 12 * n,
        "K5_pcg_spmv": 12 * nnz + 20 * n,
        "K6_normal_current": 36 * m + 8 * n,
        "copy_c128": 32 * n,
    }


def cpu_baseline(mesh, A, state, opt_kw, target_seconds=15.0, max_steps=40, terms=(), currents=None):
    """Time the oracle (NumPy/SciPy port of the reference step: SuperLU + sparse matvecs) on
    this host, starting from the GPU run's post-warm-up state.  Setup (operator build, LU
    factorisation) is excluded, like the GPU path's setup."""
    from oracle import OracleSolver

    o = SimpleNamespace(skip_time=0.0, terminal_psi=0.0, **opt_kw)
    t0 = time.perf_counter()
    solver = OracleSolver(mesh, A, 1.0, 5.79, 10.0, o, terminals=terms,
                          current_func=None if currents is None else (lambda t: currents))
    setup_s = time.perf_counter() - t0
    solver.tentative_dt = state["tentative_dt"]
    psi, mu = state["psi"].copy(), state["mu"].copy()
    t, dt = state["time"], state["dt"]
    step0 = 10**6  # past the adaptive window, like the GPU run after warm-up
    # one untimed step (first-touch effects), then time
    dt, psi, mu, _, _ = solver.update({"step": step0, "time": t, "dt": dt}, None, dt, psi=psi, mu=mu)
    n_done, t_start = 0, time.perf_counter()
    while n_done < max_steps:
        dt, psi, mu, _, _ = solver.update({"step": step0 + 1 + n_done, "time": t, "dt": dt}, None, dt, psi=psi, mu=mu)
        t += dt
        n_done += 1
        if time.perf_counter() - t_start > target_seconds:
            break
    elapsed = time.perf_counter() - t_start
    return dict(
        value=n_done / elapsed,
        unit="steps/s",
        cores=1,
        kind="port",
        sample=f"{n_done} steps of the same workload from the GPU run's post-warm-up state; "
               f"oracle = NumPy/SciPy restatement (scipy SuperLU solve + sparse matvecs, single thread); "
               f"setup excluded ({setup_s:.0f} s, mostly LU factorisation); host has {os.cpu_count()} logical cores",
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default=os.environ.get("TDGL_BENCH_WORKLOAD", "1M"), choices=list(WORKLOADS))
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--check-every", type=int, default=0, help="0 = auto (predicted)")
    ap.add_argument("--smoother", default="chebyshev", choices=["chebyshev", "jacobi"])
    ap.add_argument("--nu", type=int, default=2, help="smoother degree on the coarse levels")
    ap.add_argument("--nu-fine", type=int, default=1, help="smoother degree on level 0")
    ap.add_argument("--extrapolate", type=int, default=2, help="initial-guess extrapolation order (0, 1, 2)")
    ap.add_argument("--cheb-lo", type=float, default=0.1, help="Chebyshev smoothing interval [cheb_lo * rho, rho]")
    ap.add_argument("--precond-fp64", action="store_true",
                    help="keep the level-0 operators of the V-cycle in fp64 (default: fp32 storage inside the fp64 CG)")
    ap.add_argument("--no-fused-restriction", action="store_true",
                    help="restrict the level-0 residual with two kernels instead of the pre-multiplied operator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--kernel-reps", type=int, default=50)
    ap.add_argument("--force-distributed", action="store_true",
                    help="use the domain-decomposition driver (RCCL communicator) even on one GPU")
    ap.add_argument("--timeout", type=int, default=1500,
                    help="hard limit in seconds for the whole run (SIGALRM ends a hung rank instead of blocking the node); 0 = none")
    args = ap.parse_args()
    if args.timeout > 0:
        import signal

        signal.alarm(args.timeout)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # load libtdgl_hip (and with it ROCm 7.2's HIP runtime + RCCL) before torch brings its own
    from tdgl_amd import _lib as _tdgl_lib

    _tdgl_lib.load()
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        from tdgl_amd.distributed import stdout_to_stderr

        dist = dist_mod
        with stdout_to_stderr():  # gloo / RCCL print banners on stdout; stdout is the JSON line
            dist.init_process_group("gloo")  # bootstrap, barrier, max-reduce; the data path is RCCL

    from tdgl_amd import SolverOptions, TDGLSolver
    from tdgl_amd.finite_volume import Mesh
    from tdgl_amd.meshgen import hex_jitter_points, triangulate

    side, desc = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    strip = isinstance(side, tuple)
    pts = hex_jitter_points(*side) if strip else hex_jitter_points(side, side)
    tri = triangulate(pts)
    mesh = Mesh.from_triangulation(pts, tri)
    n, m = len(mesh.sites), len(mesh.edge_mesh.edges)
    log(f"rank {rank}: mesh {n} sites / {m} edges in {time.perf_counter() - t0:.1f} s")
    A = uniform_A(mesh, 0.0 if strip else B_FIELD)
    terms, currents = (), None
    if strip:
        em_ = mesh.edge_mesh
        bidx = em_.boundary_edge_indices

        def terminal(name, x0):
            pos = np.flatnonzero(np.isclose(em_.centers[bidx, 0], x0))
            return dict(name=name, boundary_edge_indices=pos, edge_indices=bidx[pos],
                        length=em_.edge_lengths[bidx][pos].sum(),
                        site_indices=np.intersect1d(np.flatnonzero(np.isclose(mesh.sites[:, 0], x0)), mesh.boundary_indices))

        terms = [terminal("source", -side[0] / 2), terminal("drain", side[0] / 2)]
        currents = {"source": 0.2 * side[1], "drain": -0.2 * side[1]}
        if world > 1:
            raise SystemExit("the strip workload is single-GPU in bench.py")
    opt_kw = dict(solve_time=1e12, dt_init=1e-4, dt_max=0.1, adaptive=True, adaptive_window=10,
                  max_solve_retries=10, adaptive_time_step_multiplier=0.25, save_every=10**9)
    opts = SolverOptions(**opt_kw, pcg_rtol=args.rtol, edge_currents_every_step=True, device_id=local_rank)
    t0 = time.perf_counter()
    psi_init, mu_init = np.ones(n, dtype=np.complex128), np.zeros(n)
    use_dd = world > 1 or args.force_distributed
    if use_dd and dist is None:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        from tdgl_amd.distributed import stdout_to_stderr

        with stdout_to_stderr():
            dist_mod.init_process_group("gloo", rank=0, world_size=1)
    if not use_dd:
        solver = TDGLSolver.from_dimensionless(mesh, opts, A, 1.0, terminal_info=terms, current_func=currents)
        solver.update_mu_boundary(0.0)
        psi_init = solver.psi_init
        ctx = solver.ctx
        n_loc, m_loc = n, m
        ctx.set_state(psi_init, mu_init)
    else:
        # ONE simulation cut into `world` pieces (strong scaling of the N=1 workload): RCB
        # partition, RCCL halo exchange + all-reduces inside tdgl_run
        from tdgl_amd.distributed import DistributedTDGL

        drun = DistributedTDGL(mesh, opts, A, 1.0, rank=rank, world=world, transport="rccl", device_id=local_rank)
        ctx = drun.ctx
        n_loc, m_loc = drun.lp.n_own, len(drun.lp.edge_local_to_global)
        drun.set_state(psi_init, mu_init)
        log(f"rank {rank}: owns {n_loc} sites, {drun.lp.n_ghost} ghosts, neighbours {drun.lp.neighbors}")
    ctx.set_poisson_options(rtol=args.rtol, max_iter=opts.pcg_max_iter, nu=args.nu, check_every=args.check_every,
                            edge_currents_every_step=True, smoother=args.smoother,
                            extrapolate=args.extrapolate, nu_fine=args.nu_fine,
                            fused_restriction=not args.no_fused_restriction, cheb_lo=args.cheb_lo,
                            precond_fp32=not args.precond_fp64)
    h = ctx.hierarchy
    log(f"rank {rank}: device setup {time.perf_counter() - t0:.1f} s; AMG levels {h.sizes}, operator complexity {h.operator_complexity:.2f}")
    ctx.begin_stage()

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- warm-up (untimed) -------------------------------------------------------------
    warm = ctx.run(args.warmup) if args.warmup > 0 else None
    barrier()
    start_state = None
    if rank == 0 and not use_dd and not args.no_cpu_baseline:
        st = ctx.get_state(supercurrent=False, normal_current=False)
        ls = ctx.loop_state()
        start_state = dict(psi=st["psi"], mu=st["mu"], time=ls["time"], dt=ls["dt"], tentative_dt=ls["tentative_dt"])

    # ---- timed region: exactly K steps -----------------------------------------------------
    ctx.profile_enable(True)
    ctx.comm_stats(reset=True)
    barrier()
    t_begin = time.perf_counter()
    res = ctx.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t_begin
    assert len(res["dt"]) == args.steps
    launches, k1_ms = ctx.profile_read()
    axp_launches, axp_ms = ctx.profile_read_pcg() if not use_dd else (0, 0.0)
    ctx.profile_enable(False)
    comm = ctx.comm_stats()
    if dist is not None:
        import torch

        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    steps_per_s = args.steps / elapsed  # one simulation, whatever the number of GPUs
    ab = algorithmic_bytes(n_loc, m_loc)  # the kernel rank 0 launches covers its own rows
    k1_avg_ms = k1_ms / max(launches, 1)
    achieved = ab["K1_psi_laplacian_spmv"] / (k1_avg_ms * 1e-3) / 1e9
    roofline = dict(
        bound="hbm",
        kernel="k_psi_laplacian<true> (SELL-64 covariant-Laplacian SpMV fused with the Poisson right-hand side)",
        achieved=round(achieved, 1),
        peak=HBM_PEAK_GBS,
        unit="GB/s",
        frac=round(achieved / HBM_PEAK_GBS, 4),
        traffic=PMC_TRAFFIC_BYTES.get(args.workload) if not use_dd else None,
        traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r01h_pmc_hbm_traffic_1M.txt",
        algorithmic_bytes_per_launch=ab["K1_psi_laplacian_spmv"],
        avg_launch_ms=round(k1_avg_ms, 5),
        launches=launches,
    )
    # the kernel that dominates the run time: the CG's fused direction update + A p (k_sell_axp), timed
    # in the run on the first 256 launches; algorithmic bytes = K5 SpMV + the direction update's 24 n
    roofline_pcg = None
    if axp_launches > 0:
        axp_alg = ab["K5_pcg_spmv"] + 24 * n_loc
        axp_avg_ms = axp_ms / axp_launches
        roofline_pcg = dict(
            bound="hbm",
            kernel="k_sell_axp (CG direction update p = z + beta p fused with q = A p and the p.q partials; "
                   "one launch per PCG iteration, the largest single share of the run time)",
            achieved=round(axp_alg / (axp_avg_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(axp_alg / (axp_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            traffic=PMC_TRAFFIC_AXP_BYTES.get(args.workload),
            traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r01h_pmc_hbm_traffic_1M.txt",
            algorithmic_bytes_per_launch=int(axp_alg), avg_launch_ms=round(axp_avg_ms, 5), launches=axp_launches,
        )
    # stand-alone kernel timings (same buffers, back-to-back launches) for the other rows
    names = {0: "K1_psi_laplacian_spmv", 2: "K2_psi_update", 3: "K3_supercurrent", 4: "K5_pcg_spmv", 6: "copy_c128"}
    kernels = {}
    for kid, name in (names.items() if not use_dd else []):
        ms = ctx.time_kernel(kid, args.kernel_reps)
        nbytes = ab[name] + (ab["K6_normal_current"] if kid == 3 else 0)
        kernels[name if kid != 3 else "K3+K6_edge_currents"] = dict(
            ms=round(ms, 5), gbs=round(nbytes / (ms * 1e-3) / 1e9, 1), frac=round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        )
    vc_ms = round(ctx.time_kernel(5, 20), 4) if not use_dd else None
    out = dict(
        metric="tdgl_steps_per_sec",
        value=round(steps_per_s, 3),
        unit="steps/s",
        n_gpus=world,
        steps=args.steps,
        warmup=args.warmup,
        ms_per_step=round(1e3 * elapsed / args.steps, 4),
        higher_is_better=True,
        scaling="strong",
        vs_baseline=None,
        dtype="f64",
        data="synthetic",
        config=dict(
            workload=f"{desc}, " + ("" if strip else f"uniform field b=B/Bc2={B_FIELD}, ") + f"adaptive dt (dt_init 1e-4, dt_max 0.1), "
                     f"PCG rtol {args.rtol:g} ({args.smoother} AMG smoother, degree {args.nu_fine} on level 0 / {args.nu} below"
                     + ("" if args.precond_fp64 else "; V-cycle operators stored in fp32, all arithmetic and the CG in fp64")
                     + "), J_s/J_n formed every step",
            sites=n, edges=m, amg_levels=h.sizes,
            parallelism="single" if world == 1 else
            f"domain decomposition (RCB, {world} ranks, ~{n // world} sites each), RCCL halo exchange + all-reduce",
        ),
        roofline=roofline,
        roofline_pcg=roofline_pcg,
        pcg=dict(mean_iterations=round(float(res["pcg_iters"].mean()), 2), max_iterations=int(res["pcg_iters"].max()),
                 vcycle_ms=vc_ms, dt_last=float(res["dt"][-1]), **ctx.poisson_stats()),
        kernels=kernels,
    )
    # per-step aggregate in SURVEY.md section 8(d)'s canonical (unfused, fp64, int32) accounting: the
    # non-Poisson kernels K1-K4, K6, K7 once plus one K5 PCG iteration (SpMV + 128 n of vector
    # operations) per iteration actually taken -- what a straightforward implementation would move
    its = float(res["pcg_iters"].mean())
    ab_g = algorithmic_bytes(n, m)
    step_bytes = (ab_g["K1_psi_laplacian_spmv"] + ab_g["K2_psi_update"] + ab_g["K3_supercurrent"] + ab_g["K4_div_rhs"]
                  + ab_g["K6_normal_current"] + 16 * n + its * (ab_g["K5_pcg_spmv"] + 128 * n))
    out["step_aggregate"] = dict(
        canonical_bytes_per_step=int(step_bytes), pcg_iterations=round(its, 2),
        achieved_gbs=round(step_bytes * steps_per_s / 1e9, 1),
        frac_of_hbm_peak=round(step_bytes * steps_per_s / 1e9 / HBM_PEAK_GBS, 4),
        note="canonical unfused fp64 bytes x steps/s: exceeds what is physically moved wherever kernels are "
             "fused or operands are stored in 16/32 bits",
    )
    if use_dd:  # what rank 0 exchanged per step (every rank issues the same sequence)
        out["comm_per_step"] = dict(
            halo_exchanges=round(comm["halos"] / args.steps, 1),
            halo_bytes_sent=int(comm["halo_bytes"] / args.steps),
            allreduces=round(comm["allreduces"] / args.steps, 1),
            allreduce_bytes=int(comm["allreduce_bytes"] / args.steps),
            neighbours=len(drun.lp.neighbors), ghost_sites=int(drun.lp.n_ghost),
            overlap=bool(ctx.comm_overlap()[0]), interior_rows=int(ctx.comm_overlap()[1]),
        )
    if start_state is not None:
        log("timing the CPU oracle (LU factorisation first; this takes a while at 1M sites)")
        out["cpu_baseline"] = cpu_baseline(mesh, A, start_state, opt_kw, target_seconds=args.cpu_seconds,
                                           terms=terms, currents=currents)
        out["speedup_vs_cpu_baseline"] = round(steps_per_s / out["cpu_baseline"]["value"], 1)
        out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 4)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
