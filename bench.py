#!/usr/bin/env python
"""Benchmark of the TDGL time-stepping hot path (BASELINE.json: "TDGL time-steps/sec on 1M-site
mesh; achieved HBM GB/s vs roofline").

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...          # N > 1 without a launcher: spawns its own N ranks
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one call of the reference's ``TDGLSolver.update`` (psi update with retries, Poisson
solve for mu, supercurrent + normal current on every edge, probe read-out, adaptive-dt
controller) on a synthetic square film in a uniform field, fields resident in HBM.  Prints ONE
JSON line on rank 0.  See DESIGN.md "Measurement" for the byte accounting.

Steady state: whatever ``--warmup`` says, the run first takes ``--preroll`` untimed steps (default
200) so that the timed window sits where the adaptive time step has opened up -- the first ~100
steps of the trajectory are cheaper and not representative.

Beyond ``value`` (the headline window) a single-GPU run carries on and reports, each with its own
``parity_vs_oracle`` where the CPU leg runs: ``vortex_window`` (the same number of steps once vortices
have entered), ``sustained`` (the ``--late-steps`` = 6,000 consecutive steps that follow, timed as a
whole: what a production run sees) and ``late_window`` (t ~ 440 tau: the adaptive step at dt_max, a
psi-update retry every few steps).
"""

import argparse
import glob
import json
import os
import signal
import socket
import subprocess
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
METRIC = "tdgl_steps_per_sec"

WORKLOADS = {
    # name: (side in xi, description)           site counts follow SURVEY.md §8(d)
    "5k": (70.0, "square film 70 xi, 5,791 sites"),
    # sizes around the switch between the direct (dense) and the iterative mu solve
    "2k": (42.0, "square film 42 xi, ~2.1k sites"),
    "4k": (58.0, "square film 58 xi, ~4k sites"),
    "9k": (88.0, "square film 88 xi, ~9k sites"),
    "12k": (101.0, "square film 101 xi, ~12k sites"),
    "16k": (117.0, "square film 117 xi, ~16k sites"),
    "23k": (140.0, "square film 140 xi, ~23k sites"),
    "120k": (320.0, "square film 320 xi, ~120k sites"),
    "160k": (370.0, "square film 370 xi, ~160k sites"),
    "350k": (550.0, "square film 550 xi, ~350k sites"),
    "450k": (622.0, "square film 622 xi, ~450k sites"),
    "600k": (720.0, "square film 720 xi, ~600k sites"),
    "60k": (226.0, "square film 226 xi, 59,377 sites"),
    "250k": (465.0, "square film 465 xi, 250,510 sites"),
    "1M": (930.0, "square film 930 xi, 1,000,431 sites"),
    "4M": (1860.0, "square film 1860 xi, 3,998,502 sites"),
    # BASELINE config 4: strip with two current terminals (short edges), I = 0.2 * Ly, zero field
    "strip500k": ((1300.0, 333.0), "strip 1300 x 333 xi, 501,077 sites, two current terminals, I = 0.2 Ly"),
    "strip120k": ((640.0, 164.0), "strip 640 x 164 xi, ~121k sites, two current terminals, I = 0.2 Ly"),
    "strip250k": ((920.0, 236.0), "strip 920 x 236 xi, ~251k sites, two current terminals, I = 0.2 Ly"),
    # the same strip just above the depairing current density (2 / (3 sqrt 3) = 0.385 in these units)
    "strip500k_ps": ((1300.0, 333.0), "strip 1300 x 333 xi, 501,077 sites, two current terminals, I = 0.39 Ly (just above the depairing current)"),
    # ... and in a perpendicular field with the sub-critical current: vortices enter at the long edges and are driven across (flux flow)
    "strip500k_ff": ((1300.0, 333.0), "strip 1300 x 333 xi, 501,077 sites, two current terminals, I = 0.2 Ly, uniform field b = 0.02 (flux flow)"),
}
STRIP_CURRENT = {"strip120k": 0.2, "strip250k": 0.2, "strip500k": 0.2, "strip500k_ps": 0.39, "strip500k_ff": 0.2}
STRIP_FIELD = {"strip120k": 0.0, "strip250k": 0.0, "strip500k": 0.0, "strip500k_ps": 0.0, "strip500k_ff": 0.02}
B_FIELD = 0.1  # B / Bc2


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
    }


def pmc_traffic(workload):
    """HBM bytes per launch from the rocprofv3 PMC passes of the CURRENT kernels: the newest
    ``profiles/r*_pmc_hbm_traffic_<workload>.json`` (written by tools/rocpd_pmc.py from two
    ``--pmc`` passes of this very command; counters cannot be read inside the process)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic_{workload}.json")))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f)["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)


def traffic_of(table, prefix):
    for name, nbytes in table.items():
        if name.startswith(prefix):
            return float(nbytes)
    return None


def cpu_baseline(mesh, A, state, opt_kw, target_seconds=15.0, max_steps=40, terms=(), currents=None, keep_at=0,
                 extra_states=()):
    """Time the oracle (NumPy/SciPy port of the reference step: SuperLU + sparse matvecs) on this host,
    starting from the GPU run's post-warm-up state -- fields, loop state and the adaptive-dt
    controller's history, so that it takes the very steps the timed GPU window took.  Setup (operator
    build, LU factorisation) is excluded, like the GPU path's setup.

    Returns ``(baseline, run, extra_runs)``: the ``cpu_baseline`` object of the JSON line, the oracle's dt
    sequence plus its fields after ``keep_at`` steps (the checker's side of ``parity_vs_oracle``), and the
    same for every further recorded state in ``extra_states`` (``(state, K)`` pairs; untimed, same
    factorisation)."""
    from oracle import OracleSolver

    o = SimpleNamespace(skip_time=0.0, terminal_psi=0.0, **opt_kw)
    t0 = time.perf_counter()
    solver = OracleSolver(mesh, A, 1.0, 5.79, 10.0, o, terminals=terms,
                          current_func=None if currents is None else (lambda t: currents))
    setup_s = time.perf_counter() - t0
    def follow(state, keep_at, max_steps, target_seconds):
        solver.tentative_dt = state["tentative_dt"]
        solver.d_psi_sq_vals = [float(v) for v in state["history"]]  # solver.py:318: the list persists
        psi, mu = state["psi"].copy(), state["mu"].copy()
        t, dt, step0 = state["time"], state["dt"], int(state["step"])
        dts, kept = [], None
        n_done, n_timed, t_start = 0, 0, None
        while n_done < max_steps:
            new_dt, psi, mu, js, jn = solver.update({"step": step0 + n_done, "time": t, "dt": dt}, None, dt, psi=psi, mu=mu)
            dts.append(float(new_dt))
            n_done += 1
            if n_done == keep_at:
                kept = dict(psi=psi.copy(), mu=mu.copy(), supercurrent=js.copy(), normal_current=jn.copy())
            dt = new_dt  # runner.py:431-433
            t += dt
            if t_start is None:
                t_start = time.perf_counter()  # the first step is untimed (first-touch effects)
            else:
                n_timed += 1
                if time.perf_counter() - t_start > target_seconds and n_done >= keep_at:
                    break
        return dict(dt=np.array(dts), kept=kept, keep_at=keep_at), n_timed, time.perf_counter() - t_start

    run, n_timed, elapsed = follow(state, keep_at, max_steps, target_seconds)
    extra_runs = [follow(st, k, k, 0.0)[0] for st, k in extra_states]
    base = dict(
        value=n_timed / elapsed,
        unit="steps/s",
        cores=1,
        kind="port",
        sample=f"{n_timed} steps (after one untimed step) of the same workload from the GPU run's post-warm-up state, "
               f"same dt controller history; oracle = NumPy/SciPy restatement (scipy SuperLU solve + sparse matvecs, "
               f"single thread); setup excluded ({setup_s:.0f} s, mostly LU factorisation); host has {os.cpu_count()} logical cores",
    )
    return base, run, extra_runs


PARITY_TOL = 1e-8
PARITY_STEPS = 20
DD_PARITY_MAX_SITES = 1_200_000  # decomposed runs: the oracle's LU of the whole film has to fit the host and a minute


def parity_block(hip_dt, hip_state, oracle_run, source):
    """``parity_vs_oracle``: the HIP path's K steps against the oracle's K steps from the same state.
    Deviations are max-abs, divided by max(1, max|reference|) of the field; mu is compared with its
    mean removed (the reference's additive constant is SuperLU round-off)."""
    K, ref = oracle_run["keep_at"], oracle_run["kept"]

    def dev(a, b):
        return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))

    out = dict(
        steps=K, source=source, tolerance=PARITY_TOL,
        dt=float(np.max(np.abs(hip_dt[:K] - oracle_run["dt"][:K])) / np.max(oracle_run["dt"][:K])),
        abs_sq_psi=dev(np.abs(hip_state["psi"]) ** 2, np.abs(ref["psi"]) ** 2),
        mu_zero_mean=dev(hip_state["mu"] - hip_state["mu"].mean(), ref["mu"] - ref["mu"].mean()),
        J_s=dev(hip_state["supercurrent"], ref["supercurrent"]),
        J_n=dev(hip_state["normal_current"], ref["normal_current"]),
        scales=dict(mu=float(np.max(np.abs(ref["mu"] - ref["mu"].mean()))), J_s=float(np.max(np.abs(ref["supercurrent"]))),
                    J_n=float(np.max(np.abs(ref["normal_current"])))),
    )
    out["ok"] = bool(all(out[k] <= PARITY_TOL for k in ("dt", "abs_sq_psi", "mu_zero_mean", "J_s", "J_n")))
    return out


def build_workload(name):
    """Synthetic mesh + inputs of one workload."""
    from tdgl_amd.finite_volume import Mesh
    from tdgl_amd.meshgen import hex_jitter_points, triangulate

    side, desc = WORKLOADS[name]
    t0 = time.perf_counter()
    strip = isinstance(side, tuple)
    pts = hex_jitter_points(*side) if strip else hex_jitter_points(side, side)
    mesh = Mesh.from_triangulation(pts, triangulate(pts))
    n, m = len(mesh.sites), len(mesh.edge_mesh.edges)
    log(f"workload {name}: mesh {n} sites / {m} edges in {time.perf_counter() - t0:.1f} s")
    A = uniform_A(mesh, STRIP_FIELD[name] if strip else B_FIELD)
    terms, currents = (), None
    if strip:
        em_ = mesh.edge_mesh
        bidx = em_.boundary_edge_indices

        def terminal(tname, x0):
            pos = np.flatnonzero(np.isclose(em_.centers[bidx, 0], x0))
            return dict(name=tname, boundary_edge_indices=pos, edge_indices=bidx[pos],
                        length=em_.edge_lengths[bidx][pos].sum(),
                        site_indices=np.intersect1d(np.flatnonzero(np.isclose(mesh.sites[:, 0], x0)), mesh.boundary_indices))

        terms = [terminal("source", -side[0] / 2), terminal("drain", side[0] / 2)]
        cur = STRIP_CURRENT[name] * side[1]
        currents = {"source": cur, "drain": -cur}
        # probe points at (+-Lx/4, 0)
        probes = [mesh.closest_site((-side[0] / 4, 0.0)), mesh.closest_site((side[0] / 4, 0.0))]
    else:
        probes = None
    return SimpleNamespace(name=name, desc=desc, strip=strip, mesh=mesh, A=A, terms=terms, currents=currents, n=n, m=m,
                           probes=probes, mesh_s=time.perf_counter() - t0)


OPT_KW = dict(solve_time=1e12, dt_init=1e-4, dt_max=0.1, adaptive=True, adaptive_window=10,
              max_solve_retries=10, adaptive_time_step_multiplier=0.25, save_every=10**9)


def self_launch(args, argv):
    """``python bench.py --gpus N`` without a launcher: become the launcher.  One child per GPU with
    the torch.distributed.run environment; rank 0's stdout (the JSON line) passes through."""
    from tdgl_amd import _lib as _tdgl_lib

    have = _tdgl_lib.device_count()
    if have < args.gpus and not (args.share_devices or args.transport == "gloo"):
        print(json.dumps(dict(metric=METRIC, value=None, unit="steps/s", n_gpus=args.gpus, steps=args.steps,
                              warmup=args.warmup, higher_is_better=True,
                              error=f"needs {args.gpus} devices, found {have}")), flush=True)
        return 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, start_new_session=True))
    deadline = time.time() + (args.timeout + 60 if args.timeout > 0 else 10**9)
    rc = 0
    try:
        for pr in procs:
            rc = max(rc, abs(pr.wait(timeout=max(1.0, deadline - time.time()))))
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for pr in procs:
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)  # exactly the process groups started above
                except ProcessLookupError:
                    pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=200,
                    help="untimed steps before the warm-up, so that the timed window is the steady state whatever --warmup is")
    ap.add_argument("--workload", default=os.environ.get("TDGL_BENCH_WORKLOAD", "1M"), choices=list(WORKLOADS))
    ap.add_argument("--rtol", type=float, default=1e-10, help="PCG stopping tolerance (the product default, SolverOptions.pcg_rtol)")
    ap.add_argument("--check-every", type=int, default=0, help="0 = auto (predicted)")
    ap.add_argument("--smoother", default="chebyshev", choices=["chebyshev", "jacobi"])
    ap.add_argument("--nu", type=int, default=2, help="smoother degree on the coarse levels")
    ap.add_argument("--nu-fine", type=int, default=1, help="smoother degree on level 0")
    ap.add_argument("--extrapolate", type=int, default=3,
                    help="initial guess of the mu solve: 0 previous, 1/2 linear/quadratic extrapolation in time, "
                         "3 projection onto the last --guess-window solutions")
    ap.add_argument("--guess-window", type=int, default=0, help="window of the projection guess, 1..16 (0 = library default, 12)")
    ap.add_argument("--cheb-lo", type=float, default=0.1, help="Chebyshev smoothing interval [cheb_lo * rho, rho]")
    ap.add_argument("--precond-fp64", action="store_true",
                    help="keep the level-0 operators of the V-cycle in fp64 (default: fp32 storage inside the fp64 CG)")
    ap.add_argument("--precond-fp32", action="store_true",
                    help="fp32 storage only (default: fp32 and, for the three level-0 operator streams, binary16)")
    ap.add_argument("--cg-flexible", action="store_true",
                    help="flexible (Polak-Ribiere) beta in the CG instead of Fletcher-Reeves")
    ap.add_argument("--no-fused-restriction", action="store_true",
                    help="restrict the level-0 residual with two kernels instead of the pre-multiplied operator")
    ap.add_argument("--no-collapse", action="store_true",
                    help="run the coarse AMG levels kernel by kernel instead of through the collapsed operators")
    ap.add_argument("--tail-cycles", type=int, default=2,
                    help="V-cycles folded into the explicit operators of the tail level (1 = the plain cycle)")
    ap.add_argument("--mu-precond", choices=["auto", "factors", "vcycle"], default="auto",
                    help="0.65 - 1.3M sites: which preconditioner the CG of the mu solve uses -- auto (default): per solve the cheaper "
                         "of the AMG V-cycle and the fp32-stored nested-dissection factors by predicted cost; factors / vcycle: always that one")
    ap.add_argument("--sub-limits", default="", help=argparse.SUPPRESS)  # "SUB_MAX,SUB2_MAX,SUB2_BLOCK,SUB2_SUPER" (tuning runs)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip `parity_vs_oracle` (it needs the CPU baseline leg: the oracle's steps are the reference side)")
    ap.add_argument("--no-probes", action="store_true", help="strip workloads: run without the two probe points")
    ap.add_argument("--vortex-window", choices=["auto", "on", "off"], default="auto",
                    help="after the headline window, run on until the order parameter has collapsed somewhere "
                         "(min |psi|^2 < 0.05; vortices / phase slips) and time a second window there; auto = single GPU")
    ap.add_argument("--vortex-max-steps", type=int, default=5000, help="steps the search for that state may take")
    ap.add_argument("--vortex-settle", type=int, default=500, help="steps between reaching that state and the window")
    ap.add_argument("--late-steps", type=int, default=6000,
                    help="steps after the vortex window before a third timed window in the long-time regime (0 = none)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--transport", choices=["auto", "ipc", "rccl", "gloo"], default="auto",
                    help="decomposed runs, one GPU per rank: ipc = peer-mapped inboxes, neighbours' kernels store into each other's "
                         "memory over xGMI (csrc/ipc.inc); rccl = ncclSend/Recv + ncclAllReduce; auto (default) = ipc if its self-test "
                         "passes on every rank, else rccl if ITS self-test passes, else an error line.  gloo = host callbacks + "
                         "torch.distributed (a dry run: message counts and sizes are real, timings are not; implies --share-devices)")
    ap.add_argument("--share-devices", action="store_true",
                    help="let the ranks share the visible GPUs (rank r on device r %% count): a dry run of the decomposition on a "
                         "box with fewer GPUs than ranks -- works with --transport ipc and gloo")
    ap.add_argument("--transport-timeout", type=int, default=240,
                    help="seconds a transport candidate may take to set up, pass its self-test and take the probe steps")
    ap.add_argument("--debug-pre", default="none", help=argparse.SUPPRESS)
    ap.add_argument("--debug-fail", default="", help=argparse.SUPPRESS)  # transports made to fail (tests of the fall-back)
    ap.add_argument("--selftest", choices=["on", "off"], default="on",
                    help="decomposed runs: before anything is timed, one exchange per pattern and one sum per kind through the "
                         "transport, checked entry by entry on every rank (DistributedTDGL.selftest); the line carries the report")
    ap.add_argument("--dist-levels", type=int, choices=[1, 2], default=2,
                    help="decomposed runs: 2 (default) = levels 0 and 1 of the AMG hierarchy distributed, one vector exchange "
                         "per PCG iteration (partition.DeepPlanner); 1 = level 0 only, everything below replicated (rounds 1-4)")
    ap.add_argument("--schur", choices=["auto", "on", "off"], default="auto",
                    help="decomposed runs: rank-level nested dissection (every rank's interior factors in fp32 + the replicated "
                         "interface complement) as the CG's second preconditioner -- ONE all-reduce of |Gamma| doubles per application; "
                         "auto = from 100k sites on (tdgl_amd/schur_dd.py)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="use the domain-decomposition driver (RCCL communicator) even on one GPU")
    ap.add_argument("--config5", choices=["auto", "on", "off"], default="auto",
                    help="also run BASELINE config 5 (4M-site film, same decomposition) and report it as `config5`; "
                         "auto = when more than one GPU is used")
    ap.add_argument("--config5-timeout", type=int, default=900,
                    help="seconds config 5 may take before the headline line is printed without it")
    ap.add_argument("--trace-iterations", default=None,
                    help="write dt / PCG iterations of every step (pre-roll included) to this .json file")
    ap.add_argument("--timeout", type=int, default=1500,
                    help="hard limit in seconds for the whole run (SIGALRM ends a hung rank instead of blocking the node); 0 = none")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.timeout > 0:
        signal.alarm(args.timeout)

    # The step loop talks to the GPU 2.5 times per step; on a shared host (the build pod runs four tenants on one
    # 256-core box, each free to start 256 BLAS threads) a descheduled launch thread shows up as lost steps/s: three
    # driver-flag runs of round 5 read 615-790 steps/s next to fifteen at 1,056-1,154 on the same code.  Best effort.
    try:
        os.nice(-10)
    except OSError:
        pass
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = args.share_devices or args.transport == "gloo"
    if share:  # dry run: every rank on the devices there are
        from tdgl_amd import _lib as _probe

        local_rank %= max(1, _probe.device_count())
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # load libtdgl_hip (and with it ROCm 7.2's HIP runtime + RCCL) before torch brings its own
    from tdgl_amd import _lib as _tdgl_lib

    _tdgl_lib.load()
    use_dd = world > 1 or args.force_distributed
    dist = None
    if use_dd:
        import torch.distributed as dist_mod

        from tdgl_amd.distributed import stdout_to_stderr

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with stdout_to_stderr():  # gloo / RCCL print banners on stdout; stdout is the JSON line
            dist.init_process_group("gloo", rank=rank, world_size=world)  # bootstrap, barrier, max-reduce; the data path is RCCL

    from tdgl_amd import SolverOptions, TDGLSolver

    if args.mu_precond != "auto":
        from tdgl_amd.hipcore import TDGLContext

        TDGLContext.PD_CHOICE = dict(factors=1, vcycle=2)[args.mu_precond]
    if args.sub_limits:
        from tdgl_amd.hipcore import TDGLContext

        for name, v in zip(("SUB_MAX_SITES", "SUB2_MAX_SITES", "SUB2_BLOCK", "SUB2_SUPER", "SUB2_SPARSE_SEP_MIN_SITES", "SUB3_MIN_SITES",
                            "SUB3_BIG"),
                           args.sub_limits.split(",")):
            if v:
                setattr(TDGLContext, name, int(v))
    opts = SolverOptions(**OPT_KW, pcg_rtol=args.rtol, edge_currents_every_step=True, device_id=local_rank)
    popt = dict(rtol=args.rtol, max_iter=opts.pcg_max_iter, nu=args.nu, check_every=args.check_every,
                edge_currents_every_step=True, smoother=args.smoother, extrapolate=args.extrapolate,
                nu_fine=args.nu_fine, fused_restriction=not args.no_fused_restriction, cheb_lo=args.cheb_lo,
                precond_fp32=(False if args.precond_fp64 else 1 if args.precond_fp32 else True), collapse=not args.no_collapse, tail_cycles=args.tail_cycles,
                guess_window=args.guess_window, flexible_cg=args.cg_flexible)

    selftests, chosen_transport = {}, {}
    PROBE_STEPS = 40  # steps a transport candidate takes before the choice (they are the first steps of the pre-roll)

    def run_workload(name, want_cpu_state):
        """Set up `name`, pre-roll + warm up, time K steps.  Returns a dict of measurements (rank 0
        holds the global mesh; in decomposed runs the other ranks receive their pieces from it)."""
        t0 = time.perf_counter()
        wl = build_workload(name) if (rank == 0 or not use_dd) else None
        if wl is not None and wl.strip and use_dd:
            raise SystemExit("the strip workload is single-GPU in bench.py")
        drun, probed = None, False
        if not use_dd:
            solver = TDGLSolver.from_dimensionless(wl.mesh, opts, wl.A, 1.0, terminal_info=wl.terms, current_func=wl.currents,
                                                   probe_points=None if args.no_probes else wl.probes)
            solver.update_mu_boundary(0.0)
            ctx = solver.ctx
            n_loc, m_loc, n, m = wl.n, wl.m, wl.n, wl.m
            ctx.set_state(solver.psi_init, np.zeros(wl.n))
        else:
            # ONE simulation cut into `world` pieces (strong scaling): RCB partition, RCCL halo exchange
            # + all-reduces inside tdgl_run.  Rank 0 meshes, partitions and builds
            # the AMG hierarchy once and scatters the pieces.
            from tdgl_amd.distributed import DistributedTDGL

            # Transport.  Every candidate ("auto": the peer-mapped one, then RCCL) is set up inside a watchdog thread (a
            # hung collective blocks inside a library, where no Python handler runs: the thread is then abandoned and
            # the candidate counts as failed on every rank), must pass the self-test on every rank, and -- if more than
            # one is in the race -- takes the first PROBE_STEPS steps of the pre-roll with a clock on the second half:
            # the fastest one continues, the others are closed.  All decisions are taken on what EVERY rank reports.
            import threading

            candidates = ["ipc", "rccl"] if args.transport == "auto" else [args.transport]
            last_resort = "gloo" if args.transport == "auto" else None  # (host callbacks: slow, but a line rather than none)
            if world == 1 and args.transport == "auto":
                candidates = ["rccl"]  # (--force-distributed on one rank: the communicator path, as in rounds 1-4)
            payload, alive = None, {}
            for tr in candidates:
                box = {}
                # every candidate's host-side collectives (piece scatter, handle exchange, the sums of the rank-level
                # dissection's set-up, the meeting before a batch) run on a process group of ITS OWN, made here on the main
                # thread: a collective still pending in a watchdog thread that was abandoned can then never pair with the
                # main thread's next collective on the default group
                with stdout_to_stderr():  # (gloo announces every new group's connections on stdout; stdout is the JSON line)
                    cand_group = dist.new_group(backend="gloo") if world > 1 else None

                def attempt(tr=tr, box=box, payload=payload, cand_group=cand_group):
                    try:
                        if tr in args.debug_fail.split(","):
                            raise RuntimeError("made to fail (--debug-fail)")
                        d = DistributedTDGL(None if wl is None else wl.mesh, opts, None if wl is None else wl.A, 1.0,
                                            rank=rank, world=world, transport=tr, device_id=local_rank, root=0,
                                            deep="auto" if args.dist_levels == 2 else False, payload=payload,
                                            schur=dict(auto="auto", on=True, off=False)[args.schur], group=cand_group)
                        box["drun"] = d
                        rep = d.selftest() if (args.selftest == "on" and world > 1) else dict(ok=True, skipped=True)
                        box["rep"] = rep
                        if rep["ok"] and len(candidates) > 1:
                            d.set_state(1.0, 0.0)
                            d.ctx.set_poisson_options(**popt)
                            d.ctx.begin_stage()
                            d.ctx.run(PROBE_STEPS // 2)
                            d.ctx.synchronize()
                            t1 = time.perf_counter()
                            d.ctx.run(PROBE_STEPS - PROBE_STEPS // 2)
                            d.ctx.synchronize()
                            box["probe_steps_per_s"] = (PROBE_STEPS - PROBE_STEPS // 2) / (time.perf_counter() - t1)
                    except Exception as exc:
                        box["error"] = f"{type(exc).__name__}: {exc}"

                if payload is None:  # the first candidate receives this rank's piece over the bootstrap group: main thread
                    attempt()
                    finished = True
                else:
                    th = threading.Thread(target=attempt, daemon=True)
                    th.start()
                    th.join(args.transport_timeout)
                    finished = not th.is_alive()
                if box.get("drun") is not None and payload is None:
                    payload = box["drun"].payload  # (the next candidate reuses the piece this rank already received)
                rep = dict(box.get("rep") or dict(ok=False))
                if not finished:
                    # (fatal for this candidate on EVERY rank: its group is never used again, its context keeps spinning
                    # kernels bounded by the transport's own time-out)
                    rep.update(ok=False, error=f"did not finish within {args.transport_timeout} s (abandoned)")
                elif "error" in box:
                    rep.update(ok=False, error=box["error"])
                rep.update(transport=tr, device=local_rank, probe_steps_per_s=box.get("probe_steps_per_s"))
                reports = [rep]
                if world > 1:
                    reports = [None] * world
                    dist.all_gather_object(reports, rep)
                ok = all(x["ok"] for x in reports)
                sps = [x.get("probe_steps_per_s") for x in reports]
                selftests.setdefault(name, []).append(dict(
                    transport=tr, ok=ok, probe_steps_per_s=None if (not ok or None in sps) else round(min(sps), 1), ranks=reports))
                if ok:
                    alive[tr] = box["drun"]
                else:
                    log(f"rank {rank}: transport {tr} is out: " + json.dumps([x for x in reports if not x["ok"]])[:2000])
                    if finished and box.get("drun") is not None:
                        box["drun"].close()
            if not alive and last_resort is not None:
                d = DistributedTDGL(None if wl is None else wl.mesh, opts, None if wl is None else wl.A, 1.0, rank=rank, world=world,
                                    transport=last_resort, device_id=local_rank, root=0,
                                    deep="auto" if args.dist_levels == 2 else False, payload=payload)
                rep = d.selftest() if args.selftest == "on" else dict(ok=True, skipped=True)
                rep.update(transport=last_resort, device=local_rank, probe_steps_per_s=None)
                reports = [None] * world
                dist.all_gather_object(reports, rep)
                ok = all(x["ok"] for x in reports)
                selftests.setdefault(name, []).append(dict(transport=last_resort, ok=ok, probe_steps_per_s=None, ranks=reports))
                if ok:
                    alive[last_resort] = d
                    if len(candidates) > 1:  # (the others took the probe steps; this one starts the run itself)
                        d.set_state(1.0, 0.0)
                        d.ctx.set_poisson_options(**popt)
                        d.ctx.begin_stage()
                        d.ctx.run(PROBE_STEPS)
            if not alive:
                raise RuntimeError("no transport passed its self-test: " + json.dumps(selftests[name])[:4000])
            rates = {t["transport"]: (t["probe_steps_per_s"] or 0.0) for t in selftests[name] if t["transport"] in alive}
            best = max(alive, key=lambda tr: rates[tr])
            for tr, d in alive.items():
                if tr != best:
                    d.close()
            drun, chosen_transport[name] = alive[best], best
            probed = len(candidates) > 1
            ctx = drun.ctx
            n_loc, m_loc, n, m = drun.lp.n_own, len(drun.lp.edge_local_to_global), drun.n_global, drun.m_global
            if not probed:
                drun.set_state(1.0, 0.0)
            log(f"rank {rank}: owns {n_loc} sites, {drun.lp.n_ghost} ghosts, neighbours {drun.lp.neighbors}")
        ctx.set_poisson_options(**popt)
        h = ctx.hierarchy
        total_s = time.perf_counter() - t0
        st = dict(getattr(ctx, "setup_times", {}))
        setup = dict(mesh=round(wl.mesh_s, 2) if wl is not None else None, reorder=round(st.get("reorder", 0.0), 2),
                     amg_host=round(st.get("amg_host", 0.0), 2), upload=round(st.get("upload", 0.0), 2), total=round(total_s, 2))
        if st.get("amg_candidates"):  # hierarchies built and probed on the device; the one with the smallest contraction stayed
            setup["amg_candidates"] = st["amg_candidates"]
        sub = getattr(ctx, "substructure", None)
        if sub:  # mid-size meshes: substructured direct solve, factors formed on the device
            setup["substructure"] = dict(host=round(st.get("substructure_host", 0.0), 2), device=round(st.get("substructure_device", 0.0), 2),
                                         parts=sub["parts"], separator=sub["separator"], built_on=sub["built_on"],
                                         mb_per_solve=round(sub["bytes_per_solve"] / 1e6, 1),
                                         **({"symmetric_tiles": sub["symmetric_tiles"]} if "symmetric_tiles" in sub else {}))
            setup["mu_solver"] = "direct (substructured: dense interior blocks + dense Schur complement)"
            if sub.get("levels", 1) >= 2:
                setup["substructure"].update(levels=sub["levels"], super_blocks=sub["super_blocks"], top_separator=sub["top_separator"],
                                             **({k: sub[k] for k in ("super_super_blocks", "top_top_separator") if k in sub}))
                setup["mu_solver"] = (f"direct ({'two' if sub['levels'] == 2 else 'three'} levels of nested dissection: dense part interiors, "
                                      "dense separators per block of the level above, dense Schur complement of the top separator)")
        elif getattr(ctx, "dense_direct", False):  # small meshes: explicit pseudo-inverse, built on the device (or the host's LAPACK)
            setup["dense_inverse"] = round(st.get("dense_inverse_device", 0.0) + st.get("dense_inverse_host", 0.0), 2)
            setup["dense_inverse_on"] = "device" if "dense_inverse_device" in st else "host"
            setup["mu_solver"] = "direct (dense pseudo-inverse)"
        else:
            setup["mu_solver"] = "amg_pcg"
        if st.get("substructure_error"):  # factors that were attempted and refused: the line must say what runs instead, and why
            setup["substructure_error"] = str(st["substructure_error"])[:300]
            log(f"rank {rank}: the nested-dissection factors were refused ({setup['substructure_error']}); mu solver: {setup['mu_solver']}")
        pd = getattr(ctx, "precond_direct", None)
        if pd:  # 0.65 - 1.3M sites: the three-level factors in fp32 as a second preconditioner of the CG
            setup["precond_direct"] = dict(host=round(st.get("substructure_host", 0.0), 2), device=round(st.get("substructure_device", 0.0), 2), **pd)
        log(f"rank {rank}: {name} set-up {total_s:.1f} s {setup}; AMG levels {h.sizes}, "
            f"operator complexity {h.operator_complexity:.2f}")
        if not probed:
            ctx.begin_stage()

        def barrier():
            ctx.synchronize()
            if dist is not None:
                dist.barrier()

        trace = []
        if probed:  # (the probe steps of the transport race were the first steps of the pre-roll)
            if args.preroll > PROBE_STEPS:
                trace.append(ctx.run(args.preroll - PROBE_STEPS))
        elif args.preroll > 0:
            trace.append(ctx.run(args.preroll))
        if args.warmup > 0:
            trace.append(ctx.run(args.warmup))
        barrier()
        start_state = None
        if args.debug_pre == "get_state":
            ctx.get_state(supercurrent=False, normal_current=False)
        elif args.debug_pre == "loop_state":
            ctx.loop_state(), ctx.controller_state()
        elif args.debug_pre == "sleep":
            time.sleep(0.3)
        elif args.debug_pre == "alloc":
            _junk = [np.empty(2_000_000, dtype=np.complex128) for _ in range(2)]
            for a in _junk:
                a[:] = 1.0
        # Decomposed runs at a size the oracle can follow: the state before and after the timed steps is assembled
        # from all ranks (a collective: every rank takes part, rank 0 keeps the result) so that rank 0 can let the
        # oracle take the same steps from the same state -- `parity_vs_oracle` of the decomposed path.
        dd_parity = (use_dd and not args.no_parity and not args.no_cpu_baseline and args.steps <= PARITY_STEPS
                     and n <= DD_PARITY_MAX_SITES)
        if want_cpu_state or dd_parity:
            st = drun.gather_state() if use_dd else ctx.get_state(supercurrent=False, normal_current=False)
            ls, cs = ctx.loop_state(), ctx.controller_state()
            if rank == 0:
                start_state = dict(psi=st["psi"], mu=st["mu"], step=ls["step"], time=ls["time"], dt=ls["dt"],
                                   tentative_dt=cs["tentative_dt"], history=cs["history"])
            # Reading the state back (24 MB through pageable memory + NumPy) leaves the GPU idle for tens of
            # milliseconds, long enough for its clocks to drop: the 20 timed steps that follow then read 1.3-1.6 ms each
            # instead of 0.9 (three of three driver-flag runs of round 5 with the CPU baseline, none of three without).
            # K1 on scratch output, state untouched, brings the clocks back before the clock starts.
            ctx.time_kernel(1, 400)
        # ---- timed region: exactly K steps ---------------------------------------------------
        ctx.profile_enable(True)
        ctx.comm_stats(reset=True)
        ctx.step_stats(reset=True)
        if getattr(ctx, "precond_direct", None):
            ctx.precond_direct_stats(reset=True)
        barrier()
        t_begin = time.perf_counter()
        res = ctx.run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t_begin
        assert len(res["dt"]) == args.steps
        trace.append(res)
        launches, k1_ms = ctx.profile_read()
        axp_launches, axp_ms = ctx.profile_read_pcg() if not use_dd else (0, 0.0)
        direct_solves, direct_ms = ctx.profile_read_direct() if not use_dd else (0, 0.0)
        ctx.profile_enable(False)
        work = ctx.step_stats()
        ev_over = ctx.profile_event_overhead(50)  # (after the timed region)
        # K1 once more, after the timed region: ONE event pair around 16 back-to-back launches on the state the timed
        # steps ended on -- the pair's own ~4.5 us is amortised over 16 launches instead of sitting in every sample
        k1_burst_ms = ctx.time_kernel(1, 16) if not use_dd else None
        comm = ctx.comm_stats()
        # parity_vs_oracle, direct form: the fields the timed steps themselves ended on
        end_state = ctx.get_state() if (want_cpu_state and not args.no_parity and args.steps <= PARITY_STEPS) else None
        if dist is not None:
            import torch

            tmax = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        # Decomposed runs: a physical check of the state the timed steps ended on, assembled from ALL ranks.  The step
        # enforces div(J_s + J_n) = 0 at every site through the mu solve; on the sites next to a cut that sum runs over
        # edges whose currents were formed from EXCHANGED psi and mu -- a stale or misplaced ghost value anywhere shows
        # up as a divergence of the size of the currents, not of the solver's tolerance.
        conservation = None
        if use_dd:
            fields = drun.gather_state()
            if rank == 0 and dd_parity:
                end_state = fields
            if rank == 0:
                em = wl.mesh.edge_mesh
                flux = (fields["supercurrent"] + fields["normal_current"]) * em.dual_edge_lengths
                div = (np.bincount(em.edges[:, 0], flux, minlength=n) - np.bincount(em.edges[:, 1], flux, minlength=n)) / wl.mesh.areas
                jmax = float(max(np.abs(fields["supercurrent"]).max(), np.abs(fields["normal_current"]).max(), 1e-300))
                part = drun.payload.get("part") if isinstance(drun.payload, dict) else None
                conservation = dict(max_abs_divergence=float(np.abs(div).max()), max_abs_current=jmax,
                                    relative=float(np.abs(div).max() / jmax), worst_site=int(np.abs(div).argmax()),
                                    note="div(J_s + J_n) of the final state assembled from all ranks, against max |J|: of the "
                                         "order of pcg_rtol if every exchanged value arrived where it belongs")
        out = SimpleNamespace(
            wl=wl, name=name, ctx=ctx, drun=drun, n=n, m=m, n_loc=n_loc, m_loc=m_loc, elapsed=elapsed, res=res,
            k1=(launches, k1_ms), k1_burst_ms=k1_burst_ms, axp=(axp_launches, axp_ms), direct=(direct_solves, direct_ms), ev_over=ev_over, comm=comm, sizes=list(h.sizes), start_state=start_state,
            end_state=end_state, work=work, setup=setup, windows={}, conservation=conservation,
            stats=dict(ctx.poisson_stats(), guess=ctx.guess_stats(), batch_prediction=ctx.pcg_prediction_stats(),
                       **(dict(direct_switching=ctx.direct_switching()) if getattr(ctx, "dense_direct", False) else {}),
                       **(dict(preconditioner=ctx.precond_direct_stats()) if getattr(ctx, "precond_direct", None) else {})),
            overlap=ctx.comm_overlap() if use_dd else None,
            its_pre=float(np.concatenate([t["pcg_iters"] for t in trace[:-1]]).mean()) if len(trace) > 1 else None,
            trace=dict(dt=np.concatenate([t["dt"] for t in trace]).tolist(),
                       pcg_iters=np.concatenate([t["pcg_iters"] for t in trace]).tolist()),
        )
        return out

    main_run = run_workload(args.workload, want_cpu_state=(rank == 0 and not use_dd and not args.no_cpu_baseline))
    def vortex_window(r):
        """The regime the headline window does not see: run on until the order parameter has collapsed
        somewhere (min |psi|^2 < 0.05 outside the terminals: vortex cores / phase slips), let the
        dynamics settle for --vortex-settle steps, then time --steps steps there."""
        ctx, wl = r.ctx, r.wl
        free = np.ones(r.n, dtype=bool)
        for t in wl.terms:
            free[np.asarray(t["site_indices"])] = False
        total, found, search, dts, its = 0, None, [], [], []
        while total < args.vortex_max_steps:
            res = ctx.run(250)
            total += len(res["dt"])
            dts.append(res["dt"])
            its.append(res["pcg_iters"])
            a2 = np.abs(ctx.get_state(mu=False, supercurrent=False, normal_current=False)["psi"][free]) ** 2
            search.append(dict(steps_after_headline=total, min_abs_sq_psi=float(a2.min()), sites_below_0p1=int((a2 < 0.1).sum()),
                               dt_last=float(res["dt"][-1]), dt_min=float(res["dt"].min()), pcg_mean=float(res["pcg_iters"].mean())))
            if found is None and a2.min() < 0.05:
                found = total
            if found is not None and total >= found + args.vortex_settle:
                break
        r.trace["dt"] += np.concatenate(dts).tolist()
        r.trace["pcg_iters"] += np.concatenate(its).tolist()
        r.trace["vortex_search"] = search
        out_w = timed_window(r, free, "vortex")
        out_w.update(state_reached=found is not None, steps_before_window=args.preroll + args.warmup + args.steps + total)
        return out_w

    def timed_window(r, free, label):
        """Time --steps steps from wherever the run stands; records the window's start state (and, for short
        windows, its end state) so that the oracle can follow the same steps (`windows[label]`)."""
        ctx = r.ctx
        want_par = r.start_state is not None and not args.no_parity
        rec = dict(start=None, dt=None, end=None)
        if want_par:  # recorded start of the window: the oracle follows it from here too
            st = ctx.get_state(supercurrent=False, normal_current=False)
            ls0, cs0 = ctx.loop_state(), ctx.controller_state()
            rec["start"] = dict(psi=st["psi"], mu=st["mu"], step=ls0["step"], time=ls0["time"], dt=ls0["dt"],
                                tentative_dt=cs0["tentative_dt"], history=cs0["history"])
        ctx.step_stats(reset=True)
        if getattr(ctx, "precond_direct", None):
            ctx.precond_direct_stats(reset=True)
        ctx.synchronize()
        t_begin = time.perf_counter()
        res = ctx.run(args.steps)
        ctx.synchronize()
        elapsed = time.perf_counter() - t_begin
        work, ls = ctx.step_stats(), ctx.loop_state()
        if want_par:
            rec["dt"] = res["dt"].copy()
            rec["end"] = ctx.get_state() if args.steps <= PARITY_STEPS else None
        r.windows[label] = rec
        r.trace["dt"] += res["dt"].tolist()
        r.trace["pcg_iters"] += res["pcg_iters"].tolist()
        a2 = np.abs(ctx.get_state(mu=False, supercurrent=False, normal_current=False)["psi"][free]) ** 2
        return dict(
            value=round(args.steps / elapsed, 3), unit="steps/s", ms_per_step=round(1e3 * elapsed / args.steps, 4), steps=args.steps,
            simulated_time=round(ls["time"], 3), min_abs_sq_psi=float(a2.min()), sites_below_0p1=int((a2 < 0.1).sum()),
            pcg=dict(mean_iterations=round(float(res["pcg_iters"].mean()), 2), max_iterations=int(res["pcg_iters"].max())),
            retries=int(work["psi_retries"]), host_syncs_per_step=round(work["host_syncs"] / max(work["steps"], 1), 2),
            dt=dict(mean=float(res["dt"].mean()), min=float(res["dt"].min()), max=float(res["dt"].max())),
            guess=ctx.guess_stats(),
            **(dict(direct_switching=ctx.direct_switching()) if getattr(ctx, "dense_direct", False) else {}),
            **(dict(preconditioner=ctx.precond_direct_stats()) if getattr(ctx, "precond_direct", None) else {}),
        )

    def late_window(r):
        """... and the long-time regime: --late-steps further steps (the adaptive step reaches dt_max, the psi
        update starts to fail there and is repeated with dt / 4, the right-hand side of the mu equation
        moves more from step to step), then a third timed window."""
        ctx, wl = r.ctx, r.wl
        free = np.ones(r.n, dtype=bool)
        for t in wl.terms:
            free[np.asarray(t["site_indices"])] = False
        ctx.step_stats(reset=True)
        done, chunks, results = 0, [], []
        t_from = ctx.loop_state()["time"]
        ctx.synchronize()
        t_begin = time.perf_counter()
        while done < args.late_steps:  # (nothing but tdgl_run inside the timed stretch: this is `sustained`)
            res = ctx.run(min(2000, args.late_steps - done))
            done += len(res["dt"])
            results.append(res)
        ctx.synchronize()
        elapsed = time.perf_counter() - t_begin
        n_acc = 0
        for res in results:
            n_acc += len(res["dt"])
            r.trace["dt"] += res["dt"].tolist()
            r.trace["pcg_iters"] += res["pcg_iters"].tolist()
            chunks.append(dict(steps=n_acc, dt_last=float(res["dt"][-1]), dt_max=float(res["dt"].max()),
                               pcg_mean=round(float(res["pcg_iters"].mean()), 2)))
        work = ctx.step_stats()
        retries_before = int(work["psi_retries"])
        all_it = np.concatenate([res["pcg_iters"] for res in results])
        r.sustained = dict(
            value=round(done / elapsed, 3), unit="steps/s", ms_per_step=round(1e3 * elapsed / done, 4), steps=done,
            simulated_time=[round(t_from, 3), round(ctx.loop_state()["time"], 3)],
            pcg=dict(mean_iterations=round(float(all_it.mean()), 2), max_iterations=int(all_it.max())),
            psi_retries=retries_before, host_syncs_per_step=round(work["host_syncs"] / max(work["steps"], 1), 2),
            note="the long continuous stretch between the vortex window and the late window, timed as a whole: what a "
                 "production run sees (adaptive dt between 0.02 and dt_max, psi-update retries included)",
        )
        out_w = timed_window(r, free, "late")
        out_w.update(steps_after_vortex_window=done, psi_retries_on_the_way=retries_before, on_the_way=chunks)
        return out_w

    vortex = late = None
    want_vortex = args.vortex_window == "on" or (args.vortex_window == "auto" and not use_dd)
    if want_vortex and rank == 0 and not use_dd:
        vortex = vortex_window(main_run)
        log(f"vortex window: {vortex}")
        if args.late_steps > 0:
            late = late_window(main_run)
            log(f"late window: {late}")
    if args.trace_iterations and rank == 0:
        with open(args.trace_iterations, "w") as f:
            json.dump(dict(workload=args.workload, preroll=args.preroll, warmup=args.warmup, steps=args.steps,
                           **main_run.trace), f)

    def roofline_k1(r):
        ab = algorithmic_bytes(r.n_loc, r.m_loc)  # the kernel rank 0 launches covers its own rows
        launches, k1_ms = r.k1
        avg = k1_ms / max(launches, 1)
        achieved = ab["K1_psi_laplacian_spmv"] / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        table, src = pmc_traffic(r.name) if not use_dd else ({}, None)
        nbytes = ab["K1_psi_laplacian_spmv"]
        net = avg - r.ev_over  # the same samples without the reading of an empty event pair
        burst = getattr(r, "k1_burst_ms", None)
        return dict(
            bound="hbm",
            kernel="k_psi_laplacian<true> (SELL-64 covariant-Laplacian SpMV fused with the Poisson right-hand side)"
                   + (f"; rank 0's {r.n_loc} owned rows" if use_dd else ""),
            achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
            # `frac` is the contract's figure: raw HIP-event readings of every launch in the timed region, each with
            # the event pair's own overhead inside.  frac_net subtracts that overhead (measured after the region);
            # frac_burst is one pair around 16 back-to-back launches after the region (overhead amortised; repeated
            # launches may find part of the operator in the Infinity Cache, so it is an upper bracket).  The kernel
            # trace under profiles/ (rocprofv3 dispatch durations, no events) should fall between frac_net and frac_burst.
            frac_net=round(nbytes / (net * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if net > 0 else None,
            frac_burst=None if not burst else round(nbytes / (burst * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            avg_launch_ms_burst=None if not burst else round(burst, 5),
            traffic=traffic_of(table, "void tdgl::k_psi_laplacian<true"),
            traffic_source=None if src is None else f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes, gfx950 correction), {src}",
            algorithmic_bytes_per_launch=ab["K1_psi_laplacian_spmv"], avg_launch_ms=round(avg, 5), launches=launches,
            # `achieved` uses the raw event readings; an empty event pair reads this much by itself
            # (rocprofv3's dispatch durations in profiles/ do not contain it)
            event_pair_overhead_ms=round(r.ev_over, 5),
        )

    def line_for(r):
        steps_per_s = args.steps / r.elapsed  # one simulation, whatever the number of GPUs
        its = float(r.res["pcg_iters"].mean())
        ab_g = algorithmic_bytes(r.n, r.m)
        step_bytes = (ab_g["K1_psi_laplacian_spmv"] + ab_g["K2_psi_update"] + ab_g["K3_supercurrent"] + ab_g["K4_div_rhs"]
                      + ab_g["K6_normal_current"] + 16 * r.n + its * (ab_g["K5_pcg_spmv"] + 128 * r.n))
        d = dict(
            value=round(steps_per_s, 3), ms_per_step=round(1e3 * r.elapsed / args.steps, 4), sites=r.n, edges=r.m,
            amg_levels=r.sizes,
            pcg=dict(mean_iterations=round(its, 2), max_iterations=int(r.res["pcg_iters"].max()),
                     mean_iterations_untimed=None if r.its_pre is None else round(r.its_pre, 2),
                     dt_last=float(r.res["dt"][-1]), **r.stats),
            roofline=roofline_k1(r),
            # per-step aggregate in SURVEY.md section 8(d)'s canonical (unfused, fp64, int32) accounting: the
            # non-Poisson kernels K1-K4, K6, K7 once plus one K5 PCG iteration (SpMV + 128 n of vector
            # operations) per iteration actually taken -- what a straightforward implementation would move
            step_aggregate=dict(
                canonical_bytes_per_step=int(step_bytes), pcg_iterations=round(its, 2),
                achieved_gbs=round(step_bytes * steps_per_s / 1e9, 1),
                frac_of_hbm_peak=round(step_bytes * steps_per_s / 1e9 / HBM_PEAK_GBS, 4),
                note="canonical unfused fp64 bytes x steps/s: exceeds what is physically moved wherever kernels are "
                     "fused or operands are stored in 16/32 bits",
            ),
        )
        if use_dd:  # what rank 0 exchanged per step (every rank issues the same sequence)
            st = selftests.get(r.name, [])
            d["transport"] = dict(
                used=chosen_transport.get(r.name),
                # one exchange per pattern + one sum per kind, checked entry by entry on every rank before anything was timed
                selftest=[dict(transport=t["transport"], ok=t["ok"], probe_steps_per_s=t.get("probe_steps_per_s"),
                               checks_per_rank=len(t["ranks"][0].get("checks", [])),
                               failures=[dict(rank=x.get("rank"), error=x.get("error"),
                                              bad=[c for c in x.get("checks", []) if not c["ok"]][:3])
                                         for x in t["ranks"] if not x["ok"]][:8]) for t in st])
            d["conservation"] = r.conservation
            d["comm_per_step"] = dict(
                halo_exchanges=round(r.comm["halos"] / args.steps, 1),
                halo_bytes_sent=int(r.comm["halo_bytes"] / args.steps),
                allreduces=round(r.comm["allreduces"] / args.steps, 1),
                allreduce_bytes=int(r.comm["allreduce_bytes"] / args.steps),
                neighbours=len(r.drun.lp.neighbors), ghost_sites=int(r.drun.lp.n_ghost),
                overlap=bool(r.overlap[0]), interior_rows=int(r.overlap[1]),
            )
            dp = getattr(r.drun, "deep", None)
            d["comm_per_step"]["decomposition"] = (
                "level 0 distributed, levels >= 1 replicated (three first-layer exchanges and a level-1-sized sum per iteration)"
                if dp is None else
                "levels 0 and 1 distributed (per-rank aggregates), levels >= 2 replicated: per iteration ONE exchange of the "
                "residual on a deep ghost zone, a level-2-sized sum, the CG's dot products")
            sch = getattr(r.drun, "schur", None)
            if sch is not None:
                d["comm_per_step"]["decomposition"] += (
                    "; second preconditioner: rank-level nested dissection (each rank's interior factors in fp32, the interface "
                    "complement's pseudo-inverse on every rank) -- per application ONE all-reduce of |Gamma| doubles, chosen per solve "
                    "against the AMG cycle by predicted cost")
                d["comm_per_step"]["rank_level_dissection"] = dict(sch, **r.ctx.precond_direct_stats())
            if dp is not None:
                d["comm_per_step"].update(
                    deep_ghost_entries=int(dp.n_ext - dp.n_own), deep_neighbours=len(set(dp.neighbors) | set(dp.send_idx)),
                    level1_rows=dict(owned=int(dp.l1_own), x_formed=int(dp.l1_x), b_formed=int(dp.l1_loc)),
                    summed_per_iteration=dict(level2_values_fp32=int(dp.M.shape[0]), dot_partials_fp64=3 * 1024))
        return d

    main_line = line_for(main_run) if rank == 0 else None
    # the kernel that dominates the run time: the CG's fused direction update + A p (k_sell_axp), timed
    # in the run on every 8th launch (up to 64 samples); algorithmic bytes = K5 SpMV + the direction update's 24 n
    roofline_pcg = None
    if rank == 0 and main_run.axp[0] > 0:
        ab = algorithmic_bytes(main_run.n_loc, main_run.m_loc)
        axp_alg = ab["K5_pcg_spmv"] + 24 * main_run.n_loc
        axp_avg_ms = main_run.axp[1] / main_run.axp[0]
        table, src = pmc_traffic(main_run.name)
        roofline_pcg = dict(
            bound="hbm",
            kernel="k_sell_axp (CG direction update p = z + beta p fused with q = A p and the p.q partials; "
                   "one launch per PCG iteration, the largest single share of the run time)",
            achieved=round(axp_alg / (axp_avg_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(axp_alg / (axp_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            # the canonical bytes above are what an unfused fp64 / int32 kernel would move; this kernel's operator
            # stream is narrower (16-bit column offsets), so what it PHYSICALLY moves (PMC counters) is less:
            # frac_traffic = PMC bytes / time, frac_traffic_net the same without the event pair's own overhead
            frac_traffic=(lambda t: None if not t else round(t / (axp_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))(
                traffic_of(table, "void tdgl::k_sell_axp")),
            frac_traffic_net=(lambda t: None if (not t or axp_avg_ms <= main_run.ev_over) else round(
                t / ((axp_avg_ms - main_run.ev_over) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))(traffic_of(table, "void tdgl::k_sell_axp")),
            traffic=traffic_of(table, "void tdgl::k_sell_axp"),
            traffic_source=None if src is None else f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes, gfx950 correction), {src}",
            algorithmic_bytes_per_launch=int(axp_alg), avg_launch_ms=round(axp_avg_ms, 5), launches=main_run.axp[0],
            event_pair_overhead_ms=round(main_run.ev_over, 5),
        )
    # direct mu solves (small / mid-size workloads): the solve's launches (k_dense_sym_tiles + k_dense_sym_finish, or
    # k_sub_down + those two + k_sub_up) bracketed by one event pair per batch of the run-ahead loop; bytes = the factors
    # streamed once per solve (the symmetric packing halves the dense inverse: what is moved, not n^2 * 8)
    # (samples of the direct solve's own event pairs only: when the time loop has paused the direct solve --
    # tdgl_direct_switching -- the timed window holds none and the object is null; the CG's A p samples of such a
    # window are `roofline_pcg`)
    roofline_direct = None
    if rank == 0 and main_run.direct[0] > 0 and main_run.setup.get("mu_solver", "amg_pcg") != "amg_pcg":
        if main_run.axp[0] == 0:
            roofline_pcg = None
        sub = getattr(main_run.ctx, "substructure", None)
        nt = (main_run.n + 127) // 128
        solve_bytes = int(sub["bytes_per_solve"]) if sub else nt * (nt + 1) // 2 * 128 * 128 * 8
        avg_ms = main_run.direct[1] / main_run.direct[0]
        roofline_direct = dict(
            bound="hbm", kernel="direct mu solve: " + (" + ".join(f"{k} x {c}" if c > 1 else k for k, c in (
                ("k_sub_down_sym", sum(bool(t) for t in sub.get("symmetric_tiles", []))),
                ("k_sub_down", sub["levels"] - sum(bool(t) for t in sub.get("symmetric_tiles", [])))) if c > 0)
                                                       + f" + k_dense_sym_tiles + k_dense_sym_finish + k_sub_up x {sub['levels']}" if sub and sub.get("levels", 1) >= 2 else
                                                       "k_sub_down + k_dense_sym_tiles + k_dense_sym_finish + k_sub_up" if sub else
                                                       "k_dense_sym_tiles + k_dense_sym_finish"),
            achieved=round(solve_bytes / (avg_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(solve_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), bytes_per_solve=solve_bytes,
            avg_solve_ms=round(avg_ms, 5), samples=main_run.direct[0], event_pair_overhead_ms=round(main_run.ev_over, 5),
            note="launch sequence bracketed by one HIP event pair per batch of the run-ahead loop (the pair's own overhead included)",
        )
    # 0.4 - 1.3M sites: one application of the fp32-stored factors as the CG's preconditioner (gather, the solve's launch
    # sequence, scatter), bracketed by an event pair each (the first 64 of the timed window); bytes = what is streamed once
    roofline_precond = None
    if rank == 0 and main_run.direct[0] > 0 and main_run.setup.get("precond_direct"):
        pdi = main_run.setup["precond_direct"]
        avg_ms = main_run.direct[1] / main_run.direct[0]
        roofline_precond = dict(
            bound="hbm", kernel="one application of the nested-dissection factors (fp32 storage) as preconditioner: k_pd_gather + "
                                + " + ".join(f"{k} x {c}" if c > 1 else k for k, c in (
                                    ("k_sub_down_sym", sum(bool(t) for t in pdi.get("symmetric_tiles", []))),
                                    ("k_sub_down_lanes", 3 - sum(bool(t) for t in pdi.get("symmetric_tiles", [])))) if c > 0)
                                + " + sparse couplings x 3 + k_dense_sym_tiles + k_dense_sym_finish + k_sub_up x 3 + k_pd_scatter",
            achieved=round(pdi["bytes_per_application"] / (avg_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(pdi["bytes_per_application"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), bytes_per_application=int(pdi["bytes_per_application"]),
            avg_application_ms=round(avg_ms, 5), samples=main_run.direct[0], event_pair_overhead_ms=round(main_run.ev_over, 5),
            note="share of the step: applications per step x this time / ms_per_step; the pair's own overhead included")
    r = main_run
    desc = WORKLOADS[args.workload][1]
    strip = isinstance(WORKLOADS[args.workload][0], tuple)
    out = None if rank != 0 else dict(
        metric=METRIC,
        value=main_line["value"],
        unit="steps/s",
        n_gpus=world,
        steps=args.steps,
        warmup=args.warmup,
        ms_per_step=main_line["ms_per_step"],
        higher_is_better=True,
        scaling="strong",
        vs_baseline=None,
        dtype="f64",
        data="synthetic",
        config=dict(
            workload=f"{desc}, " + ("" if strip else f"uniform field b=B/Bc2={B_FIELD}, ") + f"adaptive dt (dt_init 1e-4, dt_max 0.1), "
                     + (f"mu solve: {r.setup['mu_solver']}" if r.setup.get("mu_solver", "amg_pcg") != "amg_pcg" else
                        f"PCG rtol {args.rtol:g} ({args.smoother} AMG smoother, degree {args.nu_fine} on level 0 / {args.nu} below"
                        + ("" if args.precond_fp64 else "; V-cycle operators stored in fp32"
                           + ("" if args.precond_fp32 else " (level 0: binary16)") + ", all arithmetic and the CG in fp64") + ")"
                        + ("" if not r.setup.get("precond_direct") else
                           "; second resident preconditioner: three levels of nested dissection with explicit factors stored in fp32, "
                           + {"auto": "the cheaper of the two per solve by predicted cost", "factors": "forced for every solve",
                              "vcycle": "never used (forced off)"}[args.mu_precond]))
                     + f", J_s/J_n formed every step; steady state: {args.preroll} pre-roll + {args.warmup} warm-up steps "
                       f"untimed, then {args.steps} timed steps at {main_line['pcg']['mean_iterations']} PCG iterations per step",
            sites=r.n, edges=r.m, amg_levels=r.sizes, preroll=args.preroll,
            parallelism="single" if world == 1 else
            f"domain decomposition (RCB, {world} ranks, ~{r.n // world} sites each), "
            + {"rccl": "RCCL halo exchange (ncclSend/Recv) + ncclAllReduce",
               "ipc": "peer-mapped transport: neighbours' kernels store into each other's hipIpc-mapped inboxes, flags polled in-kernel, one stream",
               "gloo": "host-callback transport"}[chosen_transport.get(args.workload, args.transport)]
            + ("; DRY RUN: the ranks share the visible GPUs -- counts and sizes of the exchanges are real, the timing is not" if share else ""),
        ),
        roofline=main_line["roofline"],
        roofline_pcg=roofline_pcg,
        roofline_direct=roofline_direct,
        roofline_precond=roofline_precond,
        pcg=main_line["pcg"],
        step_aggregate=main_line["step_aggregate"],
        # what a step costs the host: synchronisations and repeated psi updates in the timed window
        host=dict(syncs_per_step=round(r.work["host_syncs"] / max(r.work["steps"], 1), 2), psi_retries=int(r.work["psi_retries"]),
                  # share of the timed window the host spent blocked waiting for the GPU (the rest: enqueueing)
                  blocked_frac=round(r.work["host_wait_s"] / max(r.work["run_s"], 1e-12), 3),
                  probes=0 if (r.wl is None or r.wl.probes is None or args.no_probes) else len(r.wl.probes)),
        # seconds before the first step (not in `value`): meshing, RCM, AMG set-up on the host, uploads
        setup_s=r.setup,
    )
    if vortex is not None:
        out["vortex_window"] = vortex
    if late is not None:
        out["late_window"] = late
        out["sustained"] = getattr(main_run, "sustained", None)
    if rank == 0 and "comm_per_step" in main_line:
        out["comm_per_step"] = main_line["comm_per_step"]
        out["transport"] = main_line.get("transport")
        out["conservation"] = main_line.get("conservation")
    # BASELINE config 5 next to the headline workload (decomposed runs).  The headline measurement is
    # complete at this point: a watchdog thread prints it if the second workload does not finish in
    # time (a hung collective blocks inside the library, where no Python signal handler runs), so the
    # driver gets its line either way.
    want5 = args.config5 == "on" or (args.config5 == "auto" and world > 1)
    if want5 and args.workload != "4M":
        import threading

        done = threading.Event()

        def watchdog():
            if not done.wait(args.config5_timeout) and rank == 0:
                out["config5"] = dict(error=f"did not finish within {args.config5_timeout} s")
                print(json.dumps(out), flush=True)
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        start_state_keep, wl_keep = main_run.start_state, main_run.wl
        try:
            if main_run.drun is not None:
                main_run.drun.close()
            else:
                main_run.ctx.close()
            r5 = run_workload("4M", want_cpu_state=False)
            if rank == 0:
                out["config5"] = dict(workload=f"{WORKLOADS['4M'][1]}, same options and decomposition (BASELINE config 5)",
                                      scaling="strong", **line_for(r5))
        except Exception as exc:  # the headline line must still go out
            if rank == 0:
                out["config5"] = dict(error=f"{type(exc).__name__}: {exc}")
        done.set()
        main_run.start_state, main_run.wl = start_state_keep, wl_keep
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    parity_failed = False
    if r.start_state is not None:
        log("timing the CPU oracle (LU factorisation first; this takes a while at 1M sites)")
        wl = r.wl
        want_parity = not args.no_parity
        K = min(args.steps, PARITY_STEPS) if want_parity else 0
        labels = [lb for lb in ("vortex", "late") if want_parity and r.windows.get(lb, {}).get("start") is not None]
        extra = [(r.windows[lb]["start"], K) for lb in labels]
        # (decomposed runs: the oracle only follows the K timed steps -- `cpu_baseline` is a single-GPU figure)
        base, oracle_run, extra_runs = cpu_baseline(wl.mesh, wl.A, r.start_state, OPT_KW,
                                                    target_seconds=0.0 if use_dd else args.cpu_seconds, max_steps=K if use_dd else 40,
                                                    terms=wl.terms, currents=wl.currents, keep_at=K, extra_states=extra)
        if not use_dd:
            out["cpu_baseline"] = base
            out["speedup_vs_cpu_baseline"] = round(out["value"] / base["value"], 1)
            base["value"] = round(base["value"], 4)

        def hip_side(st0, timed_dt, end_state):
            """dt sequence and fields of the HIP path after K steps from the recorded state st0."""
            if end_state is not None:  # the timed window itself was K steps long
                return timed_dt, end_state, "the timed steps themselves"
            # the timed window is longer than the oracle can follow: its first K steps are taken again
            # from the recorded start state (fields, loop state, dt controller history)
            r.ctx.set_state(st0["psi"], st0["mu"])
            r.ctx.set_loop_state(st0["step"], st0["time"], st0["dt"])
            r.ctx.set_controller_state(st0["tentative_dt"], st0["history"])
            replay = r.ctx.run(K)
            redo = float(np.max(np.abs(replay["dt"] - timed_dt[:K])) / np.max(replay["dt"]))
            return replay["dt"], r.ctx.get_state(), (
                f"first {K} of the {args.steps} timed steps, taken again from the recorded start state "
                f"(dt sequence of the replay vs the timed run: {redo:.1e} relative)")

        if want_parity:
            hip_dt, hip_state, source = hip_side(r.start_state, r.res["dt"], r.end_state)
            if use_dd:
                source += f", fields assembled from {world} rank(s) of the decomposed run"
            out["parity_vs_oracle"] = parity_block(hip_dt, hip_state, oracle_run, source)
            parity_failed = not out["parity_vs_oracle"]["ok"]
            log(f"parity vs oracle: {out['parity_vs_oracle']}")
            # ... and inside the later windows (reported; only the headline window decides the exit status: a
            # psi update that fails by a hair on one side and passes on the other would be a property of
            # the trajectory, not of the kernels)
            for lb, xr in zip(labels, extra_runs):
                w = r.windows[lb]
                v_dt, v_state, v_source = hip_side(w["start"], w["dt"], w["end"])
                out[lb + "_window"]["parity_vs_oracle"] = parity_block(v_dt, v_state, xr, v_source)
                log(f"parity vs oracle in the {lb} window: {out[lb + '_window']['parity_vs_oracle']}")
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        log(f"PARITY FAILURE: a deviation from the oracle exceeds {PARITY_TOL:g}")
        sys.exit(3)


def _libraries_present():
    """The libraries are build products that travel with the tree; should one be missing in a single-process run,
    build it here (ranks of a multi-process run must not race for the compiler: there the tree has to be built first)."""
    lib_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "py-tdgl_amd", "tdgl_amd", "lib")
    if all(os.path.exists(os.path.join(lib_dir, f)) for f in ("libtdgl_hip.so", "libtdgl_mesh.so")):
        return
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("bench.py: libraries not built -- run `python -c 'import __graft_entry__ as g; g.build()'` first")
    import __graft_entry__ as entry

    entry.build()


if __name__ == "__main__":
    _libraries_present()
    main()
