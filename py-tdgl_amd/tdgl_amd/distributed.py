"""One-process-per-GPU driver: domain decomposition of one simulation over the GPUs of a node.

Usage (every rank of a `torch.distributed` job runs the same code):

    import torch.distributed as dist
    dist.init_process_group("gloo")          # bootstrap / gather only; the data path is RCCL
    run = DistributedTDGL(mesh, options, A, rank=dist.get_rank(), world=dist.get_world_size())
    run.set_state(psi0, mu0); run.begin_stage()
    out = run.run(100)                       # identical dt sequence on every rank
    fields = run.gather_state()              # global psi, mu, J_s, J_n on every rank

The mesh is cut by recursive coordinate bisection (`partition.rcb_partition`).  With ``root=0`` only
rank 0 holds the global mesh: it builds the partition and the AMG hierarchy once and scatters each
rank's piece (`prepare_payloads`); without it every rank cuts its own piece from the global mesh.  The exchange of ghost values and the
all-reduces run inside `tdgl_run` on the context's stream: transport "rccl" (ncclSend/Recv, ncclAllReduce), or
transport "ipc" -- neighbours' kernels store straight into each other's device memory (hipIpc-mapped inboxes, flags
polled by the receiving kernel: csrc/ipc.inc; the ranks of one node, also when they share a GPU).  Transport
"gloo" routes them through host callbacks and torch.distributed instead -- slow, used by the test
suite so that several ranks can share one GPU.
"""

import contextlib
import ctypes
import os
import sys

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL (must precede the runtime)

from .amg import build_hierarchy
from .hipcore import TDGLContext, poisson_matrix
from .partition import (DeepPlanner, build_local_problem, deep_plan_applicable, link_deep_plans, local_hierarchy_level0,
                        rcb_partition)


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner through C stdio on stdout when a communicator is created;
    programs whose stdout is machine-read (bench.py prints one JSON line) route it to stderr."""
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def limit_host_threads(world: int) -> int:
    """One process per GPU on ONE host: without this every rank's NumPy/SciPy (BLAS, OpenMP) and
    torch open a thread per core, `world` times over -- measured on a 256-core box with 8 ranks
    sharing it: 322 s of set-up and 25x slower host-side exchanges, against 1 s with 3 ranks.  Gives
    each rank its share of the cores.  Returns the limit."""
    n = max(1, (os.cpu_count() or 1) // max(1, int(world)))
    try:
        import threadpoolctl

        threadpoolctl.threadpool_limits(limits=n)
    except Exception:  # pragma: no cover - optional dependency
        pass
    try:
        import torch

        torch.set_num_threads(n)
    except Exception:  # pragma: no cover
        pass
    return n


def prepare_payloads(mesh, world, link_exponents, epsilon=1.0, **kw):
    """`prepare_payloads_for` all ranks."""
    return prepare_payloads_for(mesh, world, range(int(world)), link_exponents, epsilon, **kw)


def prepare_payloads_for(mesh, world, ranks, link_exponents, epsilon=1.0, *, terminal_info=(), mu_boundary=None,
                         probe_points=None, screening=None, max_coarse=None, hierarchy=None, deep="auto", plan_kw=None,
                         schur=True):
    """Everything the ranks of a `world`-way run need, computed ONCE (by the root rank or ahead of
    time): the partition, each rank's sub-mesh + halo plan, its slice of AMG level 0 and of the
    inputs, and the coarse levels (replicated, one shared object).  Returns a list of `world`
    dicts; `DistributedTDGL(payload=...)` consumes one.  The counterpart of the reference's
    single `MeshOperators.build_operators()` (operators.py:282-308) for a decomposed mesh.

    ``deep``: ``"auto"`` (default) distributes TWO levels of the hierarchy with one vector exchange per PCG iteration
    (`partition.DeepPlanner`) whenever level 1 is an intermediate level of the collapsed chain (meshes from ~30k
    sites on); ``False`` keeps everything below level 0 replicated (small meshes fall back to that by themselves);
    ``True`` insists.  ``plan_kw``: arguments of `amg.collapsed_operators` (tests: ``tail_rows`` / ``dense_rows`` small
    enough to make level 1 an intermediate level of a small hierarchy)."""
    em = mesh.edge_mesh
    n, m = len(mesh.sites), len(em.edges)
    fixed = (
        np.concatenate([np.asarray(t["site_indices"] if isinstance(t, dict) else t.site_indices)
                        for t in terminal_info]).astype(np.int64)
        if len(terminal_info) else np.array([], dtype=np.int64)
    )
    part = rcb_partition(mesh.sites, int(world))
    if hierarchy is None:
        A_glob = poisson_matrix(em.edges, em.dual_edge_lengths / em.edge_lengths, n)
        # (level 0 is the distributed one, so the hierarchy needs at least one coarser level; the aggregates of
        # level 0 stay inside ranks, so that level 1 can be distributed as well)
        hierarchy = build_hierarchy(A_glob, max_coarse=max_coarse or min(600, max(8, n // 4)),
                                    part=part if deep else None)
    coarse = dict(levels=list(hierarchy.levels[1:]), coarse_pinv=hierarchy.coarse_pinv,
                  sizes=hierarchy.sizes, operator_complexity=hierarchy.operator_complexity)
    # the collapsed coarse chain involves only the replicated levels: built once here for the default
    # smoother settings, not once per rank (`TDGLContext._refresh_collapsed` falls back to building
    # it when the options differ)
    from .amg import collapsed_operators

    coarse["plan_key"] = (2, 1, 0.1, True, 2)  # nu, smoother (chebyshev), cheb_lo, collapse, tail_cycles
    coarse["plan"] = collapsed_operators(hierarchy, 2, "chebyshev", 0.1, tail_cycles=2, **(plan_kw or {}))
    use_deep = bool(deep) and deep_plan_applicable(hierarchy, coarse["plan"])
    if deep is True and not use_deep:
        raise ValueError("deep=True: level 1 of this hierarchy is not an intermediate level of the collapsed chain "
                         f"(sizes {hierarchy.sizes})")
    planner = None
    if use_deep:
        planner = DeepPlanner(hierarchy, coarse["plan"], part)
        # levels >= 2 are the replicated ones now; level 1 travels as per-rank slices
        coarse = dict(coarse, levels=list(hierarchy.levels[2:]), rho1=float(hierarchy.levels[1].rho))
    A_e = np.asarray(link_exponents, dtype=float)
    eps = np.asarray(epsilon, dtype=float) * np.ones(n)
    mu_b = np.zeros(len(em.boundary_edge_indices)) if mu_boundary is None else np.asarray(mu_boundary, dtype=float)
    probes = None if probe_points is None else np.asarray(probe_points, dtype=np.int64)
    # rank-level nested dissection (schur_dd.py): the interface between the ranks and its numbering, once for the job
    is_gamma = gamma_gid = None
    n_gamma = 0
    if schur and int(world) > 1:
        from .schur_dd import gamma_numbering, interface_cover

        is_gamma = interface_cover(em.edges, part)
        gamma_gid, n_gamma = gamma_numbering(is_gamma)
    out = []
    deep_plans = {}
    if use_deep:  # send lists come from the OTHER ranks' receive lists: all pieces are cut together
        all_lps = {r: build_local_problem(mesh, part, r, fixed_sites=fixed) for r in range(int(world))}
        deep_plans = {r: planner.cut(all_lps[r]) for r in range(int(world))}
        link_deep_plans(deep_plans, all_lps)
    for r in ranks:
        lp = all_lps[r] if use_deep else build_local_problem(mesh, part, r, fixed_sites=fixed)
        l2g = lp.local_to_global
        pay = dict(
            rank=r, world=int(world), n_global=n, m_global=m, lp=lp,
            level0=None if use_deep else local_hierarchy_level0(hierarchy, lp), deep=deep_plans.get(r),
            coarse=coarse, link_exponents=A_e[lp.edge_local_to_global], epsilon=eps[l2g],
            mu_boundary=mu_b[lp.boundary_positions], n_probes=0 if probes is None else len(probes),
        )
        if is_gamma is not None and n_gamma >= 2:
            pay["schur"] = dict(is_gamma=is_gamma[l2g], gid=gamma_gid[l2g], n_gamma=int(n_gamma))
        if probes is not None:
            g2l = np.full(n, -1, dtype=np.int64)
            g2l[l2g[: lp.n_own]] = np.arange(lp.n_own)
            loc = g2l[probes]
            pay["probe_mine"] = np.flatnonzero(loc >= 0)
            pay["probe_local"] = loc[pay["probe_mine"]]
        if screening is not None:
            pay["screening"] = dict(sites=np.asarray(screening["sites"]), areas=np.asarray(screening["areas"]),
                                    edge_centers=np.asarray(screening["edge_centers"])[lp.edge_local_to_global])
        out.append(pay)
    return out


def _own_payload(mesh, world, rank, link_exponents, epsilon, terminal_info, mu_boundary, probe_points, screening,
                 max_coarse, deep="auto", plan_kw=None):
    """Legacy construction: every rank holds the global mesh and prepares only its own piece (the
    global hierarchy is still built on every rank -- fine for small problems and tests)."""
    pieces = prepare_payloads_for(mesh, world, [rank], link_exponents, epsilon, terminal_info=terminal_info,
                                  mu_boundary=mu_boundary, probe_points=probe_points, screening=screening,
                                  max_coarse=max_coarse, deep=deep, plan_kw=plan_kw)
    return pieces[0]


class DistributedTDGL:
    """One rank of a domain-decomposed run.

    Three ways to construct (every rank of the job calls the constructor):

    * ``root=None`` (default): every rank holds the global ``mesh`` and cuts its own piece
      (small problems, tests);
    * ``root=k``: only rank k needs ``mesh`` / ``link_exponents`` / ... (the others pass ``None``);
      it runs `prepare_payloads` once and scatters the pieces over the bootstrap process group, so
      the global mesh and the AMG set-up exist once per job, not once per rank;
    * ``payload=...``: a piece prepared ahead of time (`prepare_payloads`)."""

    def __init__(self, mesh, options, link_exponents=None, epsilon=1.0, u=5.79, gamma=10.0, *, rank, world,
                 terminal_info=(), mu_boundary=None, probe_points=None, transport="rccl", device_id=None,
                 overlap="auto", screening=None, root=None, payload=None, max_coarse=None, deep="auto", plan_kw=None,
                 schur="auto", schur_blocks=None, schur_choice=None, group=None):
        """``group``: the process group every host-side collective of this object runs on (default: the default group).  A
        launcher that sets candidates up inside watchdog threads it may abandon (bench.py's transport race) gives each
        candidate its OWN group (`dist.new_group()` on the main thread), so that a collective still pending in an abandoned
        thread can never pair with the main thread's next collective."""
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank, self.world = int(rank), int(world)
        if self.world > 1:
            limit_host_threads(self.world)
        self.mesh = mesh
        self.options = options
        options.validate()
        if payload is None:
            if root is None:
                payload = _own_payload(mesh, self.world, self.rank, link_exponents, epsilon, terminal_info,
                                       mu_boundary, probe_points, screening, max_coarse, deep, plan_kw)
            else:
                pieces = None
                if self.rank == int(root):
                    pieces = prepare_payloads(mesh, self.world, link_exponents, epsilon, terminal_info=terminal_info,
                                              mu_boundary=mu_boundary, probe_points=probe_points,
                                              screening=screening, max_coarse=max_coarse, deep=deep, plan_kw=plan_kw)
                if self.world > 1:
                    got = [None]
                    dist.scatter_object_list(got, pieces, src=int(root), group=group)
                    payload = got[0]
                else:
                    payload = pieces[0]
        self.payload = payload
        self.transport = transport
        self.lp = lp = payload["lp"]
        self.n_global, self.m_global = payload["n_global"], payload["m_global"]
        dev = self.rank if device_id is None else device_id
        self.ctx = ctx = TDGLContext(
            lp.mesh, fixed_sites=lp.fixed_sites, fix_psi=(options.terminal_psi is not None), u=u, gamma=gamma,
            device_id=dev, n_owned=lp.n_own,
        )
        ctx.set_halo_plan(lp)
        self.deep = payload.get("deep")
        if self.deep is not None:
            ctx.set_deep_halo_plan(self.deep)
        ctx.set_comm_overlap(overlap)  # halo exchanges hidden behind the ghost-free rows
        if self.world > 1 or transport in ("rccl", "ipc"):  # (one rank: the decomposed sequence with world = 1)
            if transport == "rccl":
                ident = [ctx.comm_unique_id() if self.rank == 0 else None]
                dist.broadcast_object_list(ident, src=0, group=group)
                with stdout_to_stderr():
                    ctx.comm_init_rccl(ident[0])
            elif transport == "ipc":
                # peer-mapped inboxes (csrc/ipc.inc): handles and slot tables travel over the bootstrap group
                mine = ctx.comm_ipc_export(self.world)
                everyone = [None] * self.world
                if self.world > 1:
                    dist.all_gather_object(everyone, mine, group=group)
                else:
                    everyone = [mine]
                ctx.comm_init_ipc([e[0] for e in everyone], [e[1] for e in everyone])
            elif transport == "gloo":
                ctx.comm_init_callbacks(self._halo_cb, self._allreduce_cb)
            else:
                raise ValueError(f"unknown transport {transport!r}")
        if self.deep is not None:
            ctx.set_poisson_options(rtol=options.pcg_rtol, max_iter=options.pcg_max_iter, nu=options.amg_smoothing_sweeps,
                                    edge_currents_every_step=options.edge_currents_every_step)
            ctx.set_hierarchy_deep(self.deep, payload["coarse"])
        else:
            ctx.set_hierarchy_sliced(payload["level0"], payload["coarse"], lp)
        self.hierarchy = ctx.hierarchy
        ctx.set_poisson_options(
            rtol=options.pcg_rtol, max_iter=options.pcg_max_iter, nu=options.amg_smoothing_sweeps,
            edge_currents_every_step=options.edge_currents_every_step,
        )
        ctx.set_link_exponents(payload["link_exponents"])
        ctx.set_epsilon(payload["epsilon"])
        ctx.set_mu_boundary(payload["mu_boundary"])
        ctx.set_controller(options.dt_init, options.dt_max, options.adaptive, options.adaptive_window,
                           options.max_solve_retries, options.adaptive_time_step_multiplier)
        # screening: global site arrays + this rank's edge centres
        self.screening = payload.get("screening")
        if self.screening is not None:
            ctx.set_screening_distributed(
                self.screening["sites"], self.screening["areas"], lp.local_to_global[: lp.n_own],
                self.screening["edge_centers"],
                max_iterations=options.max_iterations_per_step, tolerance=options.screening_tolerance,
                step_size=options.screening_step_size, step_drag=options.screening_step_drag,
            )
        # probes: each rank reads the ones it owns
        self.n_probes = payload["n_probes"]
        if self.n_probes:
            self._probe_mine = payload["probe_mine"]
            ctx.set_probes(payload["probe_local"])
        # the CG's second preconditioner: rank-level nested dissection (schur_dd.py; one all-reduce of |Gamma| doubles
        # per application where the distributed AMG cycle needs an exchange and two sums per iteration)
        self.schur = None
        sp = payload.get("schur")
        want = sp is not None and self.world > 1 and self.screening is None and (
            schur is True or (schur == "auto" and self.SCHUR_MIN_SITES <= self.n_global and lp.n_own <= TDGLContext.PD_MAX_SITES
                              and sp["n_gamma"] <= self.SCHUR_MAX_INTERFACE))
        if want:
            import torch

            from .schur_dd import build_piece

            def reducer(op):
                def f(a):
                    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
                    dist.all_reduce(t, op=op, group=group)
                    return t.numpy()
                return f

            piece = build_piece(lp, sp["is_gamma"], sp["gid"], sp["n_gamma"], blocks=schur_blocks)
            if ctx.build_schur_precond(piece, reducer(dist.ReduceOp.SUM), reducer(dist.ReduceOp.MAX), choice=schur_choice):
                self.schur = ctx.precond_direct

    # meshes from here on get the rank-level dissection by default (`schur="auto"`), interfaces up to this many sites
    SCHUR_MIN_SITES = 100_000
    SCHUR_MAX_INTERFACE = 16_000

    # -- gloo transport (tests) -----------------------------------------------------------------
    def _halo_cb(self, send, send_off, recv, recv_off, ranks):
        import torch

        dist = self.dist
        reqs, bufs = [], []
        # (an empty list on this side is an empty list on the other side: both skip the message)
        for k, nb in enumerate(ranks):
            t = torch.empty(int(recv_off[k + 1] - recv_off[k]), dtype=torch.float64)
            bufs.append(t)
            if t.numel():
                reqs.append(dist.irecv(t, src=int(nb), group=self.group))
        for k, nb in enumerate(ranks):
            if send_off[k + 1] > send_off[k]:
                reqs.append(dist.isend(torch.from_numpy(send[send_off[k]:send_off[k + 1]].copy()), dst=int(nb), group=self.group))
        for r in reqs:
            r.wait()
        for k in range(len(ranks)):
            recv[recv_off[k]:recv_off[k + 1]] = bufs[k].numpy()

    def _allreduce_cb(self, buf, op):
        import torch

        t = torch.from_numpy(buf.copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MAX, group=self.group)
        buf[:] = t.numpy()

    # -- first contact ----------------------------------------------------------------------------
    def selftest(self):
        """One exchange per pattern and one sum per kind through the transport in use, on values that are functions
        of the GLOBAL site id / of the rank, checked entry by entry.  Every rank calls it; returns a report
        (``dict(ok, rank, device, neighbours, checks=[...])``) instead of raising, so that a launcher can print all
        ranks' findings.  (`tdgl_comm_test_halo`, `tdgl_comm_test_allreduce`)"""
        lp, ctx, w = self.lp, self.ctx, self.world
        self._meet()
        report = dict(rank=self.rank, world=w, neighbours=list(lp.neighbors), n_own=int(lp.n_own), ghosts=int(lp.n_ghost), checks=[])

        def f(gid, c=0):  # exactly representable, different for every site and component
            return (np.asarray(gid, dtype=np.float64) * 4.0 + c + 1.0)

        def check(name, got, want, who):
            bad = np.flatnonzero(got != want)
            entry = dict(name=name, ok=len(bad) == 0, entries=int(len(want)))
            if len(bad):
                k = int(bad[0])
                entry.update(first_bad=k, owner=int(who[k]) if who is not None else None, got=float(got[k]), want=float(want[k]),
                             n_bad=int(len(bad)))
            report["checks"].append(entry)

        try:
            l2g = lp.local_to_global
            part_of_ghost = None
            for width in (1, 2):  # mu / the PCG vectors; psi
                v = np.full((len(l2g), width), -1.0)
                for c in range(width):
                    v[: lp.n_own, c] = f(l2g[: lp.n_own], c)
                got = ctx.comm_test_halo(v.ravel(), width=width).reshape(len(l2g), width)
                want = np.stack([f(l2g, c) for c in range(width)], axis=1)
                owner = np.concatenate([np.full(b - a, nb) for nb, (a, b) in sorted(lp.recv_range.items(), key=lambda kv: kv[1][0])]) \
                    if lp.neighbors else np.empty(0, dtype=int)
                part_of_ghost = np.repeat(owner, width) if len(owner) else None
                check(f"first-layer exchange, width {width}", got[lp.n_own:].ravel(), want[lp.n_own:].ravel(), part_of_ghost)
            dp = self.deep
            if dp is not None:
                v = np.full(dp.n_ext, -1.0)
                v[: dp.n_own] = f(dp.ext_to_global[: dp.n_own])
                got = ctx.comm_test_halo(v, width=1, deep=True)
                who = np.full(dp.n_ext, -1)
                for nb, idx in dp.recv_idx.items():
                    who[idx] = nb
                check("deep exchange of the residual", got[dp.n_own:], f(dp.ext_to_global[dp.n_own:]), who[dp.n_own:])
            x = np.arange(3072, dtype=np.float64) + 1000.0 * (self.rank + 1)
            check("sum of 3 x 1024 partials", ctx.comm_test_allreduce(x), np.arange(3072.0) * w + 1000.0 * w * (w + 1) / 2, None)
            check("maximum", ctx.comm_test_allreduce(np.array([float(self.rank), -float(self.rank)]), op="max"),
                  np.array([float(w - 1), 0.0]), None)
            n2 = 20000 if dp is None else int(dp.M.shape[0])
            y = (np.arange(n2) % 257).astype(np.float64) + self.rank  # small integers: exact in fp32
            check("level-2-sized sum carried as fp32", ctx.comm_test_allreduce(y, as_f32=True),
                  (np.arange(n2) % 257) * float(w) + w * (w - 1) / 2, None)
            report["ok"] = all(c["ok"] for c in report["checks"])
        except Exception as exc:  # a transport error (RCCL, a peer that never answered): report it
            report["ok"] = False
            report["error"] = f"{type(exc).__name__}: {exc}"
        return report

    # -- inputs -----------------------------------------------------------------------------------
    def set_mu_boundary(self, mu_boundary_global):
        self.ctx.set_mu_boundary(np.asarray(mu_boundary_global, dtype=float)[self.lp.boundary_positions])

    def set_state(self, psi_global, mu_global):
        """Global arrays (every rank passes the same ones), or scalars for a uniform state."""
        l2g = self.lp.local_to_global
        psi = np.full(len(l2g), psi_global, dtype=complex) if np.ndim(psi_global) == 0 else np.asarray(psi_global)[l2g]
        mu = np.full(len(l2g), mu_global, dtype=float) if np.ndim(mu_global) == 0 else np.asarray(mu_global, dtype=float)[l2g]
        if np.ndim(psi_global) == 0 and self.options.terminal_psi is not None and len(self.lp.fixed_sites):
            psi[self.lp.fixed_sites] = self.options.terminal_psi  # solver.py:285-287
        self.ctx.set_state(psi, mu)

    def begin_stage(self):
        self.ctx.begin_stage()

    # -- stepping -----------------------------------------------------------------------------------
    def _meet(self):
        """Peer-mapped transport: the ranks meet on the host before a batch of kernels is queued.  The in-kernel waits
        are bounded (`tdgl_comm_ipc_set_timeout`; two minutes unless told otherwise), and nothing else bounds the skew
        between ranks -- one of them saving a snapshot, a slow file system -- so the host takes it out here."""
        if self.world > 1 and self.transport == "ipc":
            self.dist.barrier(group=self.group)

    def run(self, max_steps, end_time=np.inf, host_barrier=True):
        """``host_barrier=False`` skips the meeting above (tests of the device-side time-out)."""
        if host_barrier:
            self._meet()
        failure = None
        try:
            res = self.ctx.run(max_steps, end_time)
        except RuntimeError as exc:
            if not (host_barrier and self.world > 1 and self.transport == "ipc"):
                raise
            failure = exc
        if host_barrier and self.world > 1 and self.transport == "ipc":
            # one rank's failed exchange is every rank's failure: the device side poisons the peers' flags, the host
            # side agrees on the outcome (a rank that has not looked at its error flag yet must not return a result)
            import torch

            flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, group=self.group)
            if failure is not None:
                raise failure
            if int(flag.item()):
                raise RuntimeError("tdgl_run failed on another rank (peer-mapped transport): the exchanges of this batch are not to be trusted")
        if self.n_probes and self.world > 1:
            import torch

            k, npb = len(res["dt"]), self.n_probes
            both = np.zeros((2, k, npb))
            if len(self._probe_mine):
                both[0][:, self._probe_mine] = res["mu"]
                both[1][:, self._probe_mine] = res["theta"]
            t = torch.from_numpy(both)
            self.dist.all_reduce(t, group=self.group)
            res["mu"], res["theta"] = t.numpy()[0], t.numpy()[1]
        return res

    def gather_state(self):
        """Global psi, mu (sites) and J_s, J_n (edges) assembled on every rank."""
        import torch

        lp = self.lp
        st = self.ctx.get_state()
        n, m = self.n_global, self.m_global
        own = lp.local_to_global[: lp.n_own]
        sites = np.zeros((3, n))
        sites[0, own], sites[1, own], sites[2, own] = st["psi"].real[: lp.n_own], st["psi"].imag[: lp.n_own], st["mu"][: lp.n_own]
        edges = np.zeros((4, m))
        mask = lp.owned_edge_mask
        ge = lp.edge_local_to_global[mask]
        edges[0, ge], edges[1, ge] = st["supercurrent"][mask], st["normal_current"][mask]
        if self.screening is not None:
            a_ind = self.ctx.induced_vector_potential()
            edges[2, ge], edges[3, ge] = a_ind[mask, 0], a_ind[mask, 1]
        if self.world > 1:
            for arr in (sites, edges):
                t = torch.from_numpy(arr)
                self.dist.all_reduce(t, group=self.group)
        out = dict(psi=sites[0] + 1j * sites[1], mu=sites[2], supercurrent=edges[0], normal_current=edges[1])
        if self.screening is not None:
            out["induced_vector_potential"] = np.column_stack([edges[2], edges[3]])
        return out

    def close(self):
        self.ctx.close()
