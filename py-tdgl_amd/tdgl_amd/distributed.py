"""One-process-per-GPU driver: domain decomposition of one simulation over the GPUs of a node.

Usage (every rank of a `torch.distributed` job runs the same code):

    import torch.distributed as dist
    dist.init_process_group("gloo")          # bootstrap / gather only; the data path is RCCL
    run = DistributedTDGL(mesh, options, A, rank=dist.get_rank(), world=dist.get_world_size())
    run.set_state(psi0, mu0); run.begin_stage()
    out = run.run(100)                       # identical dt sequence on every rank
    fields = run.gather_state()              # global psi, mu, J_s, J_n on every rank

The mesh is cut by recursive coordinate bisection (`partition.rcb_partition`), every rank builds
the same global AMG hierarchy and uploads its slice.  The exchange of ghost values and the
all-reduces run inside `tdgl_run` over RCCL on the context's stream (transport "rccl"); transport
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
from .partition import build_local_problem, rcb_partition


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


class DistributedTDGL:
    def __init__(self, mesh, options, link_exponents, epsilon=1.0, u=5.79, gamma=10.0, *, rank, world,
                 terminal_info=(), mu_boundary=None, probe_points=None, transport="rccl", device_id=None,
                 overlap="auto", screening=None):
        import torch.distributed as dist

        self.dist = dist
        self.rank, self.world = int(rank), int(world)
        self.mesh = mesh
        self.options = options
        options.validate()
        em = mesh.edge_mesh
        n = len(mesh.sites)
        fixed = (
            np.concatenate([np.asarray(t["site_indices"] if isinstance(t, dict) else t.site_indices)
                            for t in terminal_info]).astype(np.int64)
            if len(terminal_info) else np.array([], dtype=np.int64)
        )
        self.fixed_sites = fixed
        self.part = rcb_partition(mesh.sites, self.world)
        self.lp = lp = build_local_problem(mesh, self.part, self.rank, fixed_sites=fixed)
        dev = self.rank if device_id is None else device_id
        self.ctx = ctx = TDGLContext(
            lp.mesh, fixed_sites=lp.fixed_sites, fix_psi=(options.terminal_psi is not None), u=u, gamma=gamma,
            device_id=dev, n_owned=lp.n_own,
        )
        ctx.set_halo_plan(lp)
        ctx.set_comm_overlap(overlap)  # halo exchanges hidden behind the ghost-free rows
        if self.world > 1 or transport == "rccl":
            if transport == "rccl":
                ident = [ctx.comm_unique_id() if self.rank == 0 else None]
                dist.broadcast_object_list(ident, src=0)
                with stdout_to_stderr():
                    ctx.comm_init_rccl(ident[0])
            elif transport == "gloo":
                ctx.comm_init_callbacks(self._halo_cb, self._allreduce_cb)
            else:
                raise ValueError(f"unknown transport {transport!r}")
        # the same global hierarchy on every rank (deterministic set-up), level 0 sliced
        A_glob = poisson_matrix(em.edges, em.dual_edge_lengths / em.edge_lengths, n)
        # (level 0 is the distributed one, so the hierarchy needs at least one coarser level)
        self.hierarchy = build_hierarchy(A_glob, max_coarse=min(600, max(8, n // 4)))
        ctx.set_hierarchy_distributed(self.hierarchy, lp)
        ctx.set_poisson_options(
            rtol=options.pcg_rtol, max_iter=options.pcg_max_iter, nu=options.amg_smoothing_sweeps,
            edge_currents_every_step=options.edge_currents_every_step,
        )
        l2g = lp.local_to_global
        ctx.set_link_exponents(np.asarray(link_exponents, dtype=float)[lp.edge_local_to_global])
        ctx.set_epsilon((np.asarray(epsilon, dtype=float) * np.ones(n))[l2g])
        self.set_mu_boundary(np.zeros(len(em.boundary_edge_indices)) if mu_boundary is None else mu_boundary)
        ctx.set_controller(options.dt_init, options.dt_max, options.adaptive, options.adaptive_window,
                           options.max_solve_retries, options.adaptive_time_step_multiplier)
        # screening: ``dict(sites[n, 2], areas[n] (scaled), edge_centers[m, 2])`` in GLOBAL numbering
        self.screening = screening
        if screening is not None:
            ctx.set_screening_distributed(
                screening["sites"], screening["areas"], l2g[: lp.n_own],
                np.asarray(screening["edge_centers"])[lp.edge_local_to_global],
                max_iterations=options.max_iterations_per_step, tolerance=options.screening_tolerance,
                step_size=options.screening_step_size, step_drag=options.screening_step_drag,
            )
        # probes: each rank reads the ones it owns
        self.probe_points = None if probe_points is None else np.asarray(probe_points, dtype=np.int64)
        if self.probe_points is not None:
            g2l = np.full(n, -1, dtype=np.int64)
            g2l[l2g[: lp.n_own]] = np.arange(lp.n_own)
            loc = g2l[self.probe_points]
            self._probe_mine = np.flatnonzero(loc >= 0)
            ctx.set_probes(loc[self._probe_mine])

    # -- gloo transport (tests) -----------------------------------------------------------------
    def _halo_cb(self, send, send_off, recv, recv_off, ranks):
        import torch

        dist = self.dist
        reqs, bufs = [], []
        for k, nb in enumerate(ranks):
            t = torch.empty(int(recv_off[k + 1] - recv_off[k]), dtype=torch.float64)
            bufs.append(t)
            reqs.append(dist.irecv(t, src=int(nb)))
        for k, nb in enumerate(ranks):
            reqs.append(dist.isend(torch.from_numpy(send[send_off[k]:send_off[k + 1]].copy()), dst=int(nb)))
        for r in reqs:
            r.wait()
        for k in range(len(ranks)):
            recv[recv_off[k]:recv_off[k + 1]] = bufs[k].numpy()

    def _allreduce_cb(self, buf, op):
        import torch

        t = torch.from_numpy(buf.copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MAX)
        buf[:] = t.numpy()

    # -- inputs -----------------------------------------------------------------------------------
    def set_mu_boundary(self, mu_boundary_global):
        self.ctx.set_mu_boundary(np.asarray(mu_boundary_global, dtype=float)[self.lp.boundary_positions])

    def set_state(self, psi_global, mu_global):
        l2g = self.lp.local_to_global
        self.ctx.set_state(np.asarray(psi_global)[l2g], np.asarray(mu_global, dtype=float)[l2g])

    def begin_stage(self):
        self.ctx.begin_stage()

    # -- stepping -----------------------------------------------------------------------------------
    def run(self, max_steps, end_time=np.inf):
        res = self.ctx.run(max_steps, end_time)
        if self.probe_points is not None and self.world > 1:
            import torch

            k, npb = len(res["dt"]), len(self.probe_points)
            both = np.zeros((2, k, npb))
            if len(self._probe_mine):
                both[0][:, self._probe_mine] = res["mu"]
                both[1][:, self._probe_mine] = res["theta"]
            t = torch.from_numpy(both)
            self.dist.all_reduce(t)
            res["mu"], res["theta"] = t.numpy()[0], t.numpy()[1]
        return res

    def gather_state(self):
        """Global psi, mu (sites) and J_s, J_n (edges) assembled on every rank."""
        import torch

        lp = self.lp
        st = self.ctx.get_state()
        n, m = len(self.mesh.sites), len(self.mesh.edge_mesh.edges)
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
                self.dist.all_reduce(t)
        out = dict(psi=sites[0] + 1j * sites[1], mu=sites[2], supercurrent=edges[0], normal_current=edges[1])
        if self.screening is not None:
            out["induced_vector_potential"] = np.column_stack([edges[2], edges[3]])
        return out

    def close(self):
        self.ctx.close()
