"""``TDGLSolver`` / ``solve``: the reference's solver API (`tdgl/solver/solver.py:88-827`,
`tdgl/solver/solve.py:9-52`) driving the HIP time-stepping core.

What runs where:

* setup (this file, Python): unit scaling of the applied vector potential and terminal
  currents, epsilon evaluation, terminal detection -- the work of
  ``TDGLSolver.__init__`` (solver.py:117-323);
* every time step (libtdgl_hip, C++/HIP): psi update with retries, Poisson solve, currents,
  probe read-out, adaptive-dt controller and the Runner loop's time/step accounting
  (``tdgl_run``); Python is entered once per ``save_every`` steps (or once per step when the
  terminal currents are a function of time, because the callable lives in Python).

Time-dependent ``applied_vector_potential`` (a ``Parameter`` with a keyword-only ``t``) and
``disorder_epsilon`` are evaluated in Python once per step and uploaded; dA/dt and the link
variables are then formed on the device (``tdgl_update_link_exponents``).
``include_screening`` runs the reference's self-consistent loop for the induced vector
potential (solver.py:522-578, 654-688) entirely on the device; Python only hands over the site /
edge coordinates and the scaled site areas (solver.py:305-309).  Not supported: HDF5 output.
"""

import inspect
import logging
import numbers
import time as _time
from typing import Callable, Dict, NamedTuple, Optional, Sequence, Union

import numpy as np

from .device import Device, TerminalInfo
from .operators import MeshOperators
from .options import SolverOptions
from .solution import DynamicsData, Solution, TDGLData

logger = logging.getLogger("solver")


def validate_terminal_currents(terminal_currents, terminal_info: Sequence[TerminalInfo],
                               solver_options: SolverOptions, num_evals: int = 100) -> None:
    """Terminal currents must name known terminals and sum to zero (solver.py:35-60)."""
    known = {t.name for t in terminal_info}

    def check(currents: Dict[str, float]):
        unknown = set(currents).difference(known)
        if unknown:
            raise ValueError(f"Unknown terminal(s) in terminal currents: {list(unknown)}.")
        total = sum(currents.values())
        if total:
            raise ValueError(f"The sum of all terminal currents must be 0 (got {total:.2e}).")

    if callable(terminal_currents):
        for t in np.random.default_rng().random(num_evals) * solver_options.solve_time:
            check(terminal_currents(t))
    else:
        check(terminal_currents)


class SolverResult(NamedTuple):
    """Result of one solve step (same fields as solver.py:63-85)."""

    dt: float
    psi: np.ndarray
    mu: np.ndarray
    supercurrent: np.ndarray
    normal_current: np.ndarray
    A_induced: np.ndarray
    A_applied: Optional[np.ndarray] = None
    epsilon: Optional[np.ndarray] = None


def uniform_field_vector_potential(x, y, B):
    """Symmetric-gauge vector potential of a uniform field ``B`` (in [field] * [length]),
    centred on the bounding box of the evaluation points -- what the reference's
    ``ConstantField`` evaluates (`tdgl/em.py:437-472`, `tdgl/sources/constant.py:7-21`)."""
    xs = x - (x.min() + np.ptp(x) / 2)
    ys = y - (y.min() + np.ptp(y) / 2)
    return np.stack([-B * ys / 2, B * xs / 2, np.zeros_like(xs)], axis=1)


class TDGLSolver:
    """Solver for a TDGL model; same constructor as the reference (solver.py:117-125)."""

    def __init__(
        self,
        device: Device,
        options: SolverOptions,
        applied_vector_potential: Union[Callable, float] = 0.0,
        terminal_currents: Union[Callable, Dict[str, float], None] = None,
        disorder_epsilon: Union[Callable, float] = 1.0,
        seed_solution: Optional[Solution] = None,
    ):
        self.device = device
        self.options = options
        options.validate()
        self.terminal_currents = terminal_currents
        self.seed_solution = seed_solution
        if device.mesh is None:
            raise ValueError("The device has no mesh: call device.make_mesh() first.")
        mesh = device.mesh
        em = mesh.edge_mesh
        xi = device.coherence_length
        self.u, self.gamma = device.layer.u, device.layer.gamma
        self.probe_points = device.probe_point_indices
        self.num_edges = len(em.edges)
        self.sites = xi * mesh.sites
        self.edge_centers = xi * em.centers
        self.z0 = device.layer.z0 * np.ones(len(self.edge_centers))

        # ---- applied vector potential on the edges (solver.py:158-189) ------------------
        self.dynamic_vector_potential = bool(
            getattr(applied_vector_potential, "time_dependent", False)
        )
        self.applied_vector_potential = applied_vector_potential
        self.A_scale = device.field_scale(options.field_units)
        ex, ey = self.edge_centers[:, 0], self.edge_centers[:, 1]
        self.vector_potential_func = None
        self._A_base = self._A_factor = self._A_ramp = None
        sep = applied_vector_potential.separable_product() if (
            self.dynamic_vector_potential and hasattr(applied_vector_potential, "separable_product")) else None
        if sep is not None:
            # A(t) = f(t) * A_static (the reference's field-ramp example): A_static stays on the
            # device, the factor is evaluated per step -- by tdgl_run itself for a LinearRamp
            factor, static = sep
            self._A_base = self.A_scale * np.asarray(static(ex, ey, self.z0))[:, :2]
            self._A_factor = factor.scalar
            self._A_ramp = factor.ramp
            self.vector_potential_func = lambda t: factor.scalar(t) * self._A_base
            A = self.vector_potential_func(0)
        elif self.dynamic_vector_potential:
            # solver.py:347-362: A(t) on the edge centres, scaled to dimensionless units
            def vector_potential_func(t):
                return self.A_scale * np.asarray(applied_vector_potential(ex, ey, self.z0, t=t))[:, :2]

            self.vector_potential_func = vector_potential_func
            A = vector_potential_func(0)
        elif callable(applied_vector_potential):
            A = self.A_scale * np.asarray(applied_vector_potential(ex, ey, self.z0))[:, :2]
        else:
            A = self.A_scale * uniform_field_vector_potential(ex, ey, float(applied_vector_potential))[:, :2]
        if A.shape != self.edge_centers.shape:
            raise ValueError(f"Unexpected shape for vector_potential: {A.shape}.")
        self.current_A_applied = A

        # ---- disorder parameter epsilon on the sites (solver.py:191-216) ----------------
        self.epsilon_func = None
        self.dynamic_epsilon = False
        self._eps_table = None
        from .parameter import SeparableEpsilon, TabulatedCurrents

        if isinstance(disorder_epsilon, SeparableEpsilon):  # factor(t) * static(r): evaluated on the device
            eps0 = disorder_epsilon.static_values(self.sites)
            self._eps_table = (eps0, disorder_epsilon.factor.times, disorder_epsilon.factor.values)
        if callable(disorder_epsilon):
            spec = inspect.getfullargspec(disorder_epsilon)
            self.dynamic_epsilon = "t" in spec.kwonlyargs
            vectorized = spec.kwonlydefaults is not None and spec.kwonlydefaults.get("vectorized", False)

            def evaluate(**kw):  # solver.py:211-214, 364-381
                if vectorized:
                    return np.asarray(disorder_epsilon(self.sites, **kw), dtype=float)
                return np.array([float(disorder_epsilon(r, **kw)) for r in self.sites])

            if self.dynamic_epsilon:
                self.epsilon_func = lambda t: evaluate(t=t)  # noqa: E731
                epsilon = evaluate(t=0)
            else:
                epsilon = evaluate()
        else:
            epsilon = float(disorder_epsilon) * np.ones(len(self.sites))
        if np.any(epsilon > 1):
            raise ValueError("The disorder parameter epsilon must be <= 1")
        self.disorder_epsilon = disorder_epsilon
        self.epsilon = epsilon

        # ---- terminals and currents (solver.py:224-265) ------------------------------------
        self.terminal_info = device.terminal_info()
        self.terminal_names = [t.name for t in self.terminal_info]
        for t in self.terminal_info:
            if t.length == 0:
                raise ValueError(
                    f"Terminal {t.name!r} does not contain any points on the boundary of the mesh."
                )
        if terminal_currents and device.probe_points is None:
            logger.warning("The terminal currents are non-null, but the device has no probe points.")
        names = self.terminal_names
        self.dynamic_currents = callable(terminal_currents)
        if terminal_currents is None:
            terminal_currents = {name: 0 for name in names}
        if self.dynamic_currents:
            raw_func = terminal_currents
        else:
            const = {name: terminal_currents.get(name, 0) for name in names}
            unknown = set(terminal_currents).difference(names)
            if unknown:
                raise ValueError(f"Unknown terminal(s) in terminal currents: {list(unknown)}.")

            def raw_func(t):
                return const

        J_scale = device.current_scale(options.current_units)
        self._current_table = terminal_currents.scaled(J_scale) if isinstance(terminal_currents, TabulatedCurrents) else None
        self.current_func = lambda t: {k: J_scale * v for k, v in raw_func(t).items()}
        validate_terminal_currents(self.current_func, self.terminal_info, options)
        # ---- screening (solver.py:305-309) ------------------------------------------------------
        # (mu_0 / 4 pi) K0 / A0 = 1 / (pi Lambda): converts the 1/r integral of the sheet
        # current (units of K0) into a vector potential in units of A0 = xi Bc2.
        self.screening = None
        if options.include_screening:
            a_scale = 1.0 / (np.pi * device.Lambda)
            self.screening = dict(
                sites=self.sites, edge_centers=self.edge_centers, areas=a_scale * mesh.areas * xi**2
            )
        self._setup(mesh)

    @classmethod
    def from_dimensionless(cls, mesh, options: SolverOptions, link_exponents, epsilon=1.0,
                           u: float = 5.79, gamma: float = 10.0, terminal_info=(),
                           current_func=None, probe_points=None, device=None,
                           vector_potential_func=None, epsilon_func=None,
                           screening=None, vector_potential_ramp=None) -> "TDGLSolver":
        """Build a solver directly from dimensionless inputs -- the arrays the reference's
        ``__init__`` ends up with (solver.py:185, 214, 225, 254-256): ``A[m, 2]``,
        ``epsilon[n]``, ``TerminalInfo`` records and ``t -> {name: dimensionless current}``.
        Used by the parity tests and the benchmark, whose configurations are stated in
        dimensionless form (b = B/Bc2, xi = 1).  ``screening``: ``{sites, edge_centers, areas}``
        (areas already multiplied by the kernel prefactor, solver.py:307-309); requires
        ``options.include_screening``.  ``vector_potential_ramp``: ``(A_base[m, 2], dict(tmin, tmax,
        initial, final))`` for ``A(t) = LinearRamp(t) * A_base`` evaluated by the library itself
        (then ``link_exponents`` must be its value at t = 0)."""
        self = object.__new__(cls)
        options.validate()
        self.device = device
        self.options = options
        self.terminal_currents = None
        self.seed_solution = None
        self.u, self.gamma = u, gamma
        self.probe_points = None if probe_points is None else [int(p) for p in probe_points]
        self.num_edges = len(mesh.edge_mesh.edges)
        self.sites = mesh.sites
        self.applied_vector_potential = None
        self.disorder_epsilon = epsilon
        # optional time dependence: t -> A[m, 2] / t -> epsilon[n], already dimensionless
        self.vector_potential_func = vector_potential_func
        self._A_base = self._A_factor = self._A_ramp = None
        if vector_potential_ramp is not None:
            from .parameter import LinearRamp

            base, ramp = vector_potential_ramp
            factor = LinearRamp(**ramp)
            self._A_base = np.asarray(base, dtype=float)
            self._A_factor, self._A_ramp = factor.scalar, factor.ramp
            self.vector_potential_func = vector_potential_func = lambda t: factor.scalar(t) * self._A_base
        self.epsilon_func = epsilon_func
        self.dynamic_vector_potential = vector_potential_func is not None
        self.dynamic_epsilon = epsilon_func is not None
        A = np.asarray(link_exponents, dtype=float)
        if A.shape != (self.num_edges, 2):
            raise ValueError(f"Unexpected shape for vector_potential: {A.shape}.")
        self.current_A_applied = A
        self.epsilon = np.asarray(epsilon, dtype=float) * np.ones(len(mesh.sites))
        if np.any(self.epsilon > 1):
            raise ValueError("The disorder parameter epsilon must be <= 1")
        info = [t if isinstance(t, TerminalInfo) else TerminalInfo(
            t["name"], t["site_indices"], t.get("edge_indices", []), t["boundary_edge_indices"], t["length"])
            for t in terminal_info]
        self.terminal_info = tuple(sorted(info, key=lambda t: t.length))
        self.terminal_names = [t.name for t in self.terminal_info]
        from .parameter import TabulatedCurrents

        self._current_table = current_func if isinstance(current_func, TabulatedCurrents) else None
        self._eps_table = None
        self.dynamic_currents = callable(current_func)
        if not self.dynamic_currents:  # None or a constant {name: current} dict
            const = {name: 0 for name in self.terminal_names}
            const.update(current_func or {})
            current_func = lambda t: const  # noqa: E731
        self.current_func = current_func
        validate_terminal_currents(self.current_func, self.terminal_info, options)
        if options.include_screening and screening is None:
            raise ValueError("include_screening=True needs the screening geometry (sites, edge_centers, areas).")
        self.screening = dict(screening) if options.include_screening else None
        self._setup(mesh)
        return self

    def _setup(self, mesh) -> None:
        """Device-side set-up shared by both constructors (solver.py:258-320)."""
        options = self.options
        em = mesh.edge_mesh
        names = self.terminal_names
        idx = [np.asarray(t.site_indices) for t in self.terminal_info]
        self.fixed_sites = np.concatenate(idx).astype(np.int64) if idx else np.array([], dtype=np.int64)
        self.terminal_current_densities = {name: 0 for name in names}

        # ---- operators on the device (solver.py:267-282) --------------------------------------
        terminal_psi = options.terminal_psi
        self.operators = MeshOperators(
            mesh,
            options.sparse_solver,
            fixed_sites=self.fixed_sites,
            fix_psi=(terminal_psi is not None),
            u=self.u,
            gamma=self.gamma,
            device_id=options.device_id,
            pcg_rtol=options.pcg_rtol,
            pcg_max_iter=options.pcg_max_iter,
            amg_smoothing_sweeps=options.amg_smoothing_sweeps,
            edge_currents_every_step=options.edge_currents_every_step,
            precond_fp32=options.pcg_precond_fp32,
        )
        self.operators.build_operators()
        self.ctx = self.operators.ctx
        if self._A_base is not None:
            self.ctx.set_link_exponents_base(self._A_base, self._A_factor(0))
            if self._A_ramp is not None:
                self.ctx.set_link_ramp(**self._A_ramp)
        else:
            self.operators.set_link_exponents(self.current_A_applied)

        # ---- initial values (solver.py:284-289) ------------------------------------------------
        self.psi_init = np.ones(len(mesh.sites), dtype=np.complex128)
        if terminal_psi is not None:
            self.psi_init[self.fixed_sites] = terminal_psi
        self.mu_init = np.zeros(len(mesh.sites))
        self.mu_boundary = np.zeros(len(em.boundary_edge_indices))
        self.ctx.set_epsilon(self.epsilon)
        self.ctx.set_mu_boundary(self.mu_boundary)
        self.ctx.set_probes(self.probe_points)
        self.ctx.set_controller(
            options.dt_init, options.dt_max, options.adaptive, options.adaptive_window,
            options.max_solve_retries, options.adaptive_time_step_multiplier,
        )
        if self.screening is not None:
            self.ctx.set_screening(
                self.screening["sites"], self.screening["edge_centers"], self.screening["areas"],
                max_iterations=options.max_iterations_per_step, tolerance=options.screening_tolerance,
                step_size=options.screening_step_size, step_drag=options.screening_step_drag,
            )
        # tabulated time dependence: uploaded once, evaluated by tdgl_run at every step's time
        self._currents_on_device = self._epsilon_on_device = False
        if self._current_table is not None and self.terminal_info:
            tab = self._current_table
            names = self.terminal_names
            dens = np.array([
                (-1.0 / term.length) * sum(tab.tables[nm].values for nm in names if nm != term.name and nm in tab.tables)
                * np.ones(len(tab.times))
                for term in self.terminal_info
            ])
            self.ctx.set_mu_boundary_table(tab.times, [np.asarray(t.boundary_edge_indices) for t in self.terminal_info], dens)
            self._currents_on_device = True
        if self._eps_table is not None:
            eps0, times, values = self._eps_table
            if max(float(np.max(f * eps0)) for f in values) > 1:
                raise ValueError("The disorder parameter epsilon must be <= 1")
            self.ctx.set_epsilon_table(eps0, times, values)
            self._epsilon_on_device = True
        self._device_holds = None  # (psi, mu) arrays known to equal the device state

    # -- boundary conditions --------------------------------------------------------------------
    def update_mu_boundary(self, time: float) -> bool:
        """solver.py:325-345.  Returns True if mu_boundary changed (and was re-uploaded)."""
        if self._currents_on_device:
            return False  # tdgl_run evaluates the tables itself (tdgl_set_mu_boundary_table)
        currents = self.current_func(time)
        changed = False
        for term in self.terminal_info:
            density = (-1 / term.length) * sum(
                currents.get(name, 0) for name in self.terminal_names if name != term.name
            )
            if density != self.terminal_current_densities[term.name]:
                self.terminal_current_densities[term.name] = density
                self.mu_boundary[term.boundary_edge_indices] = density
                changed = True
        if changed:
            self.ctx.set_mu_boundary(self.mu_boundary)
        return changed

    def update_dynamic_inputs(self, time: float, dt_prev: float) -> None:
        """solver.py:626-648: re-evaluate A(t) (-> link variables and dA/dt with the previous
        step's dt) and epsilon(t) before a step."""
        if self._A_ramp is not None:
            pass  # tdgl_run evaluates the ramp itself (tdgl_set_link_ramp)
        elif self._A_base is not None:
            self.ctx.update_link_scale(self._A_factor(time), dt_prev)
        elif self.dynamic_vector_potential:
            self.current_A_applied = np.asarray(self.vector_potential_func(time), dtype=float)
            self.ctx.update_link_exponents(self.current_A_applied, dt_prev)
        if self.dynamic_epsilon and self._epsilon_on_device:
            self.epsilon = np.asarray(self.epsilon_func(time), dtype=float)  # host copy for the saved steps
        elif self.dynamic_epsilon:
            self.epsilon = np.asarray(self.epsilon_func(time), dtype=float)
            if np.any(self.epsilon > 1):
                raise ValueError("The disorder parameter epsilon must be <= 1")
            self.ctx.set_epsilon(self.epsilon)

    # -- one step through the reference's method seam ---------------------------------------------
    def update(self, state: Dict[str, numbers.Real], running_state, dt: float, *, psi, mu,
               supercurrent=None, normal_current=None, induced_vector_potential=None,
               applied_vector_potential=None, epsilon=None) -> SolverResult:
        """One call of ``TDGLSolver.update`` (solver.py:580-714).

        Compatibility entry point: the fields travel host -> device -> host on every call.
        ``solve()`` does not use it; it keeps the state resident and batches steps.
        """
        ctx = self.ctx
        held = self._device_holds
        if held is None or held[0] is not psi or held[1] is not mu:
            ctx.set_state(psi, mu)
        ctx.set_loop_state(state["step"], state["time"], state.get("dt", dt))
        self.update_mu_boundary(state["time"])
        self.update_dynamic_inputs(state["time"], dt)
        if self.screening is not None and induced_vector_potential is not None:
            ctx.set_induced_vector_potential(induced_vector_potential)
        res = ctx.run(1, np.inf)
        out = ctx.get_state()
        # The identity test above skips the upload when the caller hands back exactly these
        # arrays; make them read-only so that an in-place edit (seeding a vortex, zeroing a
        # region) fails loudly instead of being ignored -- edit a copy and pass that.
        out["psi"].setflags(write=False)
        out["mu"].setflags(write=False)
        self._device_holds = (out["psi"], out["mu"])
        step_dt = float(res["dt"][0])
        if running_state is not None:
            running_state.append("dt", step_dt)
            if self.screening is not None:
                running_state.append("screening_iterations", int(res["screening_iterations"][0]))
            if self.probe_points is not None:
                running_state.append("mu", res["mu"][0])
                running_state.append("theta", res["theta"][0])
        if self.screening is not None:
            a_ind = ctx.induced_vector_potential()
        else:
            a_ind = np.zeros((self.num_edges, 2)) if induced_vector_potential is None else induced_vector_potential
        extra = []
        if self._A_base is not None:
            self.current_A_applied = ctx.link_scale() * self._A_base
        if self.dynamic_vector_potential:
            extra.append(self.current_A_applied)
        if self.dynamic_epsilon:
            extra.append(self.epsilon)
        return SolverResult(step_dt, out["psi"], out["mu"], out["supercurrent"], out["normal_current"], a_ind, *extra)

    # -- the whole simulation ------------------------------------------------------------------------
    def solve(self) -> Optional[Solution]:
        """Run thermalisation + simulation stages with the reference's loop semantics
        (runner.py:288-454) and return the saved steps in memory."""
        opts = self.options
        opts.validate()
        ctx = self.ctx
        t_start = _time.perf_counter()
        if self.seed_solution is None:
            psi0, mu0 = self.psi_init, self.mu_init
        else:
            if self.seed_solution.device != self.device:
                raise ValueError("The seed_solution.device must be equal to the device being simulated.")
            seed = self.seed_solution.tdgl_data
            if len(seed.psi) != len(self.device.mesh.sites):  # (equal devices may carry different meshes)
                raise ValueError(
                    f"The seed solution has {len(seed.psi)} sites, the device's mesh {len(self.device.mesh.sites)}.")
            psi0, mu0 = seed.psi, seed.mu
        ctx.set_state(psi0, mu0)
        if self.screening is not None:  # solver.py:738-745
            a0 = None if self.seed_solution is None else self.seed_solution.tdgl_data.induced_vector_potential
            ctx.set_induced_vector_potential(np.zeros((self.num_edges, 2)) if a0 is None else a0)
        ctx.set_controller(
            opts.dt_init, opts.dt_max, opts.adaptive, opts.adaptive_window,
            opts.max_solve_retries, opts.adaptive_time_step_multiplier,
        )
        saved = []
        dyn = dict(dt=[], time=[], mu=[], theta=[], iters=[], scr=[])
        n_steps = {"Thermalizing": 0, "Simulating": 0}
        # Streaming output (runner.py:104-183): with SolverOptions.output_file every saved step is
        # written when it is taken, and only the latest one is kept in memory.
        handler = None
        sizes = {"dt": 1}
        if self.probe_points is not None:
            sizes["mu"] = sizes["theta"] = len(self.probe_points)
        if self.screening is not None:
            sizes["screening_iterations"] = 1
        from .io import DataHandler, RunningState, write_solution_group

        running = RunningState(sizes, opts.save_every)
        if opts.output_file is not None:
            handler = DataHandler(opts.output_file, file_factory=getattr(self, "_h5_file_factory", None))
            try:
                handler.__enter__()
                handler.save_mesh(self.device.mesh)
                fixed = {}
                if not self.dynamic_vector_potential:
                    fixed["applied_vector_potential"] = self.current_A_applied
                if not self.dynamic_epsilon:
                    fixed["epsilon"] = self.epsilon
                handler.save_fixed_values(fixed)
            except BaseException:
                handler.close()  # no open files / stray .tmp left behind
                raise

        def save_step(final=False):
            ls = ctx.loop_state()
            if self.dynamic_epsilon and self._epsilon_on_device:
                # the reference saves the epsilon its last update() evaluated (solver.py:645-648): at the
                # time of the last step taken
                t_last = ls["time"] if (final or ls["step"] == 0) else ls["time"] - ls["dt"]
                self.epsilon = np.asarray(self.epsilon_func(max(t_last, 0.0)), dtype=float)
            if ls["step"] == 0 and not saved and self.seed_solution is None:
                js = jn = np.zeros(self.num_edges)  # reference initial values (solver.py:736-737)
                st = ctx.get_state(supercurrent=False, normal_current=False)
            else:
                st = ctx.get_state()
                js, jn = st["supercurrent"], st["normal_current"]
            a_ind = ctx.induced_vector_potential() if self.screening is not None else None
            if self._A_base is not None:
                self.current_A_applied = ctx.link_scale() * self._A_base
            data = TDGLData(ls["step"], ls["time"], ls["dt"], st["psi"], st["mu"], js, jn,
                            applied_vector_potential=self.current_A_applied, epsilon=self.epsilon,
                            induced_vector_potential=a_ind)
            if handler is None:
                saved.append(data)
                return
            fields = dict(psi=data.psi, mu=data.mu, supercurrent=js, normal_current=jn,
                          induced_vector_potential=np.zeros((self.num_edges, 2)) if a_ind is None else a_ind)
            if self.dynamic_vector_potential:
                fields["applied_vector_potential"] = self.current_A_applied
            if self.dynamic_epsilon:
                fields["epsilon"] = self.epsilon
            state = dict(step=int(ls["step"]), time=float(ls["time"]), dt=float(ls["dt"]))
            handler.save_time_step(state, fields, None if ls["step"] == 0 else running.export())
            saved[:] = [data]
            saved_meta.append((data.step, data.time))

        saved_meta = []

        def run_stage(name, end_time, save):
            ctx.begin_stage()
            i = 0
            while True:
                if i % opts.save_every == 0:  # runner.py:398-401
                    if save:
                        save_step()
                    running.clear()
                per_step = ((self.dynamic_currents and not self._currents_on_device)
                            or (self.dynamic_epsilon and not self._epsilon_on_device)
                            or (self.dynamic_vector_potential and self._A_ramp is None))
                chunk = 1 if per_step else opts.save_every - (i % opts.save_every)
                ls = ctx.loop_state()
                self.update_mu_boundary(ls["time"] if self.dynamic_currents else 0.0)
                self.update_dynamic_inputs(ls["time"], ls["dt"])
                t_before = ls["time"]
                res = ctx.run(chunk, end_time)
                k = len(res["dt"])
                n_steps[name] += k
                cols = {"dt": res["dt"]}
                if res["mu"] is not None:
                    cols["mu"], cols["theta"] = res["mu"], res["theta"]
                if self.screening is not None:
                    cols["screening_iterations"] = res["screening_iterations"]
                # (the step that ends the loop is written into the buffer but not counted, runner.py:429-432)
                running.extend({name_: v[:k - 1] if res["reached_end"] else v for name_, v in cols.items()})
                if res["reached_end"]:
                    for name_, v in cols.items():
                        running.append(name_, np.asarray(v[k - 1]).reshape(-1))
                if save:
                    dyn["dt"].append(res["dt"])
                    times = t_before + np.concatenate([[0.0], np.cumsum(res["dt"][:-1])])
                    dyn["time"].append(times)
                    dyn["iters"].append(res["pcg_iters"])
                    dyn["scr"].append(res["screening_iterations"])
                    if res["mu"] is not None:
                        dyn["mu"].append(res["mu"])
                        dyn["theta"].append(res["theta"])
                if res["reached_end"]:
                    i += k - 1
                    break
                i += k
            if save and (i % opts.save_every):
                save_step(final=True)

        try:
            if opts.skip_time:
                run_stage("Thermalizing", opts.skip_time, False)
            run_stage("Simulating", opts.solve_time, True)
            ctx.synchronize()
        except BaseException:
            if handler is not None:  # what has been saved stays on disk
                handler.close()
            raise
        total = _time.perf_counter() - t_start
        cat = lambda xs: np.concatenate(xs) if xs else np.array([])  # noqa: E731
        dynamics = DynamicsData(
            dt=cat(dyn["dt"]),
            time=cat(dyn["time"]),
            mu=cat(dyn["mu"]).T if dyn["mu"] else None,
            theta=cat(dyn["theta"]).T if dyn["theta"] else None,
            pcg_iterations=cat(dyn["iters"]),
            screening_iterations=cat(dyn["scr"]) if self.screening is not None else None,
        )
        solution = Solution(
            device=self.device,
            options=opts,
            saved_steps=saved,
            dynamics=dynamics,
            dynamic_vector_potential=self.dynamic_vector_potential,
            dynamic_epsilon=self.dynamic_epsilon,
            applied_vector_potential=self.applied_vector_potential,
            terminal_currents=self.terminal_currents,
            disorder_epsilon=self.disorder_epsilon,
            total_seconds=total,
            stats=dict(
                steps_thermalizing=n_steps["Thermalizing"],
                steps_simulating=n_steps["Simulating"],
                mean_pcg_iterations=float(dynamics.pcg_iterations.mean()) if len(dynamics.pcg_iterations) else 0.0,
                # which mu solve the mesh size selected (hipcore.TDGLContext.build_poisson)
                mu_solver=("direct (substructured)" if getattr(self.ctx, "substructure", None) else
                           "direct (dense inverse)" if getattr(self.ctx, "dense_direct", False) else
                           "pcg (AMG V-cycle or fp32-stored nested-dissection factors, by predicted cost)"
                           if getattr(self.ctx, "precond_direct", None) else "amg_pcg"),
                # the direct solves' in-loop guard: largest ||b - A mu|| / ||b|| over the checked steps
                # (one per batch of queued attempts), how many were checked, and whether a check above
                # 1e-9 sent the run back to AMG-PCG (never observed; the factors deliver 1e-14)
                **{"mu_residual_" + k: v for k, v in self.ctx.direct_stats().items()},
            ),
        )
        if handler is not None:
            solution.path = handler.output_path
            solution.saved_step_index = saved_meta  # (step, time) of every group data/<k> on disk
            write_solution_group(handler.output_file, solution)  # solver.py:815-826 -> Solution.to_hdf5()
            handler.close()
        return solution


def solve(
    device: Device,
    options: SolverOptions,
    applied_vector_potential: Union[Callable, float] = 0,
    terminal_currents: Union[Callable, Dict[str, float], None] = None,
    disorder_epsilon: Union[float, Callable] = 1,
    seed_solution: Optional[Solution] = None,
) -> Union[Solution, None]:
    """Solve a TDGL model (`tdgl.solve`, tdgl/solver/solve.py:9-52)."""
    solver = TDGLSolver(
        device=device,
        options=options,
        applied_vector_potential=applied_vector_potential,
        terminal_currents=terminal_currents,
        disorder_epsilon=disorder_epsilon,
        seed_solution=seed_solution,
    )
    return solver.solve()
