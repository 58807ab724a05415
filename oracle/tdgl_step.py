"""Oracle: one TDGL time step and the surrounding time loop (TEST INFRASTRUCTURE).

Restates, in dimensionless units and without screening:

* `psi_update`        <- `TDGLSolver.solve_for_psi_squared`   (tdgl/solver/solver.py:383-439)
* `OracleSolver._euler_step`  <- `adaptive_euler_step`        (solver.py:441-487)
* `OracleSolver._observables` <- `solve_for_observables`      (solver.py:489-520)
* `OracleSolver._set_terminal_bc` <- `update_mu_boundary`     (solver.py:325-345)
* `OracleSolver.update`       <- `TDGLSolver.update`          (solver.py:580-714)
* `run_time_loop`             <- `Runner.run/_run_stage`      (tdgl/solver/runner.py:288-454)

Inputs are the arrays the reference's `TDGLSolver.__init__` (solver.py:117-323) hands to
its own step: dimensionless link exponents ``A[m,2]``, ``epsilon[n]``, ``u``, ``gamma``,
terminal descriptions and a dimensionless current function.
"""

import itertools
from types import SimpleNamespace

import numpy as np

from .fv_operators import FVOperators


def psi_update(psi, abs_sq_psi, mu, epsilon, gamma, u, dt, psi_laplacian):
    """Implicit-Euler update of the order parameter (solver.py:383-439,
    docs/background.rst:273-353).  Returns ``(psi_new, |psi_new|^2)`` or ``None`` when the
    quadratic for ``|psi_new|^2`` has a negative discriminant somewhere or floating point
    overflow/invalid occurs (both mean "retry with a smaller dt")."""
    phase = np.exp(-1j * mu * dt)
    z = phase * gamma**2 / 2 * psi
    with np.errstate(all="raise"):
        try:
            w = z * abs_sq_psi + phase * (
                psi
                + (dt / u)
                * np.sqrt(1 + gamma**2 * abs_sq_psi)
                * ((epsilon - abs_sq_psi) * psi + psi_laplacian @ psi)
            )
            c = w.real * z.real + w.imag * z.imag
            b = 2 * c + 1
            w2 = np.absolute(w) ** 2
            disc = b**2 - 4 * np.absolute(z) ** 2 * w2
        except Exception:
            return None
    if np.any(disc < 0):
        return None
    new_sq = (2 * w2) / (b + np.sqrt(disc))
    return w - z * new_sq, new_sq


class OracleSolver:
    """Dimensionless TDGL stepper with the reference's per-step semantics.

    Args:
        mesh: object with ``sites, areas, edge_mesh`` (see `fv_operators`).
        link_exponents: ``A[m, 2]`` (dimensionless applied vector potential on edges).
        epsilon: ``epsilon[n]``.
        u, gamma: material parameters (tdgl/device/layer.py:30-31).
        options: object with ``dt_init, dt_max, adaptive, adaptive_window,
            max_solve_retries, adaptive_time_step_multiplier, terminal_psi``.
        terminals: sequence of objects/dicts with ``name, site_indices,
            boundary_edge_indices`` (positions within ``boundary_edge_indices``) and
            ``length`` -- the fields of the reference's ``TerminalInfo`` that the step uses.
        current_func: ``t -> {name: I}`` in dimensionless current units (i.e. already
            multiplied by ``J_scale``, solver.py:251-256); ``None`` = no terminals driven.
        probe_points: site indices whose mu / arg(psi) are recorded each step.
    """

    def __init__(
        self,
        mesh,
        link_exponents,
        epsilon,
        u,
        gamma,
        options,
        terminals=(),
        current_func=None,
        probe_points=None,
        vector_potential_func=None,
        epsilon_func=None,
        screening=None,
    ):
        """``screening``: ``None`` or a dict ``{areas, sites, edge_centers, tolerance,
        max_iterations, step_size, step_drag}`` (the arrays of solver.py:306-314, i.e. areas
        already multiplied by the screening scale) -- switches on the self-consistent induced
        vector potential (solver.py:522-578, 654-688).

        ``vector_potential_func(t) -> A[m, 2]`` / ``epsilon_func(t) -> eps[n]`` (both already
        dimensionless) switch on the reference's time-dependent branches (solver.py:626-648)."""
        self.mesh = mesh
        self.vector_potential_func = vector_potential_func
        self.epsilon_func = epsilon_func
        self.screening = screening
        if screening is not None:
            d = screening["edge_centers"][:, None, :] - screening["sites"][None, :, :]
            # K[i, j] = area_j / |r_edge_i - r_site_j|   (tdgl/solver/screening.py:12-42)
            self._inv_r_area = screening["areas"][None, :] / np.sqrt((d**2).sum(axis=2))
            self.A_induced = np.zeros_like(np.asarray(link_exponents, dtype=float))
            self.last_screening_iterations = 0
        self.current_A = np.asarray(link_exponents, dtype=float)
        self.edge_unit = mesh.edge_mesh.directions / np.linalg.norm(
            mesh.edge_mesh.directions, axis=1)[:, None]
        self.u, self.gamma = u, gamma
        self.options = options
        self.terminals = [
            t if not isinstance(t, dict) else SimpleNamespace(**t) for t in terminals
        ]
        # the reference sorts terminals by length (tdgl/device/device.py:256)
        self.terminals.sort(key=lambda t: t.length)
        self.terminal_names = [t.name for t in self.terminals]
        if current_func is None:
            current_func = lambda t: {name: 0 for name in self.terminal_names}  # noqa: E731
        self.current_func = current_func
        self.terminal_current_densities = {name: 0 for name in self.terminal_names}
        self.probe_points = probe_points

        if self.terminals:
            fixed = np.concatenate(
                [np.asarray(t.site_indices) for t in self.terminals], dtype=np.int64
            )
        else:
            fixed = np.array([], dtype=np.int64)
        self.fixed_sites = fixed
        fix_psi = options.terminal_psi is not None
        self.operators = FVOperators(mesh, fixed_sites=fixed, fix_psi=fix_psi)
        self.operators.build_operators()
        self.operators.set_link_exponents(link_exponents)

        n = len(mesh.sites)
        self.num_edges = len(mesh.edge_mesh.edges)
        self.epsilon = np.asarray(epsilon, dtype=float) * np.ones(n)
        self.psi_init = np.ones(n, dtype=np.complex128)
        if fix_psi:
            self.psi_init[fixed] = options.terminal_psi
        self.mu_init = np.zeros(n)
        self.mu_boundary = np.zeros(len(mesh.edge_mesh.boundary_edge_indices))
        # adaptive-dt controller state (solver.py:316-320)
        self.d_psi_sq_vals = []
        self.tentative_dt = options.dt_init
        self.dt_max = options.dt_max if options.adaptive else options.dt_init

    # -- pieces of one step -----------------------------------------------------------
    def _set_terminal_bc(self, time):
        """solver.py:325-345: ``J_i = -(1/L_i) sum_{j != i} I_j`` on terminal ``i``'s
        boundary edges, rewritten only when it changes."""
        currents = self.current_func(time)
        for term in self.terminals:
            density = (-1 / term.length) * sum(
                currents.get(name, 0) for name in self.terminal_names if name != term.name
            )
            if density != self.terminal_current_densities[term.name]:
                self.terminal_current_densities[term.name] = density
                self.mu_boundary[term.boundary_edge_indices] = density

    def _euler_step(self, step, psi, abs_sq_psi, mu, dt):
        """solver.py:441-487: shrink dt by ``adaptive_time_step_multiplier`` until the
        update succeeds; at most ``max_solve_retries + 1`` shrinks, adaptive runs only."""
        opts = self.options
        lap = self.operators.psi_laplacian
        args = (psi, abs_sq_psi, mu, self.epsilon, self.gamma, self.u)
        result = psi_update(*args, dt, lap)
        for retries in itertools.count():
            if result is not None:
                break
            if not opts.adaptive or retries > opts.max_solve_retries:
                raise RuntimeError(
                    f"Solver failed to converge in {opts.max_solve_retries}"
                    f" retries at step {step} with dt = {dt:.2e}."
                    f" Try using a smaller dt_init."
                )
            dt = dt * opts.adaptive_time_step_multiplier
            result = psi_update(*args, dt, lap)
        return result[0], result[1], dt

    def _observables(self, psi, dA_dt=0.0):
        """solver.py:489-520: supercurrent, Poisson solve for mu, normal current."""
        ops = self.operators
        js = ops.get_supercurrent(psi)
        rhs = (ops.divergence @ (js - dA_dt)) - (ops.mu_boundary_laplacian @ self.mu_boundary)
        mu = ops.mu_laplacian_lu(rhs)
        jn = -(ops.mu_gradient @ mu) - dA_dt
        return mu, js, jn

    def _update_dynamic_inputs(self, time, dt):
        """solver.py:626-648: new A(t) -> dA/dt along the edges (with the PREVIOUS step's dt)
        and new link variables; new epsilon(t)."""
        dA_dt = 0.0
        if self.vector_potential_func is not None:
            new_A = np.asarray(self.vector_potential_func(time), dtype=float)
            dA_dt = np.einsum("ij, ij -> i", (new_A - self.current_A) / dt, self.edge_unit)
            if not np.allclose(new_A, self.current_A):
                self.operators.set_link_exponents(new_A)
            self.current_A = new_A
        if self.epsilon_func is not None:
            self.epsilon = np.asarray(self.epsilon_func(time), dtype=float)
        return dA_dt

    def _site_average(self, edge_field):
        """`Mesh.get_quantity_on_site` (tdgl/finite_volume/mesh.py:203-243): mean over the
        incident edges of F_e * e_hat, divided by 2."""
        edges = self.mesh.edge_mesh.edges
        n = len(self.mesh.sites)
        verts = np.concatenate([edges[:, 0], edges[:, 1]])
        counts = np.bincount(verts, minlength=n)
        out = np.empty((n, 2))
        for k in range(2):
            flux = edge_field * self.edge_unit[:, k]
            out[:, k] = np.bincount(verts, weights=np.concatenate([flux, flux]), minlength=n) / counts / 2
        return out

    def _update_with_screening(self, step, psi, mu, old_sq, dA_dt):
        """The screening loop of solver.py:654-688.  Faithful to the reference, including that
        psi / mu / dt carry over from one screening iteration to the next while |psi|^2 in the
        update formula stays that of the step's starting psi."""
        sc = self.screening
        alpha, beta = sc["step_size"], sc["step_drag"]
        A_ind, velocity = self.A_induced, 0.0
        error = np.inf
        dt = self.tentative_dt
        for it in itertools.count():
            if error < sc["tolerance"]:
                break
            if it > sc["max_iterations"]:
                raise RuntimeError(
                    f"Screening calculation failed to converge at step {step} after"
                    f" {sc['max_iterations']} iterations. Relative error in"
                    f" induced vector potential: {error:.2e}"
                    f" (tolerance: {sc['tolerance']:.2e})."
                )
            self.operators.set_link_exponents(self.current_A + A_ind)
            psi, new_sq, dt = self._euler_step(step, psi, old_sq, mu, dt)
            mu, js, jn = self._observables(psi, dA_dt)
            # solver.py:522-578: Polyak (heavy-ball) update of the induced vector potential
            new_A = self._inv_r_area @ self._site_average(js + jn)
            dA = new_A - A_ind
            velocity = (1 - beta) * velocity + alpha * dA
            A_ind = A_ind + velocity
            denom = np.maximum(np.linalg.norm(A_ind, axis=1), 1e-20)
            error = float(np.max(np.linalg.norm(dA, axis=1) / denom))
        self.A_induced = A_ind
        self.last_screening_iterations = it
        return psi, new_sq, dt, mu, js, jn

    # -- one step ---------------------------------------------------------------------
    def update(self, state, running_state, dt, *, psi, mu, **_unused):
        """solver.py:580-714 without screening.

        ``dt`` (the previous step's dt) only enters dA/dt; the step itself always runs with
        ``self.tentative_dt`` (solver.py:666-668).  Returns ``(dt, psi, mu, J_s, J_n)``.
        """
        opts = self.options
        step, time = state["step"], state["time"]
        self._set_terminal_bc(time)
        dA_dt = self._update_dynamic_inputs(time, dt)
        old_sq = np.absolute(psi) ** 2
        if self.screening is not None:
            psi, new_sq, dt, mu, js, jn = self._update_with_screening(step, psi, mu, old_sq, dA_dt)
        else:
            dt = self.tentative_dt
            psi, new_sq, dt = self._euler_step(step, psi, old_sq, mu, dt)
            mu, js, jn = self._observables(psi, dA_dt)
        if running_state is not None:
            running_state.append("dt", dt)
            if self.screening is not None:
                running_state.append("screening_iterations", self.last_screening_iterations)
            if self.probe_points is not None:
                running_state.append("mu", mu[self.probe_points])
                running_state.append("theta", np.angle(psi[self.probe_points]))
        if opts.adaptive:
            self.d_psi_sq_vals.append(float(np.absolute(new_sq - old_sq).max()))
            window = opts.adaptive_window
            if step > window:
                new_dt = opts.dt_init / max(1e-10, np.mean(self.d_psi_sq_vals[-window:]))
                self.tentative_dt = np.clip(0.5 * (new_dt + dt), 0, self.dt_max)
        return dt, psi, mu, js, jn


class StepLog:
    """Per-step scalar record (stands in for the reference's RunningState + DataHandler,
    runner.py:29-221, without HDF5)."""

    def __init__(self):
        self.rows = {}
        self._cur = {}

    def append(self, name, value):
        self._cur[name] = np.array(value, dtype=float, copy=True)

    def commit(self):
        for key, val in self._cur.items():
            self.rows.setdefault(key, []).append(val)
        self._cur = {}

    def array(self, name):
        return np.array(self.rows.get(name, []))


def run_time_loop(solver, options, psi=None, mu=None, on_save=None, max_steps=None):
    """The reference's time loop (runner.py:288-454) around ``solver.update``.

    Reproduced semantics:
    * optional thermalisation stage ``[0, skip_time]`` without saving, then time/step are
      reset to 0 but ``Runner.dt`` and the solver's controller state are NOT (runner.py:303-318);
    * each iteration saves (if ``i % save_every == 0``) BEFORE stepping, steps, and tests
      ``time >= end_time`` BEFORE advancing time, so one step past ``end_time`` is taken
      and the final state is saved with the pre-step time (runner.py:396-433, 452-453);
    * ``state["dt"]`` of iteration ``i`` is the dt used by step ``i-1`` (``dt_init`` at i=0).

    ``on_save(stage, i, time, dt, psi, mu, js, jn)`` is called where the reference calls
    ``DataHandler.save_time_step``.  ``max_steps`` bounds each stage (for benchmarking).
    Returns a dict with the final fields, the step log and loop bookkeeping.
    """
    n, m = len(solver.mesh.sites), solver.num_edges
    psi = solver.psi_init.copy() if psi is None else psi
    mu = solver.mu_init.copy() if mu is None else mu
    js, jn = np.zeros(m), np.zeros(m)
    log = StepLog()
    book = {"calls": 0, "saves": [], "stages": []}
    runner = SimpleNamespace(time=0.0, dt=options.dt_init)

    def stage(name, end_time, save):
        nonlocal psi, mu, js, jn
        i = 0
        for i in itertools.count():
            state = {"step": i, "time": runner.time, "dt": runner.dt}
            if i % options.save_every == 0 and save:
                book["saves"].append((name, i, runner.time, runner.dt))
                if on_save is not None:
                    on_save(name, i, runner.time, runner.dt, psi, mu, js, jn)
            new_dt, psi, mu, js, jn = solver.update(state, log, runner.dt, psi=psi, mu=mu)
            book["calls"] += 1
            if save:
                log.commit()
            else:
                log._cur = {}
            if runner.time >= end_time or (max_steps is not None and i + 1 >= max_steps):
                break
            runner.dt = new_dt
            runner.time += runner.dt
        if save and (i % options.save_every):
            book["saves"].append((name, i, runner.time, runner.dt))
            if on_save is not None:
                on_save(name, i, runner.time, runner.dt, psi, mu, js, jn)
        book["stages"].append((name, i + 1, runner.time))

    runner.time = 0.0
    if options.skip_time:
        stage("Thermalizing", options.skip_time, False)
    runner.time = 0.0
    stage("Simulating", options.solve_time, True)
    return dict(psi=psi, mu=mu, supercurrent=js, normal_current=jn, log=log, book=book)
