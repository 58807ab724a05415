"""Solver options: same field names, defaults and validation messages as the reference's
``tdgl.SolverOptions`` (`tdgl/solver/options.py:19-166`), so scripts carry over unchanged.

Differences, all at the backend seam:

* there is exactly one compute path (HIP kernels on MI355X).  ``gpu`` and
  ``sparse_solver`` are accepted for source compatibility; ``gpu`` has no effect (there is no CPU
  path to fall back to).  The reference's solver names (``superlu`` ...) select the library's own
  choice of mu solve by mesh size: an explicit pseudo-inverse up to 5k sites, a substructured direct
  solve up to 150k sites (both exact to round-off, checked at set-up against ``min(1e-11, pcg_rtol)``
  and, in the time loop, by a residual check once per batch of steps: `Solution.stats`), the
  AMG-preconditioned CG above; ``sparse_solver="amg_pcg"`` forces the iterative solve at every size;
* ``pcg_rtol`` / ``pcg_max_iter`` / ``amg_smoothing_sweeps`` / ``pcg_precond_fp32`` control the
  iterative solve and are ignored by the direct ones;
* results are returned in memory (``tdgl_amd.solution.Solution``); ``output_file`` additionally
  writes them in the reference's HDF5 layout at the end of the run (`tdgl_amd/io.py`, needs
  h5py); ``monitor`` (the reference's live viewer) is accepted and ignored.
"""

from dataclasses import dataclass
from enum import Enum
from typing import Union


class SolverOptionsError(ValueError):
    pass


class SparseSolver(Enum):
    """Names accepted for ``SolverOptions.sparse_solver`` (reference: options.py:10-16) plus
    the native one.  The reference's names let the library pick the mu solve by mesh size (direct up
    to 150k sites, AMG-PCG above); ``AMG_PCG`` forces the iterative solve."""

    SUPERLU = "superlu"
    UMFPACK = "umfpack"
    PARDISO = "pardiso"
    CUPY = "cupy"
    AMG_PCG = "amg_pcg"


def precond_storage_mode(value) -> int:
    """Storage of the V-cycle's operators as the C ABI's code: 0 fp64, 1 fp32, 2 fp32 + binary16 on
    level 0.  Booleans of any flavour (``True``, ``numpy.bool_`` -- what an HDF5 attribute comes back
    as) mean 2 / 0; the integers 0, 1, 2 mean themselves.  Compared by value, never by identity."""
    import numpy as np

    if isinstance(value, (bool, np.bool_)):
        return 2 if bool(value) else 0
    mode = int(value)
    if mode not in (0, 1, 2):
        raise SolverOptionsError(f"pcg_precond_fp32 must be a bool or 0, 1, 2 (got {value!r}).")
    return mode


@dataclass
class SolverOptions:
    solve_time: float  # simulated time after thermalisation, in units of tau_0
    skip_time: float = 0.0  # thermalisation time simulated first, not recorded
    dt_init: float = 1e-6  # first (and, if not adaptive, only) time step
    dt_max: float = 1e-1  # cap of the adaptive time step
    adaptive: bool = True  # adapt dt to the rate of change of |psi|^2
    adaptive_window: int = 10  # number of recent steps averaged by the dt controller
    max_solve_retries: int = 10  # dt reductions allowed within one step
    adaptive_time_step_multiplier: float = 0.25  # dt factor per retry, in (0, 1)
    output_file: Union[str, None] = None  # HDF5 file in the reference's layout (needs h5py)
    terminal_psi: Union[float, complex, None] = 0.0  # psi pinned on terminal sites; None = free
    gpu: bool = False  # accepted, no effect: the HIP path is the only path
    sparse_solver: Union[SparseSolver, str] = SparseSolver.SUPERLU  # "amg_pcg" forces the iterative mu solve
    pause_on_interrupt: bool = True  # accepted, unused
    save_every: int = 100  # steps between saved snapshots
    progress_interval: int = 0  # accepted, unused
    monitor: bool = False  # accepted, unused (no live viewer)
    monitor_update_interval: float = 1.0  # accepted, unused
    field_units: str = "mT"  # units of applied fields
    current_units: str = "uA"  # units of terminal currents
    include_screening: bool = False
    max_iterations_per_step: int = 1000  # screening only
    screening_tolerance: float = 1e-3  # screening only
    screening_step_size: float = 0.1  # screening only
    screening_step_drag: float = 0.5  # screening only
    # --- native Poisson-solve controls (no reference counterpart) ---
    pcg_rtol: float = 1e-10  # stopping test of the iterative mu solve, ||b - A mu|| <= pcg_rtol ||b|| (sweep: profiles/EXPERIMENTS.md, round 4)
    pcg_max_iter: int = 500
    amg_smoothing_sweeps: int = 2  # Chebyshev degree of the AMG smoother
    # storage of the V-cycle's operators (arithmetic and the CG stay fp64): True = fp32 and, on level 0,
    # binary16; 1 = fp32 only; False = fp64
    pcg_precond_fp32: Union[bool, int] = True
    edge_currents_every_step: bool = True
    device_id: int = 0

    def validate(self) -> None:
        def fail(msg):
            raise SolverOptionsError(msg)

        if self.dt_init > self.dt_max:
            fail("dt_init must be less than or equal to dt_max.")
        tp = self.terminal_psi
        if tp is not None and not (0 <= abs(tp) <= 1):
            fail(f"terminal_psi must be None or have absolute value in [0, 1] (got {tp}).")
        mult = self.adaptive_time_step_multiplier
        if not (0 < mult < 1):
            fail(f"adaptive_time_step_multiplier must be in (0, 1) (got {mult}).")
        if not (0 < self.screening_step_drag <= 1):
            fail(f"screening_step_drag must be in (0, 1] (got {self.screening_step_drag}).")
        if self.screening_step_size <= 0:
            fail(f"screening_step_size must be in > 0 (got {self.screening_step_size}).")
        if self.screening_tolerance <= 0:
            fail(f"screening_tolerance must be in > 0 (got {self.screening_tolerance}).")
        solver = self.sparse_solver
        if isinstance(solver, str):
            try:
                solver = SparseSolver[solver.upper()]
            except KeyError:
                valid = list(SparseSolver.__members__.keys())
                fail(f"sparse solver must be one of {valid!r}, got {solver}.")
            self.sparse_solver = solver
        if not isinstance(self.sparse_solver, SparseSolver):
            fail(f"sparse solver must be a SparseSolver or str, got {self.sparse_solver!r}.")
        if self.include_screening and self.max_iterations_per_step < 1:
            fail("max_iterations_per_step must be >= 1.")
        if not (self.pcg_rtol > 0):
            fail(f"pcg_rtol must be > 0 (got {self.pcg_rtol}).")
        if self.pcg_max_iter < 1 or self.amg_smoothing_sweeps < 1:
            fail("pcg_max_iter and amg_smoothing_sweeps must be >= 1.")
        precond_storage_mode(self.pcg_precond_fp32)
        if self.adaptive_window < 0:
            fail(f"adaptive_window must be >= 0 (got {self.adaptive_window}).")
        if self.save_every < 1:
            fail(f"save_every must be >= 1 (got {self.save_every}).")
