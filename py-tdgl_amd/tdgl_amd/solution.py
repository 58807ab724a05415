"""Result of a simulation.

The reference's ``Solution`` (`tdgl/solution/solution.py:59-1090`) wraps an HDF5 file and
offers post-processing/plotting; that layer is out of scope here.  This class carries what the
solver produced: the fields at every saved step (without ``SolverOptions.output_file``) or at the
last saved step (with it: the others were streamed to ``path`` in the reference's layout,
`tdgl_amd.io.DataHandler`), the per-step scalars (``dt``, probe ``mu``/``theta``) and the
configuration.
"""

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


@dataclass
class TDGLData:
    """Fields at one saved step (cf. `tdgl/solution/data.py:69-170`)."""

    step: int
    time: float
    dt: float
    psi: np.ndarray
    mu: np.ndarray
    supercurrent: np.ndarray
    normal_current: np.ndarray
    applied_vector_potential: Optional[np.ndarray] = None
    epsilon: Optional[np.ndarray] = None
    induced_vector_potential: Optional[np.ndarray] = None  # include_screening only


@dataclass
class DynamicsData:
    """Per-step scalars of the saved stage (cf. `tdgl/solution/data.py:173-330`)."""

    dt: np.ndarray
    time: np.ndarray
    mu: Optional[np.ndarray] = None      # [n_probe, n_steps]
    theta: Optional[np.ndarray] = None   # [n_probe, n_steps]
    pcg_iterations: Optional[np.ndarray] = None
    screening_iterations: Optional[np.ndarray] = None  # include_screening only

    def voltage(self, i: int = 0, j: int = 1) -> np.ndarray:
        """mu_i - mu_j between two probe points, per step."""
        return self.mu[i] - self.mu[j]

    def phase_difference(self, i: int = 0, j: int = 1) -> np.ndarray:
        return np.unwrap(self.theta[i] - self.theta[j])


@dataclass
class Solution:
    device: object
    options: object
    saved_steps: List[TDGLData] = field(default_factory=list)
    dynamics: Optional[DynamicsData] = None
    applied_vector_potential: object = None
    terminal_currents: object = None
    disorder_epsilon: object = None
    total_seconds: float = 0.0
    stats: Dict[str, float] = field(default_factory=dict)
    solve_step: int = -1
    dynamic_vector_potential: bool = False
    dynamic_epsilon: bool = False
    path: Optional[str] = None                 # the streamed HDF5 file (SolverOptions.output_file)
    saved_step_index: Optional[list] = None    # streaming: (step, time) of every group data/<k> on disk

    def to_hdf5(self, file) -> None:
        """Write the saved steps in the reference's DataHandler layout (`tdgl_amd.io`); ``file``
        is a path (needs h5py) or an open h5py-like group."""
        from .io import write_solution_h5

        write_solution_h5(self, file, self.dynamic_vector_potential, self.dynamic_epsilon)

    @property
    def tdgl_data(self) -> TDGLData:
        """The saved step selected by ``solve_step`` (default: the last one)."""
        return self.saved_steps[self.solve_step]

    @property
    def times(self) -> np.ndarray:
        return np.array([s.time for s in self.saved_steps])

    # -- sheet current densities on the sites, in current_units / length_units -------------------
    def _k0(self) -> float:
        """K0 in ``current_units / length_units`` (`tdgl/solution/solution.py:192`)."""
        from .device import CURRENT_UNITS, LENGTH_UNITS

        dev = self.device
        return dev.K0 * LENGTH_UNITS[dev.length_units] / CURRENT_UNITS[self.options.current_units]

    def _site_vector(self, quantity_on_edges) -> np.ndarray:
        # magnitude * unit direction of the edge->site average (`tdgl/solution/data.py:48-65`)
        return self.device.mesh.get_quantity_on_site(quantity_on_edges)

    @property
    def supercurrent_density(self) -> np.ndarray:
        """K_s on the sites, shape (n, 2) (`tdgl/solution/solution.py:186-194`)."""
        return self._k0() * self._site_vector(self.tdgl_data.supercurrent)

    @property
    def normal_current_density(self) -> np.ndarray:
        return self._k0() * self._site_vector(self.tdgl_data.normal_current)

    @property
    def current_density(self) -> np.ndarray:
        """Total sheet current density K = K_s + K_n (`tdgl/solution/solution.py:230-237`)."""
        return self.supercurrent_density + self.normal_current_density

    def current_through_cut(self, x0: float, physical: bool = False) -> float:
        """Total sheet current crossing the vertical line x = x0, summed over the mesh edges that
        cross it: ``sum_e (J_s + J_n)_e s_e sign`` with ``s_e`` the Voronoi dual length.  Discretely
        conserved.  By default dimensionless with ``x0`` in units of
        xi; ``physical=True``: ``x0`` in ``length_units``, result in ``options.current_units``."""
        if physical:
            xi = self.device.coherence_length
            scale = self.device.current_scale(self.options.current_units)
            return self.current_through_cut(x0 / xi) / scale * xi
        mesh = self.device.mesh
        em = mesh.edge_mesh
        d = self.tdgl_data
        xa = mesh.sites[em.edges[:, 0], 0]
        xb = mesh.sites[em.edges[:, 1], 0]
        crossing = (xa < x0) != (xb < x0)
        sign = np.where(xa < x0, 1.0, -1.0)
        j = (d.supercurrent + d.normal_current) * em.dual_edge_lengths * sign
        return float(j[crossing].sum())
