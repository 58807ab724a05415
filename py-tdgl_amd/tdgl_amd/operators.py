"""``MeshOperators``: the finite-volume operators of the reference
(`tdgl/finite_volume/operators.py:233-394`) held as device-resident data behind the C ABI.

The reference keeps SciPy sparse matrices (``psi_laplacian``, ``psi_gradient``,
``divergence``, ``mu_laplacian`` + its LU, ``mu_boundary_laplacian``, ``mu_gradient``).
Here the same attributes are light handles whose ``@`` launches the corresponding HIP
kernel, so code written against the reference seam (``ops.psi_laplacian @ psi``,
``ops.get_supercurrent(psi)``, ``ops.mu_laplacian_lu(rhs)``) keeps working.
"""

from typing import Union

import numpy as np

from .finite_volume import Mesh
from .hipcore import TDGLContext


class _DeviceOperator:
    """``op @ x`` -> one kernel launch through the C ABI."""

    def __init__(self, name, shape, apply):
        self.name, self.shape, self._apply = name, shape, apply

    def __matmul__(self, x):
        x = np.asarray(x)
        if x.ndim != 1 or x.shape[0] != self.shape[1]:
            raise ValueError(f"dimension mismatch: {self.name} is {self.shape[0]}x{self.shape[1]}, operand {x.shape}")
        return self._apply(x)

    def __repr__(self):
        return f"<HIP operator {self.name} {self.shape[0]}x{self.shape[1]}>"


class MeshOperators:
    """Finite-volume operators for a mesh, resident on the MI355X.

    Args (as in the reference, operators.py:245-252): ``mesh``, ``sparse_solver`` (the reference's names select
    nothing here: the mu solve is chosen by mesh size -- explicit pseudo-inverse, one to three levels of nested
    dissection, CG with the AMG V-cycle and the fp32-stored factors as its two preconditioners, AMG-PCG alone;
    `hipcore.TDGLContext.build_poisson` -- and ``"amg_pcg"`` forces the last of these at every size), ``use_cupy``
    (ignored), ``fixed_sites``, ``fix_psi``.  Extra keyword arguments configure the device and the Poisson solve.
    """

    def __init__(
        self,
        mesh: Mesh,
        sparse_solver=None,
        use_cupy: bool = False,
        fixed_sites: Union[np.ndarray, None] = None,
        fix_psi: bool = True,
        *,
        u: float = 5.79,
        gamma: float = 10.0,
        device_id: int = 0,
        pcg_rtol: float = 1e-10,
        pcg_max_iter: int = 500,
        amg_smoothing_sweeps: int = 2,
        edge_currents_every_step: bool = True,
        reorder="rcm",
        precond_fp32: bool = True,
    ):
        self.mesh = mesh
        self.areas = mesh.areas
        self.edges = mesh.edge_mesh.edges
        self.edge_directions = mesh.edge_mesh.directions
        self.sparse_solver = sparse_solver
        self.fixed_sites = fixed_sites
        self.fix_psi = fix_psi
        self.link_exponents = None
        self._opts = dict(
            u=u, gamma=gamma, device_id=device_id, pcg_rtol=pcg_rtol, pcg_max_iter=pcg_max_iter,
            nu=amg_smoothing_sweeps, edge_currents_every_step=edge_currents_every_step,
            reorder=reorder, precond_fp32=precond_fp32,
        )
        self.ctx: Union[TDGLContext, None] = None
        self.hierarchy = None
        self.psi_laplacian = self.psi_gradient = None
        self.divergence = self.mu_laplacian = self.mu_boundary_laplacian = self.mu_gradient = None
        self.mu_laplacian_lu = None

    def build_operators(self) -> None:
        """Upload the mesh, build the SELL site graph and set up the AMG hierarchy for the mu
        solve (the counterpart of build_operators + LU factorisation, operators.py:282-308)."""
        o = self._opts
        from .options import SparseSolver

        forced_iterative = self.sparse_solver in (SparseSolver.AMG_PCG, "amg_pcg", "AMG_PCG")
        self.ctx = TDGLContext(
            self.mesh, fixed_sites=self.fixed_sites, fix_psi=self.fix_psi, u=o["u"],
            gamma=o["gamma"], device_id=o["device_id"], reorder=o["reorder"], direct_solve=not forced_iterative,
        )
        self.hierarchy = self.ctx.build_poisson(
            rtol=o["pcg_rtol"], max_iter=o["pcg_max_iter"], nu=o["nu"],
            edge_currents_every_step=o["edge_currents_every_step"],
        )
        from .options import precond_storage_mode

        mode = precond_storage_mode(o["precond_fp32"])
        if mode != self.ctx.poisson_options["precond_fp32"]:  # (the default is 2: fp32 + binary16 on level 0)
            po = dict(self.ctx.poisson_options)
            po["smoother"] = "jacobi" if po["smoother"] == 0 else "chebyshev"
            po["precond_fp32"] = mode
            self.ctx.set_poisson_options(**po)
        n, m = self.ctx.n, self.ctx.m
        ctx = self.ctx
        self.mu_gradient = _DeviceOperator("mu_gradient", (m, n), lambda mu: -ctx.normal_current(mu))
        self.divergence = _DeviceOperator("divergence", (n, m), ctx.apply_divergence)
        self.mu_laplacian = _DeviceOperator("mu_laplacian", (n, n), ctx.apply_mu_laplacian)
        self.mu_boundary_laplacian = _DeviceOperator(
            "mu_boundary_laplacian", (n, ctx.n_boundary), ctx.apply_mu_boundary_laplacian)
        self.mu_laplacian_lu = lambda rhs: ctx.poisson_solve(rhs)[0]

    def set_link_exponents(self, link_exponents: np.ndarray) -> None:
        """operators.py:310-383: recompute the link variables and the covariant operators."""
        if self.ctx is None:
            raise RuntimeError("build_operators() must be called before set_link_exponents().")
        self.link_exponents = np.asarray(link_exponents)
        self.ctx.set_link_exponents(self.link_exponents)
        n = self.ctx.n
        ctx = self.ctx
        self.psi_laplacian = _DeviceOperator("psi_laplacian", (n, n), ctx.apply_psi_laplacian)
        self.psi_gradient = _DeviceOperator("psi_gradient", (ctx.m, n), ctx.apply_psi_gradient)

    def get_supercurrent(self, psi: np.ndarray):
        """operators.py:385-394."""
        return self.ctx.supercurrent(psi)
