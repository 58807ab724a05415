"""tdgl_amd: MI355X-native TDGL time-stepping core behind py-tdgl's solver API.

Public surface (mirrors `tdgl/__init__.py:1-23` of the reference for the solver path):
``Layer, Polygon, Device, SolverOptions, SolverOptionsError, SparseSolver, solve, TDGLSolver,
Solution`` and the ``geometry`` helpers.
"""

from . import geometry  # noqa: F401
from .device import Device, Layer, Polygon, TerminalInfo  # noqa: F401
from .finite_volume import EdgeMesh, Mesh  # noqa: F401
from .operators import MeshOperators  # noqa: F401
from .parameter import (  # noqa: F401
    CompositeParameter, Constant, ConstantField, LinearRamp, Parameter, PiecewiseLinear, Scale, SeparableEpsilon,
    TabulatedCurrents,
)
from .options import SolverOptions, SolverOptionsError, SparseSolver  # noqa: F401
from .solution import BiotSavartField, DynamicsData, Fluxoid, Solution, TDGLData  # noqa: F401
from .solver import SolverResult, TDGLSolver, solve  # noqa: F401

__version__ = "0.1.0"
