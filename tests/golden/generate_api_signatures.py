"""Record the call signatures of the reference's solver-path API as data (`api_signatures.json`).

Run by hand in the build container (where `/root/reference` exists):
    python tests/golden/generate_api_signatures.py
The fixture holds, per callable, the parameter names, kinds and (repr of) defaults -- what a caller
can see -- and, for classes, the public method / property names.  `tests/test_host_logic.py` holds
the package against it, so that a drift of the drop-in surface is caught without the reference.
"""

import inspect
import json
import os
import sys

sys.path.insert(0, os.path.dirname(__file__))
from _reference_shim import import_reference  # noqa: E402


def signature(func):
    out = []
    for p in inspect.signature(func).parameters.values():
        if p.name == "self":
            continue
        default = None if p.default is inspect.Parameter.empty else repr(p.default)
        out.append([p.name, p.kind.name, default])
    return out


def public_names(cls):
    return sorted(n for n in dir(cls) if not n.startswith("_"))


def main():
    tdgl = import_reference()
    from tdgl.finite_volume.operators import MeshOperators
    from tdgl.parameter import CompositeParameter, Parameter
    from tdgl.solution.data import DynamicsData, TDGLData
    from tdgl.solver.options import SolverOptions
    from tdgl.solver.solver import TDGLSolver

    import dataclasses

    data = {
        "version": tdgl.__version__,
        "signatures": {
            "solve": signature(tdgl.solve),
            "TDGLSolver.__init__": signature(TDGLSolver.__init__),
            "TDGLSolver.update": signature(TDGLSolver.update),
            "TDGLSolver.solve": signature(TDGLSolver.solve),
            "TDGLSolver.update_mu_boundary": signature(TDGLSolver.update_mu_boundary),
            "MeshOperators.__init__": signature(MeshOperators.__init__),
            "MeshOperators.set_link_exponents": signature(MeshOperators.set_link_exponents),
            "MeshOperators.get_supercurrent": signature(MeshOperators.get_supercurrent),
            "Parameter.__init__": signature(Parameter.__init__),
            "Parameter.__call__": signature(Parameter.__call__),
            "CompositeParameter.__init__": signature(CompositeParameter.__init__),
            "DynamicsData.mean_voltage": signature(DynamicsData.mean_voltage),
            "DynamicsData.voltage": signature(DynamicsData.voltage),
            "DynamicsData.from_hdf5": signature(DynamicsData.from_hdf5),
            "TDGLData.from_hdf5": signature(TDGLData.from_hdf5),
        },
        "SolverOptions.fields": [[f.name, "MISSING" if f.default is dataclasses.MISSING else repr(f.default)]
                                 for f in dataclasses.fields(SolverOptions)],
        "SolverResult.fields": list(tdgl.solver.solver.SolverResult._fields),
        "MeshOperators.attributes": ["psi_laplacian", "psi_gradient", "divergence", "mu_laplacian", "mu_laplacian_lu",
                                     "mu_boundary_laplacian", "mu_gradient"],
    }
    path = os.path.join(os.path.dirname(__file__), "api_signatures.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
