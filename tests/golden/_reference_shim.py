"""Import the upstream reference (`/root/reference`, py-tdgl v0.8.3) in the build container.

Only used by `generate_golden.py` (run by hand in the build container, where
`/root/reference` exists) to produce the committed `.npz` fixtures.  Nothing in the
test-suite, `bench.py` or `__graft_entry__.py` imports this module at run time: the
reference's Python does not travel to the GPU box.

The reference imports h5py / numba / meshpy / shapely / pint / IPython at module scope.
None of them is installed here and none of them is touched by the solver hot path
(`TDGLSolver.update` and `MeshOperators`), so they are replaced by inert stand-in
modules before `import tdgl` (recipe: SURVEY.md Appendix B).
"""

import sys
from unittest import mock

REFERENCE_ROOT = "/root/reference"

_ABSENT = [
    "h5py",
    "numba",
    "meshpy",
    "meshpy.triangle",
    "shapely",
    "shapely.geometry",
    "shapely.geometry.polygon",
    "shapely.ops",
    "shapely.affinity",
    "shapely.errors",
    "shapely.validation",
    "pint",
    "IPython",
    "IPython.display",
]


def _passthrough_jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


def import_reference():
    """Returns the imported reference package (``tdgl``)."""
    for name in _ABSENT:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    sys.modules["numba"].njit = _passthrough_jit
    sys.modules["numba"].prange = range
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import tdgl  # noqa: E402

    return tdgl
