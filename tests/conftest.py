import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The libraries are build products (git-ignored): make sure they exist and are current before anything imports
    # them -- the meshes of nearly every test come from libtdgl_mesh.so.  (A no-op when they are up to date; on a
    # machine without the compilers the tests that need a library fail on their own.)
    try:
        import __graft_entry__ as entry

        entry.build()
    except Exception as exc:  # pragma: no cover
        print(f"conftest: could not build the libraries ({exc!r})", file=sys.stderr)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


_PRODUCT_DENSE_MAX_SITES = None
_PRODUCT_SUB_MAX_SITES = None
_PRODUCT_SUB2_MAX_SITES = None


@pytest.fixture(autouse=True, scope="session")
def _iterative_mu_solve_unless_asked():
    """The product solves the mu equation of meshes up to `TDGLContext.DENSE_MAX_SITES` sites with one
    dense matrix-vector product (`tdgl_poisson_set_dense_inverse`).  Most fixtures are that small, so
    the suites keep the AMG-PCG path under test by default (session-wide, so that module-scoped contexts
    see it too); tests that request the ``direct_solve`` fixture (tests/test_hip_direct.py re-runs the
    trajectory suite that way) get the product default back."""
    global _PRODUCT_DENSE_MAX_SITES, _PRODUCT_SUB_MAX_SITES, _PRODUCT_SUB2_MAX_SITES
    try:
        from tdgl_amd.hipcore import TDGLContext
    except Exception:  # (library not built: the tests that need it fail on their own)
        yield
        return
    _PRODUCT_DENSE_MAX_SITES, _PRODUCT_SUB_MAX_SITES = TDGLContext.DENSE_MAX_SITES, TDGLContext.SUB_MAX_SITES
    _PRODUCT_SUB2_MAX_SITES = TDGLContext.SUB2_MAX_SITES
    TDGLContext.DENSE_MAX_SITES = TDGLContext.SUB_MAX_SITES = TDGLContext.SUB2_MAX_SITES = 0
    yield
    TDGLContext.DENSE_MAX_SITES, TDGLContext.SUB_MAX_SITES = _PRODUCT_DENSE_MAX_SITES, _PRODUCT_SUB_MAX_SITES
    TDGLContext.SUB2_MAX_SITES = _PRODUCT_SUB2_MAX_SITES


@pytest.fixture
def direct_solve(monkeypatch):
    from tdgl_amd.hipcore import TDGLContext

    # the product defaults cover the reference's documented mesh sizes with a direct solve (one level, then two)
    assert _PRODUCT_DENSE_MAX_SITES >= 4000 and _PRODUCT_SUB_MAX_SITES >= 16000 and _PRODUCT_SUB2_MAX_SITES >= 60000
    monkeypatch.setattr(TDGLContext, "DENSE_MAX_SITES", _PRODUCT_DENSE_MAX_SITES)
    monkeypatch.setattr(TDGLContext, "SUB_MAX_SITES", _PRODUCT_SUB_MAX_SITES)
    monkeypatch.setattr(TDGLContext, "SUB2_MAX_SITES", _PRODUCT_SUB2_MAX_SITES)
    assert _PRODUCT_SUB2_MAX_SITES >= 250_000  # ... and BASELINE config 2
    return _PRODUCT_DENSE_MAX_SITES


@pytest.fixture
def substructured_solve(monkeypatch):
    """Every mesh from 200 sites up takes the substructured direct mu solve (parts of ~150 sites), which the
    product uses between `DENSE_MAX_SITES` and `SUB_MAX_SITES`."""
    from tdgl_amd.hipcore import TDGLContext

    monkeypatch.setattr(TDGLContext, "DENSE_MAX_SITES", 199)
    monkeypatch.setattr(TDGLContext, "SUB_MAX_SITES", 10 ** 9)
    monkeypatch.setattr(TDGLContext, "SUB_BLOCK", 150)
    return 150


@pytest.fixture(params=["symmetric_tiles", "whole_blocks"])
def two_level_solve(request, monkeypatch):
    """Every mesh from 200 sites up takes the TWO-level substructured direct mu solve (parts of ~60 sites inside
    super-blocks of ~500), which the product uses between `SUB_MAX_SITES` and `SUB2_MAX_SITES`.  Twice: with the G blocks
    of every level of small parts stored as the 16 x 16 tiles on or below their diagonal (what the product's first level
    is from a few tens of thousands of sites on; `TDGL_PD_SYM=2` extends it to the few parts of a test mesh) and with
    whole blocks everywhere (`TDGL_PD_SYM=0`).  Returns which."""
    from tdgl_amd.hipcore import TDGLContext

    monkeypatch.setenv("TDGL_PD_SYM", "2" if request.param == "symmetric_tiles" else "0")
    monkeypatch.setattr(TDGLContext, "DENSE_MAX_SITES", 199)
    monkeypatch.setattr(TDGLContext, "SUB_MAX_SITES", 199)
    monkeypatch.setattr(TDGLContext, "SUB2_MAX_SITES", 10 ** 9)
    monkeypatch.setattr(TDGLContext, "SUB2_BLOCK", 60)
    monkeypatch.setattr(TDGLContext, "SUB2_SUPER", 500)
    monkeypatch.setattr(TDGLContext, "SUB3_MIN_SITES", 10 ** 9)
    return request.param


@pytest.fixture
def three_level_solve(two_level_solve, monkeypatch):
    """... and from 200 sites up three levels (parts of ~60 sites, super-blocks of ~500, super-super-blocks of ~3,000),
    which the product uses from `SUB3_MIN_SITES` on."""
    from tdgl_amd.hipcore import TDGLContext

    monkeypatch.setattr(TDGLContext, "SUB3_MIN_SITES", 200)
    monkeypatch.setattr(TDGLContext, "SUB3_BIG", 3000)
    return two_level_solve


@pytest.fixture
def precond_direct_solve(three_level_solve, monkeypatch):
    """Every mesh from 200 sites up carries the three-level factors as the CG's PRECONDITIONER (fp32 storage, the
    context in reverse Cuthill-McKee order: `tdgl_poisson_set_substructure_precond`), which the product does between
    `SUB2_MAX_SITES` and `PD_MAX_SITES` (0.4 - 1.3 million sites) -- and every solve uses them (`PD_CHOICE` 1; the
    product lets the library choose per solve).  With tiles and with whole blocks like `two_level_solve`; returns which."""
    from tdgl_amd.hipcore import TDGLContext

    monkeypatch.setattr(TDGLContext, "SUB2_MAX_SITES", 199)
    monkeypatch.setattr(TDGLContext, "PD_MAX_SITES", 10 ** 9)
    monkeypatch.setattr(TDGLContext, "PD_CHOICE", 1)
    return three_level_solve
