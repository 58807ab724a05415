"""Symmetric tiles on the way down (k_sub_down_sym) against whole blocks: the set-up check of the factors as
preconditioner (white noise from a zero guess) under TDGL_PD_SYM = 0 / 1 / 2.  `python tools/diag_sym.py [side]`"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

if len(sys.argv) > 2 and sys.argv[2] == "child":
    from helpers import synthetic_mesh
    from tdgl_amd.hipcore import TDGLContext

    mesh = synthetic_mesh(float(sys.argv[1]))
    TDGLContext.AMG_CANDIDATES = 1
    TDGLContext.SUB_MAX_SITES, TDGLContext.SUB2_MAX_SITES = 1000, 2000
    ctx = TDGLContext(mesh)
    ctx.build_poisson(rtol=1e-10)
    print(json.dumps(dict(sym=os.environ.get("TDGL_PD_SYM"), n=ctx.n, pd=ctx.precond_direct, error=ctx.setup_times.get("substructure_error"))), flush=True)
    ctx.close()
else:
    side = sys.argv[1] if len(sys.argv) > 1 else "300"
    for v in ("0", "1", "2"):
        env = dict(os.environ, TDGL_PD_SYM=v)
        subprocess.run([sys.executable, os.path.abspath(__file__), side, "child"], env=env, timeout=600)
