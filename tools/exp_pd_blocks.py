"""Block sizes of the three-level factors used as the CG's preconditioner (0.65 - 1.3M sites): per configuration
(part size, super-block size, super-super-block size) the measured time of one application, its bytes, the iterations
and the residual of the set-up check (white noise from a zero guess).  `python tools/exp_pd_blocks.py [side]`"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from helpers import synthetic_mesh  # noqa: E402
from tdgl_amd.hipcore import TDGLContext  # noqa: E402

side = float(sys.argv[1]) if len(sys.argv) > 1 else 930.0
configs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [
    (160, 4096, 32768), (128, 4096, 32768), (192, 4096, 32768), (224, 6144, 49152), (160, 3072, 24576), (160, 6144, 49152),
    (128, 3072, 24576), (192, 6144, 65536)]
mesh = synthetic_mesh(side)
TDGLContext.AMG_CANDIDATES = 1
for block, sup, big in configs:
    TDGLContext.SUB2_BLOCK, TDGLContext.SUB2_SUPER, TDGLContext.SUB3_BIG = block, sup, big
    t0 = time.perf_counter()
    ctx = TDGLContext(mesh)
    ctx.build_poisson(rtol=1e-10)
    pd = ctx.precond_direct or dict(error=ctx.setup_times.get("substructure_error"))
    print(json.dumps(dict(block=block, super=sup, big=big, setup_s=round(time.perf_counter() - t0, 1),
                          host_s=round(ctx.setup_times.get("substructure_host", 0.0), 1), **pd)), flush=True)
    ctx.close()
