"""Cost of a device-wide barrier inside one launch against a kernel boundary (tdgl_time_kernel ids 8-15):
python tools/bench_barrier.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "py-tdgl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from helpers import synthetic_mesh, uniform_field_A  # noqa: E402
from tdgl_amd import _lib  # noqa: E402
from tdgl_amd.hipcore import TDGLContext  # noqa: E402

mesh = synthetic_mesh(300)
ctx = TDGLContext(mesh)
ctx.set_link_exponents(uniform_field_A(mesh, 0.1))
ctx.set_epsilon(1.0)
ctx.set_state(np.ones(ctx.n, dtype=complex), np.zeros(ctx.n))
print("two dependent trivial kernels, one stream: %.2f us per kernel" % (ctx.time_kernel(9, 200) * 1e3 / 2))
names = {10: "grid.sync(), 32 workgroups", 11: "grid.sync(), 256 workgroups", 12: "counter + agent fences, 32 workgroups",
         13: "counter + agent fences, 256 workgroups", 14: "counter + agent fences, 32 workgroups on one XCD",
         15: "counter + L2-local fences, 32 workgroups on one XCD"}
for k, name in names.items():
    try:
        ms = ctx.time_kernel(k, 20)
        per = ms * 1e3 / (200 if k <= 11 else 100)
        msg = _lib.load().tdgl_last_error(ctx._ctx).decode()
        print(f"id {k} {name}: {per:.2f} us per barrier   [{msg}]", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"id {k} {name}: FAILED {exc}", flush=True)
gib = 2.0**30
print("HBM-cold streaming ceilings (1 GiB working set):")
ms = ctx.time_kernel(16, 20); print(f"  read-only: {gib / (ms * 1e-3) / 1e12:.2f} TB/s ({ms * 1e3:.1f} us per GiB)")
ms = ctx.time_kernel(17, 20); print(f"  copy (512 MiB -> 512 MiB): {gib / (ms * 1e-3) / 1e12:.2f} TB/s moved")
for k, name in ((18, "read-only, 8 loads per lane in flight"), (19, "read-only, 16 workgroups per CU"), (20, "read-only, one workgroup per 512 entries")):
    ms = ctx.time_kernel(k, 20); print(f"  {name}: {gib / (ms * 1e-3) / 1e12:.2f} TB/s")
ms = ctx.time_kernel(21, 10)
print(f"one workgroup (1024 threads, one CU) reading an L2-resident 2 MiB buffer: {50 * 2.0**21 / (ms * 1e-3) / 1e9:.0f} GB/s")
