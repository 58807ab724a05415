"""Dev benchmark: price of cross-stream dependencies (tdgl_time_kernel ids 8 and 9)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
import numpy as np
from helpers import synthetic_mesh, uniform_field_A
from tdgl_amd.hipcore import TDGLContext
mesh = synthetic_mesh(30)
ctx = TDGLContext(mesh)
ctx.build_poisson()
ctx.set_link_exponents(uniform_field_A(mesh, 0.1)); ctx.set_epsilon(np.ones(ctx.n)); ctx.set_state(np.ones(ctx.n, dtype=complex), np.zeros(ctx.n))
for rep in range(3):
    a, b = ctx.time_kernel(8, 2000), ctx.time_kernel(9, 2000)
    print(f"two kernels across streams {1e3*a:.2f} us, in one stream {1e3*b:.2f} us, two cross-stream dependencies cost {1e3*(a-b):.2f} us")
