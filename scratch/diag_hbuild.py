import sys, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n = 1024
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, 10)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
ctx.set_planned_list(False)
b = pkg.make_batch(n, 10, "a1", seed=0xA1 + 2)
out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
print("wave 0: MFMA section mean %.0f max %.0f | inside the r loops mean %.0f | lower tiles mean %.1f" % (buf[:, 4].mean(), buf[:, 4].max(), buf[:, 5].mean(), buf[:, 6].mean()))
