import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, 10)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
b = pkg.make_batch(256, 10, "a1", seed=0xA2)
ctx.set_lpt_schedule(False)
out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((256, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, 256)
for r in (9,):
    print("robot", r, "status", out["status"][r] & 0xff, "iters", out["status"][r] >> 8)
    for k in range(7):
        v = int(buf[r, 2 * k]); t = np.array([buf[r, 2 * k + 1]], np.int64).view(np.float64)[0]
        print("   it %d: kp %d tp %d q %d full %d have_z %d  t %.6e" % (k + 1, v >> 32, (v >> 24) & 0xff, (v >> 16) & 0xff, v & 1, (v >> 1) & 1, t))
