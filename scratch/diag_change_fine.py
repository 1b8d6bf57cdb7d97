"""The stretch of a working-set change between B3 and the next row's pick, in parts (build with -DQR_GI_STAMPS -DQR_GI_STAMPS_FINE)."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
h = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G.setup_a1(ctx, pkg, h)
ctx.set_warm_start(False)
b = pkg.make_batch(n, h, "a1", seed=0xA3)
out = G.run_mpc(ctx, pkg, b); out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
it = ((out["status"] >> 8) & 0xffff).astype(np.float64)
cs = buf[:, 8:14].astype(np.float64)
names = ["z, x, u", "bookkeeping to the loop top", "slacks + select", "wave_min_d", "checks + pick", "rest of the change"]
m = it > 4
per = (cs[m] / it[m, None]).mean(0)
print("n %d h %d, %d robots, changes mean %.1f | per change: %s | sum %.0f" % (n, h, m.sum(), it[m].mean(), " | ".join("%s %.0f" % (nm, v) for nm, v in zip(names, per)), per.sum()))
