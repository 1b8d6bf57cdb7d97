"""Where a working-set change spends its cycles as the set grows (build with -DQR_GI_STAMPS): wave 0's stamps per change, by final q."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
h = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 1024
G.setup_a1(ctx, pkg, h)
ctx.set_warm_start(False)                      # cold: every change is a one-row step, the set grows from 0 to its final size
b = pkg.make_batch(n, h, "a1", seed=0xA3)
out = G.run_mpc(ctx, pkg, b); out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
it = ((out["status"] >> 8) & 0xffff).astype(np.float64)
qf = buf[:, 14]
cs = buf[:, 8:14].astype(np.float64)
names = ["scan+pick (0)", "w, delta, d (1)", "barrier X1 (2)", "r partial + B2 (3)", "r, steps, flags (4)", "barrier B3 (5)"]
for lo, hi in ((1, 12), (12, 24), (24, 36), (36, 48), (48, 65)):
    m = (qf >= lo) & (qf < hi) & (it > 0)
    if not m.any(): continue
    per = (cs[m] / it[m, None]).mean(0)
    print("final q in [%d,%d): %4d robots, iterations mean %.1f | per change: %s | sum %.0f" % (lo, hi, m.sum(), it[m].mean(), " ".join("%s %.0f" % (nm.split(" (")[0], v) for nm, v in zip(names, per)), per.sum()))
