"""Phase stamps of the h = 16 kernel on the A1 half of the mixed bench batch (instrumented kernels)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h, n = 16, 1024
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=4)
for b in seq:
    out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
d = np.diff(buf[:, :7], axis=1).astype(np.float64)
names = ["load+srbd", "H/g build", "sweep inv", "x0", "GI", "out"]
it = G.iterations(out["status"])
for k, nm in enumerate(names):
    print("  %-10s mean %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
tot = (buf[:, 6] - buf[:, 0])
print("total mean %.0f p90 %.0f max %.0f ; iters mean %.1f max %d; nls mean %.1f max %d; final q mean %.1f max %d" % (tot.mean(), np.percentile(tot, 90), tot.max(), it.mean(), it.max(), (buf[:, 7] // 3).mean(), (buf[:, 7] // 3).max(), buf[:, 14].mean(), buf[:, 14].max()))
t0, t6 = buf[:, 12].astype(np.float64), buf[:, 13].astype(np.float64)
span = t6.max() - t0.min()
print("span %.1f us, mean solve %.1f us, max %.1f us, slots busy on average %.0f" % (span / 100, (t6 - t0).mean() / 100, (t6 - t0).max() / 100, (t6 - t0).sum() / span))
order = np.argsort(-tot)[:10]
print("top-10: (total k-cycles, GI k-cycles, iters, nls, q)", [(int(tot[o] // 1000), int(d[o, 4] // 1000), int(it[o]), int(buf[o, 7] // 3), int(buf[o, 14])) for o in order])
