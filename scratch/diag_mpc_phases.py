import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, 10)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = pkg.make_batch(n, 10, "a1", seed=0xA3)
out = G.run_mpc(ctx, pkg, b)
out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
d = np.diff(buf[:, :7], axis=1).astype(np.float64)
names = ["load+srbd", "H/g build", "sweep inv", "x0", "GI", "out"]
it = out["status"] >> 8
print("clock64 ticks per phase (mean / p50 / max):")
for k, nm in enumerate(names):
    print("  %-10s mean %9.0f  p50 %9.0f  max %9.0f" % (nm, d[:, k].mean(), np.median(d[:, k]), d[:, k].max()))
tot = (buf[:, 6] - buf[:, 0])
print("total per WG mean %.0f max %.0f ; GI ticks per iteration mean %.0f ; iters mean %.1f max %d; ns mean %.0f" % (tot.mean(), tot.max(), (d[:, 4] / np.maximum(it, 1)).mean(), it.mean(), it.max(), buf[:, 7].mean()))
print("span first start -> last end: %.0f ticks" % (buf[:, 6].max() - buf[:, 0].min()))

print("sweep stamps per pivot (panel write, barrier, P^-1, block updates): mean", (buf[:, 8:13] / np.maximum(buf[:, 7:8] // 3, 1)).mean(0).round(0), " nls=40 robots:", (buf[buf[:, 7] == 120, 8:13] / 40.0).mean(0).round(0))
sub = buf[:, 8:14].astype(np.float64)
nm2 = ["update,bookkeeping,scan (0)", "w,delta,d,publish (1)", "X1 wait (2)", "r partial + B2 (3)", "r,dr,t1,t2,flags (4)", "B3 wait (5)"]
for k, nm in enumerate(nm2):
    print("  GI %-12s per-iter mean %8.0f   share %.2f" % (nm, (sub[:, k] / np.maximum(it, 1)).mean(), sub[:, k].sum() / d[:, 4].sum()))
print("final q mean %.1f max %d" % (buf[:, 14].mean(), buf[:, 14].max()))
j = int(np.argmax(it))
print("worst robot %d: iters %d, final q %d, nls %d, GI ticks %d (%.0f / iter), sweep %d" % (j, it[j], buf[j, 14], buf[j, 7] // 3, d[j, 4], d[j, 4] / it[j], d[j, 2]))
for k, nm in enumerate(nm2):
    print("  worst GI %-28s per-iter %8.0f" % (nm, sub[j, k] / it[j]))
order = np.argsort(-tot)[:8]
print("top-8 totals:", [(int(o), int(tot[o]), int(it[o]), int(buf[o, 7] // 3)) for o in order])
