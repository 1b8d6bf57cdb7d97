"""Cycles of the block rebuild and of the restore rounds of a warm start (build with -DQR_DIAG_REFAC)."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
h, n = 10, 1024
G.setup_a1(ctx, pkg, h)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=4)
for b in seq:
    out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
q = buf[:, 11]; reb = buf[:, 10]; rst = buf[:, 9]; drops = buf[:, 13]
m = q > 0
print("robots with a rebuild", m.sum(), "| q mean %.1f max %d" % (q[m].mean(), q[m].max()))
print("rebuild cycles mean %.0f max %.0f | per row %.0f" % (reb[m].mean(), reb[m].max(), (reb[m] / q[m]).mean()))
print("restore cycles mean %.0f max %.0f" % (rst[m].mean(), rst[m].max()))
for lo, hi in ((1, 16), (16, 32), (32, 48), (48, 65)):
    k = m & (q >= lo) & (q < hi)
    if k.any(): print("  q in [%d,%d): %d robots, rebuild mean %.0f, restore mean %.0f" % (lo, hi, k.sum(), reb[k].mean(), rst[k].mean()))
print("S build mean %.0f | sweep mean %.0f (per row %.0f) | rest (negate + W fill + barriers) mean %.0f" % (buf[m, 2].mean(), buf[m, 3].mean(), (buf[m, 3] / q[m]).mean(), (reb[m] - buf[m, 2] - buf[m, 3]).mean()))
