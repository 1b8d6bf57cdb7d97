"""Who are the slow robots at h = 16 (build with -DQR_DIAG_REFAC)?  Per robot: iterations, the warm guess's size, restore drops, final q."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
h = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1024
G.setup_a1(ctx, pkg, h)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=5)
for b in seq:
    out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
it = (out["status"] >> 8) & 0xffff
qg = buf[:, 11]; reb = buf[:, 8]; drops = buf[:, 13]; qf = buf[:, 14]; nls = buf[:, 7] // 3
order = np.argsort(-it)[:14]
print("h", h, "iterations mean %.1f max %d; guess size mean %.1f; restore drops mean %.1f max %d" % (it.mean(), it.max(), qg.mean(), drops.mean(), drops.max()))
print("robot iters guess_q rebuilds drops final_q nls  rebuild_cycles restore_cycles")
for r in order:
    print(r, it[r], qg[r], reb[r], drops[r], qf[r], nls[r], buf[r, 10], buf[r, 9])
big = it > 60
print("robots over 60 iterations:", big.sum(), "| of them with a failed rebuild (>=100):", (reb[big] >= 100).sum(), "| drops mean %.1f" % drops[big].mean() if big.any() else "")
