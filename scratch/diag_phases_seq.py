"""Phase stamps of the MPC kernel over a temporally coherent sequence (warm start on): clock64 ticks per phase and per robot class."""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
h = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0xA1 + 2
seq = pkg.make_batch_sequence(n, h, "a1", seed=seed, steps=5)
names = ["load+srbd", "H/g build", "sweep inv", "x0", "active set", "out"]
for warm in (False, True):
    ctx.set_warm_start(warm)
    for b in seq:
        out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64)
    lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    d = np.diff(buf[:, :7], axis=1).astype(np.float64)
    it = (out["status"] >> 8) & 0xffff
    nls = buf[:, 7] // 3
    tot = buf[:, 6] - buf[:, 0]
    print("warm", warm, ": phase ticks mean / max;  flagged", int(((out["status"] & 0xff) != 0).sum()))
    for k, nm in enumerate(names):
        print("  %-10s mean %9.0f  max %9.0f" % (nm, d[:, k].mean(), d[:, k].max()))
    print("  total mean %.0f max %.0f | iters mean %.1f max %d | span %.0f" % (tot.mean(), tot.max(), it.mean(), it.max(), buf[:, 6].max() - buf[:, 0].min()))
    for lo, hi_ in ((0, 26), (26, 36), (36, 99)):
        m = (nls >= lo) & (nls < hi_)
        if m.any():
            print("  nls in [%d, %d): %4d robots, total mean %.0f max %.0f, sweep mean %.0f, active set mean %.0f max %.0f, iters mean %.1f" % (
                lo, hi_, m.sum(), tot[m].mean(), tot[m].max(), d[m, 2].mean(), d[m, 4].mean(), d[m, 4].max(), it[m].mean()))
    top = np.argsort(-tot)[:6]
    print("  top totals (robot, ticks, iters, nls, final q):", [(int(o), int(tot[o]), int(it[o]), int(nls[o]), int(buf[o, 14])) for o in top])
