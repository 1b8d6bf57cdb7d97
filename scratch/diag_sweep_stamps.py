"""Within-pivot stamps of the block sweep (build with QRGPU_EXTRA_FLAGS=-DQR_SWEEP_STAMPS): wave 0's ticks per pivot spent
[0] before the barrier (publishing), [1] in the barrier, [2] loading the panel / P^-1, [3] updating its blocks."""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
h = 10
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
n = 1024
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=4)
for b in seq: out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
nls = np.maximum(buf[:, 7] // 3, 1)
v = buf[:, 8:12].astype(np.float64)
print("per pivot (mean over robots): publish %.0f  barrier %.0f  load+Pinv %.0f  update %.0f  | total/pivot %.0f, nls mean %.1f" % (
    *(v / nls[:, None]).mean(0), (v.sum(1) / nls).mean(), nls.mean()))
d = np.diff(buf[:, :7], axis=1).astype(np.float64)
print("sweep phase mean %.0f, max %.0f" % (d[:, 2].mean(), d[:, 2].max()))
for lo, hi_ in ((0, 26), (26, 36), (36, 99)):
    m = (nls >= lo) & (nls < hi_)
    if m.any(): print("  nls [%d,%d): %d robots  per pivot: " % (lo, hi_, m.sum()), np.round((v[m] / nls[m, None]).mean(0)))
