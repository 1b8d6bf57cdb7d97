"""The slowest robot of draw 1 (all-stance, ~60 active rows): where its active-set cycles go.  Build with -DQR_DIAG_REFAC."""
import sys, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n, h = 1024, 10
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
ctx.set_planned_list(False)
seq = pkg.make_batch_sequence(n, h, "a1", seed=1163, steps=6)
prev = np.zeros((n, 16), np.int64)
for k, b in enumerate(seq):
    out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    it = (out["status"] >> 8) & 0xffff
    nls = (b["gait"] > 0).sum(1)
    cand = np.where(nls == 40)[0]
    j = cand[np.argmax(buf[cand, 10] + buf[cand, 9])]
    print("step %d: robot %d nls %d iters %d status %d | rebuild q %d: %d cycles (S build %d, sweep of S %d) | restore %d cycles, drops %d | final q %d" % (
        k, j, nls[j], it[j], out["status"][j] & 0xff, buf[j, 11], buf[j, 10], buf[j, 2], buf[j, 3], buf[j, 9], buf[j, 13] - prev[j, 13], buf[j, 14]))
    prev = buf
