"""The hard all-stance robot of draw 1 inside the planned (eight-wave, whole-CU) launch: phase stamps."""
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
seq = pkg.make_batch_sequence(n, h, "a1", seed=1163, steps=8)
names = ["load+srbd", "H/g build", "sweep", "x0", "active set", "out"]
for k, b in enumerate(seq):
    out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    it = (out["status"] >> 8) & 0xffff
    j = 1023
    d = np.diff(buf[j, :7])
    wall = (buf[:, 13] - buf[:, 12]) / 100.0
    t0 = buf[:, 12]; t1 = buf[:, 13]
    last = np.argsort(-t1)[:3]
    print("        starts %.0f us after the launch's first workgroup; last finishers (robot, start, length us): %s" % ((t0[j] - t0.min()) / 100.0, [(int(r), round((t0[r] - t0.min()) / 100.0), round((t1[r] - t0[r]) / 100.0)) for r in last]))
    print("step %d: robot %d iterations %d final q %d | %s | total %d cycles, %.0f us by the wall clock (launch span %.0f us)" % (
        k, j, it[j], buf[j, 14], " ".join("%s %d" % (nm, v) for nm, v in zip(names, d)), buf[j, 6] - buf[j, 0], wall[j], (buf[:, 13].max() - buf[:, 12].min()) / 100.0))
