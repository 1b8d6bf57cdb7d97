"""What co-residency costs a robot: mean phase cycles of the main MPC pass when a launch puts one workgroup on a CU (256 robots), two (512,
every slot taken once) or two rounds of two (1024).  Same generator, warm steps, in-kernel stamps."""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h = 10
for n in (128, 256, 512, 1024):
    ctx = pkg.Context(0, 4096, 16)
    G.setup_a1(ctx, pkg, h)
    lib = ctx._lib
    lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.qrgpu_debug_cycles(ctx._h, None, 0)
    seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA3, steps=6)
    for b in seq:
        out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64)
    lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    t0, t6 = buf[:, 12].astype(np.float64), buf[:, 13].astype(np.float64)
    good = t6 > t0
    ph = np.diff(buf[good][:, :7], axis=1).astype(np.float64)
    nls = (buf[good][:, 7] // 3)
    it = ((out["status"][good] >> 8) & 0xffff)
    print("n %4d: span %6.1f us, mean solve %5.1f us | mean cycles load %5.0f H %6.0f sweep %6.0f x0 %5.0f active set %6.0f out %5.0f | nls mean %.1f, changes mean %.1f" % (
        n, (t6[good].max() - t0[good].min()) / 100, (t6 - t0)[good].mean() / 100, *ph.mean(axis=0), nls.mean(), it.mean()))
    del ctx
