"""h = 16, warm sequence: wave 0's sub-phase stamps (build with QRGPU_GI_STAMPS=1) of the hardest robots, per iteration of the change loop."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
h, n = 16, 1024
G.setup_a1(ctx, pkg, h)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=4)
for b in seq:
    out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
it = G.iterations(out["status"]).astype(np.float64)
qf = buf[:, 14]
cs = buf[:, 8:14].astype(np.float64)
gi = (buf[:, 5] - buf[:, 4]).astype(np.float64)
names = ["scan+pick", "w,delta,d", "X1", "r partial+B2", "r,steps,flags", "B3+rest"]
for lo, hi in ((1, 24), (24, 48), (48, 64), (64, 97)):
    m = (qf >= lo) & (qf < hi) & (it > 0)
    if not m.any(): continue
    per = (cs[m] / it[m, None]).mean(0)
    print("final q in [%d,%d): %4d robots, iterations mean %.1f, GI mean %.0f k | stamped per iteration: %s | stamped sum %.0f of %.0f per iteration" % (
        lo, hi, m.sum(), it[m].mean(), gi[m].mean() / 1e3, " ".join("%s %.0f" % (nm, v) for nm, v in zip(names, per)), per.sum(), (gi[m] / it[m]).mean()))
o = np.argsort(-gi)[:5]
for k in o:
    print("robot %d: GI %.0f k, iterations %d, final q %d, nls %d, stamped %s" % (k, gi[k] / 1e3, it[k], qf[k], buf[k, 7] // 3, (cs[k] / 1e3).round(0)))
