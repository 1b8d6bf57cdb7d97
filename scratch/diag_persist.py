"""Are the slow robots the same from tick to tick?  Iterations / guess size / final q over consecutive ticks (build with -DQR_DIAG_REFAC)."""
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
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=7)
its, qg, qf = [], [], []
for b in seq:
    out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    its.append((out["status"] >> 8) & 0xffff); qg.append(buf[:, 11].copy()); qf.append(buf[:, 14].copy())
its, qg, qf = np.array(its), np.array(qg), np.array(qf)
print("corr of iteration counts between consecutive ticks:", [round(float(np.corrcoef(its[k], its[k + 1])[0, 1]), 2) for k in range(2, 6)])
worst = np.argsort(-its[-1])[:10]
for r in worst:
    print("robot %4d: iterations %s | guess %s | final q %s" % (r, its[2:, r].tolist(), qg[2:, r].tolist(), qf[2:, r].tolist()))
b0, b1 = seq[-2], seq[-1]
d = np.abs(b1["mpc_state"] - b0["mpc_state"])
print("state change of the worst vs the median robot (|delta| of rows 3-5 v, 10-12 w):", d[worst][:, [3, 4, 5, 10, 11, 12]].mean(0).round(3), np.median(d[:, [3, 4, 5, 10, 11, 12]], 0).round(3))
