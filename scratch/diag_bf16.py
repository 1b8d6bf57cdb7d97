import sys, numpy as np
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from conftest import load_pkg
import gpu_helpers as G, oracle_py as O
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 256, 16)
h, n = 10, 32
G.setup_a1(ctx, pkg, h)
b = pkg.make_batch(n, h, "a1", seed=601)
ctx.set_warm_start(False)
Hx, gx = G.run_assemble(ctx, pkg, b)
ex = G.run_mpc(ctx, pkg, b)
ctx.set_hessian_mode("bf16x3")
Hs, gs = G.run_assemble(ctx, pkg, b)
sp = G.run_mpc(ctx, pkg, b)
# float64 evaluation of the same closed form: H64 = sum over exact products of the fp32 operands? use the oracle's fp32 H vs a float64 product of float32 factors
for i in range(3):
    m = np.isfinite(Hx[i])
    d = np.abs(Hs[i][m].astype(np.float64) - Hx[i][m])
    print("robot", i, "max|dH| %.3e, max|H| %.3e, ratio %.2e; median rel per entry %.2e; 2 alpha = 8e-6" % (d.max(), np.abs(Hx[i][m]).max(), d.max() / np.abs(Hx[i][m]).max(), np.median(d / np.maximum(np.abs(Hx[i][m]), 1e-30))))
    asym = np.abs(Hx[i] - Hx[i].T)[m & m.T]
    print("   the exact H's own asymmetry max %.3e" % asym.max())
print("force rel diff bf16x3 vs exact:", (np.abs(sp["force"] - ex["force"]).max(1) / np.maximum(1, np.abs(ex["force"]).max(1))).max())
