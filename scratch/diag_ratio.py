"""Smallest z'c / c'Mc among the rows each solve ADDS (how close to dependent an accepted row gets).  Build with -DQR_DIAG_REFAC."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
ctx.set_warm_start(False)
for h, n, exc in ((10, 2048, 1.0), (10, 2048, 2.0), (16, 1024, 1.0), (5, 1024, 1.0)):
    ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    b = pkg.make_batch(n, h, "a1", seed=0xA1 + 2, excite=exc)
    out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    r = buf[:, 15].view(np.float64)
    it = (out["status"] >> 8) & 0xffff
    r = r[it > 2]
    print("h", h, "excite", exc, "flagged", int(((out["status"] & 0xff) != 0).sum()), "min ratio percentiles [0, 0.1, 1, 10, 50]%:",
          np.percentile(r, [0, 0.1, 1, 10, 50]), "count <1e-9:", int((r < 1e-9).sum()), "<1e-7:", int((r < 1e-7).sum()), "<1e-5", int((r < 1e-5).sum()))
