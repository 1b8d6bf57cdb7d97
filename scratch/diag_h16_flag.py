import sys, numpy as np
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from conftest import load_pkg
import gpu_helpers as G, oracle_py as O
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
h, n = 16, 512
ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
b = pkg.make_batch(n, h, "a1", seed=0xA1 + 2, excite=1.0)
for warm in (False, True, True):
    ctx.set_warm_start(warm)
    out = G.run_mpc(ctx, pkg, b)
    fl = out["status"] & 0xff; it = (out["status"] >> 8) & 0xffff
    bad = np.where(fl != 0)[0]
    print("warm", warm, "flagged", bad, fl[bad], "iters", it[bad], "mean it", it.mean())
    for i in bad[:4]:
        u, st, rc = O.mpc_solve(pkg.mpc_cfg("a1"), h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        print("  robot", i, "oracle rc", rc, st, "nls", int((b["gait"][i] > 0).sum()), "force err", np.abs(out["force"][i] - u[:12]).max())
# refactor diagnostics (build with QRGPU_EXTRA_FLAGS=-DQR_DIAG_REFAC)
import ctypes as C
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
ctx.set_warm_start(False)
# zero the debug buffer through a run: slots 8, 13 accumulate -> read twice and subtract
buf0 = np.zeros((n, 16), np.int64); buf1 = np.zeros((n, 16), np.int64)
out = G.run_mpc(ctx, pkg, b); lib.qrgpu_debug_cycles(ctx._h, buf0.ctypes.data, n)
out = G.run_mpc(ctx, pkg, b); lib.qrgpu_debug_cycles(ctx._h, buf1.ctypes.data, n)
for i in np.where(((out["status"] & 0xff) != 0) | ((buf1[:, 8] - buf0[:, 8]) != 0))[0][:8]:
    d = buf1[i]
    print("robot", i, "status", out["status"][i] & 0xff, "rebuilds", d[8] - buf0[i][8], "restore drops", d[13] - buf0[i][13], "q at exit", d[12],
          "exit residuals", [np.int64(x).view(np.float64) if False else np.array([x], np.int64).view(np.float64)[0] for x in d[9:12]])
