"""Warm start diagnostics: rebuild success, restore drops, iterations (build with -DQR_DIAG_REFAC)."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
h, n = int(sys.argv[1]) if len(sys.argv) > 1 else 10, 1024
ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=6)
ctx.set_warm_start(True)
prev = np.zeros((n, 16), np.int64)
for k, b in enumerate(seq):
    out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    reb = buf[:, 8] - prev[:, 8]; drops = buf[:, 13] - prev[:, 13]; prev = buf
    it = (out["status"] >> 8) & 0xffff
    print("step", k, "flagged", int(((out["status"] & 0xff) != 0).sum()), "iters mean %.1f max %d" % (it.mean(), it.max()),
          "| rebuild ok %d failed %d none %d" % ((reb % 100 > 0).sum(), (reb >= 100).sum(), (reb == 0).sum()), "| restore drops mean %.2f max %d" % (drops.mean(), drops.max()))
