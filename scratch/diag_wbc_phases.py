import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, 10)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
n = 1024
if os.environ.get("SEQ") == "1":        # a coherent sequence: the last tick's stamps, every tick warm from its predecessor
    for b in pkg.make_batch_sequence(n, 10, "a1", seed=0xA3, steps=6, excite=1.0):
        out = G.run_tick(ctx, pkg, b, want_qdes=True)
else:
    b = pkg.make_batch(n, 10, "a1", seed=0xA3)
    out = G.run_tick(ctx, pkg, b, want_qdes=True); out = G.run_tick(ctx, pkg, b, want_qdes=True)
buf = np.zeros((n, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, -n)
buf[:, 6] = buf[:, 5]
d = np.diff(buf[:, :10], axis=1).astype(np.float64)
names = ["load", "per-leg dyn (A)", "base block", "A^-1", "wait for wave 1", "-", "WBIC recursion", "QP", "store"]
for k, nm in enumerate(names[:9]):
    print("  %-16s mean %8.0f  p50 %8.0f  max %8.0f" % (nm, d[:, k].mean(), np.median(d[:, k]), d[:, k].max()))
print("total mean %.0f max %.0f" % ((buf[:, 9] - buf[:, 0]).mean(), (buf[:, 9] - buf[:, 0]).max()))

it = buf[:, 10].astype(np.float64); qp = d[:, 7]
A = np.stack([np.ones(n), it], 1); coef, *_ = np.linalg.lstsq(A, qp, rcond=None)
print("QP: iterations mean %.1f p50 %d max %d ; cycles ~ %.0f + %.0f per iteration (least squares)" % (it.mean(), np.median(it), it.max(), coef[0], coef[1]))

print("wave 1: reaches the meeting point at %.0f (wave 0 at %.0f), ends at %.0f (wave 0 at %.0f)" % ((buf[:, 11] - buf[:, 0]).mean(), (buf[:, 4] - buf[:, 0]).mean(), (buf[:, 13] - buf[:, 0]).mean(), (buf[:, 9] - buf[:, 0]).mean()))
print("WBIC: contact part %.0f, first task %.0f, remaining tasks %.0f" % ((buf[:, 12] - buf[:, 5]).mean(), (buf[:, 14] - buf[:, 12]).mean(), (buf[:, 7] - buf[:, 14]).mean()))
