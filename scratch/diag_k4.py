"""Phase-2 stamps of the main pass (build with QRGPU_EXTRA_FLAGS=-DQR_K4_STAMPS): when every wave of the workgroup ends its share of the H / g
build (cycles after the phase's start, wave 0's stamp), and wave 0's unit loop split into set-up / chains / tile dump."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = pkg.make_batch(n, 10, "a1", seed=0xA3)
out = G.run_mpc(ctx, pkg, b)
out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
t1 = buf[:, 1]
print("end of phase 2 per wave, cycles after TS1 (mean over robots): " + " ".join("%6.0f" % (buf[:, 8 + w] - t1).mean() for w in range(8)))
print("TS2 (wave 0 after its share) %.0f ; TS3 (after the sweep) %.0f" % ((buf[:, 2] - t1).mean(), (buf[:, 3] - t1).mean()))
print("wave 0 units: setup %.0f  chains %.0f  dump %.0f cycles" % (buf[:, 4].mean(), buf[:, 5].mean(), buf[:, 6].mean()))
