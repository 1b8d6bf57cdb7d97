"""Per-robot solve times, phases, sizes and change counts of consecutive ticks of a coherent sequence (instrumented build, slot stamps):
raw material for the dispatch-order predictor (offline: scratch/analyze_predict.py)."""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = 10
ctx = pkg.Context(0, max(n, 4096), 16)
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.set_planned_list(False)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=8)
bufs = []; its = []; gaits = []
path = list(range(8)) + list(range(6, -1, -1))            # the bench walks the sequence back and forth
lib.qrgpu_debug_cycles(ctx._h, None, 0)
for k in path:
    out = G.run_mpc(ctx, pkg, seq[k])
    buf = np.zeros((n, 16), np.int64)
    lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    bufs.append(buf.copy()); its.append((out["status"] >> 8) & 0xffff); gaits.append(seq[k]["gait"].copy())
np.savez_compressed('/root/repo/gpurun_out/slots/predict.npz', buf=np.stack(bufs), it=np.stack(its), gait=np.stack(gaits))
print("saved", len(bufs))
