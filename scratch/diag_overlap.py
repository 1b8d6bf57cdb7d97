"""Timeline of overlapped ticks from the kernels' stamps (library built with -DQR_TIMELINE): per tick, on one clock, when its gate came up and opened,
first / last main-pass workgroup start, last solve, first / last WBC workgroup, trailing launch."""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, gpu_helpers as G
pkg = conftest.load_pkg()
n, h = int(os.environ.get("N", 1024)), 10
draw = int(os.environ.get("DRAW", 2))
K = int(os.environ.get("K", 30))
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2 + 1000 * draw, steps=8, excite=1.0)
walk = list(range(8)) + list(range(6, 0, -1))
ctx = pkg.Context(0, n, 16)
G.setup_a1(ctx, pkg, h)
ctx.set_torque_epilogue(True, True)
print("overlap on:", ctx.set_tick_overlap(os.environ.get("OV", "1") == "1", strict=False), ctx.last_error())
S = pkg.to_soa
dev = [[ctx.alloc(S(b[k]).shape).upload(S(b[k])) for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd")] for b in seq]
prev = ctx.alloc((3, n)); prev.zero()
outs = [dict(force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32)) for _ in range(2)]
def step(i):
    ds, dt_, dg, dfb, dcmd = dev[walk[i % len(walk)]]
    o = outs[i & 1]
    ctx.tick_batch(n, ds, dt_, dg, dfb, dcmd, prev, o["force"], o["tau"], o["status"], qdes=o["qdes"])
for i in range(12):
    step(i)
ctx.sync()
lib = ctx._lib
lib.qrgpu_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
assert lib.qrgpu_debug_timeline(ctx._h, None) == 0
t0 = time.perf_counter()
for i in range(12, 12 + K):
    step(i)
ctx.sync()
print("%.1f us per tick" % ((time.perf_counter() - t0) / K * 1e6))
tl = np.zeros((65, 8), np.int64)
lib.qrgpu_debug_timeline(ctx._h, tl.ctypes.data_as(C.c_void_p))
g2 = np.zeros((64, 2), np.int64)
lib.qrgpu_debug_gate2(ctx._h, g2.ctypes.data_as(C.c_void_p))
ep = int(tl[64, 0])
base = tl[(ep - K + 1) & 63][0]
f = lambda v: "%8.1f" % ((v - base) / 100.0) if 0 < v < 0x7fffffffffffffff else "%8s" % "-"
print("epoch   gate up  gate open | first main  last main start  last solve | first WBC  last WBC | trail start  trail end")
for e in range(ep - K + 1, ep + 1):
    r = tl[e & 63]; g = g2[e & 63]
    print("%5d  %s %s | %s %s %s | %s %s | %s %s" % (e, f(g[0]), f(g[1]), f(r[0]), f(r[1]), f(r[2]), f(r[3]), f(r[4]), f(r[5]), f(r[6])))
ctx.close()
