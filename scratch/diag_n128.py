import sys, time, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
n, h = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 10
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
ctx.set_torque_epilogue(hip_comp=True, clip=True)
lib = ctx._lib; lib.qrgpu_debug_lists.argtypes = [C.c_void_p, C.c_void_p]
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2 + 1000 * 5, steps=8)
S = pkg.to_soa
dev = [dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
            fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"]))) for b in seq]
prev = ctx.alloc((3, n)); force = ctx.alloc((12, n)); tau = ctx.alloc((12, n)); st = ctx.alloc((n,), np.int32)
walk = list(range(8)) + list(range(6, 0, -1))
for i in range(20):
    d = dev[walk[i % len(walk)]]
    ctx.sync(); t0 = time.perf_counter()
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], prev, force, tau, st)
    ctx.sync(); dt = (time.perf_counter() - t0) * 1e3
    c = np.zeros(8, np.int32); lib.qrgpu_debug_lists(ctx._h, c.ctypes.data)
    s = st.download(); it = G.iterations(s)
    nls = (seq[walk[i % len(walk)]]["gait"].reshape(n, h, 4).sum((1, 2))).astype(int)
    print("tick %2d: %.3f ms  lists(rescue, planned) %s  max it %d (robot %d, stance leg-steps %d)  flags %d" % (i, dt, c[:4].tolist(), it.max(), it.argmax(), nls[it.argmax()], int((G.flags(s) != 0).sum())))
