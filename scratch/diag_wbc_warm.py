"""WBC relaxation QP with and without the warm start (QRGPU_WBC_WARM): changes and cycles of the QP per robot over a coherent sequence
(instrumented kernel: serial tick)."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
n, h = 1024, 10
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=8)
S = pkg.to_soa
prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
force, tau, st = ctx.alloc((12, n)), ctx.alloc((12, n)), ctx.alloc((n,), np.int32)
for k, b in enumerate(seq):
    d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
             fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])))
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], prev, force, tau, st)
    ctx.sync()
    buf = np.zeros((n + 8, 16), np.int64)
    lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, -(n + 8))
    buf = buf[:n]
    qp = (buf[:, 8] - buf[:, 7]) / 1000.0; its = buf[:, 10]; tot = (buf[:, 9] - buf[:, 0]) / 1000.0
    good = (buf[:, 9] > buf[:, 0]) & (qp > 0)
    print("tick %d: QP k cycles mean %.1f p90 %.1f max %.1f; changes mean %.2f max %d; whole workgroup (wave 0) mean %.1f max %.1f" % (
        k, qp[good].mean(), np.percentile(qp[good], 90), qp[good].max(), its[good].mean(), its[good].max(), tot[good].mean(), tot[good].max()))
    for v in d.values(): v.free()
