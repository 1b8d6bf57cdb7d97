import sys, time, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h, n = 10, 512
ctx = pkg.Context(0, 1024, 16)
lib = ctx._lib
G.setup_a1(ctx, pkg, h)
ctx.set_tick_pipeline(True)
trot = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=0.0, frac_three_leg=0.0)
stance = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=1.0, frac_three_leg=0.0, excite=2.0)
def mix(k):
    b = {key: (v.copy() if isinstance(v, np.ndarray) else v) for key, v in trot.items()}
    idx = np.arange(k) * (n // max(k, 1)) + 3
    for key in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[key][idx] = stance[key][idx]
    return b
few, many = mix(8), mix(120)
S = pkg.to_soa
def upload(b):
    return dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
                gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])))
dF, dM = upload(few), upload(many)
d_prev = ctx.alloc((3, n)).upload(S(many["prev_ori_vel"]))
force, tau, status = ctx.alloc((12, n)), ctx.alloc((12, n)), ctx.alloc((n,), np.int32)
def tick(d):
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d_prev, force, tau, status)
has = hasattr(lib, "qrgpu_debug_lists")
def lists():
    if not has: return None
    lib.qrgpu_debug_lists.argtypes = [C.c_void_p, C.c_void_p]
    c = np.zeros(8, np.int32); lib.qrgpu_debug_lists(ctx._h, c.ctypes.data); return c.tolist()
for i in range(4):
    tick(dF); ctx.sync(); print("few tick", i, "lists", lists())
t0 = time.perf_counter()
tick(dM); tick(dM); ctx.sync()
print("two queued ticks: %.2f ms" % ((time.perf_counter() - t0) * 1e3), "lists", lists(), "timeouts", int((G.flags(status.download()) & 0x02000000 != 0).sum()))
t0 = time.perf_counter()
tick(dM); tick(dF); tick(dM); tick(dM); ctx.sync()
print("four queued ticks: %.2f ms" % ((time.perf_counter() - t0) * 1e3), "lists", lists(), "timeouts", int((G.flags(status.download()) & 0x02000000 != 0).sum()))
