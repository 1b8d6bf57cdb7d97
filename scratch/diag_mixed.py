import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
n, h = 1024, 16
ctx = pkg.Context(0, n, 16)
ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
ba = pkg.make_batch(n // 2, h, "a1", seed=0xA1 + 2, excite=1.0); bl = pkg.make_batch(n // 2, h, "lite3", seed=0x173, excite=1.0)
b = dict(ba)
for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
    b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
b["n"] = n
tid = pkg.shard.interleave_types(n, 2)
out = G.run_tick(ctx, pkg, b, type_id=tid)
it = out["status"] >> 8
for t, nm in ((0, "a1"), (1, "lite3")):
    m = tid == t
    print("%-6s iterations mean %.1f p90 %d p99 %d max %d ; flagged %d" % (nm, it[m].mean(), np.percentile(it[m], 90), np.percentile(it[m], 99), it[m].max(), ((out["status"][m] & 0xff) != 0).sum()))
k = np.argsort(-it)[:8]
print("top:", [(int(i), int(it[i]), "lite3" if tid[i] else "a1", int(b["gait"][i].sum())) for i in k])
import ctypes as C
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
out = G.run_tick(ctx, pkg, b, type_id=tid); out = G.run_tick(ctx, pkg, b, type_id=tid)
buf = np.zeros((n, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
d = np.diff(buf[:, :7], axis=1).astype(np.float64)
names = ["load+srbd", "H/g build", "sweep inv", "x0", "GI", "out"]
for t, nm in ((0, "a1"), (1, "lite3")):
    m = tid == t
    print(nm, " ".join("%s %.0f/%.0f" % (names[k], d[m, k].mean(), d[m, k].max()) for k in range(6)), "| total mean %.0f max %.0f" % ((buf[m, 6] - buf[m, 0]).mean(), (buf[m, 6] - buf[m, 0]).max()))
tot = buf[:, 6] - buf[:, 0]
k = np.argsort(-tot)[:6]
print("slowest:", [(int(i), int(tot[i]), int(it[i]), "lite3" if tid[i] else "a1", int(b["gait"][i].sum()), int(buf[i, 14])) for i in k])
print("span %.0f" % (buf[:, 6].max() - buf[:, 0].min()))
