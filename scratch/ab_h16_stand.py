"""h = 16, a shard in which most robots stand (all stance: the class that cannot share a CU): the two-to-a-CU main pass against one workgroup per CU.
usage: ab_h16_stand.py [frac_all_stance] [n]"""
import sys, os, time, numpy as np
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
h, n = 16, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = pkg.Context(0, n, h)
G.setup_a1(ctx, pkg, h)
b = pkg.make_batch(n, h, "a1", seed=77, frac_all_stance=frac, frac_three_leg=0.0)
S = pkg.to_soa
d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
         gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), fb=ctx.alloc((37, n)).upload(S(b["fb_state"])),
         cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
         force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32))
def tick():
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"], None)
for _ in range(12):
    tick()
ctx.sync()
t0 = time.perf_counter()
K = 60
for _ in range(K):
    tick()
ctx.sync()
dt = (time.perf_counter() - t0) / K
st = d["status"].download()
print("QRGPU_H16_TWO=%s frac_all_stance %.2f n %d: %.3f ms per tick, %.3f M ticks/s, flagged %d" % (os.environ.get("QRGPU_H16_TWO", "default"), frac, n, dt * 1e3, n / dt / 1e6, int((G.flags(st) != 0).sum())))
