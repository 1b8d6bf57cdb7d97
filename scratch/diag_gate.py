import sys, time, numpy as np, os
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h, n = 10, 256
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
b = pkg.make_batch(n, h, "a1", seed=0x6A7E)
for _ in range(4): ref = G.run_tick(ctx, pkg, b)       # (the first calls of a new batch size end with a stream sync: past those)
S = pkg.to_soa
d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
         fb=ctx.alloc((37, n)), cmd=ctx.alloc((67, n)), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
         force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32)))
big = ctx.alloc((1 << 28,))
pin_fb = ctx.alloc_pinned((37, n)); pin_fb.array[...] = S(b["fb_state"])
pin_cmd = ctx.alloc_pinned((67, n)); pin_cmd.array[...] = S(b["wbc_cmd"])
d["fb"].zero(); d["cmd"].zero()
ctx.sync()
t0 = time.perf_counter()
for _ in range(int(os.environ.get("NFILL", "200"))):
    big.zero()
t1 = time.perf_counter()
d["fb"].copy_from_pinned(pin_fb); d["cmd"].copy_from_pinned(pin_cmd)
ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"])
t2 = time.perf_counter()
ctx.sync()
t3 = time.perf_counter()
st = d["status"].download(); tau = d["tau"].download().T
print("enqueue fills %.1f ms, enqueue tick %.2f ms, until sync %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3))
print("result flags %d ref flags %d finite %d max tau diff %.3g" % (int((G.flags(st) != 0).sum()), int((G.flags(ref["status"]) != 0).sum()), int(np.isfinite(tau).all()), float(np.abs(tau - ref["tau"]).max())))
