import sys, time, numpy as np
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
n, h = 1024, 10
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
t0 = time.perf_counter()
ctx.comm_init_rank(pkg.qrgpu.comm_unique_id(), 1, 0)
print("comm init %.2f s" % (time.perf_counter() - t0))
b = pkg.make_batch(n, h, "a1", seed=31)
S = pkg.to_soa
d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
         fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
         force=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32))
tau = [ctx.alloc((12, n)), ctx.alloc((12, n))]
tau_all = ctx.alloc((1, 12, n))
def run(k, gather, fence=True):
    ctx.sync(); t0 = time.perf_counter()
    for i in range(k):
        slot = i & 1
        if gather and fence: ctx.allgather_fence(slot)
        ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], tau[slot], d["status"])
        if gather: ctx.allgather_tau(tau[slot], n, tau_all, slot, of_tick=OF_TICK)
    if gather: ctx.comm_sync()
    ctx.sync()
    return (time.perf_counter() - t0) / k * 1e3
import os
OF_TICK = os.environ.get('OF_TICK', '1') == '1'
for rep in range(3):
    print("ms per tick: without gather %.4f, with gather %.4f, with gather but no fence (timing only) %.4f" % (run(100, False), run(100, True), run(100, True, False)))
