"""1024 robots as K sub-batches on K contexts (K streams), stepped in turn with no sync: what a host that splits its population gets, against one
context with all 1024 (coherent sequences, the bench's draw 0)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
import os
MODE = os.environ.get('MODE', 'tick')
h, N = 10, 1024
S = pkg.to_soa
def run(K, steps=200, warm=20):
    n = N // K
    ctxs = [pkg.Context(0, n, 16) for _ in range(K)]
    data = []
    for k, ctx in enumerate(ctxs):
        G.setup_a1(ctx, pkg, h); ctx.set_torque_epilogue(hip_comp=True, clip=True)
        seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2 + 77 * k, steps=8)
        dev = [dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
                    fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"]))) for b in seq]
        data.append(dict(dev=dev, prev=ctx.alloc((3, n)), force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), st=ctx.alloc((n,), np.int32), qd=ctx.alloc((24, n)), n=n))
    walk = list(range(8)) + list(range(6, 0, -1))
    def step(i):
        for ctx, D in zip(ctxs, data):
            d = D["dev"][walk[i % len(walk)]]
            if MODE == "mpc": ctx.mpc_solve_batch(D["n"], d["state"], d["traj"], d["gait"], d["fb"].row(13), D["force"], D["tau"], D["st"])
            elif MODE == "wbc": ctx.wbc_run_batch(D["n"], d["fb"], d["cmd"], D["prev"], D["tau"], D["qd"], D["st"])
            else: ctx.tick_batch(D["n"], d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], D["prev"], D["force"], D["tau"], D["st"], qdes=D["qd"])
    for i in range(warm): step(i)
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps): step(i)
    t_enq = time.perf_counter() - t0
    for c in ctxs: c.sync()
    el = time.perf_counter() - t0
    print("   (K = %d: host enqueue %.1f us per tick_batch call; %.1f us per step in all)" % (K, t_enq / steps / K * 1e6, el / steps * 1e6))
    flags = sum(int((G.flags(D["st"].download()) != 0).sum()) for D in data)
    for c in ctxs: c.close()
    return N * steps / el, flags
import os
if os.environ.get("ONE512"):
    N = 512
    v, fl = run(1); print("one context, 512 robots: %.3f M ticks/s" % (v / 1e6))
    N = 2048
    v, fl = run(2); print("two contexts, 1024 robots each: %.3f M ticks/s" % (v / 1e6))
    v, fl = run(1); print("one context, 2048 robots: %.3f M ticks/s" % (v / 1e6))
else:
    for K in (1, 2, 4, 1, 2):
        v, fl = run(K)
        print("K = %d sub-batches of %d robots: %.3f M ticks/s, flags %d" % (K, N // K, v / 1e6, fl))
