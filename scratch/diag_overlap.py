"""Timeline of overlapped ticks from the kernels' stamps (library built with -DQR_TIMELINE): per tick, on one clock, when its gate came up and opened,
first / last main-pass workgroup start, last solve, first / last WBC workgroup, trailing launch."""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, gpu_helpers as G
pkg = conftest.load_pkg()
n, h = int(os.environ.get("N", 1024)), int(os.environ.get("H", 10))
draw = int(os.environ.get("DRAW", 2))
K = int(os.environ.get("K", 30))
mixed = os.environ.get("MIXED") == "1"
d_type = None
ctx = pkg.Context(0, n, 16)
if not mixed:
    seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2 + 1000 * draw, steps=8, excite=1.0)
    G.setup_a1(ctx, pkg, h)
else:
    sa = pkg.make_batch_sequence(n // 2, h, "a1", seed=0xA1 + 2 + 1000 * draw, steps=8, excite=1.0)
    sl = pkg.make_batch_sequence(n // 2, h, "lite3", seed=0xA1 + 2 + 1000 * draw + 0xD2, steps=8, excite=1.0)
    seq = []
    for ba, bl in zip(sa, sl):
        b = dict(ba); b["n"] = n
        for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
            b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
        seq.append(b)
    ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
    d_type = ctx.alloc((n,), np.int32).upload(pkg.shard.interleave_types(n, 2))
walk = list(range(8)) + list(range(6, 0, -1))
ctx.set_torque_epilogue(True, True)
print("overlap on:", ctx.set_tick_overlap(os.environ.get("OV", "1") == "1", strict=False), ctx.last_error())
S = pkg.to_soa
dev = [[ctx.alloc(S(b[k]).shape).upload(S(b[k])) for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd")] for b in seq]
prev = ctx.alloc((3, n)); prev.zero()
outs = [dict(force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32)) for _ in range(2)]
def step(i):
    ds, dt_, dg, dfb, dcmd = dev[walk[i % len(walk)]]
    o = outs[i & 1]
    ctx.tick_batch(n, ds, dt_, dg, dfb, dcmd, prev, o["force"], o["tau"], o["status"], d_type, qdes=o["qdes"])
for i in range(12):
    step(i)
    if os.environ.get("WATCH") == "1":
        ctx.sync()
        fl = G.flags(outs[i & 1]["status"].download())
        print("   warm-up tick %d flags:" % i, {hex(int(v)): int((fl == v).sum()) for v in np.unique(fl)}, ctx.tick_overlap_stats())
ctx.sync()
for o in outs:
    fl = G.flags(o["status"].download())
    print("   after warm-up flags:", {hex(int(v)): int((fl == v).sum()) for v in np.unique(fl)})
lib = ctx._lib
lib.qrgpu_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
have_tl = lib.qrgpu_debug_timeline(ctx._h, None) == 0
t0 = time.perf_counter()
for i in range(12, 12 + K):
    step(i)
ctx.sync()
print("%.1f us per tick; overlap stats %s" % ((time.perf_counter() - t0) / K * 1e6, ctx.tick_overlap_stats()))
for o in outs:
    fl = G.flags(o["status"].download())
    print("   flags:", {hex(int(v)): int((fl == v).sum()) for v in np.unique(fl)})
lists = np.zeros(8, np.int32)
lib.qrgpu_debug_lists.argtypes = [C.c_void_p, C.c_void_p]
lib.qrgpu_debug_lists(ctx._h, lists.ctypes.data_as(C.c_void_p))
print("   lists of the last lane: rescue %s planned %s go %d plan-abort %d gate-abort %d plan epoch %d" % (lists[0:2].tolist(), lists[2:4].tolist(), lists[4], lists[5], lists[6], lists[7]))
st_ = outs[0]["status"].download()
bad = np.nonzero(G.flags(st_) != 0)[0]
gait0 = seq[walk[(12 + K - 2) % len(walk)]]["gait"]
print("   flagged robots (first 12):", bad[:12].tolist(), "their stance leg-steps:", [int(gait0[i].sum()) for i in bad[:12]], "iterations:", G.iterations(st_)[bad[:12]].tolist())
print("   stance leg-steps histogram of all robots >= 43:", int((gait0.reshape(n, -1).sum(1) >= 43).sum()))
if not have_tl:
    ctx.close(); sys.exit(0)
g2 = np.zeros((128, 2), np.int64)
lib.qrgpu_debug_gate2(ctx._h, g2.ctypes.data_as(C.c_void_p))
sv = np.zeros((2, 16, 1024), np.int64)
lib.qrgpu_debug_timeline_solves.argtypes = [C.c_void_p, C.c_void_p]
lib.qrgpu_debug_timeline_solves(ctx._h, sv.ctypes.data_as(C.c_void_p))
trc = np.zeros((16, 1024), np.int64)
lib.qrgpu_debug_timeline_trace.argtypes = [C.c_void_p, C.c_void_p]
lib.qrgpu_debug_timeline_trace(ctx._h, trc.ctypes.data_as(C.c_void_p))
pls = np.zeros(4096 + 64, np.int64)
lib.qrgpu_debug_timeline_plans.argtypes = [C.c_void_p, C.c_void_p]
lib.qrgpu_debug_timeline_plans(ctx._h, pls.ctypes.data_as(C.c_void_p))
tl = np.zeros((65, 8), np.int64)
lib.qrgpu_debug_timeline(ctx._h, tl.ctypes.data_as(C.c_void_p))
ep = int(tl[64, 0])
base = tl[(ep - K + 1) & 63][0]
f = lambda v: "%8.1f" % ((v - base) / 100.0) if 0 < v < 0x7fffffffffffffff else "%8s" % "-"
print("epoch   gate up  gate open | first main  last main start  last solve | first WBC  last WBC | trail start  trail end | planned first start, last end")
for e in range(ep - K + 1, ep + 1):
    r = tl[e & 63]; g = g2[e & 63]
    pl = g2[64 + (e & 63)]
    print("%5d  %s %s | %s %s %s | %s %s | %s %s | %s %s" % (e, f(g[0]), f(g[1]), f(r[0]), f(r[1]), f(r[2]), f(r[3]), f(r[4]), f(r[5]), f(r[6]), f(pl[0]), f(pl[1])))
print("list length: what each tick's planning left (for the lane's next tick) and what that tick's planned workgroups read:")
for e in range(ep - K + 1, ep + 1):
    pl_ = pls[4096 + (e & 63)]; rd = pls[(e & 63) * 64:(e & 63) * 64 + 64]; rd = rd[rd != 0]
    print("   epoch %d: planning left %s (parity %d); planned workgroups (%d of grid %s, stride %s) read %s under parity %s" % (e, (pl_ & 0xffff) if pl_ else "-", (pl_ >> 16) & 1, len(rd),
          sorted(set(((rd >> 20) & 0xfff).tolist())), sorted(set(((rd >> 32) & 1).tolist())), sorted(set((rd & 0xffff).tolist())), sorted(set(((rd >> 16) & 1).tolist()))))
if n <= 1024:
    print("per-robot cross-tick waits over 1 ms (us) and publish times, by epoch (last 14):")
    eps = list(range(max(ep - 13, ep - K + 1), ep + 1))
    w = sv[1][[e & 15 for e in eps]] >> 8; wk = sv[1][[e & 15 for e in eps]] & 15
    pb = sv[0][[e & 15 for e in eps]] >> 8; pk = sv[0][[e & 15 for e in eps]] & 7
    pe = (sv[0][[e & 15 for e in eps]] >> 4) & 15
    pk = sv[0][[e & 15 for e in eps]] & 15
    missing = [(e, np.nonzero(pe[j] != (e & 15))[0].tolist()) for j, e in enumerate(eps)]
    print("   robots nobody solved, by epoch:", [(e, m[:8]) for e, m in missing if m])
    badall = sorted(set(np.nonzero(G.flags(outs[0]["status"].download()) != 0)[0].tolist()) | set(np.nonzero(G.flags(outs[1]["status"].download()) != 0)[0].tolist()))
    for e, m_ in missing:
        for r in m_[:4]:
            t_ = int(trc[e & 15, r]); ok_ = ((t_ >> 32) & 15) == (e & 15)
            print("      epoch %d robot %d: trace bits %s%s; the ticks before: %s" % (e, r, hex(t_ & 0xffff), "" if ok_ else " (STALE: no trace at all)", [hex(int(trc[(e - k) & 15, r]) & 0xffff) for k in (1, 2, 3)]))
    print("   flagged robots, either buffer:", badall[:16], "stance leg-steps (second last tick):", [int(gait0[i].sum()) for i in badall[:16]])
    slow = np.nonzero((w > 100000).any(0))[0]
    slow = np.array(sorted(set(slow.tolist()) | set(badall[:4])), dtype=np.int64)
    print("   robots with a long wait:", len(slow), slow[:20].tolist())
    for r in slow[:6].tolist() + [int(x) for x in np.nonzero(gait0.reshape(n, -1).sum(1) >= 43)[0][:3]]:
        print("   robot %d (stance leg-steps now %d):" % (r, int(gait0[r].sum())))
        for j, e in enumerate(eps):
            print("      epoch %d: waited %8.1f us (kind %d%s)  published at %s (kind %d%s)%s" % (e, w[j, r] / 100.0, wk[j, r] & 7, " GAVE UP" if wk[j, r] & 8 else "", f(pb[j, r]), pk[j, r] & 7, " to rescue" if pk[j, r] & 8 else "", "" if pe[j, r] == (e & 15) else "  <-- STALE RECORD: nobody solved it"))
ctx.close()
