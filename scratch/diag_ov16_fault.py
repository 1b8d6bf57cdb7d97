"""QRGPU_OV_FAULT=2 at h = 16 (tests/test_gpu_overlap.py::test_h16_hand_overs_nobody_takes_are_never_silent) with the -DQR_TIMELINE build: which launches
touched the robots whose torque is wrong without a flag (QR_TRACE bits: 1 main pass saw it, 2 skipped it, 4 handed on at once, 8 planned entry, 16 rescue
entry taken, 32 reached its wait, 64 handed on after its solve)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest, gpu_helpers as G
import test_gpu_overlap as T
pkg = conftest.load_pkg()
h, n = 16, 1024
seq = pkg.make_batch_sequence(n, h, "a1", seed=0x16AC, steps=6, excite=1.0)[:int(os.environ.get("TICKS", 6))]
ser, prevs, prev_s, _ = T._run_sequence(pkg, seq, h, n, "piped", True)
# the overlapped run by hand, so that the context is still there for the trace
ctx = pkg.Context(0, n, 16)
G.setup_a1(ctx, pkg, h)
ctx.set_planned_list(True); assert ctx.set_tick_overlap(True); ctx.set_torque_epilogue(hip_comp=True, clip=True)
lib = ctx._lib
lib.qrgpu_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
print("timeline build:", lib.qrgpu_debug_timeline(ctx._h, None) == 0)
S = pkg.to_soa
d_prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
bufs = [dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
             fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])),
             force=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
             qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32))) for b in seq]
ctx.sync()
for d in bufs:
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d_prev, d["force"], d["tau"], d["status"], qdes=d["qdes"])
ctx.sync()
print("stats", ctx.tick_overlap_stats(), ctx.last_error())
trc = np.zeros((16, 1024), np.int64)
lib.qrgpu_debug_timeline_trace.argtypes = [C.c_void_p, C.c_void_p]
lib.qrgpu_debug_timeline_trace(ctx._h, trc.ctypes.data_as(C.c_void_p))
tl = np.zeros((65, 8), np.int64)
lib.qrgpu_debug_timeline(ctx._h, tl.ctypes.data_as(C.c_void_p))
ep = int(tl[64, 0])
for k, (x, d) in enumerate(zip(ser, bufs)):
    e = ep - len(bufs) + 1 + k
    st = d["status"].download(); tau = d["tau"].download().T
    fl = G.flags(st)
    et = (np.abs(x["tau"] - tau) / np.maximum(1.0, np.abs(x["tau"]))).max(1)
    bad = np.nonzero((fl == 0) & (G.flags(x["status"]) == 0) & ~(et <= 1e-5))[0]
    big = np.asarray(seq[k]["gait"]).reshape(n, -1).sum(1) >= 43
    kinds = {}
    for r in range(n):
        t = int(trc[e & 15, r]); key = hex(t & 0xffff) if ((t >> 32) & 15) == (e & 15) else "none"
        kinds[key] = kinds.get(key, 0) + 1
    print("tick %d (epoch %d): flagged %d (time-out %d), wrong without a flag %d; trace words of all robots: %s" % (k, e, int((fl != 0).sum()), int(((fl & 0x02000000) != 0).sum()), bad.size, kinds))
    for r in bad[:6]:
        t = int(trc[e & 15, r])
        print("     robot %d big %s err %.3f trace %s" % (r, bool(big[r]), et[r], hex(t & 0xffff)))
tlr = np.zeros((4, n), np.int32)
lib.qrgpu_debug_timeline_robots.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
print("robots rc", lib.qrgpu_debug_timeline_robots(ctx._h, tlr.ctypes.data_as(C.c_void_p), n))
t0 = int(tlr[0].min())
print("last tick, per robot (us after the first WBC workgroup started): WBC started / flag seen / WBC done / the solve raised its flag")
for r in list(bad[:6]) + [int(x) for x in np.nonzero((fl == 0) & (et <= 1e-5))[0][:3]]:
    print("     robot %d: %s" % (r, [round((int(tlr[j, r]) - t0) / 100.0, 1) for j in range(4)]))
ctx.close()
