"""Overlapped ticks on a population with many list robots, tick by tick with a sync: which tick's join gives up, and what the lists say."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, gpu_helpers as G
pkg = conftest.load_pkg()
import ctypes as C
h, n = 10, 512
seq = pkg.make_batch_sequence(n, h, "a1", seed=0x91BE, steps=6, frac_all_stance=0.3, excite=1.5)
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
print("overlap", ctx.set_tick_overlap(True, strict=False), ctx.last_error())
ctx.set_torque_epilogue(True, True)
S = pkg.to_soa
prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
sync_each = os.environ.get("SYNC", "1") == "1"
bufs = []
for b in seq:
    d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
             fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)),
             qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32))
    bufs.append(d)
ctx.sync()
cnt0 = np.zeros(12, np.int32); ctx._lib.qrgpu_debug_counters(ctx._h, cnt0.ctypes.data_as(C.c_void_p))
TL = os.environ.get("TL") == "1"
if TL:
    ctx._lib.qrgpu_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
    print("timeline on:", ctx._lib.qrgpu_debug_timeline(ctx._h, None))
for k, d in enumerate(bufs):
    t0 = time.perf_counter()
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], prev, d["force"], d["tau"], d["status"], qdes=d["qdes"])
    if sync_each:
        try:
            ctx.sync()
        except Exception as e:
            print("tick", k, "sync failed:", str(e)[:80])
        cnt = np.zeros(12, np.int32)
        ctx._lib.qrgpu_debug_counters(ctx._h, cnt.ctypes.data_as(C.c_void_p))
        print("   counters (dev/host): main_started %d/%d wbc_finished %d/%d tick_done %d/%d lane1 %d/%d lane2 %d/%d epoch %d" % tuple(cnt[:11].tolist()))
        sv = np.zeros(n, np.uint32); wd = np.zeros(n, np.uint32)
        ctx._lib.qrgpu_debug_words(ctx._h, sv.ctypes.data_as(C.c_void_p), wd.ctypes.data_as(C.c_void_p), n)
        print("   solved:", {int(v): int((sv == v).sum()) for v in np.unique(sv)}, " wbc_done:", {int(v): int((wd == v).sum()) for v in np.unique(wd)},
              " robots with old wbc_done:", np.nonzero(wd != wd.max())[0][:24].tolist())
        lists = np.zeros(8, np.int32)
        ctx._lib.qrgpu_debug_lists(ctx._h, lists.ctypes.data_as(C.c_void_p))
        st = d["status"].download()
        fl = G.flags(st)
        print("tick %d: %.1f ms; lists %s; flags %s; stats %s" % (k, 1e3 * (time.perf_counter() - t0), lists.tolist(), {hex(int(v)): int((fl == v).sum()) for v in np.unique(fl)}, ctx.tick_overlap_stats()))
try:
    ctx.sync()
except Exception as e:
    print("final sync failed:", str(e)[:80])
for k, d in enumerate(bufs):
    fl = G.flags(d["status"].download())
    print("tick", k, {hex(int(v)): int((fl == v).sum()) for v in np.unique(fl)})
if TL:
    tl = np.zeros((65, 8), np.int64)
    ctx._lib.qrgpu_debug_timeline(ctx._h, tl.ctypes.data_as(C.c_void_p))
    ep = int(tl[64, 0])
    base = tl[(ep - len(bufs) + 1) & 63][0]
    names = ["first main", "last main start", "last solve", "first WBC", "last WBC done", "trail start", "trail end", "2nd WBC end"]
    print("epoch  " + "  ".join("%14s" % s_ for s_ in names))
    for e in range(ep - len(bufs) + 1, ep + 1):
        r = tl[e & 63]
        print("%5d  " % e + "  ".join("%14.1f" % ((v - base) / 100.0) if 0 < v < 0x7fffffffffffffff else "%14s" % "-" for v in r))
ctx.close()
