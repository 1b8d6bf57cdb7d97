"""Per-CU timelines of the main MPC launch (build with QRGPU_EXTRA_FLAGS=-DQR_SLOT_STAMPS): for every second-round robot, the gap between its
start and the end of the robot whose slot it took (the nearest earlier end on the same CU), from the shared 100 MHz clock."""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = 10
ctx = pkg.Context(0, max(n, 4096), 16)
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
ctx.set_planned_list(False)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=6)
for b in seq:
    out = G.run_mpc(ctx, pkg, b)
buf = np.zeros((n, 16), np.int64)
lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
t0, t6, hw = buf[:, 12].astype(np.float64) / 100, buf[:, 13].astype(np.float64) / 100, buf[:, 15]
good = t6 > t0
base = t0[good].min()
t0 -= base; t6 -= base
hwid = hw & 0xffffffff; xcc = (hw >> 32) & 0xf
cu = ((xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf))
print("distinct CUs:", len(np.unique(cu[good])), " robots:", int(good.sum()), " span %.1f us" % (t6[good].max()))
gaps = []; idle_tail = []
for c in np.unique(cu[good]):
    idx = np.where(good & (cu == c))[0]
    idx = idx[np.argsort(t0[idx])]
    ends = sorted(t6[idx])
    used = set()
    for k in idx:
        if t0[k] < 15.0: continue            # first round
        prev = [e for e in ends if e <= t0[k] + 0.5 and e not in used]
        if prev:
            e = max(prev); used.add(e); gaps.append(t0[k] - e)
gaps = np.array(gaps)
print("second-round starts: %d; gap end -> next start on the same CU (us): mean %.2f median %.2f p10 %.2f p90 %.2f max %.2f" % (
    len(gaps), gaps.mean(), np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90), gaps.max()))
st = np.sort(t0[good]); print("start of the last-started robot %.1f us; first-round starts: p50 %.2f p99 %.2f max(<15us) %.2f" % (st[-1], np.median(st[st < 15]), np.percentile(st[st < 15], 99), st[st < 15].max()))
print("robots per CU: min %d max %d" % (min(np.bincount(np.unique(cu[good], return_inverse=True)[1])), max(np.bincount(np.unique(cu[good], return_inverse=True)[1]))))
# the last finishers and the robot before them in their slot
late = np.argsort(-t6 * good)[:8]
for k in late:
    same = np.where(good & (cu == cu[k]))[0]
    print("  robot %4d: start %.1f len %.1f end %.1f | its CU ran:" % (k, t0[k], t6[k] - t0[k], t6[k]), sorted([(round(t0[j], 1), round(t6[j], 1)) for j in same]))
# what the same robots would take under list scheduling without gaps: 512 slots, robots in the order they actually started, each slot taking
# the next robot the moment it is free (durations as measured)
import heapq
d = (t6 - t0)[good]; order = np.argsort(t0[good])
for gap in (0.0, 0.5, 3.3):
    slots = [0.0] * 512; heapq.heapify(slots); end = 0.0; last_start = 0.0
    for k in order:
        s = heapq.heappop(slots); s2 = s + (gap if s > 0 else 0.0); e = s2 + d[k]; end = max(end, e); last_start = max(last_start, s2); heapq.heappush(slots, e)
    print("list scheduling, start order as measured, gap %.1f us: span %.1f us, last start %.1f" % (gap, end, last_start))
o2 = np.argsort(-d)
slots = [0.0] * 512; heapq.heapify(slots); end = 0.0
for k in o2:
    s = heapq.heappop(slots); e = s + d[k]; end = max(end, e); heapq.heappush(slots, e)
print("list scheduling, exact longest-first, no gap: span %.1f us; sum/512 = %.1f us, max %.1f" % (end, d.sum() / 512, d.max()))
np.save('/root/repo/gpurun_out/slots/t.npy', np.stack([t0, t6, cu.astype(np.float64)]))
