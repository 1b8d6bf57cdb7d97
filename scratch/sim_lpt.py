"""How much of the launch span is scheduling: per-robot solve times of consecutive ticks (the kernel's own stamps, 100 MHz clock),
list scheduling simulated on 512 slots with different dispatch orders."""
import sys, ctypes as C, heapq
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n, h = 1024, 10
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
ctx.set_planned_list(False)
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=8)
costs = []
for b in seq:
    out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    costs.append((buf[:, 13] - buf[:, 12]) / 100.0)
    span = (buf[:, 13].max() - buf[:, 12].min()) / 100.0
    print("measured span %.1f us, mean solve %.1f, max %.1f" % (span, costs[-1].mean(), costs[-1].max()))
costs = np.array(costs)
def sched(order, c, slots=512):
    heap = [0.0] * slots
    for i in order:
        t = heapq.heappop(heap); heapq.heappush(heap, t + c[i])
    return max(heap)
for k in range(3, 8):
    c = costs[k]
    prev = costs[k - 1]
    ema = 0.5 * costs[k - 1] + 0.3 * costs[k - 2] + 0.2 * costs[k - 3]
    # what the hardware does today: eight pools (XCDs) of 64 slots, robots [128 x, 128 x + 128) in pool x, each pool longest-first by the previous tick
    chunked = max(sched(128 * x + np.argsort(-prev[128 * x:128 * x + 128]), c, 64) for x in range(8))
    # one global longest-first order dealt round-robin to the pools
    g = np.argsort(-prev)
    dealt = max(sched(g[x::8], c, 64) for x in range(8))
    print("          eight pools of 64 slots: chunk-local order %.0f, global order dealt round-robin %.0f" % (chunked, dealt))
    print("tick %d: corr(prev, now) %.2f | makespan: slot order %.0f, by previous tick %.0f, by 3-tick average %.0f, by true cost (ideal LPT) %.0f, lower bound %.0f" % (
        k, np.corrcoef(prev, c)[0, 1], sched(range(n), c), sched(np.argsort(-prev), c), sched(np.argsort(-ema), c), sched(np.argsort(-c), c), max(c.max(), c.sum() / 512)))
