"""Concurrency of the main MPC launch from the kernel's own stamps: per XCD (the clock is read per XCD), how many robots are in flight
over time, the launch span, and sum(solve time) / (slots x span)."""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = pkg.Context(0, max(n, 4096), 16)
h = int(sys.argv[4]) if len(sys.argv) > 4 else 10
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
planned = len(sys.argv) > 2 and sys.argv[2] == 'planned'
ctx.set_planned_list(planned)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0xA1 + 2
seq = pkg.make_batch_sequence(n, h, "a1", seed=seed, steps=6)
for lpt in (True, False):
    ctx.set_lpt_schedule(lpt)
    for b in seq:
        out = G.run_mpc(ctx, pkg, b)
    buf = np.zeros((n, 16), np.int64)
    lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    t0, t6 = buf[:, 12], buf[:, 13]                 # 100 MHz wall clock, shared by every CU
    good = t6 > t0
    t0, t6 = t0[good].astype(np.float64), t6[good].astype(np.float64)
    span = t6.max() - t0.min()
    tot = (t6 - t0).sum()
    ts = t0.min() + (np.arange(24) + 0.5) / 24 * span
    inflight = [(int(((t0 <= t) & (t6 > t)).sum())) for t in ts]
    print("lpt %s: robots %d, span %.1f us, mean solve %.1f us, max solve %.1f us, average robots in flight %.0f (512 slots at h <= 11, 256 at h = 16), in flight over time: %s" % (
        lpt, good.sum(), span / 100, (t6 - t0).mean() / 100, (t6 - t0).max() / 100, tot / span, inflight))
    order = np.argsort(t0)
    print("   start times (us) of robots 0, 256, 511, 512, 600, 768, 1023 in start order:", [round((t0[order[k]] - t0.min()) / 100, 1) for k in (0, 256, 511, 512, 600, 768, min(1023, len(order) - 1))])
    nls_all = (buf[:, 7] // 3)[good]
    big = np.where((t6 - t0) > 0)[0][np.argsort(-(t6 - t0))[:4]]
    print("   longest solves: (start us, length us, nls, final q)", [(round((t0[k] - t0.min()) / 100, 1), round((t6[k] - t0[k]) / 100, 1), int(nls_all[k]), int(buf[good][k, 14])) for k in big])
    kbig = big[0]
    ph = np.diff(buf[good][kbig, :7]).astype(np.float64)
    itb = ((out["status"][good][kbig] >> 8) & 0xffff)
    print("   longest robot: phase cycles load %d, H %d, sweep %d, x0 %d, active set %d, out %d | iterations %d | status %d" % (*ph, itb, out["status"][good][kbig] & 0xff))
    late = np.argsort(-t6)[:5]
    print("   last finishers: (start us, length us)", [(round((t0[k] - t0.min()) / 100, 1), round((t6[k] - t0[k]) / 100, 1)) for k in late])
