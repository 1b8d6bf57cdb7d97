"""Slot utilisation of the main MPC launch: sum of per-robot solve times / (resident slots x launch span), from the kernel's own stamps."""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = pkg.Context(0, max(n, 4096), 16)
h = 10
G.setup_a1(ctx, pkg, h)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
ctx.set_planned_list(False)
names = ["load+srbd", "H/g build", "sweep inv", "x0", "active set", "out"]
for seed in (0xA1 + 2, 0xA1 + 2 + 1000):
    seq = pkg.make_batch_sequence(n, h, "a1", seed=seed, steps=6)
    for warm, lpt in ((True, True), (True, False), (False, True)):
        ctx.set_warm_start(warm); ctx.set_lpt_schedule(lpt)
        for b in seq:
            out = G.run_mpc(ctx, pkg, b)
        buf = np.zeros((n, 16), np.int64)
        lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
        ok = (out["status"] & 0x4) == 0
        t0, t6 = buf[ok, 0], buf[ok, 6]
        tot = (t6 - t0).astype(np.float64)
        span = t6.max() - t0.min()
        d = np.diff(buf[ok, :7], axis=1).astype(np.float64)
        print("seed %x warm %s lpt %s: robots %d, per-robot total mean %.0f max %.0f, span %.0f, slots*span/sum = %.2f (1 = perfect packing on 512 slots), phases mean %s" % (
            seed, warm, lpt, ok.sum(), tot.mean(), tot.max(), span, 512.0 * span / tot.sum(), dict(zip(names, d.mean(0).round(0)))))
        # start-time histogram: how many robots start in the first 5% of the span
        st = (t0 - t0.min()) / span
        print("   robots started in the first 2%% of the span: %d; last start at %.2f of the span; longest robot started at %.2f" % ((st < 0.02).sum(), st.max(), st[np.argmax(tot)]))
