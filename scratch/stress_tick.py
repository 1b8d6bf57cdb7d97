"""Stress: full MPC+WBC ticks of 4096-robot batches over seeds and excitation levels against the threaded CPU oracle, every robot;
run-to-run determinism; status flags.  Prints one line per batch and a summary."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as O
O.build()
pkg = load_pkg()
n = 4096
ctx = pkg.Context(0, n, 16)
worst_f = worst_t = 0.0; nflag = 0; nbad = 0; total = 0
for robot, h in (("a1", 10), ("lite3", 10), ("a1", 5)):
    ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), h); ctx.wbc_setup_packed(0, pkg.model_desc(robot))
    for seed in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (11, 12, 13)):
        for ex in (0.3, 1.0, 2.0):
            b = pkg.make_batch(n, h, robot, seed=seed * 100 + int(ex * 10), excite=ex)
            with G.cold_start(ctx):
                o1 = G.run_tick(ctx, pkg, b); o2 = G.run_tick(ctx, pkg, b)
            det = np.array_equal(o1["tau"], o2["tau"]) and np.array_equal(o1["force"], o2["force"])
            ws = G.run_tick(ctx, pkg, b)          # warm start from whatever the previous batch left in the slots (a stale guess)
            we = G.run_tick(ctx, pkg, b)          # warm start from this batch's own working sets
            f, tau, st, sec, prev = O.tick_batch(1, pkg.mpc_cfg(robot), h, pkg.model_desc(robot)[:3], pkg.model_desc(robot), b["mpc_state"], b["traj"], b["gait"],
                                                 b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=32)
            line = []
            for name, o in (("cold", o1), ("stale", ws), ("warm", we)):
                flags = G.flags(o["status"]) != 0
                ok = ~flags & (st == 0)
                ef = (np.abs(o["force"] - f).max(1) / np.maximum(1.0, np.abs(f).max(1)))[ok]
                et = (np.abs(o["tau"] - tau) / np.maximum(1.0, np.abs(tau))).max(1)[ok]
                bad = int((ef > 1e-5).sum() + (et > 1e-4).sum())
                worst_f = max(worst_f, ef.max()); worst_t = max(worst_t, et.max()); nflag += int(flags.sum()); nbad += bad; total += n
                line.append("%s: flagged %d %s, force %.1e, torque %.1e, over tol %d, iters mean %.1f max %d" % (
                    name, flags.sum(), dict(zip(*[x.tolist() for x in np.unique(G.flags(o["status"][flags]), return_counts=True)])) if flags.any() else "", ef.max(), et.max(), bad, ((o["status"] >> 8) & 0xffff).mean(), ((o["status"] >> 8) & 0xffff).max()))
            print("%-5s h=%2d seed %d excite %.1f (oracle nonzero %d, cold runs bit-identical %s) | %s" % (robot, h, seed, ex, (st != 0).sum(), det, " | ".join(line)), flush=True)
print("TOTAL %d robot-ticks: flagged %d, over tolerance %d, worst force %.2e, worst torque %.2e" % (total, nflag, nbad, worst_f, worst_t))
