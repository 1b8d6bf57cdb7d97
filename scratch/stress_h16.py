"""Stress at horizon 16 (BASELINE config 5 in small): mixed A1 + Lite3 batches, every robot against the threaded CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as O
O.build(); pkg = load_pkg()
n, h = 2048, 16
ctx = pkg.Context(0, n, 16)
for r, t in (("a1", 0), ("lite3", 1)):
    ctx.mpc_setup_packed(t, pkg.mpc_cfg(r), h); ctx.wbc_setup_packed(t, pkg.model_desc(r))
tot_flag = tot_bad = 0
for seed in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (21, 22)):
    for ex in (0.3, 1.0):
        ba = pkg.make_batch(n // 2, h, "a1", seed=seed * 10 + 1, excite=ex); bl = pkg.make_batch(n // 2, h, "lite3", seed=seed * 10 + 2, excite=ex)
        def il(a, c):
            out = np.empty((a.shape[0] * 2,) + a.shape[1:], a.dtype); out[0::2] = a; out[1::2] = c; return out
        b = {k: (il(ba[k], bl[k]) if isinstance(ba[k], np.ndarray) else ba[k]) for k in ba}
        b["n"] = n
        tid = np.tile(np.array([0, 1], np.int32), n // 2)
        with G.cold_start(ctx):
            o = G.run_tick(ctx, pkg, b, type_id=tid)
        ow = G.run_tick(ctx, pkg, b, type_id=tid)      # warm start from the previous batch's slots (stale), then from its own
        ow = G.run_tick(ctx, pkg, b, type_id=tid)
        f = np.zeros((n, 12), np.float32); tau = np.zeros((n, 12), np.float32); st = np.zeros(n, np.int32)
        for r, t in (("a1", 0), ("lite3", 1)):
            m = tid == t
            fr, tr, sr, _, _ = O.tick_batch(1, pkg.mpc_cfg(r), h, pkg.model_desc(r)[:3], pkg.model_desc(r), b["mpc_state"][m], b["traj"][m], b["gait"][m],
                                            b["fb_state"][m], b["wbc_cmd"][m], b["prev_ori_vel"][m].copy(), nthreads=32)
            f[m], tau[m], st[m] = fr, tr, sr
        flags = G.flags(o["status"]) != 0
        ok = ~flags & (st == 0)
        ef = (np.abs(o["force"] - f).max(1) / np.maximum(1.0, np.abs(f).max(1)))[ok]; et = (np.abs(o["tau"] - tau) / np.maximum(1.0, np.abs(tau))).max(1)[ok]
        bad = int((ef > 1e-5).sum() + (et > 1e-4).sum()); tot_flag += int(flags.sum()); tot_bad += bad
        fw = G.flags(ow["status"]) != 0
        okw = ~fw & (st == 0)
        efw = (np.abs(ow["force"] - f).max(1) / np.maximum(1.0, np.abs(f).max(1)))[okw]; etw = (np.abs(ow["tau"] - tau) / np.maximum(1.0, np.abs(tau))).max(1)[okw]
        badw = int((efw > 1e-5).sum() + (etw > 1e-4).sum()); tot_flag += int(fw.sum()); tot_bad += badw
        print("h=16 seed %d excite %.1f: cold flagged %d %s (oracle nonzero %d), max rel force err %.2e, torque %.2e, over tol %d, iters max %d | warm flagged %d, force %.2e, torque %.2e, over tol %d, iters mean %.1f (cold %.1f)"
              % (seed, ex, flags.sum(), np.unique(G.flags(o["status"][flags])), (st != 0).sum(), ef.max(), et.max(), bad, ((o["status"] >> 8) & 0xffff).max(),
                 fw.sum(), efw.max(), etw.max(), badw, ((ow["status"] >> 8) & 0xffff).mean(), ((o["status"] >> 8) & 0xffff).mean()), flush=True)
print("TOTAL flagged %d over tolerance %d" % (tot_flag, tot_bad))
