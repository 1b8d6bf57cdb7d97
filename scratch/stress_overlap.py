"""Stress of the overlapped tick: temporally coherent sequences of 1024-robot batches, every tick of a sequence queued WITHOUT a sync (own output
arrays per tick), every robot of every tick against the threaded CPU oracle; status flags; chained / unchained counts.  One line per sequence."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
import oracle_py as O
O.build()
pkg = load_pkg()
n, K = 1024, 12
worst_f = worst_t = 0.0; nflag = nto = nbad = total = chained = unchained = 0
CASES = [(c.split(":")[0], int(c.split(":")[1])) for c in os.environ["CASES"].split(",")] if os.environ.get("CASES") else [("a1", 10), ("lite3", 10), ("a1", 5)]
for robot, h in CASES:
    for seed in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (11, 12, 13)):
        for ex in (0.3, 1.0, 2.0):
            ctx = pkg.Context(0, n, 16)
            ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), h); ctx.wbc_setup_packed(0, pkg.model_desc(robot))
            ctx.set_torque_epilogue(True, True)
            if os.environ.get("OV", "1") == "1": assert ctx.set_tick_overlap(True)
            seq = pkg.make_batch_sequence(n, h, robot, seed=seed * 100 + int(ex * 10), steps=K, excite=ex)
            S = pkg.to_soa
            prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
            bufs = [dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
                         fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])),
                         force=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
                         qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32))) for b in seq]
            ctx.sync()
            for d in bufs:
                ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], prev, d["force"], d["tau"], d["status"], qdes=d["qdes"])
            ctx.sync()
            st_ = ctx.tick_overlap_stats(); chained += st_[0]; unchained += st_[1]
            sf = sb = sto = 0; wf = wt = 0.0
            for k, (d, b) in enumerate(zip(bufs, seq)):
                prev_in = seq[0]["prev_ori_vel"] if k == 0 else seq[k - 1]["wbc_cmd"][:, 12:15]        # quirk 4: this tick's memory is last tick's vBody_Ori_des
                f, tau, st, sec, pv = O.tick_batch(1, pkg.mpc_cfg(robot), h, pkg.model_desc(robot)[:3], pkg.model_desc(robot), b["mpc_state"], b["traj"], b["gait"],
                                                   b["fb_state"], b["wbc_cmd"], np.ascontiguousarray(prev_in, np.float32).copy(), nthreads=32, epilogue=3)
                status = d["status"].download(); fo = d["force"].download().T; to = d["tau"].download().T
                flags = G.flags(status) != 0
                ok = ~flags & (st == 0)
                ef = (np.abs(fo - f).max(1) / np.maximum(1.0, np.abs(f).max(1)))[ok]
                et = (np.abs(to - tau) / np.maximum(1.0, np.abs(tau))).max(1)[ok]
                if os.environ.get("KINDS") == "1" and flags.any(): print("      tick %d flag kinds:" % k, {hex(int(v)): int((G.flags(status) == v).sum()) for v in np.unique(G.flags(status)) if v})
                sf += int(flags.sum()); sto += int(((G.flags(status) & 0x02000000) != 0).sum()); sb += int((ef > 1e-5).sum() + (et > 1e-4).sum() + (~np.isfinite(to)).any(1).sum())
                wf = max(wf, float(ef.max())); wt = max(wt, float(et.max()))
            nflag += sf; nto += sto; nbad += sb; total += n * K; worst_f = max(worst_f, wf); worst_t = max(worst_t, wt)
            print("%-5s h=%2d seed %d excite %.1f: %d ticks (chained %d, not %d): flagged %d (time-outs %d), over tolerance %d, worst force %.1e torque %.1e" % (
                robot, h, seed, ex, K, st_[0], st_[1], sf, sto, sb, wf, wt), flush=True)
            ctx.close()
print("TOTAL %d robot-ticks on the overlapped form (chained ticks %d, unchained %d): flagged %d (per-robot waits that gave up: %d), over tolerance %d, worst force %.2e, worst torque %.2e" % (
    total, chained, unchained, nflag, nto, nbad, worst_f, worst_t))
