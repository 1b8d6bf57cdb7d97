#!/usr/bin/env python3
"""Replay a tick log (include/qrgpu_ticklog.h) through the GPU path and compare with the recorded forces and torques.

    python tools/qr_replay.py info   run.qrtl
    python tools/qr_replay.py replay run.qrtl [--stateful] [--device 0] [--tol-tau 1e-4] [--tol-force 1e-5]
    python tools/qr_replay.py record-synthetic out.qrtl --robots 64 --ticks 100 [--horizon 10] [--robot a1]   (outputs = this library's)

Exit code 0 when every unflagged robot-tick is inside the tolerances, 1 otherwise.  Needs an MI355X and the built library
(no CPU fallback: without them it fails loudly)."""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_pkg():
    d = os.path.join(ROOT, "quadruped-robot_amd")
    spec = importlib.util.spec_from_file_location("quadruped_robot_amd", os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules["quadruped_robot_amd"] = m
    spec.loader.exec_module(m)
    return m


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("info"); a.add_argument("log")
    a = sub.add_parser("replay"); a.add_argument("log"); a.add_argument("--stateful", action="store_true"); a.add_argument("--device", type=int, default=0)
    a.add_argument("--tol-tau", type=float, default=1e-4); a.add_argument("--tol-force", type=float, default=1e-5)
    a = sub.add_parser("record-synthetic"); a.add_argument("log"); a.add_argument("--robots", type=int, default=64); a.add_argument("--ticks", type=int, default=100)
    a.add_argument("--horizon", type=int, default=10); a.add_argument("--robot", default="a1"); a.add_argument("--seed", type=int, default=1); a.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    pkg = _load_pkg()
    if args.cmd == "info":
        r = pkg.ticklog.TickLogReader(args.log)
        print("%s: %d ticks x %d robots, horizon %d, robot '%s', %d bytes per tick" % (args.log, r.ticks, r.n_robots, r.horizon, r.robot,
                                                                                         4 * r.n_robots * pkg.ticklog.words_per_robot(r.horizon)))
        print("mpc_cfg: dt %.4g mu %.3g fmax %.4g mass %.4g alpha %.3g" % (r.mpc_cfg[0], r.mpc_cfg[1], r.mpc_cfg[2], r.mpc_cfg[3], r.mpc_cfg[19]))
        return 0
    if args.cmd == "replay":
        r = pkg.ticklog.TickLogReader(args.log)
        ctx = pkg.Context(device_id=args.device, max_batch=r.n_robots, horizon_max=max(16, r.horizon))
        try:
            pkg.replay.setup_from_log(ctx, r)
            res = pkg.replay.replay(ctx, r, stateful=args.stateful)
        finally:
            ctx.close()
        print("%d robot-ticks: worst force error %.3e, worst torque error %.3e (relative), flagged %d (recorded %d)"
              % (res["robot_ticks"], res["worst_force"], res["worst_tau"], res["flagged"], res["recorded_flagged"]))
        ok = res["worst_force"] <= args.tol_force and res["worst_tau"] <= args.tol_tau
        print("PASS" if ok else "FAIL (tolerances: force %.1e, torque %.1e)" % (args.tol_force, args.tol_tau))
        return 0 if ok else 1
    ctx = pkg.Context(device_id=args.device, max_batch=args.robots, horizon_max=max(16, args.horizon))
    try:
        ctx.mpc_setup_packed(0, pkg.mpc_cfg(args.robot), args.horizon); ctx.wbc_setup_packed(0, pkg.model_desc(args.robot))
        stream = pkg.make_batch(args.robots * args.ticks, args.horizon, args.robot, seed=args.seed)
        keys = ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel")
        batches = [dict({k: stream[k][t * args.robots:(t + 1) * args.robots] for k in keys}, n=args.robots, horizon=args.horizon) for t in range(args.ticks)]
        pkg.replay.record(ctx, args.log, batches, pkg.mpc_cfg(args.robot), pkg.model_desc(args.robot), args.robot)
    finally:
        ctx.close()
    print("wrote", args.log)
    return 0


if __name__ == "__main__":
    sys.exit(main())
