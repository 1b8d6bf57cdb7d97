"""Writes tests/golden/ticklog_a1_h10.qrtl: 12 ticks of 2 A1 robots, horizon 10, inputs from the synthetic workload generator and
outputs from the CPU oracle (oracle/, fp64 WBC) driven statefully -- the WBC memory of tick k is what tick k-1 left.
Run from the repository root:  python tests/golden/make_ticklog.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg          # noqa: E402
import oracle_py as O                  # noqa: E402


def oracle_sequence(pkg, n_robots, ticks, h, robot, seed, excite):
    """-> list of (inputs incl. prev_ori_vel before the tick, force, tau, status)"""
    cfg, md = pkg.mpc_cfg(robot), pkg.model_desc(robot)
    stream = pkg.make_batch(n_robots * ticks, h, robot, seed=seed, excite=excite)        # tick k, robot r = row k * n_robots + r
    prev = np.zeros((n_robots, 3), np.float32)
    out = []
    for k in range(ticks):
        sl = slice(k * n_robots, (k + 1) * n_robots)
        b = {key: stream[key][sl] for key in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd")}
        b["prev_ori_vel"] = prev.copy()
        f, tau, st, _, prev_new = O.tick_batch(1, cfg, h, md[:3], md, b["mpc_state"], b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"], prev.copy())
        out.append((b, f, tau, st))
        prev = prev_new
    return out


def main():
    O.build()
    pkg = load_pkg()
    n, ticks, h, robot = 2, 12, 10, "a1"
    path = os.path.join(ROOT, "tests", "golden", "ticklog_a1_h10.qrtl")
    with pkg.ticklog.TickLogWriter(path, n, h, pkg.mpc_cfg(robot), pkg.ticklog.model15(pkg.model_desc(robot)), robot) as w:
        for b, f, tau, st in oracle_sequence(pkg, n, ticks, h, robot, seed=0x71C, excite=0.8):
            w.append(b, f, tau, st)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
