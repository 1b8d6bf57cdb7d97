"""BASELINE.json configs[0] as SURVEY.md 8d(1) sizes it: ONE A1 robot, horizon 10, 2 000 consecutive control ticks (dt = 0.002 s) of a
temporally coherent state stream, driven with the reference's cadence -- the MPC re-solves on every tick of the first 50 and on every 15th
after that (qr_mpc_stance_leg_controller.cpp:342), the WBC computes on every second call and re-applies its last torques on the others
(qr_wbc_locomotion_controller.cpp:111,133), Fr_des = the latest MPC forces (:408), UpdateLegCMD overwrites stance legs only (:205-219).

The reference ships no recorded data: the state stream is workload.make_batch_sequence (seeded; the same robot 2 ms later:
position, attitude and joints integrate their rates, the gait phase advances so that the contact table scrolls).  The fixture holds the
generator's parameters and the CPU oracle's OUTPUTS only (forces of every MPC solve, the leg command of every tick, fp64 WBC) -- inputs are
regenerated from the seed by the test.   python tests/golden/make_config0.py   ->  tests/golden/config0_a1_h10_2000.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg          # noqa: E402
import oracle_py as O                  # noqa: E402

PARAMS = dict(robot="a1", horizon=10, ticks=2000, seed=0xC0F0, dt=0.002, excite=0.6)


def stream(pkg, p=PARAMS):
    """The 2 000 input rows (one robot): list index = control tick."""
    return pkg.make_batch_sequence(1, p["horizon"], p["robot"], seed=p["seed"], steps=p["ticks"], dt=p["dt"], frac_all_stance=0.0, frac_three_leg=0.0,
                                   excite=p["excite"])


def mpc_tick(k):
    return k % 15 == 0 or k < 50


def drive_oracle(pkg, seq, p=PARAMS):
    cfg, md = pkg.mpc_cfg(p["robot"]), pkg.model_desc(p["robot"])
    h = p["horizon"]
    f_cpu = np.zeros(12)
    prev = np.zeros(3)
    tau_last = np.zeros(12)
    forces, tua_all, n_active, iters = [], [], [], []
    for k, b in enumerate(seq):
        if mpc_tick(k):
            u, st, rc = O.mpc_solve(cfg, h, b["mpc_state"][0], b["traj"][0], b["gait"][0])
            assert rc == 0, (k, rc)
            f_cpu = u[:12].copy()
            forces.append(f_cpu); n_active.append(st["n_active"]); iters.append(st["iters"])
        cmd = b["wbc_cmd"][0].copy()
        cmd[51:63] = f_cpu.astype(np.float32)
        if k % 2 == 0:
            w = O.wbc_run(md, b["fb_state"][0].astype(np.float64), cmd.astype(np.float64), prev_ori_vel=prev, dtype=np.float64)
            assert w["rc"] == 0, k
            prev = w["prev_ori_vel"]
            tau_last = w["tau"]
        tua = np.zeros(12)
        stance = np.repeat(cmd[63:67] != 0, 3)
        tua[stance] = tau_last[stance]
        tua_all.append(tua)
    return np.array(forces), np.array(tua_all), np.array(n_active), np.array(iters)


def main():
    O.build()
    pkg = load_pkg()
    seq = stream(pkg)
    forces, tua, n_active, iters = drive_oracle(pkg, seq)
    # two checksums of the regenerated inputs, so that a changed generator is noticed before it is blamed on the kernels
    chk = np.array([sum(float(np.abs(b["mpc_state"]).sum()) for b in seq), sum(float(np.abs(b["fb_state"]).sum()) for b in seq),
                    sum(float(b["gait"].sum()) for b in seq)])
    path = os.path.join(ROOT, "tests", "golden", "config0_a1_h10_2000.npz")
    np.savez_compressed(path, params=np.array([PARAMS["horizon"], PARAMS["ticks"], PARAMS["seed"]]), dt_excite=np.array([PARAMS["dt"], PARAMS["excite"]]),
                        mpc_forces=forces, leg_cmd_tua=tua.astype(np.float32), mpc_n_active=n_active.astype(np.int16), mpc_iters=iters.astype(np.int16),
                        input_checksums=chk)
    print(path, os.path.getsize(path) // 1024, "KiB;", len(forces), "MPC solves, working sets of up to", n_active.max(), "rows;",
          "contact patterns seen:", sorted({tuple(int(x) for x in b["wbc_cmd"][0, 63:67]) for b in seq}))


if __name__ == "__main__":
    main()
