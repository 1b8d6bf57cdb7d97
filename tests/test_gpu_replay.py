"""BASELINE config 0 in small: ONE A1 robot, horizon 10, a replay of 60 control ticks through the single-robot host interfaces
(MPCInterface = SetupProblem / SolveMPCKernel / GetMPCSolution, WbcLocomotionController::Run) with the reference's scheduling --
MPC every 15th tick and on each of the first 50 (qr_mpc_stance_leg_controller.cpp:342), WBC computing on every second call
(qr_wbc_locomotion_controller.cpp:111,133), Fr_des = the latest MPC forces (:408) -- against the oracle driven the same way.
The state stream is synthetic (the reference ships no recorded data)."""
import numpy as np

import gpu_helpers as G
import pytest

pytestmark = pytest.mark.gpu


def test_single_robot_replay(pkg, oracle):
    h, ticks = 10, 60
    ctx = pkg.Context(device_id=0, max_batch=1, horizon_max=16)
    try:
        cfg, md = pkg.mpc_cfg("a1"), pkg.model_desc("a1")
        mpc = pkg.MPCInterface(ctx)
        mpc.SetupProblem(cfg[0], h, cfg[1], cfg[2], cfg[3], cfg[4:7], cfg[7:19], cfg[19])
        ctx.wbc_setup_packed(0, md)
        wbc = pkg.WbcLocomotionController(ctx)
        stream = pkg.make_batch(ticks, h, "a1", seed=0xC0F, excite=0.5)         # tick k = "robot" k of a synthetic batch
        f_gpu = np.zeros(12); f_cpu = np.zeros(12)
        prev_cpu = np.zeros(3, np.float32); tau_cpu_last = np.zeros(12, np.float32)
        worst_f = worst_t = 0.0
        n_mpc = n_wbc = 0
        for k in range(ticks):
            s = stream["mpc_state"][k]
            if k % 15 == 0 or k < 50:
                mpc.SolveMPCKernel(s[0:3], s[3:6], s[6:10], s[10:13], s[13:25], s[25:28], stream["traj"][k], stream["gait"][k])
                assert G.flags(mpc.status) == 0
                f_gpu = np.array([mpc.GetMPCSolution(i) for i in range(12)])
                u, st, rc = oracle.mpc_solve(cfg, h, s, stream["traj"][k], stream["gait"][k])
                assert rc == 0
                f_cpu = u[:12]
                worst_f = max(worst_f, np.abs(f_gpu - f_cpu).max() / max(1.0, np.abs(f_cpu).max()))
                n_mpc += 1
            cmd = stream["wbc_cmd"][k].copy()
            cmd[51:63] = f_gpu.astype(np.float32)                               # wbcData.Fr_des = f (:408)
            tua = np.zeros(12, np.float32)
            wbc.Run(stream["fb_state"][k], cmd, tua)
            if k % 2 == 0:                                                      # the oracle recomputes on the same cadence, with its own forces
                cmd_o = stream["wbc_cmd"][k].copy(); cmd_o[51:63] = f_cpu.astype(np.float32)
                w = oracle.wbc_run(md, stream["fb_state"][k].astype(np.float64), cmd_o.astype(np.float64), prev_ori_vel=prev_cpu.astype(np.float64),
                                   dtype=np.float64)
                prev_cpu = w["prev_ori_vel"].astype(np.float32)
                tau_cpu_last = w["tau"]
                n_wbc += 1
            stance = np.repeat(stream["wbc_cmd"][k, 63:67] != 0, 3)
            err = np.abs(tua[stance] - tau_cpu_last[stance]) / np.maximum(1.0, np.abs(tau_cpu_last[stance]))
            worst_t = max(worst_t, err.max(initial=0))
            assert np.all(tua[~stance] == 0)                                    # UpdateLegCMD overwrites stance legs only (:205-219)
        assert n_mpc == 50 and n_wbc == 30          # ticks 0..49 solve every tick; 60 would be the next multiple of 15
        assert worst_f <= 1e-5 and worst_t <= 1e-4, (worst_f, worst_t)
    finally:
        ctx.close()


def test_config0_2000_tick_coherent_sequence(pkg):
    """BASELINE.json configs[0] at the size SURVEY.md 8d(1) gives it: ONE A1 robot, horizon 10, 2 000 consecutive control ticks of a temporally
    coherent stream (tests/golden/make_config0.py: the same robot 2 ms later, the contact table scrolling through trot and the all-stance
    hand-overs) through the single-robot drop-in interfaces -- MPCInterface (SetupProblem / SolveMPCKernel / GetMPCSolution) on the
    reference's cadence (qr_mpc_stance_leg_controller.cpp:342), WbcLocomotionController::Run on every tick (computing on every second,
    qr_wbc_locomotion_controller.cpp:111) -- statefully, against the committed outputs of the CPU oracle driven the same way
    (tests/golden/config0_a1_h10_2000.npz: outputs only, the inputs are regenerated from the seed).  Warm starts carry over from solve to
    solve here as on a robot."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("make_config0", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_config0.py"))
    M = importlib.util.module_from_spec(spec); spec.loader.exec_module(M)
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config0_a1_h10_2000.npz"))
    P = M.PARAMS
    assert [int(x) for x in fx["params"]] == [P["horizon"], P["ticks"], P["seed"]]
    seq = M.stream(pkg)
    chk = np.array([sum(float(np.abs(b["mpc_state"]).sum()) for b in seq), sum(float(np.abs(b["fb_state"]).sum()) for b in seq), sum(float(b["gait"].sum()) for b in seq)])
    assert np.allclose(chk, fx["input_checksums"], rtol=1e-12), "the workload generator no longer reproduces the fixture's inputs"
    h = P["horizon"]
    ctx = pkg.Context(device_id=0, max_batch=1, horizon_max=16)
    try:
        cfg, md = pkg.mpc_cfg("a1"), pkg.model_desc("a1")
        mpc = pkg.MPCInterface(ctx)
        mpc.SetupProblem(cfg[0], h, cfg[1], cfg[2], cfg[3], cfg[4:7], cfg[7:19], cfg[19])
        ctx.wbc_setup_packed(0, md)
        wbc = pkg.WbcLocomotionController(ctx)
        f_gpu = np.zeros(12)
        worst_f = worst_t = 0.0
        n_mpc = 0
        for k, b in enumerate(seq):
            s = b["mpc_state"][0]
            if M.mpc_tick(k):
                mpc.SolveMPCKernel(s[0:3], s[3:6], s[6:10], s[10:13], s[13:25], s[25:28], b["traj"][0], b["gait"][0])
                assert G.flags(mpc.status) == 0, k
                f_gpu = np.array([mpc.GetMPCSolution(i) for i in range(12)])
                f_ref = fx["mpc_forces"][n_mpc]
                worst_f = max(worst_f, np.abs(f_gpu - f_ref).max() / max(1.0, np.abs(f_ref).max()))
                n_mpc += 1
            cmd = b["wbc_cmd"][0].copy()
            cmd[51:63] = f_gpu.astype(np.float32)                               # wbcData.Fr_des = f (:408)
            tua = np.zeros(12, np.float32)
            wbc.Run(b["fb_state"][0], cmd, tua)
            assert G.flags(wbc.status) == 0, k
            ref = fx["leg_cmd_tua"][k]
            worst_t = max(worst_t, (np.abs(tua - ref) / np.maximum(1.0, np.abs(ref))).max())
        assert n_mpc == len(fx["mpc_forces"]) == 180
        assert worst_f <= 1e-5 and worst_t <= 1e-4, (worst_f, worst_t)          # north_star's tolerance on the torque, 1e-5 of the force scale on the forces
    finally:
        ctx.close()
