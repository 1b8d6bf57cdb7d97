"""GPU parity of the batched velocity estimator (qrgpu_estimator_update_batch) against the oracle over a tick sequence.
Reference: qr_robot_velocity_estimator.cpp:77-133, qr_robot.cpp:62-72; Kalman step pinned to the compiled TinyEKF on the CPU side.

Bar: leg kinematics within 2e-6 (device sinf/cosf vs libm), everything downstream within 1e-5 absolute (velocities of order 1 m/s);
the filters themselves replay the reference's operations exactly."""
import numpy as np

import gpu_helpers as G
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("window,ticks", [(120, 150), (8, 60)])
def test_estimator_sequence_parity(gpu_ctx, pkg, oracle, window, ticks):
    W = pkg.workload
    n = 300
    cfg = W.estimator_cfg("a1", window=window)
    x, stamp = W.make_estimator_sequence(n, ticks, seed=window)
    S = gpu_ctx.estimator_state_doubles(window)
    assert S == 96 + 3 * window
    d_state = gpu_ctx.alloc((S, n), np.float64).upload(np.zeros((S, n)))
    d_in = gpu_ctx.alloc((54, n)); d_tick = gpu_ctx.alloc((n,), np.uint32); d_out = gpu_ctx.alloc((42, n))
    outs = []
    for k in range(ticks):
        d_in.upload(pkg.to_soa(x[k])); d_tick.upload(stamp[k])
        gpu_ctx.estimator_update_batch(n, cfg, d_in, d_tick, d_state, d_out)
        if k % 10 == 9 or k == ticks - 1 or k < 3:
            outs.append((k, d_out.download().T.copy()))
    gpu_ctx.sync()
    ref = [oracle.estimator_run(cfg, x[:, r], stamp[:, r]) for r in range(0, n, 7)]
    for k, o in outs:
        for j, r in enumerate(range(0, n, 7)):
            e = ref[j][k]
            assert np.abs(o[r, 12:24] - e[12:24]).max() <= 2e-6, (k, r)                       # foot positions
            assert np.abs(o[r, 24:36] - e[24:36]).max() <= 2e-5 * max(1.0, np.abs(e[24:36]).max())   # J dq
            assert np.array_equal(o[r, 0:3], e[0:3]), (k, r)                                  # acceleration window: no kinematics involved
            assert np.abs(o[r, 3:12] - e[3:12]).max() <= 1e-5, (k, r, np.abs(o[r, 3:12] - e[3:12]).max())
            # pose estimator: odometry, stance-foot height (NaN where no foot stands), absolute height, yaw
            assert np.allclose(o[r, 36:42], e[36:42], rtol=0, atol=2e-5, equal_nan=True), (k, r, o[r, 36:42], e[36:42])
    for v in (d_state, d_in, d_tick, d_out):
        v.free()


def test_sensor_to_torque_pipeline_stays_on_device(gpu_ctx, pkg, oracle):
    """Raw sensor rows -> estimator kernel -> state packing kernel -> MPC+WBC tick, no host copy in between: the torques equal the
    oracle's tick on the state the oracle's estimator produces from the same sensor stream."""
    W = pkg.workload
    n, h, ticks = 96, 10, 40
    cfg = W.estimator_cfg("a1", window=16)
    x, stamp = W.make_estimator_sequence(n, ticks, seed=33)
    x[:, :, 45:54] = np.eye(3, dtype=np.float32).reshape(-1)               # level ground
    gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); gpu_ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    b = pkg.make_batch(n, h, "a1", seed=34, excite=0.3)                     # trajectory, contact table and WBC commands of the tick
    S = pkg.to_soa
    d_state = gpu_ctx.alloc((gpu_ctx.estimator_state_doubles(16), n), np.float64).upload(np.zeros((gpu_ctx.estimator_state_doubles(16), n)))
    d_in = gpu_ctx.alloc((54, n)); d_tick = gpu_ctx.alloc((n,), np.uint32); d_est = gpu_ctx.alloc((42, n))
    for k in range(ticks):
        d_in.upload(S(x[k])); d_tick.upload(stamp[k])
        gpu_ctx.estimator_update_batch(n, cfg, d_in, d_tick, d_state, d_est)
    rpy = b["mpc_state"][:, 25:28].copy()
    d_rpy = gpu_ctx.alloc((3, n)).upload(S(rpy))
    d_mpc = gpu_ctx.alloc((28, n)); d_fb = gpu_ctx.alloc((37, n))
    com = np.asarray(W.ROBOTS["a1"]["com_offset"], np.float32)
    gpu_ctx.pack_state_batch(n, com, d_in, d_est, d_rpy, d_mpc, d_fb)
    d = dict(traj=gpu_ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=gpu_ctx.alloc((4 * h, n)).upload(S(b["gait"])),
             cmd=gpu_ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=gpu_ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
             force=gpu_ctx.alloc((12, n)), tau=gpu_ctx.alloc((12, n)), status=gpu_ctx.alloc((n,), np.int32))
    gpu_ctx.tick_batch(n, d_mpc, d["traj"], d["gait"], d_fb, d["cmd"], d["prev"], d["force"], d["tau"], d["status"])
    gpu_ctx.sync()
    tau = d["tau"].download().T; status = d["status"].download()
    mpc_g = d_mpc.download().T; fb_g = d_fb.download().T
    # the same chain on the CPU: oracle estimator -> packing restated here -> oracle tick
    mpc_o = np.zeros((n, 28), np.float32); fb_o = np.zeros((n, 37), np.float32)
    for r in range(n):
        e = oracle.estimator_run(cfg, x[:, r], stamp[:, r])[-1]
        q4 = x[-1, r, 6:10].astype(np.float32); w_, a, b_, c_ = q4
        R = np.array([[1 - 2 * (b_ * b_ + c_ * c_), 2 * (a * b_ - w_ * c_), 2 * (a * c_ + w_ * b_)],
                      [2 * (a * b_ + w_ * c_), 1 - 2 * (a * a + c_ * c_), 2 * (b_ * c_ - w_ * a)],
                      [2 * (a * c_ - w_ * b_), 2 * (b_ * c_ + w_ * a), 1 - 2 * (a * a + b_ * b_)]], np.float32)
        mpc_o[r, 0:3] = e[36:39]; mpc_o[r, 3:6] = e[3:6]; mpc_o[r, 6:10] = q4; mpc_o[r, 10:13] = e[9:12]; mpc_o[r, 25:28] = rpy[r]
        for leg in range(4):
            mpc_o[r, 13 + 3 * leg:16 + 3 * leg] = R @ (e[12 + 3 * leg:15 + 3 * leg] - com)
        fb_o[r, 0:4] = q4; fb_o[r, 4:7] = e[36:39]; fb_o[r, 7:10] = x[-1, r, 10:13]; fb_o[r, 10:13] = e[6:9]
        fb_o[r, 13:25] = x[-1, r, 17:29]; fb_o[r, 25:37] = x[-1, r, 29:41]
    assert np.abs(mpc_g - mpc_o).max() <= 2e-5 and np.abs(fb_g - fb_o).max() <= 2e-5
    # torques: GPU tick on the GPU-packed state against the oracle tick on that same state (the estimator difference is tested above)
    f_o, tau_o, st_o, _, _ = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), mpc_g, b["traj"], b["gait"], fb_g,
                                               b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=4)
    ok = (G.flags(status) == 0) & (st_o == 0)
    assert ok.sum() >= n - 2
    assert np.all(np.abs(tau[ok] - tau_o[ok]) <= 1e-4 * np.maximum(1.0, np.abs(tau_o[ok])))
