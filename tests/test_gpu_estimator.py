"""GPU parity of the batched velocity estimator (qrgpu_estimator_update_batch) against the oracle over a tick sequence.
Reference: qr_robot_velocity_estimator.cpp:77-133, qr_robot.cpp:62-72; Kalman step pinned to the compiled TinyEKF on the CPU side.

Bar: leg kinematics within 2e-6 (device sinf/cosf vs libm), everything downstream within 1e-5 absolute (velocities of order 1 m/s);
the filters themselves replay the reference's operations exactly."""
import numpy as np
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
