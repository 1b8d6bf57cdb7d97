"""CPU: the oracle's velocity estimator (qr_robot_velocity_estimator.cpp:77-133) -- its Kalman step pinned against the reference's own
TinyEKF<3,3> compiled from /root/reference (oracle/_ref), the rest by properties."""
import numpy as np


def test_ekf_matches_reference_tinyekf(oracle, ref):
    rng = np.random.default_rng(4)
    for acc_var, sen_var in ((0.1, 0.1), (0.01, 0.5), (1.0, 0.02)):
        dv = 0.01 * rng.standard_normal((400, 3)); z = np.cumsum(dv, 0) + 0.05 * rng.standard_normal((400, 3))
        ours, bad = oracle.ekf3_run(np.float32(acc_var), np.float32(sen_var), dv, z)
        theirs, bad_ref = oracle.ref_tinyekf_run(acc_var, sen_var, dv, z)
        assert bad == 0 and bad_ref == 0
        assert np.array_equal(ours, theirs)                       # same operations in the same order: bit for bit


def test_estimator_properties(pkg, oracle):
    W = pkg.workload
    cfg = W.estimator_cfg("a1", window=16)
    x, stamp = W.make_estimator_sequence(4, 120, seed=9)
    for r in range(4):
        out = oracle.estimator_run(cfg, x[:, r], stamp[:, r])
        assert np.all(np.isfinite(out))
        # filtered acceleration = plain mean of the last <= 20 samples (the Neumaier sum is exact to rounding)
        for k in (0, 5, 19, 20, 57, 119):
            lo = max(0, k - 19)
            np.testing.assert_allclose(out[k, 0:3], x[lo:k + 1, r, 3:6].astype(np.float64).mean(0), rtol=0, atol=2e-6)
        # world/base velocity are related by the attitude; angular velocity is the rotated gyro
        for k in (3, 60, 119):
            q = x[k, r, 6:10].astype(np.float64); w_, a, b, c = q
            R = np.array([[1 - 2 * (b * b + c * c), 2 * (a * b - w_ * c), 2 * (a * c + w_ * b)],
                          [2 * (a * b + w_ * c), 1 - 2 * (a * a + c * c), 2 * (b * c - w_ * a)],
                          [2 * (a * c - w_ * b), 2 * (b * c + w_ * a), 1 - 2 * (a * a + b * b)]])
            np.testing.assert_allclose(out[k, 6:9], R.T @ out[k, 3:6], atol=1e-6)
            np.testing.assert_allclose(out[k, 9:12], R @ x[k, r, 10:13], atol=1e-6)
        # foot velocity = J dq, checked by finite differences of the foot position map
        k = 50
        geom = cfg[:3]; ho = cfg[7:19]
        q0 = x[k, r, 17:29].astype(np.float32); dq = x[k, r, 29:41]
        h = 1e-3
        fd = (oracle.foot_positions(geom, ho, (q0 + h * dq).astype(np.float32)) - oracle.foot_positions(geom, ho, (q0 - h * dq).astype(np.float32))) / (2 * h)
        np.testing.assert_allclose(out[k, 24:36], fd.reshape(-1), atol=2e-3)
        np.testing.assert_allclose(out[k, 12:24], oracle.foot_positions(geom, ho, q0).reshape(-1), atol=1e-6)


def test_estimator_constant_velocity_converges(pkg, oracle):
    """Stance on all four feet, base moving at constant velocity, noiseless: the estimate converges to it."""
    W = pkg.workload
    cfg = W.estimator_cfg("a1", window=30)
    T = 400
    x = np.zeros((T, 41), np.float32)
    x[:, 6] = 1.0                                     # identity attitude
    x[:, 0:3] = (0, 0, 9.81)                          # accelerometer reads gravity only
    x[:, 13:17] = 1
    x[:, 17:29] = np.tile([0.0, 0.9, -1.8], 4)
    geom = cfg[:3]
    v_body = np.array([0.4, -0.1, 0.0])
    # joint rates that make every stance foot move at -v_body in the base frame: dq = J^-1 (-v)
    for leg in range(4):
        J = oracle.leg_jacobian(geom, x[0, 17 + 3 * leg:20 + 3 * leg], leg).reshape(3, 3)
        x[:, 29 + 3 * leg:32 + 3 * leg] = np.linalg.solve(J.astype(np.float64), -v_body)
    stamp = (1000 + 2 * np.arange(T)).astype(np.uint32)
    out = oracle.estimator_run(cfg, x, stamp)
    np.testing.assert_allclose(out[-1, 3:6], v_body, atol=2e-3)
    np.testing.assert_allclose(out[-1, 6:9], v_body, atol=2e-3)
