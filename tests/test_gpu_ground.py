"""GPU parity of the ground-plane estimator kernel (qrgpu_ground_update_batch) against the oracle over tick sequences.
Reference: qrGroundSurfaceEstimator::Update / GetNormalVector / ComputeControlFrame (qr_ground_surface_estimator.cpp:40-70,151-206).
Bar: the same double arithmetic on both sides -- 1e-6 absolute on every output (float outputs; atan2 / asin / sincos differ by an ulp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_sequences(n, T, seed):
    rng = np.random.default_rng(seed)
    hips = np.array([(0.18, -0.13), (0.18, 0.13), (-0.18, -0.13), (-0.18, 0.13)])
    x = np.zeros((T, n, 23), np.float32)
    # contacts: trot-like toggling with random phase, so that "all four down, one newly" happens at irregular times
    ph = rng.uniform(0, 1, (n, 4)); per = rng.integers(6, 14, (n, 1))
    for t in range(T):
        x[t, :, 0:4] = (np.fmod(ph + t / per, 1.0) < 0.8)
    plane = np.stack([rng.uniform(-0.35, -0.2, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)], 1)
    for l in range(4):
        fx = hips[l, 0] + rng.uniform(-0.06, 0.06, (T, n)); fy = hips[l, 1] + rng.uniform(-0.04, 0.04, (T, n))
        x[:, :, 4 + 3 * l] = fx; x[:, :, 5 + 3 * l] = fy
        x[:, :, 6 + 3 * l] = plane[:, 0] + plane[:, 1] * fx + plane[:, 2] * fy + rng.uniform(-0.01, 0.01, (T, n))
    x[:, :, 16:19] = rng.uniform(-1, 1, (T, n, 3))
    # orientation: a slowly turning base with some roll and pitch (yaw crosses +-pi for some robots: the reference filters the angles as they are)
    yaw = rng.uniform(-3.1, 3.1, (1, n)) + np.cumsum(rng.uniform(-0.05, 0.08, (T, n)), 0)
    roll = rng.uniform(-0.3, 0.3, (T, n)); pitch = rng.uniform(-0.3, 0.3, (T, n))
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    x[:, :, 19] = cr * cp * cy + sr * sp * sy; x[:, :, 20] = sr * cp * cy - cr * sp * sy
    x[:, :, 21] = cr * sp * cy + sr * cp * sy; x[:, :, 22] = cr * cp * sy - sr * sp * cy
    return x


def test_ground_sequence_vs_oracle(gpu_ctx, pkg, oracle):
    n, T = 300, 60
    x = make_sequences(n, T, seed=11)
    d_in = gpu_ctx.alloc((23, n)); d_out = gpu_ctx.alloc((32, n))
    d_st = gpu_ctx.alloc((13, n), np.float64).upload(np.full((13, n), np.nan))          # reset must not depend on what was there
    d_est = gpu_ctx.alloc((54, n)).upload(np.full((54, n), -7.0, np.float32))
    outs = np.zeros((T, n, 32), np.float32)
    for t in range(T):
        d_in.upload(pkg.to_soa(x[t]))
        gpu_ctx.ground_update_batch(n, d_in, d_st, d_out, d_est, reset=(t == 0))
        outs[t] = d_out.download().T
    est = d_est.download().T
    fired = 0
    for r in range(n):
        o = oracle.ground_run(x[:, r])
        assert np.array_equal(outs[:, r, 31], o[:, 31]), r                     # the same ticks fire
        fired += int(o[:, 31].sum())
        err = np.abs(outs[:, r, :31].astype(np.float64) - o[:, :31])
        assert err.max() < 1e-6, (r, np.unravel_index(err.argmax(), err.shape), err.max())
        assert np.array_equal(est[r, 45:54], outs[-1, r, 13:22]) and np.all(est[r, :45] == -7.0)
    assert fired > 5 * n                                                        # the batch exercised the fit many times per robot
    for v in (d_in, d_out, d_st, d_est):
        v.free()


def test_ground_without_outputs_keeps_state(gpu_ctx, pkg, oracle):
    """d_ground_out = NULL: the state still advances (the next call with outputs agrees with the oracle's second tick)."""
    n = 64
    x = make_sequences(n, 2, seed=2)
    x[0, :, 0] = 0; x[1, :, 0:4] = 1                                            # tick 1 fires for everybody
    d_in = gpu_ctx.alloc((23, n)); d_out = gpu_ctx.alloc((32, n)); d_st = gpu_ctx.alloc((13, n), np.float64)
    d_in.upload(pkg.to_soa(x[0])); gpu_ctx.ground_update_batch(n, d_in, d_st, None, None, reset=True)
    d_in.upload(pkg.to_soa(x[1])); gpu_ctx.ground_update_batch(n, d_in, d_st, d_out, None)
    g = d_out.download().T
    for r in range(n):
        o = oracle.ground_run(x[:, r])
        assert g[r, 31] == 1 and np.abs(g[r, :31].astype(np.float64) - o[1, :31]).max() < 1e-6
    for v in (d_in, d_out, d_st):
        v.free()
