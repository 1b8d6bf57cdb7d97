"""GPU parity of the swing-leg target kernel (qrgpu_swing_targets_batch) against the oracle.
Reference: qrRaibertSwingLegController::GetAction, ADVANCED_TROT case (qr_swing_leg_controller.cpp:362-424).
Bar: trajectory points and world-frame images within 2e-6 m (same fp32 operations); joint targets within 2e-5 rad (acosf / asinf / atan2f
of the device library against libm); legs not flagged as swinging are left untouched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_swing_targets_parity(gpu_ctx, pkg, oracle):
    W = pkg.workload
    n = 777
    cfg = W.estimator_cfg("a1"); geom, ho = cfg[:3], cfg[7:19]
    x = W.make_swing_batch(n, seed=8)
    S = pkg.to_soa
    sentinel = np.float32(-777.0)
    d_in = gpu_ctx.alloc((58, n)).upload(S(x))
    d_cmd = gpu_ctx.alloc((67, n)).upload(np.full((67, n), sentinel, np.float32))
    d_tgt = gpu_ctx.alloc((12, n)).upload(np.full((12, n), sentinel, np.float32))
    d_q = gpu_ctx.alloc((24, n)).upload(np.full((24, n), sentinel, np.float32))
    gpu_ctx.swing_targets_batch(n, cfg, d_in, d_cmd, d_tgt, d_q)
    gpu_ctx.sync()
    cmd = d_cmd.download().T; tgt = d_tgt.download().T; qd = d_q.download().T
    assert np.all(cmd[:, :15] == sentinel) and np.all(cmd[:, 51:] == sentinel)          # only the foot-task rows are the kernel's
    for i in range(n):
        o = oracle.swing_targets(geom, ho, x[i], np.full(72, sentinel, np.float32))
        g = np.concatenate([cmd[i, 15:51], tgt[i], qd[i]])
        assert np.array_equal(g == sentinel, o == sentinel), i                            # same legs written
        m = o != sentinel
        assert np.abs(g[:48][m[:48]] - o[:48][m[:48]]).max(initial=0) <= 2e-6, i
        # (an unreachable foothold gives NaN angles -> replaced by the current ones, and NaN joint velocities, as in the reference)
        assert np.allclose(g[48:][m[48:]], o[48:][m[48:]], rtol=0, atol=2e-5, equal_nan=True), (i, g[48:60], o[48:60])
    for v in (d_in, d_cmd, d_tgt, d_q):
        v.free()


def test_swing_velocity_mode_parity(gpu_ctx, pkg, oracle):
    """qrgpu_swing_velocity_batch (VELOCITY_LOCOMOTION case, qr_swing_leg_controller.cpp:285-309, 408-424) against the oracle: targets and
    trajectory points within 2e-6 m, joint targets within 2e-5 rad, legs that are not flagged untouched."""
    W = pkg.workload
    n = 700
    cfg = W.estimator_cfg("a1"); geom, ho = cfg[:3], cfg[7:19]
    vd = W.swing_velocity_cfg("a1")
    x = W.make_swing_velocity_batch(n, seed=9)
    sentinel = np.float32(-777.0)
    d_in = gpu_ctx.alloc((53, n)).upload(pkg.to_soa(x))
    d_out = gpu_ctx.alloc((48, n)).upload(np.full((48, n), sentinel, np.float32))
    gpu_ctx.swing_velocity_batch(n, cfg, vd, d_in, d_out)
    gpu_ctx.sync()
    g = d_out.download().T
    written = 0
    for i in range(n):
        o = oracle.swing_velocity(geom, ho, vd, x[i], np.full(48, sentinel, np.float32))
        assert np.array_equal(g[i] == sentinel, o == sentinel), i
        m = o != sentinel
        written += int(m.sum())
        assert np.abs(g[i, :24][m[:24]] - o[:24][m[:24]]).max(initial=0) <= 2e-6, (i, np.abs(g[i, :24] - o[:24]).max())
        assert np.allclose(g[i, 24:][m[24:]], o[24:][m[24:]], rtol=0, atol=2e-5, equal_nan=True), (i, g[i, 24:36], o[24:36])
    assert written > 20 * n
    d_in.free(); d_out.free()
