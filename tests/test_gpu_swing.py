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
