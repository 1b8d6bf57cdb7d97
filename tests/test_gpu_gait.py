"""GPU parity of the open-loop gait generator kernel (qrgpu_gait_update_batch) against the oracle over a tick sequence.
Reference: qrOpenLoopGaitGenerator::Update / Schedule (qr_openloop_gait_generator.cpp:126-249).  Bar: bit-exact (plain float arithmetic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gait_sequence_bit_exact(gpu_ctx, pkg, oracle):
    W = pkg.workload
    n, ticks, dt = 500, 900, 0.002
    cfg = W.gait_cfg(wait_time=0.06)
    contacts = W.make_gait_contacts(n, ticks, cfg, seed=4)
    t = (np.arange(ticks) * dt).astype(np.float32)
    d_state = gpu_ctx.alloc((52, n)).upload(np.full((52, n), np.nan, np.float32))      # reset must not depend on what was there
    d_c = gpu_ctx.alloc((4, n)); d_out = gpu_ctx.alloc((24, n))
    d_fe = gpu_ctx.alloc((64, n)).upload(np.full((64, n), -7.0, np.float32))
    snaps = {}
    for k in range(ticks):
        d_c.upload(pkg.to_soa(contacts[k]))
        gpu_ctx.gait_update_batch(n, cfg, float(t[k]), d_c, d_state, d_out, d_fe, reset=(k == 0))
        if k % 50 == 49 or k < 3:
            snaps[k] = d_out.download().T.copy()
    fe = d_fe.download().T
    for r in range(0, n, 5):
        o = oracle.gait_run(cfg, t, contacts[:, r])
        for k, g in snaps.items():
            assert np.array_equal(g[r], o[k]), (r, k, g[r], o[k])
        # rows 42-61 of the front-end input after the last tick
        assert np.array_equal(fe[r, 42:46], o[-1, 0:4]) and np.array_equal(fe[r, 50:54], o[-1, 4:8])
        assert np.array_equal(fe[r, 54:58], o[-1, 8:12]) and np.array_equal(fe[r, 58:62], o[-1, 12:16]) and np.all(fe[r, 46:50] == cfg[4])
        assert np.all(fe[r, :42] == -7.0) and np.all(fe[r, 62:] == -7.0)
    # the batch exercised the hold and the early contact
    allout = np.stack([oracle.gait_run(cfg, t, contacts[:, r]) for r in range(0, n, 5)])
    assert (allout[:, :, 12:16] == 2).any()
    for v in (d_state, d_c, d_out, d_fe):
        v.free()
