"""GPU parity of the swing-leg selection + foothold heuristic (qrgpu_footholds_batch) against the oracle, and the chain
gait generator -> footholds -> swing targets on the device.
Reference: qrRaibertSwingLegController::Update (qr_swing_leg_controller.cpp:211-236), qrFootholdPlanner::ComputeHeuristicFootHold
(qr_foothold_planner.cpp:110-239).  Bar: same fp32 operations with contraction off; only sinf/cosf of the roll angle may differ
from libm by an ulp -> 1e-6 m."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_footholds_parity(gpu_ctx, pkg, oracle):
    W = pkg.workload
    n = 1500
    S = pkg.to_soa
    for robot in ("a1", "lite3"):
        desc = W.foothold_cfg(robot)
        x = W.make_foothold_batch(n, robot, seed=21)
        sentinel = np.float32(-777.0)
        d_in = gpu_ctx.alloc((46, n)).upload(S(x))
        d_sw = gpu_ctx.alloc((58, n)).upload(np.full((58, n), sentinel, np.float32))
        gpu_ctx.footholds_batch(n, desc, d_in, d_sw)
        gpu_ctx.sync()
        g = d_sw.download().T
        worst = 0.0
        for i in range(n):
            o = oracle.footholds(desc, x[i], np.full(58, sentinel, np.float32))
            assert np.array_equal(g[i] == sentinel, o == sentinel), i                       # same rows written
            assert np.array_equal(g[i, :8], o[:8]), i                                       # flags and phases exactly
            worst = max(worst, float(np.abs(g[i, 24:36] - o[24:36]).max()))
        assert worst <= 1e-6, worst
        d_in.free(); d_sw.free()


def test_gait_to_swing_targets_chain_on_device(gpu_ctx, pkg, oracle):
    """gait kernel -> foothold kernel (reading the gait kernel's arrays) -> swing-target kernel, against the same chain of oracles."""
    W = pkg.workload
    n, ticks, dt = 64, 130, 0.002
    S = pkg.to_soa
    gcfg, fcfg, ecfg = W.gait_cfg(), W.foothold_cfg("a1"), W.estimator_cfg("a1")
    contacts = W.make_gait_contacts(n, ticks, gcfg, seed=5)
    fh = W.make_foothold_batch(n, "a1", seed=6)
    sw = W.make_swing_batch(n, seed=7)
    d_st = gpu_ctx.alloc((52, n)).upload(np.zeros((52, n), np.float32))
    d_go = gpu_ctx.alloc((24, n)); d_ct = gpu_ctx.alloc((4, n))
    d_fh = gpu_ctx.alloc((46, n)).upload(S(fh)); d_sw = gpu_ctx.alloc((58, n)).upload(S(sw))
    d_cmd = gpu_ctx.alloc((67, n)).upload(np.zeros((67, n), np.float32))
    sw_o = sw.copy()
    for k in range(ticks):
        d_ct.upload(S(contacts[k]))
        gpu_ctx.gait_update_batch(n, gcfg, k * dt, d_ct, d_st, d_go, reset=(k == 0))
        gpu_ctx.footholds_batch(n, fcfg, d_fh, d_sw, gait_state=d_st, gait_out=d_go)
    gpu_ctx.swing_targets_batch(n, ecfg, d_sw, d_cmd)
    gpu_ctx.sync()
    g_sw = d_sw.download().T; g_cmd = d_cmd.download().T
    # the same chain with the oracle's foothold step, fed tick by tick with the gait kernel's arrays (the gait kernel itself is pinned by
    # test_gpu_gait.py): rows a leg keeps while it is not swinging come from earlier ticks, so all ticks are replayed
    d_st.upload(np.zeros((52, n), np.float32)); d_sw.upload(S(sw))
    for k in range(ticks):
        d_ct.upload(S(contacts[k]))
        gpu_ctx.gait_update_batch(n, gcfg, k * dt, d_ct, d_st, d_go, reset=(k == 0))
        gpu_ctx.sync()
        st_k, go_k = d_st.download().T, d_go.download().T
        for i in range(n):
            x = fh[i].copy()
            x[0:4] = st_k[i, 16:20]; x[4:8] = st_k[i, 20:24]; x[8:12] = go_k[i, 20:24]; x[12:16] = go_k[i, 4:8]
            sw_o[i] = oracle.footholds(fcfg, x, sw_o[i])
    assert np.array_equal(g_sw[:, :8], sw_o[:, :8])
    assert np.abs(g_sw[:, 24:36] - sw_o[:, 24:36]).max() <= 1e-6
    swung = int(g_sw[:, :4].sum())
    assert 0 < swung < 4 * n
    for i in range(n):
        o = oracle.swing_targets(ecfg[:3], ecfg[7:19], sw_o[i], np.zeros(72, np.float32))
        assert np.abs(g_cmd[i, 15:51] - o[:36]).max() <= 5e-6, i
    for v in (d_st, d_go, d_ct, d_fh, d_sw, d_cmd):
        v.free()
