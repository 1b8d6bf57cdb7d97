"""GPU parity of the batched MPC front-end (qrgpu_mpc_frontend_batch) against the oracle restatement of
MPCStanceLegController::SetupCommand / Run / UpdateMPC (qr_mpc_stance_leg_controller.cpp:158-382).

Bar: bit-exact float32, except the three outputs that carry std::sin (height / pitch compensation: bodyHeight, rpyComp[1]),
where libm and the device library may differ in the last bit of a double sine -> at most 1 float ulp."""
import numpy as np

import gpu_helpers as G
import pytest

from gpu_helpers import setup_a1, tau_tol

pytestmark = pytest.mark.gpu
SIN_CMD_ROWS = [2, 10]            # pBody_des z (bodyHeight), pBody_RPY_des pitch
SIN_TRAJ_COLS = [1, 5]


def _run_gpu(ctx, pkg, fe, st, h, L=2, prev_traj=None):
    n = fe.shape[0]
    S = pkg.to_soa
    d_in = ctx.alloc((64, n)).upload(S(fe)); d_st = ctx.alloc((8, n)).upload(S(st))
    t0 = np.full((n, 12 * h), np.nan, np.float32) if prev_traj is None else prev_traj
    d_traj = ctx.alloc((12 * h, n)).upload(S(t0)); d_gait = ctx.alloc((4 * h, n))
    d_cmd = ctx.alloc((67, n)).upload(np.full((67, n), np.nan, np.float32)); d_upd = ctx.alloc((n,), np.int32)
    ctx.mpc_frontend_batch(n, d_in, d_st, d_traj, d_gait, d_cmd, d_upd, num_horizon_l=L)
    ctx.sync()
    out = dict(traj=d_traj.download().T.copy(), gait=d_gait.download().T.copy(), cmd=d_cmd.download().T.copy(),
               state=d_st.download().T.copy(), updated=d_upd.download())
    for v in (d_in, d_st, d_traj, d_gait, d_cmd, d_upd):
        v.free()
    return out


def _ulp_close(a, b, ulps=1):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


@pytest.mark.parametrize("h,L", [(10, 2), (5, 2), (16, 3)])
def test_frontend_parity(gpu_ctx, pkg, oracle, h, L):
    setup_a1(gpu_ctx, pkg, h)
    n = 1000                                   # not a multiple of the block size: ragged tail
    fe, st = pkg.workload.make_frontend_batch(n, seed=h)
    g = _run_gpu(gpu_ctx, pkg, fe, st, h, L)
    exact_cmd = [r for r in list(range(15)) + [63, 64, 65, 66] if r not in SIN_CMD_ROWS]
    exact_traj = [c for c in range(12) if c not in SIN_TRAJ_COLS]
    n_upd = 0
    for i in range(n):
        o = oracle.mpc_frontend(h, L, fe[i], st[i])
        assert g["updated"][i] == o["updated"]
        assert np.array_equal(g["state"][i], o["state"]), i
        assert np.array_equal(g["gait"][i], o["gait"]), i
        assert np.array_equal(g["cmd"][i, exact_cmd[:13]], o["wbc15"][exact_cmd[:13]]), i
        assert np.array_equal(g["cmd"][i, 63:67], o["contact"])
        assert _ulp_close(g["cmd"][i, SIN_CMD_ROWS], o["wbc15"][SIN_CMD_ROWS])
        assert np.isnan(g["cmd"][i, 15:63]).all()                       # foot tasks / Fr_des rows are not the front-end's
        tg = g["traj"][i].reshape(h, 12)
        if o["updated"]:
            n_upd += 1
            to = o["traj"].reshape(h, 12)
            assert np.array_equal(tg[:, exact_traj], to[:, exact_traj]), i
            assert _ulp_close(tg[:, SIN_TRAJ_COLS], to[:, SIN_TRAJ_COLS])
        else:
            assert np.isnan(tg).all()                                   # no re-plan: trajectory left alone
    assert 0 < n_upd < n


def test_frontend_sequence_feeds_tick(gpu_ctx, pkg, oracle):
    """Thirty consecutive ticks with persistent fe_state, then the trajectory / table / wbc rows the front-end wrote drive one
    full MPC+WBC tick whose torques match the oracle fed with the oracle front-end's outputs."""
    h, n = 10, 64
    setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(n, horizon=h, robot="a1", seed=77, excite=0.3)
    fe, st = pkg.workload.make_frontend_batch(n, seed=21)
    st[:, 7] = 40                                                       # crosses the 50-tick boundary of the cadence
    # tie the front-end's robot state to the MPC batch (position, attitude, contacts of step 0)
    shift = b["mpc_state"][:, 0:2] - fe[:, 6:8]
    for l in range(4):
        fe[:, 14 + 3 * l:16 + 3 * l] += shift; fe[:, 26 + 3 * l:28 + 3 * l] += shift
    fe[:, 62:64] += shift
    fe[:, 6:9] = b["mpc_state"][:, 0:3]; fe[:, 10:14] = b["mpc_state"][:, 6:10]; fe[:, 9] = b["mpc_state"][:, 27]
    fe[:, 38:42] = b["gait"][:, 0:4]
    fe[:, 0] = b["traj"][:, 5]; fe[:, 2] = 0
    st[:, 3] = b["mpc_state"][:, 27]; st[:, 4:6] = b["mpc_state"][:, 0:2]
    fe[:, 3:6] *= 0.3
    S = pkg.to_soa
    d_in = gpu_ctx.alloc((64, n)).upload(S(fe)); d_st = gpu_ctx.alloc((8, n)).upload(S(st))
    d_traj = gpu_ctx.alloc((12 * h, n)).upload(S(b["traj"])); d_gait = gpu_ctx.alloc((4 * h, n))
    d_cmd = gpu_ctx.alloc((67, n)).upload(S(b["wbc_cmd"])); d_upd = gpu_ctx.alloc((n,), np.int32)
    o_st = st.copy(); o_traj = b["traj"].copy(); o_gait = np.zeros((n, 4 * h), np.float32); o_cmd = b["wbc_cmd"].copy()
    upd_hist = []
    for k in range(30):
        gpu_ctx.mpc_frontend_batch(n, d_in, d_st, d_traj, d_gait, d_cmd, d_upd)
        for i in range(n):
            o = oracle.mpc_frontend(h, 2, fe[i], o_st[i])
            o_st[i] = o["state"]; o_gait[i] = o["gait"]; o_cmd[i, :15] = o["wbc15"]; o_cmd[i, 63:67] = o["contact"]
            if o["updated"]:
                o_traj[i] = o["traj"]
        upd_hist.append(int(d_upd.download().sum()))
    gpu_ctx.sync()
    # counters 40..49 re-plan every tick (< 50); of 50..69 only 60 does (% 15)
    assert upd_hist[:10] == [n] * 10 and upd_hist[20] == n and sum(upd_hist[10:]) == n
    assert np.array_equal(d_st.download().T, o_st)
    assert np.array_equal(d_gait.download().T, o_gait)
    g_traj = d_traj.download().T.copy(); g_gait = d_gait.download().T.copy(); g_cmd = d_cmd.download().T.copy()
    assert np.allclose(g_traj, o_traj, rtol=0, atol=1e-6)
    assert np.allclose(g_cmd, o_cmd, rtol=0, atol=1e-6)
    # one full tick on what the front-end produced (device buffers handed straight to qrgpu_tick_batch)
    d = dict(state=gpu_ctx.alloc((28, n)).upload(S(b["mpc_state"])), fb=gpu_ctx.alloc((37, n)).upload(S(b["fb_state"])),
             prev=gpu_ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])), force=gpu_ctx.alloc((12, n)), tau=gpu_ctx.alloc((12, n)),
             status=gpu_ctx.alloc((n,), np.int32))
    gpu_ctx.tick_batch(n, d["state"], d_traj, d_gait, d["fb"], d_cmd, d["prev"], d["force"], d["tau"], d["status"])
    gpu_ctx.sync()
    tau = d["tau"].download().T; status = d["status"].download()
    assert np.all(G.flags(status) == 0), np.unique(G.flags(status))
    f_o, tau_o, st_o, _, _ = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"],
                                               g_traj, g_gait, b["fb_state"], g_cmd, b["prev_ori_vel"].copy(), nthreads=4)
    assert np.all(st_o == 0)
    assert np.all(np.abs(tau - tau_o) <= tau_tol(tau_o, 1e-4)), np.abs(tau - tau_o).max()
    for v in list(d.values()) + [d_in, d_st, d_traj, d_gait, d_cmd, d_upd]:
        v.free()
