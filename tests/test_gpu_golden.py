"""-m gpu: the HIP path against the committed golden fixtures (reference solver outputs + oracle outputs)."""
import numpy as np
import pytest

import golden_io
import gpu_helpers as G

pytestmark = pytest.mark.gpu


def test_mpc_golden(gpu_ctx, pkg):
    rows = golden_io.load("mpc_golden.npz")
    groups = {}
    for r in rows:
        groups.setdefault((str(r["robot"]), int(r["h"])), []).append(r)
    for (robot, h), rs in groups.items():
        gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), h)
        gpu_ctx.wbc_setup_packed(0, pkg.model_desc(robot))
        n = len(rs)
        fb = np.zeros((n, 37), np.float32)
        fb[:, 0:4] = [r["quat"] for r in rs]; fb[:, 13:25] = [r["q"] for r in rs]
        b = dict(n=n, horizon=h, mpc_state=np.stack([r["mpc_state"] for r in rs]), traj=np.stack([r["traj"] for r in rs]),
                 gait=np.stack([r["gait"] for r in rs]), fb_state=fb)
        out = G.run_mpc(gpu_ctx, pkg, b)
        assert np.all(G.flags(out["status"]) == 0), (robot, h, G.flags(out["status"]))
        for i, r in enumerate(rs):
            scale = max(1.0, np.abs(r["f_qpoases_sym"]).max())
            # reference qpOASES on the symmetric data, converged: same unique optimum
            assert np.abs(out["force"][i] - r["f_qpoases_sym"]).max() <= 2e-6 * scale, (robot, h, i)
            assert np.abs(out["force"][i] - r["f_oracle"]).max() <= 2e-6 * scale
            assert np.all(np.abs(out["tau"][i] - r["tau_oracle"]) <= G.tau_tol(r["tau_oracle"], 1e-5))


def test_mpc_vs_reference_solver_as_called(gpu_ctx, pkg):
    """The HIP path against the reference's own solver call: qpOASES 3.2.0 compiled from the reference tree, fed the asymmetric fp32 H with
    nWSR = 100 exactly as qr_mpc_interface.cpp:418-438 does (tests/golden/make_golden.py; rows on which that call runs into its nWSR cap
    or fails are recorded and skipped here).  The reference's answer moves when it is handed H^T instead of the (nominally symmetric) H;
    the stated QP depends on H only through (H + H^T)/2, and the kernel's answer must be
      * within that ambiguity of the as-called answer:  |f_gpu - f(H)| <= 0.55 |f(H^T) - f(H)| (+ 1e-5 of the force scale), same in torque;
      * the MIDPOINT of the reference's two answers to 1e-5 relative force / 1.5e-4 relative torque;
      * per horizon no further from the as-called answer than the committed table tests/golden/parity_as_called.json:
        2.6e-3 (h = 10, the worst rows being those of bench.py's own batch) and 1.2e-3 (h = 16) relative force."""
    import json, os
    rows = golden_io.load("mpc_golden.npz")
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_as_called.json")))
    groups = {}
    for r in rows:
        groups.setdefault((str(r["robot"]), int(r["h"])), []).append(r)
    worst = {}
    n_checked = n_cap = 0
    for (robot, h), rs in groups.items():
        gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), h)
        gpu_ctx.wbc_setup_packed(0, pkg.model_desc(robot))
        n = len(rs)
        fb = np.zeros((n, 37), np.float32)
        fb[:, 0:4] = [r["quat"] for r in rs]; fb[:, 13:25] = [r["q"] for r in rs]
        b = dict(n=n, horizon=h, mpc_state=np.stack([r["mpc_state"] for r in rs]), traj=np.stack([r["traj"] for r in rs]),
                 gait=np.stack([r["gait"] for r in rs]), fb_state=fb)
        out = G.run_mpc(gpu_ctx, pkg, b)
        assert np.all(G.flags(out["status"]) == 0), (robot, h)
        for i, r in enumerate(rs):
            nwsr, rc = int(r["qpoases_as_called_nwsr"][0]), int(r["qpoases_as_called_nwsr"][1])
            if rc != 0 or nwsr >= 100:
                n_cap += 1                  # the reference returned a non-optimal point: nothing to be close to
                continue
            n_checked += 1
            f, fc, ft = out["force"][i].astype(np.float64), r["f_qpoases_as_called"], r["f_qpoases_transposed"]
            tau, tc, tt = out["tau"][i].astype(np.float64), r["tau_qpoases_as_called"].astype(np.float64), r["tau_qpoases_transposed"].astype(np.float64)
            fs = max(1.0, np.abs(fc).max())
            ts = np.maximum(1.0, np.abs(tc))
            ef, af = np.abs(f - fc).max(), np.abs(ft - fc).max()
            assert ef <= 0.55 * af + 1e-5 * fs, (robot, h, i, ef, af)
            assert np.all(np.abs(tau - tc) <= 0.55 * np.abs(tt - tc).max() + 1.5e-4 * ts), (robot, h, i)
            assert np.abs(f - 0.5 * (fc + ft)).max() <= 1e-5 * fs, (robot, h, i, np.abs(f - 0.5 * (fc + ft)).max() / fs)
            assert np.all(np.abs(tau - 0.5 * (tc + tt)) <= 1.5e-4 * ts), (robot, h, i)
            w = worst.setdefault(h, [0.0, 0.0])
            w[0] = max(w[0], ef / fs); w[1] = max(w[1], (np.abs(tau - tc) / ts).max())
    assert n_checked >= 50 and n_cap >= 1
    for h, (wf, wt) in worst.items():
        t = table["h%d" % h]
        assert wf <= 1.02 * t["max_rel_force"] + 2e-6 and wt <= 1.02 * t["max_rel_torque"] + 2e-5, (h, wf, wt, t)
    G.setup_a1(gpu_ctx, pkg, 10)


def test_wbc_golden(gpu_ctx, pkg):
    rows = golden_io.load("wbc_golden.npz")
    for robot in ("a1", "lite3"):
        rs = [r for r in rows if str(r["robot"]) == robot]
        gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), 10)
        gpu_ctx.wbc_setup_packed(0, pkg.model_desc(robot))
        n = len(rs)
        b = dict(n=n, horizon=10, fb_state=np.stack([r["fb_state"] for r in rs]), wbc_cmd=np.stack([r["wbc_cmd"] for r in rs]),
                 prev_ori_vel=np.stack([r["prev"] for r in rs]))
        out = G.run_wbc(gpu_ctx, pkg, b)
        assert np.all(out["status"] == 0)
        dbg = G.run_fb_debug(gpu_ctx, pkg, b)
        for i, r in enumerate(rs):
            assert np.all(np.abs(out["tau"][i] - r["tau64"]) <= G.tau_tol(r["tau64"], 1e-6)), (robot, i, np.abs(out["tau"][i] - r["tau64"]).max())
            assert np.all(np.abs(out["tau"][i] - r["tau32"]) <= G.tau_tol(r["tau32"], 1e-4))
            assert np.abs(out["qdes"][i] - r["qdes64"]).max() <= 1e-5
            assert np.abs(dbg["H"][i] - r["H"]).max() <= 2e-6 and np.abs(dbg["Jc"][i] - r["Jc"]).max() <= 1e-6
    G.setup_a1(gpu_ctx, pkg, 10)


def test_wbc_relaxation_qp_vs_quadprogpp_as_called(gpu_ctx, pkg):
    """the kernel's torque and optimalFr against the reference's QuadProg++ (compiled from the reference tree) called as
    qr_wholebody_impulse_ctrl.cpp:113 calls it on the QP the tick assembles (:129-167, 232-247): 48 cases, 32 of them fed the MPC's own
    forces as the tick does, up to 7 inequality rows active.  G is diagonal: one optimum, nothing to choose between.
      * against the QP assembled in double (the kernel's arithmetic): 1e-6 * max(1, |.|) on tau, optimalFr and z -- only fp32 output rounding left;
      * against the QP assembled in float as the reference computes it: north_star's 1e-4 * max(1, |tau|)."""
    rows = golden_io.load("wbc_golden.npz")
    n_active_rows = 0
    for robot in ("a1", "lite3"):
        rs = [r for r in rows if str(r["robot"]) == robot]
        gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), 10)
        gpu_ctx.wbc_setup_packed(0, pkg.model_desc(robot))
        n = len(rs)
        b = dict(n=n, horizon=10, fb_state=np.stack([r["fb_state"] for r in rs]), wbc_cmd=np.stack([r["wbc_cmd"] for r in rs]),
                 prev_ori_vel=np.stack([r["prev"] for r in rs]))
        out = G.run_wbc_inspect(gpu_ctx, pkg, b)
        plain = G.run_wbc(gpu_ctx, pkg, b)
        assert np.all(out["status"] == 0)
        assert np.array_equal(out["tau"], plain["tau"])              # the instrumented kernel is the timed kernel
        for i, r in enumerate(rs):
            tol = lambda ref, rel: rel * np.maximum(1.0, np.abs(ref))
            nz = r["z_quadprogpp64"].size
            assert np.all(np.abs(out["tau"][i] - r["tau_quadprogpp64"]) <= tol(r["tau_quadprogpp64"], 1e-6)), (robot, i)
            assert np.all(np.abs(out["fr"][i] - r["fr_quadprogpp64"]) <= tol(r["fr_quadprogpp64"], 1e-6)), (robot, i)
            assert np.all(np.abs(out["z"][i, :nz] - r["z_quadprogpp64"]) <= tol(r["z_quadprogpp64"], 1e-6)) and np.all(out["z"][i, nz:] == 0)
            assert np.all(np.abs(out["tau"][i] - r["tau_quadprogpp32"]) <= tol(r["tau_quadprogpp32"], 1e-4)), (robot, i)
            assert np.all(np.abs(out["fr"][i] - r["fr_quadprogpp32"]) <= tol(r["fr_quadprogpp32"], 1e-4)), (robot, i)
            n_active_rows += int(r["n_active"][0])
    assert n_active_rows >= 60
    G.setup_a1(gpu_ctx, pkg, 10)


def _tick_rows(gpu_ctx, pkg):
    """The full tick (K12 on, K14 tail on) of every row of mpc_golden.npz on the GPU, grouped by (robot, horizon).  -> [(row, tau, force)]"""
    rows = golden_io.load("mpc_golden.npz")
    groups = {}
    for r in rows:
        groups.setdefault((str(r["robot"]), int(r["h"])), []).append(r)
    res = []
    gpu_ctx.set_torque_epilogue(hip_comp=True, clip=True)
    try:
        for (robot, h), rs in groups.items():
            gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), h)
            gpu_ctx.wbc_setup_packed(0, pkg.model_desc(robot))
            b = dict(n=len(rs), horizon=h, mpc_state=np.stack([r["mpc_state"] for r in rs]), traj=np.stack([r["traj"] for r in rs]),
                     gait=np.stack([r["gait"] for r in rs]), fb_state=np.stack([r["fb_state"] for r in rs]),
                     wbc_cmd=np.stack([r["wbc_cmd"] for r in rs]), prev_ori_vel=np.stack([r["prev"] for r in rs]))
            out = G.run_tick(gpu_ctx, pkg, b, want_qdes=True)
            assert np.all(G.flags(out["status"]) == 0), (robot, h)
            res += [(r, out["tau"][i].astype(np.float64), out["force"][i].astype(np.float64)) for i, r in enumerate(rs)]
    finally:
        gpu_ctx.set_torque_epilogue(False, False)
        G.setup_a1(gpu_ctx, pkg, 10)
    return res


def test_full_tick_torque_vs_reference_as_called(gpu_ctx, pkg):
    """VERDICT r2 item 1(a), the metric's own quantity: the K14 torque of the FULL tick (MPC -> WBC -> merge, abad compensation, clip) on the
    GPU against the reference's forces -- qpOASES called as qr_mpc_interface.cpp:428-438 calls it, asymmetric fp32 H, nWSR = 100 -- carried
    through the oracle's WBC and K14 tail (tests/golden/make_golden.py: tau_tick_as_called; _transposed: the same call handed H^T; _literal:
    handed the QP assembled by the reference's other fp32 route).  86 rows, 24 of them robots 0..23 of bench.py's own batch.  Asserted:
      * against the oracle's tick (the stated QP, fp32 WBC as the reference computes): 1e-4 * max(1, |tau|) -- north_star's tolerance;
      * against the as-called answer: within 0.55 of the reference's own H <-> H^T spread (+1.5e-4), and at its midpoint to 3e-4
        (exactly so in the forces; the WBC between forces and torques is only piecewise linear);
      * the RAW figures per horizon are those of the committed table (parity_as_called.json: 3.6e-2 at h = 10, 2.5e-2 at h = 16 -- beyond
        1e-4; test_north_star_torque_bar_vs_reference_as_called below states that as a test of its own);
      * wherever the reference's answer is itself determined to 1e-4 (its H <-> H^T spread below 1e-4) the GPU torque is within 1e-4 of it."""
    import json, os
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_as_called.json")))
    worst, above, determined = {}, {}, 0
    n_checked = 0
    for r, tau, f in _tick_rows(gpu_ctx, pkg):
        h = int(r["h"])
        to = r["tau_tick_oracle"].astype(np.float64)
        assert np.all(np.abs(tau - to) <= G.tau_tol(to, 1e-4)), (h, np.abs(tau - to).max())
        assert np.all(np.abs(tau - r["tau_tick_oracle64"]) <= G.tau_tol(to, 2e-5))
        if int(r["qpoases_as_called_nwsr"][1]) != 0 or int(r["qpoases_as_called_nwsr"][0]) >= 100:
            continue
        n_checked += 1
        tc, tt = r["tau_tick_as_called"].astype(np.float64), r["tau_tick_transposed"].astype(np.float64)
        ts = np.maximum(1.0, np.abs(tc))
        spread = np.abs(tt - tc).max()
        assert np.all(np.abs(tau - tc) <= 0.55 * spread + 1.5e-4 * ts), (h, np.abs(tau - tc).max(), spread)
        assert np.all(np.abs(tau - 0.5 * (tc + tt)) <= 3e-4 * ts), h      # (the WBC's relaxation QP is only piecewise linear in Fr_des)
        raw = (np.abs(tau - tc) / ts).max()
        for key in ["h%d" % h] + (["h%d_bench_batch" % h] if int(r["bench_batch"][0]) else []):
            worst[key] = max(worst.get(key, 0.0), raw)
            above[key] = above.get(key, 0) + int(raw > 1e-4)
        if (np.abs(tt - tc) / ts).max() <= 1e-4:
            determined += 1
            assert raw <= 1e-4, (h, raw)
    assert n_checked >= 75 and determined >= 20
    for key, w in worst.items():
        t = table[key]
        assert w <= 1.02 * t["max_rel_tick_torque"] + 2e-5, (key, w, t["max_rel_tick_torque"])
        assert abs(above[key] - t["rows_above_1e_4_tick_torque"]) <= 2, (key, above[key], t["rows_above_1e_4_tick_torque"])
    assert worst["h10_bench_batch"] > 1e-2           # the figure bench.py's line reports on its own batch is covered by the committed table


@pytest.mark.xfail(strict=True, reason="north_star's 1e-4 relative torque is NOT met against the reference's solver as called (DESIGN.md 2): "
                                       "qpOASES fed the asymmetric fp32 H answers 3.6e-2 away at h = 10 -- and 4e-2 away from ITSELF when the same "
                                       "QP is assembled by its other fp32 route.  Strict: if this ever passes, the statement must be rewritten.")
def test_north_star_torque_bar_vs_reference_as_called(gpu_ctx, pkg):
    for r, tau, f in _tick_rows(gpu_ctx, pkg):
        if int(r["qpoases_as_called_nwsr"][1]) != 0 or int(r["qpoases_as_called_nwsr"][0]) >= 100:
            continue
        tc = r["tau_tick_as_called"].astype(np.float64)
        assert np.all(np.abs(tau - tc) <= 1e-4 * np.maximum(1.0, np.abs(tc)))
