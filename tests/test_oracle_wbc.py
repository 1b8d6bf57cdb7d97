"""CPU: the oracle's rigid-body model and WBC tick (K8-K14): analytic invariants, float64 cross-checks,
golden fixtures (tests/golden/wbc_golden.npz)."""
import numpy as np
import pytest

import golden_io


def test_pinv_and_lu_match_numpy(oracle):
    rng = np.random.default_rng(0)
    for r, c in ((3, 18), (6, 18), (12, 18), (12, 12), (3, 3), (18, 6)):
        A = rng.normal(size=(r, c)).astype(np.float32)
        P = oracle.pinv(A, 1e-3)
        assert np.abs(P - np.linalg.pinv(A.astype(np.float64))).max() < 2e-5
    # singular values at/below the threshold are dropped (strict '>'), 1x1 compares the entry itself
    U, _ = np.linalg.qr(rng.normal(size=(6, 6))); V, _ = np.linalg.qr(rng.normal(size=(18, 18)))
    s = np.array([2.0, 1.0, 0.5, 0.1, 5e-4, 1e-5])
    A = (U * s) @ V[:6]
    P = oracle.pinv(A.astype(np.float32), 1e-3)
    ref = (V[:6].T * np.where(s > 1e-3, 1 / s, 0.0)) @ U.T
    assert np.abs(P - ref).max() < 1e-3
    assert oracle.pinv(np.array([[-5.0]], np.float32), 1e-4)[0, 0] == 0.0          # quirk 7
    assert abs(oracle.pinv(np.array([[4.0]], np.float32), 1e-4)[0, 0] - 0.25) < 1e-7
    M = rng.normal(size=(18, 18)); M = (M @ M.T + 18 * np.eye(18)).astype(np.float32)
    assert np.abs(oracle.lu_inverse(M) - np.linalg.inv(M.astype(np.float64))).max() < 1e-6


def test_rigid_body_invariants(oracle, pkg):
    md = pkg.model_desc("a1")
    b = pkg.make_batch(6, 10, "a1", seed=3)
    for i in range(6):
        s = b["fb_state"][i].astype(np.float64)
        s[:4] /= np.linalg.norm(s[:4])
        r = oracle.fb_compute(md, s, np.float64)
        H = r["H"]
        assert np.abs(H - H.T).max() < 1e-12 and np.linalg.eigvalsh(H).min() > 0
        m_tot = 6 + 4 * (np.float32(0.696) + np.float32(1.013) + np.float32(0.166)) + 12 * np.float32(1e-8)
        assert np.allclose(np.diag(H)[3:6], m_tot, rtol=1e-7)              # total mass incl. rotors, body frame
        assert abs(np.linalg.norm(r["G"][3:6]) - m_tot * 9.81) < 1e-4
        qd = np.r_[s[7:13], s[25:37]]
        for leg in range(4):                                               # v_foot = Jc * qdot
            assert np.abs(r["Jc"][leg] @ qd - r["vGC"][leg]).max() < 1e-9
            for j in range(12):                                            # Jc = d p_foot / d q
                s2 = s.copy(); s2[13 + j] += 1e-6
                num = (oracle.fb_compute(md, s2, np.float64)["pGC"][leg] - r["pGC"][leg]) / 1e-6
                assert np.abs(num - r["Jc"][leg][:, 6 + j]).max() < 1e-5
        # zero velocity => no Coriolis
        s0 = s.copy(); s0[7:13] = 0; s0[25:37] = 0
        assert np.abs(oracle.fb_compute(md, s0, np.float64)["C"]).max() < 1e-14
        # kinetic energy from H equals the sum over the analytic leg/foot picture only loosely; check positivity
        assert qd @ H @ qd > 0


def test_wbc_satisfies_floating_base_dynamics(oracle, pkg):
    """The relaxation QP's equality rows: (A qdd + C + G - Jc' f)[0:6] = 0 for the returned qdd, f."""
    md = pkg.model_desc("a1")
    b = pkg.make_batch(12, 10, "a1", seed=8)
    for i in range(12):
        s = b["fb_state"][i].astype(np.float64); c = b["wbc_cmd"][i].astype(np.float64)
        w = oracle.wbc_run(md, s, c, dtype=np.float64)
        assert w["rc"] == 0
        fb = oracle.fb_compute(md, s, np.float64)
        stance = [l for l in range(4) if c[63 + l] != 0]
        gen = fb["H"] @ w["qddot"] + fb["C"] + fb["G"]
        for k, leg in enumerate(stance):
            gen -= fb["Jc"][leg].T @ w["fr"][3 * k:3 * k + 3]
        assert np.abs(gen[:6]).max() < 1e-8
        assert np.abs(gen[6:] - w["tau"]).max() < 1e-9
        for k, leg in enumerate(stance):                                   # friction pyramid, mu = 0.4
            f = w["fr"][3 * k:3 * k + 3]
            assert f[2] >= -1e-9 and abs(f[0]) <= 0.4 * f[2] + 1e-8 and abs(f[1]) <= 0.4 * f[2] + 1e-8
        # fp32 evaluation (what the reference does) stays within 1e-4 * max(1,|tau|) of the fp64 one
        w32 = oracle.wbc_run(md, b["fb_state"][i], b["wbc_cmd"][i], dtype=np.float32)
        assert np.all(np.abs(w32["tau"] - w["tau"]) <= 1e-4 * np.maximum(1, np.abs(w["tau"])))


def test_golden_wbc(oracle):
    rows = golden_io.load("wbc_golden.npz")
    assert len(rows) == 48
    active = 0
    for r in rows:
        w = oracle.wbc_run(r["model"], r["fb_state"].astype(np.float64), r["wbc_cmd"].astype(np.float64),
                           r["prev"].astype(np.float64), dtype=np.float64)
        assert w["rc"] == 0
        assert np.abs(w["tau"] - r["tau64"]).max() <= 1e-9 and np.abs(w["qdes"] - r["qdes64"]).max() <= 1e-9
        assert np.abs(w["fr"] - r["fr64"]).max() <= 1e-8
        fb = oracle.fb_compute(r["model"], r["fb_state"].astype(np.float64), np.float64)
        for k in ("H", "G", "C", "Jc", "Jcdqd", "pGC"):
            assert np.abs(fb[k] - r[k]).max() <= 1e-10
        active += int(r["n_active"][0]) > 0
    assert active >= 2        # the fixtures exercise binding friction rows


def test_motor_tail_of_the_tick(oracle, pkg):
    """K14 tail as the locomotion state applies it (qr_fsm_state_locomotion.cpp:131-156, qr_safety_checker.cpp:48-66): +-0.9 N m on every abad
    motor before the WBC overwrites its stance legs, then the +-23 N m clip."""
    b = pkg.make_batch(32, 10, "a1", seed=5)
    args = (pkg.mpc_cfg("a1"), 10, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"])
    _, tau0, st, _, _ = oracle.tick_batch(1, *args, b["prev_ori_vel"].copy())
    _, tau3, _, _, _ = oracle.tick_batch(1, *args, b["prev_ori_vel"].copy(), epilogue=3)
    _, tau1, _, _, _ = oracle.tick_batch(1, *args, b["prev_ori_vel"].copy(), epilogue=1)
    stance = np.repeat(b["wbc_cmd"][:, 63:67] != 0, 3, axis=1)
    comp = np.tile(np.array([-0.9, 0, 0, 0.9, 0, 0], np.float32), 2)
    expect = np.where(stance, tau0, (tau0.astype(np.float64) + comp).astype(np.float32))
    assert np.array_equal(tau1, expect)
    assert np.array_equal(tau3, np.clip(expect, -23, 23))
    # MPC-only tick: nothing overwrites, the compensation is on every leg
    _, m0, _, _, _ = oracle.tick_batch(0, *args, b["prev_ori_vel"].copy())
    _, m1, _, _, _ = oracle.tick_batch(0, *args, b["prev_ori_vel"].copy(), epilogue=1)
    assert np.array_equal(m1, (m0.astype(np.float64) + comp).astype(np.float32))


def test_relaxation_qp_as_called_vs_compiled_quadprogpp():
    """wbc_golden.npz holds, per case, the relaxation QP exactly as the tick assembles it (qr_wholebody_impulse_ctrl.cpp:129-167, 232-247; float
    and double assembly) with z from the reference's QuadProg++ called as :113 calls it, and tau / optimalFr of the tick finished with that z.
    The oracle re-assembles the same QP bit for bit, its own solver lands on QuadProg++'s z (G is diagonal and positive: one optimum, no
    H <-> H^T question), and so do its torques."""
    import golden_io
    import oracle_py as O
    rows = golden_io.load("wbc_golden.npz")
    assert len(rows) == 48
    active = 0
    for r in rows:
        md = r["model"]
        for tag, dt in (("32", np.float32), ("64", np.float64)):
            q = O.wbc_qp(md, r["fb_state"], r["wbc_cmd"], r["prev"], dtype=dt)
            for k in ("G", "g0", "CE", "ce0", "CI", "ci0"):
                assert np.array_equal(q[k], r["qp%s_%s" % (tag, k)]), (tag, k)
            assert np.allclose(np.diag(np.diag(q["G"])), q["G"]) and np.all(np.diag(q["G"]) > 0)
            zs = max(1.0, np.abs(r["z_quadprogpp%s" % tag]).max())
            assert np.abs(q["z"] - r["z_quadprogpp%s" % tag]).max() <= 1e-10 * zs
            assert np.all(np.abs(q["tau"].astype(np.float64) - r["tau_quadprogpp%s" % tag]) <= 1e-6 * np.maximum(1.0, np.abs(r["tau_quadprogpp%s" % tag])))
            assert np.all(np.abs(q["fr"].astype(np.float64) - r["fr_quadprogpp%s" % tag]) <= 1e-6 * np.maximum(1.0, np.abs(r["fr_quadprogpp%s" % tag])))
            fin = O.wbc_qp(md, r["fb_state"], r["wbc_cmd"], r["prev"], dtype=dt, z_in=r["z_quadprogpp%s" % tag])
            assert np.array_equal(fin["tau"], r["tau_quadprogpp%s" % tag]) and np.array_equal(fin["fr"], r["fr_quadprogpp%s" % tag])
        active += int(r["n_active"][0]) > 0
    assert active >= 24           # the MPC-fed cases end with up to 7 inequality rows active


def test_live_quadprogpp_on_the_wbc_qp(ref, pkg):
    """The same against the compiled QuadProg++ itself (oracle/_ref, build container only) on fresh tick-fed cases."""
    O = ref
    b = pkg.make_batch(16, 10, "a1", seed=2077)
    md, cfg = pkg.model_desc("a1"), pkg.mpc_cfg("a1")
    for i in range(16):
        u, _, rc = O.mpc_solve(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        cmd = b["wbc_cmd"][i].copy(); cmd[51:63] = u[:12]
        q = O.wbc_qp(md, b["fb_state"][i], cmd, b["prev_ori_vel"][i], dtype=np.float32)
        z, fval = O.ref_quadprog(q["G"], q["g0"], q["CE"], q["ce0"], q["CI"], q["ci0"])
        assert np.isfinite(fval) and np.abs(z - q["z"]).max() <= 1e-10 * max(1.0, np.abs(z).max())


def test_config0_fixture_is_what_the_oracle_gives(pkg):
    """tests/golden/config0_a1_h10_2000.npz (the 2 000-tick single-robot sequence of BASELINE.json configs[0]): regenerated inputs match the
    stored checksums and the oracle, driven with the reference's cadence, reproduces the stored forces and leg commands."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("make_config0", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_config0.py"))
    M = importlib.util.module_from_spec(spec); spec.loader.exec_module(M)
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config0_a1_h10_2000.npz"))
    seq = M.stream(pkg)
    assert len(seq) == 2000
    forces, tua, n_active, iters = M.drive_oracle(pkg, seq)
    assert np.array_equal(forces, fx["mpc_forces"]) and np.array_equal(tua.astype(np.float32), fx["leg_cmd_tua"])
    assert forces.shape == (180, 12) and n_active.max() >= 20
    # temporally coherent: consecutive MPC solves 15 ticks (0.03 s) apart differ by a small fraction of the force scale
    late = forces[50:]
    assert np.median(np.abs(np.diff(late, axis=0)).max(1)) < 0.25 * np.abs(late).max()
