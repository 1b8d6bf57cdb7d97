"""CPU: the oracle's MPC assembly + solve (K1-K7).

Pins: golden vectors from the compiled reference qpOASES (tests/golden/mpc_golden.npz), live qpOASES when
oracle/_ref is present, float64 scipy expm for the closed-form discretisation, and the documented
ambiguity of the reference's own result (H is symmetric only to fp32 rounding)."""
import numpy as np
import pytest
from scipy.linalg import expm

import golden_io


def _rot(q):
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _skew(r):
    return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])


def _dense_float64_qp(cfg, h, s, traj, gait):
    """SolveMPC (qr_mpc_interface.cpp:359-425) written out literally in float64 with scipy's expm."""
    dt, mu, fmax, mass = [float(v) for v in cfg[:4]]
    Ib = np.diag(cfg[4:7].astype(np.float64)); w = np.r_[cfg[7:19].astype(np.float64), 0.0]; alpha = float(cfg[19])
    p, v, quat, om, r, rpy = s[0:3], s[3:6], s[6:10], s[10:13], s[13:25].reshape(4, 3), s[25:28]
    R = _rot(quat)
    Iw = R @ Ib @ R.T
    A = np.zeros((13, 13)); B = np.zeros((13, 12))
    A[3:6, 9:12] = np.eye(3); A[11, 12] = 1; A[0:3, 6:9] = R.T
    for b in range(4):
        B[6:9, 3 * b:3 * b + 3] = np.linalg.inv(Iw) @ _skew(r[b].astype(np.float64)); B[9:12, 3 * b:3 * b + 3] = np.eye(3) / mass
    M = np.zeros((25, 25)); M[:13, :13] = A; M[:13, 13:] = B
    E = expm(M * dt); Adt = E[:13, :13]; Bdt = E[:13, 13:]
    pw = [np.eye(13)]
    for i in range(h):
        pw.append(Adt @ pw[-1])
    Aqp = np.vstack(pw[1:]); Bqp = np.zeros((13 * h, 12 * h))
    for rr in range(h):
        for c in range(rr + 1):
            Bqp[13 * rr:13 * rr + 13, 12 * c:12 * c + 12] = pw[rr - c] @ Bdt
    x0 = np.r_[rpy, p, om, v, -9.8].astype(np.float64)
    Xd = np.zeros(13 * h)
    for i in range(h):
        Xd[13 * i:13 * i + 12] = traj[12 * i:12 * i + 12]
    L = np.diag(np.tile(w, h))
    H = 2 * (Bqp.T @ L @ Bqp + alpha * np.eye(12 * h))
    g = 2 * Bqp.T @ L @ (Aqp @ x0 - Xd)
    return H, g


def test_assembly_matches_float64_dense_formula(oracle, pkg):
    """Closed-form fp32 assembly == literal dense float64 evaluation (expm + powers) to fp32 rounding."""
    cfg = pkg.mpc_cfg("a1")
    for h, seed in ((10, 5), (5, 6), (16, 7)):
        b = pkg.make_batch(3, h, "a1", seed=seed)
        for i in range(3):
            H, g, ub = oracle.mpc_assemble(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            H64, g64 = _dense_float64_qp(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            assert np.abs(H - H64).max() <= 3e-6 * np.abs(H64).max()
            assert np.abs(g - g64).max() <= 3e-6 * np.abs(g64).max()
            # literal fp32 route (Pade expm + repeated products) agrees to fp32 rounding as well
            Hl, gl, _ = oracle.mpc_assemble(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i], literal=True)
            assert np.abs(Hl - H64).max() <= 5e-6 * np.abs(H64).max()
            assert np.array_equal(ub.reshape(-1, 5)[:, 4], b["gait"][i] * cfg[2]) and np.all(ub.reshape(-1, 5)[:, :4] == np.float32(5e10))


def test_golden_mpc_vs_reference_qpoases(oracle, pkg):
    """Golden vectors: forces from the reference's compiled qpOASES.
      * symmetric data (H+H')/2, converged: the oracle must agree to 1e-7 relative (same unique optimum);
      * exactly as the reference calls it (fp32-asymmetric H, nWSR=100): agreement only within the
        reference's own ambiguity, measured by handing qpOASES H' instead of H."""
    rows = golden_io.load("mpc_golden.npz")
    assert len(rows) == 86          # 62 of rounds 1-2 + robots 0..23 of bench.py's own first batch (round 3)
    worst_sym, worst_called, worst_ambig = 0.0, 0.0, 0.0
    n_cap = 0
    for r in rows:
        h = int(r["h"])
        u, st, rc = oracle.mpc_solve(r["cfg"], h, r["mpc_state"], r["traj"], r["gait"])
        assert rc == 0
        scale = max(1.0, np.abs(r["u_qpoases_sym"]).max())
        assert np.abs(u - r["u_oracle"]).max() <= 1e-9 * scale            # the oracle itself is stable
        e = np.abs(u - r["u_qpoases_sym"]).max() / scale
        worst_sym = max(worst_sym, e)
        H, g, ub = oracle.mpc_assemble(r["cfg"], h, r["mpc_state"], r["traj"], r["gait"])
        assert np.array_equal(g, r["g"]) and np.array_equal(H[:12, :12], r["H_first_block"])
        Hd = H.astype(np.float64)
        assert np.allclose([Hd.sum(), np.abs(Hd).sum(), np.trace(Hd)], r["H_checksum"], rtol=1e-12)
        tau = oracle.mpc_force_to_torque(pkg.model_desc(str(r["robot"]))[:3], r["quat"], r["q"], u[:12])
        assert np.abs(tau - r["tau_oracle"]).max() <= 1e-6
        if int(r["qpoases_as_called_nwsr"][1]) == 0 and int(r["qpoases_as_called_nwsr"][0]) < 100:     # the reference converged within nWSR = 100
            f0 = max(1.0, np.abs(u[:12]).max())
            ec = np.abs(u[:12] - r["f_qpoases_as_called"]).max() / f0
            ea = np.abs(r["f_qpoases_transposed"] - r["f_qpoases_as_called"]).max() / f0
            worst_called = max(worst_called, ec)
            worst_ambig = max(worst_ambig, ea)
            # row by row: within the reference's own H <-> H^T ambiguity, in fact its midpoint
            assert ec <= 0.55 * ea + 1e-5, (h, ec, ea)
            assert np.abs(u[:12] - 0.5 * (r["f_qpoases_as_called"] + r["f_qpoases_transposed"])).max() <= 1e-5 * f0
        else:
            n_cap += 1
    assert worst_sym <= 1e-7, worst_sym
    # the oracle sits between qpOASES(H) and qpOASES(H'); both are "the reference"
    assert worst_called <= max(0.55 * worst_ambig, 1e-6), (worst_called, worst_ambig)
    assert worst_ambig < 2e-2
    assert n_cap == 5                   # rows of the fixture on which the reference's call ran into its nWSR = 100 cap (all at h = 16)


def test_parity_table_matches_fixture(oracle, pkg):
    """tests/golden/parity_as_called.json (the tables of DESIGN.md 2) is what the fixture's rows give -- first-step forces, the MPC-only J^T f
    torque and the full tick's K14 torque (the metric's own quantity), the latter recomputed here through the oracle's tick tail from the
    stored forces -- and says what it is quoted for: against the reference's solver AS CALLED the 1e-4 relative torque of north_star is not
    met (3.6e-2 on bench.py's own batch), and is not met by the reference against itself either: handed the same QP assembled by its other
    fp32 route (a few ulps of H and g) its own full-tick torque moves by up to 4e-2, beyond 1e-4 on more rows than ours."""
    import json, os
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_as_called.json")))
    rows = golden_io.load("mpc_golden.npz")
    rel_t = lambda a, b: float((np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))).max())
    for key, sel in (("h5", lambda r: int(r["h"]) == 5), ("h10", lambda r: int(r["h"]) == 10), ("h16", lambda r: int(r["h"]) == 16),
                     ("h10_bench_batch", lambda r: int(r["bench_batch"][0]) == 1)):
        rs = [r for r in rows if sel(r) and int(r["qpoases_as_called_nwsr"][1]) == 0 and int(r["qpoases_as_called_nwsr"][0]) < 100]
        wf = max(np.abs(r["f_oracle"] - r["f_qpoases_as_called"]).max() / max(1.0, np.abs(r["f_qpoases_as_called"]).max()) for r in rs)
        assert abs(wf - tab[key]["max_rel_force"]) <= 1e-12 and tab[key]["converged"] == len(rs)
        wk = 0.0
        for r in rs:
            geom, md = r["model"][:3], r["model"]
            for name, f in (("tau_tick_oracle", r["f_oracle"]), ("tau_tick_as_called", r["f_qpoases_as_called"]), ("tau_tick_transposed", r["f_qpoases_transposed"])):
                t, _ = oracle.tick_from_forces(geom, md, r["fb_state"], r["wbc_cmd"], r["prev"], f, 1, 3)
                assert np.array_equal(t, r[name]), (key, name)
            wk = max(wk, rel_t(r["tau_tick_oracle"], r["tau_tick_as_called"]))
            # the full tick's torque in the kernel's arithmetic (WBC in double) is the float one to the fp32 oracle's own 1e-4 bar
            assert np.all(np.abs(r["tau_tick_oracle64"] - r["tau_tick_oracle"]) <= 1e-4 * np.maximum(1.0, np.abs(r["tau_tick_oracle"])))
        assert abs(wk - tab[key]["max_rel_tick_torque"]) <= 1e-12
    assert tab["h10_bench_batch"]["rows"] == 24 and tab["h10_bench_batch"]["converged"] == 24
    assert 2e-3 < tab["h10"]["max_rel_force"] < 3e-3 and tab["h16"]["max_rel_force"] < 1.3e-3      # (h = 10: bench.py's batch is the worst group, 2.6e-3)
    # what the statement of DESIGN.md 2 rests on
    for key in ("h10", "h16"):
        t = tab[key]
        assert t["max_rel_tick_torque"] > 1e-4                                          # north_star's bar is missed against the as-called answer ...
        assert t["max_rel_tick_torque"] <= 0.55 * t["ambiguity_rel_tick_torque"]        # ... by half the reference's own H <-> H^T spread ...
        assert t["literal_route_rel_tick_torque"] > 1e-2                                # ... and the reference misses it against itself by more,
        assert t["reference_rows_above_1e_4_tick_torque_between_its_routes"] >= t["rows_above_1e_4_tick_torque"]    # on at least as many rows


def test_live_qpoases_symmetric(ref, pkg):
    cfg = pkg.mpc_cfg("a1")
    b = pkg.make_batch(8, 10, "a1", seed=77, excite=0.5)
    A = ref.mpc_constraint_matrix(10)
    for i in range(8):
        H, g, ub = ref.mpc_assemble(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        u, st, rc = ref.mpc_solve(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        Hd = H.astype(np.float64)
        x, info = ref.ref_qpoases_mpc(0.5 * (Hd + Hd.T), g.astype(np.float64), A, np.zeros(200), ub, nWSR=2000)
        assert info["init_rc"] == 0 and rc == 0
        assert np.abs(u - x).max() <= 1e-7 * max(1.0, np.abs(x).max())


def test_swing_variables_are_zero_and_constraints_hold(oracle, pkg):
    cfg = pkg.mpc_cfg("a1")
    b = pkg.make_batch(16, 10, "a1", seed=12)
    for i in range(16):
        u, st, rc = oracle.mpc_solve(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        f = u.reshape(40, 3); gt = b["gait"][i]
        assert np.all(f[gt == 0] == 0)
        assert np.all(f[:, 2] >= -1e-9) and np.all(f[:, 2] <= np.float64(cfg[2]) * gt + 1e-7)
        assert np.all(np.abs(f[:, 0]) <= 0.45 * f[:, 2] + 1e-7) and np.all(np.abs(f[:, 1]) <= 0.45 * f[:, 2] + 1e-7)


def test_leg_kinematics(oracle, pkg):
    """AnalyticalLegJacobian is the derivative of FootPositionInHipFrame (QS/robots/qr_robot.cpp:127-172)."""
    geom = pkg.model_desc("a1")[:3]
    hip = np.array(pkg.ROBOTS["a1"]["hip_offset"], np.float32).reshape(12)
    rng = np.random.default_rng(1)
    q = (np.tile([0.0, 0.8, -1.6], 4) + rng.uniform(-0.2, 0.2, 12)).astype(np.float64)
    p0 = oracle.foot_positions(geom, hip, q).astype(np.float64)
    for leg in range(4):
        J = oracle.leg_jacobian(geom, q[3 * leg:3 * leg + 3], leg)
        for j in range(3):
            dq = q.copy(); dq[3 * leg + j] += 1e-3
            num = (oracle.foot_positions(geom, hip, dq).astype(np.float64) - p0)[3 * leg:3 * leg + 3] / 1e-3
            assert np.abs(num - J[:, j]).max() < 2e-3


def test_solve_from_given_hessian_and_gradient(oracle, pkg):
    """mpc_solve_hg -- the checker of BASELINE configs[4]'s bf16-limb Hessian mode (tests/test_gpu_mpc.py::test_bf16x3_solve_vs_oracle_on_its_own_hessian):
    the stated QP solved from a GIVEN fp32 (H, g).  Handed the oracle's own assembly it is mpc_solve bit for bit; entries of swing variables are
    never read; another rounding of H (one ulp on every entry) gives that other QP's optimum -- which the compiled qpOASES confirms on the
    symmetrised data when oracle/_ref is there."""
    ref_ok = oracle.ref() is not None
    for name, h, seed in (("a1", 10, 41), ("lite3", 16, 42)):
        cfg = pkg.mpc_cfg(name)
        b = pkg.make_batch(4, h, name, seed=seed)
        for i in range(4):
            s, traj, gait = b["mpc_state"][i], b["traj"][i], b["gait"][i]
            H, g, ub = oracle.mpc_assemble(cfg, h, s, traj, gait)
            u0, _, rc0 = oracle.mpc_solve(cfg, h, s, traj, gait)
            u1, _, rc1 = oracle.mpc_solve_hg(cfg, h, gait, H, g)
            assert rc0 == 0 and rc1 == 0 and np.array_equal(u0, u1)
            free = np.repeat(gait > 0, 3)
            Hn = H.copy(); gn = g.copy()
            Hn[~free, :] = np.nan; Hn[:, ~free] = np.nan; gn[~free] = np.nan
            u2, _, rc2 = oracle.mpc_solve_hg(cfg, h, gait, Hn, gn)
            assert rc2 == 0 and np.array_equal(u0, u2)
            # another rounding of H: every entry one ulp up
            Hp = np.nextafter(H, np.float32(np.inf))
            u3, _, rc3 = oracle.mpc_solve_hg(cfg, h, gait, Hp, g)
            assert rc3 == 0 and np.all(np.isfinite(u3))
            if ref_ok and i < 2:
                Hs = 0.5 * (Hp.astype(np.float64) + Hp.astype(np.float64).T)
                A = oracle.mpc_constraint_matrix(h, float(cfg[1]))
                xq, info = oracle.ref_qpoases_mpc(Hs, g.astype(np.float64), A, np.zeros(20 * h), ub.astype(np.float64), nWSR=2000)
                assert info["init_rc"] == 0
                assert np.abs(xq - u3).max() <= 2e-6 * max(1.0, np.abs(u3).max()), (name, i, np.abs(xq - u3).max())
