"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for oracle/_ref):

    python tests/golden/make_golden.py

Fixtures are data only: seeded synthetic inputs (quadruped-robot_amd/workload.py) and the outputs of
  * the reference's vendored solvers compiled from /root/reference (oracle/_ref):
      qpOASES 3.2.0 called exactly as qr_mpc_interface.cpp:428-438 does, and with nWSR raised;
      QuadProg++ solve_quadprog as qr_wholebody_impulse_ctrl.cpp:113 does;
  * our CPU restatement (oracle/libqr_oracle.so).
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O   # noqa: E402

spec = importlib.util.spec_from_file_location("workload", os.path.join(ROOT, "quadruped-robot_amd", "workload.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)
OUT = os.path.dirname(os.path.abspath(__file__))


def mpc_cases():
    rows = []
    # (the first five groups are round 1's 30 rows, unchanged; the h = 16 / full-range groups after them include rows on which the
    # reference's qpOASES runs into its nWSR = 100 cap: qpoases_as_called_nwsr[0] == 100 or init_rc != 0 -- kept, and marked.  The last
    # group (round 3) is robots 0..23 of bench.py's own first batch -- make_batch(1024, seed = 0xA1 + 2, excite = 1.0), the 24 robots its
    # config.parity_vs_reference_as_called figure is taken on -- so that the committed table covers what the bench line reports.)
    groups = [("a1", 10, 12, 1001, 1.0, None), ("a1", 10, 6, 1002, 0.3, None), ("a1", 5, 4, 1003, 1.0, None),
              ("a1", 16, 4, 1004, 0.3, None), ("lite3", 10, 4, 1005, 1.0, None),
              ("a1", 16, 14, 1006, 1.0, None), ("lite3", 16, 6, 1007, 1.0, None), ("a1", 5, 4, 1008, 0.3, None), ("a1", 10, 8, 1009, 1.0, None),
              ("a1", 10, 1024, 0xA1 + 2, 1.0, 24)]
    for robot, h, n, seed, excite, take in groups:
        b = W.make_batch(n, h, robot, seed=seed, excite=excite)
        cfg = W.mpc_cfg(robot)
        md = W.model_desc(robot)
        A = O.mpc_constraint_matrix(h)
        for i in range(n if take is None else take):
            H, g, ub = O.mpc_assemble(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            u, st, rc = O.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            assert rc == 0
            Hd = H.astype(np.float64); gd = g.astype(np.float64)
            x_ref, info = O.ref_qpoases_mpc(Hd, gd, A, np.zeros(20 * h), ub, nWSR=100)          # as the reference calls it
            x_avg, info_avg = O.ref_qpoases_mpc(0.5 * (Hd + Hd.T), gd, A, np.zeros(20 * h), ub, nWSR=2000)   # converged, symmetric data
            x_t, info_t = O.ref_qpoases_mpc(Hd.T.copy(), gd, A, np.zeros(20 * h), ub, nWSR=2000)
            # the same call on the same QP assembled by the OTHER fp32 route (Pade expm + repeated products as qr_mpc_interface.cpp:257-293
            # evaluates them, restated in mpc_assemble_literal): a few ulps away in H and g, as another Eigen build of the reference would be
            Hl, gl, _ = O.mpc_assemble(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i], literal=True)
            x_lit, info_lit = O.ref_qpoases_mpc(Hl.astype(np.float64), gl.astype(np.float64), A, np.zeros(20 * h), ub, nWSR=100)
            assert info_avg["init_rc"] == 0
            geom = md[:3]
            fbs, cmd, prev = b["fb_state"][i], b["wbc_cmd"][i], b["prev_ori_vel"][i]
            tq = lambda f12: O.mpc_force_to_torque(geom, fbs[0:4], fbs[13:25], f12)
            # the metric's own quantity: the K14 torque of the FULL tick (MPC forces -> WBC with Fr_des := forces -> stance / swing merge,
            # abad compensation, clip; qr_mpc_interface.cpp:428-438 -> qr_wbc_locomotion_controller.cpp:108-135 -> qr_fsm_state_locomotion.cpp:141-151),
            # the WBC in float as the reference computes it
            tk = lambda f12: O.tick_from_forces(geom, md, fbs, cmd, prev, f12, 1, 3)[0]
            rows.append(dict(robot=robot, h=h, cfg=cfg, mpc_state=b["mpc_state"][i], traj=b["traj"][i], gait=b["gait"][i],
                             q=fbs[13:25], quat=fbs[0:4], fb_state=fbs, wbc_cmd=cmd, prev=prev, model=md,
                             g=g, H_checksum=np.array([H.astype(np.float64).sum(), np.abs(H.astype(np.float64)).sum(), np.trace(Hd)]),
                             H_first_block=H[:12, :12].copy(),
                             f_oracle=u[:12].copy(), u_oracle_norm=np.array([np.linalg.norm(u)]),
                             f_qpoases_as_called=x_ref[:12].copy(), qpoases_as_called_nwsr=np.array([info["nWSR"], info["init_rc"]]),
                             f_qpoases_sym=x_avg[:12].copy(), u_qpoases_sym=x_avg.copy(), u_oracle=u.copy(),
                             f_qpoases_transposed=x_t[:12].copy(), tau_oracle=tq(u[:12]), n_active=np.array([st["n_active"]]),
                             f_qpoases_literal=x_lit[:12].copy(), qpoases_literal_nwsr=np.array([info_lit["nWSR"], info_lit["init_rc"]]),
                             tau_qpoases_as_called=tq(x_ref[:12]), tau_qpoases_transposed=tq(x_t[:12]), tau_qpoases_literal=tq(x_lit[:12]),
                             tau_tick_oracle=tk(u[:12]), tau_tick_as_called=tk(x_ref[:12]), tau_tick_transposed=tk(x_t[:12]),
                             tau_tick_literal=tk(x_lit[:12]),
                             tau_tick_oracle64=O.tick_from_forces(geom, md, fbs, cmd, prev, u[:12], 1, 3, wbc_fp64=True)[0],
                             excite=np.array([excite]), bench_batch=np.array([0 if take is None else 1])))
    return rows


def wbc_cases():
    rows = []
    rng = np.random.default_rng(77)
    # (groups three and four, round 3: Fr_des := the MPC's own first-step forces, as the tick feeds the WBC -- qr_mpc_stance_leg_controller.cpp:408 --
    # at SURVEY 8d's full excitation: the MPC's pyramid (mu = 0.45) is wider than the WBC's (0.4), so 0-7 inequality rows end up active)
    for robot, n, seed, from_mpc in (("a1", 12, 2001, False), ("lite3", 4, 2002, False), ("a1", 24, 2003, True), ("lite3", 8, 2004, True)):
        b = W.make_batch(n, 10, robot, seed=seed)
        md = W.model_desc(robot)
        for i in range(n):
            cmd = b["wbc_cmd"][i].copy()
            if from_mpc:
                u, _, rc = O.mpc_solve(W.mpc_cfg(robot), 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
                assert rc == 0
                cmd[51:63] = u[:12]
            elif i % 3 == 1:       # Fr_des on the MPC pyramid edge -> WBC inequalities bind
                fr = cmd[51:63].reshape(4, 3)
                fr[:, 0] = 0.45 * fr[:, 2]; fr[:, 1] = -0.3 * fr[:, 2]
                cmd[51:63] = fr.reshape(12)
            if not from_mpc and i % 4 == 3:
                cmd[63:67] = (1, 1, 1, 1); cmd[51:63] = np.tile(np.array([2.0, -1.0, 33.0], np.float32), 4)
            prev = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
            r64 = O.wbc_run(md, b["fb_state"][i].astype(np.float64), cmd.astype(np.float64), prev.astype(np.float64), dtype=np.float64)
            r32 = O.wbc_run(md, b["fb_state"][i], cmd, prev, dtype=np.float32)
            fb = O.fb_compute(md, b["fb_state"][i].astype(np.float64), np.float64)
            row = dict(robot=robot, model=md, fb_state=b["fb_state"][i], wbc_cmd=cmd, prev=prev,
                       tau64=r64["tau"], tau32=r32["tau"], qdes64=r64["qdes"], qddes64=r64["qddes"], fr64=r64["fr"],
                       H=fb["H"], G=fb["G"], C=fb["C"], Jc=fb["Jc"], Jcdqd=fb["Jcdqd"], pGC=fb["pGC"],
                       n_active=np.array([r64["qp"]["n_active"]]))
            # The relaxation QP exactly as the tick assembles it (SetCost / SetEqualityConstraint / SetInequalityConstraint,
            # qr_wholebody_impulse_ctrl.cpp:129-167, 232-247) solved by the reference's QuadProg++ as :113 calls it, and the tick finished with
            # that z (GetSolution :210-228): `32` = assembled in float as the reference computes (then promoted to double for the solver, as
            # qpG / qpCE / qpCI are), `64` = the same formulas in double (what the kernel evaluates).  G is diagonal: no H <-> H^T question here.
            for tag, dt in (("32", np.float32), ("64", np.float64)):
                q = O.wbc_qp(md, b["fb_state"][i], cmd, prev, dtype=dt)
                z, fval = O.ref_quadprog(q["G"], q["g0"], q["CE"], q["ce0"], q["CI"], q["ci0"])
                assert np.isfinite(fval)
                fin = O.wbc_qp(md, b["fb_state"][i], cmd, prev, dtype=dt, z_in=z)
                row.update({"qp%s_G" % tag: q["G"], "qp%s_g0" % tag: q["g0"], "qp%s_CE" % tag: q["CE"], "qp%s_ce0" % tag: q["ce0"],
                            "qp%s_CI" % tag: q["CI"], "qp%s_ci0" % tag: q["ci0"], "z_oracle%s" % tag: q["z"], "z_quadprogpp%s" % tag: z,
                            "f_quadprogpp%s" % tag: np.array([fval]), "tau_quadprogpp%s" % tag: fin["tau"], "fr_quadprogpp%s" % tag: fin["fr"]})
            rows.append(row)
    return rows


def qp_cases():
    """WBC-shaped QPs solved by the reference's QuadProg++ (and the known-answer demo of main.cc)."""
    rng = np.random.default_rng(99)
    rows = []
    for nc in (0, 1, 2, 3, 4, 2, 2, 4):
        nz, p, m = 6 + 3 * nc, 6, max(6 * nc, 1)
        G = np.diag(np.r_[np.full(6, 0.1), np.ones(3 * nc)])
        CE = np.zeros((nz, p))
        A6 = rng.normal(size=(6, 6)); A6 = A6 @ A6.T + 6 * np.eye(6)
        CE[:6, :] = A6.T
        if nc:
            CE[6:, :] = rng.normal(size=(3 * nc, 6))
        ce0 = rng.normal(size=p) * 5
        CI = np.zeros((nz, m)); ci0 = np.zeros(m)
        mu = 0.4
        Uf = np.array([[0, 0, 1], [1, 0, mu], [-1, 0, mu], [0, 1, mu], [0, -1, mu], [0, 0, -1.0]])
        for k in range(nc):
            CI[6 + 3 * k:9 + 3 * k, 6 * k:6 * k + 6] = Uf.T
            fr = np.array([rng.normal() * 8, rng.normal() * 8, 30 + rng.normal() * 10])
            ci0[6 * k:6 * k + 6] = Uf @ fr - np.array([0, 0, 0, 0, 0, -132.4])
        x_ref, f_ref = O.ref_quadprog(G, np.zeros(nz), CE, ce0, CI, ci0)
        rows.append(dict(G=G, g0=np.zeros(nz), CE=CE, ce0=ce0, CI=CI, ci0=ci0, x_quadprogpp=x_ref, f_quadprogpp=np.array([f_ref])))
    # QX/QuadProgpp/src/main.cc:8-20 : x = [1, 2], f = 12
    G = np.array([[4, -2], [-2, 4.0]]); g0 = np.array([6.0, 0]); CE = np.array([[1.0], [1.0]]); ce0 = np.array([-3.0])
    CI = np.array([[1, 0, 1], [0, 1, 1.0]]); ci0 = np.array([0, 0, -2.0])
    x_ref, f_ref = O.ref_quadprog(G, g0, CE, ce0, CI, ci0)
    rows.append(dict(G=G, g0=g0, CE=CE, ce0=ce0, CI=CI, ci0=ci0, x_quadprogpp=x_ref, f_quadprogpp=np.array([f_ref])))
    return rows


def vmc_golden():
    """Force-balance QP: fp32 data from our assembly, x from the reference's QuadProg++ called as qr_qp_torque_optimizer.cpp:242-276 does."""
    cfg = W.vmc_cfg("a1"); geom = W.model_desc("a1")[:3]
    vin, q = W.make_vmc_batch(48, sloped=0.25, seed=3003)
    Gs, As, xs, infs = [], [], [], []
    for i in range(48):
        G, a, CI, b = O.vmc_assemble(cfg, vin[i])
        x, f = O.ref_quadprog(G.T.astype(np.float64), -a.astype(np.float64), np.zeros((12, 0)), np.zeros(0), CI.astype(np.float64), -b.astype(np.float64))
        Gs.append(G); As.append(a); xs.append(x); infs.append(not np.isfinite(f))
    np.savez_compressed(os.path.join(OUT, "vmc_golden.npz"), cfg=cfg, geom=geom, vin=vin, q=q, G=np.array(Gs), a=np.array(As),
                        x_quadprog=np.array(xs), quadprog_inf=np.array(infs))
    print("vmc_golden.npz: 48 cases, %d with QuadProg++ returning +inf" % int(np.sum(infs)))


def vmc_world_golden():
    """World-frame overload of the force-balance QP (:304-398): tilted base, per-leg force-window ratios; x from the reference's QuadProg++."""
    cfg = W.vmc_cfg("a1"); geom = W.model_desc("a1")[:3]
    vin, q, ratio = W.make_vmc_world_batch(40, seed=3004)
    Gs, As, bs, xs, infs = [], [], [], [], []
    for i in range(40):
        G, a, CI, b = O.vmc_assemble(cfg, vin[i], ratio[i])
        x, f = O.ref_quadprog(G.T.astype(np.float64), -a.astype(np.float64), np.zeros((12, 0)), np.zeros(0), CI.astype(np.float64), -b.astype(np.float64))
        Gs.append(G); As.append(a); bs.append(b); xs.append(x); infs.append(not np.isfinite(f))
    np.savez_compressed(os.path.join(OUT, "vmc_world_golden.npz"), cfg=cfg, geom=geom, vin=vin, q=q, ratio=ratio, G=np.array(Gs), a=np.array(As),
                        b=np.array(bs), x_quadprog=np.array(xs), quadprog_inf=np.array(infs))
    print("vmc_world_golden.npz: 40 cases, %d with QuadProg++ returning +inf" % int(np.sum(infs)))


def ekf_golden():
    """States of the reference's TinyEKF<3,3> (compiled from /root/reference) over three 200-step sequences."""
    rng = np.random.default_rng(4004)
    out = {}
    for j, (av, sv) in enumerate(((0.1, 0.1), (0.01, 0.5), (1.0, 0.02))):
        dv = 0.01 * rng.standard_normal((200, 3)); z = np.cumsum(dv, 0) + 0.05 * rng.standard_normal((200, 3))
        x, bad = O.ref_tinyekf_run(av, sv, dv, z)
        assert bad == 0
        out["var%d" % j] = np.array([av, sv], np.float32); out["dv%d" % j] = dv; out["z%d" % j] = z; out["x%d" % j] = x
    np.savez_compressed(os.path.join(OUT, "ekf_golden.npz"), **out)
    print("ekf_golden.npz: 3 sequences x 200 steps")


def save(name, rows):
    flat = {"count": np.array([len(rows)])}
    for i, r in enumerate(rows):
        for k, v in r.items():
            flat["%03d_%s" % (i, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name), **flat)
    print(name, len(rows), "cases", os.path.getsize(os.path.join(OUT, name)) // 1024, "KiB")


def parity_table(rows):
    """Worst deviation of the oracle (= the kernel to 2e-6) from the reference's solver exactly as the reference calls it, per horizon,
    over the rows on which that call converged: first-step forces, the MPC-only J^T f torque, and -- the metric's own quantity -- the K14
    torque of the full tick.  Beside it how far the reference's OWN answer moves on the same rows when it is handed H^T (the same matrix up
    to the fp32 rounding of its assembly), or the QP assembled by its other fp32 route (`literal`: a few ulps of H and g, as another Eigen
    build would give).  The rows of bench.py's own batch also appear as a group of their own ("h10_bench_batch").  -> dict, also written to
    parity_as_called.json (the tables of DESIGN.md 2)."""
    import json
    tab = {}

    def rel_t(a, b):
        return float((np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))).max())

    for r in rows:
        h = int(r["h"])
        keys = ["h%d" % h] + (["h%d_bench_batch" % h] if int(r["bench_batch"][0]) else [])
        nwsr, rc = int(r["qpoases_as_called_nwsr"][0]), int(r["qpoases_as_called_nwsr"][1])
        for key in keys:
            t = tab.setdefault(key, dict(rows=0, converged=0, nwsr_cap_or_failed=0, max_rel_force=0.0, max_rel_torque=0.0, max_rel_tick_torque=0.0,
                                         ambiguity_rel_force=0.0, ambiguity_rel_torque=0.0, ambiguity_rel_tick_torque=0.0,
                                         literal_route_rel_force=0.0, literal_route_rel_torque=0.0, literal_route_rel_tick_torque=0.0, literal_route_rows=0,
                                         rows_above_1e_4_torque=0, rows_above_1e_4_tick_torque=0, reference_rows_above_1e_4_tick_torque_between_its_routes=0))
            t["rows"] += 1
            if rc != 0 or nwsr >= 100:
                t["nwsr_cap_or_failed"] += 1
                continue
            t["converged"] += 1
            fc = r["f_qpoases_as_called"]
            fs = max(1.0, np.abs(fc).max())
            ef = np.abs(r["f_oracle"] - fc).max() / fs
            et = rel_t(r["tau_oracle"], r["tau_qpoases_as_called"])
            ek = rel_t(r["tau_tick_oracle"], r["tau_tick_as_called"])
            t["max_rel_force"] = max(t["max_rel_force"], float(ef)); t["max_rel_torque"] = max(t["max_rel_torque"], et)
            t["max_rel_tick_torque"] = max(t["max_rel_tick_torque"], ek)
            t["ambiguity_rel_force"] = max(t["ambiguity_rel_force"], float(np.abs(r["f_qpoases_transposed"] - fc).max() / fs))
            t["ambiguity_rel_torque"] = max(t["ambiguity_rel_torque"], rel_t(r["tau_qpoases_transposed"], r["tau_qpoases_as_called"]))
            t["ambiguity_rel_tick_torque"] = max(t["ambiguity_rel_tick_torque"], rel_t(r["tau_tick_transposed"], r["tau_tick_as_called"]))
            t["rows_above_1e_4_torque"] += int(et > 1e-4); t["rows_above_1e_4_tick_torque"] += int(ek > 1e-4)
            lw, lrc = int(r["qpoases_literal_nwsr"][0]), int(r["qpoases_literal_nwsr"][1])
            if lrc == 0 and lw < 100:
                t["literal_route_rows"] += 1
                t["literal_route_rel_force"] = max(t["literal_route_rel_force"], float(np.abs(r["f_qpoases_literal"] - fc).max() / fs))
                t["literal_route_rel_torque"] = max(t["literal_route_rel_torque"], rel_t(r["tau_qpoases_literal"], r["tau_qpoases_as_called"]))
                lk = rel_t(r["tau_tick_literal"], r["tau_tick_as_called"])
                t["literal_route_rel_tick_torque"] = max(t["literal_route_rel_tick_torque"], lk)
                t["reference_rows_above_1e_4_tick_torque_between_its_routes"] += int(lk > 1e-4)
    out = {k: tab[k] for k in sorted(tab)}
    json.dump(out, open(os.path.join(OUT, "parity_as_called.json"), "w"), indent=1)
    return out


def nwsr_cap_census(n=40):
    """How often the reference's solve runs into its own nWSR = 100 cap on SURVEY.md 8d's input ranges: n seeded A1 robots per
    horizon, qpOASES called exactly as qr_mpc_interface.cpp:428-438 does.  Written to nwsr_cap_census.json."""
    import json
    out = {}
    for h in (10, 16):
        b = W.make_batch(n, h, "a1", seed=0xA1 + 2, excite=1.0)
        cfg = W.mpc_cfg("a1")
        A = O.mpc_constraint_matrix(h)
        hits, fails, nws = 0, 0, []
        for i in range(n):
            H, g, ub = O.mpc_assemble(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            x, info = O.ref_qpoases_mpc(H.astype(np.float64), g.astype(np.float64), A, np.zeros(20 * h), ub, nWSR=100)
            nws.append(info["nWSR"]); hits += int(info["nWSR"] >= 100 or info["init_rc"] != 0); fails += int(info["init_rc"] != 0)
        out["h%d" % h] = dict(robots=n, seed=0xA1 + 2, excite=1.0, hit_cap_or_failed=hits, init_returned_error=fails,
                             nwsr_median=float(np.median(nws)), nwsr_max=int(max(nws)))
    json.dump(out, open(os.path.join(OUT, "nwsr_cap_census.json"), "w"), indent=1)
    return out


if __name__ == "__main__":
    assert O.ref() is not None, "oracle/_ref is required to generate fixtures"
    if len(sys.argv) > 1 and sys.argv[1] == "wbc":
        save("wbc_golden.npz", wbc_cases())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mpc":          # only the MPC fixture and the artefacts derived from it
        rows = mpc_cases()
        save("mpc_golden.npz", rows)
        print(parity_table(rows))
        print(nwsr_cap_census())
        sys.exit(0)
    rows = mpc_cases()
    save("mpc_golden.npz", rows)
    parity_table(rows)
    nwsr_cap_census()
    save("wbc_golden.npz", wbc_cases())
    vmc_golden()
    vmc_world_golden()
    ekf_golden()
    save("qp_golden.npz", qp_cases())
