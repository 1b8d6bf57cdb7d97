"""CPU: the oracle's force-balance (VMC) QP -- qr_qp_torque_optimizer.cpp:190-301 -- pinned against the reference's own QuadProg++
(compiled from /root/reference into oracle/_ref) and the committed golden vectors."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vmc_golden.npz")


def _ref_inputs(G, a, CI, b):
    # exactly the conversion of qr_qp_torque_optimizer.cpp:242-270: GG[i][j] = G(j,i), aa = -a, CICI = Ci, bb = -b
    return G.T.astype(np.float64), -a.astype(np.float64), np.zeros((12, 0)), np.zeros(0), CI.astype(np.float64), -b.astype(np.float64)


def test_vmc_assembly_properties(pkg, oracle):
    W = pkg.workload
    cfg = W.vmc_cfg("a1")
    vin, q = W.make_vmc_batch(32, sloped=0.5, seed=5)
    for i in range(32):
        G, a, CI, b = oracle.vmc_assemble(cfg, vin[i])
        # float64 evaluation of the same formulas
        R = vin[i, 22:31].reshape(3, 3).astype(np.float64)
        I = R @ np.diag([0.24, 0.80, 1.0]) @ R.T
        M = np.zeros((6, 12))
        for l in range(4):
            x = R @ vin[i, 3 * l:3 * l + 3].astype(np.float64)
            S = np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]])
            M[:3, 3 * l:3 * l + 3] = np.eye(3) / 13.0
            M[3:, 3 * l:3 * l + 3] = np.linalg.inv(I) @ S
        Q = np.diag(cfg[10:16].astype(np.float64))
        G64 = M.T @ Q @ M + 1e-4 + 1e-4 * np.eye(12)
        gv = np.concatenate([vin[i, 31:34], np.zeros(3)]).astype(np.float64)
        a64 = (gv + vin[i, 12:18]) @ Q @ M
        assert np.abs(G - G64).max() <= 2e-6 * np.abs(G64).max()
        assert np.abs(a - a64).max() <= 1e-5 * max(1.0, np.abs(a64).max())
        # constraint rows: every row touches one foot; swing feet carry the contradictory 1e-7 pair (:79-81)
        for l in range(4):
            assert np.count_nonzero(CI[:, 2 * l]) <= 3 and np.all(CI[:3 * l, 2 * l] == 0) and np.all(CI[3 * l + 3:, 2 * l] == 0)
            if vin[i, 18 + l] == 0:
                assert b[2 * l] == np.float32(1e-7) and b[2 * l + 1] == np.float32(1e-7)
            else:
                assert np.isclose(b[2 * l], 0.01 * 13 * 9.8) and np.isclose(b[2 * l + 1], -10 * 13 * 9.8)


def test_vmc_matches_reference_quadprog(pkg, oracle, ref):
    """x from the oracle == x from QuadProg++ as the reference calls it, including the 'infeasible' (+inf) returns of swing-foot ticks."""
    W = pkg.workload
    cfg = W.vmc_cfg("a1"); geom = pkg.model_desc("a1")[:3]
    vin, q = W.make_vmc_batch(150, sloped=0.3, seed=11)
    n_inf = 0
    for i in range(150):
        G, a, CI, b = oracle.vmc_assemble(cfg, vin[i])
        force, tau, x, st, rc = oracle.vmc_solve(cfg, geom, vin[i], q[i])
        xr, fr = oracle.ref_quadprog(*_ref_inputs(G, a, CI, b))
        assert np.abs(x - xr).max() <= 1e-8 * max(1.0, np.abs(xr).max()), i
        assert (rc == 1) == (not np.isfinite(fr))
        n_inf += rc == 1
        if rc == 0:       # converged: KKT of the mirrored-lower-triangle QP
            Gs = np.tril(G.astype(np.float64)); Gs = Gs + np.tril(Gs, -1).T
            s = CI.T.astype(np.float64) @ x - b
            assert s.min() >= -1e-6
        assert np.all(np.isfinite(force)) and np.all(np.isfinite(tau))
    assert 0 < n_inf < 150


def test_vmc_golden(pkg, oracle):
    g = np.load(GOLD)
    cfg = g["cfg"]; geom = g["geom"]
    for i in range(g["vin"].shape[0]):
        force, tau, x, st, rc = oracle.vmc_solve(cfg, geom, g["vin"][i], g["q"][i])
        assert np.abs(x - g["x_quadprog"][i]).max() <= 1e-8 * max(1.0, np.abs(g["x_quadprog"][i]).max())
        assert (rc == 1) == bool(g["quadprog_inf"][i])
        Gq, aq, CIq, bq = oracle.vmc_assemble(cfg, g["vin"][i])
        assert np.array_equal(Gq, g["G"][i]) and np.array_equal(aq, g["a"][i])       # the fp32 assembly has not drifted


WGOLD = os.path.join(os.path.dirname(__file__), "golden", "vmc_world_golden.npz")


def test_vmc_world_frame_overload(pkg, oracle, ref):
    """World-frame overload (qr_qp_torque_optimizer.cpp:304-398): the same QP with Rcb = rotMat, world gravity / axes and per-leg
    force-window ratios.  Assembly properties (the double 9.8 of :133-134, rotation invariance of the physics) and x against QuadProg++."""
    W = pkg.workload
    cfg = W.vmc_cfg("a1"); geom = pkg.model_desc("a1")[:3]
    vin, q, ratio = W.make_vmc_world_batch(120, seed=17)
    n_inf = n_walk = 0
    for i in range(120):
        G, a, CI, b = oracle.vmc_assemble(cfg, vin[i], ratio[i])
        for l in range(4):
            if vin[i, 18 + l] == 0:
                assert b[2 * l] == np.float32(1e-7) and b[2 * l + 1] == np.float32(1e-7)
            else:
                assert b[2 * l] == np.float32(np.float64(np.float32(ratio[i, l] * np.float32(13.0))) * 9.8)
                assert b[2 * l + 1] == np.float32(np.float64(np.float32(-ratio[i, 4 + l] * np.float32(13.0))) * 9.8)
        n_walk += ratio[i, 0] != np.float32(0.01)
        # constraint normals are the world axes: rows 2l = e_z, friction rows mu e_z +- e_x / e_y
        assert np.array_equal(CI[0:3, 0], [0, 0, 1]) and np.array_equal(CI[0:3, 8], np.float32([1, 0, 0.5])) and np.array_equal(CI[0:3, 10], np.float32([0, 1, 0.5]))
        force, tau, x, st, rc = oracle.vmc_solve(cfg, geom, vin[i], q[i], ratio[i])
        xr, fr = oracle.ref_quadprog(*_ref_inputs(G, a, CI, b))
        assert np.abs(x - xr).max() <= 1e-8 * max(1.0, np.abs(xr).max()), i
        assert (rc == 1) == (not np.isfinite(fr))
        n_inf += rc == 1
        # forces come back in the base frame: R^T (world solution)
        R = vin[i, 22:31].reshape(3, 3).astype(np.float64)
        assert np.abs(force.reshape(4, 3) - (-x.reshape(4, 3)) @ R).max() <= 1e-5 * max(1.0, np.abs(x).max())
    assert 0 < n_inf < 120 and 0 < n_walk < 120
    # a level robot with the trot ratios is the control-frame overload's problem up to the rounding of the force window
    lvl = vin[0].copy(); lvl[22:31] = np.eye(3, dtype=np.float32).reshape(-1)
    trot = np.float32([0.01] * 4 + [10.0] * 4)
    Gw, aw, CIw, bw = oracle.vmc_assemble(cfg, lvl, trot)
    Gc, ac, CIc, bc = oracle.vmc_assemble(cfg, lvl)
    assert np.array_equal(Gw, Gc) and np.array_equal(aw, ac) and np.array_equal(CIw, CIc) and np.abs(bw - bc).max() <= 1e-4


def test_vmc_world_golden(pkg, oracle):
    g = np.load(WGOLD)
    for i in range(g["vin"].shape[0]):
        force, tau, x, st, rc = oracle.vmc_solve(g["cfg"], g["geom"], g["vin"][i], g["q"][i], g["ratio"][i])
        assert np.abs(x - g["x_quadprog"][i]).max() <= 1e-8 * max(1.0, np.abs(g["x_quadprog"][i]).max())
        assert (rc == 1) == bool(g["quadprog_inf"][i])
        Gq, aq, CIq, bq = oracle.vmc_assemble(g["cfg"], g["vin"][i], g["ratio"][i])
        assert np.array_equal(Gq, g["G"][i]) and np.array_equal(aq, g["a"][i]) and np.array_equal(bq, g["b"][i])
