"""CPU: the oracle's dense Goldfarb-Idnani solver, pinned against the reference's vendored solvers.

  * known answer of QX/QuadProgpp/src/main.cc:8-20  (x = [1, 2], f = 12)
  * QuadProg++ solutions of WBC-shaped QPs            (tests/golden/qp_golden.npz, and live via oracle/_ref)
  * qpOASES 3.2.0 solutions of the MPC QP             (tests/golden/mpc_golden.npz, and live via oracle/_ref)
"""
import numpy as np
import pytest

import golden_io


def test_known_answer_quadprogpp_demo(oracle):
    G = np.array([[4, -2], [-2, 4.0]]); g0 = np.array([6.0, 0]); CE = np.array([[1.0], [1.0]]); ce0 = np.array([-3.0])
    CI = np.array([[1, 0, 1], [0, 1, 1.0]]); ci0 = np.array([0, 0, -2.0])
    x, lam, st, rc = oracle.qp_solve(G, g0, CE, ce0, CI, ci0)
    assert rc == 0 and np.allclose(x, [1.0, 2.0], atol=1e-12) and abs(st["obj"] - 12.0) < 1e-10


def test_golden_quadprogpp(oracle):
    rows = golden_io.load("qp_golden.npz")
    assert len(rows) >= 9
    for r in rows:
        x, lam, st, rc = oracle.qp_solve(r["G"], r["g0"], r["CE"], r["ce0"], r["CI"], r["ci0"])
        assert rc == 0
        assert np.abs(x - r["x_quadprogpp"]).max() <= 1e-8 * max(1.0, np.abs(x).max())
        assert abs(st["obj"] - float(r["f_quadprogpp"][0])) <= 1e-8 * max(1.0, abs(st["obj"]))


def test_live_quadprogpp_random(ref):
    """Random strictly convex QPs with equalities and inequalities vs the compiled reference QuadProg++."""
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(3, 19)); p = int(rng.integers(0, min(n, 7))); m = int(rng.integers(1, 25))
        A = rng.normal(size=(n, n)); G = A @ A.T + np.eye(n)
        g0 = rng.normal(size=n) * 3
        CE = rng.normal(size=(n, p)); x_feas = rng.normal(size=n); ce0 = -(CE.T @ x_feas)
        CI = rng.normal(size=(n, m)); ci0 = -(CI.T @ x_feas) + rng.uniform(0.0, 2.0, m)      # x_feas is strictly feasible
        x, lam, st, rc = ref.qp_solve(G, g0, CE, ce0, CI, ci0)
        xr, fr = ref.ref_quadprog(G, g0, CE, ce0, CI, ci0)
        assert rc == 0 and np.isfinite(fr)
        assert np.abs(x - xr).max() <= 1e-7 * max(1.0, np.abs(xr).max()), (trial, np.abs(x - xr).max())
        # KKT: stationarity with non-negative inequality multipliers, complementarity
        assert np.all(lam >= -1e-9)
        s = CI.T @ x + ci0
        assert np.all(s >= -1e-8) and np.abs(lam * s).max() <= 1e-6


def test_degenerate_pyramid_corner(oracle):
    """f = 0 is optimal: five pyramid rows active on three variables (linearly dependent working set)."""
    im = 1 / 0.45
    CI = np.array([[im, -im, 0, 0, 0, 0], [0, 0, im, -im, 0, 0], [1, 1, 1, 1, 1, -1.0]])
    ci0 = np.array([0, 0, 0, 0, 0, 100.0])
    G = np.diag([1.0, 2.0, 3.0]); g0 = np.array([0.3, -0.2, 5.0])          # pushes f_z negative
    x, lam, st, rc = oracle.qp_solve(G, g0, None, None, CI, ci0)
    assert rc == 0 and np.abs(x).max() <= 1e-12
