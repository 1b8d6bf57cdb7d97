"""CPU: properties of the oracle's MPC front-end (qr_mpc_stance_leg_controller.cpp:158-382 restated in oracle/qr_oracle_frontend.cpp)."""
import numpy as np


def test_frontend_properties(pkg, oracle):
    wl = pkg.workload
    h, L = 10, 2
    fe, st = wl.make_frontend_batch(64, seed=3)
    for i in range(64):
        o = oracle.mpc_frontend(h, L, fe[i], st[i])
        it = int(st[i, 7])
        # cadence (:342): every 15th tick and each of the first 50
        assert o["updated"] == int(it % 15 == 0 or it < 50)
        assert o["state"][7] == it + 1
        # filter + clip (:175-179)
        assert -1.0 <= o["state"][0] <= 2.0 and -0.6 <= o["state"][1] <= 0.6
        g = o["gait"].reshape(h, 4)
        assert np.array_equal(g[0], (fe[i, 38:42] != 0).astype(np.float32))          # row 0 = measured contacts (:301-303)
        assert set(np.unique(g)) <= {0.0, 1.0}
        for j in range(4):
            if int(fe[i, 58 + j]) == 2:
                assert g[1:, j].all()                                                # EARLY_CONTACT keeps the leg in the table (:293)
        if o["updated"]:
            tr = o["traj"].reshape(h, 12)
            assert abs(tr[0, 3] - fe[i, 6]) <= 0.1 + 1e-6 and abs(tr[0, 4] - fe[i, 7]) <= 0.1 + 1e-6      # clip to p +- 0.1 (:352-353)
            np.testing.assert_allclose(np.diff(tr[:, 2]), 0.06 * o["state"][2], atol=2e-6)              # yaw ramp (:372)
            np.testing.assert_allclose(np.diff(tr[:, 3]), 0.06 * o["wbc15"][3], atol=2e-6)
            assert np.all(tr[:, 5] == o["wbc15"][2]) and np.all(tr[:, 8] == o["state"][2])
            assert np.all(tr[:, [0, 6, 7, 11]] == 0)
        else:
            assert np.isnan(o["traj"]).all()                                         # trajAll untouched between re-plans
        # height compensation only with a swing leg (:233-241), at most 0.02
        dh = o["wbc15"][2] - fe[i, 0]
        assert -1e-7 <= dh <= 0.02 + 1e-6
        if not (fe[i, 54:58] == 0).any():
            assert dh == 0
        # vBody_des is the rotated filtered command: its norm is that of (xVelDes, yVelDes) up to the roll/pitch tilt
        assert np.hypot(o["wbc15"][3], o["wbc15"][4]) <= np.hypot(o["state"][0], o["state"][1]) * (1 + 1e-5) + 1e-6


def test_frontend_yaw_wrap(pkg, oracle):
    fe, st = pkg.workload.make_frontend_batch(4, seed=5)
    fe[:, 5] = 1.0; st[:, 2] = 1.0
    fe[0, 9] = 0.0; st[0, 3] = np.float32(np.pi) - 1e-4                # crosses +pi: wraps by M_2PI (:187-188)
    o = oracle.mpc_frontend(10, 2, fe[0], st[0])
    assert o["state"][3] < -3.0
    fe[1, 9] = 3.0; st[1, 3] = -3.1                                    # yaw > pi/2 and plan < 0: unwrapped upwards (:196-197)
    o = oracle.mpc_frontend(10, 2, fe[1], st[1])
    assert o["state"][3] > 3.0
    fe[2, 9] = -3.0; st[2, 3] = 3.1
    o = oracle.mpc_frontend(10, 2, fe[2], st[2])
    assert o["state"][3] < -3.0


def test_frontend_sequence_matches_closed_form(pkg, oracle):
    """Constant command, stance-only gait: the filters are first-order low-passes with known closed form."""
    fe, st = pkg.workload.make_frontend_batch(1, seed=9)
    f, s = fe[0].copy(), np.zeros(8, np.float32)
    f[3:6] = (0.5, 0.1, 0.2); f[9] = 0.0; f[10:14] = (1, 0, 0, 0); f[54:62] = 1; f[38:42] = 1
    K = 200
    for _ in range(K):
        s = oracle.mpc_frontend(10, 2, f, s)["state"]
    np.testing.assert_allclose(s[0], 0.5 * (1 - 0.99 ** K), rtol=1e-4)
    np.testing.assert_allclose(s[1], 0.1 * (1 - 0.995 ** K), rtol=1e-4)
    np.testing.assert_allclose(s[2], 0.2 * (1 - 0.97 ** K), rtol=1e-4)
    yaw = sum(0.002 * 0.2 * (1 - 0.97 ** (k + 1)) for k in range(K))
    np.testing.assert_allclose(s[3], yaw, rtol=1e-3)
    assert s[7] == K
