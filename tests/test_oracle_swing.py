"""CPU: properties of the oracle's swing-leg targets (qr_swing_leg_controller.cpp:362-424, ADVANCED_TROT on horizontal terrain)."""
import numpy as np


def test_swing_targets_properties(pkg, oracle):
    W = pkg.workload
    cfg = W.estimator_cfg("a1"); geom, ho = cfg[:3], cfg[7:19]
    x = W.make_swing_batch(128, seed=2)
    for i in range(128):
        o = oracle.swing_targets(geom, ho, x[i])
        q4 = x[i, 39:43].astype(np.float64); w_, a, b, c = q4
        R = np.array([[1 - 2 * (b * b + c * c), 2 * (a * b - w_ * c), 2 * (a * c + w_ * b)],
                      [2 * (a * b + w_ * c), 1 - 2 * (a * a + c * c), 2 * (b * c - w_ * a)],
                      [2 * (a * c - w_ * b), 2 * (b * c + w_ * a), 1 - 2 * (a * a + b * b)]])
        bp = x[i, 36:39].astype(np.float64)
        for leg in range(4):
            sl = slice(3 * leg, 3 * leg + 3)
            if x[i, leg] == 0:
                assert np.isnan(o[0:12][sl]).all() and np.isnan(o[48:60][sl]).all()      # stance legs untouched
                continue
            ph = float(x[i, 4 + leg]); st = x[i, 12:24][sl].astype(np.float64); tg = x[i, 24:36][sl].astype(np.float64)
            np.testing.assert_allclose(o[36:48][sl], R @ tg + bp, atol=2e-6)               # foothold in the world frame
            np.testing.assert_allclose(o[12:24][sl], x[i, 43:46], atol=0)                  # vFoot_des = base velocity (the spline has xd = 0)
            assert np.all(o[24:36][sl] == 0)
            if ph > 1.001:
                continue
            pb = R.T @ (o[0:12][sl].astype(np.float64) - bp)                               # back to the base frame
            np.testing.assert_allclose(pb[:2], (1 - ph) * st[:2] + ph * tg[:2], atol=3e-6)
            mid = max(st[2], tg[2]) + 0.1                                                  # parabola through start, apex at half phase, end
            z_expect = np.polyval(np.polyfit([0, 0.5, 1], [st[2], mid, tg[2]], 2), ph)
            np.testing.assert_allclose(pb[2], z_expect, atol=3e-6)
            # inverse kinematics: forward kinematics of the joint targets gives the point back
            ang = o[48:60][sl]
            if i == 127 and leg == 0:
                assert np.array_equal(ang[1:], x[i, 46:58][sl][1:])                       # unreachable: NaN angles replaced by the current ones
                continue
            qfull = x[i, 46:58].copy(); qfull[sl] = ang
            fk = oracle.foot_positions(geom, ho, qfull).reshape(4, 3)[leg]
            np.testing.assert_allclose(fk, pb, atol=5e-6)
            assert np.all(o[60:72][sl] == 0)
