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


def _rot(quat):
    w, x, y, z = [float(v) for v in quat]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_foothold_heuristic_properties(pkg, oracle):
    """qrRaibertSwingLegController::Update (swing-leg selection) + qrFootholdPlanner::ComputeHeuristicFootHold restated
    (qr_swing_leg_controller.cpp:211-236, qr_foothold_planner.cpp:110-239): pinned by what the formulas imply (Eigen cannot be compiled here)."""
    W = pkg.workload
    desc = W.foothold_cfg("a1")
    ho, hp, hl, kp, clr = desc[:12].reshape(4, 3), desc[12:24].reshape(4, 3), desc[24], desc[25:28], desc[28]
    x = W.make_foothold_batch(400, seed=3)
    side = np.array([-1, 1, -1, 1])
    seen = dict(stance=0, early=0, hold=0, clip=0, free=0)
    for i in range(x.shape[0]):
        prev = np.full(58, -777.0, np.float32)
        o = oracle.footholds(desc, x[i], prev)
        R = _rot(x[i, 33:37])
        for leg in range(4):
            st, allow = int(x[i, leg]), x[i, 4 + leg] != 0
            skip = (st == 1 and allow) or st == 2
            assert o[leg] == (0.0 if skip else 1.0)
            if skip:
                assert o[4 + leg] == -777.0 and np.all(o[24 + 3 * leg:27 + 3 * leg] == -777.0)      # the planner's members keep their values
                seen["early" if st == 2 else "stance"] += 1
                continue
            f = o[24 + 3 * leg:27 + 3 * leg].astype(np.float64)
            if not allow:
                # the foot stays where it is relative to the default hip, 2 cm lower and 5 mm towards the body axis, in the world-aligned frame
                t = R @ (x[i, 21 + 3 * leg:24 + 3 * leg] - hp[leg])
                t[1] += -0.005 if t[1] > 0.01 else (0.005 if t[1] < -0.01 else 0.0)
                t[2] -= 0.02
                assert np.abs(f - (R.T @ t + hp[leg])).max() < 2e-6 and o[4 + leg] == 1.0
                seen["hold"] += 1
                continue
            assert o[4 + leg] == x[i, 12 + leg]                                                  # phase = normalizedPhase
            hv = R @ (x[i, 40:43] + np.cross(x[i, 43:46], ho[leg])); hv[2] = 0
            tv = x[i, 16:19] + x[i, 19] * np.array([-ho[leg, 1], ho[leg, 0], 0])
            dP = R.T @ (tv * x[i, 8 + leg] - kp * (tv - hv))
            clipped = np.abs(dP[:2]).max() > 0.2
            dP = np.array([np.clip(dP[0], -0.2, 0.2), np.clip(dP[1], -0.2, 0.2), 0.0])
            c, s = np.cos(x[i, 37]), np.sin(x[i, 37])
            expect = dP + np.array([ho[leg, 0], ho[leg, 1], 0]) + np.array([0, c * hl * side[leg], -s * hl * side[leg]]) - R.T @ np.array([0, 0, x[i, 20] - clr])
            assert np.abs(f - expect).max() < 3e-6, (i, leg, f, expect)
            seen["clip" if clipped else "free"] += 1
    assert all(v > 0 for v in seen.values()), seen
    # a robot standing still, level, with zero command: the foothold is under the hip joint, body height minus clearance below it
    z = np.zeros(46, np.float32); z[0:4] = 0; z[4:8] = 1; z[20] = 0.28; z[33] = 1.0
    o = oracle.footholds(desc, z)
    for leg in range(4):
        assert np.allclose(o[24 + 3 * leg:27 + 3 * leg], [ho[leg, 0], ho[leg, 1] + hl * side[leg], -(0.28 - clr)], atol=1e-7)


def test_swing_velocity_mode_properties(pkg, oracle):
    """Velocity-mode swing action (qr_swing_leg_controller.cpp:285-309): the Raibert target, the warped phase, the parabola, the IK."""
    W = pkg.workload
    cfg = W.estimator_cfg("a1"); geom, ho = cfg[:3], cfg[7:19]
    vd = W.swing_velocity_cfg("a1")
    x = W.make_swing_velocity_batch(160, seed=3)
    for i in range(160):
        o = oracle.swing_velocity(geom, ho, vd, x[i])
        dR = x[i, 28:37].reshape(3, 3).astype(np.float64)
        w_, a, b, c = x[i, 37:41].astype(np.float64)
        Rbw = np.array([[1 - 2 * (b * b + c * c), 2 * (a * b - w_ * c), 2 * (a * c + w_ * b)],
                        [2 * (a * b + w_ * c), 1 - 2 * (a * a + c * c), 2 * (b * c - w_ * a)],
                        [2 * (a * c - w_ * b), 2 * (b * c + w_ * a), 1 - 2 * (a * a + b * b)]])        # baseRMat
        for leg in range(4):
            sl = slice(3 * leg, 3 * leg + 3)
            if x[i, leg] == 0:
                assert np.isnan(o[0:12][sl]).all() and np.isnan(o[24:36][sl]).all()
                continue
            hp = vd[0:12][sl].astype(np.float64)
            tw = np.array([-hp[1], hp[0], 0.0])
            hh = dR @ (x[i, 20:23].astype(np.float64) + float(x[i, 23]) * tw); hh[2] = 0
            tv = x[i, 24:27].astype(np.float64) + float(x[i, 27]) * tw
            tgt = dR.T @ (hh * float(vd[12 + leg]) / 2 - vd[16:19].astype(np.float64) * (tv - hh)) + np.array([hp[0], hp[1], 0]) - Rbw.T @ np.array([0, 0, float(vd[19])])
            np.testing.assert_allclose(o[0:12][sl], tgt, atol=3e-6)
            p = float(x[i, 4 + leg])
            ph = 0.8 * np.sin(p * np.pi) if p <= 0.5 else 0.8 + (p - 0.5) * 0.4
            st = x[i, 8:20][sl].astype(np.float64)
            pb = o[12:24][sl].astype(np.float64)
            np.testing.assert_allclose(pb[:2], (1 - ph) * st[:2] + ph * tgt[:2], atol=3e-6)
            mid = max(st[2], tgt[2]) + 0.1
            np.testing.assert_allclose(pb[2], np.polyval(np.polyfit([0, 0.5, 1], [st[2], mid, tgt[2]], 2), ph), atol=3e-6)
            if p == 0.0:
                np.testing.assert_allclose(pb, st, atol=1e-6)                                  # lift-off point at the start of the swing
            if p == 1.0:
                np.testing.assert_allclose(pb, tgt, atol=3e-6)                                 # the warp ends at 0.8 + 0.5 * 0.4 = 1: on the target
            ang = o[24:36][sl]
            if i == 159 and leg == 0:
                assert np.array_equal(ang[1:], x[i, 41:53][sl][1:])                           # unreachable: NaN angles replaced by the current ones
                continue
            qfull = x[i, 41:53].copy(); qfull[sl] = ang
            fk = oracle.foot_positions(geom, ho, qfull).reshape(4, 3)[leg]
            np.testing.assert_allclose(fk, pb, atol=5e-6)
            assert np.all(o[36:48][sl] == 0)
    # a standing robot with no command: the target lies under the hip, desiredHeight - footClearance below the base
    y = np.zeros(53, np.float32); y[0:4] = 1; y[4:8] = 1.0; y[28] = y[32] = y[36] = 1; y[37] = 1; y[41:53] = np.tile([0.0, 0.9, -1.8], 4)
    y[8:20] = (ho.reshape(4, 3) + np.array([0, 0, -0.26], np.float32)).reshape(12)
    o = oracle.swing_velocity(geom, ho, vd, y)
    np.testing.assert_allclose(o[0:12].reshape(4, 3), np.column_stack([vd[0:12].reshape(4, 3)[:, :2], np.full(4, -0.26)]), atol=1e-6)
