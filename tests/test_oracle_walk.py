"""CPU: properties of the oracle's walk gait generator (qrWalkGaitGenerator::Update, qr_walk_gait_generator.cpp:202-288) and of the
force-window ratios its sub-states select (TorqueStanceLegController::UpdateFRatio, walk branch, qr_torque_stance_leg_controller.cpp:125-168).
The class needs Eigen and yaml-cpp (not in this image): pinned by the closed form of the schedule, not by a compiled reference."""
import numpy as np

STANCE, LOAD, UNLOAD, FULL, TRUE_SWING = 1, 5, 6, 7, 8


def test_schedule_closed_form(pkg, oracle):
    cfg = pkg.workload.walk_cfg()
    T, dt = 6000, 0.002                                   # 12 s: more than one 10 s cycle
    t = (np.arange(T) * dt).astype(np.float32)
    o = oracle.walk_run(cfg, t, np.ones((T, 4), np.float32))
    c = np.ones((T, 4), np.float32); c[o[:, 8:12] == TRUE_SWING] = 0          # feet lift off when their leg is in true swing (the schedule itself
    o = oracle.walk_run(cfg, t, c)                                           #  does not depend on the contacts)
    full = 7.5 / 0.75
    init = cfg[8:12].astype(np.float64)
    for l in range(4):
        ph = np.fmod(init[l] * full + t.astype(np.float64), full) / full
        assert np.abs(o[:, l] - ph).max() < 2e-6
        psc = (ph - 0.75) / 0.25
        want = np.where(ph <= 0.75, STANCE, np.where(psc < 0.2, FULL, np.where(psc < 0.5, UNLOAD, np.where(psc < 0.8, TRUE_SWING, LOAD))))
        # a sub-state switches on the first tick AFTER its boundary (the index advances one tick late, :252-258), and float rounding at
        # the boundaries: compare away from them
        edge = np.zeros(T, bool)
        for b in (0.75, 0.75 + 0.25 * 0.2, 0.75 + 0.25 * 0.5, 0.75 + 0.25 * 0.8, 1.0, 0.0):
            edge |= np.abs(ph - b) < 2 * dt / full + 1e-6
        assert np.array_equal(o[~edge, 8 + l], want[~edge].astype(np.float32)), l
        assert np.all((o[:, 4 + l] >= -1e-6) & (o[:, 4 + l] <= 1 + 2 * dt / (full * 0.25 * 0.2)))
        # legState is only ever SWING / STANCE, curLegState trails desiredLegState by one tick
        assert set(np.unique(o[:, 12 + l])) <= {0.0, 1.0}
        assert np.array_equal(o[1:, 16 + l], o[:-1, 8 + l])
    # the legs take turns: never two legs outside STANCE
    assert ((o[:, 8:12] != STANCE).sum(1) <= 1).all()
    # contacts / force windows of the sub-states
    des, nph = o[:, 8:12], o[:, 4:8]
    cont, fmin, fmax = o[:, 29:33], o[:, 33:37], o[:, 37:41]
    assert np.all(fmin == np.float32(0.001))
    assert np.all(cont[des == TRUE_SWING] == 0) and np.all(fmax[des == TRUE_SWING] == np.float32(0.002))
    assert np.all(cont[des != TRUE_SWING] == 1)
    assert np.all(fmax[(des == STANCE) | (des == FULL)] == 10.0)
    m = des == LOAD
    assert np.allclose(fmax[m], 10 * np.maximum(0.001, nph[m]), rtol=1e-6)
    m = des == UNLOAD
    assert np.allclose(fmax[m], 10 * np.maximum(0.001, 1 - nph[m] / 0.75), rtol=1e-5, atol=1e-6)
    # moveBasePhase ramps over the stance-like head of the swing quarter (full_stance + unload_force = 0.5 of it), then holds 1
    l = 1
    ph = np.fmod(init[l] * full + t.astype(np.float64), full) / full
    k = np.nonzero((ph > 0.76) & (ph < 0.87))[0]
    assert np.allclose(o[k, 28], ((ph[k] - 0.75) / 0.25) / 0.5, atol=1e-4)
    k = np.nonzero((ph > 0.88) & (ph < 0.99))[0]
    assert np.all(o[k, 28] == 1.0)


def test_contact_events(pkg, oracle):
    cfg = pkg.workload.walk_cfg()
    T, dt = 5500, 0.002
    t = (np.arange(T) * dt).astype(np.float32)
    base = oracle.walk_run(cfg, t, np.ones((T, 4), np.float32))
    # with every foot always reporting contact, a leg in true swing past 10 % of it reads EARLY_CONTACT (plan swing, actual stance)
    ts = base[:, 8:12] == TRUE_SWING
    late = ts & (base[:, 4:8] >= 0.1)
    assert late.any() and np.all(base[:, 20:24][late] == 2) and np.all(base[:, 20:24][ts & ~late] == 0)
    early = base[:, 20:24] == 2
    assert np.allclose(base[:, 37:41][early], 10 * np.minimum(0.01, np.abs(base[:, 4:8][early] - 0.8)), rtol=1e-6)
    assert np.all(base[:, 29:33][early] == 1)
    # a foot in the air while its leg should stand (past 10 % of the stance): LOSE_CONTACT, still a contact for the force distribution
    c = np.ones((T, 4), np.float32)
    k = np.nonzero((base[:, 8] == STANCE) & (base[:, 4] > 0.3) & (base[:, 4] < 0.4))[0]
    c[k, 0] = 0
    o = oracle.walk_run(cfg, t, c)
    assert np.all(o[k, 20] == 3) and np.all(o[k, 29] == 1) and np.all(o[k, 37] == 10.0)
    assert np.array_equal(o[k, 24], o[k, 0])                       # detectedEventTickPhase = phaseInFullCycle of the event
    # a foot that duly lifts off in true swing: plain SWING
    c = np.ones((T, 4), np.float32); c[base[:, 8 + 2] == TRUE_SWING, 2] = 0
    o = oracle.walk_run(cfg, t, c)
    assert np.all(o[base[:, 10] == TRUE_SWING, 22] == 0)


def test_stop_freezes_stance_legs(pkg, oracle):
    """robot->stop: curLegState only follows desiredLegState for legs that are in SWING (:218-220)."""
    cfg = pkg.workload.walk_cfg()
    T, dt = 3000, 0.002
    t = (np.arange(T) * dt).astype(np.float32)
    stop = np.zeros(T, np.int32); stop[500:] = 1
    o = oracle.walk_run(cfg, t, np.ones((T, 4), np.float32), stop)
    free = oracle.walk_run(cfg, t, np.ones((T, 4), np.float32))
    assert np.array_equal(o[:500], free[:500])
    # desired states keep following the clock, the current state of a standing leg does not change any more
    assert np.array_equal(o[:, 8:12], free[:, 8:12])
    for l in range(4):
        k0 = 500
        if o[k0, 16 + l] == STANCE:
            assert np.all(o[k0:, 16 + l] == STANCE)
