"""CPU: properties of the oracle's open-loop gait generator (qr_openloop_gait_generator.cpp:126-249, advanced_trot)."""
import numpy as np


def test_nominal_trot(pkg, oracle):
    W = pkg.workload
    cfg = W.gait_cfg()
    T, dt = 1500, 0.002
    full = 0.5 / 0.6
    t = (np.arange(T) * dt).astype(np.float32)
    # every foot always reports contact: the generator never holds, and the phases are the closed form -- up to one tick of slip per
    # cycle, because the clock restart of Schedule (:212-216) fires on the first tick strictly after a full period
    ph = np.stack([np.fmod(cfg[8 + l] * full + t.astype(np.float64), full) / full for l in range(4)], 1)
    o = oracle.gait_run(cfg, t, np.ones((T, 4), np.float32))
    err = np.abs(o[:, 0:4] - ph); err = np.minimum(err, 1 - err)
    assert err.max() < 4 * dt / full + 1e-4
    des = o[:, 8:12]
    assert np.array_equal(des == 1, o[:, 0:4] < 0.6)                         # STANCE while the phase is below the duty factor
    assert np.all((o[:, 4:8] >= 0) & (o[:, 4:8] <= 1))
    # diagonal pairs move together, the two pairs half a cycle apart
    assert np.array_equal(des[:, 0], des[:, 3]) and np.array_equal(des[:, 1], des[:, 2])
    sw = des[:, 0] == 0
    assert 0.35 < sw.mean() < 0.45                                           # 40 % of the cycle in swing
    rem = o[sw, 20]
    assert rem.max() <= full - 0.5 + 1e-6 and rem.min() >= 0


def test_lost_contact_holds_the_schedule(pkg, oracle):
    """A foot that should have landed but has not: the phase clock stops (resetTime advances) until it lands or wait_time passes."""
    W = pkg.workload
    cfg = W.gait_cfg(wait_time=0.05)
    T, dt = 1200, 0.002
    full = 0.5 / 0.6
    t = (np.arange(T) * dt).astype(np.float32)
    contact = np.ones((T, 4), np.float32)
    nominal = oracle.gait_run(cfg, t, contact)
    k0 = int(np.argmax((nominal[1:, 8] == 1) & (nominal[:-1, 8] == 0))) + 1      # first swing -> stance switch of leg 0
    c2 = contact.copy(); c2[k0 - 2:k0 + 200, 0] = 0                                  # leg 0 stays in the air
    held = oracle.gait_run(cfg, t, c2)
    # during the hold the phases of all legs freeze
    frozen = held[k0 + 3:k0 + 20, 0:4]
    assert np.abs(frozen - frozen[0]).max() < 1e-6
    assert not np.allclose(held[k0 + 60, 0:4], frozen[0], atol=1e-3)                # released after wait_time (25 ticks) and running again
