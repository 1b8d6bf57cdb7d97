"""GPU parity of the walk gait kernel (qrgpu_walk_gait_update_batch) against the oracle over tick sequences, and its hand-over to the
world-frame force distribution.  Reference: qrWalkGaitGenerator::Update (qr_walk_gait_generator.cpp:202-288), UpdateFRatio's walk branch
(qr_torque_stance_leg_controller.cpp:125-168).  Bar: bit-exact (plain float arithmetic with the same float / double mix)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_contacts(n, T, seed):
    rng = np.random.default_rng(seed)
    c = np.ones((T, n, 4), np.float32)
    # random stretches without contact (some fall into stance phases -> LOSE_CONTACT, some into true swing -> no EARLY_CONTACT there)
    for _ in range(6):
        k0 = rng.integers(0, T - 80, (n, 4)); ln = rng.integers(5, 80, (n, 4))
        for r in range(n):
            for l in range(4):
                c[k0[r, l]:k0[r, l] + ln[r, l], r, l] = 0
    return c


@pytest.mark.parametrize("variant", ["yaml", "short_cycle_swing_start"])
def test_walk_sequence_bit_exact(gpu_ctx, pkg, oracle, variant):
    W = pkg.workload
    n = 200
    if variant == "yaml":
        cfg, T, dt, t0 = W.walk_cfg(), 1500, 0.008, 3.0                 # 12 s of a 10 s cycle, starting away from zero (Update takes absolute time)
    else:                                                              # a leg that starts in its swing quarter, a dropped sub-state, another order
        cfg, T, dt, t0 = W.walk_cfg(stance_duration=1.5, duty_factor=0.75, initial_leg_phase=(0.9, 0.1, 0.6, 0.35), initial_leg_state=(0, 1, 1, 1),
                                    state_switch=(7, 6, 8, 5), state_ratio=(0.005, 0.4, 0.4, 0.2)), 1500, 0.002, 0.0
    t = (t0 + np.arange(T) * dt).astype(np.float32)
    contacts = make_contacts(n, T, seed=5)
    stop = np.zeros(T, np.int32); stop[900:1100] = 1
    d_state = gpu_ctx.alloc((33, n)).upload(np.full((33, n), np.nan, np.float32))       # reset = 2 must not depend on what was there
    d_c = gpu_ctx.alloc((4, n)); d_out = gpu_ctx.alloc((41, n)); d_ratio = gpu_ctx.alloc((8, n))
    d_vmc = gpu_ctx.alloc((37, n)).upload(np.full((37, n), -7.0, np.float32))
    snaps = {}
    for k in range(T):
        d_c.upload(pkg.to_soa(contacts[k]))
        gpu_ctx.walk_gait_update_batch(n, cfg, float(t[k]), d_c, d_state, d_out, d_ratio, d_vmc, stop=bool(stop[k]), reset=2 if k == 0 else 0)
        if k % 37 == 36 or k < 3 or k == T - 1:
            snaps[k] = d_out.download().T.copy()
    ratio = d_ratio.download().T; vmc = d_vmc.download().T
    seen = set()
    for r in range(0, n, 4):
        o = oracle.walk_run(cfg, t, contacts[:, r], stop)
        for k, g in snaps.items():
            assert np.array_equal(g[r], o[k]), (r, k, g[r], o[k])
        assert np.array_equal(ratio[r], o[-1, 33:41]) and np.array_equal(vmc[r, 18:22], o[-1, 29:33])
        assert np.all(vmc[r, :18] == -7.0) and np.all(vmc[r, 22:] == -7.0)
        seen |= set(np.unique(o[:, 8:12]).astype(int)) | {100 + int(v) for v in np.unique(o[:, 20:24])}
    want = {1, 5, 6, 8, 100, 101, 102, 103} | ({7} if variant == "yaml" else set())
    assert want <= seen, seen                                          # every sub-state and every detected state occurred
    for v in (d_state, d_c, d_out, d_ratio, d_vmc):
        v.free()


def test_walk_bad_arguments(gpu_ctx, pkg):
    W = pkg.workload
    n = 8
    d_c = gpu_ctx.alloc((4, n)); d_state = gpu_ctx.alloc((33, n))
    with pytest.raises(pkg.QrgpuError):                                # ratios that do not add up to one (the reference asserts, :124)
        gpu_ctx.walk_gait_update_batch(n, W.walk_cfg(state_ratio=(0.2, 0.3, 0.3, 0.3)), 0.0, d_c, d_state, reset=2)
    with pytest.raises(pkg.QrgpuError):                                # USERDEFINED_SWING legs are not built
        gpu_ctx.walk_gait_update_batch(n, W.walk_cfg(duty_factor=0.0), 0.0, d_c, d_state, reset=2)
    d_c.free(); d_state.free()
