"""CPU: properties of the oracle's ground-plane estimator (qrGroundSurfaceEstimator::Update / GetNormalVector / ComputeControlFrame,
qr_ground_surface_estimator.cpp:40-70,151-206).  The class needs Eigen and yaml-cpp (not in this image), so it is pinned by what its
formulas must satisfy, not by a compiled reference."""
import numpy as np

HIPS = [(0.18, -0.13), (0.18, 0.13), (-0.18, -0.13), (-0.18, 0.13)]


def quat_from_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    return np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy])


def make_in(T, plane=(-0.3, 0.0, 0.0), rpy=(0, 0, 0), contact=None, rng=None, jitter=0.0):
    x = np.zeros((T, 23), np.float32)
    x[:, 0:4] = 1 if contact is None else contact
    for l, (px, py) in enumerate(HIPS):
        fx = px + (rng.uniform(-jitter, jitter, T) if rng is not None else 0)
        fy = py + (rng.uniform(-jitter, jitter, T) if rng is not None else 0)
        x[:, 4 + 3 * l] = fx; x[:, 5 + 3 * l] = fy
        x[:, 6 + 3 * l] = plane[0] + plane[1] * x[:, 4 + 3 * l].astype(np.float64) + plane[2] * x[:, 5 + 3 * l].astype(np.float64)
    x[:, 16:19] = (0.1, -0.2, 0.3)
    x[:, 19:23] = quat_from_rpy(*rpy)
    return x


def test_plane_through_four_coplanar_feet(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        plane = (rng.uniform(-0.35, -0.2), rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3))
        x = make_in(2, plane, rng=rng, jitter=0.05)
        x[0, 0] = 0                                   # three feet down, then the fourth lands: the fit fires on tick 1
        o = oracle.ground_run(x)
        assert o[0, 31] == 0 and o[1, 31] == 1
        assert np.allclose(o[0, 0:3], 0) and np.allclose(o[0, 3:6], (0, 0, 1))          # Reset(): a = 0, n = (0, 0, 1)
        assert np.allclose(o[1, 0:3], plane, atol=2e-6)
        nrm = np.array([-plane[1], -plane[2], 1.0]); nrm /= np.linalg.norm(nrm)
        assert np.allclose(o[1, 3:6], nrm, atol=1e-6) and abs(np.linalg.norm(o[1, 3:6]) - 1) < 1e-6


def test_least_squares_when_the_feet_are_not_coplanar(oracle):
    rng = np.random.default_rng(5)
    x = make_in(2, (-0.3, 0.05, -0.02), rng=rng, jitter=0.04)
    x[:, 6 + 3 * 2] += 0.03                          # one foot 3 cm off the plane
    x[0, 1] = 0
    o = oracle.ground_run(x)
    W = np.stack([np.ones(4), x[1, 4:16:3], x[1, 5:16:3]], 1).astype(np.float64)
    a = np.linalg.lstsq(W, x[1, 6:16:3].astype(np.float64), rcond=None)[0]
    assert np.allclose(o[1, 0:3], a, atol=2e-6)


def test_update_gating(oracle):
    """The fit needs all four feet down and at least one of them newly so (:42-56)."""
    T = 12
    c = np.ones((T, 4), np.float32)
    c[0] = (1, 1, 1, 0); c[1] = (1, 1, 1, 1); c[2] = (1, 1, 1, 1)         # fires at 1 only
    c[3] = (0, 1, 1, 1); c[4] = (1, 1, 1, 0)                              # a new contact, but only three feet down: no fit
    c[5] = (1, 1, 1, 1)                                                   # leg 3 lands: fires
    c[6] = (0, 0, 1, 1); c[7] = (1, 1, 1, 1)                              # two land at once: fires
    x = make_in(T, contact=c)
    o = oracle.ground_run(x)
    assert list(np.nonzero(o[:, 31])[0]) == [1, 5, 7]


def test_control_frame_follows_the_heading(oracle):
    """Flat-world assumption (:168): the frame is the base's yaw, filtered with ratio 0.8, roll = 0 and (since the world normal is
    vertical) pitch = 0, whatever the base's roll and pitch."""
    T = 40
    c = np.ones((T, 4), np.float32); c[::2, 0] = 0                        # leg 0 re-lands every other tick -> an update every other tick
    yaw = 0.7
    x = make_in(T, plane=(-0.3, 0.1, 0.05), rpy=(0.2, -0.15, yaw), contact=c)
    o = oracle.ground_run(x)
    upd = np.nonzero(o[:, 31])[0]
    assert len(upd) == T // 2
    k = np.arange(1, len(upd) + 1)
    assert np.allclose(o[upd, 8], yaw * (1 - 0.2 ** k), atol=1e-6)        # first-order filter from 0
    assert np.allclose(o[:, 6], 0) and np.allclose(o[:, 7], 0, atol=1e-7)
    for t in (1, 9, T - 1):
        R = o[t, 13:22].reshape(3, 3).astype(np.float64)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
        y = o[t, 8]
        assert np.allclose(R, [[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]], atol=1e-6)       # groundRMat: frame -> world
        q = o[t, 9:13]
        assert np.allclose(q, [np.cos(y / 2), 0, 0, np.sin(y / 2)], atol=1e-6)
        # baseRInControlFrame = groundRMat' * baseRMat
        e0, e1, e2, e3 = x[t, 19:23].astype(np.float64)
        B = np.array([[1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2)],
                      [2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3), 2 * (e2 * e3 - e0 * e1)],
                      [2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)]])
        assert np.allclose(o[t, 22:31].reshape(3, 3), R.T @ B, atol=2e-6)
