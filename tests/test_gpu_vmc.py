"""GPU parity of the force-balance (VMC) stance QP (qrgpu_vmc_force_batch) against the oracle and the QuadProg++ golden vectors.
Reference: ComputeContactForce, qr_qp_torque_optimizer.cpp:190-301; MapContactForceToJointTorques, qr_robot.cpp:241-251.

Bar: forces within 1e-5 * max(1, |f|max) of the oracle (which equals the reference's QuadProg++ to 1e-8 on the same fp32 data, CPU test),
torques within 1e-4 * max(1, |tau|) (north_star tolerance); the 'QuadProg++ returned +inf' flag must match case by case."""
import os

import numpy as np

import gpu_helpers as G
import pytest

from gpu_helpers import tau_tol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "vmc_golden.npz")


def _run(ctx, pkg, vin, q, type_id=None):
    n = vin.shape[0]
    S = pkg.to_soa
    d_in = ctx.alloc((37, n)).upload(S(vin)); d_q = ctx.alloc((12, n)).upload(S(q))
    d_f = ctx.alloc((12, n)); d_t = ctx.alloc((12, n)); d_s = ctx.alloc((n,), np.int32)
    tid = ctx.alloc((n,), np.int32).upload(type_id) if type_id is not None else None
    ctx.vmc_force_batch(n, d_in, d_q, d_f, d_t, d_s, tid)
    ctx.sync()
    out = dict(force=d_f.download().T.copy(), tau=d_t.download().T.copy(), status=d_s.download())
    for v in (d_in, d_q, d_f, d_t, d_s):
        v.free()
    if tid is not None:
        tid.free()
    return out


@pytest.mark.parametrize("n,sloped,excite", [(1000, 0.0, 1.0), (333, 0.5, 1.0), (64, 0.2, 3.0)])
def test_vmc_parity(gpu_ctx, pkg, oracle, n, sloped, excite):
    W = pkg.workload
    cfg = W.vmc_cfg("a1"); geom = pkg.model_desc("a1")[:3]
    gpu_ctx.vmc_setup_packed(0, cfg, geom)
    vin, q = W.make_vmc_batch(n, sloped=sloped, excite=excite, seed=n)
    g = _run(gpu_ctx, pkg, vin, q)
    flags = G.flags(g["status"])
    assert np.all((flags & ~0x80) == 0), np.unique(flags)
    n_inf = 0
    for i in range(n):
        force, tau, x, st, rc = oracle.vmc_solve(cfg, geom, vin[i], q[i])
        assert bool(flags[i] & 0x80) == (rc == 1), (i, flags[i], rc)
        n_inf += rc == 1
        assert np.abs(g["force"][i] - force).max() <= 1e-5 * max(1.0, np.abs(force).max()), (i, np.abs(g["force"][i] - force).max())
        assert np.all(np.abs(g["tau"][i] - tau) <= tau_tol(tau, 1e-4)), i
    assert n_inf > 0


def test_vmc_golden_quadprog(gpu_ctx, pkg):
    """Forces against the reference's QuadProg++ output stored in tests/golden/vmc_golden.npz (no oracle in the loop)."""
    g = np.load(GOLD)
    gpu_ctx.vmc_setup_packed(0, g["cfg"], g["geom"])
    out = _run(gpu_ctx, pkg, g["vin"], g["q"])
    for i in range(g["vin"].shape[0]):
        R = g["vin"][i, 22:31].reshape(3, 3).astype(np.float64)
        X = -g["x_quadprog"][i].reshape(4, 3)
        f_ref = (X @ R).reshape(-1)                                  # (X * Rcb)^T flattened column-major = force[3*leg+axis]
        assert np.abs(out["force"][i] - f_ref).max() <= 1e-5 * max(1.0, np.abs(f_ref).max()), i
        assert bool(out["status"][i] & 0x80) == bool(g["quadprog_inf"][i])


def test_vmc_edge_cases(gpu_ctx, pkg, oracle):
    W = pkg.workload
    cfg = W.vmc_cfg("a1"); geom = pkg.model_desc("a1")[:3]
    gpu_ctx.vmc_setup_packed(0, cfg, geom)
    vin, q = W.make_vmc_batch(8, seed=1)
    vin[0, 18:22] = 0                       # no foot in contact: four contradictory pairs
    vin[1, 18:22] = 1; vin[1, 12:18] = 0    # standing still: f_z = m g / 4 each, nothing binds but f >= fmin
    vin[2, 12:18] = (40, 0, 0, 0, 0, 0)     # hard forward acceleration: friction rows bind
    vin[3, 12:18] = (0, 0, 200, 0, 0, 0)    # upward: fmax binds
    vin[4, 12:18] = (0, 0, -30, 0, 0, 0)    # downward beyond gravity: fmin binds
    g = _run(gpu_ctx, pkg, vin, q)
    for i in range(8):
        force, tau, x, st, rc = oracle.vmc_solve(cfg, geom, vin[i], q[i])
        assert np.abs(g["force"][i] - force).max() <= 1e-5 * max(1.0, np.abs(force).max()), (i, g["force"][i], force)
        assert bool(g["status"][i] & 0x80) == (rc == 1)
    f1 = g["force"][1].reshape(4, 3)
    assert abs(-f1[:, 2].sum() - 13 * 9.8) < 0.05 * 13 * 9.8          # the feet push down with the robot's weight (leg force = -GRF)
    f3 = g["force"][3].reshape(4, 3)
    assert np.all(-f3[:, 2] <= 10 * 13 * 9.8 * (1 + 1e-6))


WGOLD = os.path.join(os.path.dirname(__file__), "golden", "vmc_world_golden.npz")


def _run_world(ctx, pkg, vin, q, ratio):
    n = vin.shape[0]
    S = pkg.to_soa
    d_in = ctx.alloc((37, n)).upload(S(vin)); d_q = ctx.alloc((12, n)).upload(S(q)); d_r = ctx.alloc((8, n)).upload(S(ratio))
    d_f = ctx.alloc((12, n)); d_t = ctx.alloc((12, n)); d_s = ctx.alloc((n,), np.int32)
    ctx.vmc_force_world_batch(n, d_in, d_r, d_q, d_f, d_t, d_s)
    ctx.sync()
    out = dict(force=d_f.download().T.copy(), tau=d_t.download().T.copy(), status=d_s.download())
    for v in (d_in, d_q, d_r, d_f, d_t, d_s):
        v.free()
    return out


def test_vmc_world_frame_parity(gpu_ctx, pkg, oracle):
    """World-frame overload (qr_qp_torque_optimizer.cpp:304-398) against the oracle, and against QuadProg++ through the golden file."""
    W = pkg.workload
    cfg = W.vmc_cfg("a1"); geom = pkg.model_desc("a1")[:3]
    gpu_ctx.vmc_setup_packed(0, cfg, geom)
    n = 800
    vin, q, ratio = W.make_vmc_world_batch(n, seed=23)
    g = _run_world(gpu_ctx, pkg, vin, q, ratio)
    flags = G.flags(g["status"])
    assert np.all((flags & ~0x80) == 0), np.unique(flags)
    n_inf = 0
    for i in range(n):
        force, tau, x, st, rc = oracle.vmc_solve(cfg, geom, vin[i], q[i], ratio[i])
        assert bool(flags[i] & 0x80) == (rc == 1), (i, flags[i], rc)
        n_inf += rc == 1
        assert np.abs(g["force"][i] - force).max() <= 1e-5 * max(1.0, np.abs(force).max()), (i, np.abs(g["force"][i] - force).max())
        assert np.all(np.abs(g["tau"][i] - tau) <= tau_tol(tau, 1e-4)), i
    assert 0 < n_inf < n
    gd = np.load(WGOLD)
    gpu_ctx.vmc_setup_packed(0, gd["cfg"], gd["geom"])
    out = _run_world(gpu_ctx, pkg, gd["vin"], gd["q"], gd["ratio"])
    for i in range(gd["vin"].shape[0]):
        R = gd["vin"][i, 22:31].reshape(3, 3).astype(np.float64)
        f_ref = ((-gd["x_quadprog"][i].reshape(4, 3)) @ R).reshape(-1)
        assert np.abs(out["force"][i] - f_ref).max() <= 1e-5 * max(1.0, np.abs(f_ref).max()), i
        assert bool(out["status"][i] & 0x80) == bool(gd["quadprog_inf"][i])
    # the ratio array is mandatory for this entry point
    with pytest.raises(pkg.QrgpuError):
        d = gpu_ctx.alloc((37, 4)); f = gpu_ctx.alloc((12, 4))
        try:
            gpu_ctx.vmc_force_world_batch(4, d, None, None, f)
        finally:
            d.free(); f.free()
