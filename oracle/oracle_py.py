"""TEST INFRASTRUCTURE -- ctypes bindings for oracle/libqr_oracle.so (our CPU
restatement) and oracle/_ref/libqr_ref.so (the reference's vendored qpOASES /
QuadProg++ compiled from /root/reference; optional).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f = np.float32
_d = np.float64


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _fp(a):
    return _ptr(a, C.c_float)


def _dp(a):
    return _ptr(a, C.c_double)


def _ip(a):
    return _ptr(a, C.c_int)


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(x) > t for x in sources if os.path.exists(x))


def build(force=False):
    """Compile libqr_oracle.so (and _ref when /root/reference exists).  make decides what is out of date."""
    so = os.path.join(_HERE, "libqr_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h")) and f != "ref_shim.cpp"] + [os.path.join(_HERE, "Makefile")]
    if force or _stale(so, srcs):
        subprocess.check_call(["make", "-C", _HERE, "-j4", "all"], stdout=subprocess.DEVNULL)
    refso = os.path.join(_HERE, "_ref", "libqr_ref.so")
    if os.path.isdir("/root/reference/quadruped/extern/qpOASES/src") and (
            force or _stale(refso, [os.path.join(_HERE, "ref_shim.cpp"), os.path.join(_HERE, "Makefile")])):
        subprocess.check_call(["make", "-C", _HERE, "-j4", "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.path.join(_HERE, "libqr_oracle.so"))
        _lib.qro_tick_batch.restype = C.c_double
    return _lib


def ref():
    """The compiled reference solvers, or None when oracle/_ref was not built."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libqr_ref.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
        _ref.ref_quadprog.restype = C.c_double
    return _ref


# ----------------------------------------------------------------------------- packed layouts
def pack_mpc_cfg(dt=0.06, mu=0.45, fmax=13 * 9.81, mass=13.0, inertia=(0.24, 0.80, 1.0),
                 weights=(10, 10, 5, 40, 60, 100, 0, 0, 0.5, 5, 5, 1), alpha=4e-6):
    return np.array([dt, mu, fmax, mass, *inertia, *weights, alpha], dtype=_f)


A1_GEOM = np.array([0.08505, 0.2, 0.2], dtype=_f)
A1_MODEL = np.array([0.08505, 0.2, 0.2, 0.267, 0.194, 0.114], dtype=_f)
LITE3_GEOM = np.array([0.0985, 0.20, 0.20], dtype=_f)
LITE3_MODEL = np.array([0.0985, 0.20, 0.20, 0.349, 0.124, 0.15], dtype=_f)


# ----------------------------------------------------------------------------- QP
def qp_solve(G, g0, CE, ce0, CI, ci0):
    """min 1/2 x'Gx+g0'x, CE'x+ce0=0, CI'x+ci0>=0 (QuadProg++ convention).  -> x, lambda, stats, rc"""
    G = np.ascontiguousarray(G, _d); g0 = np.ascontiguousarray(g0, _d)
    n = g0.size
    CE = np.ascontiguousarray(CE, _d).reshape(n, -1) if CE is not None and np.size(CE) else np.zeros((n, 0))
    CI = np.ascontiguousarray(CI, _d).reshape(n, -1) if CI is not None and np.size(CI) else np.zeros((n, 0))
    ce0 = np.ascontiguousarray(ce0, _d) if ce0 is not None else np.zeros(0)
    ci0 = np.ascontiguousarray(ci0, _d) if ci0 is not None else np.zeros(0)
    p, m = CE.shape[1], CI.shape[1]
    x = np.zeros(n); lam = np.zeros(max(m, 1)); st = np.zeros(4, np.int32); obj = C.c_double(0)
    rc = lib().qro_qp_solve(n, _dp(G), _dp(g0), p, _dp(CE), _dp(ce0), m, _dp(CI), _dp(ci0), _dp(x), _dp(lam), _ip(st), C.byref(obj))
    return x, lam[:m], dict(iters=int(st[0]), adds=int(st[1]), drops=int(st[2]), n_active=int(st[3]), obj=obj.value), rc


def ref_quadprog(G, g0, CE, ce0, CI, ci0):
    r = ref()
    G = np.ascontiguousarray(G, _d).copy(); g0 = np.ascontiguousarray(g0, _d).copy()
    n = g0.size
    CE = np.ascontiguousarray(CE, _d).reshape(n, -1); CI = np.ascontiguousarray(CI, _d).reshape(n, -1)
    ce0 = np.ascontiguousarray(ce0, _d); ci0 = np.ascontiguousarray(ci0, _d)
    x = np.zeros(n)
    f = r.ref_quadprog(n, CE.shape[1], CI.shape[1], _dp(G), _dp(g0), _dp(CE), _dp(ce0), _dp(CI), _dp(ci0), _dp(x))
    return x, f


def ref_qpoases_mpc(H, g, A, lbA, ubA, nWSR=100):
    """Exactly the reference's solver call (qr_mpc_interface.cpp:428-438).  -> x, info"""
    r = ref()
    H = np.ascontiguousarray(H, _d); g = np.ascontiguousarray(g, _d); A = np.ascontiguousarray(A, _d)
    lbA = np.ascontiguousarray(lbA, _d); ubA = np.ascontiguousarray(ubA, _d)
    n, m = g.size, lbA.size
    x = np.zeros(n); nw = C.c_int(0); prc = C.c_int(0); obj = C.c_double(0)
    rc = r.ref_qpoases_mpc(n, m, _dp(H), _dp(g), _dp(A), _dp(lbA), _dp(ubA), int(nWSR), _dp(x), C.byref(nw), C.byref(prc), C.byref(obj))
    return x, dict(init_rc=rc, nWSR=nw.value, primal_rc=prc.value, obj=obj.value)


def mpc_constraint_matrix(horizon, mu=0.45):
    """fmat of ResizeQPMats (qr_mpc_interface.cpp:230-240) as float32 -> float64, 20h x 12h."""
    im = np.float32(1.0) / np.float32(mu)
    blk = np.array([[im, 0, 1], [-im, 0, 1], [0, im, 1], [0, -im, 1], [0, 0, 1]], dtype=_f)
    A = np.zeros((20 * horizon, 12 * horizon), dtype=_f)
    for i in range(4 * horizon):
        A[5 * i:5 * i + 5, 3 * i:3 * i + 3] = blk
    return A.astype(_d)


# ----------------------------------------------------------------------------- MPC
def mpc_assemble(cfg, horizon, state28, traj, gait, literal=False):
    n, m = 12 * horizon, 20 * horizon
    H = np.zeros((n, n), _f); g = np.zeros(n, _f); ub = np.zeros(m, _f)
    state28 = np.ascontiguousarray(state28, _f); traj = np.ascontiguousarray(traj, _f); gait = np.ascontiguousarray(gait, _f)
    lib().qro_mpc_assemble(_fp(cfg), horizon, _fp(state28), _fp(traj), _fp(gait), int(literal), _fp(H), _fp(g), _fp(ub))
    return H, g, ub


def mpc_solve(cfg, horizon, state28, traj, gait, literal=False):
    n = 12 * horizon
    u = np.zeros(n); st = np.zeros(4, np.int32)
    state28 = np.ascontiguousarray(state28, _f); traj = np.ascontiguousarray(traj, _f); gait = np.ascontiguousarray(gait, _f)
    rc = lib().qro_mpc_solve(_fp(cfg), horizon, _fp(state28), _fp(traj), _fp(gait), int(literal), _dp(u), _ip(st))
    return u, dict(iters=int(st[0]), adds=int(st[1]), drops=int(st[2]), n_active=int(st[3])), rc


def mpc_solve_hg(cfg, horizon, gait, H, g):
    """The stated QP solved from a given fp32 (H, g) (whatever assembled them) with the bounds / friction rows of `gait` and `cfg`.  -> u, stats, rc"""
    n = 12 * horizon
    u = np.zeros(n); st = np.zeros(4, np.int32)
    H = np.ascontiguousarray(H, _f); g = np.ascontiguousarray(g, _f); gait = np.ascontiguousarray(gait, _f)
    assert H.shape == (n, n) and g.shape == (n,)
    rc = lib().qro_mpc_solve_hg(_fp(cfg), horizon, _fp(gait), _fp(H), _fp(g), _dp(u), _ip(st))
    return u, dict(iters=int(st[0]), adds=int(st[1]), drops=int(st[2]), n_active=int(st[3])), rc


def mpc_force_to_torque(geom, quat, q12, f12):
    tau = np.zeros(12, _f)
    lib().qro_mpc_force_to_torque(_fp(np.ascontiguousarray(geom, _f)), _fp(np.ascontiguousarray(quat, _f)),
                                  _fp(np.ascontiguousarray(q12, _f)), _dp(np.ascontiguousarray(f12, _d)), _fp(tau))
    return tau


def foot_positions(geom, hip_offset12, q12):
    out = np.zeros(12, _f)
    lib().qro_foot_positions(_fp(np.ascontiguousarray(geom, _f)), _fp(np.ascontiguousarray(hip_offset12, _f)),
                             _fp(np.ascontiguousarray(q12, _f)), _fp(out))
    return out


def leg_jacobian(geom, q3, leg):
    J = np.zeros(9, _f)
    lib().qro_leg_jacobian(_fp(np.ascontiguousarray(geom, _f)), _fp(np.ascontiguousarray(q3, _f)), int(leg), _fp(J))
    return J.reshape(3, 3)


def rpy_to_quat(rpy):
    q = np.zeros(4, _f)
    lib().qro_rpy_to_quat(_fp(np.ascontiguousarray(rpy, _f)), _fp(q))
    return q


# ----------------------------------------------------------------------------- dynamics / WBC
def fb_compute(model, state37, dtype=_f):
    t = dtype
    H = np.zeros((18, 18), t); G = np.zeros(18, t); Cq = np.zeros(18, t)
    Jc = np.zeros((4, 3, 18), t); Jcd = np.zeros((4, 3), t); p = np.zeros((4, 3), t); v = np.zeros((4, 3), t)
    s = np.ascontiguousarray(state37, t)
    conv = _fp if t == _f else _dp
    fn = lib().qro_fb_compute_f32 if t == _f else lib().qro_fb_compute_f64
    fn(_fp(np.ascontiguousarray(model, _f)), conv(s), conv(H), conv(G), conv(Cq), conv(Jc), conv(Jcd), conv(p), conv(v))
    return dict(H=H, G=G, C=Cq, Jc=Jc, Jcdqd=Jcd, pGC=p, vGC=v)


def wbc_run(model, state37, cmd67, prev_ori_vel=None, dtype=_f):
    t = dtype
    conv = _fp if t == _f else _dp
    s = np.ascontiguousarray(state37, t); c = np.ascontiguousarray(cmd67, t)
    prev = np.zeros(3, t) if prev_ori_vel is None else np.ascontiguousarray(prev_ori_vel, t).copy()
    tau = np.zeros(12, t); qdes = np.zeros(12, t); qddes = np.zeros(12, t); fr = np.zeros(12, t); qdd = np.zeros(18, t)
    st = np.zeros(4, np.int32)
    fn = lib().qro_wbc_run_f32 if t == _f else lib().qro_wbc_run_f64
    rc = fn(_fp(np.ascontiguousarray(model, _f)), conv(s), conv(c), conv(prev), conv(tau), conv(qdes), conv(qddes), conv(fr), conv(qdd), _ip(st))
    return dict(tau=tau, qdes=qdes, qddes=qddes, fr=fr, qddot=qdd, prev_ori_vel=prev, rc=rc,
                qp=dict(iters=int(st[0]), adds=int(st[1]), drops=int(st[2]), n_active=int(st[3])))


def wbc_qp(model, state37, cmd67, prev_ori_vel=None, dtype=_f, z_in=None):
    """The WBC tick with its relaxation QP laid open (qr_wholebody_impulse_ctrl.cpp:113, :129-206, :232-247): the QP as the oracle assembles it in
    QuadProg++ layout (G [n,n], g0 [n], CE [n,p], ce0 [p], CI [n,m], ci0 [m]), the solver's z, and tau / optimalFr of the tick -- finished
    with `z_in` instead of the oracle's own solution when that is given (the compiled QuadProg++'s).  prev_ori_vel is not modified."""
    t = dtype
    conv = _fp if t == _f else _dp
    s = np.ascontiguousarray(state37, t); c = np.ascontiguousarray(cmd67, t)
    prev = np.zeros(3, t) if prev_ori_vel is None else np.ascontiguousarray(prev_ori_vel, t)
    dims = np.zeros(3, np.int32)
    G = np.zeros(18 * 18); g0 = np.zeros(18); CE = np.zeros(18 * 6); ce0 = np.zeros(6); CI = np.zeros(18 * 24); ci0 = np.zeros(24); z = np.zeros(18)
    tau = np.zeros(12, t); fr = np.zeros(12, t)
    zi = np.ascontiguousarray(z_in, _d) if z_in is not None else None
    fn = lib().qro_wbc_qp_f32 if t == _f else lib().qro_wbc_qp_f64
    rc = fn(_fp(np.ascontiguousarray(model, _f)), conv(s), conv(c), conv(prev), _dp(zi) if zi is not None else None, _ip(dims), _dp(G), _dp(g0), _dp(CE),
            _dp(ce0), _dp(CI), _dp(ci0), _dp(z), conv(tau), conv(fr))
    n, p, m = (int(x) for x in dims)
    return dict(n=n, p=p, m=m, G=G[:n * n].reshape(n, n).copy(), g0=g0[:n].copy(), CE=CE[:n * p].reshape(n, p).copy(), ce0=ce0[:p].copy(),
                CI=CI[:n * m].reshape(n, m).copy(), ci0=ci0[:m].copy(), z=z[:n].copy(), tau=tau, fr=fr, rc=rc)


def tick_from_forces(geom, model, fb_state37, wbc_cmd67, prev_ori_vel, f12, mode=1, epilogue=0, wbc_fp64=False, want_qdes=False):
    """One robot's tick behind the MPC solve from given first-step forces: K7 torque map, abad compensation, WBC fed with Fr_des := f12,
    stance / swing merge, clip (qr_oracle_capi.cpp tick_tail).  -> tau[12] float32, prev (updated copy) (, qdes[24])"""
    prev = np.ascontiguousarray(prev_ori_vel, _f).copy()
    tau = np.zeros(12, _f); qdes = np.zeros(24, _f)
    lib().qro_tick_from_forces(_fp(np.ascontiguousarray(geom, _f)), _fp(np.ascontiguousarray(model, _f)), _fp(np.ascontiguousarray(fb_state37, _f)),
                               _fp(np.ascontiguousarray(wbc_cmd67, _f)), _fp(prev), _dp(np.ascontiguousarray(f12, _d)), int(mode), int(epilogue),
                               int(bool(wbc_fp64)), _fp(tau), _fp(qdes))
    return (tau, prev, qdes) if want_qdes else (tau, prev)


def pinv(A, thr):
    A = np.ascontiguousarray(A, _f)
    out = np.zeros((A.shape[1], A.shape[0]), _f)
    lib().qro_pinv_f32(A.shape[0], A.shape[1], _fp(A), C.c_double(float(thr)), _fp(out))
    return out


def lu_inverse(A):
    A = np.ascontiguousarray(A, _f)
    out = np.zeros_like(A)
    lib().qro_lu_inverse_f32(A.shape[0], _fp(A), _fp(out))
    return out


def tick_batch(mode, cfg, horizon, geom, model, mpc_state, traj, gait, fb_state, wbc_cmd, prev_ori_vel, nthreads=1, epilogue=0, want_qdes=False):
    """Robot-major (AoS) batched tick on host threads.  -> force[n,12], tau[n,12], status[n], seconds, prev (, qdes[n,24] with want_qdes)
    epilogue: 1 = abad hip compensation, 2 = +-23 N m clip (K14 tail, qr_fsm_state_locomotion.cpp:141-151, qr_safety_checker.cpp:48-66)."""
    n = mpc_state.shape[0]
    force = np.zeros((n, 12), _f); tau = np.zeros((n, 12), _f); status = np.zeros(n, np.int32)
    qdes = np.zeros((n, 24), _f) if want_qdes else None
    arrs = [np.ascontiguousarray(a, _f) for a in (mpc_state, traj, gait, fb_state, wbc_cmd)]
    prev = np.ascontiguousarray(prev_ori_vel, _f)
    sec = lib().qro_tick_batch(n, int(nthreads), int(mode), _fp(cfg), horizon, _fp(np.ascontiguousarray(geom, _f)),
                               _fp(np.ascontiguousarray(model, _f)), _fp(arrs[0]), _fp(arrs[1]), _fp(arrs[2]), _fp(arrs[3]),
                               _fp(arrs[4]), _fp(prev), _fp(force), _fp(tau), _ip(status), int(epilogue), _fp(qdes) if want_qdes else None)
    if want_qdes:
        return force, tau, status, sec, prev, qdes
    return force, tau, status, sec, prev


def mpc_frontend(horizon, num_horizon_l, in64, st8, dt=0.002, dt_mpc=0.06):
    """-> traj[12h] (or None when this tick does not re-plan), gait[4h], wbc15, contact4, new state, updated flag"""
    i = np.ascontiguousarray(in64, _f); st = np.ascontiguousarray(st8, _f).copy()
    traj = np.full(12 * horizon, np.nan, _f); gait = np.zeros(4 * horizon, _f); wbc = np.zeros(15, _f); ct = np.zeros(4, _f)
    upd = C.c_int(0)
    lib().qro_mpc_frontend(int(horizon), int(num_horizon_l), C.c_float(dt), C.c_float(dt_mpc), _fp(i), _fp(st), _fp(traj), _fp(gait), _fp(wbc), _fp(ct), C.byref(upd))
    return dict(traj=traj, gait=gait, wbc15=wbc, contact=ct, state=st, updated=upd.value)


def vmc_assemble(cfg20, in37, ratio8=None):
    """fp32 QP data of ComputeContactForce: G[12,12], a[12], CI[12,24] (= A^T), b[24]"""
    G = np.zeros((12, 12), _f); a = np.zeros(12, _f); CI = np.zeros((12, 24), _f); b = np.zeros(24, _f)
    r8 = np.ascontiguousarray(ratio8, _f) if ratio8 is not None else None
    lib().qro_vmc_assemble(_fp(np.ascontiguousarray(cfg20, _f)), _fp(np.ascontiguousarray(in37, _f)), _fp(r8) if r8 is not None else None, _fp(G), _fp(a), _fp(CI), _fp(b))
    return G, a, CI, b


def vmc_solve(cfg20, geom3, in37, q12=None, ratio8=None):
    """-> force[12] (3*leg+axis, base frame), tau[12] or None, x[12] (raw QuadProg solution), stats, rc"""
    force = np.zeros(12, _f); tau = np.zeros(12, _f); x = np.zeros(12); st = np.zeros(4, np.int32)
    qa = np.ascontiguousarray(q12, _f) if q12 is not None else None
    r8 = np.ascontiguousarray(ratio8, _f) if ratio8 is not None else None
    rc = lib().qro_vmc_solve(_fp(np.ascontiguousarray(cfg20, _f)), _fp(np.ascontiguousarray(geom3, _f)), _fp(np.ascontiguousarray(in37, _f)),
                             _fp(r8) if r8 is not None else None, _fp(qa) if qa is not None else None, _fp(force), _fp(tau) if qa is not None else None, _dp(x), _ip(st))
    return force, (tau if qa is not None else None), x, dict(iters=int(st[0]), adds=int(st[1]), drops=int(st[2]), n_active=int(st[3])), rc


def ekf3_run(qvar, rvar, deltaV, z):
    """Our restatement of TinyEKF<3,3> with the velocity estimator's model, stepped from a fresh filter.  -> x after each step, failures"""
    dv = np.ascontiguousarray(deltaV, _d); zz = np.ascontiguousarray(z, _d)
    out = np.zeros_like(dv)
    lib().qro_ekf3_run.argtypes = [C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    bad = lib().qro_ekf3_run(float(qvar), float(rvar), dv.shape[0], _dp(dv), _dp(zz), _dp(out))
    return out, bad


def ref_tinyekf_run(acc_var, sensor_var, deltaV, z):
    """The reference's own TinyEKF<3,3> (compiled from /root/reference into oracle/_ref), driven as qrRobotVelocityEstimator does."""
    r = ref()
    dv = np.ascontiguousarray(deltaV, _d); zz = np.ascontiguousarray(z, _d)
    out = np.zeros_like(dv)
    r.ref_tinyekf_run.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    bad = r.ref_tinyekf_run(float(acc_var), float(sensor_var), dv.shape[0], _dp(dv), _dp(zz), _dp(out))
    return out, bad


def estimator_run(cfg20, in54, tick):
    """A sequence of velocity + pose estimator updates of one robot from fresh estimators.  in54 [T][54], tick [T] (ms) -> out [T][42]"""
    a = np.ascontiguousarray(in54, _f); t = np.ascontiguousarray(tick, np.uint32)
    assert a.shape[1] == 54
    out = np.zeros((a.shape[0], 42), _f)
    lib().qro_estimator_run(_fp(np.ascontiguousarray(cfg20, _f)), a.shape[0], _fp(a), t.ctypes.data_as(C.c_void_p), _fp(out))
    return out


def footholds(desc29, in46, swing_in58=None):
    """Swing-leg selection + foothold heuristic (qr_oracle_swing.cpp footholds).  -> swing_in[58] with rows 0-7, 24-35 updated"""
    out = np.zeros(58, _f) if swing_in58 is None else np.ascontiguousarray(swing_in58, _f).copy()
    lib().qro_footholds(_fp(np.ascontiguousarray(desc29, _f)), _fp(np.ascontiguousarray(in46, _f)), _fp(out))
    return out


def swing_targets(geom3, hip_offset12, in58, out72_prev=None):
    """Swing-leg targets (ADVANCED_TROT, horizontal terrain).  -> out[72]; rows of stance legs keep out72_prev (NaN if not given)."""
    out = np.full(72, np.nan, _f) if out72_prev is None else np.ascontiguousarray(out72_prev, _f).copy()
    lib().qro_swing_targets(_fp(np.ascontiguousarray(geom3, _f)), _fp(np.ascontiguousarray(hip_offset12, _f)), _fp(np.ascontiguousarray(in58, _f)), _fp(out))
    return out


def walk_run(cfg26, time, contact, stop=None):
    """Walk gait generator of one robot as constructed + Reset(0): time [T], contact [T][4] -> out [T][41] (phaseInFullCycle,
    normalizedPhase, desiredLegState, legState, curLegState, detectedLegState, detectedEventTickPhase, moveBasePhase, contacts, fMinRatio,
    fMaxRatio)."""
    t = np.ascontiguousarray(time, _f); c = np.ascontiguousarray(contact, _f)
    out = np.zeros((t.shape[0], 41), _f)
    st = np.ascontiguousarray(stop, np.int32) if stop is not None else None
    lib().qro_walk_run(_fp(np.ascontiguousarray(cfg26, _f)), t.shape[0], _fp(t), _fp(c), _ip(st) if st is not None else None, _fp(out))
    return out


def ground_run(in23):
    """Ground-plane estimator of one robot from Reset(): in23 [T][23] (contact[4], footPositionsInBaseFrame[12], basePosition[3], quat[4])
    -> out [T][32] (a[3], n[3], controlFrameRPY[3], controlFrameOrientation[4], groundRMat[9], baseRInControlFrame[9], updated)."""
    a = np.ascontiguousarray(in23, _f)
    assert a.ndim == 2 and a.shape[1] == 23
    out = np.zeros((a.shape[0], 32), _f)
    lib().qro_ground_run(a.shape[0], _fp(a), _fp(out))
    return out


def swing_velocity(geom3, hip_offset12, desc20, in53, out48_prev=None):
    """Swing-leg action of the velocity mode (qr_swing_leg_controller.cpp:285-309, 408-424).  -> out[48]: footTargetPosition[12],
    footPositionInBaseFrame[12], joint angle targets[12], joint velocity targets[12]; legs that are not flagged keep out48_prev (NaN)."""
    out = np.full(48, np.nan, _f) if out48_prev is None else np.ascontiguousarray(out48_prev, _f).copy()
    lib().qro_swing_velocity(_fp(np.ascontiguousarray(geom3, _f)), _fp(np.ascontiguousarray(hip_offset12, _f)), _fp(np.ascontiguousarray(desc20, _f)),
                             _fp(np.ascontiguousarray(in53, _f)), _fp(out))
    return out


def gait_run(cfg19, time, contact, stop=None):
    """Open-loop gait generator of one robot from Reset(0): time [T], contact [T][4] -> out [T][24]
    (phaseInFullCycle, normalizedPhase, desiredLegState, legState, curLegState, swingTimeRemaining)."""
    t = np.ascontiguousarray(time, _f); c = np.ascontiguousarray(contact, _f)
    out = np.zeros((t.shape[0], 24), _f)
    st = np.ascontiguousarray(stop, np.int32) if stop is not None else None
    lib().qro_gait_run(_fp(np.ascontiguousarray(cfg19, _f)), t.shape[0], _fp(t), _fp(c), _ip(st) if st is not None else None, _fp(out))
    return out
