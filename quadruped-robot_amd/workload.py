"""Deterministic synthetic per-tick inputs for batches of quadrupeds.

Pure numpy host code.  Produces, robot-major (AoS, shape [n, k], float32):
  mpc_state[n,28] = p, v_world, quat_wxyz, w_world, r (3x4 col-major: foot - CoM, world aligned), rpy
                    -- the arguments of SolveMPCKernel (qr_mpc_interface.h:200) as
                    SolveDenseMPC prepares them (qr_mpc_stance_leg_controller.cpp:385-399)
  traj[n,12h]      -- trajAll of UpdateMPC (qr_mpc_stance_leg_controller.cpp:361-376)
  gait[n,4h]       -- mpcTable, row-major h x 4 (:283-303)
  fb_state[n,37]   = quat_wxyz, pos, [omega_body, v_body], q, qd  (FBModelState, floating_base_model.hpp:29-44)
  wbc_cmd[n,67]    = qrWbcCtrlData fields (qr_state_dataflow.h:133-192), contact as 0/1 floats
The device API takes the transposed (SoA, [k, n]) arrays: see qrgpu.to_soa().
"""
import numpy as np

f32 = np.float32

ROBOTS = {
    # Q/config/a1_sim/a1_sim.yaml, stance_leg_controller.yaml; SURVEY.md 8 "Constants"
    "a1": dict(type_id=0, mass=13.0, inertia=(0.24, 0.80, 1.0),
               weights=(10, 10, 5, 40, 60, 100, 0, 0, 0.5, 5, 5, 1),
               hip_l=0.08505, upper_l=0.2, lower_l=0.2, body_size=(0.267, 0.194, 0.114),
               com_offset=(-0.008, 0.005, 0.0),
               hip_offset=((0.1805, -0.047, 0.0), (0.1805, 0.047, 0.0), (-0.1805, -0.047, 0.0), (-0.1805, 0.047, 0.0)),
               default_hip_position=((0.185, -0.135, 0.0), (0.185, 0.135, 0.0), (-0.185, -0.135, 0.0), (-0.185, 0.135, 0.0))),
    # Q/config/lite3_sim/robot.yaml, stance_leg_controller.yaml
    "lite3": dict(type_id=1, mass=8.742, inertia=(0.24, 0.8, 1.0),
                  weights=(20, 20, 10, 40, 40, 150, 0.5, 1, 1, 5, 5, 10),
                  hip_l=0.0985, upper_l=0.20, lower_l=0.20, body_size=(0.349, 0.124, 0.15),
                  com_offset=(-0.012, 0.0, 0.0),
                  hip_offset=((0.175, -0.062, 0.0), (0.175, 0.062, 0.0), (-0.175, -0.062, 0.0), (-0.175, 0.062, 0.0)),
                  default_hip_position=((0.1745, -0.15, 0.0), (0.1745, 0.15, 0.0), (-0.1745, -0.15, 0.0), (-0.1745, 0.15, 0.0))),
}
DT_MPC = 0.06      # qr_mpc_stance_leg_controller.cpp:43
MU_MPC = 0.45      # :90
ALPHA = 4e-6       # :85


def mpc_cfg(robot="a1"):
    """Packed float32[20]: dt, mu, fmax, mass, inertia[3], weights[12], alpha (SetupProblem arguments, :83-90)."""
    r = ROBOTS[robot]
    return np.array([DT_MPC, MU_MPC, r["mass"] * 9.81, r["mass"], *r["inertia"], *r["weights"], ALPHA], dtype=f32)


def model_desc(robot="a1"):
    """Packed float32[6]: hip_l, upper_l, lower_l, body_size[3] (the YAML-dependent part of BuildDynamicModel)."""
    r = ROBOTS[robot]
    return np.array([r["hip_l"], r["upper_l"], r["lower_l"], *r["body_size"]], dtype=f32)


def _rot_body_to_world(rpy):
    """R = Rz(yaw) Ry(pitch) Rx(roll), batched [n,3,3]."""
    cr, sr = np.cos(rpy[:, 0]), np.sin(rpy[:, 0])
    cp, sp = np.cos(rpy[:, 1]), np.sin(rpy[:, 1])
    cy, sy = np.cos(rpy[:, 2]), np.sin(rpy[:, 2])
    R = np.empty((rpy.shape[0], 3, 3))
    R[:, 0, 0] = cy * cp; R[:, 0, 1] = cy * sp * sr - sy * cr; R[:, 0, 2] = cy * sp * cr + sy * sr
    R[:, 1, 0] = sy * cp; R[:, 1, 1] = sy * sp * sr + cy * cr; R[:, 1, 2] = sy * sp * cr - cy * sr
    R[:, 2, 0] = -sp;     R[:, 2, 1] = cp * sr;                R[:, 2, 2] = cp * cr
    return R


def _quat_from_rpy(rpy):
    """(w,x,y,z) of R = Rz Ry Rx."""
    hr, hp, hy = rpy[:, 0] / 2, rpy[:, 1] / 2, rpy[:, 2] / 2
    cr, sr, cp, sp, cy, sy = np.cos(hr), np.sin(hr), np.cos(hp), np.sin(hp), np.cos(hy), np.sin(hy)
    q = np.stack([cr * cp * cy + sr * sp * sy,
                  sr * cp * cy - cr * sp * sy,
                  cr * sp * cy + sr * cp * sy,
                  cr * cp * sy - sr * sp * cy], axis=1)
    return q


def _foot_positions_base(r, q):
    """qrRobot::FootPositionsInBaseFrame (QS/robots/qr_robot.cpp:127-146,175-184), batched [n,4,3]."""
    n = q.shape[0]
    out = np.empty((n, 4, 3))
    lu, ll = r["upper_l"], r["lower_l"]
    for leg in range(4):
        tab, thip, tknee = q[:, 3 * leg], q[:, 3 * leg + 1], q[:, 3 * leg + 2]
        sh = r["hip_l"] * (-1.0) ** (leg + 1)
        ld = np.sqrt(lu * lu + ll * ll + 2 * lu * ll * np.cos(tknee))
        eff = thip + tknee / 2
        ox, oz, oy = -ld * np.sin(eff), -ld * np.cos(eff), sh
        out[:, leg, 0] = ox + r["hip_offset"][leg][0]
        out[:, leg, 1] = np.cos(tab) * oy - np.sin(tab) * oz + r["hip_offset"][leg][1]
        out[:, leg, 2] = np.sin(tab) * oy + np.cos(tab) * oz + r["hip_offset"][leg][2]
    return out


def _draw_population(n, horizon, robot, seed, frac_all_stance, frac_three_leg, excite):
    """Every random draw of one batch, in a fixed order (seeded batches are part of the tests' contract), as a dict of float64 arrays."""
    r = ROBOTS[robot]
    rng = np.random.default_rng(seed)
    U = rng.uniform
    e = float(excite)
    v = dict(robot=robot, horizon=horizon, n=n, excite=e)
    v["rpy"] = np.stack([e * U(-0.15, 0.15, n), e * U(-0.15, 0.15, n), U(-np.pi, np.pi, n)], axis=1)
    v["pos"] = np.stack([U(-1, 1, n), U(-1, 1, n), 0.27 + U(-0.03, 0.03, n)], axis=1)
    v["v_w"] = np.stack([U(-0.5, 0.5, n), U(-0.5, 0.5, n), e * U(-0.1, 0.1, n)], axis=1)
    v["w_w"] = e * U(-0.5, 0.5, (n, 3))
    # joints: stand pose (0, 0.8+-0.2, -1.6+-0.3), abad U(-0.2,0.2)
    q = np.empty((n, 12)); v["qd"] = U(-1, 1, (n, 12))
    for leg in range(4):
        q[:, 3 * leg] = U(-0.2, 0.2, n)
        q[:, 3 * leg + 1] = 0.8 + U(-0.2, 0.2, n)
        q[:, 3 * leg + 2] = -1.6 + U(-0.3, 0.3, n)
    v["q"] = q
    v["phase0"] = U(0, 1, n)
    kind = U(0, 1, n)
    v["all_st"] = kind < frac_all_stance
    v["three"] = (kind >= frac_all_stance) & (kind < frac_all_stance + frac_three_leg)
    v["sw_leg"] = rng.integers(0, 4, n)
    # command (UpdateMPC :361-376): at excite < 1 it is blended towards the current motion
    R = _rot_body_to_world(v["rpy"])
    vdes_b = np.stack([U(-0.5, 1.0, n), U(-0.3, 0.3, n), np.zeros(n)], axis=1)
    v_b_now = np.einsum("nji,nj->ni", R, v["v_w"])
    vdes_b[:, :2] = v_b_now[:, :2] + e * (vdes_b[:, :2] - v_b_now[:, :2])     # e=1: independent command
    v["vdes_b"] = vdes_b
    v["yaw_rate"] = v["w_w"][:, 2] + e * (U(-0.5, 0.5, n) - v["w_w"][:, 2])
    v["d_yaw"] = U(-0.05, 0.05, n)
    v["d_x0"] = U(-0.05, 0.05, n); v["d_y0"] = U(-0.05, 0.05, n)
    v["d_height"] = U(-0.01, 0.01, n)
    v["d_foot"] = U(-0.05, 0.05, (n, 4, 3))
    v["v_foot"] = U(-0.5, 0.5, (n, 12))
    v["a_foot"] = U(-2.0, 2.0, (n, 12))
    v["_rng"] = rng
    return v


def _assemble(v):
    """The tick's input arrays from a population's variables (deterministic)."""
    r = ROBOTS[v["robot"]]
    n, h = v["n"], v["horizon"]
    rpy, pos, v_w, w_w, q, qd = v["rpy"], v["pos"], v["v_w"], v["w_w"], v["q"], v["qd"]
    R = _rot_body_to_world(rpy)
    quat = _quat_from_rpy(rpy)
    foot_b = _foot_positions_base(r, q)                                   # [n,4,3]
    r_w = np.einsum("nij,nlj->nli", R, foot_b - np.asarray(r["com_offset"]))   # R (foot - comOffset)
    # gait table: trot, duty 0.6, random phase; dPhase = 1/(numHorizonL*h), numHorizonL = 2  (:50,:284)
    duty = 0.6
    offs = np.array([0.0, 0.5, 0.5, 0.0])
    dphase = 1.0 / (2 * h)
    ph = (v["phase0"][:, None, None] + offs[None, None, :] + dphase * np.arange(h)[None, :, None]) % 1.0
    gait = (ph < duty).astype(np.float64)                                  # [n,h,4]
    gait[v["all_st"]] = 1.0
    for i in np.nonzero(v["three"])[0]:
        gait[i] = 1.0
        gait[i, :, v["sw_leg"][i]] = 0.0
    contact = gait[:, 0, :].copy()
    # reference trajectory (UpdateMPC :361-376)
    vdes_w = np.einsum("nij,nj->ni", R, v["vdes_b"])
    yaw_rate = v["yaw_rate"]
    yaw_des = rpy[:, 2] + v["d_yaw"]
    x0 = pos[:, 0] + v["d_x0"]; y0 = pos[:, 1] + v["d_y0"]
    height = np.full(n, 0.27)
    traj = np.zeros((n, h, 12))
    traj[:, :, 5] = height[:, None]
    traj[:, :, 8] = yaw_rate[:, None]
    traj[:, :, 9] = vdes_w[:, 0:1]; traj[:, :, 10] = vdes_w[:, 1:2]
    k = np.arange(h)[None, :]
    traj[:, :, 2] = yaw_des[:, None] + DT_MPC * k * yaw_rate[:, None]
    traj[:, :, 3] = x0[:, None] + DT_MPC * k * vdes_w[:, 0:1]
    traj[:, :, 4] = y0[:, None] + DT_MPC * k * vdes_w[:, 1:2]
    # MPC state
    mpc_state = np.concatenate([pos, v_w, quat, w_w, r_w.reshape(n, 12), rpy], axis=1)
    # floating-base state (body-frame velocities)
    w_b = np.einsum("nji,nj->ni", R, w_w); v_b = np.einsum("nji,nj->ni", R, v_w)
    fb_state = np.concatenate([quat, pos, w_b, v_b, q, qd], axis=1)
    # WBC command
    foot_w = pos[:, None, :] + np.einsum("nij,nlj->nli", R, foot_b)
    pBody = np.stack([x0, y0, height + v["d_height"]], axis=1)
    cmd = np.zeros((n, 67))
    cmd[:, 0:3] = pBody
    cmd[:, 3:6] = np.stack([vdes_w[:, 0], vdes_w[:, 1], np.zeros(n)], axis=1)
    cmd[:, 9:12] = np.stack([np.zeros(n), np.zeros(n), yaw_des], axis=1)
    cmd[:, 12:15] = np.stack([np.zeros(n), np.zeros(n), yaw_rate], axis=1)
    cmd[:, 15:27] = (foot_w + v["d_foot"]).reshape(n, 12)
    cmd[:, 27:39] = v["v_foot"]
    cmd[:, 39:51] = v["a_foot"]
    nst = np.maximum(contact.sum(axis=1), 1.0)
    cmd[:, 51:63] = (contact[:, :, None] * np.array([0.0, 0.0, 1.0]) * (r["mass"] * 9.81 / nst)[:, None, None]).reshape(n, 12)
    cmd[:, 63:67] = contact
    return dict(robot=v["robot"], horizon=h, n=n,
                mpc_state=mpc_state.astype(f32), traj=traj.reshape(n, 12 * h).astype(f32),
                gait=gait.reshape(n, 4 * h).astype(f32), fb_state=fb_state.astype(f32), wbc_cmd=cmd.astype(f32),
                prev_ori_vel=np.zeros((n, 3), f32))


def make_batch(n, horizon=10, robot="a1", seed=0xA1, frac_all_stance=0.05, frac_three_leg=0.05, excite=1.0):
    """n robots of one type.  Returns a dict of float32 AoS arrays (see module docstring).

    excite scales the tracking errors the MPC has to remove (roll/pitch, angular rate,
    velocity mismatch, vertical velocity).  excite=1.0 is SURVEY.md 8d's full range, under
    which a good share of the QPs need more than qpOASES' nWSR=100 working-set changes (the
    reference then returns a non-optimal point)."""
    return _assemble(_draw_population(n, horizon, robot, seed, frac_all_stance, frac_three_leg, excite))


def make_batch_sequence(n, horizon=10, robot="a1", seed=0xA1, steps=8, dt=0.03, frac_all_stance=0.05, frac_three_leg=0.05, excite=1.0):
    """`steps` consecutive batches of the SAME n robots, dt seconds apart (0.03 s = the reference's MPC cadence of 15 control ticks,
    qr_mpc_stance_leg_controller.cpp:342): batch 0 is make_batch(seed); after that every robot moves on -- position and attitude integrate
    the velocities, joint angles their rates, velocities take a bounded random step (<= 0.6 m/s^2, 0.6 rad/s^2), everything is reflected
    back into SURVEY.md 8d's ranges, the gait phase advances by dt / 0.8333 s (stance 0.5 s at duty 0.6) so that the contact table scrolls,
    and the commands stay.  Temporally coherent, never the same batch twice.  Returns a list of batch dicts."""
    v = _draw_population(n, horizon, robot, seed, frac_all_stance, frac_three_leg, excite)
    rng = v.pop("_rng")
    e = v["excite"]
    out = [_assemble(v)]

    def reflect(x, rate, lo, hi):
        over, under = x > hi, x < lo
        x = np.where(over, 2 * hi - x, np.where(under, 2 * lo - x, x))
        if rate is not None:
            rate = np.where(over | under, -rate, rate)
        return x, rate

    dv = 0.018 if dt == 0.03 else 0.6 * dt          # <= 0.6 m/s^2 (rad/s^2) over one step
    for _ in range(1, steps):
        v = dict(v)
        v["v_w"] = v["v_w"] + rng.uniform(-dv, dv, (n, 3)) * np.array([1.0, 1.0, e * 0.2])
        v["w_w"] = v["w_w"] + e * rng.uniform(-dv, dv, (n, 3))
        for c, lim in ((0, 0.5), (1, 0.5), (2, e * 0.1 + 1e-9)):
            v["v_w"][:, c], _ = reflect(v["v_w"][:, c], None, -lim, lim)
        v["w_w"], _ = reflect(v["w_w"], None, -e * 0.5 - 1e-9, e * 0.5 + 1e-9)
        v["pos"] = v["pos"] + dt * v["v_w"]
        v["pos"][:, 2], _ = reflect(v["pos"][:, 2], None, 0.24, 0.30)
        rpy = v["rpy"] + dt * v["w_w"]
        w = v["w_w"].copy()
        for c in (0, 1):
            rpy[:, c], w[:, c] = reflect(rpy[:, c], w[:, c], -e * 0.15 - 1e-9, e * 0.15 + 1e-9)
        rpy[:, 2] = (rpy[:, 2] + np.pi) % (2 * np.pi) - np.pi
        v["rpy"], v["w_w"] = rpy, w
        q, qd = v["q"] + dt * v["qd"], v["qd"].copy()
        for j, (mid, half) in enumerate(((0.0, 0.2), (0.8, 0.2), (-1.6, 0.3))):
            q[:, j::3], qd[:, j::3] = reflect(q[:, j::3], qd[:, j::3], mid - half, mid + half)
        v["q"], v["qd"] = q, qd
        v["phase0"] = (v["phase0"] + dt / (0.5 / 0.6)) % 1.0
        out.append(_assemble(v))
    return out


def to_soa(a):
    """[n,k] robot-major -> [k,n] field-major contiguous (robot index fastest: coalesced device loads)."""
    return np.ascontiguousarray(np.asarray(a).T)


def make_frontend_batch(n, seed=0xFE, tick=0):
    """Synthetic per-tick inputs of the MPC front-end (layout: include/qrgpu.h fe_in / fe_state), AoS [n][64] and [n][8].

    A trot-like gait generator state: diagonal leg pairs half a cycle apart, duty factor 0.6, legs past the duty
    factor swinging; a few robots carry an EARLY_CONTACT / lost-contact leg, yaw near +-pi and backwards commands so
    every branch of SetupCommand / Run is taken somewhere in the batch."""
    rng = np.random.default_rng(seed)
    fe = np.zeros((n, 64), f32)
    st = np.zeros((n, 8), f32)
    fe[:, 0] = 0.28 + 0.02 * rng.standard_normal(n)                     # des height
    fe[:, 1] = 0.02 * rng.standard_normal(n)
    fe[:, 2] = 0.05 * rng.standard_normal(n)
    fe[:, 3] = rng.uniform(-1.5, 2.5, n)                                # x vel cmd (clipped to [-1, 2] by the filter state)
    fe[:, 4] = rng.uniform(-0.8, 0.8, n)
    fe[:, 5] = rng.uniform(-1.0, 1.0, n)
    fe[:, 6:8] = rng.uniform(-5, 5, (n, 2))
    fe[:, 8] = 0.28 + 0.03 * rng.standard_normal(n)
    yaw = rng.uniform(-np.pi, np.pi, n)
    yaw[: n // 8] = np.sign(yaw[: n // 8]) * rng.uniform(3.0, np.pi, n // 8)      # near the +-pi seam
    rpy = np.stack([0.05 * rng.standard_normal(n), 0.05 * rng.standard_normal(n), yaw], 1)
    fe[:, 9] = yaw
    fe[:, 10:14] = _quat_from_rpy(rpy)
    hip = np.array([[0.18, -0.13], [0.18, 0.13], [-0.18, -0.13], [-0.18, 0.13]])
    phase0 = rng.uniform(0, 1, n)
    duty = 0.6
    for leg in range(4):
        ph = (phase0 + (0.0 if leg in (0, 3) else 0.5)) % 1.0
        c, s = np.cos(yaw), np.sin(yaw)
        fx = fe[:, 6] + c * hip[leg, 0] - s * hip[leg, 1] + 0.03 * rng.standard_normal(n)
        fy = fe[:, 7] + s * hip[leg, 0] + c * hip[leg, 1] + 0.03 * rng.standard_normal(n)
        fe[:, 14 + 3 * leg] = fx; fe[:, 15 + 3 * leg] = fy; fe[:, 16 + 3 * leg] = 0.01 * rng.standard_normal(n)
        fe[:, 26 + 3 * leg] = fx + 0.1 * rng.standard_normal(n); fe[:, 27 + 3 * leg] = fy + 0.1 * rng.standard_normal(n)
        stance = ph < duty
        fe[:, 42 + leg] = ph
        fe[:, 46 + leg] = duty
        fe[:, 50 + leg] = np.where(stance, ph / duty, (ph - duty) / (1 - duty))
        des = np.where(stance, 1, 0)                                     # LegState: SWING 0, STANCE 1
        leg_state = des.copy()
        early = (~stance) & (rng.uniform(0, 1, n) < 0.1)
        leg_state[early] = 2                                            # EARLY_CONTACT
        lost = stance & (rng.uniform(0, 1, n) < 0.05)
        leg_state[lost] = 3                                             # LOSE_CONTACT
        fe[:, 54 + leg] = des
        fe[:, 58 + leg] = leg_state
        fe[:, 38 + leg] = ((leg_state == 1) | (leg_state == 2)).astype(f32)
    fe[:, 62:64] = fe[:, 6:8] + 0.05 * rng.standard_normal((n, 2))
    st[:, 0] = rng.uniform(-1.2, 2.2, n); st[:, 1] = rng.uniform(-0.7, 0.7, n); st[:, 2] = rng.uniform(-1, 1, n)
    st[:, 3] = yaw + 0.02 * rng.standard_normal(n)
    st[:, 4:6] = fe[:, 6:8] + 0.2 * rng.standard_normal((n, 2))        # some beyond the +-0.1 clip of UpdateMPC
    st[:, 6] = fe[:, 0]
    st[:, 7] = tick + rng.integers(0, 120, n)                          # both sides of the <50 / %15 cadence
    return fe, st


def vmc_cfg(robot="a1", acc_weight=(1., 1., 1., 10., 10., 1.), reg_weight=1e-4, friction=0.5, fmin_ratio=0.01, fmax_ratio=10.0):
    """Packed force-balance QP parameters (qrgpu_vmc_setup): mass, total_inertia[9] (Eigen column-major map of the YAML list),
    acc_weight[6] (stance_leg_controller.yaml), regWeight, frictionCoef, fMinRatio, fMaxRatio (qr_qp_torque_optimizer.h:144-153)."""
    r = ROBOTS[robot]
    I = np.diag(np.asarray(r["inertia"], f32)).reshape(-1, order="F")
    return np.array([r["mass"], *I, *acc_weight, reg_weight, friction, fmin_ratio, fmax_ratio], dtype=f32)


def make_vmc_batch(n, robot="a1", seed=0xB2, sloped=0.0, excite=1.0):
    """Synthetic per-tick inputs of the force-balance QP, AoS [n][37] (layout: include/qrgpu.h vmc_in) plus joint angles [n][12].

    Stance patterns: all four, trot pairs, three legs and a single leg; desired accelerations up to `excite` x (6 m/s^2, 20 rad/s^2)
    so that friction and force-window rows bind on part of the batch.  `sloped` > 0 gives a fraction of robots a pitched control frame
    (Rcb != I, tilted normal and gravity) as the non-PLANE terrain branch builds it."""
    rng = np.random.default_rng(seed)
    r = ROBOTS[robot]
    q = np.tile(np.array([0.0, 0.9, -1.8], f32), (n, 4)) + 0.25 * rng.standard_normal((n, 12)).astype(f32)
    vin = np.zeros((n, 37), f32)
    hip_off = np.array([[0.1805, -0.047, 0], [0.1805, 0.047, 0], [-0.1805, -0.047, 0], [-0.1805, 0.047, 0]], f32)
    # foot positions in the base frame from the leg kinematics (qr_robot.cpp:127-146)
    for leg in range(4):
        tab, thip, tknee = q[:, 3 * leg], q[:, 3 * leg + 1], q[:, 3 * leg + 2]
        sh = r["hip_l"] * (1.0 if leg % 2 == 1 else -1.0)
        ld = np.sqrt(r["upper_l"] ** 2 + r["lower_l"] ** 2 + 2 * r["upper_l"] * r["lower_l"] * np.cos(tknee))
        eff = thip + tknee / 2
        ox, oz, oy = -ld * np.sin(eff), -ld * np.cos(eff), sh
        vin[:, 3 * leg + 0] = ox + hip_off[leg, 0]
        vin[:, 3 * leg + 1] = np.cos(tab) * oy - np.sin(tab) * oz + hip_off[leg, 1]
        vin[:, 3 * leg + 2] = np.sin(tab) * oy + np.cos(tab) * oz + hip_off[leg, 2]
    scale = np.array([6, 6, 6, 20, 20, 20], f32) * excite
    vin[:, 12:18] = scale * rng.uniform(-1, 1, (n, 6)).astype(f32)
    pat = rng.integers(0, 10, n)
    contacts = np.ones((n, 4), f32)
    contacts[pat == 0] = (1, 0, 0, 1); contacts[pat == 1] = (0, 1, 1, 0)
    contacts[pat == 2] = (1, 0, 0, 1); contacts[pat == 3] = (0, 1, 1, 0)
    contacts[pat == 4] = (0, 1, 1, 1); contacts[pat == 5] = (1, 1, 0, 1)
    contacts[pat == 6] = (1, 0, 0, 0)
    vin[:, 18:22] = contacts
    vin[:, 22:31] = np.eye(3, dtype=f32).reshape(-1)
    vin[:, 31:34] = (0, 0, 9.8)
    vin[:, 34:37] = (0, 0, 1)
    ns = int(sloped * n)
    if ns:
        pitch = rng.uniform(-0.4, 0.4, ns)
        for i in range(ns):
            c, s = np.cos(pitch[i]), np.sin(pitch[i])
            Rc = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], f32)              # control frame pitched about y
            vin[i, 22:31] = Rc.T.reshape(-1)
            vin[i, 31:34] = Rc.T @ np.array([0, 0, 9.8], f32)
            vin[i, 34:37] = (-s, 0, c)
    return vin, q


def make_vmc_world_batch(n, robot="a1", seed=0xB3, excite=1.0):
    """Inputs of the world-frame overload of the force-balance QP (qr_qp_torque_optimizer.cpp:304-398): vmc_in as make_vmc_batch with
    Rcb = rotMat of a random base attitude, gvec = (0,0,9.8), normal = e_z; plus per-leg ratios [n][8] = fMinRatio[4], fMaxRatio[4] --
    the trot values (0.01, 10) for most robots, the walk mode's loading / unloading values (0.001, 10 * phase) for the rest
    (qr_torque_stance_leg_controller.cpp:128-152).  -> vin [n][37], q [n][12], ratio [n][8]"""
    rng = np.random.default_rng(seed)
    vin, q = make_vmc_batch(n, robot, seed=seed + 1, sloped=0.0, excite=excite)
    rpy = np.stack([0.3 * rng.uniform(-1, 1, n), 0.3 * rng.uniform(-1, 1, n), rng.uniform(-np.pi, np.pi, n)], 1)
    R = _rot_body_to_world(rpy)
    vin[:, 22:31] = R.reshape(n, 9).astype(f32)
    vin[:, 31:34] = (0, 0, 9.8)
    vin[:, 34:37] = (0, 0, 1)
    ratio = np.zeros((n, 8), f32)
    ratio[:, 0:4] = 0.01; ratio[:, 4:8] = 10.0
    walk = rng.random(n) < 0.4
    ph = rng.uniform(0.001, 1.0, (n, 4)).astype(f32)
    ratio[walk, 0:4] = 0.001
    ratio[walk, 4:8] = np.where(rng.random((int(walk.sum()), 4)) < 0.5, 10.0, 10.0 * ph[walk])
    return vin, q, ratio


def estimator_cfg(robot="a1", time_step=0.002, accelerometer_variance=0.1, sensor_variance=0.1, window=120, body_height=0.28):
    """Packed estimator parameters: leg lengths, robot->timeStep, the three user_parameters.yaml values, hip offsets[12], robot->bodyHeight."""
    r = ROBOTS[robot]
    return np.array([r["hip_l"], r["upper_l"], r["lower_l"], time_step, accelerometer_variance, sensor_variance, window,
                     *np.asarray(r["hip_offset"], f32).reshape(-1), body_height], dtype=f32)


def make_estimator_sequence(n, ticks, seed=0xE5, dt_ms=2):
    """Synthetic sensor streams for n robots over `ticks` control ticks: [ticks][n][54] floats + [ticks][n] uint32 millisecond stamps.
    Layout per tick (include/qrgpu.h est_in): baseAccInBaseFrame[3], baseLinearAcceleration[3], quat_wxyz[4], rpyRate[3], footContact[4],
    q[12], dq[12], desiredLegState[4], groundOrientationMat[9].  A slow body motion with noise, trot contacts, a flight phase (no
    contact) for some robots, a pitched ground frame for every fifth robot."""
    rng = np.random.default_rng(seed)
    x = np.zeros((ticks, n, 54), f32)
    gpitch = np.where(np.arange(n) % 5 == 2, 0.2, 0.0)
    gmat = np.zeros((n, 9), f32)
    gmat[:, 0] = np.cos(gpitch); gmat[:, 2] = np.sin(gpitch); gmat[:, 4] = 1; gmat[:, 6] = -np.sin(gpitch); gmat[:, 8] = np.cos(gpitch)
    stamp = np.zeros((ticks, n), np.uint32)
    t0 = rng.integers(1, 5000, n)
    phase0 = rng.uniform(0, 1, n)
    yaw0 = rng.uniform(-np.pi, np.pi, n)
    qn = np.tile(np.array([0.0, 0.9, -1.8], f32), (n, 4))
    for k in range(ticks):
        t = k * dt_ms * 1e-3
        rpy = np.stack([0.05 * np.sin(3 * t + phase0), 0.05 * np.cos(2 * t + phase0), yaw0 + 0.3 * t], 1)
        x[k, :, 6:10] = _quat_from_rpy(rpy)
        x[k, :, 10:13] = np.stack([0.15 * np.cos(3 * t + phase0), -0.1 * np.sin(2 * t + phase0), 0.3 + 0 * phase0], 1) + 0.02 * rng.standard_normal((n, 3))
        acc = np.stack([0.5 * np.sin(5 * t + phase0), 0.3 * np.cos(4 * t + phase0), 9.81 + 0.4 * np.sin(7 * t + phase0)], 1)
        x[k, :, 0:3] = acc + 0.1 * rng.standard_normal((n, 3))
        x[k, :, 3:6] = acc - np.array([0, 0, 9.81]) + 0.1 * rng.standard_normal((n, 3))
        ph = (phase0 + t / 0.5) % 1.0
        c = np.stack([ph < 0.6, (ph + 0.5) % 1 < 0.6, (ph + 0.5) % 1 < 0.6, ph < 0.6], 1).astype(f32)
        c[(np.arange(n) % 7 == 3) & (k % 40 > 30)] = 0                   # a flight phase: the filter falls back on its own estimate
        x[k, :, 13:17] = c
        x[k, :, 41:45] = c                                               # desiredLegState: STANCE = 1, SWING = 0
        x[k, :, 45:54] = gmat
        x[k, :, 17:29] = qn + 0.2 * np.sin(6 * t + phase0)[:, None] * np.array([0.3, 1, -1] * 4, f32) + 0.01 * rng.standard_normal((n, 12))
        x[k, :, 29:41] = 1.5 * np.cos(6 * t + phase0)[:, None] * np.array([0.3, 1, -1] * 4, f32) + 0.05 * rng.standard_normal((n, 12))
        stamp[k] = t0 + k * dt_ms + (rng.uniform(0, 1, n) < 0.05)        # an occasional late sample
    return x, stamp


def make_swing_batch(n, robot="a1", seed=0x5E):
    """Synthetic inputs of the swing-leg target kernel, AoS [n][58] (layout: include/qrgpu.h swing_in): trot pairs swinging at random
    phases (a few at exactly 0 and 1 and slightly beyond), lift-off points under the hips, footholds a step ahead."""
    rng = np.random.default_rng(seed)
    r = ROBOTS[robot]
    x = np.zeros((n, 58), f32)
    pair = rng.integers(0, 4, n)
    flags = np.zeros((n, 4), f32)
    flags[pair == 0] = (1, 0, 0, 1); flags[pair == 1] = (0, 1, 1, 0); flags[pair == 2] = (0, 0, 0, 0); flags[pair == 3] = (1, 0, 0, 0)
    x[:, 0:4] = flags
    ph = rng.uniform(0, 1, (n, 4)).astype(f32)
    ph[: n // 16] = 0.0; ph[n // 16: n // 8] = 1.0; ph[n // 8: n // 8 + 4] = 1.0005
    x[:, 4:8] = ph
    x[:, 8:12] = rng.uniform(0.15, 0.3, (n, 4))
    hip = np.asarray(r["hip_offset"], f32)
    side = np.array([-1, 1, -1, 1], f32) * r["hip_l"]
    for leg in range(4):
        base = hip[leg] + np.array([0, side[leg], -0.28], f32)
        x[:, 12 + 3 * leg:15 + 3 * leg] = base + 0.03 * rng.standard_normal((n, 3))
        x[:, 24 + 3 * leg:27 + 3 * leg] = base + np.array([0.08, 0, 0], f32) + 0.03 * rng.standard_normal((n, 3))
    x[:, 36:39] = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), 0.28 + 0.02 * rng.standard_normal(n)], 1)
    rpy = np.stack([0.05 * rng.standard_normal(n), 0.05 * rng.standard_normal(n), rng.uniform(-np.pi, np.pi, n)], 1)
    x[:, 39:43] = _quat_from_rpy(rpy)
    x[:, 43:46] = rng.uniform(-0.5, 0.5, (n, 3))
    x[:, 46:58] = np.tile(np.array([0.0, 0.9, -1.8], f32), (n, 4))
    x[-1, 24:27] = (2.0, 0.0, -0.3)                      # an unreachable foothold: NaN angles fall back to the current ones
    x[-1, 0] = 1; x[-1, 4] = 0.9
    return x


def swing_velocity_cfg(robot="a1", stance_duration=0.3, swing_kp=(0.03, 0.03, 0.03), desired_height=0.27, foot_clearance=0.01, com_offset=(0.0, 0.0, 0.0)):
    """Packed float32[20] for the velocity-mode swing action: GetDefaultHipPosition() + comOffset [12], stanceDuration[4] (trot: 0.3),
    swingKp.trot[3], desiredHeight - footClearance (user_parameters.yaml)."""
    r = ROBOTS[robot]
    hp = np.asarray(r["default_hip_position"], f32).reshape(4, 3) + np.asarray(com_offset, f32)
    return np.array([*hp.reshape(12), *([stance_duration] * 4), *swing_kp, desired_height - foot_clearance], dtype=f32)


def make_swing_velocity_batch(n, robot="a1", seed=0x5F):
    """Synthetic inputs of the velocity-mode swing kernel, AoS [n][53] (layout: include/qrgpu.h swing_vel_in)."""
    rng = np.random.default_rng(seed)
    r = ROBOTS[robot]
    x = np.zeros((n, 53), f32)
    pair = rng.integers(0, 4, n)
    flags = np.zeros((n, 4), f32)
    flags[pair == 0] = (1, 0, 0, 1); flags[pair == 1] = (0, 1, 1, 0); flags[pair == 2] = (0, 0, 0, 0); flags[pair == 3] = (1, 1, 1, 1)
    x[:, 0:4] = flags
    ph = rng.uniform(0, 1, (n, 4)).astype(f32)
    ph[: n // 16] = 0.0; ph[n // 16: n // 8] = 1.0; ph[n // 8: n // 8 + 8] = 0.5
    x[:, 4:8] = ph
    hip = np.asarray(r["hip_offset"], f32)
    side = np.array([-1, 1, -1, 1], f32) * r["hip_l"]
    for leg in range(4):
        base = hip[leg] + np.array([0, side[leg], -0.26], f32)
        x[:, 8 + 3 * leg:11 + 3 * leg] = base + 0.03 * rng.standard_normal((n, 3))
    x[:, 20:23] = rng.uniform(-0.6, 0.6, (n, 3)); x[:, 22] *= 0.2
    x[:, 23] = rng.uniform(-1, 1, n)
    x[:, 24:27] = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.3, 0.3, n), np.zeros(n)], 1)
    x[:, 27] = rng.uniform(-1, 1, n)
    rpy = np.stack([0.08 * rng.standard_normal(n), 0.08 * rng.standard_normal(n), rng.uniform(-np.pi, np.pi, n)], 1)
    q = _quat_from_rpy(rpy)
    x[:, 37:41] = q
    # dR = baseRInControlFrame = groundRMat' * baseRMat with the control frame at the base's heading: the base's roll / pitch remain
    rp = rpy.copy(); rp[:, 2] = 0
    qq = _quat_from_rpy(rp).astype(np.float64)
    e0, e1, e2, e3 = qq.T
    x[:, 28:37] = np.stack([1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2), 2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3),
                            2 * (e2 * e3 - e0 * e1), 2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)], 1)
    x[:, 41:53] = np.tile(np.array([0.0, 0.9, -1.8], f32), (n, 4))
    x[-1, 8:11] = (3.0, 0.0, -0.3); x[-1, 0] = 1; x[-1, 4] = 0.0       # an unreachable lift-off point at phase 0: NaN angles fall back to the current ones
    return x


def foothold_cfg(robot="a1", swing_kp=(0.16, 0.16, 0.16), foot_clearance=0.01):
    """Packed float32[29]: hip_offset[12], default_hip_position[12], hip_l, swing_kp[3], foot_clearance
    (robot_params.hip_offset / default_hip_positions, user_parameters.yaml swingKp.advanced_trot / footClearance)."""
    r = ROBOTS[robot]
    return np.array([*np.asarray(r["hip_offset"], f32).reshape(12), *np.asarray(r["default_hip_position"], f32).reshape(12), r["hip_l"],
                     *swing_kp, foot_clearance], dtype=f32)


def make_foothold_batch(n, robot="a1", seed=0xF0):
    """Synthetic inputs of the foothold kernel, AoS [n][46] (layout: include/qrgpu.h fh_in): trotting leg states with a few early
    contacts and lost contacts, some robots holding their schedule (allowSwitchLegState = 0 on a leg), commands up to the clip."""
    rng = np.random.default_rng(seed)
    r = ROBOTS[robot]
    x = np.zeros((n, 46), f32)
    pair = rng.integers(0, 3, n)
    st = np.ones((n, 4), f32)
    st[pair == 0] = (0, 1, 1, 0); st[pair == 1] = (1, 0, 0, 1)
    odd = rng.random((n, 4))
    st = np.where(odd < 0.03, 2, np.where(odd > 0.97, 3, st)).astype(f32)          # EARLY_CONTACT / LOSE_CONTACT
    x[:, 0:4] = st
    x[:, 4:8] = (rng.random((n, 4)) > 0.1)
    x[:, 8:12] = rng.uniform(0.0, 0.25, (n, 4))
    x[:, 12:16] = rng.uniform(0.0, 1.0, (n, 4))
    x[:, 16:19] = np.stack([rng.uniform(-0.5, 1.5, n), rng.uniform(-0.5, 0.5, n), np.zeros(n)], 1)
    x[: n // 8, 16] = 3.0                                                        # beyond the 0.2 m clip
    x[:, 19] = rng.uniform(-1.5, 1.5, n)
    x[:, 20] = 0.28 + 0.02 * rng.standard_normal(n)
    hip = np.asarray(r["hip_offset"], f32)
    side = np.array([-1, 1, -1, 1], f32) * r["hip_l"]
    for leg in range(4):
        x[:, 21 + 3 * leg:24 + 3 * leg] = hip[leg] + np.array([0, side[leg], -0.27], f32) + 0.04 * rng.standard_normal((n, 3))
    rpy = np.stack([0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(n), rng.uniform(-np.pi, np.pi, n)], 1)
    x[:, 33:37] = _quat_from_rpy(rpy)
    x[:, 37:40] = rpy
    x[:, 40:43] = np.stack([rng.uniform(-0.5, 1.5, n), rng.uniform(-0.5, 0.5, n), 0.1 * rng.standard_normal(n)], 1)
    x[:, 43:46] = rng.uniform(-1.0, 1.0, (n, 3))
    return x


def gait_cfg(stance_duration=0.5, duty_factor=0.6, initial_leg_phase=(0.5, 0.0, 0.0, 0.5), initial_leg_state=(1, 1, 1, 1),
             contact_detection_phase_threshold=0.5, wait_time=1.0, advanced_trot=True):
    """Packed open-loop gait parameters (config/a1_sim/openloop_gait_generator.yaml, gait "advanced_trot")."""
    return np.array([*([stance_duration] * 4), *([duty_factor] * 4), *initial_leg_phase, *initial_leg_state, contact_detection_phase_threshold,
                     wait_time, 1.0 if advanced_trot else 0.0], dtype=f32)


def walk_cfg(stance_duration=7.5, duty_factor=0.75, initial_leg_phase=(0.5, 0.0, 0.75, 0.25), initial_leg_state=(1, 1, 1, 1),
             contact_detection_phase_threshold=0.1, state_switch=(7, 6, 8, 5), state_ratio=(0.2, 0.3, 0.3, 0.2)):
    """Packed walk gait parameters (config/a1_sim/openloop_gait_generator.yaml, gait "walk"; SubLegState: load_force 5, unload_force 6,
    full_stance 7, true_swing 8): stance_duration[4], duty_factor[4], initial_leg_phase[4], initial_leg_state[4], threshold, n_states,
    state_switch[4], state_ratio[4]."""
    sw = list(state_switch) + [0] * (4 - len(state_switch)); sr = list(state_ratio) + [0.0] * (4 - len(state_ratio))
    return np.array([*([stance_duration] * 4), *([duty_factor] * 4), *initial_leg_phase, *initial_leg_state, contact_detection_phase_threshold,
                     len(state_switch), *sw, *sr], dtype=f32)


def make_gait_contacts(n, ticks, cfg19, seed=0x6A, dt=0.002):
    """Foot contact streams [ticks][n][4] that mostly follow the nominal trot, with late touch-downs (lost contact at the end of a
    swing: exercises the hold of Schedule()) and early touch-downs (EARLY_CONTACT) sprinkled in."""
    rng = np.random.default_rng(seed)
    full = cfg19[0] / cfg19[4]
    out = np.zeros((ticks, n, 4), f32)
    late = rng.uniform(0, 1, (n, 4)) < 0.15
    early = rng.uniform(0, 1, (n, 4)) < 0.15
    for k in range(ticks):
        t = k * dt
        for l in range(4):
            ph = np.fmod(cfg19[8 + l] * full + t, full) / full
            nominal = ph < cfg19[4]
            c = nominal.copy() if isinstance(nominal, np.ndarray) else np.full(n, nominal)
            c = np.where(late[:, l] & (ph < 0.08), False, c)                  # still in the air 8 % into the stance phase
            c = np.where(early[:, l] & (ph > 0.9), True, c)                   # touches down before the swing ends
            out[k, :, l] = c
    return out
