// ============================================================================
// Header-only C++ adapters: the reference's own entry points, bodies replaced by
// calls into libqrgpu.so.  Drop-in plan:
//
//   * quadruped/src/controllers/mpc/qr_mpc_interface.cpp keeps its three public
//     functions (declared at quadruped/include/quadruped/controllers/mpc/qr_mpc_interface.h:157,200,215);
//     their bodies become the ones below (namespace Quadruped).  MPCStanceLegController
//     (qr_mpc_stance_leg_controller.cpp:90,399,404) is untouched.
//   * qrWbcLocomotionController<T>::Run (quadruped/src/controllers/wbc/qr_wbc_locomotion_controller.cpp:108-135)
//     becomes qrgpu_adapters::WbcRun below; ctor/UpdateLegCMD/cadence are untouched.
//
// The templates only need the handful of members the reference types already have
// (operator[] / data() on Eigen vectors, the qrRobot getters, qrWbcCtrlData fields), so the
// header compiles against the real Eigen/ROS types at an integration site and against the
// stubs of tests/stubs/ in this repository (Eigen and ROS are not installed here).
// ============================================================================
#pragma once
#include <cstdio>
#include <cstring>

#include "qrgpu.h"

namespace qrgpu_adapters {

// One process-wide context, mirroring the reference's file-static MPC state
// (qr_mpc_interface.cpp:35-104: one robot per process, not re-entrant).
struct Global {
    qrgpu_ctx *ctx = nullptr;
    int horizon = 0;
    bool has_solved = false;          // has_solved, qr_mpc_interface.cpp:41
    double q_soln[12] = {0};          // only indices 0..11 are ever read (qr_mpc_stance_leg_controller.cpp:404)
    int status = 0;
    float prev_ori_vel[3] = {0, 0, 0};
};
inline Global &global()
{
    static Global g;
    if (!g.ctx) {
        if (qrgpu_create(0, 1, QRGPU_MAX_HORIZON, &g.ctx) != QRGPU_OK) {
            std::fprintf(stderr, "[qrgpu] no usable gfx950 device: the MPC/WBC path has no CPU fallback\n");
            g.ctx = nullptr;
        }
    }
    return g;
}

}  // namespace qrgpu_adapters

namespace Quadruped {

// void SetupProblem(double dt, int horizon, double frictionCoeff, double fMax, double totalMass,
//                   float *inertia, float *weight, float alpha)        -- qr_mpc_interface.h:157
inline void SetupProblem(double dt, int horizon, double frictionCoeff, double fMax, double totalMass,
                         float *inertia, float *weight, float alpha)
{
    std::printf("SetupProblem: f_max = %f, mass = %f, horizon = %d\n", fMax, totalMass, horizon);   // as the reference (:162)
    auto &g = qrgpu_adapters::global();
    if (!g.ctx) return;
    g.horizon = horizon;
    g.has_solved = false;
    int rc = qrgpu_mpc_setup(g.ctx, 0, (float)dt, horizon, (float)frictionCoeff, (float)fMax, (float)totalMass, inertia, weight, alpha);
    if (rc != QRGPU_OK) std::fprintf(stderr, "[qrgpu] mpc_setup failed (%d): %s\n", rc, qrgpu_last_error(g.ctx));
}

// void SolveMPCKernel(Vec3<float>& p, Vec3<float>& v, Quat<float>& q, Vec3<float>& w,
//                     Eigen::Matrix<float,3,4>& r, Vec3<float>& rpy, float* state_trajectory, float* gait)   -- :200
// Quat<float> is (w, x, y, z) (qr_mpc_interface.cpp:344-347); r.data() is column-major 3x4.
template <class V3, class Q4, class M34>
inline void SolveMPCKernel(V3 &p, V3 &v, Q4 &q, V3 &w, M34 &r, V3 &rpy, float *state_trajectory, float *gait)
{
    auto &g = qrgpu_adapters::global();
    if (!g.ctx) return;
    const float pp[3] = {p[0], p[1], p[2]}, vv[3] = {v[0], v[1], v[2]}, ww[3] = {w[0], w[1], w[2]};
    const float qq[4] = {q[0], q[1], q[2], q[3]}, rr[3] = {rpy[0], rpy[1], rpy[2]};
    int rc = qrgpu_mpc_solve1(g.ctx, 0, pp, vv, qq, ww, r.data(), rr, state_trajectory, gait, nullptr, g.q_soln, nullptr, &g.status);
    if (rc != QRGPU_OK || ((unsigned)g.status & QRGPU_ST_FLAG_MASK)) std::printf("failed to solve!\n");        // the reference's only error channel (:440-442)
    g.has_solved = true;
}

// double GetMPCSolution(int index)                                                       -- :215
inline double GetMPCSolution(int index)
{
    auto &g = qrgpu_adapters::global();
    if (!g.has_solved || index < 0 || index >= 12) return 0.0;
    return g.q_soln[index];
}

}  // namespace Quadruped

namespace qrgpu_adapters {

// Body of qrWbcLocomotionController<T>::Run's compute branch (UpdateModel + ContactTaskUpdate +
// FindConfiguration + MakeTorque, qr_wbc_locomotion_controller.cpp:111-127).  Robot: qrRobot-like getters;
// WbcData: qrWbcCtrlData; writes jointTorqueCmd / desiredJPos / desiredJVel (anything indexable).
template <class Robot, class WbcData, class VecOut>
inline int WbcRun(Robot *robot, const WbcData *d, VecOut &jointTorqueCmd, VecOut &desiredJPos, VecOut &desiredJVel)
{
    auto &g = global();
    if (!g.ctx) return QRGPU_ERR_NO_DEVICE;
    float st[37], cmd[67];
    auto quat = robot->GetBaseOrientation();          // :141-146
    auto pos = robot->GetBasePosition();
    auto vb = robot->GetBaseVelocityInBaseFrame();
    auto q = robot->GetMotorAngles();
    auto dq = robot->GetMotorVelocities();
    auto om = robot->GetBaseRollPitchYawRate();
    for (int i = 0; i < 4; ++i) st[i] = quat[i];
    for (int i = 0; i < 3; ++i) { st[4 + i] = pos[i]; st[7 + i] = om[i]; st[10 + i] = vb[i]; }
    for (int i = 0; i < 12; ++i) { st[13 + i] = q[i]; st[25 + i] = dq[i]; }
    for (int i = 0; i < 3; ++i) {
        cmd[i] = d->pBody_des[i]; cmd[3 + i] = d->vBody_des[i]; cmd[6 + i] = d->aBody_des[i];
        cmd[9 + i] = d->pBody_RPY_des[i]; cmd[12 + i] = d->vBody_Ori_des[i];
    }
    for (int l = 0; l < 4; ++l) {
        for (int i = 0; i < 3; ++i) {
            cmd[15 + 3 * l + i] = d->pFoot_des[l][i]; cmd[27 + 3 * l + i] = d->vFoot_des[l][i];
            cmd[39 + 3 * l + i] = d->aFoot_des[l][i]; cmd[51 + 3 * l + i] = d->Fr_des[l][i];
        }
        cmd[63 + l] = d->contact_state[l] ? 1.f : 0.f;
    }
    float tau[12], qd[12], qdd[12];
    int status = 0;
    int rc = qrgpu_wbc_run1(g.ctx, 0, st, cmd, g.prev_ori_vel, tau, qd, qdd, &status);
    if (rc != QRGPU_OK) return rc;
    for (int i = 0; i < 12; ++i) { jointTorqueCmd[i] = tau[i]; desiredJPos[i] = qd[i]; desiredJVel[i] = qdd[i]; }
    return status;
}

// BuildDynamicModel's YAML-dependent values + the controller gains (qrgpu_wbc_setup); call once from the
// qrWbcLocomotionController constructor.
inline int WbcSetup(float hip_l, float upper_l, float lower_l)
{
    auto &g = global();
    if (!g.ctx) return QRGPU_ERR_NO_DEVICE;
    qrgpu_model_desc d;
    qrgpu_model_desc_default(&d);
    d.hip_l = hip_l; d.upper_l = upper_l; d.lower_l = lower_l;
    return qrgpu_wbc_setup(g.ctx, 0, &d);
}

// Body of Quadruped::ComputeContactForce, control-frame overload (quadruped/src/controllers/balance_controller/
// qr_qp_torque_optimizer.cpp:190-301), for the call TorqueStanceLegController::GetAction makes (:500).  The caller keeps the terrain
// branch of :214-223 and hands over what it produced: Rcb (row-major 3x3; identity on PLANE / PLUM_PILES), g.head(3), surfaceNormal.
// Robot: GetFootPositionsInBaseFrame() -> something with (row, col) access like Eigen::Matrix<float,3,4>.  Writes the 3x4 force matrix
// through out(row, col).  Per-type constants (mass, inertia, weights, ratios) go in once through VmcSetup.
template <class Robot, class V6, class V4, class M34>
inline int VmcContactForce(Robot *robot, const float Rcb[9], const float g3[3], const float normal[3], const V6 &desiredAcc, const V4 &contacts, M34 &out)
{
    auto &g = global();
    if (!g.ctx) return QRGPU_ERR_NO_DEVICE;
    float in[37];
    auto fp = robot->GetFootPositionsInBaseFrame();
    for (int l = 0; l < 4; ++l) for (int i = 0; i < 3; ++i) in[3 * l + i] = fp(i, l);
    for (int i = 0; i < 6; ++i) in[12 + i] = desiredAcc[i];
    for (int l = 0; l < 4; ++l) in[18 + l] = contacts[l] ? 1.f : 0.f;
    for (int i = 0; i < 9; ++i) in[22 + i] = Rcb[i];
    for (int i = 0; i < 3; ++i) { in[31 + i] = g3[i]; in[34 + i] = normal[i]; }
    float f[12];
    int status = 0;
    int rc = qrgpu_vmc_force1(g.ctx, 0, in, nullptr, f, nullptr, &status);
    if (rc != QRGPU_OK) return rc;
    for (int l = 0; l < 4; ++l) for (int i = 0; i < 3; ++i) out(i, l) = f[3 * l + i];
    return status;
}

// The world-frame overload, ComputeContactForce(robot, desiredAcc, contacts, accWeight, normal, tangent1, tangent2, fMinRatio, fMaxRatio, ...)
// (qr_qp_torque_optimizer.cpp:304-398) as TorqueStanceLegController::GetAction calls it with the identity's columns (:490-498):
// rotMat = quaternionToRotationMatrix(robot->GetBaseOrientation()).transpose(), row-major; V4 ratios are the controller's Vec4 members.
template <class Robot, class V6, class V4b, class V4min, class V4max, class M34>
inline int VmcContactForceWorld(Robot *robot, const float rotMat[9], const V6 &desiredAcc, const V4b &contacts, const V4min &fMinRatio, const V4max &fMaxRatio, M34 &out)
{
    auto &g = global();
    if (!g.ctx) return QRGPU_ERR_NO_DEVICE;
    float in[37], ratio[8];
    auto fp = robot->GetFootPositionsInBaseFrame();
    for (int l = 0; l < 4; ++l) for (int i = 0; i < 3; ++i) in[3 * l + i] = fp(i, l);
    for (int i = 0; i < 6; ++i) in[12 + i] = desiredAcc[i];
    for (int l = 0; l < 4; ++l) { in[18 + l] = contacts[l] ? 1.f : 0.f; ratio[l] = fMinRatio[l]; ratio[4 + l] = fMaxRatio[l]; }
    for (int i = 0; i < 9; ++i) in[22 + i] = rotMat[i];
    in[31] = 0.f; in[32] = 0.f; in[33] = 9.8f; in[34] = 0.f; in[35] = 0.f; in[36] = 1.f;
    float f[12];
    int status = 0;
    int rc = qrgpu_vmc_force_world1(g.ctx, 0, in, ratio, nullptr, f, nullptr, &status);
    if (rc != QRGPU_OK) return rc;
    for (int l = 0; l < 4; ++l) for (int i = 0; i < 3; ++i) out(i, l) = f[3 * l + i];
    return status;
}

inline int VmcSetup(const qrgpu_vmc_desc &d)
{
    auto &g = global();
    if (!g.ctx) return QRGPU_ERR_NO_DEVICE;
    return qrgpu_vmc_setup(g.ctx, 0, &d);
}

}  // namespace qrgpu_adapters
