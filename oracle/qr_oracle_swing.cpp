// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Swing-leg targets of the MPC/WBC locomotion mode (second part): restates, for LocomotionMode::ADVANCED_TROT on
// horizontal terrain (dR = robotBaseR = I, QS/controllers/qr_swing_leg_controller.cpp:271-276),
//   qrRaibertSwingLegController::GetAction, ADVANCED_TROT case           QS/controllers/qr_swing_leg_controller.cpp:362-398, 408-424
//   SwingFootTrajectory::ResetFootTrajectory / GenerateTrajectoryPoint    QI/controllers/qr_foot_trajectory_generator.h:368-372, QS/.../qr_foot_trajectory_generator.cpp:322-343
//   qrFootParabolaPatternGenerator::GenerateTrajectory (XYLinear_ZParabola) QS/controllers/qr_foot_trajectory_generator.cpp:187-215
//   qrQuadraticSpline::getPoint(t, mid, out)                              QS/utils/qr_geometry.cpp:157-190   (xd = xdd = 0 in the reference)
//   robotics::math::invertRigidTransform                                  QI/utils/qr_se3.h:442-449
//   qrRobot::ComputeMotorAnglesFromFootLocalPosition / FootPositionInHipFrameToJointAngle / ComputeMotorVelocityFromFootLocalVelocity
//                                                                         QS/robots/qr_robot.cpp:106-124, 200-219
// Eigen's Isometry / Quaternion arithmetic cannot be compiled here: "parity unpinned" at that boundary; pinned by properties
// (trajectory end points, apex height, IK o FK = identity) in tests/test_oracle_swing.py.
#include "qr_oracle.h"

namespace qro {

// in[58]: swing flag[4] (leg is in swingFootIds), footholdPlanner->phase[4], swingDuration[4], phaseSwitchFootGlobalPos[12] (3*leg+axis),
//         desiredFootholds[12] (base frame), basePosition[3], quat_wxyz[4], baseVInWorldFrame[3], motor angles q[12].
// out[72]: pFoot_des[12], vFoot_des[12], aFoot_des[12] (world), footTargetPositionsInWorldFrame[12], joint angle targets[12],
//          joint velocity targets[12]; entries of stance legs are left untouched (the caller's previous values).
void swing_targets(const LegGeom &geo, const float hip_offset[12], const float in[58], float out[72])
{
    const float *flag = in, *phase_in = in + 4, *swingDur = in + 8, *pswitch = in + 12, *foothold = in + 24, *bp = in + 36, *quat = in + 39, *bv = in + 43, *q = in + 46;
    Q4<float> qq = {{quat[0], quat[1], quat[2], quat[3]}};
    M3<float> Rt = quaternionToRotationMatrix(qq);           // world -> body; its transpose rotates body -> world
    auto to_world = [&](const float p[3], float o[3]) {      // invertRigidTransform(basePosition, quat, p): t + R p
        for (int i = 0; i < 3; ++i) o[i] = (Rt[0][i] * p[0] + Rt[1][i] * p[1] + Rt[2][i] * p[2]) + bp[i];
    };
    for (int leg = 0; leg < 4; ++leg) {
        if (flag[leg] == 0.f) continue;
        const float *tgt = foothold + 3 * leg, *st = pswitch + 3 * leg;
        const float phase = phase_in[leg];
        const float H = 0.1f;
        to_world(tgt, out + 36 + 3 * leg);                    // desiredStateCommand->footTargetPositionsInWorldFrame (:366-367)
        // XYLinear_ZParabola with duration 1, start = phaseSwitchFootGlobalPos, end = robotBaseR * target = target
        float pw[3] = {0, 0, 0}, vw[3] = {0, 0, 0}, aw[3] = {0, 0, 0};
        if (!(phase < 0.f - 1e-3) && !(phase >= 0.f + 1.f + 1e-3)) {
            pw[0] = (1 - phase) * st[0] + phase * tgt[0];
            pw[1] = (1 - phase) * st[1] + phase * tgt[1];
            const float mid = std::max(tgt[2], st[2]) + H;
            float dt = phase - 0.f;
            if (dt > 1.f) dt = 1.f;
            if (!(dt < 0.)) {
                const float mid_phase = 0.5;
                const float d1 = mid - st[2], d2 = tgt[2] - st[2];
                const float d3 = std::pow((double)mid_phase, 2) - mid_phase;
                const float ca = (d1 - d2 * mid_phase) / d3;
                const float cb = (d2 * std::pow((double)mid_phase, 2) - d1) / d3;
                const float cc = st[2];
                pw[2] = ca * std::pow((double)phase, 2) + cb * phase + cc;
            }
        }
        float pb[3] = {pw[0], pw[1], pw[2]}, vb[3] = {0, 0, 0};      // robotBaseR^T * (...) with robotBaseR = I
        if (phase < 1.0) for (int i = 0; i < 3; ++i) vb[i] = vw[i] / swingDur[leg];
        to_world(pb, out + 3 * leg);                                      // pFoot_des (:393)
        for (int i = 0; i < 3; ++i) { out[12 + 3 * leg + i] = bv[i] + vb[i]; out[24 + 3 * leg + i] = aw[i]; }      // (:394-395)
        // joint targets: IK of the base-frame point, J^-1 of the base-frame velocity (:408-411)
        const float sgn = ((leg + 1) % 2 == 0) ? 1.f : -1.f;             // pow(-1, leg + 1)
        const float sh = geo.hip_l * sgn;
        const float x = pb[0] - hip_offset[3 * leg], y = pb[1] - hip_offset[3 * leg + 1], z = pb[2] - hip_offset[3 * leg + 2];
        const float lu = geo.upper_l, ll = geo.lower_l;
        float tK = -std::acos(((x * x + y * y + z * z) - (sh * sh + lu * lu + ll * ll)) / (2 * ll * lu));
        const float l = std::sqrt(lu * lu + ll * ll + 2 * lu * ll * std::cos(tK));
        float tH = std::asin(-x / l) - tK / 2;
        const float c1 = sh * y - l * std::cos(tH + tK / 2) * z;
        const float s1 = l * std::cos(tH + tK / 2) * y + sh * z;
        float tA = std::atan2(s1, c1);
        float ang[3] = {tA, tH, tK};
        float J[9];
        analytical_leg_jacobian(geo, ang, leg, J);
        // dq = J^-1 v by cofactors
        const float det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
        const float id = 1.f / det;
        const float Ji[9] = {(J[4] * J[8] - J[5] * J[7]) * id, (J[2] * J[7] - J[1] * J[8]) * id, (J[1] * J[5] - J[2] * J[4]) * id,
                             (J[5] * J[6] - J[3] * J[8]) * id, (J[0] * J[8] - J[2] * J[6]) * id, (J[2] * J[3] - J[0] * J[5]) * id,
                             (J[3] * J[7] - J[4] * J[6]) * id, (J[1] * J[6] - J[0] * J[7]) * id, (J[0] * J[4] - J[1] * J[3]) * id};
        for (int i = 0; i < 3; ++i) {
            float a = ang[i];
            if (std::isnan(a)) a = q[3 * leg + i];                        // (:415-418) unreachable target: keep the current angle
            out[48 + 3 * leg + i] = a;
            out[60 + 3 * leg + i] = Ji[3 * i] * vb[0] + Ji[3 * i + 1] * vb[1] + Ji[3 * i + 2] * vb[2];
        }
    }
}

// Swing-leg action of the velocity mode (the trot of the VMC / force-balance path), QS/controllers/qr_swing_leg_controller.cpp:285-309 + 408-424:
// the Raibert target in the base frame from the hip's horizontal velocity (estimated base velocity + yaw rate x hip lever, through
// dR = baseRInControlFrame: the velocity mode runs on terrain type SLOPE, :271-276), XYLinear_ZParabola between the lift-off point and that
// target (height 0.1, duration 1) at the warped phase of SwingFootTrajectory::GenerateTrajectoryPoint(phaseModule = true)
// (QS/controllers/qr_foot_trajectory_generator.cpp:322-343), leg IK and J^-1 v.
// desc[20]: hipPositions + comOffset [12] (3*leg+axis), stanceDuration[4], swingKp[3] (user_parameters swingKp.trot), desiredHeight - footClearance
// in[53]: swing flag[4], normalizedPhase[4], phaseSwitchFootLocalPos[12], estimated base velocity (base frame)[3], yaw rate, desiredSpeed[3]
//         (stateDes 6..8), desiredTwistingSpeed (stateDes 11), dR[9] (baseRInControlFrame, row-major), quat_wxyz[4], motor angles[12]
// out[48]: footTargetPosition[12] (base frame), footPositionInBaseFrame[12], joint angle targets[12], joint velocity targets[12];
//          entries of legs that are not flagged are left untouched.
void swing_velocity_mode(const LegGeom &geo, const float hip_offset[12], const float desc[20], const float in[53], float out[48])
{
    const float *flag = in, *nphase = in + 4, *pswitch = in + 8, *bvel = in + 20, *sp = in + 24, *dR = in + 28, *quat = in + 37, *q = in + 41;
    const float yawDot = in[23], twist = in[27];
    Q4<float> qq = {{quat[0], quat[1], quat[2], quat[3]}};
    const M3<float> Rwb = quaternionToRotationMatrix(qq);       // = baseRMat^T: robotBaseR.transpose() * v = Rwb v
    for (int leg = 0; leg < 4; ++leg) {
        if (flag[leg] == 0.f) continue;
        const float ho[3] = {desc[3 * leg], desc[3 * leg + 1], desc[3 * leg + 2]};
        const float tw[3] = {-ho[1], ho[0], 0.f};
        float hv[3], hh[3], tgtv[3];
        for (int i = 0; i < 3; ++i) hv[i] = bvel[i] + yawDot * tw[i];
        for (int i = 0; i < 3; ++i) hh[i] = (dR[3 * i] * hv[0] + dR[3 * i + 1] * hv[1]) + dR[3 * i + 2] * hv[2];
        hh[2] = 0.f;
        for (int i = 0; i < 3; ++i) tgtv[i] = sp[i] + twist * tw[i];
        float u[3];
        for (int i = 0; i < 3; ++i) u[i] = hh[i] * desc[12 + leg] / 2.0f - desc[16 + i] * (tgtv[i] - hh[i]);
        const float dh[3] = {0.f, 0.f, desc[19]};
        float tgt[3];
        for (int i = 0; i < 3; ++i) {
            const float a = (dR[i] * u[0] + dR[3 + i] * u[1]) + dR[6 + i] * u[2];                    // dR^T u
            const float b = (Rwb[i][0] * dh[0] + Rwb[i][1] * dh[1]) + Rwb[i][2] * dh[2];            // robotBaseR^T desiredHeight
            const float off = (i < 2) ? ho[i] : 0.f;
            tgt[i] = (a + off) - b;
        }
        const float *st = pswitch + 3 * leg;
        // phase warp (phaseModule = true): double arithmetic on the float phase
        const float inputPhase = nphase[leg];
        float phase;
        if (inputPhase <= 0.5) phase = (float)(0.8 * std::sin(inputPhase * M_PI));
        else phase = (float)(0.8 + (inputPhase - 0.5) * 0.4);
        float pw[3] = {0, 0, 0}, vw[3] = {0, 0, 0};
        if (!(phase < 0.f - 1e-3) && !(phase >= 0.f + 1.f + 1e-3)) {
            pw[0] = (1 - phase) * st[0] + phase * tgt[0];
            pw[1] = (1 - phase) * st[1] + phase * tgt[1];
            const float mid = std::max(tgt[2], st[2]) + 0.1f;
            float dtp = phase - 0.f;
            if (dtp > 1.f) dtp = 1.f;
            if (!(dtp < 0.)) {
                const float mid_phase = 0.5;
                const float d1 = mid - st[2], d2 = tgt[2] - st[2];
                const float d3 = std::pow((double)mid_phase, 2) - mid_phase;
                const float ca = (d1 - d2 * mid_phase) / d3;
                const float cb = (d2 * std::pow((double)mid_phase, 2) - d1) / d3;
                const float cc = st[2];
                pw[2] = ca * std::pow((double)phase, 2) + cb * phase + cc;
            }
        }
        for (int i = 0; i < 3; ++i) { out[3 * leg + i] = tgt[i]; out[12 + 3 * leg + i] = pw[i]; }
        const float sgn = ((leg + 1) % 2 == 0) ? 1.f : -1.f;
        const float sh = geo.hip_l * sgn;
        const float x = pw[0] - hip_offset[3 * leg], y = pw[1] - hip_offset[3 * leg + 1], z = pw[2] - hip_offset[3 * leg + 2];
        const float lu = geo.upper_l, ll = geo.lower_l;
        float tK = -std::acos(((x * x + y * y + z * z) - (sh * sh + lu * lu + ll * ll)) / (2 * ll * lu));
        const float l = std::sqrt(lu * lu + ll * ll + 2 * lu * ll * std::cos(tK));
        float tH = std::asin(-x / l) - tK / 2;
        const float c1 = sh * y - l * std::cos(tH + tK / 2) * z;
        const float s1 = l * std::cos(tH + tK / 2) * y + sh * z;
        float tA = std::atan2(s1, c1);
        float ang[3] = {tA, tH, tK};
        float J[9];
        analytical_leg_jacobian(geo, ang, leg, J);
        const float det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
        const float id = 1.f / det;
        const float Ji[9] = {(J[4] * J[8] - J[5] * J[7]) * id, (J[2] * J[7] - J[1] * J[8]) * id, (J[1] * J[5] - J[2] * J[4]) * id,
                             (J[5] * J[6] - J[3] * J[8]) * id, (J[0] * J[8] - J[2] * J[6]) * id, (J[2] * J[3] - J[0] * J[5]) * id,
                             (J[3] * J[7] - J[4] * J[6]) * id, (J[1] * J[6] - J[0] * J[7]) * id, (J[0] * J[4] - J[1] * J[3]) * id};
        for (int i = 0; i < 3; ++i) {
            float a = ang[i];
            if (std::isnan(a)) a = q[3 * leg + i];
            out[24 + 3 * leg + i] = a;
            out[36 + 3 * leg + i] = Ji[3 * i] * vw[0] + Ji[3 * i + 1] * vw[1] + Ji[3 * i + 2] * vw[2];
        }
    }
}

}  // namespace qro

extern "C" void qro_swing_velocity(const float *geom3, const float *hip_offset12, const float *desc20, const float *in53, float *out48)
{
    qro::LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2];
    qro::swing_velocity_mode(g, hip_offset12, desc20, in53, out48);
}

extern "C" void qro_swing_targets(const float *geom3, const float *hip_offset12, const float *in58, float *out72)
{
    qro::LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2];
    qro::swing_targets(g, hip_offset12, in58, out72);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Swing-leg selection + Raibert-type foothold heuristic of the MPC/WBC mode: restates
//   qrRaibertSwingLegController::Update, default branch (which legs are in swingFootIds)     QS/controllers/qr_swing_leg_controller.cpp:211-236
//   qrFootholdPlanner::ComputeHeuristicFootHold                                               QS/planner/qr_foothold_planner.cpp:110-239
// on flat ground (groundRMat = I, so dR = baseRInControlFrame = baseRMat, QS/robots/qr_robot.cpp:70-71; the ground estimator is not built).
// One thing is NOT mirrored: for a backwards command (stateDes(6) < -0.01) the reference executes `footTargetPosition(0,2) -= 0.02;
// footTargetPosition(0,3) -= 0.02;` on a 3x1 vector (:222-225) -- an out-of-bounds access (an assertion in debug builds, a stray stack
// write otherwise); it is treated as a no-op here.  Eigen's fixed-size products cannot be compiled here: "parity unpinned" at that
// boundary, pinned by properties in tests/test_oracle_swing.py (like the swing targets above).
// desc[29]: hip_offset[12] (robot->hipOffset, 3*leg+axis), default_hip_position[12], hip_l, swing_kp[3], foot_clearance
// in[46]:   legState[4], allowSwitchLegState[4], swingTimeRemaining[4], normalizedPhase[4], desiredSpeed[3] (stateDes 6..8),
//           desiredTwistingSpeed (stateDes 11), stateDes(2), footPositionsInBaseFrame[12], quat_wxyz[4], rpy[3],
//           baseVelocityInBaseFrame[3], baseRollPitchYawRate[3]
// swing_in[58] (layout above): rows 0-3 (leg is in swingFootIds) are always written; for those legs rows 4-7 (planner phase) and
//           24-35 (desiredFootholds, base frame); the other legs keep their previous values, as the planner's members do.
namespace qro {
void footholds(const float desc[29], const float in[46], float swing_in[58])
{
    const float *abad = desc, *hipPos = desc + 12, hipLen = desc[24], *kp = desc + 25, clearance = desc[28];
    const float *legState = in, *allow = in + 4, *srem = in + 8, *nphase = in + 12, *vdes = in + 16, wdes = in[19], hdes = in[20];
    const float *footB = in + 21, *quat = in + 33, *rpy = in + 37, *vb = in + 40, *w = in + 43;
    Q4<float> qq = {{quat[0], quat[1], quat[2], quat[3]}};
    const M3<float> Rt = quaternionToRotationMatrix(qq);                      // world -> body; baseRMat = its transpose
    auto R = [&](int i, int j) { return Rt[j][i]; };                          // robotBaseR = dR
    const float side_sign[4] = {-1.f, 1.f, -1.f, 1.f};
    const float dh[3] = {0.f, 0.f, hdes - clearance};
    for (int leg = 0; leg < 4; ++leg) {
        const int st = (int)legState[leg];
        const bool skip = (st == 1 /*STANCE*/ && allow[leg] != 0.f) || st == 2 /*EARLY_CONTACT*/;
        swing_in[leg] = skip ? 0.f : 1.f;
        if (skip) continue;
        const float *ho = abad + 3 * leg;
        const float twist[3] = {-ho[1], ho[0], 0.f};
        const float cr[3] = {w[1] * ho[2] - w[2] * ho[1], w[2] * ho[0] - w[0] * ho[2], w[0] * ho[1] - w[1] * ho[0]};
        const float hv0[3] = {vb[0] + cr[0], vb[1] + cr[1], vb[2] + cr[2]};
        float hv[3];
        for (int i = 0; i < 3; ++i) hv[i] = (R(i, 0) * hv0[0] + R(i, 1) * hv0[1]) + R(i, 2) * hv0[2];
        hv[2] = 0.f;
        const float tv[3] = {vdes[0] + wdes * twist[0], vdes[1] + wdes * twist[1], vdes[2] + wdes * twist[2]};
        float ftp[3];
        float phase;
        if (allow[leg] == 0.f) {
            const float d[3] = {footB[3 * leg] - hipPos[3 * leg], footB[3 * leg + 1] - hipPos[3 * leg + 1], footB[3 * leg + 2] - hipPos[3 * leg + 2]};
            float t[3];
            for (int i = 0; i < 3; ++i) t[i] = (R(i, 0) * d[0] + R(i, 1) * d[1]) + R(i, 2) * d[2];
            if (t[1] > 0.01 + 0.00 * (-side_sign[leg])) t[1] -= 0.005; else if (t[1] < -0.01 + 0.00 * side_sign[leg]) t[1] += 0.005;
            t[2] -= 0.02;
            for (int i = 0; i < 3; ++i) ftp[i] = ((R(0, i) * t[0] + R(1, i) * t[1]) + R(2, i) * t[2]) + hipPos[3 * leg + i];
            phase = 1.0f;
        } else {
            const float s = srem[leg];
            float u[3];
            for (int i = 0; i < 3; ++i) u[i] = tv[i] * s - kp[i] * (tv[i] - hv[i]);
            float dP[3];
            for (int i = 0; i < 3; ++i) dP[i] = (R(0, i) * u[0] + R(1, i) * u[1]) + R(2, i) * u[2];
            const float th = 0.2f;
            dP[0] = dP[0] < -th ? -th : (dP[0] > th ? th : dP[0]);
            dP[1] = dP[1] < -th ? -th : (dP[1] > th ? th : dP[1]);
            dP[2] = 0;
            const float iy = hipLen * side_sign[leg];
            const float c = std::cos(rpy[0]), sn = std::sin(rpy[0]);                 // coordinateRotation(X, roll) * (0, iy, 0)
            const float rr[3] = {(0.f * 1.f + 0.f * iy) + 0.f * 0.f, (0.f * 0.f + c * iy) + sn * 0.f, (0.f * 0.f + -sn * iy) + c * 0.f};
            const float a[3] = {ho[0], ho[1], 0.f};
            for (int i = 0; i < 3; ++i) ftp[i] = (dP[i] + a[i]) + rr[i];
            for (int i = 0; i < 3; ++i) ftp[i] -= (R(0, i) * dh[0] + R(1, i) * dh[1]) + R(2, i) * dh[2];
            phase = nphase[leg];
        }
        swing_in[4 + leg] = phase;
        for (int i = 0; i < 3; ++i) swing_in[24 + 3 * leg + i] = ftp[i];
    }
}
}  // namespace qro

extern "C" void qro_footholds(const float *desc29, const float *in46, float *swing_in58) { qro::footholds(desc29, in46, swing_in58); }
