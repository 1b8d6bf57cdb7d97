// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// MPC front-end: command filtering, desired pose integration, reference
// trajectory and contact table, restating per control tick
//   MPCStanceLegController::SetupCommand   QS/controllers/mpc/qr_mpc_stance_leg_controller.cpp:158-204 (after UpdateDesCommand)
//   MPCStanceLegController::Run            :207-334   (without the SolveDenseMPC call)
//   MPCStanceLegController::UpdateMPC      :337-382   (trajectory only)
// Expressions keep the reference's float/double mix (double literals promote, results narrow on assignment).
#include "qr_oracle.h"

namespace qro {

static inline float clipf(float c, float lo, float hi) { return c < lo ? lo : (c > hi ? hi : c); }   // QI/utils/qr_algebra.h:57-65

void mpc_frontend(int horizon, int numHorizonL, float dt, float dtMPC, const float in[64], float st[8], float *traj, float *gait, float *wbc15 /*pBody vBody aBody rpy ori*/,
                  float contact_out[4], int *mpc_updated)
{
    // dt = 0.002, dtMPC = 0.06 in the reference (:43-44)
    const double kPI = 3.14159265358979323846, k2PI = 6.28318530718;   // M_PI, M_2PI (QI/utils/qr_ctypes.h:51)
    const float des_height = in[0], des_pitch = in[2];
    const float x_vel_cmd = in[3], y_vel_cmd = in[4], yaw_vel_cmd = in[5];
    const float *p = in + 6;
    const float yawCurrent = in[9];
    const float *pFoot = in + 14, *footTarget = in + 26, *contacts = in + 38, *phase = in + 42, *duty = in + 46, *nphase = in + 50;
    const float *desLegState = in + 54, *legState = in + 58, *startXY = in + 62;
    float xVelDes = st[0], yVelDes = st[1], yawTurnRate = st[2], yawDesTrue = st[3];
    float posDes[3] = {st[4], st[5], st[6]};
    const int iterationCounter = (int)st[7];

    // ---- SetupCommand (:163-203)
    float bodyHeight = des_height;
    const float x_filter(0.01f), y_filter(0.005f), yaw_filter(0.03f);
    xVelDes = xVelDes * (1 - x_filter) + x_vel_cmd * x_filter;
    yVelDes = yVelDes * (1 - y_filter) + y_vel_cmd * y_filter;
    yawTurnRate = yawTurnRate * (1 - yaw_filter) + yaw_vel_cmd * yaw_filter;
    xVelDes = clipf(xVelDes, -1.0f, 2.0f);
    yVelDes = clipf(yVelDes, -0.6f, 0.6f);
    yawDesTrue = yawDesTrue + dt * yawTurnRate;
    if (yawDesTrue >= kPI) yawDesTrue -= k2PI;
    else if (yawDesTrue <= -kPI) yawDesTrue += k2PI;
    if (yawCurrent > kPI / 2 && yawDesTrue < 0) yawDesTrue += k2PI;
    else if (yawCurrent < -kPI / 2 && yawDesTrue > 0) yawDesTrue -= k2PI;
    const float pitchDes = des_pitch;

    // ---- Run (:212-332)
    // baseRMat = quaternionToRotationMatrix(q)^T  (QS/robots/qr_robot.cpp:70)
    Q4<float> qq = {{in[10], in[11], in[12], in[13]}};
    M3<float> Rt = quaternionToRotationMatrix(qq);
    float Rm[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rm[i][j] = Rt[j][i];
    float vDesWorld[3];
    for (int i = 0; i < 3; ++i) vDesWorld[i] = Rm[i][0] * xVelDes + Rm[i][1] * yVelDes + Rm[i][2] * 0.f;
    posDes[0] += dt * vDesWorld[0];
    posDes[1] += dt * vDesWorld[1];
    posDes[2] += dt * 0.f;
    posDes[2] = 0.99 * (bodyHeight + (bodyHeight - p[2])) + 0.01 * posDes[2];
    float rpyComp[3] = {0.f, pitchDes, 0.f};
    for (int leg = 0; leg < 4; ++leg) {
        if ((int)desLegState[leg] == 0 /*SWING*/) {
            bodyHeight += 0.02 * std::sin(nphase[leg] * kPI);
            if (x_vel_cmd < -0.01) rpyComp[1] = rpyComp[1] - 0.1 * std::sin(nphase[leg] * kPI);
            break;
        }
    }
    float comDest[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int a = 0; a < 3; ++a) comDest[a] += (contacts[i] == 0.f) ? footTarget[3 * i + a] : pFoot[3 * i + a];
    for (int a = 0; a < 3; ++a) comDest[a] /= 4.f;
    float t = 1;
    const float dutyF = duty[0];
    if ((int)desLegState[0] == 0) t = phase[0] - dutyF;
    else if ((int)desLegState[1] == 0) t = phase[1] - dutyF;
    else if (phase[0] < phase[1]) t = phase[0] + (1 - dutyF);
    else t = phase[1] + (1 - dutyF);
    t *= 2.0f;
    for (int axis = 0; axis < 2; ++axis) posDes[axis] = (1 - t) * startXY[axis] + t * comDest[axis];
    const float dPhase = 1.0 / (numHorizonL * horizon);
    for (int i = 0; i < horizon; ++i)
        for (int j = 0; j < 4; ++j) {
            float ith = phase[j] + i * dPhase;
            while (ith > 1.0) ith -= 1.0;
            gait[4 * i + j] = (ith < duty[j] || (int)legState[j] == 2 /*EARLY_CONTACT*/) ? 1.f : 0.f;
        }
    for (int j = 0; j < 4; ++j) gait[j] = (contacts[j] != 0.f) ? 1.f : 0.f;

    // ---- UpdateMPC (:342-381)
    const int iterationsInaMPC = (int)std::round(dtMPC / dt);
    if (iterationCounter % (iterationsInaMPC / 2) == 0 || iterationCounter < 50) {
        const float xStart = clipf(posDes[0], p[0] - 0.1f, p[0] + 0.1f), yStart = clipf(posDes[1], p[1] - 0.1f, p[1] + 0.1f);
        posDes[0] = xStart; posDes[1] = yStart;
        const float ti[12] = {rpyComp[0], rpyComp[1], yawDesTrue, xStart, yStart, bodyHeight, 0.f, 0.f, yawTurnRate, vDesWorld[0], vDesWorld[1], 0.f};
        for (int i = 0; i < horizon; ++i) {
            for (int j = 0; j < 12; ++j) traj[12 * i + j] = ti[j];
            if (i == 0) traj[2] = yawDesTrue;
            else {
                traj[12 * i + 2] = traj[12 * (i - 1) + 2] + dtMPC * yawTurnRate;
                traj[12 * i + 3] = traj[12 * (i - 1) + 3] + dtMPC * vDesWorld[0];
                traj[12 * i + 4] = traj[12 * (i - 1) + 4] + dtMPC * vDesWorld[1];
            }
        }
        *mpc_updated = 1;
    } else *mpc_updated = 0;

    // ---- wbcData (:307-332)
    const float offx = Rm[0][0] * 0.018f + Rm[0][1] * 0.f + Rm[0][2] * 0.f, offy = Rm[1][0] * 0.018f + Rm[1][1] * 0.f + Rm[1][2] * 0.f;
    wbc15[0] = posDes[0] + offx; wbc15[1] = posDes[1] + offy; wbc15[2] = bodyHeight;
    wbc15[3] = vDesWorld[0]; wbc15[4] = vDesWorld[1]; wbc15[5] = 0.f;
    wbc15[6] = wbc15[7] = wbc15[8] = 0.f;
    wbc15[9] = rpyComp[0]; wbc15[10] = rpyComp[1]; wbc15[11] = yawDesTrue;
    wbc15[12] = 0.f; wbc15[13] = 0.f; wbc15[14] = yawTurnRate;
    for (int j = 0; j < 4; ++j) contact_out[j] = (contacts[j] != 0.f) ? 1.f : 0.f;
    st[0] = xVelDes; st[1] = yVelDes; st[2] = yawTurnRate; st[3] = yawDesTrue;
    st[4] = posDes[0]; st[5] = posDes[1]; st[6] = posDes[2];
    st[7] = (float)(iterationCounter + 1);
}

}  // namespace qro

extern "C" void qro_mpc_frontend(int horizon, int numHorizonL, float dt, float dtMPC, const float *in64, float *st8, float *traj, float *gait, float *wbc15,
                                 float *contact4, int *mpc_updated)
{
    qro::mpc_frontend(horizon, numHorizonL, dt, dtMPC, in64, st8, traj, gait, wbc15, contact4, mpc_updated);
}
