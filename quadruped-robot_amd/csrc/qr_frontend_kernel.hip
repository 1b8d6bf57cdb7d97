// ============================================================================
// MPC front-end, one thread per (robot, horizon step): command filter, desired-pose integration,
// contact table and reference trajectory of
//   MPCStanceLegController::SetupCommand / Run / UpdateMPC
//   (quadruped/src/controllers/mpc/qr_mpc_stance_leg_controller.cpp:158-204, 207-334, 337-382)
// for n robots whose gait-generator / estimator outputs are device resident.  Pure streaming work:
// (64 + 2*8 + 16h + 19 + 1) floats per robot, SoA so every load and store is a coalesced 256 B row
// segment per wave; bound by HBM, no LDS, no cross-lane traffic.
//
// The float/double mix of each expression is the reference's (double literals promote, assignment narrows),
// and contraction is off, so results are bit-identical to the CPU restatement except where std::sin's last
// bit differs between libm and the device library.
// ============================================================================
#include <hip/hip_runtime.h>
#include "qr_device_types.h"

namespace qrgpu {

#pragma clang fp contract(off)
__device__ __forceinline__ float fe_clip(float c, float lo, float hi) { return c < lo ? lo : (c > hi ? hi : c); }

// blockDim = (64 robots, horizon): thread (x, k) recomputes the short scalar prefix of robot x and writes row k of its contact table
// and reference trajectory, so that the h-long store loops of one robot run side by side; thread k = 0 also writes the
// controller memory, the WBC rows and the re-plan flag.  The barrier separates every read of fe_state from its rewrite.
__global__ void __launch_bounds__(1024) qr_frontend_kernel(int n, int horizon, int numHorizonL, float dt, float dtMPC, const float *__restrict__ fin,
                                                           float *__restrict__ fst, float *__restrict__ g_traj, float *__restrict__ g_gait,
                                                           float *__restrict__ g_cmd, int *__restrict__ g_updated)
{
#pragma clang fp contract(off)
    const int krow = threadIdx.y;
    const int i_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i_raw < n;
    const int i = live ? i_raw : n - 1;                 // out-of-range threads shadow the last robot (no stores) so that all reach the barrier
    const size_t N = (size_t)n;
#define FIN(f) fin[(size_t)(f) * N + i]
    const double kPI = 3.14159265358979323846, k2PI = 6.28318530718;        // M_PI, M_2PI (utils/qr_ctypes.h:51)
    float bodyHeight = FIN(0);
    const float pitchDes = FIN(2);
    const float x_vel_cmd = FIN(3), y_vel_cmd = FIN(4), yaw_vel_cmd = FIN(5);
    const float px = FIN(6), py = FIN(7), pz = FIN(8), yawCurrent = FIN(9);
    float xVelDes = fst[0 * N + i], yVelDes = fst[1 * N + i], yawTurnRate = fst[2 * N + i], yawDesTrue = fst[3 * N + i];
    float posx = fst[4 * N + i], posy = fst[5 * N + i], posz = fst[6 * N + i];
    const int iterationCounter = (int)fst[7 * N + i];
    __syncthreads();

    // SetupCommand (:163-203)
    const float x_filter = 0.01f, y_filter = 0.005f, yaw_filter = 0.03f;
    xVelDes = xVelDes * (1 - x_filter) + x_vel_cmd * x_filter;
    yVelDes = yVelDes * (1 - y_filter) + y_vel_cmd * y_filter;
    yawTurnRate = yawTurnRate * (1 - yaw_filter) + yaw_vel_cmd * yaw_filter;
    xVelDes = fe_clip(xVelDes, -1.0f, 2.0f);
    yVelDes = fe_clip(yVelDes, -0.6f, 0.6f);
    yawDesTrue = yawDesTrue + dt * yawTurnRate;
    if ((double)yawDesTrue >= kPI) yawDesTrue = (float)((double)yawDesTrue - k2PI);
    else if ((double)yawDesTrue <= -kPI) yawDesTrue = (float)((double)yawDesTrue + k2PI);
    if ((double)yawCurrent > kPI / 2 && yawDesTrue < 0) yawDesTrue = (float)((double)yawDesTrue + k2PI);
    else if ((double)yawCurrent < -kPI / 2 && yawDesTrue > 0) yawDesTrue = (float)((double)yawDesTrue - k2PI);

    // Run (:212-303).  baseRMat = body -> world rotation of the (w,x,y,z) quaternion.
    const float e0 = FIN(10), e1 = FIN(11), e2 = FIN(12), e3 = FIN(13);
    const float R00 = 1 - 2 * (e2 * e2 + e3 * e3), R01 = 2 * (e1 * e2 - e0 * e3);
    const float R10 = 2 * (e1 * e2 + e0 * e3), R11 = 1 - 2 * (e1 * e1 + e3 * e3);
    const float R02 = 2 * (e1 * e3 + e0 * e2), R12 = 2 * (e2 * e3 - e0 * e1);
    const float vwx = R00 * xVelDes + R01 * yVelDes + R02 * 0.f;
    const float vwy = R10 * xVelDes + R11 * yVelDes + R12 * 0.f;
    posx += dt * vwx;
    posy += dt * vwy;
    posz += dt * 0.f;
    posz = (float)(0.99 * (double)(bodyHeight + (bodyHeight - pz)) + 0.01 * (double)posz);
    float rpy1 = pitchDes;
    int swing0 = -1;
#pragma unroll
    for (int leg = 3; leg >= 0; --leg) if ((int)FIN(54 + leg) == 0) swing0 = leg;     // first SWING leg (LegState::SWING == 0)
    if (swing0 >= 0) {
        const double s = sin((double)FIN(50 + swing0) * kPI);
        bodyHeight = (float)((double)bodyHeight + 0.02 * s);
        if ((double)x_vel_cmd < -0.01) rpy1 = (float)((double)rpy1 - 0.1 * s);
    }
    float cdx = 0.f, cdy = 0.f;
    float ct[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        ct[l] = FIN(38 + l);
        const int base = (ct[l] == 0.f) ? 26 : 14;
        cdx += FIN(base + 3 * l);
        cdy += FIN(base + 3 * l + 1);
    }
    cdx /= 4.f; cdy /= 4.f;
    const float ph0 = FIN(42), ph1 = FIN(43), dutyF = FIN(46);
    float t;
    if ((int)FIN(54) == 0) t = ph0 - dutyF;
    else if ((int)FIN(55) == 0) t = ph1 - dutyF;
    else if (ph0 < ph1) t = ph0 + (1 - dutyF);
    else t = ph1 + (1 - dutyF);
    t *= 2.0f;
    posx = (1 - t) * FIN(62) + t * cdx;
    posy = (1 - t) * FIN(63) + t * cdy;

    const float dPhase = (float)(1.0 / (double)(numHorizonL * horizon));
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v;
            if (krow == 0) v = (ct[j] != 0.f) ? 1.f : 0.f;                  // row 0 = measured contacts (:301-303)
            else {
                float ith = FIN(42 + j) + krow * dPhase;
                while ((double)ith > 1.0) ith = (float)((double)ith - 1.0);
                v = (ith < FIN(46 + j) || (int)FIN(58 + j) == 2) ? 1.f : 0.f;    // LegState::EARLY_CONTACT stays in the table
            }
            g_gait[(size_t)(4 * krow + j) * N + i] = v;
        }
    }

    // UpdateMPC (:342-381): re-plan twice per MPC period and on each of the first 50 ticks
    const int iterationsInaMPC = (int)roundf(dtMPC / dt);
    const int period = iterationsInaMPC / 2 > 0 ? iterationsInaMPC / 2 : 1;
    const bool upd = (iterationCounter % period == 0) || iterationCounter < 50;
    if (upd) {
        posx = fe_clip(posx, px - 0.1f, px + 0.1f);
        posy = fe_clip(posy, py - 0.1f, py + 0.1f);
        if (live) {
            float yaw = yawDesTrue, x = posx, y = posy;
            for (int k = 0; k < krow; ++k) { yaw = yaw + dtMPC * yawTurnRate; x = x + dtMPC * vwx; y = y + dtMPC * vwy; }   // the reference's running sums (:372-374)
            float *tr = g_traj + (size_t)(12 * krow) * N + i;
            tr[0] = 0.f; tr[N] = rpy1; tr[2 * N] = yaw; tr[3 * N] = x; tr[4 * N] = y; tr[5 * N] = bodyHeight;
            tr[6 * N] = 0.f; tr[7 * N] = 0.f; tr[8 * N] = yawTurnRate; tr[9 * N] = vwx; tr[10 * N] = vwy; tr[11 * N] = 0.f;
        }
    }
    if (!live || krow != 0) return;
    if (g_updated) g_updated[i] = upd ? 1 : 0;

    // wbcData (:307-332)
    if (g_cmd) {
        g_cmd[0 * N + i] = posx + (R00 * 0.018f + R01 * 0.f + R02 * 0.f);
        g_cmd[1 * N + i] = posy + (R10 * 0.018f + R11 * 0.f + R12 * 0.f);
        g_cmd[2 * N + i] = bodyHeight;
        g_cmd[3 * N + i] = vwx; g_cmd[4 * N + i] = vwy; g_cmd[5 * N + i] = 0.f;
        g_cmd[6 * N + i] = 0.f; g_cmd[7 * N + i] = 0.f; g_cmd[8 * N + i] = 0.f;
        g_cmd[9 * N + i] = 0.f; g_cmd[10 * N + i] = rpy1; g_cmd[11 * N + i] = yawDesTrue;
        g_cmd[12 * N + i] = 0.f; g_cmd[13 * N + i] = 0.f; g_cmd[14 * N + i] = yawTurnRate;
#pragma unroll
        for (int l = 0; l < 4; ++l) g_cmd[(size_t)(63 + l) * N + i] = (ct[l] != 0.f) ? 1.f : 0.f;
    }
    fst[0 * N + i] = xVelDes; fst[1 * N + i] = yVelDes; fst[2 * N + i] = yawTurnRate; fst[3 * N + i] = yawDesTrue;
    fst[4 * N + i] = posx; fst[5 * N + i] = posy; fst[6 * N + i] = posz;
    fst[7 * N + i] = (float)(iterationCounter + 1);
#undef FIN
}

}  // namespace qrgpu
