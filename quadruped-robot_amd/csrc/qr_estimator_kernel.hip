// ============================================================================
// Base velocity estimator + the leg kinematics it reads, one thread per robot (first part):
//   qrRobot::UpdateDataFlow              quadruped/src/robots/qr_robot.cpp:62-72, 187-197   (foot Jacobians, positions, velocities)
//   qrRobotVelocityEstimator::Update     quadruped/src/estimators/qr_robot_velocity_estimator.cpp:77-133
//   qrRobotPoseEstimator::Update         quadruped/src/estimators/qr_robot_pose_estimator.cpp:68-165   (stance-foot height, planar odometry)
//   qrMovingWindowFilter                 quadruped/include/quadruped/estimators/qr_moving_window_filter.hpp:150-186, 236-263
//   TinyEKF<3,3>                         quadruped/extern/TinyEKF/src/TinyEKF.h:103-124, tiny_ekf.c:17-93, 292-332
// A streaming kernel with per-robot memory: the Kalman state, the two Neumaier window sums and their ring buffers live in a
// [field][robot] array of doubles of which one tick touches 32 header fields and 6 ring slots.  Every operation of the filters is
// the reference's, in its order and type, contraction off: with equal foot kinematics the outputs are bit-identical to the CPU
// restatement (whose Kalman step is bit-identical to the reference's compiled TinyEKF); sinf / cosf of the leg kinematics differ
// from libm in the last bit, which is what the tolerance of tests/test_gpu_estimator.py covers.
// ============================================================================
#include <hip/hip_runtime.h>
#include "qr_device_types.h"
#include "qr_wave_helpers.h"

namespace qrgpu {

namespace {
#pragma clang fp contract(off)
__device__ __forceinline__ void mulmat3(const double *a, const double *b, double *c)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
#pragma unroll
            for (int l = 0; l < 3; ++l) acc += a[3 * i + l] * b[3 * l + j];
            c[3 * i + j] = acc;
        }
}
// Inverse of a symmetric positive-definite 3x3 (row-major) through its Cholesky factor, written out entry by entry.
// Returns 1 when a pivot is not positive.  The Kalman step is compared bit for bit with the reference's compiled filter
// (tests/test_oracle_estimator.py), whose inverse runs factor -> inverse of the factor -> product of the inverse factors; bit identity
// needs those roundings in that sequence, so every statement below is one IEEE operation (contraction off) in that dependency order,
// including the `0.0 -` / `0.0 +` starts of the accumulated sums.
__device__ __forceinline__ int spd3_inverse(const double *A, double *inv)
{
#pragma clang fp contract(off)
    // factor A = L L^T: d0..d2 the diagonal of L, l10 l20 l21 below it
    if (A[0] <= 0) return 1;
    const double d0 = sqrt(A[0]);
    const double l10 = A[1] / d0, l20 = A[2] / d0;
    const double s11 = A[4] - l10 * l10;
    if (s11 <= 0) return 1;
    const double d1 = sqrt(s11);
    const double l21 = (A[5] - l10 * l20) / d1;
    const double s22 = (A[8] - l21 * l21) - l20 * l20;
    if (s22 <= 0) return 1;
    const double d2 = sqrt(s22);
    // K = L^-1 (lower triangular), column by column
    const double k00 = 1 / d0;
    const double k10 = (0.0 - l10 * k00) / d1;
    const double k20 = ((0.0 - l20 * k00) - l21 * k10) / d2;
    const double k11 = 1 / d1;
    const double k21 = (0.0 - l21 * k11) / d2;
    const double k22 = 1 / d2;
    // A^-1 = K^T K, upper triangle then mirrored
    const double i00 = (k00 * k00 + k10 * k10) + k20 * k20;
    const double i01 = (0.0 + k10 * k11) + k20 * k21;
    const double i02 = 0.0 + k20 * k22;
    const double i11 = k11 * k11 + k21 * k21;
    const double i12 = 0.0 + k21 * k22;
    const double i22 = k22 * k22;
    inv[0] = i00; inv[1] = i01; inv[2] = i02;
    inv[3] = i01; inv[4] = i11; inv[5] = i12;
    inv[6] = i02; inv[7] = i12; inv[8] = i22;
    return 0;
}

// Neumaier update of (sum, corr) by v
template <typename T> __device__ __forceinline__ void neumaier(T &sum, T &corr, T v)
{
#pragma clang fp contract(off)
    const T ns = sum + v;
    const T as = sum < 0 ? -sum : sum, av = v < 0 ? -v : v;
    if (as >= av) corr += (sum - ns) + v;
    else corr += (v - ns) + sum;
    sum = ns;
}
}  // namespace

#define EST_LAST 0
#define EST_X 1
#define EST_P 4
#define EST_VB 13
#define EST_VSUM 16
#define EST_VCORR 19
#define EST_VCNT 22
#define EST_VHEAD 23
#define EST_ASUM 24
#define EST_ACORR 27
#define EST_ACNT 30
#define EST_AHEAD 31
#define EST_POSE 32          /* x, y, theta, absoluteHight */
#define EST_AWIN 36
#define EST_VWIN 96

__global__ void __launch_bounds__(64) qr_estimator_kernel(int n, EstimatorDesc D, const float *__restrict__ g_in, const unsigned *__restrict__ g_tick,
                                                          double *__restrict__ st, float *__restrict__ g_out)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define IN(f) g_in[(size_t)(f) * N + i]
#define ST(f) st[(size_t)(f) * N + i]
    // ---- UpdateDataFlow: leg kinematics (FootPositionInHipFrame :127-146, AnalyticalLegJacobian :148-172)
    float footP[12], footV[12];
#pragma unroll
    for (int leg = 0; leg < 4; ++leg) {
        const float t0 = IN(17 + 3 * leg), t1 = IN(18 + 3 * leg), t2 = IN(19 + 3 * leg);
        const float d0 = IN(29 + 3 * leg), d1 = IN(30 + 3 * leg), d2 = IN(31 + 3 * leg);
        const float sh = D.hip_l * ((leg & 1) ? 1.f : -1.f);
        const float lu = D.upper_l, ll = D.lower_l;
        const float legDist = sqrtf(lu * lu + ll * ll + 2 * lu * ll * cosf(t2));
        const float eff = t1 + t2 / 2;
        const float offX = -legDist * sinf(eff), offZ = -legDist * cosf(eff), offY = sh;
        footP[3 * leg + 0] = offX + D.hip_offset[3 * leg + 0];
        footP[3 * leg + 1] = cosf(t0) * offY - sinf(t0) * offZ + D.hip_offset[3 * leg + 1];
        footP[3 * leg + 2] = sinf(t0) * offY + cosf(t0) * offZ + D.hip_offset[3 * leg + 2];
        float J[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) leg_jacobian_column(j, t0, t1, t2, sh, lu, ll, J[0][j], J[1][j], J[2][j]);
#pragma unroll
        for (int r = 0; r < 3; ++r) footV[3 * leg + r] = J[r][0] * d0 + J[r][1] * d1 + J[r][2] * d2;
    }
    // ---- AccFilter: 3-vector window of 20 (:79-80)
    float facc[3];
    {
        const int cnt = (int)ST(EST_ACNT), head = (int)ST(EST_AHEAD);
        const int len = cnt >= 20 ? cnt - 1 : cnt;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float sum = (float)ST(EST_ASUM + a), corr = (float)ST(EST_ACORR + a);
            if (cnt >= 20) neumaier<float>(sum, corr, -(float)ST(EST_AWIN + 20 * a + head));
            const float v = IN(3 + a);
            neumaier<float>(sum, corr, v);
            ST(EST_AWIN + 20 * a + head) = (double)v;
            ST(EST_ASUM + a) = (double)sum; ST(EST_ACORR + a) = (double)corr;
            facc[a] = (sum + corr) / (float)(len + 1);
        }
        ST(EST_ACNT) = (double)(len + 1); ST(EST_AHEAD) = (double)((head + 1) % 20);
    }
    // ---- ComputeDeltaTime (:64-75)
    const unsigned tick = g_tick[i], last = (unsigned)ST(EST_LAST);
    float deltaTime;
    if ((double)last < 1e-5) deltaTime = D.time_step;
    else deltaTime = (float)((double)(tick - last) / 1000.);
    ST(EST_LAST) = (double)tick;
    // ---- attitude, calibrated acceleration, contact-leg observation (:85-103)
    const float e0 = IN(6), e1 = IN(7), e2 = IN(8), e3 = IN(9);
    float R[3][3];
    R[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); R[0][1] = 2 * (e1 * e2 - e0 * e3); R[0][2] = 2 * (e1 * e3 + e0 * e2);
    R[1][0] = 2 * (e1 * e2 + e0 * e3); R[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); R[1][2] = 2 * (e2 * e3 - e0 * e1);
    R[2][0] = 2 * (e1 * e3 - e0 * e2); R[2][1] = 2 * (e2 * e3 + e0 * e1); R[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
    const float s0 = IN(0), s1 = IN(1), s2 = IN(2);
    float cal[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) cal[r] = R[r][0] * s0 + R[r][1] * s1 + R[r][2] * s2;
    cal[2] = (float)((double)cal[2] - 9.81);
    const double deltaV[3] = {(double)(cal[0] * deltaTime), (double)(cal[1] * deltaTime), (double)(cal[2] * deltaTime)};
    const float wx = IN(10), wy = IN(11), wz = IN(12);
    float mean[3] = {0.f, 0.f, 0.f};
    int num = 0;
#pragma unroll
    for (int leg = 0; leg < 4; ++leg) {
        if (IN(13 + leg) == 0.f) continue;
        const float p0 = footP[3 * leg], p1 = footP[3 * leg + 1], p2 = footP[3 * leg + 2];
        const float cx = 0 * p0 + (-wz) * p1 + wy * p2;
        const float cy = wz * p0 + 0 * p1 + (-wx) * p2;
        const float cz = (-wy) * p0 + wx * p1 + 0 * p2;
        const float vB0 = footV[3 * leg] + cx, vB1 = footV[3 * leg + 1] + cy, vB2 = footV[3 * leg + 2] + cz;
#pragma unroll
        for (int r = 0; r < 3; ++r) mean[r] += (-R[r][0]) * vB0 + (-R[r][1]) * vB1 + (-R[r][2]) * vB2;
        ++num;
    }
    double z[3];
    if (num > 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { mean[r] /= (float)num; z[r] = (double)mean[r]; }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) z[r] = (double)(float)ST(EST_VB + r);
    }
    // ---- TinyEKF<3,3>::step with fx = x + deltaV, F = H = I (:104-109)
    double x[3], P[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) x[r] = ST(EST_X + r);
#pragma unroll
    for (int r = 0; r < 9; ++r) P[r] = ST(EST_P + r);
    {
        const double qv = (double)D.accelerometer_variance, rv = (double)D.sensor_variance;
        const double F[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double fx[3] = {x[0] + deltaV[0], x[1] + deltaV[1], x[2] + deltaV[2]};
        double tmp0[9], Pp[9], tmp1[9], tmp2[9], tmp3[9], tmp4[9], tmp5[3], G[9];
        mulmat3(F, P, tmp0);
        mulmat3(tmp0, F, Pp);                        // Ft = F
#pragma unroll
        for (int r = 0; r < 3; ++r) Pp[4 * r] += qv;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (r % 4 != 0) Pp[r] += 0.0;
        mulmat3(Pp, F, tmp1);                        // Ht = H = I
        mulmat3(F, Pp, tmp2);
        mulmat3(tmp2, F, tmp3);
#pragma unroll
        for (int r = 0; r < 3; ++r) tmp3[4 * r] += rv;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (r % 4 != 0) tmp3[r] += 0.0;
        if (!spd3_inverse(tmp3, tmp4)) {
            mulmat3(tmp1, tmp4, G);
#pragma unroll
            for (int r = 0; r < 3; ++r) tmp5[r] = z[r] - fx[r];
#pragma unroll
            for (int r = 0; r < 3; ++r) { double y = 0; for (int j = 0; j < 3; ++j) y += tmp5[j] * G[3 * r + j]; x[r] = fx[r] + y; }
            mulmat3(G, F, tmp0);
#pragma unroll
            for (int r = 0; r < 9; ++r) tmp0[r] = -tmp0[r];
#pragma unroll
            for (int r = 0; r < 3; ++r) tmp0[4 * r] += 1;
            mulmat3(tmp0, Pp, P);
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) ST(EST_X + r) = x[r];
#pragma unroll
    for (int r = 0; r < 9; ++r) ST(EST_P + r) = P[r];
    // ---- three scalar windows of W doubles on float(x) (:111-116)
    float vw[3];
    {
        const int W = D.window;
        const int cnt = (int)ST(EST_VCNT), head = (int)ST(EST_VHEAD);
        const int len = cnt >= W ? cnt - 1 : cnt;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double sum = ST(EST_VSUM + a), corr = ST(EST_VCORR + a);
            double *slot = st + (size_t)(EST_VWIN + W * a + head) * N + i;
            if (cnt >= W) neumaier<double>(sum, corr, -*slot);
            const double v = (double)(float)x[a];
            neumaier<double>(sum, corr, v);
            *slot = v;
            ST(EST_VSUM + a) = sum; ST(EST_VCORR + a) = corr;
            vw[a] = (float)((sum + corr) / (double)(len + 1));
        }
        ST(EST_VCNT) = (double)(len + 1); ST(EST_VHEAD) = (double)((head + 1) % W);
    }
    // ---- outputs (:117-132)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float vb = R[0][r] * vw[0] + R[1][r] * vw[1] + R[2][r] * vw[2];
        ST(EST_VB + r) = (double)vb;
        g_out[(size_t)r * N + i] = facc[r];
        g_out[(size_t)(3 + r) * N + i] = vw[r];
        g_out[(size_t)(6 + r) * N + i] = vb;
        g_out[(size_t)(9 + r) * N + i] = R[r][0] * wx + R[r][1] * wy + R[r][2] * wz;
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) { g_out[(size_t)(12 + r) * N + i] = footP[r]; g_out[(size_t)(24 + r) * N + i] = footV[r]; }
    // ---- qrRobotPoseEstimator::Update (:68-165): same ticks, hence the same deltaTime
    {
        int nct = 0;
        float hs = 0.f, hc = 0.f;
        const float g2 = IN(47), g5 = IN(50), g8 = IN(53);            // third column of groundOrientationMat = row 2 of its transpose
#pragma unroll
        for (int leg = 0; leg < 4; ++leg) {
            const bool st_ = (int)IN(41 + leg) == 1;                    // LegState::STANCE
            const float p0 = footP[3 * leg], p1 = footP[3 * leg + 1], p2 = footP[3 * leg + 2];
            float w[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) w[r] = R[r][0] * p0 + R[r][1] * p1 + R[r][2] * p2;
            const float cz = g2 * w[0] + g5 * w[1] + g8 * w[2];
            const float c = st_ ? 1.f : 0.f;
            hc += (-cz) * c;
            hs += (-w[2]) * c;
            nct += st_ ? 1 : 0;
        }
        const float height = nct ? hs / (float)nct : D.body_height;
        const float hctrl = nct ? hc / (float)nct : __builtin_nanf("");
        const float vX = (float)ST(EST_VB), vY = (float)ST(EST_VB + 1), vZ = (float)ST(EST_VB + 2);
        const float theta = (float)ST(EST_POSE + 2);
        const double ct = cos((double)theta), sn = sin((double)theta);
        const float deltaX = (float)(((double)vX * ct - (double)vY * sn) * (double)deltaTime);
        const float deltaY = (float)(((double)vX * sn + (double)vY * ct) * (double)deltaTime);
        const float px = (float)ST(EST_POSE) + deltaX, py = (float)ST(EST_POSE + 1) + deltaY;
        const float ah = (float)ST(EST_POSE + 3) + vZ * deltaTime;
        const float th = theta + wz * deltaTime;
        ST(EST_POSE) = (double)px; ST(EST_POSE + 1) = (double)py; ST(EST_POSE + 2) = (double)th; ST(EST_POSE + 3) = (double)ah;
        g_out[(size_t)36 * N + i] = px; g_out[(size_t)37 * N + i] = py; g_out[(size_t)38 * N + i] = height;
        g_out[(size_t)39 * N + i] = hctrl; g_out[(size_t)40 * N + i] = ah; g_out[(size_t)41 * N + i] = th;
    }
#undef IN
#undef ST
}

// Glue between the estimators and the tick, one thread per robot: what MPCStanceLegController::SolveDenseMPC (:385-399) and
// qrWbcLocomotionController::UpdateModel (:136-156) read from qrRobot / stateDataFlow, laid out as the tick's mpc_state[28] and
// fb_state[37].  foot2ComInWorldFrame = baseRMat * (footPositionsInBaseFrame - comOffset); rpy is an input (GetBaseRollPitchYaw).
__global__ void __launch_bounds__(64) qr_pack_state_kernel(int n, float c0, float c1, float c2, const float *__restrict__ g_in, const float *__restrict__ g_est,
                                                           const float *__restrict__ g_rpy, float *__restrict__ g_mpc, float *__restrict__ g_fb)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define IN(f) g_in[(size_t)(f) * N + i]
#define ES(f) g_est[(size_t)(f) * N + i]
    const float e0 = IN(6), e1 = IN(7), e2 = IN(8), e3 = IN(9);
    float R[3][3];
    R[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); R[0][1] = 2 * (e1 * e2 - e0 * e3); R[0][2] = 2 * (e1 * e3 + e0 * e2);
    R[1][0] = 2 * (e1 * e2 + e0 * e3); R[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); R[1][2] = 2 * (e2 * e3 - e0 * e1);
    R[2][0] = 2 * (e1 * e3 - e0 * e2); R[2][1] = 2 * (e2 * e3 + e0 * e1); R[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
    if (g_mpc) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            g_mpc[(size_t)r * N + i] = ES(36 + r);                 // basePosition
            g_mpc[(size_t)(3 + r) * N + i] = ES(3 + r);            // baseVInWorldFrame
            g_mpc[(size_t)(10 + r) * N + i] = ES(9 + r);           // baseWInWorldFrame
            g_mpc[(size_t)(25 + r) * N + i] = g_rpy[(size_t)r * N + i];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) g_mpc[(size_t)(6 + r) * N + i] = IN(6 + r);
#pragma unroll
        for (int leg = 0; leg < 4; ++leg) {
            const float p0 = ES(12 + 3 * leg) - c0, p1 = ES(13 + 3 * leg) - c1, p2 = ES(14 + 3 * leg) - c2;
#pragma unroll
            for (int r = 0; r < 3; ++r) g_mpc[(size_t)(13 + 3 * leg + r) * N + i] = R[r][0] * p0 + R[r][1] * p1 + R[r][2] * p2;
        }
    }
    if (g_fb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) g_fb[(size_t)r * N + i] = IN(6 + r);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            g_fb[(size_t)(4 + r) * N + i] = ES(36 + r);
            g_fb[(size_t)(7 + r) * N + i] = IN(10 + r);            // omegaBody = GetBaseRollPitchYawRate
            g_fb[(size_t)(10 + r) * N + i] = ES(6 + r);            // GetBaseVelocityInBaseFrame
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) { g_fb[(size_t)(13 + r) * N + i] = IN(17 + r); g_fb[(size_t)(25 + r) * N + i] = IN(29 + r); }
    }
#undef IN
#undef ES
}

// Swing-leg targets of the MPC/WBC mode (ADVANCED_TROT, horizontal terrain), one thread per robot:
//   qrRaibertSwingLegController::GetAction      quadruped/src/controllers/qr_swing_leg_controller.cpp:362-398, 408-424
//   qrFootParabolaPatternGenerator / qrQuadraticSpline   quadruped/src/controllers/qr_foot_trajectory_generator.cpp:187-215, quadruped/src/utils/qr_geometry.cpp:157-190
//   leg inverse kinematics                      quadruped/src/robots/qr_robot.cpp:106-124, 200-219
// Writes, for the legs flagged as swinging only: rows 15-50 of wbc_cmd (pFoot_des, vFoot_des, aFoot_des), the foot target in the
// world frame (an input of the MPC front-end) and the joint angle / velocity targets of the swing-leg position command.
__global__ void __launch_bounds__(64) qr_swing_kernel(int n, EstimatorDesc D, const float *__restrict__ g_in, float *__restrict__ g_cmd, float *__restrict__ g_tgt_world,
                                                      float *__restrict__ g_qdes)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define IN(f) g_in[(size_t)(f) * N + i]
    const float e0 = IN(39), e1 = IN(40), e2 = IN(41), e3 = IN(42);
    float R[3][3];                                                        // body -> world
    R[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); R[0][1] = 2 * (e1 * e2 - e0 * e3); R[0][2] = 2 * (e1 * e3 + e0 * e2);
    R[1][0] = 2 * (e1 * e2 + e0 * e3); R[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); R[1][2] = 2 * (e2 * e3 - e0 * e1);
    R[2][0] = 2 * (e1 * e3 - e0 * e2); R[2][1] = 2 * (e2 * e3 + e0 * e1); R[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
    const float bp[3] = {IN(36), IN(37), IN(38)}, bv[3] = {IN(43), IN(44), IN(45)};
#pragma unroll
    for (int leg = 0; leg < 4; ++leg) {
        if (IN(leg) == 0.f) continue;
        const float phase = IN(4 + leg), swingDur = IN(8 + leg);
        const float st[3] = {IN(12 + 3 * leg), IN(13 + 3 * leg), IN(14 + 3 * leg)}, tg[3] = {IN(24 + 3 * leg), IN(25 + 3 * leg), IN(26 + 3 * leg)};
        if (g_tgt_world) {
#pragma unroll
            for (int r = 0; r < 3; ++r) g_tgt_world[(size_t)(3 * leg + r) * N + i] = (R[r][0] * tg[0] + R[r][1] * tg[1] + R[r][2] * tg[2]) + bp[r];
        }
        float pw[3] = {0.f, 0.f, 0.f};
        if (!((double)phase < 0.0 - 1e-3) && !((double)phase >= 0.0 + 1.0 + 1e-3)) {
            pw[0] = (1 - phase) * st[0] + phase * tg[0];
            pw[1] = (1 - phase) * st[1] + phase * tg[1];
            const float mid = (tg[2] > st[2] ? tg[2] : st[2]) + 0.1f;
            if (!(phase < 0.f)) {
                const float d1 = mid - st[2], d2 = tg[2] - st[2];
                const float d3 = (float)(0.25 - 0.5);
                const float ca = (d1 - d2 * 0.5f) / d3;
                const float cb = (float)(((double)d2 * 0.25 - (double)d1) / (double)d3);
                pw[2] = (float)((double)ca * ((double)phase * (double)phase) + (double)(cb * phase) + (double)st[2]);
            }
        }
        float vb[3] = {0.f, 0.f, 0.f};
        if ((double)phase < 1.0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) vb[r] = 0.f / swingDur;
        }
        if (g_cmd) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                g_cmd[(size_t)(15 + 3 * leg + r) * N + i] = (R[r][0] * pw[0] + R[r][1] * pw[1] + R[r][2] * pw[2]) + bp[r];
                g_cmd[(size_t)(27 + 3 * leg + r) * N + i] = bv[r] + vb[r];
                g_cmd[(size_t)(39 + 3 * leg + r) * N + i] = 0.f;
            }
        }
        if (g_qdes) {
            const float sh = D.hip_l * ((leg & 1) ? 1.f : -1.f);
            const float x = pw[0] - D.hip_offset[3 * leg], y = pw[1] - D.hip_offset[3 * leg + 1], z = pw[2] - D.hip_offset[3 * leg + 2];
            const float lu = D.upper_l, ll = D.lower_l;
            const float tK = -acosf(((x * x + y * y + z * z) - (sh * sh + lu * lu + ll * ll)) / (2 * ll * lu));
            const float l = sqrtf(lu * lu + ll * ll + 2 * lu * ll * cosf(tK));
            const float tH = asinf(-x / l) - tK / 2;
            const float c1 = sh * y - l * cosf(tH + tK / 2) * z;
            const float s1 = l * cosf(tH + tK / 2) * y + sh * z;
            const float tA = atan2f(s1, c1);
            const float ang[3] = {tA, tH, tK};
            float J[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) leg_jacobian_column(j, tA, tH, tK, sh, lu, ll, J[0][j], J[1][j], J[2][j]);
            const float det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
            const float id = 1.f / det;
            const float Ji[3][3] = {{(J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id, (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id, (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id},
                                    {(J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id, (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id, (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id},
                                    {(J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id, (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id, (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id}};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float a = ang[r];
                if (a != a) a = IN(46 + 3 * leg + r);                     // unreachable target: keep the current angle (:415-418)
                g_qdes[(size_t)(3 * leg + r) * N + i] = a;
                g_qdes[(size_t)(12 + 3 * leg + r) * N + i] = Ji[r][0] * vb[0] + Ji[r][1] * vb[1] + Ji[r][2] * vb[2];
            }
        }
    }
#undef IN
}

// Open-loop gait generator, one thread per robot: qrOpenLoopGaitGenerator::Update + Schedule
// (quadruped/src/gait/qr_openloop_gait_generator.cpp:126-207, 210-249; legs with a non-zero duty factor).  State [52][n] floats:
// resetTime, lastTime, cumDt, gaitCycle, then per leg cur, last, desired, legState, allow, firstSwing, firstStance, phaseInFullCycle,
// normalizedPhase, contactStartPhase, swingTimeRemaining, (spare); `fresh` != 0 applies Reset(0) first.  Plain float arithmetic and
// fmodf (exact): bit-identical to the CPU restatement.
__global__ void __launch_bounds__(64) qr_gait_kernel(int n, GaitDesc D, float currentTime, int stop, int fresh, const float *__restrict__ g_contact,
                                                     float *__restrict__ st, float *__restrict__ g_out, float *__restrict__ g_fe)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define ST(f) st[(size_t)(f) * N + i]
    float reset_time, last_time, cum_dt, gait_cycle;
    int cur[4], last[4], desired[4], leg[4], allow[4], fsw[4], fst[4];
    float phase[4], nphase[4], csp[4], srem[4];
    if (fresh) {
        reset_time = last_time = cum_dt = gait_cycle = 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l) { cur[l] = last[l] = desired[l] = leg[l] = D.initial_leg_state[l]; allow[l] = 1; fsw[l] = fst[l] = 0; phase[l] = nphase[l] = csp[l] = srem[l] = 0.f; }
    } else {
        reset_time = ST(0); last_time = ST(1); cum_dt = ST(2); gait_cycle = ST(3);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            cur[l] = (int)ST(4 + l); last[l] = (int)ST(8 + l); desired[l] = (int)ST(12 + l); leg[l] = (int)ST(16 + l); allow[l] = (int)ST(20 + l);
            fsw[l] = (int)ST(24 + l); fst[l] = (int)ST(28 + l); phase[l] = ST(32 + l); nphase[l] = ST(36 + l); csp[l] = ST(40 + l); srem[l] = ST(44 + l);
        }
    }
    float full[4], swingDur[4], ct[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) { full[l] = D.stance_duration[l] / D.duty_factor[l]; swingDur[l] = full[l] - D.stance_duration[l]; ct[l] = g_contact[(size_t)l * N + i]; }
    float tsr = currentTime;
    if (reset_time + full[0] < tsr) { reset_time = tsr; gait_cycle += 1.f; }
    tsr -= reset_time;
#pragma unroll
    for (int l = 0; l < 4; ++l) allow[l] = 1;
    if (D.advanced_trot) {
#pragma unroll
        for (int l = 0; l < 4; ++l) if (cur[l] == 0 && desired[l] == 1 && ct[l] == 0.f) allow[l] = 0;
        if (allow[0] + allow[1] + allow[2] + allow[3] < 4) {
            const float dt_ = currentTime - last_time;
            cum_dt += dt_;
            if (cum_dt > D.wait_time) {
#pragma unroll
                for (int l = 0; l < 4; ++l) allow[l] = 1;
            } else reset_time += dt_;
        } else cum_dt = 0.f;
    }
    const bool all_allowed = allow[0] + allow[1] + allow[2] + allow[3] == 4;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        if (!all_allowed) continue;
        if (!stop || (stop && last[l] == 0)) { last[l] = cur[l]; cur[l] = desired[l]; }
        const float augmented = D.initial_leg_phase[l] * full[l] + tsr;
        phase[l] = fmodf(augmented, full[l]) / full[l];
        const float ratio = D.duty_factor[l];
        if (phase[l] < ratio) { desired[l] = 1; nphase[l] = phase[l] / ratio; }
        else {
            desired[l] = 0;
            nphase[l] = (phase[l] - ratio) / (1 - ratio);
            if (cur[l] == 1) { fsw[l] = 1; csp[l] = 0.f; fst[l] = 0; srem[l] = swingDur[l]; }
            else { fsw[l] = 0; srem[l] = swingDur[l] * (1 - nphase[l]); }
        }
        if (leg[l] == 2 && desired[l] == 0) continue;
        leg[l] = desired[l];
        if (nphase[l] < D.contact_detection_phase_threshold) continue;
        if (leg[l] == 0 && ct[l] != 0.f) { leg[l] = 2; csp[l] = phase[l] - 1.0f; }
        if (cur[l] == 0 && (leg[l] == 2 || leg[l] == 1)) { fst[l] = 1; fsw[l] = 0; }
    }
    ST(0) = reset_time; ST(1) = currentTime; ST(2) = cum_dt; ST(3) = gait_cycle;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        ST(4 + l) = (float)cur[l]; ST(8 + l) = (float)last[l]; ST(12 + l) = (float)desired[l]; ST(16 + l) = (float)leg[l]; ST(20 + l) = (float)allow[l];
        ST(24 + l) = (float)fsw[l]; ST(28 + l) = (float)fst[l]; ST(32 + l) = phase[l]; ST(36 + l) = nphase[l]; ST(40 + l) = csp[l]; ST(44 + l) = srem[l];
        if (g_out) {
            g_out[(size_t)l * N + i] = phase[l]; g_out[(size_t)(4 + l) * N + i] = nphase[l]; g_out[(size_t)(8 + l) * N + i] = (float)desired[l];
            g_out[(size_t)(12 + l) * N + i] = (float)leg[l]; g_out[(size_t)(16 + l) * N + i] = (float)cur[l]; g_out[(size_t)(20 + l) * N + i] = srem[l];
        }
        if (g_fe) {          // rows 42-61 of the MPC front-end's input
            g_fe[(size_t)(42 + l) * N + i] = phase[l]; g_fe[(size_t)(46 + l) * N + i] = D.duty_factor[l]; g_fe[(size_t)(50 + l) * N + i] = nphase[l];
            g_fe[(size_t)(54 + l) * N + i] = (float)desired[l]; g_fe[(size_t)(58 + l) * N + i] = (float)leg[l];
        }
    }
#undef ST
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Which legs swing + where they should land: qrRaibertSwingLegController::Update (default branch, QS/controllers/
// qr_swing_leg_controller.cpp:211-236) and qrFootholdPlanner::ComputeHeuristicFootHold (QS/planner/qr_foothold_planner.cpp:110-239) on
// flat ground (dR = baseRMat).  One thread per robot; same operation order as oracle/qr_oracle_swing.cpp (footholds), contraction off.
// g_in [46][n] (include/qrgpu.h fh_in); rows 0-15 come from the gait kernel's arrays instead when those are given (legState and
// allowSwitchLegState are rows 16-23 of gait_state, normalizedPhase / swingTimeRemaining rows 4-7 / 20-23 of gait_out).
// Writes rows 0-3 of swing_in for every leg, rows 4-7 and 24-35 for the swinging ones.
__global__ void __launch_bounds__(64) qr_foothold_kernel(int n, FootholdDesc D, const float *__restrict__ g_in, const float *__restrict__ g_gait_state,
                                                         const float *__restrict__ g_gait_out, float *__restrict__ g_swing)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define FI(f) g_in[(size_t)(f) * N + i]
    const float e0 = FI(33), e1 = FI(34), e2 = FI(35), e3 = FI(36);
    float R[3][3];                                                        // baseRMat: body -> world (= dR on flat ground)
    R[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); R[0][1] = 2 * (e1 * e2 - e0 * e3); R[0][2] = 2 * (e1 * e3 + e0 * e2);
    R[1][0] = 2 * (e1 * e2 + e0 * e3); R[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); R[1][2] = 2 * (e2 * e3 - e0 * e1);
    R[2][0] = 2 * (e1 * e3 - e0 * e2); R[2][1] = 2 * (e2 * e3 + e0 * e1); R[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
    const float vdes[3] = {FI(16), FI(17), FI(18)}, wdes = FI(19), hdes = FI(20);
    const float roll = FI(37);
    const float vb[3] = {FI(40), FI(41), FI(42)}, w[3] = {FI(43), FI(44), FI(45)};
    const float dh2 = hdes - D.foot_clearance;
    const float c = cosf(roll), sn = sinf(roll);
#pragma unroll
    for (int leg = 0; leg < 4; ++leg) {
        const int st = (int)(g_gait_state ? g_gait_state[(size_t)(16 + leg) * N + i] : FI(leg));
        const float allow = g_gait_state ? g_gait_state[(size_t)(20 + leg) * N + i] : FI(4 + leg);
        const bool skip = (st == 1 && allow != 0.f) || st == 2;
        g_swing[(size_t)leg * N + i] = skip ? 0.f : 1.f;
        if (skip) continue;
        const float side = (leg & 1) ? 1.f : -1.f;
        const float ho[3] = {D.hip_offset[3 * leg], D.hip_offset[3 * leg + 1], D.hip_offset[3 * leg + 2]};
        const float twist[3] = {-ho[1], ho[0], 0.f};
        const float cr[3] = {w[1] * ho[2] - w[2] * ho[1], w[2] * ho[0] - w[0] * ho[2], w[0] * ho[1] - w[1] * ho[0]};
        const float hv0[3] = {vb[0] + cr[0], vb[1] + cr[1], vb[2] + cr[2]};
        float hv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) hv[r] = (R[r][0] * hv0[0] + R[r][1] * hv0[1]) + R[r][2] * hv0[2];
        hv[2] = 0.f;
        const float tv[3] = {vdes[0] + wdes * twist[0], vdes[1] + wdes * twist[1], vdes[2] + wdes * twist[2]};
        float ftp[3], phase;
        if (allow == 0.f) {
            const float hp[3] = {D.default_hip_position[3 * leg], D.default_hip_position[3 * leg + 1], D.default_hip_position[3 * leg + 2]};
            const float d[3] = {FI(21 + 3 * leg) - hp[0], FI(22 + 3 * leg) - hp[1], FI(23 + 3 * leg) - hp[2]};
            float t[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) t[r] = (R[r][0] * d[0] + R[r][1] * d[1]) + R[r][2] * d[2];
            if ((double)t[1] > 0.01 + 0.00 * (double)(-side)) t[1] = (float)((double)t[1] - 0.005);
            else if ((double)t[1] < -0.01 + 0.00 * (double)side) t[1] = (float)((double)t[1] + 0.005);
            t[2] = (float)((double)t[2] - 0.02);
#pragma unroll
            for (int r = 0; r < 3; ++r) ftp[r] = ((R[0][r] * t[0] + R[1][r] * t[1]) + R[2][r] * t[2]) + hp[r];
            phase = 1.0f;
        } else {
            const float s = g_gait_out ? g_gait_out[(size_t)(20 + leg) * N + i] : FI(8 + leg);
            float u[3], dP[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) u[r] = tv[r] * s - D.swing_kp[r] * (tv[r] - hv[r]);
#pragma unroll
            for (int r = 0; r < 3; ++r) dP[r] = (R[0][r] * u[0] + R[1][r] * u[1]) + R[2][r] * u[2];
            const float th = 0.2f;
            dP[0] = dP[0] < -th ? -th : (dP[0] > th ? th : dP[0]);
            dP[1] = dP[1] < -th ? -th : (dP[1] > th ? th : dP[1]);
            dP[2] = 0.f;
            const float iy = D.hip_l * side;
            const float rr[3] = {(0.f * 1.f + 0.f * iy) + 0.f * 0.f, (0.f * 0.f + c * iy) + sn * 0.f, (0.f * 0.f + -sn * iy) + c * 0.f};
            const float a[3] = {ho[0], ho[1], 0.f};
#pragma unroll
            for (int r = 0; r < 3; ++r) ftp[r] = (dP[r] + a[r]) + rr[r];
#pragma unroll
            for (int r = 0; r < 3; ++r) ftp[r] -= (R[0][r] * 0.f + R[1][r] * 0.f) + R[2][r] * dh2;
            phase = g_gait_out ? g_gait_out[(size_t)(4 + leg) * N + i] : FI(12 + leg);
        }
        g_swing[(size_t)(4 + leg) * N + i] = phase;
#pragma unroll
        for (int r = 0; r < 3; ++r) g_swing[(size_t)(24 + 3 * leg + r) * N + i] = ftp[r];
    }
#undef FI
}

// Ground-plane fit and control frame, one thread per robot, the reference's double arithmetic on the float robot state:
//   qrGroundSurfaceEstimator::Update / GetNormalVector / ComputeControlFrame    quadruped/src/estimators/qr_ground_surface_estimator.cpp:40-70,151-206
// (run by qrStateEstimatorContainer::Update in front of the robot estimator, quadruped/include/quadruped/estimators/qr_state_estimator_container.h:76-81).
// The update fires when all four feet are in contact and one of them newly so; the control frame assumes a flat ground in the world
// (nInWorldFrame := (0, 0, 1), :168), i.e. it is the base's heading, low-pass filtered as roll / pitch / yaw (ratio 0.8), roll := 0.
// g_in [23][n]: footContact[4], footPositionsInBaseFrame[12] (3*leg+axis), basePosition[3], quat_wxyz[4].  g_st [13][n] doubles:
// lastContactState[4], a[3], n[3], controlFrameRPY[3].  g_out [32][n]: a, n, controlFrameRPY, controlFrameOrientation[4], groundRMat[9]
// (row-major), baseRInControlFrame[9], updated.  g_est_in (may be null): the estimator's input array, whose rows 45-53
// (GetAlignedDirections) receive groundRMat.
__global__ void __launch_bounds__(64) qr_ground_kernel(int n, int fresh, const float *__restrict__ g_in, double *__restrict__ g_st, float *__restrict__ g_out,
                                                       float *__restrict__ g_est_in)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define IN(f) g_in[(size_t)(f) * N + i]
#define ST(f) g_st[(size_t)(f) * N + i]
    double a[3], nv[3], rpy[3];
    bool last[4];
    if (fresh) {                     // Reset(): a = 0, n = (0, 0, 1), controlFrameRPY = 0, lastContactState = 0
        for (int k = 0; k < 3; ++k) { a[k] = 0.0; nv[k] = (k == 2) ? 1.0 : 0.0; rpy[k] = 0.0; }
        for (int k = 0; k < 4; ++k) last[k] = false;
    } else {
        for (int k = 0; k < 4; ++k) last[k] = ST(k) != 0.0;
        for (int k = 0; k < 3; ++k) { a[k] = ST(4 + k); nv[k] = ST(7 + k); rpy[k] = ST(10 + k); }
    }
    bool shouldUpdate = false;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool c = IN(k) != 0.f;
        if (c) { if (!last[k]) shouldUpdate = true; ++cnt; }
        ST(k) = c ? 1.0 : 0.0;
    }
    const float qf0 = IN(19), qf1 = IN(20), qf2 = IN(21), qf3 = IN(22);
    const bool upd = !(cnt <= 3 || !shouldUpdate);
    if (upd) {
        double W[4][3], pZ[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) { W[l][0] = 1.0; W[l][1] = (double)IN(4 + 3 * l); W[l][2] = (double)IN(5 + 3 * l); pZ[l] = (double)IN(6 + 3 * l); }
        double ww[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { double acc = 0; for (int l = 0; l < 4; ++l) acc += W[l][r] * W[l][c]; ww[r][c] = acc; }
        double inv[3][3];
        {
            const double c00 = ww[1][1] * ww[2][2] - ww[1][2] * ww[2][1], c01 = ww[1][2] * ww[2][0] - ww[1][0] * ww[2][2], c02 = ww[1][0] * ww[2][1] - ww[1][1] * ww[2][0];
            const double det = ww[0][0] * c00 + ww[0][1] * c01 + ww[0][2] * c02;
            const double id = 1.0 / det;
            inv[0][0] = c00 * id; inv[1][0] = c01 * id; inv[2][0] = c02 * id;
            inv[0][1] = (ww[0][2] * ww[2][1] - ww[0][1] * ww[2][2]) * id; inv[1][1] = (ww[0][0] * ww[2][2] - ww[0][2] * ww[2][0]) * id; inv[2][1] = (ww[0][1] * ww[2][0] - ww[0][0] * ww[2][1]) * id;
            inv[0][2] = (ww[0][1] * ww[1][2] - ww[0][2] * ww[1][1]) * id; inv[1][2] = (ww[0][2] * ww[1][0] - ww[0][0] * ww[1][2]) * id; inv[2][2] = (ww[0][0] * ww[1][1] - ww[0][1] * ww[1][0]) * id;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double acc = 0;
#pragma unroll
            for (int l = 0; l < 4; ++l) { double m = 0; for (int c = 0; c < 3; ++c) m += inv[r][c] * W[l][c]; acc += m * pZ[l]; }
            a[r] = acc;
        }
        const double factor = sqrt(a[1] * a[1] + a[2] * a[2] + 1);
        nv[0] = -a[1] / factor; nv[1] = -a[2] / factor; nv[2] = 1.0 / factor;
        // base x axis (first column of quaternionToRotationMatrix(quat)^T) in double
        const double e0 = (double)qf0, e1 = (double)qf1, e2 = (double)qf2, e3 = (double)qf3;
        double x[3] = {1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 + e0 * e3), 2 * (e1 * e3 - e0 * e2)};
        const double nW[3] = {0, 0, 1};
        double y[3] = {nW[1] * x[2] - nW[2] * x[1], nW[2] * x[0] - nW[0] * x[2], nW[0] * x[1] - nW[1] * x[0]};
        { const double nn = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]); y[0] /= nn; y[1] /= nn; y[2] /= nn; }
        x[0] = y[1] * nW[2] - y[2] * nW[1]; x[1] = y[2] * nW[0] - y[0] * nW[2]; x[2] = y[0] * nW[1] - y[1] * nW[0];
        { const double nn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); x[0] /= nn; x[1] /= nn; x[2] /= nn; }
        // rotationMatrixToRPY(R^T), R = [x y n]: rotationMatrixToQuaternion transposes its argument again, so r = R
        const double r00 = x[0], r01 = y[0], r02 = nW[0], r10 = x[1], r11 = y[1], r12 = nW[1], r20 = x[2], r21 = y[2], r22 = nW[2];
        double q0, q1, q2, q3;
        const double tr = r00 + r11 + r22;
        if (tr > 0.0) { const double S = sqrt(tr + 1.0) * 2.0; q0 = 0.25 * S; q1 = (r21 - r12) / S; q2 = (r02 - r20) / S; q3 = (r10 - r01) / S; }
        else if (r00 > r11 && r00 > r22) { const double S = sqrt(1.0 + r00 - r11 - r22) * 2.0; q0 = (r21 - r12) / S; q1 = 0.25 * S; q2 = (r01 + r10) / S; q3 = (r02 + r20) / S; }
        else if (r11 > r22) { const double S = sqrt(1.0 + r11 - r00 - r22) * 2.0; q0 = (r02 - r20) / S; q1 = (r01 + r10) / S; q2 = 0.25 * S; q3 = (r12 + r21) / S; }
        else { const double S = sqrt(1.0 + r22 - r00 - r11) * 2.0; q0 = (r10 - r01) / S; q1 = (r02 + r20) / S; q2 = (r12 + r21) / S; q3 = 0.25 * S; }
        double nr[3];
        const double as = fmin(-2. * (q1 * q3 - q0 * q2), .99999);
        nr[2] = atan2(2 * (q1 * q2 + q0 * q3), q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3);
        nr[1] = asin(as);
        nr[0] = atan2(2 * (q2 * q3 + q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3);
        const double ratio = 0.8;
        for (int k = 0; k < 3; ++k) rpy[k] = (1 - ratio) * rpy[k] + ratio * nr[k];
        rpy[0] = 0;
        for (int k = 0; k < 3; ++k) { ST(4 + k) = a[k]; ST(7 + k) = nv[k]; ST(10 + k) = rpy[k]; }
    } else if (fresh) {
        for (int k = 0; k < 3; ++k) { ST(4 + k) = a[k]; ST(7 + k) = nv[k]; ST(10 + k) = rpy[k]; }
    }
    // R = rpyToRotMat(rpy)^T, rpyToRotMat = Rx(r) Ry(p) Rz(y) of coordinate-transform matrices (qr_se3.h:72-89,109-116)
    double Rm[3][3];
    {
        double sr, cr, sp, cp, sy, cy;
        sincos(rpy[0], &sr, &cr); sincos(rpy[1], &sp, &cp); sincos(rpy[2], &sy, &cy);
        const double X[9] = {1, 0, 0, 0, cr, sr, 0, -sr, cr}, Y[9] = {cp, 0, -sp, 0, 1, 0, sp, 0, cp}, Z[9] = {cy, sy, 0, -sy, cy, 0, 0, 0, 1};
        double XY[9], M[9];
        mulmat3(X, Y, XY); mulmat3(XY, Z, M);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rm[r][c] = M[3 * c + r];
    }
    if (g_out || g_est_in) {
        float g[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g[r][c] = (float)Rm[r][c];
        if (g_est_in) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g_est_in[(size_t)(45 + 3 * r + c) * N + i] = g[r][c];
        if (g_out) {
            // controlFrameOrientation = rpyToQuat(rpy) = rotationMatrixToQuaternion(rpyToRotMat(rpy)): r = rpyToRotMat^T = Rm
            double q0, q1, q2, q3;
            const double r00 = Rm[0][0], r01 = Rm[0][1], r02 = Rm[0][2], r10 = Rm[1][0], r11 = Rm[1][1], r12 = Rm[1][2], r20 = Rm[2][0], r21 = Rm[2][1], r22 = Rm[2][2];
            const double tr = r00 + r11 + r22;
            if (tr > 0.0) { const double S = sqrt(tr + 1.0) * 2.0; q0 = 0.25 * S; q1 = (r21 - r12) / S; q2 = (r02 - r20) / S; q3 = (r10 - r01) / S; }
            else if (r00 > r11 && r00 > r22) { const double S = sqrt(1.0 + r00 - r11 - r22) * 2.0; q0 = (r21 - r12) / S; q1 = 0.25 * S; q2 = (r01 + r10) / S; q3 = (r02 + r20) / S; }
            else if (r11 > r22) { const double S = sqrt(1.0 + r11 - r00 - r22) * 2.0; q0 = (r02 - r20) / S; q1 = (r01 + r10) / S; q2 = 0.25 * S; q3 = (r12 + r21) / S; }
            else { const double S = sqrt(1.0 + r22 - r00 - r11) * 2.0; q0 = (r10 - r01) / S; q1 = (r02 + r20) / S; q2 = (r12 + r21) / S; q3 = 0.25 * S; }
            // baseRMat (float, qr_robot.cpp:70) = quaternionToRotationMatrix(quat)^T
            const float e0 = qf0, e1 = qf1, e2 = qf2, e3 = qf3;
            float B[3][3];
            B[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); B[0][1] = 2 * (e1 * e2 - e0 * e3); B[0][2] = 2 * (e1 * e3 + e0 * e2);
            B[1][0] = 2 * (e1 * e2 + e0 * e3); B[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); B[1][2] = 2 * (e2 * e3 - e0 * e1);
            B[2][0] = 2 * (e1 * e3 - e0 * e2); B[2][1] = 2 * (e2 * e3 + e0 * e1); B[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
#define OUT(f) g_out[(size_t)(f) * N + i]
            for (int k = 0; k < 3; ++k) { OUT(k) = (float)a[k]; OUT(3 + k) = (float)nv[k]; OUT(6 + k) = (float)rpy[k]; }
            OUT(9) = (float)q0; OUT(10) = (float)q1; OUT(11) = (float)q2; OUT(12) = (float)q3;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    OUT(13 + 3 * r + c) = g[r][c];
                    float acc = 0.f;
                    for (int k = 0; k < 3; ++k) acc += g[k][r] * B[k][c];
                    OUT(22 + 3 * r + c) = acc;
                }
            OUT(31) = upd ? 1.f : 0.f;
#undef OUT
        }
    }
#undef IN
#undef ST
}

// Walk gait generator and the force windows its sub-states select, one thread per robot, plain float arithmetic with the reference's
// float / double mix:
//   qrWalkGaitGenerator::Update                                   quadruped/src/gait/qr_walk_gait_generator.cpp:202-288
//   qrGaitGenerator::Reset                                       quadruped/include/quadruped/gait/qr_gait.h:76-87
//   TorqueStanceLegController::UpdateFRatio (walk branch)         quadruped/src/controllers/balance_controller/qr_torque_stance_leg_controller.cpp:125-168
// st [QRGPU_WALK_STATE_FLOATS = 33][n]: cur[4], desired[4], leg[4], detected[4], stateIndexOfLegs[4], phase[4], normalizedPhase[4],
// detectedEventTickPhase[4], moveBasePhase.  fresh: 2 = as constructed (+ Reset(0)), 1 = Reset() only (stateIndexOfLegs and the detection
// members survive a Reset in the reference).  g_out [41][n]: phaseInFullCycle[4], normalizedPhase[4], desiredLegState[4], legState[4],
// curLegState[4], detectedLegState[4], detectedEventTickPhase[4], moveBasePhase, contacts[4], fMinRatio[4], fMaxRatio[4].
// g_ratio [8][n] (may be null): fMinRatio, fMaxRatio as qrgpu_vmc_force_world_batch takes them; g_vmc_in (may be null): rows 18-21 (contacts).
__global__ void __launch_bounds__(64) qr_walk_gait_kernel(int n, WalkDesc D, float currentTime, int stop, int fresh, const float *__restrict__ g_contact,
                                                          float *__restrict__ st, float *__restrict__ g_out, float *__restrict__ g_ratio, float *__restrict__ g_vmc_in)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define ST(f) st[(size_t)(f) * N + i]
    float mbp = (fresh == 2) ? 0.f : ST(32);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        int cur, desired, leg, detected, sidx;
        float phase, nphase, evp;
        if (fresh) {
            cur = desired = leg = D.initial_leg_state[l]; nphase = 0.f;
            if (fresh == 2) { sidx = D.state_index0[l]; phase = 0.f; detected = 0; evp = 0.f; }
            else { sidx = (int)ST(16 + l); phase = ST(20 + l); detected = (int)ST(12 + l); evp = ST(28 + l); }
        } else {
            cur = (int)ST(l); desired = (int)ST(4 + l); leg = (int)ST(8 + l); detected = (int)ST(12 + l); sidx = (int)ST(16 + l);
            phase = ST(20 + l); nphase = ST(24 + l); evp = ST(28 + l);
        }
        const bool contact = g_contact[(size_t)l * N + i] != 0.f;
        if (!stop || (stop && cur == 0)) cur = desired;
        const float augmentedTime = D.initial_leg_phase[l] * D.full[l] + currentTime;
        phase = fmodf(augmentedTime, D.full[l]) / D.full[l];
        const float ratio = D.duty_factor[l];
        if (phase <= ratio) {
            if (cur != 1) sidx = 0;
            desired = 1; leg = 1;
            nphase = phase / ratio;
        } else {
            desired = 0; leg = 0;
            nphase = (float)((double)(phase - ratio) / (1.0 - (double)ratio));
        }
        if (desired == 0) {
            int idx = sidx;
            const float start = D.accum[idx], end = D.accum[idx + 1];
            const float psc = (float)((double)(phase - ratio) / (1.0 - (double)ratio));
            if (psc <= end && psc >= start) {
                desired = D.que[idx];
                nphase = (psc - start) / (end - start);
            } else {
                idx += 1;
                if (idx > D.nq - 1) idx = D.nq - 1;      // (the reference indexes past its queue here: only if a tick were longer than a sub-state)
                desired = D.que[idx];
                sidx = idx;
                nphase = (psc - D.accum[idx]) / D.ratio[idx];
            }
            mbp = (psc < D.true_swing_start_in_swing) ? psc / D.true_swing_start_in_swing : 1.0f;
        }
        detected = (desired != 1) ? 0 : 1;
        if (!(nphase < D.contact_detection_phase_threshold)) {
            if (desired == 8 && contact) { detected = 2; evp = phase; }
            else if (desired == 1 && !contact) { detected = 3; evp = phase; }
        }
        ST(l) = (float)cur; ST(4 + l) = (float)desired; ST(8 + l) = (float)leg; ST(12 + l) = (float)detected; ST(16 + l) = (float)sidx;
        ST(20 + l) = phase; ST(24 + l) = nphase; ST(28 + l) = evp;
        // UpdateFRatio, walk branch
        float ph = nphase, cont, fmax;
        const float fmin = 0.001f;
        if (detected == 1 || detected == 3) { cont = 1.f; fmax = 10.0f; }
        else if (detected == 2) { cont = 1.f; const float t = fabsf(ph - 0.8f); fmax = 10.0f * fminf(0.01f, t); }
        else if (desired == 5) { cont = 1.f; fmax = 10.0f * fmaxf(0.001f, ph); }
        else if (desired == 6) { cont = 1.f; ph = ph / (3.f / 4.0f); fmax = 10.0f * fmaxf(0.001f, 1.0f - ph); }
        else if (desired == 8) { cont = 0.f; fmax = 0.002f; }
        else { cont = 1.f; fmax = 10.0f; }
        if (g_out) {
            g_out[(size_t)l * N + i] = phase; g_out[(size_t)(4 + l) * N + i] = nphase; g_out[(size_t)(8 + l) * N + i] = (float)desired;
            g_out[(size_t)(12 + l) * N + i] = (float)leg; g_out[(size_t)(16 + l) * N + i] = (float)cur; g_out[(size_t)(20 + l) * N + i] = (float)detected;
            g_out[(size_t)(24 + l) * N + i] = evp; g_out[(size_t)(29 + l) * N + i] = cont; g_out[(size_t)(33 + l) * N + i] = fmin; g_out[(size_t)(37 + l) * N + i] = fmax;
        }
        if (g_ratio) { g_ratio[(size_t)l * N + i] = fmin; g_ratio[(size_t)(4 + l) * N + i] = fmax; }
        if (g_vmc_in) g_vmc_in[(size_t)(18 + l) * N + i] = cont;
    }
    ST(32) = mbp;
    if (g_out) g_out[(size_t)28 * N + i] = mbp;
#undef ST
}

// Swing-leg action of the velocity mode (the trot of the force-balance path), one thread per robot:
//   qrRaibertSwingLegController::GetAction, VELOCITY_LOCOMOTION case    quadruped/src/controllers/qr_swing_leg_controller.cpp:285-309, 408-424
//   SwingFootTrajectory::GenerateTrajectoryPoint(phaseModule = true)   quadruped/src/controllers/qr_foot_trajectory_generator.cpp:322-343
// g_in [53][n]: swing flag[4], normalizedPhase[4], phaseSwitchFootLocalPos[12], estimated base velocity (base frame)[3], yaw rate,
// desiredSpeed[3], desiredTwistingSpeed, dR[9] (baseRInControlFrame, row-major: rows 22-30 of the ground kernel's output), quat_wxyz[4],
// motor angles[12].  g_out [48][n], for the flagged legs: footTargetPosition[12] (base frame), footPositionInBaseFrame[12], joint angle
// targets[12], joint velocity targets[12].
__global__ void __launch_bounds__(64) qr_swing_velocity_kernel(int n, EstimatorDesc D, SwingVelDesc V, const float *__restrict__ g_in, float *__restrict__ g_out)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)n;
#define IN(f) g_in[(size_t)(f) * N + i]
    const float e0 = IN(37), e1 = IN(38), e2 = IN(39), e3 = IN(40);
    float Rwb[3][3];                                                      // world -> body = baseRMat^T
    Rwb[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); Rwb[1][0] = 2 * (e1 * e2 - e0 * e3); Rwb[2][0] = 2 * (e1 * e3 + e0 * e2);
    Rwb[0][1] = 2 * (e1 * e2 + e0 * e3); Rwb[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); Rwb[2][1] = 2 * (e2 * e3 - e0 * e1);
    Rwb[0][2] = 2 * (e1 * e3 - e0 * e2); Rwb[1][2] = 2 * (e2 * e3 + e0 * e1); Rwb[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
    float dR[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dR[k] = IN(28 + k);
    const float bvel[3] = {IN(20), IN(21), IN(22)}, yawDot = IN(23), sp[3] = {IN(24), IN(25), IN(26)}, twist = IN(27);
#pragma unroll
    for (int leg = 0; leg < 4; ++leg) {
        if (IN(leg) == 0.f) continue;
        const float ho[3] = {V.hip_pos_com[3 * leg], V.hip_pos_com[3 * leg + 1], V.hip_pos_com[3 * leg + 2]};
        const float tw[3] = {-ho[1], ho[0], 0.f};
        float hv[3], hh[3], tgtv[3], u[3], tg[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) hv[r] = bvel[r] + yawDot * tw[r];
#pragma unroll
        for (int r = 0; r < 3; ++r) hh[r] = (dR[3 * r] * hv[0] + dR[3 * r + 1] * hv[1]) + dR[3 * r + 2] * hv[2];
        hh[2] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) tgtv[r] = sp[r] + twist * tw[r];
#pragma unroll
        for (int r = 0; r < 3; ++r) u[r] = hh[r] * V.stance_duration[leg] / 2.0f - V.swing_kp[r] * (tgtv[r] - hh[r]);
        const float dh[3] = {0.f, 0.f, V.desired_height};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float a = (dR[r] * u[0] + dR[3 + r] * u[1]) + dR[6 + r] * u[2];
            const float b = (Rwb[r][0] * dh[0] + Rwb[r][1] * dh[1]) + Rwb[r][2] * dh[2];
            const float off = (r < 2) ? ho[r] : 0.f;
            tg[r] = (a + off) - b;
        }
        const float st[3] = {IN(8 + 3 * leg), IN(9 + 3 * leg), IN(10 + 3 * leg)};
        const float inputPhase = IN(4 + leg);
        float phase;
        if (inputPhase <= 0.5f) phase = (float)(0.8 * sin((double)inputPhase * 3.14159265358979323846));
        else phase = (float)(0.8 + ((double)inputPhase - 0.5) * 0.4);
        float pw[3] = {0.f, 0.f, 0.f};
        if (!((double)phase < 0.0 - 1e-3) && !((double)phase >= 0.0 + 1.0 + 1e-3)) {
            pw[0] = (1 - phase) * st[0] + phase * tg[0];
            pw[1] = (1 - phase) * st[1] + phase * tg[1];
            const float mid = (tg[2] > st[2] ? tg[2] : st[2]) + 0.1f;
            if (!(phase < 0.f)) {
                const float d1 = mid - st[2], d2 = tg[2] - st[2];
                const float d3 = (float)(0.25 - 0.5);
                const float ca = (d1 - d2 * 0.5f) / d3;
                const float cb = (float)(((double)d2 * 0.25 - (double)d1) / (double)d3);
                pw[2] = (float)((double)ca * ((double)phase * (double)phase) + (double)(cb * phase) + (double)st[2]);
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { g_out[(size_t)(3 * leg + r) * N + i] = tg[r]; g_out[(size_t)(12 + 3 * leg + r) * N + i] = pw[r]; }
        const float sh = D.hip_l * ((leg & 1) ? 1.f : -1.f);
        const float x = pw[0] - D.hip_offset[3 * leg], y = pw[1] - D.hip_offset[3 * leg + 1], z = pw[2] - D.hip_offset[3 * leg + 2];
        const float lu = D.upper_l, ll = D.lower_l;
        const float tK = -acosf(((x * x + y * y + z * z) - (sh * sh + lu * lu + ll * ll)) / (2 * ll * lu));
        const float l = sqrtf(lu * lu + ll * ll + 2 * lu * ll * cosf(tK));
        const float tH = asinf(-x / l) - tK / 2;
        const float c1 = sh * y - l * cosf(tH + tK / 2) * z;
        const float s1 = l * cosf(tH + tK / 2) * y + sh * z;
        const float tA = atan2f(s1, c1);
        const float ang[3] = {tA, tH, tK};
        // J^-1 * (the generator's zero velocity): zero, or NaN where the Jacobian is (an unreachable target), as the reference's product is
        float J[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) leg_jacobian_column(j, tA, tH, tK, sh, lu, ll, J[0][j], J[1][j], J[2][j]);
        const float det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        const float id = 1.f / det;
        const float Ji[3][3] = {{(J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id, (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id, (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id},
                                {(J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id, (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id, (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id},
                                {(J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id, (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id, (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id}};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float a = ang[r];
            if (a != a) a = IN(41 + 3 * leg + r);                         // unreachable target: keep the current angle (:415-418)
            g_out[(size_t)(24 + 3 * leg + r) * N + i] = a;
            g_out[(size_t)(36 + 3 * leg + r) * N + i] = Ji[r][0] * 0.f + Ji[r][1] * 0.f + Ji[r][2] * 0.f;
        }
    }
#undef IN
}

}  // namespace qrgpu
