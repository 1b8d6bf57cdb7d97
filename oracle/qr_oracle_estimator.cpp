// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Base velocity estimator + the leg kinematics it reads (first part):
//   qrRobot::UpdateDataFlow                  QS/robots/qr_robot.cpp:62-72  (foot Jacobians, foot positions / velocities in the base frame)
//   qrRobotVelocityEstimator::Update         QS/estimators/qr_robot_velocity_estimator.cpp:77-133
//   qrRobotPoseEstimator::Update             QS/estimators/qr_robot_pose_estimator.cpp:68-165 (height from the stance feet, planar odometry;
//                                            called right after the velocity estimator, qr_robot_estimator.cpp:81-82)
//   qrMovingWindowFilter (Neumaier sums)     QI/estimators/qr_moving_window_filter.hpp:150-186, 236-263
//   TinyEKF<3,3>::model / ekf_step           QX/TinyEKF/src/TinyEKF.h:103-124, QX/TinyEKF/src/tiny_ekf.c:292-332 (+ cholsl :17-93)
// The Kalman step is restated operation by operation (same loops, same accumulation order) and pinned against the reference's own
// TinyEKF compiled into oracle/_ref (tests/test_oracle_estimator.py).
#include "qr_oracle.h"

namespace qro {

namespace {
// tiny_ekf.c helpers, n = m = 3, row-major
void mulmat3(const double *a, const double *b, double *c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            c[3 * i + j] = 0;
            for (int l = 0; l < 3; ++l) c[3 * i + j] += a[3 * i + l] * b[3 * l + j];
        }
}
int cholsl3(const double *A, double *a, double *p)
{
    const int n = 3;
    for (int i = 0; i < 9; ++i) a[i] = A[i];
    for (int i = 0; i < n; i++)                      // choldc1
        for (int j = i; j < n; j++) {
            double sum = a[i * n + j];
            for (int k = i - 1; k >= 0; k--) sum -= a[i * n + k] * a[j * n + k];
            if (i == j) { if (sum <= 0) return 1; p[i] = std::sqrt(sum); }
            else a[j * n + i] = sum / p[i];
        }
    for (int i = 0; i < n; i++) {                    // choldcsl
        a[i * n + i] = 1 / p[i];
        for (int j = i + 1; j < n; j++) {
            double sum = 0;
            for (int k = i; k < j; k++) sum -= a[j * n + k] * a[k * n + i];
            a[j * n + i] = sum / p[j];
        }
    }
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) a[i * n + j] = 0.0;     // cholsl
    for (int i = 0; i < n; i++) {
        a[i * n + i] *= a[i * n + i];
        for (int k = i + 1; k < n; k++) a[i * n + i] += a[k * n + i] * a[k * n + i];
        for (int j = i + 1; j < n; j++) for (int k = j; k < n; k++) a[i * n + j] += a[k * n + i] * a[k * n + j];
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) a[i * n + j] = a[j * n + i];
    return 0;
}
}  // namespace

// One TinyEKF<3,3> step with the estimator's model: fx = x + deltaV, F = I, hx = fx, H = I (TinyEKF.h:103-124).
int ekf3_step(double x[3], double P[9], double qvar, double rvar, const double deltaV[3], const double z[3])
{
    double F[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Q[9] = {qvar, 0, 0, 0, qvar, 0, 0, 0, qvar}, R[9] = {rvar, 0, 0, 0, rvar, 0, 0, 0, rvar};
    double fx[3] = {x[0] + deltaV[0], x[1] + deltaV[1], x[2] + deltaV[2]}, hx[3] = {fx[0], fx[1], fx[2]};
    double tmp0[9], Ft[9], Pp[9], Ht[9], tmp1[9], tmp2[9], tmp3[9], tmp4[9], tmp5[3], G[9];
    mulmat3(F, P, tmp0);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ft[3 * j + i] = F[3 * i + j];
    mulmat3(tmp0, Ft, Pp);
    for (int i = 0; i < 9; ++i) Pp[i] += Q[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ht[3 * j + i] = H[3 * i + j];
    mulmat3(Pp, Ht, tmp1);
    mulmat3(H, Pp, tmp2);
    mulmat3(tmp2, Ht, tmp3);
    for (int i = 0; i < 9; ++i) tmp3[i] += R[i];
    if (cholsl3(tmp3, tmp4, tmp5)) return 1;
    mulmat3(tmp1, tmp4, G);
    for (int i = 0; i < 3; ++i) tmp5[i] = z[i] - hx[i];
    for (int i = 0; i < 3; ++i) { double y = 0; for (int j = 0; j < 3; ++j) y += tmp5[j] * G[3 * i + j]; tmp2[i] = y; }     // mulvec
    for (int i = 0; i < 3; ++i) x[i] = fx[i] + tmp2[i];
    mulmat3(G, H, tmp0);
    for (int i = 0; i < 9; ++i) tmp0[i] = -tmp0[i];
    for (int i = 0; i < 3; ++i) tmp0[4 * i] += 1;
    mulmat3(tmp0, Pp, P);
    return 0;
}

// Neumaier moving-window average over a ring buffer (CalculateAverage, :236-263 / :171-186).  T is the filter's arithmetic type.
template <typename T>
static T window_average(T *win, int W, int &count, int &head, T &sum, T &corr, T value)
{
    auto neumaier = [&](T v) {
        const T ns = sum + v;
        if (std::abs(sum) >= std::abs(v)) corr += (sum - ns) + v;
        else corr += (v - ns) + sum;
        sum = ns;
    };
    int len = count;
    if (len >= W) { neumaier(-win[head]); --len; }
    neumaier(value);
    win[head] = value;                  // the oldest slot is the one just released (or the next free one while filling)
    head = (head + 1) % W;
    count = len + 1;
    return (sum + corr) / (T)(len + 1);
}

// in[54]: sensorAcc[3] (baseAccInBaseFrame), baseLinearAcceleration[3], quat_wxyz[4], rpyRate[3], footContact[4], q[12], dq[12],
//         desiredLegState[4], groundOrientationMat[9] (GetAlignedDirections, row-major; identity on a plane); tick (ms) separately.
// out[42]: filteredAcc[3], baseVInWorldFrame[3], baseVelocityInBaseFrame[3], baseWInWorldFrame[3], footPositionsInBaseFrame[12],
//         footVelocitiesInBaseFrame[12], basePosition[3], heightInControlFrame, absoluteHight, estimatedPose yaw.
void estimator_update(const EstimatorConfig &cfg, const float in[54], unsigned tick, EstimatorState &s, float out[42])
{
    const float *sensorAcc = in, *linAcc = in + 3, *quat = in + 6, *rpyRate = in + 10, *contact = in + 13, *q = in + 17, *dq = in + 29;
    // UpdateDataFlow: foot Jacobians, velocities, positions in the base frame (:62-72, :187-197)
    LegGeom geo; geo.hip_l = cfg.hip_l; geo.upper_l = cfg.upper_l; geo.lower_l = cfg.lower_l;
    float footP[12], footV[12];
    foot_positions_in_base_frame(geo, cfg.hip_offset, q, footP);
    for (int leg = 0; leg < 4; ++leg) {
        float J[9];
        analytical_leg_jacobian(geo, q + 3 * leg, leg, J);
        for (int i = 0; i < 3; ++i) footV[3 * leg + i] = J[3 * i] * dq[3 * leg] + J[3 * i + 1] * dq[3 * leg + 1] + J[3 * i + 2] * dq[3 * leg + 2];
    }
    // AccFilter (:79-80)
    float facc[3];
    {
        int cnt = s.acc_count, head = s.acc_head, c2 = cnt, h2 = head;
        for (int a = 0; a < 3; ++a) { c2 = cnt; h2 = head; facc[a] = window_average<float>(s.acc_win[a], 20, c2, h2, s.acc_sum[a], s.acc_corr[a], linAcc[a]); }
        s.acc_count = c2; s.acc_head = h2;
    }
    // ComputeDeltaTime (:64-75)
    float deltaTime;
    if ((double)s.last_timestamp < 1e-5) deltaTime = cfg.time_step;
    else deltaTime = (tick - s.last_timestamp) / 1000.;
    s.last_timestamp = tick;
    // rotMat = quaternionToRotationMatrix(q)^T  (body -> world)
    Q4<float> qq = {{quat[0], quat[1], quat[2], quat[3]}};
    M3<float> Rt = quaternionToRotationMatrix(qq);
    float R[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = Rt[j][i];
    float cal[3];
    for (int i = 0; i < 3; ++i) cal[i] = R[i][0] * sensorAcc[0] + R[i][1] * sensorAcc[1] + R[i][2] * sensorAcc[2];
    cal[2] -= 9.81;
    const double deltaV[3] = {cal[0] * deltaTime, cal[1] * deltaTime, cal[2] * deltaTime};
    // observed base velocity from the contact legs (:94-103)
    float mean[3] = {0, 0, 0};
    int num = 0;
    for (int leg = 0; leg < 4; ++leg) {
        if (contact[leg] == 0.f) continue;
        const float *p = footP + 3 * leg, *v = footV + 3 * leg;
        // vectorToSkewMat(rpyRate) * p
        const float cx = 0 * p[0] + (-rpyRate[2]) * p[1] + rpyRate[1] * p[2];
        const float cy = rpyRate[2] * p[0] + 0 * p[1] + (-rpyRate[0]) * p[2];
        const float cz = (-rpyRate[1]) * p[0] + rpyRate[0] * p[1] + 0 * p[2];
        const float vB[3] = {v[0] + cx, v[1] + cy, v[2] + cz};
        for (int i = 0; i < 3; ++i) mean[i] += (-R[i][0]) * vB[0] + (-R[i][1]) * vB[1] + (-R[i][2]) * vB[2];
        ++num;
    }
    double z[3];
    if (num > 0) { for (int i = 0; i < 3; ++i) { mean[i] /= num; z[i] = mean[i]; } }
    else for (int i = 0; i < 3; ++i) z[i] = s.est_vel_base[i];          // estimatedVelocity holds the BASE-frame value of the last call (:130)
    ekf3_step(s.x, s.P, (double)cfg.accelerometer_variance, (double)cfg.sensor_variance, deltaV, z);
    const float xf[3] = {(float)s.x[0], (float)s.x[1], (float)s.x[2]};
    float vw[3];
    {
        int cnt = s.vel_count, head = s.vel_head, c2 = cnt, h2 = head;
        for (int a = 0; a < 3; ++a) {
            c2 = cnt; h2 = head;
            vw[a] = (float)window_average<double>(s.vel_win[a].data(), cfg.window, c2, h2, s.vel_sum[a], s.vel_corr[a], (double)xf[a]);
        }
        s.vel_count = c2; s.vel_head = h2;
    }
    for (int i = 0; i < 3; ++i) {
        out[i] = facc[i];
        out[3 + i] = vw[i];
        s.est_vel_base[i] = R[0][i] * vw[0] + R[1][i] * vw[1] + R[2][i] * vw[2];          // rotMat^T * v
        out[6 + i] = s.est_vel_base[i];
        out[9 + i] = R[i][0] * rpyRate[0] + R[i][1] * rpyRate[1] + R[i][2] * rpyRate[2];
    }
    for (int i = 0; i < 12; ++i) { out[12 + i] = footP[i]; out[24 + i] = footV[i]; }

    // ---- qrRobotPoseEstimator::Update (its own lastTimestamp sees the same ticks: same deltaTime)
    const float *desLeg = in + 41, *gm = in + 45;
    int nct = 0;
    for (int leg = 0; leg < 4; ++leg) nct += ((int)desLeg[leg] == 1 /*STANCE*/) ? 1 : 0;
    float height, hctrl = 0.f;
    if (nct == 0) { height = cfg.body_height; hctrl = std::nanf(""); }      // heightInControlFrame is left as it was (:103-105)
    else {
        float hs = 0.f, hc = 0.f;
        for (int leg = 0; leg < 4; ++leg) {
            const float *p = footP + 3 * leg;
            float w[3];
            for (int i = 0; i < 3; ++i) w[i] = R[i][0] * p[0] + R[i][1] * p[1] + R[i][2] * p[2];              // rotMat * footPositions
            const float cz = gm[2] * w[0] + gm[5] * w[1] + gm[8] * w[2];                                         // row 2 of groundOrientationMat^T
            const float c = ((int)desLeg[leg] == 1) ? 1.f : 0.f;
            hc += (-cz) * c;
            hs += (-w[2]) * c;
        }
        hctrl = hc / nct;
        height = hs / nct;
    }
    {   // ComputePose (:137-165): planar odometry with the base-frame velocity the velocity estimator has just left behind
        const float vX = s.est_vel_base[0], vY = s.est_vel_base[1], vZ = s.est_vel_base[2], vTheta = rpyRate[2];
        const float theta = s.pose_theta;
        const float deltaX = (vX * std::cos((double)theta) - vY * std::sin((double)theta)) * deltaTime;
        const float deltaY = (vX * std::sin((double)theta) + vY * std::cos((double)theta)) * deltaTime;
        const float deltaTheta = vTheta * deltaTime;
        s.pose_x += deltaX; s.pose_y += deltaY;
        s.abs_height += vZ * deltaTime;
        s.pose_theta = theta + deltaTheta;
    }
    out[36] = s.pose_x; out[37] = s.pose_y; out[38] = height; out[39] = hctrl; out[40] = s.abs_height; out[41] = s.pose_theta;
}

}  // namespace qro

extern "C" {
int qro_ekf3_run(double qvar, double rvar, int nsteps, const double *deltaV, const double *z, double *x_out)
{
    double x[3] = {0, 0, 0}, P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int bad = 0;
    for (int k = 0; k < nsteps; ++k) {
        bad += qro::ekf3_step(x, P, qvar, rvar, deltaV + 3 * k, z + 3 * k);
        for (int i = 0; i < 3; ++i) x_out[3 * k + i] = x[i];
    }
    return bad;
}
// A sequence of nticks updates of ONE robot from a fresh estimator: in [nticks][54], tick [nticks], out [nticks][42].
// cfg20: hip_l, upper_l, lower_l, time_step, accelerometerVariance, sensorVariance, window, hip_offset[12], body_height.
void qro_estimator_run(const float *cfg, int nticks, const float *in, const unsigned *tick, float *out)
{
    qro::EstimatorConfig c;
    c.hip_l = cfg[0]; c.upper_l = cfg[1]; c.lower_l = cfg[2]; c.time_step = cfg[3]; c.accelerometer_variance = cfg[4]; c.sensor_variance = cfg[5];
    c.window = (int)cfg[6];
    for (int i = 0; i < 12; ++i) c.hip_offset[i] = cfg[7 + i];
    c.body_height = cfg[19];
    qro::EstimatorState s(c.window);
    for (int k = 0; k < nticks; ++k) qro::estimator_update(c, in + 54 * k, tick[k], s, out + 42 * k);
}
}
