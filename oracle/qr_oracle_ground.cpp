// TEST INFRASTRUCTURE -- CPU restatement of the reference's ground-plane estimator; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// qrGroundSurfaceEstimator::Update / GetNormalVector / ComputeControlFrame   QS/estimators/qr_ground_surface_estimator.cpp:40-70,151-206
// (the estimator qrStateEstimatorContainer::Update runs in front of the robot estimator, QI/estimators/qr_state_estimator_container.h:76-81),
// in the reference's double arithmetic on the float robot state.
//
// Parity unpinned against a reference binary: the class needs Eigen and yaml-cpp, neither of which is in this image.  Pinned by properties
// the formulas must satisfy (tests/test_oracle_ground.py): the plane through four coplanar feet, its unit normal, the gating of the update
// on a fresh fourth contact, the filtered control-frame yaw, roll = 0, orthonormal frames.
#include "qr_oracle.h"

#include <cmath>
#include <cstring>

namespace qro {

// quatToRPY  QI/utils/qr_se3.h:210-223
static V3<double> quat_to_rpy(const Q4<double> &q)
{
    V3<double> rpy;
    const double as = std::min(-2. * (q[1] * q[3] - q[0] * q[2]), .99999);
    rpy[2] = std::atan2(2 * (q[1] * q[2] + q[0] * q[3]), q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3]);
    rpy[1] = std::asin(as);
    rpy[0] = std::atan2(2 * (q[2] * q[3] + q[0] * q[1]), q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3]);
    return rpy;
}

void ground_reset(GroundState &s)      // Reset(): :71-128 (the terrain bookkeeping is configuration, not state)
{
    std::memset(&s, 0, sizeof(s));
    s.n[2] = 1.0;
}

// in[23]: footContact[4], footPositionsInBaseFrame[12] (3*leg+axis), basePosition[3], quat_wxyz[4]
// out[32]: a[3], n[3] (base frame), controlFrameRPY[3], controlFrameOrientation[4], groundRMat[9] (row-major), baseRInControlFrame[9], updated
void ground_update(const float in[23], GroundState &s, float out[32])
{
    bool contact[4];
    bool shouldUpdate = false;
    int N = 0;
    for (int i = 0; i < 4; ++i) {
        contact[i] = in[i] != 0.f;
        if (contact[i]) { if (!s.last_contact[i]) shouldUpdate = true; ++N; }
    }
    for (int i = 0; i < 4; ++i) s.last_contact[i] = contact[i];
    Q4<double> quat; for (int i = 0; i < 4; ++i) quat[i] = (double)in[19 + i];
    const bool upd = !(N <= 3 || !shouldUpdate);
    if (upd) {
        // the plane z(x, y) = a0 + a1 x + a2 y through the four feet (base frame): a = (W'W)^-1 W' pZ, W = [1 x y]   (:58-66)
        double W[4][3], pZ[4];
        for (int l = 0; l < 4; ++l) { W[l][0] = 1.0; W[l][1] = (double)in[4 + 3 * l]; W[l][2] = (double)in[4 + 3 * l + 1]; pZ[l] = (double)in[4 + 3 * l + 2]; }
        double ww[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double acc = 0; for (int l = 0; l < 4; ++l) acc += W[l][i] * W[l][j]; ww[i][j] = acc; }
        // fixed-size 3 x 3 inverse: cofactors over the determinant
        double inv[3][3];
        {
            const double c00 = ww[1][1] * ww[2][2] - ww[1][2] * ww[2][1], c01 = ww[1][2] * ww[2][0] - ww[1][0] * ww[2][2], c02 = ww[1][0] * ww[2][1] - ww[1][1] * ww[2][0];
            const double det = ww[0][0] * c00 + ww[0][1] * c01 + ww[0][2] * c02;
            const double id = 1.0 / det;
            inv[0][0] = c00 * id; inv[1][0] = c01 * id; inv[2][0] = c02 * id;
            inv[0][1] = (ww[0][2] * ww[2][1] - ww[0][1] * ww[2][2]) * id; inv[1][1] = (ww[0][0] * ww[2][2] - ww[0][2] * ww[2][0]) * id; inv[2][1] = (ww[0][1] * ww[2][0] - ww[0][0] * ww[2][1]) * id;
            inv[0][2] = (ww[0][1] * ww[1][2] - ww[0][2] * ww[1][1]) * id; inv[1][2] = (ww[0][2] * ww[1][0] - ww[0][0] * ww[1][2]) * id; inv[2][2] = (ww[0][0] * ww[1][1] - ww[0][1] * ww[1][0]) * id;
        }
        // (ww.inverse() * W.transpose()) * pZ, the order the expression is written in
        for (int i = 0; i < 3; ++i) {
            double acc = 0;
            for (int l = 0; l < 4; ++l) { double m = 0; for (int j = 0; j < 3; ++j) m += inv[i][j] * W[l][j]; acc += m * pZ[l]; }
            s.a[i] = acc;
        }
        // GetNormalVector(true)   :151-159
        const double factor = std::sqrt(s.a[1] * s.a[1] + s.a[2] * s.a[2] + 1);
        s.n[0] = -s.a[1] / factor; s.n[1] = -s.a[2] / factor; s.n[2] = 1.0 / factor;
        // ComputeControlFrame()   :162-206: the ground is assumed flat in the world (nInWorldFrame := (0, 0, 1)), so the frame is the base's
        // heading: x = base x axis projected on the horizontal plane
        const M3<double> BaseR = transpose(quaternionToRotationMatrix(quat));
        double x[3] = {BaseR[0][0], BaseR[1][0], BaseR[2][0]}, nW[3] = {0, 0, 1};
        double y[3] = {nW[1] * x[2] - nW[2] * x[1], nW[2] * x[0] - nW[0] * x[2], nW[0] * x[1] - nW[1] * x[0]};
        { const double nn = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]); for (double &v : y) v /= nn; }
        x[0] = y[1] * nW[2] - y[2] * nW[1]; x[1] = y[2] * nW[0] - y[0] * nW[2]; x[2] = y[0] * nW[1] - y[1] * nW[0];
        { const double nn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); for (double &v : x) v /= nn; }
        M3<double> R;
        for (int i = 0; i < 3; ++i) { R[i][0] = x[i]; R[i][1] = y[i]; R[i][2] = nW[i]; }
        const V3<double> newRPY = quat_to_rpy(rotationMatrixToQuaternion(transpose(R)));       // rotationMatrixToRPY(R.transpose())
        const double ratio = 0.8;
        for (int i = 0; i < 3; ++i) s.rpy[i] = (1 - ratio) * s.rpy[i] + ratio * newRPY[i];
        s.rpy[0] = 0;
    }
    V3<double> rpy; for (int i = 0; i < 3; ++i) rpy[i] = s.rpy[i];
    const M3<double> R = transpose(rpyToRotMat(rpy));
    const Q4<double> qcf = rpyToQuat(rpy);
    // stateDataFlow: groundRMat = R.cast<float>(); baseRInControlFrame = groundRMat' * baseRMat (float; baseRMat as qrRobot::UpdateDataFlow, qr_robot.cpp:70-71)
    Q4<float> qf; for (int i = 0; i < 4; ++i) qf[i] = in[19 + i];
    const M3<float> baseR = transpose(quaternionToRotationMatrix(qf));
    float g[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) g[i][j] = (float)R[i][j];
    for (int i = 0; i < 3; ++i) { out[i] = (float)s.a[i]; out[3 + i] = (float)s.n[i]; out[6 + i] = (float)s.rpy[i]; }
    for (int i = 0; i < 4; ++i) out[9 + i] = (float)qcf[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            out[13 + 3 * i + j] = g[i][j];
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) acc += g[k][i] * baseR[k][j];
            out[22 + 3 * i + j] = acc;
        }
    out[31] = upd ? 1.f : 0.f;
}

}  // namespace qro

extern "C" void qro_ground_run(int nticks, const float *in /*[nticks][23]*/, float *out /*[nticks][32]*/)
{
    qro::GroundState s;
    qro::ground_reset(s);
    for (int k = 0; k < nticks; ++k) qro::ground_update(in + 23 * k, s, out + 32 * k);
}
