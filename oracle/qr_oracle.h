// ============================================================================
//  qr_oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
//  Plain-C++ CPU restatement of the convex-MPC + WBC hot path of
//  TopHillRobotics/quadruped-robot (reference files cited per function as
//  "QS/..." = quadruped/src/..., "QI/..." = quadruped/include/quadruped/...).
//  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
//  load this library; libqrgpu.so never links or calls it.
//
//  Parity status: the reference has NO tests / golden vectors for this path
//  (SURVEY.md 4).  The QP solvers are pinned against the reference's own
//  vendored qpOASES 3.2.0 / QuadProg++ compiled from /root/reference
//  (oracle/_ref, tests/golden/*).  The Eigen glue around them (fp32 assembly,
//  rigid-body dynamics) cannot be compiled here (Eigen/ROS absent) and is
//  pinned only by analytic invariants and float64 numpy cross-checks:
//  "parity unpinned" at the Eigen boundary.
// ============================================================================
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cassert>

namespace qro {

// ---------------------------------------------------------------------------
// Tiny dense row-major matrix (stands in for Eigen DMat<T>/DVec<T>).
// All arithmetic is done in T with plain, left-to-right accumulation.
// ---------------------------------------------------------------------------
template <typename T>
struct Mat {
    int r = 0, c = 0;
    std::vector<T> d;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, T(0)) {}
    static Mat Identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = T(1); return m; }
    static Mat Zero(int r_, int c_) { return Mat(r_, c_); }
    T &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    const T &operator()(int i, int j) const { return d[(size_t)i * c + j]; }
    T &operator[](int i) { return d[i]; }                 // vector access
    const T &operator[](int i) const { return d[i]; }
    int rows() const { return r; }
    int cols() const { return c; }
    int size() const { return r * c; }
    void setZero() { std::fill(d.begin(), d.end(), T(0)); }
    Mat t() const { Mat m(c, r); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m(j, i) = (*this)(i, j); return m; }
    Mat block(int i0, int j0, int nr, int nc) const {
        Mat m(nr, nc);
        for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) m(i, j) = (*this)(i0 + i, j0 + j);
        return m;
    }
    void setBlock(int i0, int j0, const Mat &b) {
        for (int i = 0; i < b.r; ++i) for (int j = 0; j < b.c; ++j) (*this)(i0 + i, j0 + j) = b(i, j);
    }
};

template <typename T> Mat<T> operator*(const Mat<T> &a, const Mat<T> &b) {
    assert(a.c == b.r);
    Mat<T> m(a.r, b.c);
    for (int i = 0; i < a.r; ++i)
        for (int j = 0; j < b.c; ++j) {
            T s = T(0);
            for (int k = 0; k < a.c; ++k) s += a(i, k) * b(k, j);
            m(i, j) = s;
        }
    return m;
}
template <typename T> Mat<T> operator+(const Mat<T> &a, const Mat<T> &b) {
    assert(a.r == b.r && a.c == b.c);
    Mat<T> m(a.r, a.c);
    for (int i = 0; i < a.size(); ++i) m.d[i] = a.d[i] + b.d[i];
    return m;
}
template <typename T> Mat<T> operator-(const Mat<T> &a, const Mat<T> &b) {
    assert(a.r == b.r && a.c == b.c);
    Mat<T> m(a.r, a.c);
    for (int i = 0; i < a.size(); ++i) m.d[i] = a.d[i] - b.d[i];
    return m;
}
template <typename T> Mat<T> operator-(const Mat<T> &a) {
    Mat<T> m(a.r, a.c);
    for (int i = 0; i < a.size(); ++i) m.d[i] = -a.d[i];
    return m;
}
template <typename T> Mat<T> operator*(T s, const Mat<T> &a) {
    Mat<T> m(a.r, a.c);
    for (int i = 0; i < a.size(); ++i) m.d[i] = s * a.d[i];
    return m;
}

// ------------------------- small fixed-size helpers ------------------------
template <typename T> struct V3 { T v[3]; T &operator[](int i) { return v[i]; } const T &operator[](int i) const { return v[i]; } };
template <typename T> struct M3 { T m[3][3]; T *operator[](int i) { return m[i]; } const T *operator[](int i) const { return m[i]; } };
template <typename T> struct Q4 { T q[4]; T &operator[](int i) { return q[i]; } const T &operator[](int i) const { return q[i]; } };  // (w,x,y,z)

// QI/utils/qr_se3.h restatements (qr_oracle_math.cpp)
template <typename T> M3<T> coordinateRotation(int axis, T theta);               // :72-89 (returns the TRANSPOSED / coordinate-transform matrix)
template <typename T> M3<T> rpyToRotMat(const V3<T> &rpy);                       // :109-116
template <typename T> Q4<T> rotationMatrixToQuaternion(const M3<T> &r1);         // :139-172
template <typename T> M3<T> quaternionToRotationMatrix(const Q4<T> &q);          // :186-203 (world->body)
template <typename T> Q4<T> rpyToQuat(const V3<T> &rpy);                         // :229-235
template <typename T> Q4<T> quatProduct(const Q4<T> &a, const Q4<T> &b);         // :291-302
template <typename T> V3<T> quaternionToso3(const Q4<T> &q);                     // :383-397
template <typename T> M3<T> mul(const M3<T> &a, const M3<T> &b);
template <typename T> V3<T> mul(const M3<T> &a, const V3<T> &b);
template <typename T> M3<T> transpose(const M3<T> &a);

// QI/utils/qr_algebra.h:119-141 -- SVD pseudo-inverse with strict '>' threshold.
template <typename T> void pseudoInverse(const Mat<T> &m, double sigmaThreshold, Mat<T> &inv);
// Thin SVD  A = U diag(s) V^T  by one-sided (Hestenes) Jacobi; stands in for Eigen::JacobiSVD.
template <typename T> void jacobiSVD(const Mat<T> &A, Mat<T> &U, std::vector<T> &s, Mat<T> &V);
// Partial-pivot LU inverse; stands in for Eigen dynamic .inverse() (qr_wholebody_impulse_ctrl.cpp:55).
template <typename T> Mat<T> luInverse(const Mat<T> &A);

// ---------------------------------------------------------------------------
// Dense strictly-convex QP, Goldfarb-Idnani dual active set, double precision.
//   min 1/2 x'Gx + g0'x   s.t.  CE'x + ce0 = 0,  CI'x + ci0 >= 0
// (the QuadProg++ convention, QX/QuadProgpp/src/QuadProg++.hh:8-23; CE is n x p,
// CI is n x m, row-major).  Returns 0 ok, 1 infeasible, 2 iteration cap.
// ---------------------------------------------------------------------------
struct QpStats { int iters = 0, adds = 0, drops = 0, n_active = 0; double obj = 0; };
int qp_solve_gi(int n, const double *G, const double *g0, int p, const double *CE, const double *ce0,
                int m, const double *CI, const double *ci0, double *x, double *lambda_ineq /*m or null*/,
                QpStats *st, int max_iter = 0, double abs_tol = 0.0);

// ---------------------------------------------------------------------------
// MPC (K1-K7).  QS/controllers/mpc/qr_mpc_interface.cpp, qr_mpc_stance_leg_controller.cpp
// ---------------------------------------------------------------------------
constexpr int kMaxHorizon = 16;   // K_MAX_GAIT_SEGMENTS, QI/controllers/mpc/qr_mpc_interface.h:33

struct MpcConfig {                // ProblemConfig + body inertia (qr_mpc_interface.h:104-144)
    float dt = 0.06f; int horizon = 10; float mu = 0.45f; float fmax = 13.f * 9.81f;
    float mass = 13.f; float inertia[3] = {0.24f, 0.80f, 1.0f};
    float weights[12] = {10, 10, 5, 40, 60, 100, 0, 0, 0.5f, 5, 5, 1}; float alpha = 4e-6f;
};
struct MpcInput {                 // arguments of SolveMPCKernel (qr_mpc_interface.h:200)
    float p[3], v[3], quat[4] /*wxyz*/, w[3], r[12] /*3x4 column-major: r[3*leg+axis]*/, rpy[3];
    float traj[12 * kMaxHorizon]; float gait[4 * kMaxHorizon];
};
struct MpcAssembly {              // fp32 QP data exactly as SolveMPC builds it (:359-425)
    int n = 0, m = 0;
    std::vector<float> H, g;      // n*n row-major, n
    std::vector<float> ub;        // m (lb = 0)
    float invmu = 0;
};
void mpc_assemble(const MpcConfig &cfg, const MpcInput &in, MpcAssembly &out);
// Literal variant of ConvertToDiscreteQP (:257-293): fp32 Pade expm + repeated products.
// Used only to quantify how far the closed form is from a literal evaluation.
void mpc_assemble_literal(const MpcConfig &cfg, const MpcInput &in, MpcAssembly &out);
// Solve the assembled QP (double, own GI solver on the swing-eliminated problem).
int mpc_solve_qp(const MpcAssembly &a, const float *gait, int horizon, double *u_out /*n*/, QpStats *st);
// K7: force -> (f_ff, torque).  qr_mpc_stance_leg_controller.cpp:402-409,139-153; QS/robots/qr_robot.cpp:148-172,241-251
struct LegGeom { float hip_l = 0.08505f, upper_l = 0.2f, lower_l = 0.2f; };
void mpc_force_to_torque(const LegGeom &geo, const float quat[4], const float q[12], const double f_world[12], float tau[12]);
void swing_velocity_mode(const struct LegGeom &geo, const float hip_offset[12], const float desc[20], const float in[53], float out[48]);
void analytical_leg_jacobian(const LegGeom &geo, const float q[3], int leg, float J[9] /*row-major*/);
void foot_positions_in_base_frame(const LegGeom &geo, const float hipOffset[12], const float q[12], float out[12]); // QS/robots/qr_robot.cpp:127-146,175-184

// ---------------------------------------------------------------------------
// Floating-base model (K8-K10).  QS/dynamics/floating_base_model.cpp, QI/dynamics/spatial.hpp,
// constants QS/robots/qr_robot_a1_sim.cpp:176-343
// ---------------------------------------------------------------------------
struct ModelDesc {                // what BuildDynamicModel hard-codes / reads from YAML
    float hip_l = 0.08505f, upper_l = 0.2f, lower_l = 0.2f;
    float body_size[3] = {0.267f, 0.194f, 0.114f};
};
template <typename T> struct FBModel;          // defined in qr_oracle_fbmodel.cpp
template <typename T> struct FBState { T quat[4]; T pos[3]; T bodyVel[6]; T q[12]; T qd[12]; };
template <typename T> struct FBResult {
    Mat<T> H{18, 18}; Mat<T> G{18, 1}; Mat<T> C{18, 1};
    Mat<T> Jc[4]; Mat<T> Jcdqd[4]; T pGC[4][3]; T vGC[4][3];   // feet only (ids 9,11,13,15)
    T totalNonRotorMass = 0;
};
template <typename T> void fb_compute(const ModelDesc &md, const FBState<T> &st, FBResult<T> &out);

// ---------------------------------------------------------------------------
// WBC (K11-K14).  QS/controllers/wbc/*.cpp
// ---------------------------------------------------------------------------
template <typename T> struct WbcCmd {          // qrWbcCtrlData, QI/controllers/qr_state_dataflow.h:133-192
    T pBody_des[3], vBody_des[3], aBody_des[3], pBody_RPY_des[3], vBody_Ori_des[3];
    T pFoot_des[4][3], vFoot_des[4][3], aFoot_des[4][3], Fr_des[4][3];
    int contact[4];
};
template <typename T> struct WbcOut {
    T tau[12];        // jointTorqueCmd
    T qdes[12], qddes[12];   // desiredJPos / desiredJVel (K12)
    T fr[12];         // optimalFr, stance feet in contact order, zero padded
    T qddot[18];
    int qp_status; QpStats qp;
};
// The relaxation QP exactly as MakeTorque hands it to solve_quadprog (qr_wholebody_impulse_ctrl.cpp:113), QuadProg++ convention, the
// matrices row-major [n][p] / [n][m] as qpCE[j][i] / qpCI[j][i] are filled (:141-147, :161-166).  When `z_in` is set the tick is finished
// with that solution instead of the oracle's own solver's (tests: the compiled QuadProg++ of oracle/_ref).
struct WbcQpIO {
    int n = 0, p = 0, m = 0;
    std::vector<double> G, g0, CE, ce0, CI, ci0, z;
    const double *z_in = nullptr;
};
// prev_ori_vel: in/out, TK::desiredVel of the orientation task from the previous call (quirk 4).
template <typename T> void wbc_run(const ModelDesc &md, const FBState<T> &st, const WbcCmd<T> &cmd,
                                   T prev_ori_vel[3], WbcOut<T> &out, WbcQpIO *qpio = nullptr);


// ---------------------------------------------------------------------------
// Force-balance (VMC) stance QP.  qr_oracle_vmc.cpp
// ---------------------------------------------------------------------------
struct VmcConfig {                // ComputeContactForce defaults (QI/controllers/balance_controller/qr_qp_torque_optimizer.h:144-153)
    float mass = 13.f; float inertia[9] = {0.24f, 0, 0, 0, 0.80f, 0, 0, 0, 1.0f};
    float acc_weight[6] = {1, 1, 1, 10, 10, 1}; float reg_weight = 1e-4f, friction = 0.5f, fmin_ratio = 0.01f, fmax_ratio = 10.f;
};
struct VmcInput {                 // per tick
    float foot_pos_base[12];      // 3*leg+axis, GetFootPositionsInBaseFrame
    float desired_acc[6];         // KP/KD output (ddqDes.head(6))
    float contacts[4];
    float Rcb[9];                 // row-major; identity on PLANE / PLUM_PILES terrain (:217-223)
    float gvec[3];                // g.head(3): (0,0,9.8) on a plane
    float normal[3];              // surfaceNormal
    // world-frame overload (:304-398): per-leg force-window ratios fMinRatio[4], fMaxRatio[4] (Vec4 arguments); null = the scalar ones
    const float *ratio8 = nullptr;
};
void vmc_assemble(const VmcConfig &c, const VmcInput &in, float G[144], float a[12], float CI[12 * 24], float b[24]);
int vmc_solve(const VmcConfig &c, const VmcInput &in, float force[12], double xout[12], QpStats *st);

// ---------------------------------------------------------------------------
// Base velocity estimator + leg kinematics (first part).  qr_oracle_estimator.cpp
// ---------------------------------------------------------------------------
struct EstimatorConfig {
    float hip_l = 0.08505f, upper_l = 0.2f, lower_l = 0.2f;
    float hip_offset[12] = {0.1805f, -0.047f, 0, 0.1805f, 0.047f, 0, -0.1805f, -0.047f, 0, -0.1805f, 0.047f, 0};
    float time_step = 0.002f;                      // robot->timeStep
    float accelerometer_variance = 0.1f, sensor_variance = 0.1f;
    int window = 60;                               // movingWindowFilterSize
    float body_height = 0.28f;                     // robot->bodyHeight (all feet in the air)
};
struct EstimatorState {
    unsigned last_timestamp = 0;
    double x[3] = {0, 0, 0}, P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float est_vel_base[3] = {0, 0, 0};
    float pose_x = 0, pose_y = 0, pose_theta = 0, abs_height = 0;     // qrRobotPoseEstimator: estimatedPose[0,1,5], robot->absoluteHight
    std::vector<double> vel_win[3]; double vel_sum[3] = {0, 0, 0}, vel_corr[3] = {0, 0, 0}; int vel_count = 0, vel_head = 0;
    float acc_win[3][20]; float acc_sum[3] = {0, 0, 0}, acc_corr[3] = {0, 0, 0}; int acc_count = 0, acc_head = 0;
    explicit EstimatorState(int W) { for (auto &w : vel_win) w.assign(W, 0.0); std::memset(acc_win, 0, sizeof(acc_win)); }
};
int ekf3_step(double x[3], double P[9], double qvar, double rvar, const double deltaV[3], const double z[3]);
void estimator_update(const EstimatorConfig &cfg, const float in[54], unsigned tick, EstimatorState &s, float out[42]);

// Ground-plane fit + control frame (qr_oracle_ground.cpp): qrGroundSurfaceEstimator, QS/estimators/qr_ground_surface_estimator.cpp:40-70,151-206
struct GroundState { bool last_contact[4]; double a[3], n[3], rpy[3]; };
void ground_reset(GroundState &s);
void ground_update(const float in[23], GroundState &s, float out[32]);

// Walk gait generator + the force-window ratios of its sub-states (qr_oracle_walk.cpp): qrWalkGaitGenerator, QS/gait/qr_walk_gait_generator.cpp;
// TorqueStanceLegController::UpdateFRatio (walk branch), QS/controllers/balance_controller/qr_torque_stance_leg_controller.cpp:125-168
struct WalkConfig {                                   // config/a1_sim/openloop_gait_generator.yaml, gait "walk"
    float stance_duration[4], duty_factor[4], initial_leg_phase[4]; int initial_leg_state[4];
    float contact_detection_phase_threshold; int n_states; int state_switch[4]; float state_ratio[4];
};
struct WalkDerived { int nq; int que[4]; float ratio[4], accum[5], true_swing_start_in_swing, full[4]; int state_index0[4]; };
struct WalkState { int cur[4], desired[4], leg[4], detected[4], state_index[4]; float phase[4], nphase[4], event_phase[4], move_base_phase; };
void walk_derive(const WalkConfig &c, WalkDerived &d);
void walk_reset(const WalkConfig &c, const WalkDerived &d, WalkState &s, bool constructed);
void walk_update(const WalkConfig &c, const WalkDerived &d, float currentTime, const float contact[4], bool stop, WalkState &s, float out[41]);

// Open-loop gait generator (qr_oracle_gait.cpp).  LegState: SWING 0, STANCE 1, EARLY_CONTACT 2.
struct GaitConfig {                                   // config/a1_sim/openloop_gait_generator.yaml, gait "advanced_trot"
    float stance_duration[4] = {0.5f, 0.5f, 0.5f, 0.5f}, duty_factor[4] = {0.6f, 0.6f, 0.6f, 0.6f}, initial_leg_phase[4] = {0.5f, 0.f, 0.f, 0.5f};
    int initial_leg_state[4] = {1, 1, 1, 1};
    float contact_detection_phase_threshold = 0.5f, wait_time = 1.0f;
    bool advanced_trot = true;
};
struct GaitState {
    float reset_time = 0, last_time = 0, cum_dt = 0, gait_cycle = 0;
    int cur[4] = {0, 0, 0, 0}, last[4] = {0, 0, 0, 0}, desired[4] = {0, 0, 0, 0}, leg[4] = {0, 0, 0, 0}, allow[4] = {1, 1, 1, 1};
    int first_swing[4] = {0, 0, 0, 0}, first_stance[4] = {0, 0, 0, 0};
    float phase[4] = {0, 0, 0, 0}, nphase[4] = {0, 0, 0, 0}, contact_start_phase[4] = {0, 0, 0, 0}, swing_remaining[4] = {0, 0, 0, 0};
};
void gait_reset(const GaitConfig &c, GaitState &s);
void gait_update(const GaitConfig &c, float currentTime, const float contact[4], bool stop, GaitState &s, float out[24]);

// Swing-leg targets, ADVANCED_TROT on horizontal terrain (second part).  qr_oracle_swing.cpp
void swing_targets(const LegGeom &geo, const float hip_offset[12], const float in[58], float out[72]);

// ---------------------------------------------------------------------------
// MPC front-end.  qr_oracle_frontend.cpp
// in[64] = des_height, des_roll, des_pitch, x_vel_cmd, y_vel_cmd, yaw_vel_cmd, basePosition[3], yawCurrent,
//          quat_wxyz[4], footPosWorld[12] (leg major), footTargetWorld[12], contacts[4], phaseInFullCycle[4],
//          dutyFactor[4], normalizedPhase[4], desiredLegState[4], legState[4], firstSwingBaseState x, y
// st[8]  = xVelDes, yVelDes, yawTurnRate, yawDesTrue, posDesiredinWorld[3], iterationCounter   (in/out)
// ---------------------------------------------------------------------------
void mpc_frontend(int horizon, int numHorizonL, float dt, float dtMPC, const float in[64], float st[8], float *traj, float *gait, float *wbc15,
                  float contact_out[4], int *mpc_updated);

}  // namespace qro
