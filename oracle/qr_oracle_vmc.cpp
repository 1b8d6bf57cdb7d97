// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Force-balance ("VMC") stance QP, SURVEY.md 8f rank 2: restates
//   ComputeMassMatrix / ComputeConstraintMatrix / ComputeObjectiveMatrix / ComputeWeightMatrix / ComputeContactForce
//   QS/controllers/balance_controller/qr_qp_torque_optimizer.cpp:31-57, 60-110, 152-179, 183-187, 190-301  (control-frame overload)
//   and the world-frame overload :304-398 with its ComputeMassMatrix / ComputeConstraintMatrix :401-427, :113-149 -- the same QP with
//   Rcb := rotMat (base -> world), g := (0,0,9.8), normal / tangents := the world axes (TorqueStanceLegController::GetAction passes the
//   identity's columns, qr_torque_stance_leg_controller.cpp:490-498) and per-leg force-window ratios; both overloads return the
//   forces rotated back to the base frame ((X * Rcb)^T :300, RigidTransform(0, quat, X^T) :397)
//   qrRobot::MapContactForceToJointTorques    QS/robots/qr_robot.cpp:241-251
// The QP itself is QuadProg++ in the reference (solve_quadprog, :276) -- pinned by oracle/_ref (tests/golden/vmc_golden.npz).
// The fp32 matrix assembly is Eigen's in the reference and cannot be compiled here: "parity unpinned" at that boundary.  The
// convention shared with the HIP kernel (bit for bit): every inner product is a k-ordered chain acc = fmaf(a_k, b_k, acc) from +0,
// everything else one IEEE fp32 operation in the written order.  With regWeight = 1e-4 against entries of order 10 the QP is
// conditioned like 1e6, so this matters for the internal-force components.
#include "qr_oracle.h"

namespace qro {

static inline float chain3(float a0, float b0, float a1, float b1, float a2, float b2)
{
    float acc = 0.f;
    acc = std::fmaf(a0, b0, acc); acc = std::fmaf(a1, b1, acc); acc = std::fmaf(a2, b2, acc);
    return acc;
}

void vmc_assemble(const VmcConfig &c, const VmcInput &in, float G[144], float a[12], float CI[12 * 24], float b[24])
{
    // Rcb * totalInertia * Rcb^T  (:225); totalInertia is the Eigen column-major map of robot_params.total_inertia
    float I0[3][3], T[3][3], Ic[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) I0[i][j] = c.inertia[i + 3 * j];
    const float (*R)[3] = reinterpret_cast<const float (*)[3]>(in.Rcb);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i][j] = chain3(R[i][0], I0[0][j], R[i][1], I0[1][j], R[i][2], I0[2][j]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ic[i][j] = chain3(T[i][0], R[j][0], T[i][1], R[j][1], T[i][2], R[j][2]);
    // 3x3 inverse by cofactors (:41)
    float cof[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            cof[i][j] = Ic[i1][j1] * Ic[i2][j2] - Ic[i1][j2] * Ic[i2][j1];
        }
    const float det = chain3(Ic[0][0], cof[0][0], Ic[0][1], cof[0][1], Ic[0][2], cof[0][2]);
    const float idet = 1.f / det;
    float Iinv[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Iinv[i][j] = cof[j][i] * idet;
    // massMat 6 x 12 (:48-56); foot positions in the control frame: Rcb * footPosBase (:226)
    float Mm[6][12];
    const float im = 1.f / c.mass;
    for (int l = 0; l < 4; ++l) {
        const float *pb = in.foot_pos_base + 3 * l;
        float x[3];
        for (int i = 0; i < 3; ++i) x[i] = chain3(R[i][0], pb[0], R[i][1], pb[1], R[i][2], pb[2]);
        const float S[3][3] = {{0.f, -x[2], x[1]}, {x[2], 0.f, -x[0]}, {-x[1], x[0], 0.f}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                Mm[i][3 * l + j] = (i == j) ? im : 0.f;
                Mm[3 + i][3 * l + j] = chain3(Iinv[i][0], S[0][j], Iinv[i][1], S[1][j], Iinv[i][2], S[2][j]);
            }
    }
    // quadTerm = (M^T Q) M + regWeight * Ones, + W = 1e-4 I (:166-172, :183-187, :236)
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 6; ++k) acc = std::fmaf(Mm[k][i] * c.acc_weight[k], Mm[k][j], acc);
            float g = acc + c.reg_weight;
            g = g + ((i == j) ? 1e-4f : 0.f);
            G[12 * i + j] = g;
        }
    // linearTerm = ((g + desiredAcc)^T Q) M (:173)
    for (int j = 0; j < 12; ++j) {
        float acc = 0.f;
        for (int k = 0; k < 6; ++k) {
            const float gk = (k < 3) ? in.gvec[k] : 0.f;
            acc = std::fmaf((gk + in.desired_acc[k]) * c.acc_weight[k], Mm[k][j], acc);
        }
        a[j] = acc;
    }
    // constraints (:60-110): rows 2l, 2l+1 normal force window, rows 8+4l.. friction pyramid; CI is A^T (12 x 24)
    for (int i = 0; i < 12 * 24; ++i) CI[i] = 0.f;
    const float fMin = c.fmin_ratio * c.mass * 9.8f, fMax = c.fmax_ratio * c.mass * 9.8f;
    const float *nrm = in.normal;
    const float t2[3] = {0.f, 1.f, 0.f};
    const float t1[3] = {t2[1] * nrm[2] - t2[2] * nrm[1], t2[2] * nrm[0] - t2[0] * nrm[2], t2[0] * nrm[1] - t2[1] * nrm[0]};   // tangent2 x normal
    for (int l = 0; l < 4; ++l) {
        for (int ax = 0; ax < 3; ++ax) {
            CI[(3 * l + ax) * 24 + 2 * l] = nrm[ax];
            CI[(3 * l + ax) * 24 + 2 * l + 1] = -nrm[ax];
            const float mn = c.friction * nrm[ax];
            CI[(3 * l + ax) * 24 + 8 + 4 * l + 0] = mn + t1[ax];
            CI[(3 * l + ax) * 24 + 8 + 4 * l + 1] = mn - t1[ax];
            CI[(3 * l + ax) * 24 + 8 + 4 * l + 2] = mn + t2[ax];
            CI[(3 * l + ax) * 24 + 8 + 4 * l + 3] = mn - t2[ax];
        }
        if (in.contacts[l] > 0.f && in.ratio8) {
            // lb = fMinRatio[leg] * mpcBodyMass * 9.8 (:133-134): float product times the double literal, rounded to float on assignment
            b[2 * l] = (float)((double)(in.ratio8[l] * c.mass) * 9.8); b[2 * l + 1] = (float)((double)(-in.ratio8[4 + l] * c.mass) * 9.8);
        }
        else if (in.contacts[l] > 0.f) { b[2 * l] = fMin; b[2 * l + 1] = -fMax; }
        else { b[2 * l] = 1e-7f; b[2 * l + 1] = 1e-7f; }
        for (int r = 0; r < 4; ++r) b[8 + 4 * l + r] = 0.f;
    }
}

// -> force[12] (3x4, force[3*leg+axis], base frame), x[12] the raw QuadProg solution.  Returns the solver status.
int vmc_solve(const VmcConfig &c, const VmcInput &in, float force[12], double xout[12], QpStats *st)
{
    float G[144], a[12], CI[288], b[24];
    vmc_assemble(c, in, G, a, CI, b);
    double GG[144], aa[12], CC[288], bb[24];
    // GG[i][j] = G(j,i) (:243-247), and QuadProg++'s Cholesky reads only GG[i][j], j >= i (QX/QuadProgpp/src/QuadProg++.cc
    // cholesky_decomposition): the reference's QP is the one of the LOWER triangle of the fp32 G, mirrored.  G itself is not
    // symmetric in fp32 ((M_ki w_k) M_kj vs (M_kj w_k) M_ki) and the problem amplifies that 1e-7 by ~1e4.
    for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) GG[12 * i + j] = (double)G[12 * (i > j ? i : j) + (i > j ? j : i)];
    for (int i = 0; i < 12; ++i) aa[i] = (double)(-a[i]);
    for (int i = 0; i < 288; ++i) CC[i] = (double)CI[i];
    for (int i = 0; i < 24; ++i) bb[i] = (double)(-b[i]);
    double x[12];
    const int rc = qp_solve_gi(12, GG, aa, 0, nullptr, nullptr, 24, CC, bb, x, nullptr, st);
    bool bad = false;
    for (int i = 0; i < 12; ++i) if (std::isnan(x[i])) bad = true;
    float X[4][3];
    for (int l = 0; l < 4; ++l) for (int j = 0; j < 3; ++j) X[l][j] = bad ? 0.f : -(float)x[3 * l + j];    // (:280-297)
    const float (*R)[3] = reinterpret_cast<const float (*)[3]>(in.Rcb);
    for (int l = 0; l < 4; ++l)
        for (int j = 0; j < 3; ++j) force[3 * l + j] = chain3(X[l][0], R[0][j], X[l][1], R[1][j], X[l][2], R[2][j]);   // (X * Rcb)^T (:300)
    if (xout) for (int i = 0; i < 12; ++i) xout[i] = x[i];
    return rc;
}

void vmc_force_to_torque(const LegGeom &geo, const float q[12], const float force[12], float tau[12])
{
    for (int leg = 0; leg < 4; ++leg) {
        float J[9];
        analytical_leg_jacobian(geo, &q[3 * leg], leg, J);
        for (int j = 0; j < 3; ++j) {       // jv^T * contact_force (qr_robot.cpp:244)
            float s = 0.f;
            for (int i = 0; i < 3; ++i) s += J[3 * i + j] * force[3 * leg + i];
            tau[3 * leg + j] = s;
        }
    }
}

}  // namespace qro

extern "C" {
// cfg20: mass, inertia[9], acc_weight[6], reg_weight, friction, fmin_ratio, fmax_ratio;  in37: include/qrgpu.h vmc_in
static void unpack(const float *cfg20, const float *in37, qro::VmcConfig &c, qro::VmcInput &in)
{
    c.mass = cfg20[0];
    for (int i = 0; i < 9; ++i) c.inertia[i] = cfg20[1 + i];
    for (int i = 0; i < 6; ++i) c.acc_weight[i] = cfg20[10 + i];
    c.reg_weight = cfg20[16]; c.friction = cfg20[17]; c.fmin_ratio = cfg20[18]; c.fmax_ratio = cfg20[19];
    for (int i = 0; i < 12; ++i) in.foot_pos_base[i] = in37[i];
    for (int i = 0; i < 6; ++i) in.desired_acc[i] = in37[12 + i];
    for (int i = 0; i < 4; ++i) in.contacts[i] = in37[18 + i];
    for (int i = 0; i < 9; ++i) in.Rcb[i] = in37[22 + i];
    for (int i = 0; i < 3; ++i) { in.gvec[i] = in37[31 + i]; in.normal[i] = in37[34 + i]; }
}
void qro_vmc_assemble(const float *cfg20, const float *in37, const float *ratio8, float *G, float *a, float *CI, float *b)
{
    qro::VmcConfig c; qro::VmcInput in; unpack(cfg20, in37, c, in); in.ratio8 = ratio8;
    qro::vmc_assemble(c, in, G, a, CI, b);
}
int qro_vmc_solve(const float *cfg20, const float *geom3, const float *in37, const float *ratio8, const float *q12, float *force, float *tau, double *x, int *stats4)
{
    qro::VmcConfig c; qro::VmcInput in; unpack(cfg20, in37, c, in); in.ratio8 = ratio8;
    qro::QpStats st;
    const int rc = qro::vmc_solve(c, in, force, x, &st);
    if (q12 && tau) { qro::LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2]; qro::vmc_force_to_torque(g, q12, force, tau); }
    if (stats4) { stats4[0] = st.iters; stats4[1] = st.adds; stats4[2] = st.drops; stats4[3] = st.n_active; }
    return rc;
}
}
