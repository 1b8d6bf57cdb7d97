// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Whole-body controller tick, restating
//   QS/controllers/wbc/qr_wbc_locomotion_controller.cpp:108-219  (Run / UpdateModel / ContactTaskUpdate)
//   QS/controllers/wbc/task_set/qr_task_body_orientation.cpp:43-98
//   QS/controllers/wbc/task_set/qr_task_body_position.cpp:43-80
//   QS/controllers/wbc/task_set/qr_task_link_position.cpp:45-85
//   QS/controllers/wbc/qr_single_contact.cpp:29-111
//   QS/controllers/wbc/qr_multitask_projection.cpp:38-124
//   QS/controllers/wbc/qr_wholebody_impulse_ctrl.cpp:50-299
// One call = one *computing* WBC tick (the reference's every-2nd-call cadence,
// :111, is the caller's business).
#include "qr_oracle.h"

namespace qro {

namespace {

template <typename T> struct Task {        // qrTask<T>, QI/controllers/wbc/task_set/qr_task.hpp
    Mat<T> Jt{3, 18}, JtDotQdot{3, 1}, xddotCmd{3, 1}, posErr{3, 1}, desiredVel{3, 1};
};
template <typename T> struct Contact {     // qrSingleContact<T>
    Mat<T> Jc{3, 18}, JcDotQdot{3, 1}, Uf{6, 3}, ineqVec{6, 1}, desiredFr{3, 1};
};

template <typename T> void weightedInverse(const Mat<T> &J, const Mat<T> &Winv, Mat<T> &Jinv, double threshold = 0.0001)
{   // qr_wholebody_impulse_ctrl.cpp:291-299
    Mat<T> temp = Winv * J.t();
    Mat<T> lambda = J * temp;
    Mat<T> lambda_inv;
    pseudoInverse(lambda, threshold, lambda_inv);
    Jinv = temp * lambda_inv;
}

}  // namespace

template <typename T>
void wbc_run(const ModelDesc &md, const FBState<T> &st, const WbcCmd<T> &cmd, T prev_ori_vel[3], WbcOut<T> &out, WbcQpIO *qpio)
{
    // ---- UpdateModel (:138-168)
    FBResult<T> fb;
    fb_compute(md, st, fb);
    const Mat<T> &A = fb.H;
    const Mat<T> &Grav = fb.G, &Cori = fb.C;
    Mat<T> Ainv = luInverse(A);                              // GetModelRes, qr_wholebody_impulse_ctrl.cpp:50-58
    const Mat<T> I18 = Mat<T>::Identity(18);

    Q4<T> quat = {{st.quat[0], st.quat[1], st.quat[2], st.quat[3]}};
    M3<T> Rot = quaternionToRotationMatrix(quat);            // world -> base
    M3<T> RotT = transpose(Rot);

    // ---- ContactTaskUpdate (:172-201)
    std::vector<Task<T>> tasks;
    std::vector<Contact<T>> contacts;
    {   // body orientation task (qr_task_body_orientation.cpp)
        Task<T> tk;
        V3<T> rpy = {{cmd.pBody_RPY_des[0], cmd.pBody_RPY_des[1], cmd.pBody_RPY_des[2]}};
        Q4<T> ori_cmd = rpyToQuat(rpy);                                              // :180
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tk.Jt(i, j) = RotT[i][j];   // UpdateTaskJacobian :84-90
        Q4<T> inv = {{quat[0], -quat[1], -quat[2], -quat[3]}};
        Q4<T> err = quatProduct(ori_cmd, inv);
        if (err[0] < T(0.)) for (int i = 0; i < 4; ++i) err[i] *= T(-1.);
        V3<T> so3 = quaternionToso3(err);
        // vel_err uses LAST call's desiredVel (quirk 4) and a body-frame omega (:68)
        V3<T> dv = {{prev_ori_vel[0] - st.bodyVel[0], prev_ori_vel[1] - st.bodyVel[1], prev_ori_vel[2] - st.bodyVel[2]}};
        V3<T> vel_err = mul(RotT, dv);
        const T Kp = T(100.), Kd = T(10.);                                           // :66-67 of the controller ctor
        for (int i = 0; i < 3; ++i) {
            tk.posErr[i] = T(1.) * so3[i];
            tk.desiredVel[i] = cmd.vBody_Ori_des[i];
            T acc = T(0);                                                            // des_acc = zeroVec3 (:182)
            T x = Kp * so3[i] + Kd * vel_err[i] + acc;
            tk.xddotCmd[i] = std::min(std::max(x, T(-10)), T(10));
        }
        for (int i = 0; i < 3; ++i) prev_ori_vel[i] = tk.desiredVel[i];
        tasks.push_back(tk);
    }
    {   // body position task (qr_task_body_position.cpp)
        Task<T> tk;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tk.Jt(i, 3 + j) = RotT[i][j];
        V3<T> vb = {{st.bodyVel[3], st.bodyVel[4], st.bodyVel[5]}};
        V3<T> vw = mul(RotT, vb);                                                    // :54
        const T Kp = T(100.), Kd = T(10.);
        for (int i = 0; i < 3; ++i) {
            tk.posErr[i] = T(1.) * (cmd.pBody_des[i] - st.pos[i]);
            tk.desiredVel[i] = cmd.vBody_des[i];
            T x = Kp * (cmd.pBody_des[i] - st.pos[i]) + Kd * (tk.desiredVel[i] - vw[i]) + cmd.aBody_des[i];
            tk.xddotCmd[i] = std::min(std::max(x, T(-10)), T(10));
        }
        tasks.push_back(tk);
    }
    const T maxFz = fb.totalNonRotorMass * T(9.81);                                  // qr_single_contact.cpp:31
    const T mu = T(0.4f);                                                            // :34
    for (int leg = 0; leg < 4; ++leg) {
        if (cmd.contact[leg]) {
            Contact<T> ct;
            ct.Jc = fb.Jc[leg];
            ct.JcDotQdot = fb.Jcdqd[leg];
            ct.Uf(0, 2) = T(1.);
            ct.Uf(1, 0) = T(1.);  ct.Uf(1, 2) = mu;
            ct.Uf(2, 0) = T(-1.); ct.Uf(2, 2) = mu;
            ct.Uf(3, 1) = T(1.);  ct.Uf(3, 2) = mu;
            ct.Uf(4, 1) = T(-1.); ct.Uf(4, 2) = mu;
            ct.Uf(5, 2) = T(-1.);
            ct.ineqVec[5] = -maxFz;
            for (int i = 0; i < 3; ++i) ct.desiredFr[i] = cmd.Fr_des[leg][i];
            contacts.push_back(ct);
        } else {   // link position task (qr_task_link_position.cpp), virtualDepend = false
            Task<T> tk;
            tk.Jt = fb.Jc[leg];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) tk.Jt(i, j) = T(0);
            tk.JtDotQdot = fb.Jcdqd[leg];
            const T Kp = T(500), Kd = T(10.);
            for (int i = 0; i < 3; ++i) {
                tk.posErr[i] = T(1.) * (cmd.pFoot_des[leg][i] - fb.pGC[leg][i]);
                tk.desiredVel[i] = cmd.vFoot_des[leg][i];
            }
            for (int i = 0; i < 3; ++i) {
                T v_error = tk.desiredVel[i] - fb.vGC[leg][i];
                tk.xddotCmd[i] = Kp * tk.posErr[i] + Kd * v_error + cmd.aFoot_des[leg][i];
            }
            tasks.push_back(tk);
        }
    }
    const int nc = (int)contacts.size();
    const int dimFr = 3 * nc, dimUf = 6 * nc;

    // ---- qrMultitaskProjection::FindConfiguration (qr_multitask_projection.cpp:38-106)
    {
        const double thr = 0.001;
        Mat<T> Nc = I18;
        if (nc > 0) {
            Mat<T> Jc(dimFr, 18);
            for (int k = 0; k < nc; ++k) Jc.setBlock(3 * k, 0, contacts[k].Jc);
            Mat<T> Jp; pseudoInverse(Jc, thr, Jp);
            Nc = I18 - Jp * Jc;
        }
        Mat<T> JtPre = tasks[0].Jt * Nc, JtPre_pinv;
        pseudoInverse(JtPre, thr, JtPre_pinv);
        Mat<T> delta_q = JtPre_pinv * tasks[0].posErr;
        Mat<T> qdot = JtPre_pinv * tasks[0].desiredVel;
        Mat<T> prev_delta_q = delta_q, prev_qdot = qdot;
        Mat<T> N_nx = I18 - JtPre_pinv * JtPre;
        Mat<T> N_pre = Nc * N_nx;
        for (size_t i = 1; i < tasks.size(); ++i) {
            const Task<T> &tk = tasks[i];
            JtPre = tk.Jt * N_pre;
            pseudoInverse(JtPre, thr, JtPre_pinv);
            delta_q = prev_delta_q + JtPre_pinv * (tk.posErr - tk.Jt * prev_delta_q);
            qdot = prev_qdot + JtPre_pinv * (tk.desiredVel - tk.Jt * prev_qdot);
            if (i < tasks.size() - 1) {
                Mat<T> Jp2; pseudoInverse(JtPre, thr, Jp2);      // BuildProjectionMatrix recomputes the pinv (:110-114)
                N_nx = I18 - Jp2 * JtPre;
                N_pre = N_pre * N_nx;
                prev_delta_q = delta_q;
                prev_qdot = qdot;
            }
        }
        for (int i = 0; i < 12; ++i) {
            out.qdes[i] = st.q[i] + delta_q[6 + i];
            out.qddes[i] = qdot[6 + i];
        }
    }

    // ---- qrWholeBodyImpulseCtrl::MakeTorque (qr_wholebody_impulse_ctrl.cpp:62-126)
    const int dimOpt = 6 + dimFr;
    Mat<T> JC(dimFr, 18), JCDotQdot(dimFr, 1), desiredFr(dimFr, 1), UF(dimUf, dimFr), ineqVec(dimUf, 1);
    Mat<T> qddot_pre(18, 1), Npre = I18;
    if (dimFr > 0) {
        for (int k = 0; k < nc; ++k) {                  // ContactBuilding :171-206
            JC.setBlock(3 * k, 0, contacts[k].Jc);
            for (int i = 0; i < 3; ++i) { JCDotQdot[3 * k + i] = contacts[k].JcDotQdot[i]; desiredFr[3 * k + i] = contacts[k].desiredFr[i]; }
            UF.setBlock(6 * k, 3 * k, contacts[k].Uf);
            for (int i = 0; i < 6; ++i) ineqVec[6 * k + i] = contacts[k].ineqVec[i];
        }
        Mat<T> JcBar;
        weightedInverse(JC, Ainv, JcBar);
        qddot_pre = JcBar * (-JCDotQdot);
        Npre = I18 - JcBar * JC;
    }
    for (size_t i = 0; i < tasks.size(); ++i) {         // :96-109
        const Task<T> &tk = tasks[i];
        Mat<T> JtPre = tk.Jt * Npre, JtBar;
        weightedInverse(JtPre, Ainv, JtBar);
        qddot_pre = qddot_pre + JtBar * (tk.xddotCmd - tk.JtDotQdot - tk.Jt * qddot_pre);
        if (i < tasks.size() - 1) Npre = Npre * (I18 - JtBar * JtPre);
    }

    // QP in QuadProg++ form, double (SetCost :232-247, SetEqualityConstraint :129-148,
    // SetInequalityConstraint :152-167, SetOptimizationSize :251-287)
    const int p = 6, m = dimFr > 0 ? dimUf : 1;
    std::vector<double> G((size_t)dimOpt * dimOpt, 0.0), g0(dimOpt, 0.0), CE((size_t)dimOpt * p, 0.0), ce0(p, 0.0),
        CI((size_t)dimOpt * m, 0.0), ci0(m, 0.0), z(dimOpt, 0.0);
    for (int i = 0; i < 6; ++i) G[(size_t)i * dimOpt + i] = (double)T(0.1);     // weightFb (:44 of the controller)
    for (int i = 0; i < dimFr; ++i) G[(size_t)(6 + i) * dimOpt + 6 + i] = (double)T(1);
    {
        Mat<T> CEm(6, dimOpt), ce0m(6, 1);
        CEm.setBlock(0, 0, A.block(0, 0, 6, 6));
        Mat<T> gen = A * qddot_pre + Cori + Grav;
        if (dimFr > 0) {
            Mat<T> JCt = JC.t();
            CEm.setBlock(0, 6, -(JCt.block(0, 0, 6, dimFr)));        // -Sf * JC^T
            gen = gen - JCt * desiredFr;
        }
        for (int i = 0; i < 6; ++i) ce0m[i] = -gen[i];               // ce0 = -Sf * (...)
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < dimOpt; ++j) CE[(size_t)j * p + i] = (double)CEm(i, j);
            ce0[i] = -(double)ce0m[i];
        }
    }
    if (dimFr > 0) {
        Mat<T> CIm(dimUf, dimOpt);
        CIm.setBlock(0, 6, UF);
        Mat<T> ci0m = ineqVec - UF * desiredFr;
        for (int i = 0; i < dimUf; ++i) {
            for (int j = 0; j < dimOpt; ++j) CI[(size_t)j * m + i] = (double)CIm(i, j);
            ci0[i] = -(double)ci0m[i];
        }
    }
    if (qpio && qpio->z_in) {        // finish the tick with somebody else's solution of this QP
        for (int i = 0; i < dimOpt; ++i) z[i] = qpio->z_in[i];
        out.qp_status = 0; out.qp = QpStats();
    } else {
        out.qp_status = qp_solve_gi(dimOpt, G.data(), g0.data(), p, CE.data(), ce0.data(), m, CI.data(), ci0.data(), z.data(), nullptr, &out.qp);
    }
    if (qpio) { qpio->n = dimOpt; qpio->p = p; qpio->m = m; qpio->G = G; qpio->g0 = g0; qpio->CE = CE; qpio->ce0 = ce0; qpio->CI = CI; qpio->ci0 = ci0; qpio->z = z; }

    for (int i = 0; i < 6; ++i) qddot_pre[i] += (T)z[i];             // :117-119

    // GetSolution (:210-228)
    Mat<T> tot_tau = A * qddot_pre + Cori + Grav;
    for (int i = 0; i < 12; ++i) out.fr[i] = T(0);
    if (dimFr > 0) {
        Mat<T> optimalFr(dimFr, 1);
        for (int i = 0; i < dimFr; ++i) { optimalFr[i] = (T)z[6 + i] + desiredFr[i]; out.fr[i] = optimalFr[i]; }
        tot_tau = tot_tau - JC.t() * optimalFr;
    }
    for (int i = 0; i < 12; ++i) out.tau[i] = tot_tau[6 + i];
    for (int i = 0; i < 18; ++i) out.qddot[i] = qddot_pre[i];
}

template void wbc_run<float>(const ModelDesc &, const FBState<float> &, const WbcCmd<float> &, float *, WbcOut<float> &, WbcQpIO *);
template void wbc_run<double>(const ModelDesc &, const FBState<double> &, const WbcCmd<double> &, double *, WbcOut<double> &, WbcQpIO *);

}  // namespace qro
