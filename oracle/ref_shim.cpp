// TEST INFRASTRUCTURE ONLY -- never linked into libqrgpu.so, never on the product path.
//
// C-ABI shim around the reference's *vendored* QP solvers, compiled where their
// sources lie under /root/reference (see oracle/Makefile, target `ref`).  Only
// this shim is ours; no reference source is copied into the repository.  The
// resulting oracle/_ref/libqr_ref.so is git-ignored.
//
//  * ref_qpoases_mpc  reproduces the exact solver call of
//    quadruped/src/controllers/mpc/qr_mpc_interface.cpp:428-438
//    (Options::setToMPC, PL_NONE, QProblem::init(H,g,A,NULL,NULL,lbA,ubA,nWSR),
//    getPrimalSolution).
//  * ref_quadprog     reproduces the call of
//    quadruped/src/controllers/wbc/qr_wholebody_impulse_ctrl.cpp:113
//    (quadprogpp::solve_quadprog with n x p / n x m column-constraint matrices,
//    quadruped/extern/QuadProgpp/src/QuadProg++.hh:8-23).
#include <qpOASES.hpp>
#include "QuadProg++.hh"

#include <limits>

#include <string.h>
#include "TinyEKF.h"

extern "C" {

// H: n*n row-major, A: m*n row-major (as EigenToOASES writes them,
// qr_mpc_interface.cpp:127-136).  Returns qpOASES' init() return value;
// *primal_rc receives getPrimalSolution()'s return value (the only one the
// reference checks, :440).
int ref_qpoases_mpc(int n, int m, const double *H, const double *g, const double *A,
                    const double *lbA, const double *ubA, int nWSR_in,
                    double *x_out, int *nWSR_out, int *primal_rc, double *obj_out)
{
    qpOASES::int_t nWSR = nWSR_in;
    qpOASES::QProblem problem(n, m);
    qpOASES::Options option;
    option.setToMPC();
    option.printLevel = qpOASES::PL_NONE;
    problem.setOptions(option);
    // init() takes non-const pointers in 3.2.0 for the dense overload? No: const.
    int rval = problem.init(H, g, A, NULL, NULL, lbA, ubA, nWSR);
    int rval2 = problem.getPrimalSolution(x_out);
    if (nWSR_out) *nWSR_out = (int)nWSR;
    if (primal_rc) *primal_rc = rval2;
    if (obj_out) *obj_out = problem.getObjVal();
    return rval;
}

// G: n*n row-major, CE: n*p row-major (column k = k-th equality), CI: n*m row-major.
// Convention (QuadProg++.hh): CE^T x + ce0 = 0, CI^T x + ci0 >= 0.
// Returns the optimal cost, +inf when infeasible.
double ref_quadprog(int n, int p, int m, const double *G, const double *g0,
                    const double *CE, const double *ce0,
                    const double *CI, const double *ci0, double *x_out)
{
    quadprogpp::Matrix<double> qG, qCE, qCI;
    quadprogpp::Vector<double> qg0, qce0, qci0, qx;
    qG.resize(0., n, n);
    qg0.resize(0., n);
    qCE.resize(0., n, p);
    qce0.resize(0., p);
    qCI.resize(0., n, m);
    qci0.resize(0., m);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) qG[i][j] = G[i * n + j];
        qg0[i] = g0[i];
        for (int j = 0; j < p; ++j) qCE[i][j] = CE[i * p + j];
        for (int j = 0; j < m; ++j) qCI[i][j] = CI[i * m + j];
    }
    for (int j = 0; j < p; ++j) qce0[j] = ce0[j];
    for (int j = 0; j < m; ++j) qci0[j] = ci0[j];
    double f = quadprogpp::solve_quadprog(qG, qg0, qCE, qce0, qCI, qci0, qx);
    for (int i = 0; i < n; ++i) x_out[i] = qx[i];
    return f;
}

// The reference's own TinyEKF<3,3> (QX/TinyEKF/src/TinyEKF.h + tiny_ekf.c), constructed and stepped exactly as
// qrRobotVelocityEstimator does (QS/estimators/qr_robot_velocity_estimator.cpp:42, :104-109): nsteps calls of step(deltaV, z).
// x_out: [nsteps][3] state after each step.
int ref_tinyekf_run(float accelerometerVariance, float sensorVariance, int nsteps, const double *deltaV, const double *z, double *x_out)
{
    TinyEKF<3, 3> f(0.f, 0.f, accelerometerVariance, sensorVariance);
    int bad = 0;
    for (int k = 0; k < nsteps; ++k) {
        double dv[3] = {deltaV[3 * k], deltaV[3 * k + 1], deltaV[3 * k + 2]}, zz[3] = {z[3 * k], z[3 * k + 1], z[3 * k + 2]};
        if (!f.step(dv, zz)) ++bad;
        for (int i = 0; i < 3; ++i) x_out[3 * k + i] = f.getX(i);
    }
    return bad;
}

}  // extern "C"
