// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Convex-MPC QP assembly and solve, following
//   QS/controllers/mpc/qr_mpc_interface.cpp            (K1-K6)
//   QS/controllers/mpc/qr_mpc_stance_leg_controller.cpp:385-410,129-156   (K7)
//   QS/robots/qr_robot.cpp:127-184,241-251             (leg kinematics for K7)
//
// Arithmetic contract (the HIP kernels replay exactly this fp32 sequence, see
// DESIGN.md "bit-exact assembly"):
//   * every inner product is a k-ordered chain  acc = fmaf(a_k, b_k, acc)
//     starting from acc = +0  (or from the stated first term);
//   * everything else is one IEEE fp32 operation per written operator;
//   * this file is compiled with -ffp-contract=off, so no other fusion occurs.
// Discretisation: dt*[A B;0 0] is nilpotent of index 3, so expm = I + M + M^2/2
// exactly and Adt^k, Adt^a*Bdt have closed forms.  The reference
// evaluates them with Eigen's fp32 Pade expm and repeated products
// (qr_mpc_interface.cpp:257-293), whose rounding cannot be reproduced without
// Eigen; mpc_assemble_literal() below restates that route to bound the gap.
#include <cstdlib>
#include "qr_oracle.h"
#include <algorithm>

namespace qro {

// Stopping threshold on the constraint slack of the MPC QP (0 = QuadProg++'s psi rule); QRO_MPC_ABS_TOL overrides for experiments.
static double mpc_abs_tol()
{
    static const double v = [] { const char *e = std::getenv("QRO_MPC_ABS_TOL"); return e ? std::atof(e) : 1e-9; }();
    return v;
}

namespace {

inline float dot3(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}
inline float det2(float a, float b, float c, float d) { return fmaf(a, b, -(c * d)); }   // a*b - c*d

struct Srbd {               // everything K1/K2 produce, in closed form
    float R[3][3];          // quat.toRotationMatrix(), body->world  (:350, "yawRotMat = rotMat" quirk 1)
    float U[4][3][3];       // I_world^-1 * [r_p]x          (B_c rows 6-8, :328)
    float T[4][3][3];       // R^T * U_p                    (A_c[0:3,6:9] * B_c rows 6-8)
    float dt, dt2, minv;
};

void build_srbd(const MpcConfig &cfg, const MpcInput &in, Srbd &s)
{
    // Eigen::Quaternion::toRotationMatrix (published algorithm), (w,x,y,z) = quat[0..3] (:344-347)
    const float w = in.quat[0], x = in.quat[1], y = in.quat[2], z = in.quat[3];
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    s.R[0][0] = 1.f - (tyy + tzz); s.R[0][1] = txy - twz;         s.R[0][2] = txz + twy;
    s.R[1][0] = txy + twz;         s.R[1][1] = 1.f - (txx + tzz); s.R[1][2] = tyz - twx;
    s.R[2][0] = txz - twy;         s.R[2][1] = tyz + twx;         s.R[2][2] = 1.f - (txx + tyy);

    // I_world = R * I_b * R^T  (:365), I_b diagonal (:172)
    float RI[3][3], Iw[3][3];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) RI[i][k] = s.R[i][k] * cfg.inertia[k];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Iw[i][j] = dot3(RI[i][0], s.R[j][0], RI[i][1], s.R[j][1], RI[i][2], s.R[j][2]);

    // I_world.inverse() (:324): 3x3 cofactor inverse.
    float cof[3][3];
    cof[0][0] = det2(Iw[1][1], Iw[2][2], Iw[1][2], Iw[2][1]);
    cof[0][1] = det2(Iw[1][2], Iw[2][0], Iw[1][0], Iw[2][2]);
    cof[0][2] = det2(Iw[1][0], Iw[2][1], Iw[1][1], Iw[2][0]);
    cof[1][0] = det2(Iw[0][2], Iw[2][1], Iw[0][1], Iw[2][2]);
    cof[1][1] = det2(Iw[0][0], Iw[2][2], Iw[0][2], Iw[2][0]);
    cof[1][2] = det2(Iw[0][1], Iw[2][0], Iw[0][0], Iw[2][1]);
    cof[2][0] = det2(Iw[0][1], Iw[1][2], Iw[0][2], Iw[1][1]);
    cof[2][1] = det2(Iw[0][2], Iw[1][0], Iw[0][0], Iw[1][2]);
    cof[2][2] = det2(Iw[0][0], Iw[1][1], Iw[0][1], Iw[1][0]);
    const float det = dot3(Iw[0][2], cof[0][2], Iw[0][1], cof[0][1], Iw[0][0], cof[0][0]);
    const float invdet = 1.0f / det;
    float Iinv[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Iinv[i][j] = cof[j][i] * invdet;

    // B_c rows 6-8: I_world_inv * crossMatrix(r_p)   (:328; crossMatrix QI/utils/qr_se3.h:92-101)
    for (int p = 0; p < 4; ++p) {
        const float rx = in.r[3 * p + 0], ry = in.r[3 * p + 1], rz = in.r[3 * p + 2];
        for (int i = 0; i < 3; ++i) {
            s.U[p][i][0] = det2(Iinv[i][1], rz, Iinv[i][2], ry);
            s.U[p][i][1] = det2(Iinv[i][2], rx, Iinv[i][0], rz);
            s.U[p][i][2] = det2(Iinv[i][0], ry, Iinv[i][1], rx);
        }
        // A_c[0:3,6:9] = R^T (:313)  =>  T_p = R^T U_p
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                s.T[p][i][j] = dot3(s.R[0][i], s.U[p][0][j], s.R[1][i], s.U[p][1][j], s.R[2][i], s.U[p][2][j]);
    }
    s.dt = cfg.dt;
    s.dt2 = cfg.dt * cfg.dt;
    s.minv = 1.0f / cfg.mass;     // m_inv = Identity / mass (:325)
}

// G[a] = Adt^a * Bdt, 13 x 12  (Bqp block (r,c) = G[r-c] for r >= c, :289)
void build_G(const Srbd &s, int h, std::vector<float> &G)
{
    G.assign((size_t)h * 13 * 12, 0.f);
    const float dtm = s.dt * s.minv;
    for (int a = 0; a < h; ++a) {
        const float ca = ((float)a + 0.5f) * s.dt2;
        const float cam = ca * s.minv;
        float *Ga = &G[(size_t)a * 156];
        for (int p = 0; p < 4; ++p)
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) {
                    Ga[(0 + i) * 12 + 3 * p + j] = ca * s.T[p][i][j];
                    Ga[(6 + i) * 12 + 3 * p + j] = s.dt * s.U[p][i][j];
                }
                Ga[(3 + i) * 12 + 3 * p + i] = cam;
                Ga[(9 + i) * 12 + 3 * p + i] = dtm;
            }
    }
}

// v = Aqp*x0 - X_d   (:362, :376-390, :412), Aqp block r = Adt^(r+1)
void build_v(const Srbd &s, const MpcInput &in, int h, std::vector<float> &v)
{
    v.assign((size_t)13 * h, 0.f);
    const float grav = -9.8f;        // x0(12), quirk 2
    for (int r = 0; r < h; ++r) {
        const float kd = (float)(r + 1) * s.dt;
        const float hk2 = (kd * kd) * 0.5f;
        float ax[13];
        for (int i = 0; i < 3; ++i) {
            float acc = in.rpy[i];
            for (int j = 0; j < 3; ++j) acc = fmaf(kd * s.R[j][i], in.w[j], acc);
            ax[i] = acc;
            ax[3 + i] = fmaf(kd, in.v[i], in.p[i]);
            ax[6 + i] = in.w[i];
            ax[9 + i] = in.v[i];
        }
        ax[5] = fmaf(hk2, grav, ax[5]);
        ax[11] = fmaf(kd, grav, ax[11]);
        ax[12] = grav;
        for (int j = 0; j < 12; ++j) v[13 * r + j] = ax[j] - in.traj[12 * r + j];
        v[13 * r + 12] = ax[12] - 0.f;
    }
}

void finish_assembly(const MpcConfig &cfg, const MpcInput &in, const std::vector<float> &Bqp,
                     const std::vector<float> &v, MpcAssembly &out)
{
    const int h = cfg.horizon, n = 12 * h, K = 13 * h, m = 20 * h;
    out.n = n; out.m = m;
    out.H.assign((size_t)n * n, 0.f);
    out.g.assign(n, 0.f);
    out.ub.assign(m, 0.f);
    // temp = 2 * Bqp^T * diag(full_weight) on the upper block triangle (:396-406)
    std::vector<float> temp((size_t)n * K, 0.f);
    float w2[13];
    for (int s = 0; s < 12; ++s) w2[s] = 2.f * cfg.weights[s];
    w2[12] = 2.f * 0.f;
    for (int i = 0; i < h; ++i)
        for (int j = i; j < h; ++j)
            for (int t = 0; t < 12; ++t)
                for (int s = 0; s < 13; ++s)
                    temp[(size_t)(12 * i + t) * K + 13 * j + s] = Bqp[(size_t)(13 * j + s) * n + 12 * i + t] * w2[s];
    // qH = temp*Bqp + 2*alpha*I ; qg = temp*(Aqp*x0 - X_d)   (:411-412)
    // Row a of temp is zero left of k = 13*(a/12) and column b of Bqp is zero above
    // k = 13*(b/12); starting the chain at k0 = 13*max(a/12, b/12) skips only
    // fmaf(0, x, acc) / fmaf(x, 0, acc) terms, which leave acc unchanged, so the
    // result is bit-identical to the dense k = 0..13h-1 chain.
    const float two_alpha = 2.f * cfg.alpha;
    for (int a = 0; a < n; ++a) {
        for (int b = 0; b < n; ++b) {
            float acc = 0.f;
            for (int k = 13 * std::max(a / 12, b / 12); k < K; ++k) acc = fmaf(temp[(size_t)a * K + k], Bqp[(size_t)k * n + b], acc);
            out.H[(size_t)a * n + b] = (a == b) ? acc + two_alpha : acc;
        }
        float acc = 0.f;
        for (int k = 13 * (a / 12); k < K; ++k) acc = fmaf(temp[(size_t)a * K + k], v[k], acc);
        out.g[a] = acc;
    }
    // U_b (:222-226, :386-389); lb = 0 (:423-425)
    for (int k = 0; k < 4 * h; ++k) {
        for (int c = 0; c < 4; ++c) out.ub[5 * k + c] = 5e10f;
        out.ub[5 * k + 4] = in.gait[k] * cfg.fmax;
    }
    out.invmu = 1.f / cfg.mu;      // mu_ (:230)
}

}  // namespace

void mpc_assemble(const MpcConfig &cfg, const MpcInput &in, MpcAssembly &out)
{
    const int h = cfg.horizon, n = 12 * h, K = 13 * h;
    Srbd s;
    build_srbd(cfg, in, s);
    std::vector<float> G, v;
    build_G(s, h, G);
    build_v(s, in, h, v);
    std::vector<float> Bqp((size_t)K * n, 0.f);
    for (int r = 0; r < h; ++r)
        for (int c = 0; c <= r; ++c)
            for (int si = 0; si < 13; ++si)
                for (int t = 0; t < 12; ++t)
                    Bqp[(size_t)(13 * r + si) * n + 12 * c + t] = G[(size_t)(r - c) * 156 + si * 12 + t];
    finish_assembly(cfg, in, Bqp, v, out);
}

// ---------------------------------------------------------------------------
// Literal route: fp32 Pade-approximant expm with scaling & squaring (the
// algorithm of Eigen's unsupported MatrixFunctions for float: Higham 2005,
// degree 3/5/7 chosen by the 1-norm), powerMats by repeated products, dense
// Aqp*x0.  Only plain (unfused) fp32 arithmetic.
// ---------------------------------------------------------------------------
namespace {
typedef std::vector<float> VF;
VF mm(const VF &a, const VF &b, int n)
{
    VF c((size_t)n * n, 0.f);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            float s = 0.f;
            for (int k = 0; k < n; ++k) s += a[(size_t)i * n + k] * b[(size_t)k * n + j];
            c[(size_t)i * n + j] = s;
        }
    return c;
}
VF solve_lu(VF A, VF B, int n)     // A X = B, partial pivoting
{
    for (int k = 0; k < n; ++k) {
        int piv = k; float best = std::fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(A[(size_t)i * n + k]) > best) { best = std::fabs(A[(size_t)i * n + k]); piv = i; }
        if (piv != k) for (int j = 0; j < n; ++j) { std::swap(A[(size_t)k * n + j], A[(size_t)piv * n + j]); std::swap(B[(size_t)k * n + j], B[(size_t)piv * n + j]); }
        for (int i = k + 1; i < n; ++i) {
            float l = A[(size_t)i * n + k] / A[(size_t)k * n + k];
            for (int j = k; j < n; ++j) A[(size_t)i * n + j] -= l * A[(size_t)k * n + j];
            for (int j = 0; j < n; ++j) B[(size_t)i * n + j] -= l * B[(size_t)k * n + j];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = n - 1; i >= 0; --i) {
            float s = B[(size_t)i * n + j];
            for (int k = i + 1; k < n; ++k) s -= A[(size_t)i * n + k] * B[(size_t)k * n + j];
            B[(size_t)i * n + j] = s / A[(size_t)i * n + i];
        }
    return B;
}
VF expm_pade_f32(const VF &M, int n)
{
    float l1 = 0.f;
    for (int j = 0; j < n; ++j) { float s = 0.f; for (int i = 0; i < n; ++i) s += std::fabs(M[(size_t)i * n + j]); l1 = std::max(l1, s); }
    VF I((size_t)n * n, 0.f);
    for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.f;
    VF A = M; int squarings = 0;
    VF U, V;
    auto lin = [&](std::initializer_list<std::pair<float, const VF *>> terms) {
        VF r((size_t)n * n, 0.f);
        for (auto &t : terms) for (size_t i = 0; i < r.size(); ++i) r[i] += t.first * (*t.second)[i];
        return r;
    };
    if (l1 < 4.258730016922831e-001f) {
        VF A2 = mm(A, A, n);
        VF tmp = lin({{1.f, &A2}, {60.f, &I}});
        U = mm(A, tmp, n);
        V = lin({{12.f, &A2}, {120.f, &I}});
    } else if (l1 < 1.880152677804762e+000f) {
        VF A2 = mm(A, A, n), A4 = mm(A2, A2, n);
        VF tmp = lin({{1.f, &A4}, {420.f, &A2}, {15120.f, &I}});
        U = mm(A, tmp, n);
        V = lin({{30.f, &A4}, {3360.f, &A2}, {30240.f, &I}});
    } else {
        const float maxnorm = 3.925724783138660f;
        int e; frexpf(l1 / maxnorm, &e);
        squarings = std::max(0, e);
        float sc = ldexpf(1.f, -squarings);
        for (auto &a : A) a *= sc;
        VF A2 = mm(A, A, n), A4 = mm(A2, A2, n), A6 = mm(A4, A2, n);
        VF tmp = lin({{1.f, &A6}, {1512.f, &A4}, {277200.f, &A2}, {8648640.f, &I}});
        U = mm(A, tmp, n);
        V = lin({{56.f, &A6}, {25200.f, &A4}, {1995840.f, &A2}, {17297280.f, &I}});
    }
    VF num((size_t)n * n), den((size_t)n * n);
    for (size_t i = 0; i < num.size(); ++i) { num[i] = U[i] + V[i]; den[i] = -U[i] + V[i]; }
    VF R = solve_lu(den, num, n);
    for (int i = 0; i < squarings; ++i) R = mm(R, R, n);
    return R;
}
}  // namespace

void mpc_assemble_literal(const MpcConfig &cfg, const MpcInput &in, MpcAssembly &out)
{
    const int h = cfg.horizon, n = 12 * h, K = 13 * h;
    Srbd s;
    build_srbd(cfg, in, s);
    // A_ct, B_ct (:296-331) stacked into ABc (:264-267)
    VF ABc(25 * 25, 0.f);
    auto at = [&](int i, int j) -> float & { return ABc[(size_t)i * 25 + j]; };
    at(3, 9) = 1.f; at(4, 10) = 1.f; at(5, 11) = 1.f; at(11, 12) = 1.f;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) at(i, 6 + j) = s.R[j][i];
    for (int p = 0; p < 4; ++p)
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) at(6 + i, 13 + 3 * p + j) = s.U[p][i][j];
            at(9 + i, 13 + 3 * p + i) = s.minv;
        }
    for (auto &a : ABc) a *= cfg.dt;
    VF E = expm_pade_f32(ABc, 25);
    VF Adt(169), Bdt(156);
    for (int i = 0; i < 13; ++i) {
        for (int j = 0; j < 13; ++j) Adt[i * 13 + j] = E[(size_t)i * 25 + j];
        for (int j = 0; j < 12; ++j) Bdt[i * 12 + j] = E[(size_t)i * 25 + 13 + j];
    }
    std::vector<VF> pw(h + 1);
    pw[0].assign(169, 0.f);
    for (int i = 0; i < 13; ++i) pw[0][i * 13 + i] = 1.f;
    for (int i = 1; i <= h; ++i) pw[i] = mm(Adt, pw[i - 1], 13);
    std::vector<float> Bqp((size_t)K * n, 0.f), v((size_t)K, 0.f);
    float x0[13] = {in.rpy[0], in.rpy[1], in.rpy[2], in.p[0], in.p[1], in.p[2], in.w[0], in.w[1], in.w[2], in.v[0], in.v[1], in.v[2], -9.8f};
    for (int r = 0; r < h; ++r) {
        for (int i = 0; i < 13; ++i) {
            float sacc = 0.f;
            for (int j = 0; j < 13; ++j) sacc += pw[r + 1][i * 13 + j] * x0[j];
            v[13 * r + i] = sacc - (i < 12 ? in.traj[12 * r + i] : 0.f);
        }
        for (int c = 0; c <= r; ++c)
            for (int i = 0; i < 13; ++i)
                for (int t = 0; t < 12; ++t) {
                    float sacc = 0.f;
                    for (int j = 0; j < 13; ++j) sacc += pw[r - c][i * 13 + j] * Bdt[j * 12 + t];
                    Bqp[(size_t)(13 * r + i) * n + 12 * c + t] = sacc;
                }
    }
    finish_assembly(cfg, in, Bqp, v, out);
}

// ---------------------------------------------------------------------------
// K6.  The reference hands (H,g,fmat,0,U_b) to qpOASES in double (:418-438).
// Swing leg-steps have U_b = 0 on the f_z row, which with the four pyramid rows
// pins that force to zero; those variables are eliminated exactly and the
// remaining strictly convex QP is solved by qp_solve_gi.  The fp32 product qH is
// symmetric only to rounding; 1/2 u'Hu depends on H through (H+H')/2 alone, so
// the stated QP's unique optimum is that of the (exactly, in double) averaged
// matrix.  qpOASES assumes symmetry and lands ~2e-4 (relative) away from it --
// by as much as it moves when handed H' instead of H (tests/test_oracle_mpc.py).
// ---------------------------------------------------------------------------
int mpc_solve_qp(const MpcAssembly &a, const float *gait, int horizon, double *u_out, QpStats *st)
{
    const int n = a.n;
    std::vector<int> idx;
    for (int k = 0; k < 4 * horizon; ++k)
        if (a.ub[5 * k + 4] > 0.f) for (int c = 0; c < 3; ++c) idx.push_back(3 * k + c);
    (void)gait;
    const int ns = (int)idx.size(), nls = ns / 3, m = 6 * nls;
    std::fill(u_out, u_out + n, 0.0);
    if (ns == 0) { if (st) *st = QpStats(); return 0; }
    std::vector<double> G((size_t)ns * ns), g0(ns), CI((size_t)ns * m, 0.0), ci0(m, 0.0), x(ns);
    for (int i = 0; i < ns; ++i) {
        for (int j = 0; j < ns; ++j) {
            int ai = idx[i], bj = idx[j];
            G[(size_t)i * ns + j] = 0.5 * ((double)a.H[(size_t)ai * n + bj] + (double)a.H[(size_t)bj * n + ai]);
        }
        g0[i] = (double)a.g[idx[i]];
    }
    const double im = (double)a.invmu;
    for (int k = 0; k < nls; ++k) {
        const int legstep = idx[3 * k] / 3;
        const int c0 = 6 * k;
        auto set = [&](int row, int col, double val) { CI[(size_t)(3 * k + row) * m + c0 + col] = val; };
        set(0, 0, im);  set(2, 0, 1.0);      //  fx/mu + fz >= 0      (f_block row 0, :232)
        set(0, 1, -im); set(2, 1, 1.0);      // -fx/mu + fz >= 0
        set(1, 2, im);  set(2, 2, 1.0);      //  fy/mu + fz >= 0
        set(1, 3, -im); set(2, 3, 1.0);      // -fy/mu + fz >= 0
        set(2, 4, 1.0);                      //  fz >= 0
        set(2, 5, -1.0); ci0[c0 + 5] = (double)a.ub[5 * legstep + 4];   // fz <= gait*fMax
    }
    int rc = qp_solve_gi(ns, G.data(), g0.data(), 0, nullptr, nullptr, m, CI.data(), ci0.data(), x.data(), nullptr, st, 0, mpc_abs_tol());
    for (int i = 0; i < ns; ++i) u_out[idx[i]] = x[i];
    return rc;
}

// ---------------------------------------------------------------------------
// K7.  QS/robots/qr_robot.cpp:148-172 (AnalyticalLegJacobian), :241-251
// (MapContactForceToJointTorques), qr_mpc_stance_leg_controller.cpp:402-409.
// ---------------------------------------------------------------------------
void analytical_leg_jacobian(const LegGeom &geo, const float t[3], int leg, float J[9])
{
    const float l_up = geo.upper_l, l_low = geo.lower_l;
    const float signedHip = geo.hip_l * ((leg + 1) % 2 == 0 ? 1.f : -1.f);      // hipLength * pow(-1, leg+1)
    const float lEff = std::sqrt(l_up * l_up + l_low * l_low + 2 * l_up * l_low * std::cos(t[2]));
    const float tEff = t[1] + t[2] / 2;
    J[0] = 0;
    J[1] = -lEff * std::cos(tEff);
    J[2] = l_low * l_up * std::sin(t[2]) * std::sin(tEff) / lEff - lEff * std::cos(tEff) / 2;
    J[3] = -signedHip * std::sin(t[0]) + lEff * std::cos(t[0]) * std::cos(tEff);
    J[4] = -lEff * std::sin(t[0]) * std::sin(tEff);
    J[5] = -l_low * l_up * std::sin(t[0]) * std::sin(t[2]) * std::cos(tEff) / lEff - lEff * std::sin(t[0]) * std::sin(tEff) / 2;
    J[6] = signedHip * std::cos(t[0]) + lEff * std::sin(t[0]) * std::cos(tEff);
    J[7] = lEff * std::sin(tEff) * std::cos(t[0]);
    J[8] = l_low * l_up * std::sin(t[2]) * std::cos(t[0]) * std::cos(tEff) / lEff + lEff * std::sin(tEff) * std::cos(t[0]) / 2;
}

void foot_positions_in_base_frame(const LegGeom &geo, const float hipOffset[12], const float q[12], float out[12])
{
    for (int leg = 0; leg < 4; ++leg) {     // FootPositionInHipFrame, qr_robot.cpp:127-146
        const float tab = q[3 * leg], thip = q[3 * leg + 1], tknee = q[3 * leg + 2];
        const float signedHip = geo.hip_l * ((leg + 1) % 2 == 0 ? 1.f : -1.f);
        const float legDist = std::sqrt(geo.upper_l * geo.upper_l + geo.lower_l * geo.lower_l + 2 * geo.upper_l * geo.lower_l * std::cos(tknee));
        const float eff = thip + tknee / 2;
        const float offXHip = -legDist * std::sin(eff), offZHip = -legDist * std::cos(eff), offYHip = signedHip;
        out[3 * leg + 0] = offXHip + hipOffset[3 * leg + 0];
        out[3 * leg + 1] = std::cos(tab) * offYHip - std::sin(tab) * offZHip + hipOffset[3 * leg + 1];
        out[3 * leg + 2] = std::sin(tab) * offYHip + std::cos(tab) * offZHip + hipOffset[3 * leg + 2];
    }
}

void mpc_force_to_torque(const LegGeom &geo, const float quat[4], const float q[12], const double f_world[12], float tau[12])
{
    // baseRMat^T = quaternionToRotationMatrix(quat)  (QS/robots/qr_robot.cpp:70)
    Q4<float> qq = {{quat[0], quat[1], quat[2], quat[3]}};
    M3<float> Rt = quaternionToRotationMatrix(qq);
    for (int leg = 0; leg < 4; ++leg) {
        float f[3] = {(float)f_world[3 * leg], (float)f_world[3 * leg + 1], (float)f_world[3 * leg + 2]};   // f(axis,leg) = GetMPCSolution (:404)
        float fff[3];
        for (int i = 0; i < 3; ++i) {       // f_ff = -R^T f (:406)
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += (-Rt[i][k]) * f[k];
            fff[i] = s;
        }
        float J[9];
        analytical_leg_jacobian(geo, &q[3 * leg], leg, J);
        for (int j = 0; j < 3; ++j) {       // tau = J^T f_ff (qr_robot.cpp:244)
            float s = 0.f;
            for (int i = 0; i < 3; ++i) s += J[3 * i + j] * fff[i];
            tau[3 * leg + j] = s;
        }
    }
}

}  // namespace qro
