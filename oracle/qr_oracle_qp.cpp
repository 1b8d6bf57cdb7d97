// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Dense strictly-convex QP by the Goldfarb-Idnani dual active-set method
// (D. Goldfarb, A. Idnani, Math. Programming 27 (1983) 1-33), written from the
// paper.  It plays the role of the two vendored solvers of the reference:
//   * qpOASES 3.2.0  for the MPC QP  (QS/controllers/mpc/qr_mpc_interface.cpp:428-438)
//   * QuadProg++     for the WBC QP  (QS/controllers/wbc/qr_wholebody_impulse_ctrl.cpp:113)
// Both QPs are strictly convex, so the optimum is unique and independent of the
// active-set strategy; tests/test_oracle_qp.py pins this solver against both
// vendored solvers (oracle/_ref) and against the known-answer problem of
// QX/QuadProgpp/src/main.cc:8-20.
//
// Convention (QuadProg++.hh:8-23):  min 1/2 x'Gx + g0'x,  CE'x + ce0 = 0,  CI'x + ci0 >= 0.
#include "qr_oracle.h"
#include <limits>
#include <algorithm>

namespace qro {

namespace {

struct GI {
    int n;
    std::vector<double> J;     // n x n, J = L^-T Q  (columns: first q span the active normals)
    std::vector<double> R;     // n x n upper-triangular (q x q used)
    std::vector<double> d, z, r, u, np;
    std::vector<int> A;        // active constraint ids (equalities: -(i+1); inequalities: i)
    int q = 0;
    double Rnorm = 1.0;

    explicit GI(int n_) : n(n_), J((size_t)n_ * n_), R((size_t)n_ * n_, 0.0), d(n_), z(n_), r(n_), u(n_ + 1), np(n_), A(n_ + 1) {}
    double &Jm(int i, int j) { return J[(size_t)i * n + j]; }
    double &Rm(int i, int j) { return R[(size_t)i * n + j]; }

    void compute_d() {   // d = J^T np
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += Jm(k, i) * np[k];
            d[i] = s;
        }
    }
    void update_z() {    // z = J2 d2
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int k = q; k < n; ++k) s += Jm(i, k) * d[k];
            z[i] = s;
        }
    }
    void update_r() {    // r = R^-1 d1
        for (int i = q - 1; i >= 0; --i) {
            double s = 0;
            for (int j = i + 1; j < q; ++j) s += Rm(i, j) * r[j];
            r[i] = (d[i] - s) / Rm(i, i);
        }
    }
    static double hyp(double a, double b) { return std::hypot(a, b); }

    bool add_constraint() {
        // Givens rotations zeroing d[n-1..q+1]; same rotations applied to J's columns.
        for (int j = n - 1; j >= q + 1; --j) {
            double cc = d[j - 1], ss = d[j];
            double h = hyp(cc, ss);
            if (h == 0.0) continue;
            d[j] = 0.0;
            ss /= h; cc /= h;
            if (cc < 0.0) { cc = -cc; ss = -ss; d[j - 1] = -h; } else d[j - 1] = h;
            double xny = ss / (1.0 + cc);
            for (int k = 0; k < n; ++k) {
                double t1 = Jm(k, j - 1), t2 = Jm(k, j);
                Jm(k, j - 1) = t1 * cc + t2 * ss;
                Jm(k, j) = xny * (t1 + Jm(k, j - 1)) - t2;
            }
        }
        if (std::fabs(d[q]) <= std::numeric_limits<double>::epsilon() * Rnorm) return false;   // linearly dependent
        for (int i = 0; i <= q; ++i) Rm(i, q) = d[i];
        Rnorm = std::max(Rnorm, std::fabs(d[q]));
        ++q;
        return true;
    }
    void delete_constraint(int l) {   // l = position in A
        for (int i = l; i < q - 1; ++i) {
            A[i] = A[i + 1];
            u[i] = u[i + 1];
            for (int j = 0; j < n; ++j) Rm(j, i) = Rm(j, i + 1);
        }
        A[q - 1] = A[q];
        u[q - 1] = u[q];
        A[q] = 0; u[q] = 0.0;
        for (int j = 0; j < q; ++j) Rm(j, q - 1) = 0.0;
        --q;
        if (q == 0) return;
        for (int j = l; j < q; ++j) {
            double cc = Rm(j, j), ss = Rm(j + 1, j);
            double h = hyp(cc, ss);
            if (h == 0.0) continue;
            cc /= h; ss /= h;
            Rm(j + 1, j) = 0.0;
            if (cc < 0.0) { Rm(j, j) = -h; cc = -cc; ss = -ss; } else Rm(j, j) = h;
            double xny = ss / (1.0 + cc);
            for (int k = j + 1; k < q; ++k) {
                double t1 = Rm(j, k), t2 = Rm(j + 1, k);
                Rm(j, k) = t1 * cc + t2 * ss;
                Rm(j + 1, k) = xny * (t1 + Rm(j, k)) - t2;
            }
            for (int k = 0; k < n; ++k) {
                double t1 = Jm(k, j), t2 = Jm(k, j + 1);
                Jm(k, j) = t1 * cc + t2 * ss;
                Jm(k, j + 1) = xny * (Jm(k, j) + t1) - t2;
            }
        }
    }
};

}  // namespace

int qp_solve_gi(int n, const double *G, const double *g0, int p, const double *CE, const double *ce0,
                int m, const double *CI, const double *ci0, double *x, double *lambda_ineq, QpStats *st, int max_iter, double abs_tol)
{
    const double inf = std::numeric_limits<double>::infinity();
    const double eps = std::numeric_limits<double>::epsilon();
    if (max_iter <= 0) max_iter = 50 * (n + m + p) + 100;
    GI s(n);
    QpStats stats;

    // Cholesky G = L L^T (lower).
    std::vector<double> L((size_t)n * n, 0.0);
    double c1 = 0.0;
    for (int i = 0; i < n; ++i) c1 += G[(size_t)i * n + i];
    for (int j = 0; j < n; ++j) {
        double sum = G[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) sum -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (sum <= 0.0) return 3;   // not positive definite
        double ljj = std::sqrt(sum);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double v = G[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = v / ljj;
        }
    }
    // J = L^-T : column j of J solves L^T J(:,j) = e_j.
    double c2 = 0.0;
    for (int j = 0; j < n; ++j) {
        for (int i = n - 1; i >= 0; --i) {
            double v = (i == j) ? 1.0 : 0.0;
            for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * s.Jm(k, j);
            s.Jm(i, j) = v / L[(size_t)i * n + i];
        }
        c2 += s.Jm(j, j);
    }
    // x = -G^-1 g0  via  L y = -g0, L^T x = y.
    {
        std::vector<double> y(n);
        for (int i = 0; i < n; ++i) {
            double v = -g0[i];
            for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * y[k];
            y[i] = v / L[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double v = y[i];
            for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * x[k];
            x[i] = v / L[(size_t)i * n + i];
        }
    }

    // Equality constraints go into the working set first.
    for (int i = 0; i < p; ++i) {
        for (int k = 0; k < n; ++k) s.np[k] = CE[(size_t)k * p + i];
        s.compute_d();
        s.update_z();
        s.update_r();
        double znp = 0, res = ce0[i], zz = 0;
        for (int k = 0; k < n; ++k) { znp += s.z[k] * s.np[k]; res += s.np[k] * x[k]; zz += s.z[k] * s.z[k]; }
        double t2 = 0.0;
        if (std::fabs(zz) > eps) t2 = -res / znp;
        for (int k = 0; k < n; ++k) x[k] += t2 * s.z[k];
        s.u[s.q] = t2;
        for (int k = 0; k < s.q; ++k) s.u[k] -= t2 * s.r[k];
        s.A[s.q] = -(i + 1);
        if (!s.add_constraint()) return 1;   // dependent equalities
    }

    std::vector<double> sv(m);
    std::vector<char> active(m, 0), excluded(m, 0);
    int status = 0;
    for (;;) {
        if (++stats.iters > max_iter) { status = 2; break; }
        // Step 1: most violated inactive inequality.
        double psi = 0.0, smin = 0.0; int ip = -1;
        for (int i = 0; i < m; ++i) {
            double v = ci0[i];
            for (int k = 0; k < n; ++k) v += CI[(size_t)k * m + i] * x[k];
            sv[i] = v;
            psi += std::min(0.0, v);
            if (!active[i] && !excluded[i] && v < smin) { smin = v; ip = i; }
        }
        // QuadProg++'s stopping rule (sum of violations against an estimate of cond(G)); abs_tol > 0 replaces it by "no row
        // violated by more than abs_tol" (the MPC QP, whose 1/(2 alpha) = 1.25e5 turns 1e-7 of slack into 1e-2 N of force)
        if (abs_tol > 0.0) { if (ip < 0 || smin >= -abs_tol) break; }
        else if (ip < 0 || std::fabs(psi) <= m * eps * c1 * c2 * 100.0) break;

        // Step 2: add ip, possibly dropping blocking constraints on the way.
        for (int k = 0; k < n; ++k) s.np[k] = CI[(size_t)k * m + ip];
        s.u[s.q] = 0.0;
        s.A[s.q] = ip;
        bool next_outer = false;
        while (!next_outer) {
            if (++stats.iters > max_iter) { status = 2; break; }
            s.compute_d();
            s.update_z();
            s.update_r();
            double t1 = inf; int l = -1;
            for (int k = p; k < s.q; ++k)
                if (s.r[k] > 0.0 && s.u[k] / s.r[k] < t1) { t1 = s.u[k] / s.r[k]; l = k; }
            double zz = 0, znp = 0;
            for (int k = 0; k < n; ++k) { zz += s.z[k] * s.z[k]; znp += s.z[k] * s.np[k]; }
            double t2 = (std::fabs(zz) > eps) ? -sv[ip] / znp : inf;
            double t = std::min(t1, t2);
            if (t >= inf) { status = 1; break; }
            if (t2 >= inf) {
                // dual step only
                for (int k = 0; k < s.q; ++k) s.u[k] -= t * s.r[k];
                s.u[s.q] += t;
                active[s.A[l]] = 0;
                s.delete_constraint(l);
                ++stats.drops;
                continue;
            }
            for (int k = 0; k < n; ++k) x[k] += t * s.z[k];
            for (int k = 0; k < s.q; ++k) s.u[k] -= t * s.r[k];
            s.u[s.q] += t;
            if (t == t2) {
                // full step: ip becomes active
                if (!s.add_constraint()) {
                    // numerically dependent on the working set: leave it out (it is satisfied to rounding)
                    excluded[ip] = 1;
                } else {
                    active[ip] = 1;
                    ++stats.adds;
                    std::fill(excluded.begin(), excluded.end(), 0);
                }
                next_outer = true;
            } else {
                active[s.A[l]] = 0;
                s.delete_constraint(l);
                ++stats.drops;
                double v = ci0[ip];
                for (int k = 0; k < n; ++k) v += CI[(size_t)k * m + ip] * x[k];
                sv[ip] = v;
            }
        }
        if (status) break;
    }

    if (lambda_ineq) {
        std::fill(lambda_ineq, lambda_ineq + m, 0.0);
        for (int k = p; k < s.q; ++k) lambda_ineq[s.A[k]] = s.u[k];
    }
    double obj = 0;
    for (int i = 0; i < n; ++i) {
        double gi = g0[i];
        for (int j = 0; j < n; ++j) gi += 0.5 * G[(size_t)i * n + j] * x[j];
        obj += gi * x[i];
    }
    stats.obj = obj;
    stats.n_active = s.q - p;
    if (st) *st = stats;
    return status;
}

}  // namespace qro
