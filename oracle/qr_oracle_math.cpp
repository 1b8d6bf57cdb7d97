// TEST INFRASTRUCTURE (see qr_oracle.h).  Math helpers: restatements of
// QI/utils/qr_se3.h and QI/utils/qr_algebra.h plus the two Eigen decompositions
// the path relies on (JacobiSVD, partial-pivot LU inverse).
#include "qr_oracle.h"
#include <algorithm>

namespace qro {

// QI/utils/qr_se3.h:72-89.  NOTE: this is the coordinate-transform (transposed)
// rotation: X -> [[1,0,0],[0,c,s],[0,-s,c]].
template <typename T> M3<T> coordinateRotation(int axis, T theta)
{
    T s = std::sin(theta), c = std::cos(theta);
    M3<T> R;
    if (axis == 0) {
        R = {{{1, 0, 0}, {0, c, s}, {0, -s, c}}};
    } else if (axis == 1) {
        R = {{{c, 0, -s}, {0, 1, 0}, {s, 0, c}}};
    } else {
        R = {{{c, s, 0}, {-s, c, 0}, {0, 0, 1}}};
    }
    return R;
}

template <typename T> M3<T> mul(const M3<T> &a, const M3<T> &b)
{
    M3<T> m;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T s = 0;
            for (int k = 0; k < 3; ++k) s += a[i][k] * b[k][j];
            m[i][j] = s;
        }
    return m;
}
template <typename T> V3<T> mul(const M3<T> &a, const V3<T> &b)
{
    V3<T> m;
    for (int i = 0; i < 3; ++i) {
        T s = 0;
        for (int k = 0; k < 3; ++k) s += a[i][k] * b[k];
        m[i] = s;
    }
    return m;
}
template <typename T> M3<T> transpose(const M3<T> &a)
{
    M3<T> m;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = a[j][i];
    return m;
}

// :109-116   Rx(r) * Ry(p) * Rz(y)  of the transposed elementary rotations.
template <typename T> M3<T> rpyToRotMat(const V3<T> &v)
{
    return mul(mul(coordinateRotation<T>(0, v[0]), coordinateRotation<T>(1, v[1])), coordinateRotation<T>(2, v[2]));
}

// :139-172   (input is transposed first)
template <typename T> Q4<T> rotationMatrixToQuaternion(const M3<T> &r1)
{
    Q4<T> q;
    M3<T> r = transpose(r1);
    T tr = r[0][0] + r[1][1] + r[2][2];
    if (tr > T(0.0)) {
        T S = std::sqrt(tr + T(1.0)) * T(2.0);
        q[0] = T(0.25) * S;
        q[1] = (r[2][1] - r[1][2]) / S;
        q[2] = (r[0][2] - r[2][0]) / S;
        q[3] = (r[1][0] - r[0][1]) / S;
    } else if ((r[0][0] > r[1][1]) && (r[0][0] > r[2][2])) {
        T S = std::sqrt(T(1.0) + r[0][0] - r[1][1] - r[2][2]) * T(2.0);
        q[0] = (r[2][1] - r[1][2]) / S;
        q[1] = T(0.25) * S;
        q[2] = (r[0][1] + r[1][0]) / S;
        q[3] = (r[0][2] + r[2][0]) / S;
    } else if (r[1][1] > r[2][2]) {
        T S = std::sqrt(T(1.0) + r[1][1] - r[0][0] - r[2][2]) * T(2.0);
        q[0] = (r[0][2] - r[2][0]) / S;
        q[1] = (r[0][1] + r[1][0]) / S;
        q[2] = T(0.25) * S;
        q[3] = (r[1][2] + r[2][1]) / S;
    } else {
        T S = std::sqrt(T(1.0) + r[2][2] - r[0][0] - r[1][1]) * T(2.0);
        q[0] = (r[1][0] - r[0][1]) / S;
        q[1] = (r[0][2] + r[2][0]) / S;
        q[2] = (r[1][2] + r[2][1]) / S;
        q[3] = T(0.25) * S;
    }
    return q;
}

// :186-203   returns the TRANSPOSE of the usual body->world matrix.
template <typename T> M3<T> quaternionToRotationMatrix(const Q4<T> &q)
{
    T e0 = q[0], e1 = q[1], e2 = q[2], e3 = q[3];
    M3<T> R;
    R[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); R[0][1] = 2 * (e1 * e2 - e0 * e3); R[0][2] = 2 * (e1 * e3 + e0 * e2);
    R[1][0] = 2 * (e1 * e2 + e0 * e3); R[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); R[1][2] = 2 * (e2 * e3 - e0 * e1);
    R[2][0] = 2 * (e1 * e3 - e0 * e2); R[2][1] = 2 * (e2 * e3 + e0 * e1); R[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
    return transpose(R);
}

template <typename T> Q4<T> rpyToQuat(const V3<T> &rpy) { return rotationMatrixToQuaternion(rpyToRotMat(rpy)); }

// :291-302
template <typename T> Q4<T> quatProduct(const Q4<T> &q1, const Q4<T> &q2)
{
    T r1 = q1[0], r2 = q2[0];
    T v1[3] = {q1[1], q1[2], q1[3]}, v2[3] = {q2[1], q2[2], q2[3]};
    T r = r1 * r2 - (v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]);
    T cx = v1[1] * v2[2] - v1[2] * v2[1];
    T cy = v1[2] * v2[0] - v1[0] * v2[2];
    T cz = v1[0] * v2[1] - v1[1] * v2[0];
    Q4<T> q;
    q[0] = r;
    q[1] = r1 * v2[0] + r2 * v1[0] + cx;
    q[2] = r1 * v2[1] + r2 * v1[1] + cy;
    q[3] = r1 * v2[2] + r2 * v1[2] + cz;
    return q;
}

// :383-397
template <typename T> V3<T> quaternionToso3(const Q4<T> &quat)
{
    V3<T> so3 = {{quat[1], quat[2], quat[3]}};
    T theta = T(2.0) * std::asin(std::sqrt(so3[0] * so3[0] + so3[1] * so3[1] + so3[2] * so3[2]));
    if (std::fabs(theta) < T(0.0000001)) {
        so3 = {{0, 0, 0}};
        return so3;
    }
    T s = std::sin(theta / T(2.0));
    for (int i = 0; i < 3; ++i) so3[i] /= s;
    for (int i = 0; i < 3; ++i) so3[i] *= theta;
    return so3;
}

// ---------------------------------------------------------------------------
// Thin SVD by one-sided Jacobi.  Works on W = A (rows>=cols) or A^T, rotating
// column pairs until mutually orthogonal; singular values are the column norms.
// Eigen::JacobiSVD is a two-sided Jacobi with QR preconditioning -- a different
// route to the same (unique up to signs/order) decomposition.
// ---------------------------------------------------------------------------
template <typename T> static void one_sided_jacobi(Mat<T> &W, Mat<T> &V)
{
    const int m = W.r, n = W.c;
    V = Mat<T>::Identity(n);
    const T eps = std::numeric_limits<T>::epsilon();
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                T alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < m; ++i) {
                    alpha += W(i, p) * W(i, p);
                    beta += W(i, q) * W(i, q);
                    gamma += W(i, p) * W(i, q);
                }
                if (std::fabs(gamma) <= eps * std::sqrt(alpha * beta) || gamma == T(0)) continue;
                rotated = true;
                T zeta = (beta - alpha) / (T(2) * gamma);
                T t = (zeta >= 0 ? T(1) : T(-1)) / (std::fabs(zeta) + std::sqrt(T(1) + zeta * zeta));
                T c = T(1) / std::sqrt(T(1) + t * t), s = c * t;
                for (int i = 0; i < m; ++i) {
                    T wp = W(i, p), wq = W(i, q);
                    W(i, p) = c * wp - s * wq;
                    W(i, q) = s * wp + c * wq;
                }
                for (int i = 0; i < n; ++i) {
                    T vp = V(i, p), vq = V(i, q);
                    V(i, p) = c * vp - s * vq;
                    V(i, q) = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
}

template <typename T> void jacobiSVD(const Mat<T> &A, Mat<T> &U, std::vector<T> &s, Mat<T> &V)
{
    const bool tall = A.r >= A.c;
    Mat<T> W = tall ? A : A.t();      // W is (max x min)
    Mat<T> Vw;
    one_sided_jacobi(W, Vw);
    const int k = W.c, m = W.r;
    s.assign(k, T(0));
    Mat<T> Uw(m, k);
    for (int j = 0; j < k; ++j) {
        T nrm = 0;
        for (int i = 0; i < m; ++i) nrm += W(i, j) * W(i, j);
        nrm = std::sqrt(nrm);
        s[j] = nrm;
        for (int i = 0; i < m; ++i) Uw(i, j) = nrm > T(0) ? W(i, j) / nrm : T(0);
    }
    if (tall) { U = Uw; V = Vw; } else { U = Vw; V = Uw; }
}

// QI/utils/qr_algebra.h:119-141
template <typename T> void pseudoInverse(const Mat<T> &matrix, double sigmaThreshold, Mat<T> &invMatrix)
{
    if (matrix.rows() == 1 && matrix.cols() == 1) {
        invMatrix = Mat<T>(1, 1);
        if (matrix(0, 0) > sigmaThreshold) invMatrix(0, 0) = T(1.0 / matrix(0, 0));   // compares the ENTRY, not |entry| (quirk 7)
        else invMatrix(0, 0) = T(0);
        return;
    }
    Mat<T> U, V;
    std::vector<T> s;
    jacobiSVD(matrix, U, s, V);
    const int k = (int)s.size();
    Mat<T> invS(k, k);
    for (int i = 0; i < k; ++i)
        if (s[i] > sigmaThreshold) invS(i, i) = T(1.0 / s[i]);     // strict '>'
    invMatrix = V * invS * U.t();
}

// Eigen's dynamic-size .inverse() is PartialPivLU; restated as Gauss-Jordan with
// row pivoting on the largest magnitude.
template <typename T> Mat<T> luInverse(const Mat<T> &A)
{
    const int n = A.r;
    assert(A.r == A.c);
    Mat<T> LU = A;
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k; T best = std::fabs(LU(k, k));
        for (int i = k + 1; i < n; ++i) if (std::fabs(LU(i, k)) > best) { best = std::fabs(LU(i, k)); piv = i; }
        if (piv != k) {
            for (int j = 0; j < n; ++j) std::swap(LU(k, j), LU(piv, j));
            std::swap(perm[k], perm[piv]);
        }
        for (int i = k + 1; i < n; ++i) {
            LU(i, k) /= LU(k, k);
            T l = LU(i, k);
            for (int j = k + 1; j < n; ++j) LU(i, j) -= l * LU(k, j);
        }
    }
    Mat<T> inv(n, n);
    for (int col = 0; col < n; ++col) {
        std::vector<T> y(n);
        for (int i = 0; i < n; ++i) {
            T s = (perm[i] == col) ? T(1) : T(0);
            for (int j = 0; j < i; ++j) s -= LU(i, j) * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            T s = y[i];
            for (int j = i + 1; j < n; ++j) s -= LU(i, j) * inv(j, col);
            inv(i, col) = s / LU(i, i);
        }
    }
    return inv;
}

#define QRO_INST(T)                                                     \
    template M3<T> coordinateRotation<T>(int, T);                       \
    template M3<T> mul<T>(const M3<T> &, const M3<T> &);                \
    template V3<T> mul<T>(const M3<T> &, const V3<T> &);                \
    template M3<T> transpose<T>(const M3<T> &);                         \
    template M3<T> rpyToRotMat<T>(const V3<T> &);                       \
    template Q4<T> rotationMatrixToQuaternion<T>(const M3<T> &);        \
    template M3<T> quaternionToRotationMatrix<T>(const Q4<T> &);        \
    template Q4<T> rpyToQuat<T>(const V3<T> &);                         \
    template Q4<T> quatProduct<T>(const Q4<T> &, const Q4<T> &);        \
    template V3<T> quaternionToso3<T>(const Q4<T> &);                   \
    template void jacobiSVD<T>(const Mat<T> &, Mat<T> &, std::vector<T> &, Mat<T> &); \
    template void pseudoInverse<T>(const Mat<T> &, double, Mat<T> &);   \
    template Mat<T> luInverse<T>(const Mat<T> &);
QRO_INST(float)
QRO_INST(double)

}  // namespace qro
