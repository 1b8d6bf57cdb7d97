// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Floating-base rigid-body model of the quadruped, restating
//   QS/dynamics/floating_base_model.cpp:469-806   (FK, bias accelerations, CRBA,
//       gravity, Coriolis, contact Jacobians)
//   QI/dynamics/spatial.hpp                        (spatial algebra, SpatialInertia)
//   QS/robots/qr_robot_a1_sim.cpp:176-343          (BuildDynamicModel constants;
//       qr_robot_lite3_sim.cpp:176-343 is literally identical)
//   QS/robots/qr_robot.cpp:89-103                  (WithLegSigns)
// with generic 6x6 matrices, exactly as the reference does (no sparsity tricks).
#include "qr_oracle.h"

namespace qro {

namespace {

template <typename T> using M6 = Mat<T>;

template <typename T> Mat<T> toMat(const M3<T> &a) { Mat<T> m(3, 3); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = a[i][j]; return m; }
template <typename T> Mat<T> skew(const T v[3])     // vectorToSkewMat, QI/utils/qr_se3.h:119-127
{
    Mat<T> m(3, 3);
    m(0, 1) = -v[2]; m(0, 2) = v[1];
    m(1, 0) = v[2];  m(1, 2) = -v[0];
    m(2, 0) = -v[1]; m(2, 1) = v[0];
    return m;
}
template <typename T> void matToSkewVec(const Mat<T> &m, T out[3])   // :130-135
{
    out[0] = T(0.5) * (m(2, 1) - m(1, 2));
    out[1] = T(0.5) * (m(0, 2) - m(2, 0));
    out[2] = T(0.5) * (m(1, 0) - m(0, 1));
}
// spatial.hpp:146-158
template <typename T> M6<T> createSXform(const Mat<T> &R, const T r[3])
{
    M6<T> X(6, 6);
    X.setBlock(0, 0, R);
    X.setBlock(3, 3, R);
    X.setBlock(3, 0, -(R * skew(r)));
    return X;
}
// spatial.hpp:27-36
template <typename T> M6<T> spatialRotation(int axis, T theta)
{
    Mat<T> R = toMat(coordinateRotation<T>(axis, theta));
    M6<T> X(6, 6);
    X.setBlock(0, 0, R);
    X.setBlock(3, 3, R);
    return X;
}
// spatial.hpp:183-192 (rotationFromSXform / translationFromSXform / invertSXform)
template <typename T> M6<T> invertSXform(const M6<T> &X)
{
    Mat<T> R = X.block(0, 0, 3, 3);
    T rr[3];
    matToSkewVec(R.t() * X.block(3, 0, 3, 3), rr);
    T r[3] = {-rr[0], -rr[1], -rr[2]};
    Mat<T> rv(3, 1); rv[0] = r[0]; rv[1] = r[1]; rv[2] = r[2];
    Mat<T> mRr = -(R * rv);
    T t[3] = {mRr[0], mRr[1], mRr[2]};
    return createSXform(R.t(), t);
}
// spatial.hpp:343-356
template <typename T> void sXFormPoint(const M6<T> &X, const T p[3], T out[3])
{
    Mat<T> R = X.block(0, 0, 3, 3);
    T rr[3];
    matToSkewVec(R.t() * X.block(3, 0, 3, 3), rr);
    Mat<T> d(3, 1);
    for (int i = 0; i < 3; ++i) d[i] = p[i] - (-rr[i]);
    Mat<T> Xp = R * d;
    for (int i = 0; i < 3; ++i) out[i] = Xp[i];
}
// spatial.hpp:63-75
template <typename T> Mat<T> motionCrossProduct(const Mat<T> &a, const Mat<T> &b)
{
    Mat<T> mv(6, 1);
    mv[0] = a[1] * b[2] - a[2] * b[1];
    mv[1] = a[2] * b[0] - a[0] * b[2];
    mv[2] = a[0] * b[1] - a[1] * b[0];
    mv[3] = a[1] * b[5] - a[2] * b[4] + a[4] * b[2] - a[5] * b[1];
    mv[4] = a[2] * b[3] - a[0] * b[5] - a[3] * b[2] + a[5] * b[0];
    mv[5] = a[0] * b[4] - a[1] * b[3] + a[3] * b[1] - a[4] * b[0];
    return mv;
}
// spatial.hpp:78-90
template <typename T> Mat<T> forceCrossProduct(const Mat<T> &a, const Mat<T> &b)
{
    Mat<T> mv(6, 1);
    mv[0] = b[2] * a[1] - b[1] * a[2] - b[4] * a[5] + b[5] * a[4];
    mv[1] = b[0] * a[2] - b[2] * a[0] + b[3] * a[5] - b[5] * a[3];
    mv[2] = b[1] * a[0] - b[0] * a[1] - b[3] * a[4] + b[4] * a[3];
    mv[3] = b[5] * a[1] - b[4] * a[2];
    mv[4] = b[3] * a[2] - b[5] * a[0];
    mv[5] = b[4] * a[0] - b[3] * a[1];
    return mv;
}

// SpatialInertia(mass, com, inertia), spatial.hpp:390-398
template <typename T> M6<T> spatialInertia(T mass, const T com[3], const Mat<T> &I)
{
    Mat<T> c = skew(com);
    M6<T> S(6, 6);
    S.setBlock(0, 0, I + mass * (c * c.t()));
    S.setBlock(0, 3, mass * c);
    S.setBlock(3, 0, mass * c.t());
    S.setBlock(3, 3, mass * Mat<T>::Identity(3));
    return S;
}
// flipAlongAxis(Y): getPseudoInertia -> X P X -> SpatialInertia(Mat4), spatial.hpp:437-449,505-534
template <typename T> M6<T> flipAlongY(const M6<T> &S)
{
    T h[3];
    matToSkewVec(S.block(0, 3, 3, 3), h);
    Mat<T> Ibar = S.block(0, 0, 3, 3);
    T m = S(5, 5);
    Mat<T> P(4, 4);
    T tr = Ibar(0, 0) + Ibar(1, 1) + Ibar(2, 2);
    P.setBlock(0, 0, T(0.5) * tr * Mat<T>::Identity(3) - Ibar);
    for (int i = 0; i < 3; ++i) { P(i, 3) = h[i]; P(3, i) = h[i]; }
    P(3, 3) = m;
    Mat<T> X = Mat<T>::Identity(4);
    X(1, 1) = T(-1);
    P = X * P * X;
    // SpatialInertia(const Mat4&)
    T hh[3] = {P(0, 3), P(1, 3), P(2, 3)};
    Mat<T> E = P.block(0, 0, 3, 3);
    T trE = E(0, 0) + E(1, 1) + E(2, 2);
    M6<T> out(6, 6);
    out.setBlock(0, 0, trE * Mat<T>::Identity(3) - E);
    out.setBlock(0, 3, skew(hh));
    out.setBlock(3, 0, skew(hh).t());
    out.setBlock(3, 3, P(3, 3) * Mat<T>::Identity(3));
    return out;
}

template <typename T> void withLegSigns(const T v[3], int leg, T out[3])   // QS/robots/qr_robot.cpp:89-103
{
    const T sx[4] = {1, 1, -1, -1}, sy[4] = {-1, 1, -1, 1};
    out[0] = sx[leg] * v[0]; out[1] = sy[leg] * v[1]; out[2] = v[2];
}

}  // namespace

// The static tree (what BuildDynamicModel constructs).  Indices follow the
// reference: 0-4 unused, 5 floating base, 6+3l / 7+3l / 8+3l = abad/hip/knee of leg l.
template <typename T> struct FBModel {
    int parent[18];
    int axis[18];
    M6<T> Xtree[18], Xrot[18], Ibody[18], Irot[18];
    T gear[18];
    int gcParent[4];
    T gcLoc[4][3];
    T gravity[3] = {0, 0, T(-9.81)};

    explicit FBModel(const ModelDesc &md)
    {
        // literals: qr_robot_a1_sim.cpp:184-271 (float literals kept as float, as in the reference)
        const T abadRotorLoc[3] = {T(0.14f), T(0.047f), T(0.f)};
        const T abadLoc[3] = {T(0.1805f), T(0.047f), T(0.f)};
        const T hipLoc[3] = {T(0), T(md.hip_l), T(0)};
        const T hipRotorLoc[3] = {T(0), T((float)0.04), T(0)};
        const T kneeLoc[3] = {T(0), T(0), T(-md.upper_l)};
        const T kneeRotorLoc[3] = {T(0), T(0), T(0)};
        const float scale_ = 1e-2;
        // rotorRotationalInertiaZ.setIdentity(); then * scale_*1e-6  (:193-198)
        Mat<T> rotZ = T((float)(scale_ * 1e-6)) * Mat<T>::Identity(3);
        Mat<T> RY = toMat(coordinateRotation<T>(1, T((float)(M_PI / 2))));
        Mat<T> RX = toMat(coordinateRotation<T>(0, T((float)(M_PI / 2))));
        Mat<T> rotX = RY * rotZ * RY.t();
        Mat<T> rotY = RX * rotZ * RX.t();

        auto m3 = [](std::initializer_list<double> v, double s) {
            Mat<T> m(3, 3); int k = 0;
            for (double x : v) { m.d[k++] = T((float)x) * T((float)s); }
            return m;
        };
        Mat<T> abadI = m3({469.2, -9.4, -0.342, -9.4, 807.5, -0.466, -0.342, -0.466, 552.9}, 1e-6);
        const T abadCOM[3] = {T(-0.0033f), T(0), T(0)};
        M6<T> abadInertia = spatialInertia<T>(T(0.696f), abadCOM, abadI);
        Mat<T> hipI = m3({5529, 4.825, 343.9, 4.825, 5139.3, 22.4, 343.9, 22.4, 1367.8}, 1e-6);
        const T hipCOM[3] = {T(-0.003237f), T(-0.022327f), T(-0.027326f)};
        M6<T> hipInertia = spatialInertia<T>(T(1.013f), hipCOM, hipI);
        Mat<T> kneeI = m3({2998, 0, -141.2, 0, 3014, 0, -141.2, 0, 32.4}, 1e-6);
        const T kneeCOM[3] = {T(0.006435f), T(0), T(-0.107f)};
        M6<T> kneeInertia = spatialInertia<T>(T(0.166f), kneeCOM, kneeI);
        const T rotorCOM[3] = {0, 0, 0};
        const T rotorMass = T(1e-8f);
        M6<T> rotorInertiaX = spatialInertia<T>(rotorMass, rotorCOM, rotX);
        M6<T> rotorInertiaY = spatialInertia<T>(rotorMass, rotorCOM, rotY);
        Mat<T> bodyI = m3({15853, 0, 0, 0, 37799, 0, 0, 0, 45654}, 1e-6);
        const T bodyCOM[3] = {0, 0, 0};
        M6<T> bodyInertia = spatialInertia<T>(T(6), bodyCOM, bodyI);

        M6<T> eye6 = Mat<T>::Identity(6), zero6(6, 6);
        for (int i = 0; i < 18; ++i) { parent[i] = 0; axis[i] = 0; Xtree[i] = eye6; Xrot[i] = eye6; Ibody[i] = zero6; Irot[i] = zero6; gear[i] = 0; }
        Ibody[5] = bodyInertia; gear[5] = 1;     // addBase, floating_base_model.cpp:257-290

        Mat<T> I3 = Mat<T>::Identity(3);
        Mat<T> Rz_pi = toMat(coordinateRotation<T>(2, T((float)M_PI)));
        const T kneeLinkY = T(0.004f);
        int body = 5; T sideSign = -1;
        for (int leg = 0; leg < 4; ++leg) {
            T loc[3];
            // abad (:281-293)
            ++body;
            withLegSigns(abadLoc, leg, loc);      Xtree[body] = createSXform(I3, loc);
            withLegSigns(abadRotorLoc, leg, loc); Xrot[body] = createSXform(I3, loc);
            Ibody[body] = sideSign < 0 ? flipAlongY(abadInertia) : abadInertia;
            Irot[body] = sideSign < 0 ? flipAlongY(rotorInertiaX) : rotorInertiaX;
            gear[body] = 1; parent[body] = 5; axis[body] = 0;
            // hip (:296-312)
            ++body;
            withLegSigns(hipLoc, leg, loc);      Xtree[body] = createSXform(I3, loc);
            withLegSigns(hipRotorLoc, leg, loc); Xrot[body] = createSXform(Rz_pi, loc);
            Ibody[body] = sideSign < 0 ? flipAlongY(hipInertia) : hipInertia;
            Irot[body] = sideSign < 0 ? flipAlongY(rotorInertiaY) : rotorInertiaY;
            gear[body] = 1; parent[body] = body - 1; axis[body] = 1;
            // knee (:318-334); the link inertia is NOT flipped (flip commented out at :322)
            ++body;
            Xtree[body] = createSXform(I3, kneeLoc);
            Xrot[body] = createSXform(I3, kneeRotorLoc);
            Ibody[body] = kneeInertia;
            Irot[body] = sideSign < 0 ? flipAlongY(rotorInertiaY) : rotorInertiaY;
            gear[body] = 1; parent[body] = body - 1; axis[body] = 1;
            // foot contact point ids 9/11/13/15 (:327,:333)
            gcParent[leg] = body;
            gcLoc[leg][0] = 0; gcLoc[leg][1] = sideSign < 0 ? kneeLinkY : -kneeLinkY; gcLoc[leg][2] = T(-md.lower_l);
            sideSign *= -1;
        }
    }
};

template <typename T> void fb_compute(const ModelDesc &md, const FBState<T> &st, FBResult<T> &out)
{
    FBModel<T> M(md);
    const int nDof = 18;
    M6<T> Xup[18], Xuprot[18], Xa[18];
    Mat<T> S[18], Srot[18], v[18], vrot[18], c[18], crot[18], avp[18], avprot[18];
    M6<T> IC[18];

    // ---- forwardKinematics (:469-524)
    Q4<T> quat = {{st.quat[0], st.quat[1], st.quat[2], st.quat[3]}};
    Mat<T> R = toMat(quaternionToRotationMatrix(quat));
    Xup[5] = createSXform(R, st.pos);
    v[5] = Mat<T>(6, 1);
    for (int i = 0; i < 6; ++i) v[5][i] = st.bodyVel[i];
    for (int i = 6; i < nDof; ++i) {
        M6<T> XJ = spatialRotation<T>(M.axis[i], st.q[i - 6]);      // jointXform(Revolute), spatial.hpp:230-249
        Xup[i] = XJ * M.Xtree[i];
        S[i] = Mat<T>(6, 1); S[i][M.axis[i]] = T(1);                 // jointMotionSubspace, :205-224
        Mat<T> vJ = st.qd[i - 6] * S[i];
        v[i] = Xup[i] * v[M.parent[i]] + vJ;
        M6<T> XJrot = spatialRotation<T>(M.axis[i], st.q[i - 6] * M.gear[i]);
        Srot[i] = M.gear[i] * S[i];
        Mat<T> vJrot = st.qd[i - 6] * Srot[i];
        Xuprot[i] = XJrot * M.Xrot[i];
        vrot[i] = Xuprot[i] * v[M.parent[i]] + vJrot;
        c[i] = motionCrossProduct(v[i], vJ);
        crot[i] = motionCrossProduct(vrot[i], vJrot);
    }
    for (int i = 5; i < nDof; ++i) Xa[i] = (M.parent[i] == 0) ? Xup[i] : Xup[i] * Xa[M.parent[i]];
    for (int k = 0; k < 4; ++k) {
        int i = M.gcParent[k];
        M6<T> Xai = invertSXform(Xa[i]);
        Mat<T> vSp = Xai * v[i];
        sXFormPoint(Xai, M.gcLoc[k], out.pGC[k]);
        // spatialToLinearVelocity: vLin + vAng x p   (spatial.hpp:279-291)
        const T *p = out.pGC[k];
        out.vGC[k][0] = vSp[3] + (vSp[1] * p[2] - vSp[2] * p[1]);
        out.vGC[k][1] = vSp[4] + (vSp[2] * p[0] - vSp[0] * p[2]);
        out.vGC[k][2] = vSp[5] + (vSp[0] * p[1] - vSp[1] * p[0]);
    }

    // ---- biasAccelerations (:587-600)
    avp[5] = Mat<T>(6, 1);
    for (int i = 6; i < nDof; ++i) {
        avp[i] = Xup[i] * avp[M.parent[i]] + c[i];
        avprot[i] = Xuprot[i] * avp[M.parent[i]] + crot[i];
    }

    // ---- contactJacobians (:541-580), feet only
    for (int k = 0; k < 4; ++k) {
        out.Jc[k] = Mat<T>(3, 18);
        out.Jcdqd[k] = Mat<T>(3, 1);
        int i = M.gcParent[k];
        Mat<T> Rai = Xa[i].block(0, 0, 3, 3).t();
        M6<T> Xc = createSXform(Rai, M.gcLoc[k]);
        Mat<T> ac = Xc * avp[i];
        Mat<T> vc = Xc * v[i];
        // spatialToLinearAcceleration(a, v) = a.tail(3) + v.head(3) x v.tail(3)   (spatial.hpp:304-316)
        out.Jcdqd[k][0] = ac[3] + (vc[1] * vc[5] - vc[2] * vc[4]);
        out.Jcdqd[k][1] = ac[4] + (vc[2] * vc[3] - vc[0] * vc[5]);
        out.Jcdqd[k][2] = ac[5] + (vc[0] * vc[4] - vc[1] * vc[3]);
        Mat<T> Xout = Xc.block(3, 0, 3, 6);
        while (i > 5) {
            Mat<T> col = Xout * S[i];
            for (int r = 0; r < 3; ++r) out.Jc[k](r, i) = col[r];
            Xout = Xout * Xup[i];
            i = M.parent[i];
        }
        out.Jc[k].setBlock(0, 0, Xout);
    }

    // ---- compositeInertias (:750-767)
    for (int i = 5; i < nDof; ++i) IC[i] = M.Ibody[i];
    for (int i = nDof - 1; i > 5; --i) {
        IC[M.parent[i]] = IC[M.parent[i]] + Xup[i].t() * IC[i] * Xup[i];
        IC[M.parent[i]] = IC[M.parent[i]] + Xuprot[i].t() * M.Irot[i] * Xuprot[i];
    }

    // ---- massMatrix (:774-806)
    out.H = Mat<T>(18, 18);
    out.H.setBlock(0, 0, IC[5]);
    for (int j = 6; j < nDof; ++j) {
        Mat<T> f = IC[j] * S[j];
        Mat<T> frot = M.Irot[j] * Srot[j];
        T hjj = 0, hr = 0;
        for (int k = 0; k < 6; ++k) { hjj += S[j][k] * f[k]; hr += Srot[j][k] * frot[k]; }
        out.H(j, j) = hjj + hr;
        f = Xup[j].t() * f + Xuprot[j].t() * frot;
        int i = M.parent[j];
        while (i > 5) {
            T hij = 0;
            for (int k = 0; k < 6; ++k) hij += S[i][k] * f[k];
            out.H(i, j) = hij;
            out.H(j, i) = hij;
            f = Xup[i].t() * f;
            i = M.parent[i];
        }
        for (int k = 0; k < 6; ++k) { out.H(k, j) = f[k]; out.H(j, k) = f[k]; }
    }

    // ---- generalizedGravityForce (:607-626)
    {
        Mat<T> ag[18], agrot[18];
        Mat<T> aG(6, 1);
        aG[3] = M.gravity[0]; aG[4] = M.gravity[1]; aG[5] = M.gravity[2];
        ag[5] = Xup[5] * aG;
        Mat<T> top = -(IC[5] * ag[5]);
        out.G = Mat<T>(18, 1);
        for (int k = 0; k < 6; ++k) out.G[k] = top[k];
        for (int i = 6; i < nDof; ++i) {
            ag[i] = Xup[i] * ag[M.parent[i]];
            agrot[i] = Xuprot[i] * ag[M.parent[i]];
            Mat<T> a = IC[i] * ag[i], b = M.Irot[i] * agrot[i];
            T s1 = 0, s2 = 0;
            for (int k = 0; k < 6; ++k) { s1 += S[i][k] * a[k]; s2 += Srot[i][k] * b[k]; }
            out.G[i] = -s1 - s2;
        }
    }

    // ---- generalizedCoriolisForce (:633-665)
    {
        Mat<T> fvp[18], fvprot[18];
        Mat<T> hfb = M.Ibody[5] * v[5];
        fvp[5] = M.Ibody[5] * avp[5] + forceCrossProduct(v[5], hfb);
        for (int i = 6; i < nDof; ++i) {
            Mat<T> hi = M.Ibody[i] * v[i];
            fvp[i] = M.Ibody[i] * avp[i] + forceCrossProduct(v[i], hi);
            Mat<T> hr = M.Irot[i] * vrot[i];
            fvprot[i] = M.Irot[i] * avprot[i] + forceCrossProduct(vrot[i], hr);
        }
        out.C = Mat<T>(18, 1);
        for (int i = nDof - 1; i > 5; --i) {
            T s1 = 0, s2 = 0;
            for (int k = 0; k < 6; ++k) { s1 += S[i][k] * fvp[i][k]; s2 += Srot[i][k] * fvprot[i][k]; }
            out.C[i] = s1 + s2;
            fvp[M.parent[i]] = fvp[M.parent[i]] + Xup[i].t() * fvp[i];
            fvp[M.parent[i]] = fvp[M.parent[i]] + Xuprot[i].t() * fvprot[i];
        }
        for (int k = 0; k < 6; ++k) out.C[k] = fvp[5][k];
    }

    // totalNonRotorMass (:447-453)
    T tm = 0;
    for (int i = 0; i < nDof; ++i) tm += M.Ibody[i](5, 5);
    out.totalNonRotorMass = tm;
}

template struct FBModel<float>;
template struct FBModel<double>;
template void fb_compute<float>(const ModelDesc &, const FBState<float> &, FBResult<float> &);
template void fb_compute<double>(const ModelDesc &, const FBState<double> &, FBResult<double> &);

}  // namespace qro
