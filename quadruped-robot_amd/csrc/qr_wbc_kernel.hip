// ============================================================================
// Whole-body-control tick for a batch of quadrupeds: two 64-lane wavefronts
// (= one workgroup) per robot, all matrices LDS-resident, fp64 arithmetic on the
// fp32 inputs.  gfx950 (MI355X) only.
//
// Replaces, per robot (reference: TopHillRobotics/quadruped-robot, QS/ = quadruped/src/):
//   K8  FloatingBaseModel::forwardKinematics / biasAccelerations   QS/dynamics/floating_base_model.cpp:469-524,587-600
//   K9  compositeInertias / massMatrix / generalizedGravityForce / generalizedCoriolisForce   :750-806,607-665
//   K10 contactJacobians                                            :541-580
//   K11 task_set/*.cpp UpdateTask, qrSingleContact::UpdateContactSpec, ContactTaskUpdate
//       (QS/controllers/wbc/qr_wbc_locomotion_controller.cpp:172-201)
//   K12 qrMultitaskProjection::FindConfiguration                    QS/controllers/wbc/qr_multitask_projection.cpp:38-106
//   K13 qrWholeBodyImpulseCtrl::GetModelRes / MakeTorque            QS/controllers/wbc/qr_wholebody_impulse_ctrl.cpp:50-299
//   K14 UpdateLegCMD (stance legs take the WBC torque)              qr_wbc_locomotion_controller.cpp:205-219
//
// Structure: wave 0 carries the torque chain (dynamics -> A^-1 -> acceleration recursion -> relaxation QP), wave 1 the velocity-dependent
// terms, the task set and the kinematic projection K12 (see the kernel's header).  Lanes 0-3 of each walk one leg through the kinematic tree
// with 3-vector / rigid-body-inertia (m, h, Ibar) algebra instead of generic 6x6 products; the 18x18 / n x 18 dense algebra (A^-1 through
// the base's Schur complement, null-space recursions over three-column task Jacobians, Gram-matrix pseudo-inverses with an eigenvalue
// guard + Jacobi fallback that reproduces pseudoInverse()'s singular-value cut, the relaxation QP by a Schur-complement Goldfarb-Idnani)
// runs lane-parallel over matrix elements.
// Rotor bodies (1e-8 kg, gear 1): their constant isotropic inertia is folded into the
// parent link on the host, the +k on H(j,j) and the k*axis coupling term are kept,
// their gravity term is exactly zero and their Coriolis term (<= 1e-7 N m) is dropped.
// ============================================================================
#include <hip/hip_runtime.h>
#include <type_traits>
#include "qr_device_types.h"
#include "qr_wave_helpers.h"

namespace qrgpu {

// Diagnostic hooks of the pipelined tick's timeline: compiled in only with -DQR_TIMELINE (see qr_mpc_kernel.hip)
#ifdef QR_TIMELINE
#define QW_P_TL pipe.tl
#define QW_P_TLR pipe.tlr
#else
#define QW_P_TL ((long long *)nullptr)
#define QW_P_TLR ((int *)nullptr)
#endif

typedef double real;

struct v3 { real x, y, z; };
__device__ __forceinline__ v3 mk(real x, real y, real z) { v3 r = {x, y, z}; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(real s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ real dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
struct m3 { real m[3][3]; };
__device__ __forceinline__ v3 mul(const m3 &A, v3 b)
{
    return mk(A.m[0][0] * b.x + A.m[0][1] * b.y + A.m[0][2] * b.z, A.m[1][0] * b.x + A.m[1][1] * b.y + A.m[1][2] * b.z,
              A.m[2][0] * b.x + A.m[2][1] * b.y + A.m[2][2] * b.z);
}
__device__ __forceinline__ v3 mulT(const m3 &A, v3 b)
{
    return mk(A.m[0][0] * b.x + A.m[1][0] * b.y + A.m[2][0] * b.z, A.m[0][1] * b.x + A.m[1][1] * b.y + A.m[2][1] * b.z,
              A.m[0][2] * b.x + A.m[1][2] * b.y + A.m[2][2] * b.z);
}
__device__ __forceinline__ m3 mul(const m3 &A, const m3 &B)
{
    m3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}
__device__ __forceinline__ m3 transpose(const m3 &A)
{
    m3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[j][i];
    return C;
}
// coordinateRotation (QI/utils/qr_se3.h:72-89): the coordinate-transform (transposed) matrix.
__device__ __forceinline__ m3 coord_rot(int axis, real th)
{
    real s, c;
    sincos(th, &s, &c);
    m3 R;
    if (axis == 0)      { R = {{{1, 0, 0}, {0, c, s}, {0, -s, c}}}; }
    else if (axis == 1) { R = {{{c, 0, -s}, {0, 1, 0}, {s, 0, c}}}; }
    else                { R = {{{c, s, 0}, {-s, c, 0}, {0, 0, 1}}}; }
    return R;
}
__device__ __forceinline__ m3 coord_rot_sc(int axis, real s, real c)
{
    m3 R;
    if (axis == 0)      { R = {{{1, 0, 0}, {0, c, s}, {0, -s, c}}}; }
    else if (axis == 1) { R = {{{c, 0, -s}, {0, 1, 0}, {s, 0, c}}}; }
    else                { R = {{{c, s, 0}, {-s, c, 0}, {0, 0, 1}}}; }
    return R;
}
// quaternionToRotationMatrix (:186-203): world -> body.
__device__ __forceinline__ m3 quat_to_rot_wb(const real *q)
{
    const real e0 = q[0], e1 = q[1], e2 = q[2], e3 = q[3];
    m3 R;
    R.m[0][0] = 1 - 2 * (e2 * e2 + e3 * e3); R.m[1][0] = 2 * (e1 * e2 - e0 * e3); R.m[2][0] = 2 * (e1 * e3 + e0 * e2);
    R.m[0][1] = 2 * (e1 * e2 + e0 * e3); R.m[1][1] = 1 - 2 * (e1 * e1 + e3 * e3); R.m[2][1] = 2 * (e2 * e3 - e0 * e1);
    R.m[0][2] = 2 * (e1 * e3 - e0 * e2); R.m[1][2] = 2 * (e2 * e3 + e0 * e1); R.m[2][2] = 1 - 2 * (e1 * e1 + e2 * e2);
    return R;
}

// Rigid-body spatial inertia [[Ibar, [h]x],[[h]x^T, m 1]] as (m, h, Ibar sym: xx yy zz xy xz yz).
struct rbi { real m; v3 h; real I[6]; };
__device__ __forceinline__ rbi rbi_load(const real *p)
{
    rbi r; r.m = p[0]; r.h = mk(p[1], p[2], p[3]);
#pragma unroll
    for (int i = 0; i < 6; ++i) r.I[i] = p[4 + i];
    return r;
}
__device__ __forceinline__ v3 rbi_Iw(const rbi &a, v3 w)
{
    return mk(a.I[0] * w.x + a.I[3] * w.y + a.I[4] * w.z, a.I[3] * w.x + a.I[1] * w.y + a.I[5] * w.z, a.I[4] * w.x + a.I[5] * w.y + a.I[2] * w.z);
}
__device__ __forceinline__ rbi rbi_add(const rbi &a, const rbi &b)
{
    rbi r; r.m = a.m + b.m; r.h = a.h + b.h;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.I[i] = a.I[i] + b.I[i];
    return r;
}
// Express a child-frame inertia in the parent frame: X^T I X with X = (E, r)  (createSXform(E, r)).
__device__ __forceinline__ rbi rbi_to_parent(const rbi &a, const m3 &E, v3 r)
{
    rbi o;
    o.m = a.m;
    const v3 hr = mulT(E, a.h);                    // E^T h
    o.h = hr + a.m * r;
    // Ibar' = E^T Ibar E - [r]x[hr]x - [h']x[r]x
    m3 I; I.m[0][0] = a.I[0]; I.m[1][1] = a.I[1]; I.m[2][2] = a.I[2];
    I.m[0][1] = I.m[1][0] = a.I[3]; I.m[0][2] = I.m[2][0] = a.I[4]; I.m[1][2] = I.m[2][1] = a.I[5];
    m3 Ir = mul(transpose(E), mul(I, E));
    // -[a]x[b]x = (a.b) 1 - b a^T
    auto add_outer = [&](v3 a_, v3 b_) {
        const real ab = dot(a_, b_);
        const real av[3] = {a_.x, a_.y, a_.z}, bv[3] = {b_.x, b_.y, b_.z};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Ir.m[i][j] += (i == j ? ab : 0.0) - bv[i] * av[j];
    };
    add_outer(r, hr);
    add_outer(o.h, r);
    o.I[0] = Ir.m[0][0]; o.I[1] = Ir.m[1][1]; o.I[2] = Ir.m[2][2];
    o.I[3] = 0.5 * (Ir.m[0][1] + Ir.m[1][0]); o.I[4] = 0.5 * (Ir.m[0][2] + Ir.m[2][0]); o.I[5] = 0.5 * (Ir.m[1][2] + Ir.m[2][1]);
    return o;
}
struct sv6 { v3 a, l; };      // spatial vector (angular; linear)
__device__ __forceinline__ sv6 xmotion(const m3 &E, v3 r, sv6 v) { sv6 o; o.a = mul(E, v.a); o.l = mul(E, v.l - cross(r, v.a)); return o; }   // X v
__device__ __forceinline__ sv6 xforceT(const m3 &E, v3 r, sv6 f) { sv6 o; o.l = mulT(E, f.l); o.a = mulT(E, f.a) + cross(r, o.l); return o; }   // X^T f
__device__ __forceinline__ sv6 rbi_mul(const rbi &I, sv6 v) { sv6 o; o.a = rbi_Iw(I, v.a) + cross(I.h, v.l); o.l = I.m * v.l - cross(I.h, v.a); return o; }
__device__ __forceinline__ sv6 crf(sv6 v, sv6 f) { sv6 o; o.a = cross(v.a, f.a) + cross(v.l, f.l); o.l = cross(v.a, f.l); return o; }   // v x* f
__device__ __forceinline__ sv6 crm(sv6 v, sv6 u) { sv6 o; o.a = cross(v.a, u.a); o.l = cross(v.a, u.l) + cross(v.l, u.a); return o; }   // v x u

__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ real wsum(real v) { return wave_sum_d(v); }

// sum_{t<k} a[t*as] b[t*bs], k <= 18: every load is issued before the first use
__device__ __forceinline__ real dot18(const real *a, int as, const real *b, int bs, int k)
{
    real av[18], bv[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) { const bool ok = t < k; av[t] = ok ? a[t * as] : 0.0; bv[t] = ok ? b[t * bs] : 0.0; }
    real s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < 18; t += 3) { s0 += av[t] * bv[t]; s1 += av[t + 1] * bv[t + 1]; s2 += av[t + 2] * bv[t + 2]; }
    return (s0 + s1) + s2;
}

// e / n for 0 <= e < 1024, 1 <= n <= 32 with rcp = ceil(65536 / n): one multiply and a shift.
__device__ __forceinline__ int fdiv16(int e, int rcp) { return (e * rcp) >> 16; }
__device__ __forceinline__ int rcp16(int n) { return (65536 + n - 1) / n; }

// C(m x n) = alpha * op(A)(m x k) * op(B)(k x n) + beta * C0 ; lane-parallel over outputs, ends with wsync.
// tA: A stored k x m (use A^T);  tB: B stored n x k (use B^T).  k <= KMAX: the k-loop is fully unrolled and
// predicated so that all 2k LDS loads of an output are in flight together (one wave per SIMD: latency is everything).
template <int KMAX>
__device__ __forceinline__ void gemm(int lane, real *C, int ldc, const real *A, int lda, bool tA, const real *B, int ldb, bool tB,
                                     int m, int n, int k, real alpha = 1.0, real beta = 0.0, const real *C0 = nullptr, int ldc0 = 0)
{
    const int rn = rcp16(n);
    const int as = tA ? lda : 1, bs = tB ? 1 : ldb;
    for (int e = lane; e < m * n; e += 64) {
        const int i = fdiv16(e, rn), j = e - i * n;
        const real *ap = tA ? A + i : A + i * lda;
        const real *bp = tB ? B + j * ldb : B + j;
        real av[KMAX], bv[KMAX];
#pragma unroll
        for (int t = 0; t < KMAX; ++t) { const bool ok = t < k; av[t] = ok ? ap[t * as] : 0.0; bv[t] = ok ? bp[t * bs] : 0.0; }
        real a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int t = 0; t < KMAX; t += 3) {
            a0 += av[t] * bv[t];
            if (t + 1 < KMAX) a1 += av[t + 1] * bv[t + 1];
            if (t + 2 < KMAX) a2 += av[t + 2] * bv[t + 2];
        }
        real v = alpha * ((a0 + a1) + a2);
        if (C0) v += beta * C0[i * ldc0 + j];
        C[i * ldc + j] = v;
    }
    wsync();
}

// In-place inverse of a symmetric positive definite n x n matrix (full storage, ld) by symmetric sweeps.
// Each lane keeps its <= UMAX elements (e = lane + 64 u) in registers for all n pivots; only the pivot column goes
// through LDS (`col`, n doubles).  n * n <= 64 * UMAX.  Returns (uniform) the smallest pivot seen.
template <int UMAX>
__device__ __forceinline__ real spd_inverse(int lane, real *A, int ld, int n, real *col)
{
    const int rn = rcp16(n);
    int ei[UMAX], ej[UMAX];
    real a[UMAX];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
        const int e = lane + 64 * u;
        const bool ok = e < n * n;
        ei[u] = ok ? fdiv16(e, rn) : -1;
        ej[u] = ok ? e - ei[u] * n : -1;
        a[u] = ok ? A[ei[u] * ld + ej[u]] : 0.0;
    }
    real minpiv = 1e300;
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int u = 0; u < UMAX; ++u) if (ej[u] == k) col[ei[u]] = a[u];
        wsync();
        const real piv = col[k];
        minpiv = piv < minpiv ? piv : minpiv;
        const real ip = fast_rcp(piv);
        real ci[UMAX], cj[UMAX];
#pragma unroll
        for (int u = 0; u < UMAX; ++u) { const bool ok = ei[u] >= 0; ci[u] = ok ? col[ei[u]] : 0.0; cj[u] = ok ? col[ej[u]] : 0.0; }
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (ei[u] == k) a[u] = (ej[u] == k) ? -ip : cj[u] * ip;
            else if (ej[u] == k) a[u] = ci[u] * ip;
            else a[u] -= ci[u] * cj[u] * ip;
        }
        wsync();                                   // col is rewritten by the next pivot
    }
#pragma unroll
    for (int u = 0; u < UMAX; ++u) if (ei[u] >= 0) A[ei[u] * ld + ej[u]] = -a[u];
    wsync();
    return minpiv;
}

// pseudoInverse() of a symmetric PSD matrix W (n x n, n <= 12) with eigenvalue cut `thr`
// (QI/utils/qr_algebra.h:119-141: singular values <= thr are dropped; strict '>').
// Fast path: plain inverse when every eigenvalue provably exceeds thr; otherwise Jacobi
// eigen-decomposition on lane 0.  Winv may not alias W.  scr: >= n*n + n doubles.
template <int UMAX = 3>
__device__ __forceinline__ void psd_pinv(int lane, const real *W, int n, real thr, real *Winv, real *scr)
{
    if (n == 1) {   // 1x1 special case compares the entry itself (quirk 7)
        if (lane == 0) Winv[0] = (W[0] > thr) ? 1.0 / W[0] : 0.0;
        wsync();
        return;
    }
    for (int e = lane; e < n * n; e += 64) Winv[e] = W[e];
    wsync();
    const real minpiv = spd_inverse<UMAX>(lane, Winv, n, n, scr);     // n * n <= 64 * UMAX (n <= 12: 144 elements)
    real fro = 0.0;
    for (int e = lane; e < n * n; e += 64) fro += Winv[e] * Winv[e];
    fro = wsum(fro);
    // lambda_min >= 1/||W^-1||_2 >= 1/||W^-1||_F
    const bool full_rank = (minpiv > 0.0) && (fro == fro) && (1.0 > thr * __builtin_sqrt(fro));
    if (full_rank) return;
    // Rank-revealing path (rare: kinematic singularities).  Cyclic Jacobi on lane 0.
    if (lane == 0) {
        real *Am = scr;            // n*n working copy
        real *V = Winv;            // eigenvectors accumulate here
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) { Am[i * n + j] = 0.5 * (W[i * n + j] + W[j * n + i]); V[i * n + j] = (i == j) ? 1.0 : 0.0; }
        for (int sweep = 0; sweep < 30; ++sweep) {
            real off = 0.0;
            for (int p = 0; p < n; ++p) for (int q2 = p + 1; q2 < n; ++q2) off += Am[p * n + q2] * Am[p * n + q2];
            if (off < 1e-40) break;
            for (int p = 0; p < n - 1; ++p)
                for (int q2 = p + 1; q2 < n; ++q2) {
                    const real apq = Am[p * n + q2];
                    if (apq == 0.0) continue;
                    const real th = (Am[q2 * n + q2] - Am[p * n + p]) / (2.0 * apq);
                    const real t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + __builtin_sqrt(th * th + 1.0));
                    const real c = 1.0 / __builtin_sqrt(t * t + 1.0), s = t * c;
                    for (int k = 0; k < n; ++k) { const real akp = Am[k * n + p], akq = Am[k * n + q2]; Am[k * n + p] = c * akp - s * akq; Am[k * n + q2] = s * akp + c * akq; }
                    for (int k = 0; k < n; ++k) { const real apk = Am[p * n + k], aqk = Am[q2 * n + k]; Am[p * n + k] = c * apk - s * aqk; Am[q2 * n + k] = s * apk + c * aqk; }
                    for (int k = 0; k < n; ++k) { const real vkp = V[k * n + p], vkq = V[k * n + q2]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q2] = s * vkp + c * vkq; }
                }
        }
        // Winv = V diag(1/l if l > thr) V^T   (singular values of a PSD matrix are |eigenvalues|)
        real *ev = scr + n * n;
        for (int i = 0; i < n; ++i) { const real l = fabs(Am[i * n + i]); ev[i] = (l > thr) ? 1.0 / Am[i * n + i] : 0.0; }
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Am[i * n + j] = V[i * n + j] * ev[j];      // V D
        real *tmp = scr + n * n + n;   // n*n more
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { real a = 0; for (int k = 0; k < n; ++k) a += Am[i * n + k] * V[j * n + k]; tmp[i * n + j] = a; }
        for (int e = 0; e < n * n; ++e) Winv[e] = tmp[e];
    }
    wsync();
}

// Inverse of a symmetric 3 x 3 W (row-major; the lower triangle is read) by cofactors, every lane for itself: iv = {00, 10, 11, 20, 21, 22}.
// Returns psd_pinv's fast-path test: leading minors > 0 and 1 / ||W^-1||_F > thr, i.e. every eigenvalue provably above the cut.
__device__ __forceinline__ bool sym3_inverse(const real *W, real thr, real iv[6])
{
    const real a = W[0], b = W[3], c = W[4], d = W[6], e = W[7], f = W[8];
    const real c00 = c * f - e * e, c10 = d * e - b * f, c20 = b * e - c * d, c11 = a * f - d * d, c21 = b * d - a * e, c22 = a * c - b * b;
    const real det = a * c00 + b * c10 + d * c20;
    const real id = 1.0 / det;
    iv[0] = c00 * id; iv[1] = c10 * id; iv[2] = c11 * id; iv[3] = c20 * id; iv[4] = c21 * id; iv[5] = c22 * id;
    const real fro = (iv[0] * iv[0] + iv[2] * iv[2] + iv[5] * iv[5]) + 2.0 * (iv[1] * iv[1] + iv[3] * iv[3] + iv[4] * iv[4]);
    return (a > 0.0) && (c22 > 0.0) && (det > 0.0) && (fro == fro) && (1.0 > thr * __builtin_sqrt(fro));
}

// LDS: what both waves share, then one workspace per wave of identical layout.  The relaxation QP of wave 0 runs over the memory of its
// finished recursion: Nq (30 x 18) = NP + T1, Sq (18 x 18) = JB + JTP + JTB.
#define QW_NP    0      // 324 null-space projector
#define QW_T1    324    // 216 (18 x dimFr, 18 x 3)
#define QW_JB    540    // 216 JcBar / pinv
#define QW_JTP   756    // 54  3 x 18
#define QW_JTB   810    // 54  18 x 3
#define QW_T2    864    // 54
#define QW_LAM   918    // 144
#define QW_LAMI  1062   // 144
#define QW_SCR   1206   // 144*2 + 16 = 304
#define QW_VEC   1510   // 3 x 18: qdd, tv, tv2 (wave 1: delta_q, qdot)
#define QW_QP    1564   // qd_ 32, qr_ 32, qu_ 32, qc0 32, qx 18, qw 18, qz 18
#define QW_SIZE  1746
#define QR_WBC_SHARED_DOUBLES 1392
#define QR_WBC_LDS_DOUBLES (QR_WBC_SHARED_DOUBLES + 2 * QW_SIZE)       // 4884 doubles = 39 072 B: four workgroups per CU

// leg ids of the contacts / swing-foot tasks, 4 bits each in `cpack` / `tpack` (no indexed local arrays -> no scratch)
#define CLEG(k) ((int)((cpack >> (4 * (k))) & 15u))
#define TLEG(k) ((int)((tpack >> (4 * (k))) & 15u))

// Output stores of a pipelined tick are written through (agent scope, sc1) so that the wave can tell the tick's join when they are in memory
// (wbc_signal_done): a plain store otherwise.
__device__ __forceinline__ void st_w(float *p, float v, bool through)
{
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ void st_w(int *p, int v, bool through)
{
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ void wbc_signal_done(int *finished, int lane)
{
    if (!finished) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have been acknowledged
    if (lane == 0) __hip_atomic_fetch_add(finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The relaxation QP of K13 and the torque store (+ K14 tail), on one wavefront.  Inputs in LDS: A (mass matrix), JC (stacked contact
// Jacobian), qdd (commanded accelerations of the recursion), Cv, Gv, cm (commands incl. Fr_des and the contact flags).  Wq: >= QW_SIZE doubles
// of workspace laid out as the QW_* offsets say (Nq over NP + T1, Sq over JB + JTP + JTB, the vectors at VEC / QP).
//
// In two parts.  wbc_qp_setup is everything that does not read Fr_des: the force-free part of gen, the constraint normals, and the 6 x 6 block
// of the floating-base equalities with its inverse.  In a pipelined tick it runs BEFORE the workgroup waits for the MPC's forces, so that only
// the right-hand sides, the changes of the working set and the store are left behind the robot's flag.  Returns true when the equalities
// are dependent (QRGPU_ST_WBC_INFEAS).  The arithmetic of every value is what it was in one piece.
__device__ __forceinline__ bool wbc_qp_setup(const int lane, const WbcConst &K, const int nc, const real *A, const real *JC, const real *qdd, const real *Cv, const real *Gv,
                                             real *W, int *sI)
{
    const int dimFr = 3 * nc;
    real *tv = W + QW_VEC + 18;
    real *Nq = W + QW_NP;          // 30 x 18 QP constraint normals
    real *Sq = W + QW_JB;          // 18 x 18 S^-1
    real *qd_ = W + QW_QP;         // 32 d
    real *qx = qd_ + 128;          // 18 z
    const int nz = 6 + dimFr, np_ = 6, mi = 6 * nc;
    // gen (tv) = A qdd + C + G - Jc^T Fr_des  (all 18 rows; rows 6.. are reused for the torque): here without the last term
    if (lane < 18) {
        real acc = Cv[lane] + Gv[lane];
        for (int k = 0; k < 18; ++k) acc += A[lane * 18 + k] * qdd[k];
        tv[lane] = acc;
    }
    // constraint normals Nq[c][0:nz]
    for (int e = lane; e < (np_ + mi) * nz; e += 64) {
        const int c = e / nz, j = e - c * nz;
        real v = 0.0;
        if (c < 6) v = (j < 6) ? A[c * 18 + j] : -JC[(j - 6) * 18 + c];
        else {
            const int u = c - 6, k = u / 6, tt = u - 6 * k;      // Uf row tt of contact k
            if (j >= 6 && (j - 6) / 3 == k) {
                const int ax = (j - 6) - 3 * k;
                // Uf rows: [0 0 1] [1 0 mu] [-1 0 mu] [0 1 mu] [0 -1 mu] [0 0 -1]   (qr_single_contact.cpp:40-62)
                if (tt == 0) v = (ax == 2) ? 1.0 : 0.0;
                else if (tt == 5) v = (ax == 2) ? -1.0 : 0.0;
                else { const int a2 = (tt - 1) >> 1; v = (ax == a2) ? (((tt - 1) & 1) ? -1.0 : 1.0) : ((ax == 2) ? (real)K.mu : 0.0); }
            }
        }
        Nq[c * 18 + j] = v;
    }
    if (lane < 18) qx[lane] = 0.0;        // g0 = 0  =>  unconstrained minimiser z = 0
    if (lane < 32) sI[32 + lane] = -1;    // constraint -> position or -1
    wsync();
    // The six floating-base equalities enter together instead of one active-set iteration each: with S_e = N_e M N_e' (6 x 6),
    // u_e = -S_e^-1 c_e, z = M N_e' u_e, S^-1 = S_e^-1, working set = {0..5}.  Same point the six equality iterations reach.
    const real iw_fb = 1.0 / (real)K.w_fb, iw_fr = 1.0 / (real)K.w_fr;
    if (lane < 36) {
        const int a = lane / 6, b2 = lane - 6 * a;
        real acc = 0.0;
        for (int j = 0; j < nz; ++j) acc += Nq[a * 18 + j] * ((j < 6) ? iw_fb : iw_fr) * Nq[b2 * 18 + j];
        Sq[a * 18 + b2] = acc;
    }
    wsync();
    real tr = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) tr += Sq[a * 18 + a];
    const real minpiv = spd_inverse<1>(lane, Sq, 18, 6, qd_);
    return !(minpiv > 1e-14 * tr);        // dependent equalities
}

__device__ __forceinline__ void wbc_qp_and_store(const int lane, const int rid, const int n, const WbcConst &K, const int nc, const unsigned cpack, const bool bad_type,
                                                 const bool eq_dependent, const real *A, const real *JC, const real *cm, real *W, int *sI,
                                                 float *__restrict__ g_tau, int *__restrict__ g_status, const int merge_tau, const int status_or, const int epilogue,
                                                 long long *__restrict__ dbgT, float *__restrict__ g_qp, const bool piped = false, float *__restrict__ g_prev = nullptr)
{
#define QW_TSF(i) do { if (dbgT && (threadIdx.x & 63) == 0 && threadIdx.x < 64) dbgT[(size_t)blockIdx.x * 16 + (i)] = clock64(); } while (0)
    int qp_iters = 0;
    const int dimFr = 3 * nc;
    real *tv = W + QW_VEC + 18;
    real *Nq = W + QW_NP;          // 30 x 18 QP constraint normals
    real *Sq = W + QW_JB;          // 18 x 18 S^-1
    real *qd_ = W + QW_QP;         // 32 d
    real *qr_ = qd_ + 32;          // 32 r
    real *qu_ = qr_ + 32;          // 32 u
    real *qc0 = qu_ + 32;          // 32 constraint offsets
    real *qx = qc0 + 32;           // 18 z
    // ---------------- relaxation QP (SetCost/SetEqualityConstraint/SetInequalityConstraint :129-167,232-247) ----------------
    //   min 1/2 z' W z,  W = diag(w_fb x6, w_fr x dimFr)
    //   equalities  i<6 :  A[i,0:6] z_fb - Jc[:,i]' z_f + gen_i = 0,   gen = (A qdd + C + G - Jc' Fr_des)[0:6]
    //   inequalities    :  Uf (Fr_des + z_f) - ineqVec >= 0
    const int nz = 6 + dimFr, np_ = 6, mi = 6 * nc;
    if (lane < 18) {
        real acc = tv[lane];
        for (int k = 0; k < dimFr; ++k) acc -= JC[k * 18 + lane] * cm[51 + 3 * CLEG(k / 3) + k % 3];
        tv[lane] = acc;
    }
    wsync();
    // offsets qc0[c]
    if (lane < np_ + mi) {
        const int c = lane;
        real v;
        if (c < 6) v = tv[c];
        else {
            const int u = c - 6, k = u / 6, tt = u - 6 * k;
            real acc = 0.0;
            for (int ax = 0; ax < 3; ++ax) acc += Nq[c * 18 + 6 + 3 * k + ax] * cm[51 + 3 * CLEG(k) + ax];     // Uf Fr_des
            v = acc - ((tt == 5) ? -(real)K.max_fz : 0.0);                                                    // - ineqVec
        }
        qc0[c] = v;
    }
    wsync();
    int stw = bad_type ? QRGPU_ST_BAD_TYPE_D : 0;
    {
        // Goldfarb-Idnani, Schur-complement form, M = W^-1 diagonal, dense normals.
        int q = 0;
        const real iw_fb = 1.0 / (real)K.w_fb, iw_fr = 1.0 / (real)K.w_fr;
        auto Minv = [&](int j) -> real { return (j < 6) ? iw_fb : iw_fr; };
        // one "add constraint c" attempt; returns 0 added, 1 dropped-one-and-retry, 2 failure/infeasible, 3 dependent
        int iter = 0;
        const int maxit = 200;
        int *iter_out = &qp_iters;
        bool fail = false;
        int *act = sI;             // active ids
        int *posi = sI + 32;       // constraint -> position or -1 (cleared by wbc_qp_setup)
        // the six floating-base equalities: S^-1 = S_e^-1 is in place (wbc_qp_setup); u_e = -S_e^-1 c_e, z = M N_e' u_e, working set = {0..5}
        int next_eq = 0;
        {
            if (eq_dependent) { stw |= QRGPU_ST_WBC_INFEAS_D; fail = true; }
            if (lane < 6) {
                real acc = 0.0;
#pragma unroll
                for (int b2 = 0; b2 < 6; ++b2) acc -= Sq[lane * 18 + b2] * qc0[b2];
                qu_[lane] = acc; act[lane] = lane; posi[lane] = lane;
            }
            wsync();
            if (lane < nz) {
                real acc = 0.0;
#pragma unroll
                for (int a = 0; a < 6; ++a) acc += Nq[a * 18 + lane] * qu_[a];
                qx[lane] = Minv(lane) * acc;
            }
            q = 6; next_eq = np_;
            wsync();
        }
        // Working-set vectors in registers: position i lives in lane i (constraint id act_r, multiplier u_r, and d, r of the current
        // change); lane c also knows whether constraint c is active.  Only S^-1, the normals and x go through LDS; uniform gathers are
        // v_readlane, reductions DPP.  (The six equalities are already in and are never dropped: every row handled here is an inequality.)
        int act_r = (lane < 6) ? lane : 0;
        real u_r = (lane < 6) ? qu_[lane] : 0.0;
        bool active_c = lane < 6;
        const real INF_ = __builtin_inf();
        while (!fail) {
            int p;
            {
                real bs = -1e-10; bool cand = false;
                if (lane >= 6 && lane < np_ + mi && !active_c) {
                    const int cj = 6 + 3 * ((lane - 6) / 6);                    // the three force unknowns of this row's contact
                    const real s_ = qc0[lane] + (Nq[lane * 18 + cj] * qx[cj] + Nq[lane * 18 + cj + 1] * qx[cj + 1] + Nq[lane * 18 + cj + 2] * qx[cj + 2]);
                    if (s_ < bs) { bs = s_; cand = true; }
                }
                const real mn = wave_min_d(bs);
                if (!(mn < -1e-10)) break;
                p = first_lane(cand && bs == mn);                               // lowest-id row holding the minimum (lane == constraint id)
                if (p < 0) break;
            }
            real up = 0.0;
            for (;;) {
                if (++iter > maxit) { stw |= QRGPU_ST_WBC_MAXITER_D; fail = true; break; }
                *iter_out = iter;
                // an inequality row touches the three force unknowns of one contact: its products are three terms, not a reduction
                const int pj = 6 + 3 * ((p - 6) / 6);
                const real pn0 = Nq[p * 18 + pj], pn1 = Nq[p * 18 + pj + 1], pn2 = Nq[p * 18 + pj + 2];
                const real delta = (pn0 * pn0 + pn1 * pn1 + pn2 * pn2) * iw_fr;
                // w = M n_p is iw_fr * n_p on those three unknowns and zero elsewhere
                const real w0 = iw_fr * pn0, w1 = iw_fr * pn1, w2 = iw_fr * pn2;
                real dq_ = 0.0;
                if (lane < q) { const real *na = Nq + act_r * 18 + pj; dq_ = na[0] * w0 + na[1] * w1 + na[2] * w2; }
                // r = S^-1 d: lane i walks row i, d_j by v_readlane
                real rq_ = 0.0;
                {
                    const real *Srow = Sq + ((lane < q) ? lane : 0) * 18;
#pragma unroll
                    for (int c = 0; c < 18; c += 6) {           // chunks of six columns, a chunk's loads in flight together
                        if (c >= q) continue;
                        real sv[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) sv[j] = Srow[c + j];                        // (row stride 18: in bounds; columns >= q are stale, masked below)
#pragma unroll
                        for (int j = 0; j < 6; ++j) rq_ += (c + j < q) ? sv[j] * readlane_d(dq_, c + j) : 0.0;
                    }
                    if (lane >= q) rq_ = 0.0;
                }
                const real dr = wave_sum_d(rq_ * dq_);
                const real zc = delta - dr;
                real t1; int lpos;
                {
                    const real tt = (lane < q && act_r >= np_ && rq_ > 0.0) ? u_r * fast_rcp(rq_) : INF_;
                    t1 = wave_min_d(tt);
                    lpos = (t1 < INF_) ? first_lane(tt == t1) : 0x7fffffff;
                }
                const real sp = qc0[p] + (pn0 * qx[pj] + pn1 * qx[pj + 1] + pn2 * qx[pj + 2]);
                const bool have_z = zc > 1e-13 * delta;
                const real izc = fast_rcp(zc);
                const real t2 = have_z ? -sp * izc : INF_;
                const real t = t1 < t2 ? t1 : t2;
                if (!(t < INF_)) { stw |= QRGPU_ST_WBC_INFEAS_D; fail = true; break; }
                if (have_z) {
                    // z = w - M N r ; x += t z
                    real acc = 0.0;
                    const int lz = (lane < nz) ? lane : 0;
#pragma unroll
                    for (int c = 0; c < 18; c += 6) {
                        if (c >= q) continue;
                        real nv[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) nv[j] = Nq[__builtin_amdgcn_readlane(act_r, c + j) * 18 + lz];     // (lanes >= q hold act_r = 0: a valid row)
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc += (c + j < q) ? nv[j] * readlane_d(rq_, c + j) : 0.0;
                    }
                    if (lane < nz) {
                        const real wl_ = (lane == pj) ? w0 : (lane == pj + 1) ? w1 : (lane == pj + 2) ? w2 : 0.0;
                        qx[lane] += t * (wl_ - Minv(lane) * acc);
                    }
                }
                u_r -= t * rq_;
                up += t;
                if (have_z && t == t2) {
                    const real isg = izc;
                    if (lane < q) qr_[lane] = rq_;
                    wsync();                                                    // r and the new x are in LDS
                    { const int rq2 = rcp16(q); for (int e = lane; e < q * q; e += 64) { const int i = fdiv16(e, rq2), j = e - i * q; Sq[i * 18 + j] += qr_[i] * qr_[j] * isg; } }
                    if (lane < q) { Sq[q * 18 + lane] = -rq_ * isg; Sq[lane * 18 + q] = -rq_ * isg; }
                    if (lane == 0) Sq[q * 18 + q] = isg;
                    if (lane == q) { act_r = p; u_r = up; }
                    if (lane == p) active_c = true;
                    ++q;
                    wsync();
                    break;
                }
                // partial or dual-only step: position lpos leaves
                {
                    const int l = lpos, last = q - 1;
                    const int cdrop = __builtin_amdgcn_readlane(act_r, l), alast = __builtin_amdgcn_readlane(act_r, last);
                    const real ulast = readlane_d(u_r, last);
                    if (lane < q) qd_[lane] = Sq[lane * 18 + l];
                    wsync();
                    const real isl = 1.0 / qd_[l];
                    { const int rq2 = rcp16(q); for (int e = lane; e < q * q; e += 64) { const int i = fdiv16(e, rq2), j = e - i * q; if (i != l && j != l) Sq[i * 18 + j] -= qd_[i] * qd_[j] * isl; } }
                    wsync();
                    if (l != last) {
                        if (lane < last) qr_[lane] = (lane == l) ? Sq[last * 18 + last] : Sq[last * 18 + lane];
                        wsync();
                        if (lane < last) { Sq[l * 18 + lane] = qr_[lane]; Sq[lane * 18 + l] = qr_[lane]; }
                        if (lane == l) { act_r = alast; u_r = ulast; }
                    }
                    if (lane == cdrop) active_c = false;
                    --q;
                    wsync();
                }
            }
        }
    }

    QW_TSF(8);
    if (dbgT && threadIdx.x == 0) dbgT[(size_t)blockIdx.x * 16 + 10] = qp_iters;
    // inspection (qrgpu_wbc_inspect_batch, instrumented build only): the QP's solution z (qpz, :113) and extraData->optimalFr = z_f + Fr_des (:216-218),
    // stance feet in contact order, zero padded
    if (g_qp) {
        float *o = g_qp + (size_t)rid * 30;
        if (lane < 18) o[lane] = (lane < nz) ? (float)qx[lane] : 0.f;
        if (lane < 12) o[18 + lane] = (lane < dimFr) ? (float)(qx[6 + lane] + cm[51 + 3 * CLEG(lane / 3) + lane % 3]) : 0.f;
    }
    // ---------------- GetSolution (:210-228) + store ----------------
    // qddot[0:6] += z[0:6];  tau = (A qddot + C + G - Jc^T (Fr_des + z_f))[6:18]
    if (lane < 12) {
        const int row = 6 + lane;
        real acc = tv[row];                                  // (A qdd_cmd + C + G - Jc^T Fr_des)[row]
        for (int k = 0; k < 6; ++k) acc += A[row * 18 + k] * qx[k];
        for (int k = 0; k < dimFr; ++k) acc -= JC[k * 18 + row] * qx[6 + k];
        const int leg = lane / 3;
        const bool stance = cm[63 + leg] != 0.0;
        if (!epilogue) {
            if (!merge_tau || stance) st_w(&g_tau[(size_t)lane * n + rid], (float)acc, piped);
        } else {
            // K14 tail (fused tick): UpdateLegCMD overwrites the stance legs (:205-219) AFTER qrFSMStateLocomotion::Run added the +-0.9 N m abad
            // compensation to every leg (QS/fsm/qr_fsm_state_locomotion.cpp:141-151), so the compensation survives on swing legs only (their
            // command is what the MPC kernel left in g_tau); then the +-23 N m clip (QS/fsm/qr_safety_checker.cpp:48-66).  legCmd.tua is a double.
            double t = stance ? (double)(float)acc
                              : (double)(piped ? __hip_atomic_load(g_tau + (size_t)lane * n + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g_tau[(size_t)lane * n + rid]);
            if (!stance && (epilogue & 1) && lane % 3 == 0) t += (double)((leg & 1) ? 0.9f : -0.9f);
            if (epilogue & 2) t = t > 23.0 ? 23.0 : (t < -23.0 ? -23.0 : t);
            st_w(&g_tau[(size_t)lane * n + rid], (float)t, piped);
        }
    }
    if (g_prev && lane < 3) st_w(&g_prev[(size_t)lane * n + rid], (float)cm[12 + lane], piped);          // desiredVel of the orientation task (quirk 4's memory)
    if (lane == 0 && g_status) {
        if (status_or & 2) stw |= QRGPU_ST_PIPE_TIMEOUT_D;
        if (status_or & 1) st_w(&g_status[rid], stw | (piped ? __hip_atomic_load(g_status + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g_status[rid]), piped);
        else st_w(&g_status[rid], stw, piped);
    }
    QW_TSF(9);
#undef QW_TSF
}

// Two wavefronts per robot, 39 KB of LDS: 4 workgroups per CU = 2 waves per SIMD, 256 VGPRs each.
//   wave 0: Jacobians, composite inertias, H, G -> A^-1                    -> prioritized acceleration recursion -> relaxation QP -> torques
//   wave 1: velocities, foot kinematics, Jcdqd, Coriolis -> the task set   -> K12 kinematic projection (when asked for) -> q_des, qd_des
// One workgroup barrier after the load, one where the two meet; everything else is wave-local (wsync).
#ifdef QR_WBC_DBG_BUILD
#define qr_wbc_kernel qr_wbc_kernel_dbg
#endif
__global__ __launch_bounds__(128, 2)
void qr_wbc_kernel(int n, const WbcConst *__restrict__ types, const int *__restrict__ type_id,
                   const float *__restrict__ g_state, const float *__restrict__ g_cmd, float *__restrict__ g_prev,
                   float *__restrict__ g_tau, float *__restrict__ g_qdes, int *__restrict__ g_status,
                   float *__restrict__ g_dbg, int merge_tau, int status_or, long long *__restrict__ dbgT,
                   const float *__restrict__ g_fr /* [12][n] Fr_des override (the MPC's forces in the fused tick) or null */,
                   int type_ready /* bit t: type t was set up */, int epilogue /* QRGPU_EPILOGUE_* bits (fused tick only) */,
                   float *__restrict__ g_qp /* [n][30] inspection: the relaxation QP's z[18] and optimalFr[12], or null */,
                   WbcPipe pipe /* pipelined tick: per-robot flags of the MPC launches running beside this one; or the list of a second pass */)
{
#ifndef QR_WBC_DBG_BUILD      // (the timed path's kernel carries neither the inspection outputs nor the cycle stamps: 2.5 % of its time; qr_wbc_kernel_dbg.hip
    dbgT = nullptr; g_dbg = nullptr; g_qp = nullptr;      //  compiles this file once more with them in, as qr_wbc_kernel_dbg, for the launches that ask for either)
#endif
#define QW_TS(i) do { if (dbgT && threadIdx.x == 0) dbgT[(size_t)blockIdx.x * 16 + (i)] = clock64(); } while (0)
#define QW_TS1(i) do { if (dbgT && threadIdx.x == 64) dbgT[(size_t)blockIdx.x * 16 + (i)] = clock64(); } while (0)
    QW_TS(0);
    int qp_iters = 0;
    int rid = xcd_robot_index((int)blockIdx.x + pipe.slot_base, n);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool gate_gave_up = pipe.gate_abort && __builtin_amdgcn_readfirstlane(*pipe.gate_abort) == (int)pipe.epoch;
    if (pipe.second) {
        // second pass of a pipelined tick: the robots the trailing MPC list launch has just re-solved (normally none) -- or, when the gate of
        // the first pass gave up, every robot (WbcPipe::gate_abort)
        if (!gate_gave_up) {
            int cnt = pipe.list_count ? *pipe.list_count : 0;
            cnt = cnt < n ? cnt : n;
            if ((int)blockIdx.x >= cnt) return;
            rid = pipe.list[blockIdx.x];
        }
    }
    if (rid < 0) return;
    if (!pipe.second && gate_gave_up) { wbc_signal_done(pipe.finished, lane); return; }
    if (QW_P_TL && threadIdx.x == 0 && !pipe.second) atomicMin(QW_P_TL + (pipe.epoch & 63u) * 8 + 3, wall_clock64());
    if (pipe.order && !pipe.second) rid = pipe.order[rid];        // (a permutation inside the XCD chunk: qr_mpc_kernel.hip, finish_order_chunk)
    if (QW_P_TLR && threadIdx.x == 0 && !pipe.second) QW_P_TLR[rid] = (int)wall_clock64();
    int tyid = type_id ? type_id[rid] : 0;
    const bool bad_type = tyid < 0 || tyid >= QR_MAX_TYPES || !((type_ready >> (tyid & (QR_MAX_TYPES - 1))) & 1);
    if (bad_type) tyid = __builtin_ctz(type_ready | (1 << QR_MAX_TYPES));     // computed with the first valid type, flagged QRGPU_ST_BAD_TYPE
    const WbcConst &K = types[tyid & (QR_MAX_TYPES - 1)];

    __shared__ real sm[QR_WBC_LDS_DOUBLES];
    real *A = sm;                  // 324  mass matrix
    real *Ai = A + 324;            // 324  A^-1
    real *JcA = Ai + 324;          // 4 x 54   foot Jacobians (3x18 each)
    real *JC = JcA + 216;          // 12 x 18  stacked contact Jacobian
    real *st = JC + 216;           // 37 state
    real *cm = st + 40;            // 67 cmd
    real *Gv = cm + 68;            // 18
    real *Cv = Gv + 18;            // 18
    real *Jcd = Cv + 18;           // 12 Jcdqd
    real *pGC = Jcd + 12;          // 12
    real *vGC = pGC + 12;          // 12
    real *tkX = vGC + 12;          // 6 tasks x 3: xddot
    real *tkE = tkX + 18;          // posErr
    real *tkV = tkE + 18;          // desiredVel
    real *legB = tkV + 18;         // 4 x 16: per-leg contributions to the base (rbi 10: wave 0, fvp 6: wave 1)
    real *sRT = legB + 64;         // 9  Rot^T (body -> world), read with run-time indices by the task Jacobians
    real *W = sm + QR_WBC_SHARED_DOUBLES + wv * QW_SIZE;     // this wave's workspace
    real *Np = W + QW_NP, *T1 = W + QW_T1, *JB = W + QW_JB, *JtP = W + QW_JTP, *JtB = W + QW_JTB, *T2 = W + QW_T2;
    real *lam = W + QW_LAM, *lamI = W + QW_LAMI, *scr = W + QW_SCR;
    real *qdd = W + QW_VEC, *tv = qdd + 18;
    real *dq1 = qdd, *dq2 = tv;                       // (wave 1's names for the same slots)
    real *Nq = W + QW_NP;          // 30 x 18 QP constraint normals
    real *Sq = W + QW_JB;          // 18 x 18 S^-1
    real *qd_ = W + QW_QP;         // 32 d
    real *qr_ = qd_ + 32;          // 32 r
    real *qu_ = qr_ + 32;          // 32 u
    real *qc0 = qu_ + 32;          // 32 constraint offsets
    real *qx = qc0 + 32;           // 18 z
    __shared__ int sI[64];         // active list of the QP
    __shared__ int sPipe;          // overlapped ticks: wave 1's wait for the robot's previous WBC pass gave up (read by wave 0 behind the meeting barrier)
    if (threadIdx.x == 0) sPipe = 0;

    // ---------------- load (wave 0: state, wave 1: commands) ----------------
    if (wv == 0) {
        if (lane < 37) st[lane] = (real)g_state[(size_t)lane * n + rid];
        for (int e = lane; e < 324; e += 64) A[e] = 0.0;
        for (int e = lane; e < 216; e += 64) JcA[e] = 0.0;
    } else if (g_tau) {
        // (pipelined tick: the MPC's forces are taken right in front of the relaxation QP, the only place that reads them, once the robot's flag is up)
        for (int i = lane; i < 67; i += 64) cm[i] = (real)((g_fr && i >= 51 && i < 63) ? (pipe.flag ? 0.f : g_fr[(size_t)(i - 51) * n + rid]) : g_cmd[(size_t)i * n + rid]);
    }
    __syncthreads();
    const real *quat = st, *pos = st + 4, *bv = st + 7, *qj = st + 13, *qdj = st + 25;
    const m3 Rwb = quat_to_rot_wb(quat);        // world -> body  (E of Xup[5])
    // contacts: stance feet; task list: 0 = body orientation, 1 = body position, then swing feet in leg order
    int nc = 0, nt = 2;
    unsigned cpack = 0, tpack = 0;      // leg ids of the contacts / swing-foot tasks, 4 bits each (no indexed local arrays -> no scratch)
    if (g_tau) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {   // readfirstlane: LDS loads count as divergent, the contact pattern is wave-uniform
            const int in_contact = __builtin_amdgcn_readfirstlane(cm[63 + l] != 0.0 ? 1 : 0);
            if (in_contact) { cpack |= (unsigned)l << (4 * nc); ++nc; } else { tpack |= (unsigned)l << (4 * (nt - 2)); ++nt; }
        }
    }
    const int dimFr = 3 * nc;

    // sines and cosines of the twelve joint angles (and, on wave 1, of the commanded roll / pitch / yaw) in one go on fifteen lanes, instead of
    // three (six) calls one after the other inside the per-leg chains; each wave keeps its own copy (scr is idle until A^-1)
    {
        real *sS = scr + 160, *sC = scr + 176;
        if (lane < 15) {
            const real th = (lane < 12) ? qj[lane] : ((wv == 1 && g_tau) ? cm[9 + lane - 12] : 0.0);
            real s_, c_;
            sincos(th, &s_, &c_);
            sS[lane] = s_; sC[lane] = c_;
        }
        wsync();
    }
    const real *sS = scr + 160, *sC = scr + 176;

    QW_TS(1);
    // ---------------- K8-K10 per leg (lanes 0-3 of both waves) ----------------
    // Wave 0 takes what the mass matrix needs (contact Jacobians, composite inertias, H, gravity), wave 1 what depends on the velocities
    // (bias accelerations, foot position / velocity, Jcdqd, Coriolis) and then the task set: two chains of half the length side by side.
    struct LegFrames { v3 r_a, r_h, r_k, loc; m3 Ea, Eh, Ek, Eabs_a, Eabs_h, Eabs_k; };
    auto leg_frames = [&](int leg) {
        LegFrames F;
        const int side = leg & 1;           // side 0: right (legs 0,2; sideSign<0), 1: left
        const real sx = (leg < 2) ? 1.0 : -1.0, sy = side ? 1.0 : -1.0;
        F.r_a = mk(sx * K.abad_loc[0], sy * K.abad_loc[1], K.abad_loc[2]);
        F.r_h = mk(0.0, sy * K.hip_l, 0.0);
        F.r_k = mk(0.0, 0.0, -K.upper_l);
        F.loc = mk(0.0, side ? -K.foot_y : K.foot_y, -K.lower_l);
        F.Ea = coord_rot_sc(0, sS[3 * leg], sC[3 * leg]); F.Eh = coord_rot_sc(1, sS[3 * leg + 1], sC[3 * leg + 1]); F.Ek = coord_rot_sc(1, sS[3 * leg + 2], sC[3 * leg + 2]);
        // absolute rotations (world -> link)
        F.Eabs_a = mul(F.Ea, Rwb); F.Eabs_h = mul(F.Eh, F.Eabs_a); F.Eabs_k = mul(F.Ek, F.Eabs_h);
        return F;
    };
    const v3 ex = mk(1, 0, 0), ey = mk(0, 1, 0);
    if (wv == 0) {
      if (lane < 4) {
        const int leg = lane, side = leg & 1;
        const LegFrames F = leg_frames(leg);
        const v3 r_a = F.r_a, r_h = F.r_h, r_k = F.r_k, loc = F.loc;
        const m3 &Ea = F.Ea, &Eh = F.Eh, &Ek = F.Ek, &Eabs_a = F.Eabs_a, &Eabs_h = F.Eabs_h, &Eabs_k = F.Eabs_k;
        // contact Jacobian columns: world velocity of the foot per unit generalized velocity
        {
            real *J = JcA + 54 * leg;
            const v3 lk = loc;                               // foot in knee frame
            const v3 lh = r_k + mulT(Ek, lk);                // foot in hip frame
            const v3 la = r_h + mulT(Eh, lh);                // foot in abad frame
            const v3 lb = r_a + mulT(Ea, la);                // foot in base frame
            const v3 ck_ = mulT(Eabs_k, cross(ey, lk)), ch_ = mulT(Eabs_h, cross(ey, lh)), ca_ = mulT(Eabs_a, cross(ex, la));
            const int c0 = 6 + 3 * leg;
            J[0 * 18 + c0] = ca_.x; J[1 * 18 + c0] = ca_.y; J[2 * 18 + c0] = ca_.z;
            J[0 * 18 + c0 + 1] = ch_.x; J[1 * 18 + c0 + 1] = ch_.y; J[2 * 18 + c0 + 1] = ch_.z;
            J[0 * 18 + c0 + 2] = ck_.x; J[1 * 18 + c0 + 2] = ck_.y; J[2 * 18 + c0 + 2] = ck_.z;
            // base: angular columns Rbw (e_i x lb), linear columns Rbw e_i
            const v3 e[3] = {mk(1, 0, 0), mk(0, 1, 0), mk(0, 0, 1)};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const v3 ang = mulT(Rwb, cross(e[i], lb)), lin = mulT(Rwb, e[i]);
                J[0 * 18 + i] = ang.x; J[1 * 18 + i] = ang.y; J[2 * 18 + i] = ang.z;
                J[0 * 18 + 3 + i] = lin.x; J[1 * 18 + 3 + i] = lin.y; J[2 * 18 + 3 + i] = lin.z;
            }
        }
        // composite inertias (rotor constants are folded into the *_eff parents on the host)
        const rbi ICk = rbi_load(K.rb[QR_RB_KNEE]);
        const rbi Ih_e = rbi_load(K.rb[QR_RB_HIP_EFF + side]);
        const rbi Ia_e = rbi_load(K.rb[QR_RB_ABAD_EFF + side]);
        const rbi ICh = rbi_add(Ih_e, rbi_to_parent(ICk, Ek, r_k));
        const rbi ICa = rbi_add(Ia_e, rbi_to_parent(ICh, Eh, r_h));
        const rbi ICa_b = rbi_to_parent(ICa, Ea, r_a);
        real *LB = legB + 16 * leg;
        LB[0] = ICa_b.m; LB[1] = ICa_b.h.x; LB[2] = ICa_b.h.y; LB[3] = ICa_b.h.z;
#pragma unroll
        for (int i = 0; i < 6; ++i) LB[4 + i] = ICa_b.I[i];
        // mass-matrix columns (massMatrix :774-806)
        const real kr = K.k_rot;
        const int ja = 6 + 3 * leg, jh = ja + 1, jk = ja + 2;
        auto base_col = [&](int j, sv6 f) {     // f expressed in the base frame
            const real fv[6] = {f.a.x, f.a.y, f.a.z, f.l.x, f.l.y, f.l.z};
#pragma unroll
            for (int i = 0; i < 6; ++i) { A[i * 18 + j] = fv[i]; A[j * 18 + i] = fv[i]; }
        };
        {   // knee
            sv6 S; S.a = ey; S.l = mk(0, 0, 0);
            sv6 f = rbi_mul(ICk, S);
            A[jk * 18 + jk] = f.a.y + kr;
            f = xforceT(Ek, r_k, f); f.a.y += kr;                 // + Xuprot^T (Irot Srot): knee rotor, E_rot = 1
            A[jh * 18 + jk] = A[jk * 18 + jh] = f.a.y;
            f = xforceT(Eh, r_h, f);
            A[ja * 18 + jk] = A[jk * 18 + ja] = f.a.x;
            f = xforceT(Ea, r_a, f);
            base_col(jk, f);
        }
        {   // hip
            sv6 S; S.a = ey; S.l = mk(0, 0, 0);
            sv6 f = rbi_mul(ICh, S);
            A[jh * 18 + jh] = f.a.y + kr;
            f = xforceT(Eh, r_h, f); f.a.x += kr * K.hiprot_ex; f.a.y += kr * K.hiprot_ey;   // hip rotor: E_rot = Rz(pi)
            A[ja * 18 + jh] = A[jh * 18 + ja] = f.a.x;
            f = xforceT(Ea, r_a, f);
            base_col(jh, f);
        }
        {   // abad
            sv6 S; S.a = ex; S.l = mk(0, 0, 0);
            sv6 f = rbi_mul(ICa, S);
            A[ja * 18 + ja] = f.a.x + kr;
            f = xforceT(Ea, r_a, f); f.a.x += kr;
            base_col(ja, f);
        }
        // gravity (:607-626): ag_i = [0; E_abs_i g], G[i] = -S_i . (IC_i ag_i) = -axis . (h_i x a_i)
        {
            const v3 gw = mk(0, 0, -9.81);
            const v3 g_a = mul(Eabs_a, gw), g_h = mul(Eabs_h, gw), g_k = mul(Eabs_k, gw);
            Gv[ja] = -cross(ICa.h, g_a).x;
            Gv[jh] = -cross(ICh.h, g_h).y;
            Gv[jk] = -cross(ICk.h, g_k).y;
        }
      }
      wsync();
      QW_TS(2);
      // ---------------- base block of H and G (lane 0) ----------------
      if (lane == 0) {
        rbi IC5 = rbi_load(K.rb[QR_RB_BASE_EFF]);
        for (int l = 0; l < 4; ++l) IC5 = rbi_add(IC5, rbi_load(legB + 16 * l));
        // H[0:6,0:6] = IC5 as a 6x6
        const real I6[3][3] = {{IC5.I[0], IC5.I[3], IC5.I[4]}, {IC5.I[3], IC5.I[1], IC5.I[5]}, {IC5.I[4], IC5.I[5], IC5.I[2]}};
        const real hx[3][3] = {{0, -IC5.h.z, IC5.h.y}, {IC5.h.z, 0, -IC5.h.x}, {-IC5.h.y, IC5.h.x, 0}};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                A[i * 18 + j] = I6[i][j];
                A[i * 18 + 3 + j] = hx[i][j];
                A[(3 + i) * 18 + j] = hx[j][i];
                A[(3 + i) * 18 + 3 + j] = (i == j) ? IC5.m : 0.0;
            }
        // G[0:6] = -IC5 [0; a5],  a5 = Rwb g
        const v3 a5 = mul(Rwb, mk(0, 0, -9.81));
        const v3 gt = cross(IC5.h, a5);
        Gv[0] = -gt.x; Gv[1] = -gt.y; Gv[2] = -gt.z; Gv[3] = -IC5.m * a5.x; Gv[4] = -IC5.m * a5.y; Gv[5] = -IC5.m * a5.z;
      }
      wsync();
      QW_TS(3);
    } else {
      if (lane < 4) {
        const int leg = lane, side = leg & 1;
        const LegFrames F = leg_frames(leg);
        const v3 r_a = F.r_a, r_h = F.r_h, r_k = F.r_k, loc = F.loc;
        const m3 &Ea = F.Ea, &Eh = F.Eh, &Ek = F.Ek, &Eabs_a = F.Eabs_a, &Eabs_h = F.Eabs_h, &Eabs_k = F.Eabs_k;
        const real d0 = qdj[3 * leg], d1 = qdj[3 * leg + 1], d2 = qdj[3 * leg + 2];
        // velocities, bias accelerations
        sv6 v5; v5.a = mk(bv[0], bv[1], bv[2]); v5.l = mk(bv[3], bv[4], bv[5]);
        sv6 va = xmotion(Ea, r_a, v5); sv6 vJa; vJa.a = d0 * ex; vJa.l = mk(0, 0, 0); va.a = va.a + vJa.a;
        sv6 ca = crm(va, vJa);
        sv6 vh = xmotion(Eh, r_h, va); sv6 vJh; vJh.a = d1 * ey; vJh.l = mk(0, 0, 0); vh.a = vh.a + vJh.a;
        sv6 ch = crm(vh, vJh);
        sv6 vk = xmotion(Ek, r_k, vh); sv6 vJk; vJk.a = d2 * ey; vJk.l = mk(0, 0, 0); vk.a = vk.a + vJk.a;
        sv6 ck = crm(vk, vJk);
        sv6 aa = ca;
        sv6 ah = xmotion(Eh, r_h, aa); ah.a = ah.a + ch.a; ah.l = ah.l + ch.l;
        sv6 ak = xmotion(Ek, r_k, ah); ak.a = ak.a + ck.a; ak.l = ak.l + ck.l;
        // Foot position / velocity exactly as forwardKinematics does it (:506-521): through the bottom-left
        // block of Xa and invertSXform / sXFormPoint, which use E^T as E^-1.  With the float-rounded (not
        // exactly unit) quaternion of the state this differs from the textbook sum of offsets by O(|q|^2-1) * 1 m,
        // which the foot task's Kp = 500 would turn into 1e-5 N m.
        {
            auto skewm = [](v3 r) { m3 S = {{{0, -r.z, r.y}, {r.z, 0, -r.x}, {-r.y, r.x, 0}}}; return S; };
            auto neg = [](const m3 &A_) { m3 C_; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C_.m[i][j] = -A_.m[i][j]; return C_; };
            auto addm = [](const m3 &A_, const m3 &B_) { m3 C_; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C_.m[i][j] = A_.m[i][j] + B_.m[i][j]; return C_; };
            auto unskew = [](const m3 &M_) { return mk(0.5 * (M_.m[2][1] - M_.m[1][2]), 0.5 * (M_.m[0][2] - M_.m[2][0]), 0.5 * (M_.m[1][0] - M_.m[0][1])); };   // matToSkewVec
            const v3 p5 = mk(pos[0], pos[1], pos[2]);
            const m3 B5 = neg(mul(Rwb, skewm(p5)));                                              // createSXform(R, pos) bottom-left
            const m3 Ba = addm(mul(neg(mul(Ea, skewm(r_a))), Rwb), mul(Ea, B5));                 // Xup[a] * Xa[5]
            const m3 Bh = addm(mul(neg(mul(Eh, skewm(r_h))), Eabs_a), mul(Eh, Ba));
            const m3 Bk = addm(mul(neg(mul(Ek, skewm(r_k))), Eabs_h), mul(Ek, Bh));
            const m3 E = Eabs_k, Et = transpose(Eabs_k);
            const v3 r1 = (-1.0) * unskew(mul(Et, Bk));                                          // invertSXform: r
            const v3 Er1 = mul(E, r1);
            const m3 BLi = mul(Et, skewm(Er1));                                                  // Xai bottom-left = -E^T [-E r]x
            const v3 rp = (-1.0) * unskew(mul(E, BLi));                                          // translationFromSXform(Xai)
            const v3 pf = mul(Et, loc - rp);                                                     // sXFormPoint
            const v3 wS = mul(Et, vk.a);
            const v3 vS = mul(BLi, vk.a) + mul(Et, vk.l);
            const v3 vf = vS + cross(wS, pf);                                                    // spatialToLinearVelocity
            pGC[3 * leg] = pf.x; pGC[3 * leg + 1] = pf.y; pGC[3 * leg + 2] = pf.z;
            vGC[3 * leg] = vf.x; vGC[3 * leg + 1] = vf.y; vGC[3 * leg + 2] = vf.z;
        }
        // Jcdqd = Rai [ (a_lin + a_ang x loc) + w x (v_lin + w x loc) ]
        {
            const v3 t = (ak.l + cross(ak.a, loc)) + cross(vk.a, vk.l + cross(vk.a, loc));
            const v3 jd = mulT(Eabs_k, t);
            Jcd[3 * leg] = jd.x; Jcd[3 * leg + 1] = jd.y; Jcd[3 * leg + 2] = jd.z;
        }
        // Coriolis (:633-665) with link inertias
        {
            const rbi Ik = rbi_load(K.rb[QR_RB_KNEE]), Ih = rbi_load(K.rb[QR_RB_HIP + side]), Ia = rbi_load(K.rb[QR_RB_ABAD + side]);
            const int ja = 6 + 3 * leg, jh = ja + 1, jk = ja + 2;
            sv6 fk = rbi_mul(Ik, ak); { sv6 c = crf(vk, rbi_mul(Ik, vk)); fk.a = fk.a + c.a; fk.l = fk.l + c.l; }
            sv6 fh = rbi_mul(Ih, ah); { sv6 c = crf(vh, rbi_mul(Ih, vh)); fh.a = fh.a + c.a; fh.l = fh.l + c.l; }
            sv6 fa = rbi_mul(Ia, aa); { sv6 c = crf(va, rbi_mul(Ia, va)); fa.a = fa.a + c.a; fa.l = fa.l + c.l; }
            Cv[jk] = fk.a.y;
            { sv6 t = xforceT(Ek, r_k, fk); fh.a = fh.a + t.a; fh.l = fh.l + t.l; }
            Cv[jh] = fh.a.y;
            { sv6 t = xforceT(Eh, r_h, fh); fa.a = fa.a + t.a; fa.l = fa.l + t.l; }
            Cv[ja] = fa.a.x;
            sv6 t = xforceT(Ea, r_a, fa);
            real *LB = legB + 16 * leg;
            LB[10] = t.a.x; LB[11] = t.a.y; LB[12] = t.a.z; LB[13] = t.l.x; LB[14] = t.l.y; LB[15] = t.l.z;
        }
      }
      wsync();
    }
    if (wv == 0) {
    // ---------------- K13 GetModelRes: A^-1 ----------------
    // The joints of different legs do not couple (H(leg a, leg b) = 0), so A = [Abb Abl; Abl' diag(L0..L3)] is inverted through the
    // 6 x 6 Schur complement of the floating base instead of 18 pivots: Li = L^-1 (cofactors), T = Abl Li, Sb = Abb - T Abl',
    // A^-1 = [Sb^-1, -Sb^-1 T; -T' Sb^-1, Li + T' Sb^-1 T].
    {
        real *Tm = scr, *Um = scr + 72;
        if (lane < 36) {
            const int l = lane / 9, e = lane - 9 * l, i = e / 3, j = e - 3 * i;
            const real *B = A + (6 + 3 * l) * 18 + 6 + 3 * l;
            const real b00 = B[0], b01 = B[1], b02 = B[2], b11 = B[19], b12 = B[20], b22 = B[38];
            const real c00 = b11 * b22 - b12 * b12, c01 = b02 * b12 - b01 * b22, c02 = b01 * b12 - b02 * b11;
            const real c11 = b00 * b22 - b02 * b02, c12 = b01 * b02 - b00 * b12, c22 = b00 * b11 - b01 * b01;
            const real idet = 1.0 / (b00 * c00 + b01 * c01 + b02 * c02);
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const real c = (lo == 0) ? (hi == 0 ? c00 : (hi == 1 ? c01 : c02)) : (lo == 1 ? (hi == 1 ? c11 : c12) : c22);
            Um[lane] = c * idet;                                    // Li, leg-major 3 x 3 blocks (parked in Um until T is formed)
        }
        wsync();
        for (int e = lane; e < 72; e += 64) {
            const int a = e / 12, c = e - 12 * a, l = c / 3, cc = c - 3 * l;
            const real *Ar = A + a * 18 + 6 + 3 * l, *Li = Um + 9 * l;
            Tm[e] = Ar[0] * Li[cc] + Ar[1] * Li[3 + cc] + Ar[2] * Li[6 + cc];
        }
        wsync();
        for (int e = lane; e < 144; e += 64) {                      // leg-leg part starts as blockdiag(Li); everything else of Ai is written below
            const int c = e / 12, d = e - 12 * c;
            Ai[(6 + c) * 18 + 6 + d] = (c / 3 == d / 3) ? Um[9 * (c / 3) + 3 * (c % 3) + d % 3] : 0.0;
        }
        if (lane < 36) {
            const int a = lane / 6, b2 = lane - 6 * a;
            real acc = A[a * 18 + b2];
#pragma unroll
            for (int c = 0; c < 12; ++c) acc -= Tm[a * 12 + c] * A[b2 * 18 + 6 + c];
            Ai[a * 18 + b2] = acc;
        }
        wsync();
        spd_inverse<1>(lane, Ai, 18, 6, tv);
        for (int e = lane; e < 72; e += 64) {
            const int a = e / 12, c = e - 12 * a;
            real acc = 0.0;
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) acc += Ai[a * 18 + b2] * Tm[b2 * 12 + c];
            Um[e] = acc;                                            // U = Sb^-1 T
            Ai[a * 18 + 6 + c] = -acc; Ai[(6 + c) * 18 + a] = -acc;
        }
        wsync();
        for (int e = lane; e < 144; e += 64) {
            const int c = e / 12, d = e - 12 * c;
            real acc = Ai[(6 + c) * 18 + 6 + d];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc += Tm[a * 12 + c] * Um[a * 12 + d];
            Ai[(6 + c) * 18 + 6 + d] = acc;
        }
        wsync();
    }

      QW_TS(4);
      // stacked contact Jacobian; RotT[i][j] = Rwb[j][i] with static indices only (a register array indexed at run time would be demoted to scratch)
      for (int e = lane; e < 54 * nc; e += 64) { const int k = e / 54; JC[e] = JcA[54 * CLEG(k) + (e - 54 * k)]; }
#pragma unroll
      for (int e = 0; e < 9; ++e) if (lane == e) sRT[e] = Rwb.m[e % 3][e / 3];
    } else {
      // ---------------- base block of C (lane 0): fvp5 + the legs' contributions; fvp5 = v5 x* (I5 v5)  (avp5 = 0) ----------------
      if (lane == 0) {
        sv6 fb; fb.a = mk(0, 0, 0); fb.l = mk(0, 0, 0);
        for (int l = 0; l < 4; ++l) { const real *LB = legB + 16 * l; fb.a = fb.a + mk(LB[10], LB[11], LB[12]); fb.l = fb.l + mk(LB[13], LB[14], LB[15]); }
        const rbi I5 = rbi_load(K.rb[QR_RB_BASE]);
        sv6 v5; v5.a = mk(bv[0], bv[1], bv[2]); v5.l = mk(bv[3], bv[4], bv[5]);
        const sv6 c5 = crf(v5, rbi_mul(I5, v5));
        Cv[0] = c5.a.x + fb.a.x; Cv[1] = c5.a.y + fb.a.y; Cv[2] = c5.a.z + fb.a.z;
        Cv[3] = c5.l.x + fb.l.x; Cv[4] = c5.l.y + fb.l.y; Cv[5] = c5.l.z + fb.l.z;
      }
      // ---------------- K11 tasks (lane 0) ----------------
      if (lane == 0 && g_tau) {
        const m3 RotT = transpose(Rwb);
        // --- orientation task (qr_task_body_orientation.cpp:43-81)
        {
            // quatDes = rpyToQuat(pBody_RPY_des): rotationMatrixToQuaternion(rpyToRotMat(rpy))
            const m3 Rr = mul(mul(coord_rot_sc(0, sS[12], sC[12]), coord_rot_sc(1, sS[13], sC[13])), coord_rot_sc(2, sS[14], sC[14]));
            const m3 r = transpose(Rr);
            real qd4[4];
            const real tr = r.m[0][0] + r.m[1][1] + r.m[2][2];
            if (tr > 0.0) {
                const real S = __builtin_sqrt(tr + 1.0) * 2.0;
                qd4[0] = 0.25 * S; qd4[1] = (r.m[2][1] - r.m[1][2]) / S; qd4[2] = (r.m[0][2] - r.m[2][0]) / S; qd4[3] = (r.m[1][0] - r.m[0][1]) / S;
            } else if ((r.m[0][0] > r.m[1][1]) && (r.m[0][0] > r.m[2][2])) {
                const real S = __builtin_sqrt(1.0 + r.m[0][0] - r.m[1][1] - r.m[2][2]) * 2.0;
                qd4[0] = (r.m[2][1] - r.m[1][2]) / S; qd4[1] = 0.25 * S; qd4[2] = (r.m[0][1] + r.m[1][0]) / S; qd4[3] = (r.m[0][2] + r.m[2][0]) / S;
            } else if (r.m[1][1] > r.m[2][2]) {
                const real S = __builtin_sqrt(1.0 + r.m[1][1] - r.m[0][0] - r.m[2][2]) * 2.0;
                qd4[0] = (r.m[0][2] - r.m[2][0]) / S; qd4[1] = (r.m[0][1] + r.m[1][0]) / S; qd4[2] = 0.25 * S; qd4[3] = (r.m[1][2] + r.m[2][1]) / S;
            } else {
                const real S = __builtin_sqrt(1.0 + r.m[2][2] - r.m[0][0] - r.m[1][1]) * 2.0;
                qd4[0] = (r.m[1][0] - r.m[0][1]) / S; qd4[1] = (r.m[0][2] + r.m[2][0]) / S; qd4[2] = (r.m[1][2] + r.m[2][1]) / S; qd4[3] = 0.25 * S;
            }
            // ori_err = quatProduct(ori_cmd, conj(link_ori))
            const real r1 = qd4[0], r2 = quat[0];
            const v3 v1 = mk(qd4[1], qd4[2], qd4[3]), v2 = mk(-quat[1], -quat[2], -quat[3]);
            real e0 = r1 * r2 - dot(v1, v2);
            v3 ev = r1 * v2 + r2 * v1 + cross(v1, v2);
            if (e0 < 0.0) { e0 = -e0; ev = (-1.0) * ev; }
            // quaternionToso3 (qr_se3.h:383-397)
            v3 so3 = ev;
            const real theta = 2.0 * asin(__builtin_sqrt(dot(so3, so3)));
            if (fabs(theta) < 0.0000001) so3 = mk(0, 0, 0);
            else { const real sn = sin(theta / 2.0); so3 = (1.0 / sn) * so3; so3 = theta * so3; }
            // vel_err = Rot^T (desiredVel_prev - omega_body)   (quirk 4)
            // (overlapped ticks: the robot's last WBC pass -- last tick's, possibly that tick's second pass on another stream -- leaves its epoch in
            //  wbc_done[robot] behind its written-through g_prev: poll for it, bounded and flagged, and read g_prev with agent-scope loads)
            if (pipe.wait_epoch) {
                const long long t0 = wall_clock64();
                int late = 0;
                // (... or for THIS tick's epoch: the robot is on the MPC's list pass and the second WBC pass -- another stream -- has already computed it;
                //  this workgroup, of the first pass, will find the robot's flag saying so and leave without storing anything)
                for (;;) {
                    const unsigned w = __hip_atomic_load(pipe.wbc_done + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (qr_epoch_reached(w, pipe.wait_epoch)) break;
                    if (wall_clock64() - t0 > pipe.wait_ticks) { late = 1; break; }
                    __builtin_amdgcn_s_sleep(32);
                }
                if (lane == 0) sPipe = late;
            }
            auto prev_ld = [&](int i_) -> real {
                const float *p_ = g_prev + (size_t)i_ * n + rid;
                return (real)(pipe.wait_epoch ? __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p_);
            };
            const v3 pv = mk(prev_ld(0), prev_ld(1), prev_ld(2));
            const v3 ve = mul(RotT, pv - mk(bv[0], bv[1], bv[2]));
            const real so[3] = {so3.x, so3.y, so3.z}, vev[3] = {ve.x, ve.y, ve.z};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                tkE[i] = so[i];
                tkV[i] = cm[12 + i];
                real x = K.kp_ori * so[i] + K.kd_ori * vev[i] + 0.0;
                tkX[i] = fmin(fmax(x, -10.0), 10.0);
            }
            // (this call's desiredVel becomes the next call's "previous" one, g_prev: stored by wave 0 at the very end -- in a pipelined tick a
            //  robot that turns out to be on the MPC's list pass is computed again by the second WBC pass and must find g_prev as it was)
        }
        // --- position task (qr_task_body_position.cpp:43-67)
        {
            const v3 vw = mul(RotT, mk(bv[3], bv[4], bv[5]));
            const real vwv[3] = {vw.x, vw.y, vw.z};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const real pe = cm[i] - pos[i];
                tkE[3 + i] = pe;
                tkV[3 + i] = cm[3 + i];
                real x = K.kp_pos * pe + K.kd_pos * (cm[3 + i] - vwv[i]) + cm[6 + i];
                tkX[3 + i] = fmin(fmax(x, -10.0), 10.0);
            }
        }
        // --- swing-foot tasks (qr_task_link_position.cpp:45-68), no clamp
        for (int t = 2; t < nt; ++t) {
            const int l = TLEG(t - 2);
            for (int i = 0; i < 3; ++i) {
                const real pe = cm[15 + 3 * l + i] - pGC[3 * l + i];
                tkE[3 * t + i] = pe;
                tkV[3 * t + i] = cm[27 + 3 * l + i];
                tkX[3 * t + i] = K.kp_foot * pe + K.kd_foot * (cm[27 + 3 * l + i] - vGC[3 * l + i]) + cm[39 + 3 * l + i];
            }
        }
    }
      QW_TS1(11);
    }
    __syncthreads();                  // the two waves meet: dynamics and A^-1 from wave 0, Jcdqd / C / the task set from wave 1
    if (g_dbg) {
        if (wv == 0) {
            float *o = g_dbg + (size_t)rid * (324 + 18 + 18 + 216 + 36);
            for (int e = lane; e < 324; e += 64) o[e] = (float)A[e];
            if (lane < 18) { o[324 + lane] = (float)Gv[lane]; o[342 + lane] = (float)Cv[lane]; }
            for (int e = lane; e < 216; e += 64) o[360 + e] = (float)JcA[e];
            if (lane < 12) { o[576 + lane] = (float)Jcd[lane]; o[588 + lane] = (float)pGC[lane]; o[600 + lane] = (float)vGC[lane]; }
        }
        if (!g_tau) return;
    }
    // A task Jacobian is non-zero in three columns only: orientation = base angular columns 0-2 (Rot^T), position = 3-5 (Rot^T), swing foot
    // = that leg's joints 6+3l.. (virtualDepend = false zeroes its base columns).  J3[i * ld + k] is entry (i, c0 + k).
#define TASK_COLS(t, c0, J3, ld)                                                                           \
    const int c0 = ((t) == 0) ? 0 : ((t) == 1) ? 3 : 6 + 3 * TLEG((t) - 2);                                \
    const real *J3 = ((t) < 2) ? sRT : JcA + 54 * TLEG((t) - 2) + c0;                                      \
    const int ld = ((t) < 2) ? 3 : 18
    // JtPre = Jt N_pre (3 x 18): three terms per entry
    auto jt_npre = [&](const real *J3, int ld, int c0) {
        if (lane < 54) {
            const int i = lane / 18, j = lane - 18 * i;
            const real *nr = Np + c0 * 18 + j, *jr = J3 + i * ld;
            JtP[lane] = (jr[0] * nr[0] + jr[1] * nr[18]) + jr[2] * nr[36];
        }
        wsync();
    };
    // N_pre <- N_pre - T JtPre  (T = N_pre pinv, 18 x 3): the rank-3 form of N_pre (I - pinv JtPre)
    auto npre_update = [&](const real *T) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = lane + 64 * u;
            if (e < 324) {
                const int i = fdiv16(e, rcp16(18)), j = e - 18 * i;
                Np[e] -= (T[3 * i] * JtP[j] + T[3 * i + 1] * JtP[18 + j]) + T[3 * i + 2] * JtP[36 + j];
            }
        }
        wsync();
    };
    // lamI (registers, every lane) <- pinv of the 3 x 3 in `lam` with eigenvalue cut thr
    auto pinv3 = [&](real thr, real iv[6]) {
        if (!sym3_inverse(lam, thr, iv)) {          // wave-uniform (every lane read the same nine numbers); rare: kinematic singularities
            psd_pinv(lane, lam, 3, thr, lamI, scr);
            iv[0] = lamI[0]; iv[1] = lamI[3]; iv[2] = lamI[4]; iv[3] = lamI[6]; iv[4] = lamI[7]; iv[5] = lamI[8];
        }
    };

    // The contact level of either recursion, for D = dimFr rows known at compile time (static trip counts, no predicated loads):
    //   dynamic (wave 0, K13):   JcBar = A^-1 Jc' (Jc A^-1 Jc')^+,   qdd = JcBar (-Jcdqd),   N = I - JcBar Jc
    //   kinematic (wave 1, K12): pinv  = Jc' (Jc Jc')^+,                                      N = I - pinv Jc
    auto contact_part = [&](auto Dc, const bool dynamic, const real thr) {
        constexpr int D = decltype(Dc)::value;
        constexpr int UM = (D * D + 63) / 64;
        if (dynamic) {
            for (int e = lane; e < 18 * D; e += 64) { const int i = e / D, j = e - D * i; T1[e] = dot18(Ai + i * 18, 1, JC + j * 18, 1, 18); }      // temp = Ainv Jc' (18 x D)
            wsync();
        }
        for (int e = lane; e < D * D; e += 64) {                                                    // lambda^-1 = Jc temp  /  Jc Jc'
            const int i = e / D, j = e - D * i;
            lam[e] = dynamic ? dot18(JC + i * 18, 1, T1 + j, D, 18) : dot18(JC + i * 18, 1, JC + j * 18, 1, 18);
        }
        wsync();
        psd_pinv<UM>(lane, lam, D, thr, lamI, scr);
        for (int e = lane; e < 18 * D; e += 64) {                                                   // JcBar = temp lambda  /  pinv = Jc' W^+
            const int i = e / D, j = e - D * i;
            real av[D], bv[D];
#pragma unroll
            for (int k = 0; k < D; ++k) { av[k] = dynamic ? T1[i * D + k] : JC[k * 18 + i]; bv[k] = lamI[k * D + j]; }
            real a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; k += 3) { a0 += av[k] * bv[k]; a1 += av[k + 1] * bv[k + 1]; a2 += av[k + 2] * bv[k + 2]; }
            JB[e] = (a0 + a1) + a2;
        }
        wsync();
        if (dynamic && lane < 18) {
            real acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc -= JB[lane * D + k] * Jcd[3 * CLEG(k / 3) + k % 3];
            qdd[lane] = acc;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = lane + 64 * u;
            if (e < 324) {
                const int i = fdiv16(e, rcp16(18)), j = e - 18 * i;
                real av[D], bv[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { av[k] = JB[i * D + k]; bv[k] = JC[k * 18 + j]; }
                real a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; k += 3) { a0 += av[k] * bv[k]; a1 += av[k + 1] * bv[k + 1]; a2 += av[k + 2] * bv[k + 2]; }
                Np[e] = ((i == j) ? 1.0 : 0.0) - ((a0 + a1) + a2);
            }
        }
        wsync();
    };

    QW_TS(5);
    // ---------------- K12 kinematic multitask projection (wave 1; only when its outputs are requested) ----------------
    if (wv == 1) {
      if (g_qdes) {
        const real thr2 = 1e-6;       // singular value > 1e-3  <=>  eigenvalue of J J^T > 1e-6
        // Nc = I - pinv(Jc) Jc
        switch (nc) {
            case 1: contact_part(std::integral_constant<int, 3>{}, false, thr2); break;
            case 2: contact_part(std::integral_constant<int, 6>{}, false, thr2); break;
            case 3: contact_part(std::integral_constant<int, 9>{}, false, thr2); break;
            case 4: contact_part(std::integral_constant<int, 12>{}, false, thr2); break;
            default:
                for (int e = lane; e < 324; e += 64) Np[e] = ((e / 18) == (e % 18)) ? 1.0 : 0.0;
                wsync();
        }
        for (int t = 0; t < nt; ++t) {
            TASK_COLS(t, c0, J3, ld);
            jt_npre(J3, ld, c0);
            if (lane < 9) { const int a = lane / 3, b2 = lane - 3 * a; lam[lane] = dot18(JtP + a * 18, 1, JtP + b2 * 18, 1, 18); }
            wsync();
            real iv[6];
            pinv3(thr2, iv);
            if (lane < 54) {                                                                  // pinv(JtPre) = JtPre^T W^+ (18 x 3)
                const int r = lane / 3, j = lane - 3 * r;
                const real i0 = j == 0 ? iv[0] : j == 1 ? iv[1] : iv[3], i1 = j == 0 ? iv[1] : j == 1 ? iv[2] : iv[4], i2 = j == 0 ? iv[3] : j == 1 ? iv[4] : iv[5];
                JtB[lane] = (JtP[r] * i0 + JtP[18 + r] * i1) + JtP[36 + r] * i2;
            }
            wsync();
            // delta_q = prev + pinv (posErr - Jt prev),  qdot likewise (lanes 0-17 / 18-35); T1 = N_pre pinv for the projector update
            real upd = 0.0;
            if (lane < 36) {
                const int which = lane / 18, i = lane - 18 * which;
                const real *prevv = which ? dq2 : dq1;
                const real *tgt = which ? (tkV + 3 * t) : (tkE + 3 * t);
                real tv3[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    tv3[k] = tgt[k];
                    if (t > 0) tv3[k] -= (J3[k * ld] * prevv[c0] + J3[k * ld + 1] * prevv[c0 + 1]) + J3[k * ld + 2] * prevv[c0 + 2];
                }
                upd = ((t > 0) ? prevv[i] : 0.0) + ((JtB[3 * i] * tv3[0] + JtB[3 * i + 1] * tv3[1]) + JtB[3 * i + 2] * tv3[2]);
            }
            if (t < nt - 1 && lane < 54) { const int r = lane / 3, j = lane - 3 * r; T1[lane] = dot18(Np + r * 18, 1, JtB + j, 3, 18); }
            wsync();
            if (lane < 36) { if (lane < 18) dq1[lane] = upd; else dq2[lane - 18] = upd; }
            if (t < nt - 1) npre_update(T1); else wsync();
        }
        if (lane < 12) {
            st_w(&g_qdes[(size_t)lane * n + rid], (float)(qj[lane] + dq1[6 + lane]), pipe.flag != nullptr);
            st_w(&g_qdes[(size_t)(12 + lane) * n + rid], (float)dq2[6 + lane], pipe.flag != nullptr);
        }
      }
      QW_TS1(13);
      wbc_signal_done(pipe.finished, lane);
      return;
    }

    // ---------------- K13 MakeTorque: prioritized acceleration recursion ----------------
    const real thrW = 1e-4;       // WeightedInverse default threshold (qr_wholebody_impulse_ctrl.hpp:110)
    switch (nc) {
        case 1: contact_part(std::integral_constant<int, 3>{}, true, thrW); break;
        case 2: contact_part(std::integral_constant<int, 6>{}, true, thrW); break;
        case 3: contact_part(std::integral_constant<int, 9>{}, true, thrW); break;
        case 4: contact_part(std::integral_constant<int, 12>{}, true, thrW); break;
        default:
            if (lane < 18) qdd[lane] = 0.0;
            for (int e = lane; e < 324; e += 64) Np[e] = ((e / 18) == (e % 18)) ? 1.0 : 0.0;
            wsync();
    }
    QW_TS(12);
    for (int t = 0; t < nt; ++t) {
        if (t == 1) QW_TS(14);
        TASK_COLS(t, c0, J3, ld);
        jt_npre(J3, ld, c0);
        if (lane < 54) { const int r = lane / 3, i = lane - 3 * r; T1[lane] = dot18(Ai + r * 18, 1, JtP + i * 18, 1, 18); }      // temp = Ainv JtPre^T (18 x 3)
        wsync();
        if (lane < 9) { const int a = lane / 3, b2 = lane - 3 * a; lam[lane] = dot18(JtP + a * 18, 1, T1 + b2, 3, 18); }        // lambda^-1 = JtPre temp
        wsync();
        real iv[6];
        pinv3(thrW, iv);
        if (lane < 54) {                                                                                                      // JtBar = temp lambda
            const int r = lane / 3, j = lane - 3 * r;
            const real i0 = j == 0 ? iv[0] : j == 1 ? iv[1] : iv[3], i1 = j == 0 ? iv[1] : j == 1 ? iv[2] : iv[4], i2 = j == 0 ? iv[3] : j == 1 ? iv[4] : iv[5];
            JtB[lane] = (T1[3 * r] * i0 + T1[3 * r + 1] * i1) + T1[3 * r + 2] * i2;
        }
        wsync();
        // qdd += JtBar (xddot - JtDotQdot - Jt qdd): every lane forms the three residuals itself; T2 = Npre JtBar for the projector update
        real upd = 0.0;
        if (lane < 18) {
            real tv3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                tv3[k] = tkX[3 * t + k] - ((t >= 2) ? Jcd[c0 - 6 + k] : 0.0) - ((J3[k * ld] * qdd[c0] + J3[k * ld + 1] * qdd[c0 + 1]) + J3[k * ld + 2] * qdd[c0 + 2]);
            upd = qdd[lane] + ((JtB[3 * lane] * tv3[0] + JtB[3 * lane + 1] * tv3[1]) + JtB[3 * lane + 2] * tv3[2]);
        }
        if (t < nt - 1 && lane < 54) { const int r = lane / 3, j = lane - 3 * r; T2[lane] = dot18(Np + r * 18, 1, JtB + j, 3, 18); }
        wsync();
        if (lane < 18) qdd[lane] = upd;
        if (t < nt - 1) npre_update(T2); else wsync();
    }

    QW_TS(7);
    const bool eq_dependent = wbc_qp_setup(lane, K, nc, A, JC, qdd, Cv, Gv, W, sI);
    int pipe_st = 0;
    if (pipe.flag) {
        // Everything up to here needed the robot's state and commands only; the relaxation QP needs the MPC's first-step forces.  The robot's
        // solve raises done_flag[robot] to this tick's epoch behind its write-through stores of force / tau / status (qr_mpc_kernel.hip): poll
        // it with agent-scope (sc1) loads -- bounded: 4 ms of the 100 MHz clock, then the robot is flagged, never silently wrong -- and take the
        // forces with loads of the same kind (they bypass this CU's L1, which may hold last tick's line).
        unsigned v = 0;
        const long long t0 = wall_clock64();
        for (;;) {
            v = __hip_atomic_load(pipe.flag + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 1) == pipe.epoch && !(pipe.wait_list && (v & 1u))) break;
            if (wall_clock64() - t0 > (((v >> 1) == pipe.epoch && pipe.wait_list) ? pipe.wait_ticks : pipe.flag_ticks)) {
                // Never silent, also when the solve is still running and will store its status word OVER the one this workgroup writes: leave
                // "gave up in this epoch" in the flag word itself (bit 31; epochs stay below 2^30) -- the solve raises the flag with an exchange
                // and, finding that value, adds QRGPU_ST_PIPE_TIMEOUT to the status word it has just stored (qr_mpc_kernel.hip).
                pipe_st = QRGPU_ST_PIPE_TIMEOUT_D;
                if (lane == 0) __hip_atomic_store(const_cast<unsigned *>(pipe.flag) + rid, 0x80000000u | (pipe.epoch << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        v = __builtin_amdgcn_readfirstlane(v);
        if (QW_P_TLR && lane == 0) QW_P_TLR[n + rid] = (int)wall_clock64();
        if (!pipe_st && (v & 1u)) { wbc_signal_done(pipe.finished, lane); return; }              // on the MPC's list pass: the second WBC pass behind that launch takes this robot
        if (lane < 12) cm[51 + lane] = (real)__hip_atomic_load(g_fr + (size_t)lane * n + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wsync();
    }
    if (pipe.wait_epoch && ((volatile int *)&sPipe)[0]) pipe_st = QRGPU_ST_PIPE_TIMEOUT_D;
    wbc_qp_and_store(lane, rid, n, K, nc, cpack, bad_type, eq_dependent, A, JC, cm, W, sI, g_tau, g_status, merge_tau, status_or | (pipe_st ? 2 : 0), epilogue, dbgT, g_qp,
                     pipe.flag != nullptr || pipe.wbc_done != nullptr, g_prev);
    if (pipe.wbc_done) {
        // overlapped ticks: g_prev of this robot is on its way to memory (written through by this wave): wait, then tell the robot's next WBC pass
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) qr_epoch_raise(pipe.wbc_done + rid, pipe.epoch);
    }
    wbc_signal_done(pipe.finished, lane);
    if (QW_P_TL && lane == 0) atomicMax(QW_P_TL + (pipe.epoch & 63u) * 8 + (pipe.second ? 7 : 4), wall_clock64());
    if (QW_P_TLR && lane == 0 && !pipe.second) QW_P_TLR[2 * n + rid] = (int)wall_clock64();
}

}  // namespace qrgpu
