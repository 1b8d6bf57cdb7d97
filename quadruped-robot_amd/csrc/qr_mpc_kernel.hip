// ============================================================================
// Convex-MPC tick for a batch of quadrupeds, one 256-thread workgroup per robot.
// gfx950 (MI355X) only.  Everything between the coalesced state load and the
// 24-float result store lives in LDS / registers.
//
// Replaces, per robot (reference: TopHillRobotics/quadruped-robot, QS/ = quadruped/src/):
//   K1  ComputeContinuousTimeStateSpaceMatrices   QS/controllers/mpc/qr_mpc_interface.cpp:296-331
//   K2  ConvertToDiscreteQP                        :257-293   (closed form: [A B;0 0]^3 = 0)
//   K3  X_d / U_b fill                             :376-390
//   K4  qH = 2 Bqp'L Bqp + 2aI, qg                 :396-412
//   K5  fmat                                       :230-240   (never materialised)
//   K6  qpOASES QProblem::init                     :428-438   (own dual active-set solver, fp64)
//   K7  force -> torque                            qr_mpc_stance_leg_controller.cpp:402-409,139-153,
//                                                  QS/robots/qr_robot.cpp:148-172,241-251
//
// Phases (barriers between them):
//   0  load state/traj/gait to LDS; 1  SRBD terms R, U_p = Iw^-1 [r_p]x, T_p = R'U_p, free
//   (stance) leg-step list, v = Aqp x0 - X_d; 2  Hessian blocks for stance x stance leg-step
//   pairs as fp32 k-ordered fmaf chains (bit-identical to the CPU oracle's dense GEMM: skipped
//   terms are exact zeros), averaged with the transposed entry in fp64 -> packed lower triangle;
//   3  in-place symmetric sweep inverse M = H^-1 (fp64, packed, LDS);  4  x = -M g;
//   5  (wave 0 only) Goldfarb-Idnani dual active set in Schur-complement form: with the pyramid
//   rows having <= 2 non-zeros, M c_p is two rows of M, S = N'MN is read off M, and S^-1 is kept
//   explicitly by bordered-inverse updates;  6  J' f torques, store.
// ============================================================================
#include <hip/hip_runtime.h>
#include "qr_device_types.h"

namespace qrgpu {

#define QR_MPC_THREADS 256

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
}
__device__ __forceinline__ float det2(float a, float b, float c, float d) { return __builtin_fmaf(a, b, -(c * d)); }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) >> 1; }
__device__ __forceinline__ int pidx(int i, int j) { return i >= j ? tri(i) + j : tri(j) + i; }

// Pyramid constraint `cid = 6*k + t` of free leg-step k (rows of f_block, :232-236, plus the
// two-sided f_z row split in two):  c'u + ci0 >= 0 with c = ca*e[ia] + cb*e[ib].
struct Cons { int ia, ib; double ca, cb; };
__device__ __forceinline__ Cons decode_cons(int cid, double im)
{
    const int k = cid / 6, t = cid - 6 * k;
    Cons c;
    c.ib = 3 * k + 2;
    if (t < 4) { c.ia = 3 * k + (t >> 1); c.ca = (t & 1) ? -im : im; c.cb = 1.0; }
    else       { c.ia = 3 * k + 2;        c.ca = (t == 4) ? 1.0 : -1.0; c.cb = 0.0; }
    return c;
}

__global__ __launch_bounds__(QR_MPC_THREADS, 2)
void qr_mpc_kernel(MpcLaunch P, const int *__restrict__ type_id, const float *__restrict__ g_state,
                   const float *__restrict__ g_traj, const float *__restrict__ g_gait, const float *__restrict__ g_q,
                   float *__restrict__ g_force, float *__restrict__ g_tau, int *__restrict__ g_status,
                   float *__restrict__ dbgH, float *__restrict__ dbgG, float *__restrict__ g_force_wbc, int force_stride)
{
    const int rid = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int n = P.n;
    if (rid >= n) return;
    const MpcType &C = P.type[type_id ? type_id[rid] : 0];
    const int h = P.horizon;
    const int NV = 12 * h, NL = 4 * h;

    // ---------------- LDS carve ----------------
    extern __shared__ double smem[];
    double *xv = smem;                 // [NV] primal
    double *wv = xv + NV;              // [NV] M c_p; sweep pivot column
    double *zv = wv + NV;              // [NV] primal step
    double *yv = zv + NV;              // [NV] g, later N r
    double *dv = yv + NV;              // [QH] N' w
    double *rv = dv + QR_QH;           // [QH] S^-1 d
    double *uv = rv + QR_QH;           // [QH] multipliers
    double *fmk = uv + QR_QH;          // [NL] f_z upper bound per free leg-step
    float *sT = (float *)(fmk + NL);   // [4][3][3]
    float *sU = sT + 36;               // [4][3][3]
    float *sSt = sU + 36;              // [28]
    float *sTraj = sSt + 28;           // [12h]
    float *sGait = sTraj + NV;         // [4h]
    float *sV = sGait + NL;            // [13h]
    int *sLs = (int *)(sV + 13 * h);   // [NL] free leg-step -> original leg-step
    int *sAct = sLs + NL;              // [QH] active constraint ids
    short *sPos = (short *)(sAct + QR_QH);   // [6 NL] constraint id -> position in sAct, or -1
    int *sMisc = (int *)(sPos + 6 * NL + ((6 * NL) & 1));   // [4]
    double *Mp = (double *)(((uintptr_t)(sMisc + 4) + 7) & ~(uintptr_t)7);

    // ---------------- phase 0: inputs ----------------
    if (tid < 28) sSt[tid] = g_state[(size_t)tid * n + rid];
    for (int i = tid; i < NV; i += QR_MPC_THREADS) sTraj[i] = g_traj[(size_t)i * n + rid];
    for (int i = tid; i < NL; i += QR_MPC_THREADS) sGait[i] = g_gait[(size_t)i * n + rid];
    __syncthreads();

    // ---------------- phase 1: SRBD terms (every thread keeps R in registers) ----------------
    float R[3][3];
    {
        const float w = sSt[6], x = sSt[7], y = sSt[8], z = sSt[9];
        const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
        const float twx = tx * w, twy = ty * w, twz = tz * w;
        const float txx = tx * x, txy = ty * x, txz = tz * x;
        const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
        R[0][0] = 1.f - (tyy + tzz); R[0][1] = txy - twz;         R[0][2] = txz + twy;
        R[1][0] = txy + twz;         R[1][1] = 1.f - (txx + tzz); R[1][2] = tyz - twx;
        R[2][0] = txz - twy;         R[2][1] = tyz + twx;         R[2][2] = 1.f - (txx + tyy);
    }
    const float dt = C.dt, dt2 = C.dt * C.dt, minv = 1.0f / C.mass;
    if (tid < 4) {
        const int p = tid;
        float RI[3][3], Iw[3][3], cof[3][3], Iinv[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) RI[i][k] = R[i][k] * C.inertia[k];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Iw[i][j] = dot3(RI[i][0], R[j][0], RI[i][1], R[j][1], RI[i][2], R[j][2]);
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
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Iinv[i][j] = cof[j][i] * invdet;
        const float rx = sSt[13 + 3 * p], ry = sSt[14 + 3 * p], rz = sSt[15 + 3 * p];
        float U[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            U[i][0] = det2(Iinv[i][1], rz, Iinv[i][2], ry);
            U[i][1] = det2(Iinv[i][2], rx, Iinv[i][0], rz);
            U[i][2] = det2(Iinv[i][0], ry, Iinv[i][1], rx);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                sU[9 * p + 3 * i + j] = U[i][j];
                sT[9 * p + 3 * i + j] = dot3(R[0][i], U[0][j], R[1][i], U[1][j], R[2][i], U[2][j]);
            }
    }
    // free (stance) leg-steps: U_b(5k+4) = gait*fMax > 0   (:387)
    if (tid < 64) {
        const bool fr = (tid < NL) && (sGait[tid < NL ? tid : 0] * C.fmax > 0.f);
        const unsigned long long mask = __ballot(fr);
        if (fr) {
            const int pos = __popcll(mask & ((1ull << tid) - 1ull));
            sLs[pos] = tid;
            fmk[pos] = (double)(sGait[tid] * C.fmax);
        }
        if (tid == 0) sMisc[0] = __popcll(mask);
    }
    // v = Aqp x0 - X_d, one horizon step per thread (wave 1 so it overlaps the above)
    if (tid >= 64 && tid < 64 + h) {
        const int r = tid - 64;
        const float grav = -9.8f;
        const float kd = (float)(r + 1) * dt;
        const float hk2 = (kd * kd) * 0.5f;
        float ax[13];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = sSt[25 + i];                                     // rpy
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = __builtin_fmaf(kd * R[j][i], sSt[10 + j], acc);
            ax[i] = acc;
            ax[3 + i] = __builtin_fmaf(kd, sSt[3 + i], sSt[i]);         // p + kd v
            ax[6 + i] = sSt[10 + i];
            ax[9 + i] = sSt[3 + i];
        }
        ax[5] = __builtin_fmaf(hk2, grav, ax[5]);
        ax[11] = __builtin_fmaf(kd, grav, ax[11]);
        ax[12] = grav;
#pragma unroll
        for (int j = 0; j < 12; ++j) sV[13 * r + j] = ax[j] - sTraj[12 * r + j];
        sV[13 * r + 12] = ax[12] - 0.f;
    }
    __syncthreads();
    const int nls = sMisc[0];
    const int ns = 3 * nls;
    const size_t lds_doubles = (size_t)P.lds_bytes / 8;
    const size_t mp_off = (size_t)(Mp - smem);
    double *Sinv = Mp + tri(ns);
    int qcap;
    {   // rows of S^-1 that fit behind the packed M
        long long rem = (long long)lds_doubles - (long long)mp_off - (long long)tri(ns);
        int qc = 0;
        if (rem > 0) { qc = (int)((__builtin_sqrt(8.0 * (double)rem + 1.0) - 1.0) * 0.5); while (tri(qc) > rem) --qc; }
        qcap = qc < QR_QH ? qc : QR_QH;
        if (qcap > ns) qcap = ns;
    }

    float w2[13];
#pragma unroll
    for (int s = 0; s < 12; ++s) w2[s] = 2.f * C.weights[s];
    w2[12] = 0.f;
    const float dtm = dt * minv;
    const float two_alpha = 2.f * C.alpha;

    // ---------------- phase 2: Hessian blocks + gradient ----------------
    const int npairs = tri(nls);
    for (int pid = tid; pid < npairs; pid += QR_MPC_THREADS) {
        int a = (int)((__builtin_sqrtf(8.f * (float)pid + 1.f) - 1.f) * 0.5f);
        while (tri(a + 1) <= pid) ++a;
        while (tri(a) > pid) --a;
        const int b = pid - tri(a);                 // a >= b
        const int la = sLs[a], lb = sLs[b];
        const int ia = la >> 2, pa = la & 3, ib = lb >> 2, pb = lb & 3;   // horizon step, leg
        float Ta[3][3], Tb[3][3], Ua[3][3], Ub[3][3], Uaw[3][3], Ubw[3][3];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Ta[s][j] = sT[9 * pa + 3 * s + j]; Tb[s][j] = sT[9 * pb + 3 * s + j];
                Ua[s][j] = dt * sU[9 * pa + 3 * s + j]; Ub[s][j] = dt * sU[9 * pb + 3 * s + j];      // G rows 6-8
                Uaw[s][j] = Ua[s][j] * w2[6 + s]; Ubw[s][j] = Ub[s][j] * w2[6 + s];                  // temp = G*2w
            }
        float hab[3][3], hba[3][3];      // hab[i][j] = H[3a+i][3b+j],  hba[j][i] = H[3b+j][3a+i]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { hab[i][j] = 0.f; hba[j][i] = 0.f; }
        const int r0 = ia > ib ? ia : ib;
        for (int r = r0; r < h; ++r) {
            const float caa = ((float)(r - ia) + 0.5f) * dt2, cab = ((float)(r - ib) + 0.5f) * dt2;
            // s = 0..2 : rows c_a * T
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float ga[3], gb[3], gaw[3], gbw[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    ga[j] = caa * Ta[s][j]; gb[j] = cab * Tb[s][j];
                    gaw[j] = ga[j] * w2[s]; gbw[j] = gb[j] * w2[s];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        hab[i][j] = __builtin_fmaf(gaw[i], gb[j], hab[i][j]);
                        hba[j][i] = __builtin_fmaf(gbw[j], ga[i], hba[j][i]);
                    }
            }
            // s = 3..5 : rows (c_a/m) e_i  -> only the (i,i) entry of the block
            {
                const float cama = caa * minv, camb = cab * minv;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    hab[i][i] = __builtin_fmaf(cama * w2[3 + i], camb, hab[i][i]);
                    hba[i][i] = __builtin_fmaf(camb * w2[3 + i], cama, hba[i][i]);
                }
            }
            // s = 6..8 : rows dt * U
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        hab[i][j] = __builtin_fmaf(Uaw[s][i], Ub[s][j], hab[i][j]);
                        hba[j][i] = __builtin_fmaf(Ubw[s][j], Ua[s][i], hba[j][i]);
                    }
            // s = 9..11 : rows (dt/m) e_i
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                hab[i][i] = __builtin_fmaf(dtm * w2[9 + i], dtm, hab[i][i]);
                hba[i][i] = __builtin_fmaf(dtm * w2[9 + i], dtm, hba[i][i]);
            }
        }
        if (a == b) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { hab[i][i] = hab[i][i] + two_alpha; hba[i][i] = hba[i][i] + two_alpha; }   // + 2 alpha I (:411)
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (a > b || j <= i) Mp[tri(3 * a + i) + 3 * b + j] = 0.5 * ((double)hab[i][j] + (double)hba[j][i]);
            }
        if (dbgH) {
            float *Hd = dbgH + (size_t)rid * NV * NV;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    Hd[(size_t)(3 * la + i) * NV + 3 * lb + j] = hab[i][j];
                    Hd[(size_t)(3 * lb + j) * NV + 3 * la + i] = hba[j][i];
                }
        }
    }
    // gradient: qg[a] = sum_k temp[a][k] v[k], one free variable per thread
    for (int e = tid; e < ns; e += QR_MPC_THREADS) {
        const int ls = sLs[e / 3], j = e % 3, ia = ls >> 2, p = ls & 3;
        const float t0 = sT[9 * p + j], t1 = sT[9 * p + 3 + j], t2 = sT[9 * p + 6 + j];
        const float u0 = (dt * sU[9 * p + j]) * w2[6], u1 = (dt * sU[9 * p + 3 + j]) * w2[7], u2 = (dt * sU[9 * p + 6 + j]) * w2[8];
        const float dw = dtm * w2[9 + j];
        float acc = 0.f;
        for (int r = ia; r < h; ++r) {
            const float ca = ((float)(r - ia) + 0.5f) * dt2;
            const float *vr = sV + 13 * r;
            acc = __builtin_fmaf((ca * t0) * w2[0], vr[0], acc);
            acc = __builtin_fmaf((ca * t1) * w2[1], vr[1], acc);
            acc = __builtin_fmaf((ca * t2) * w2[2], vr[2], acc);
            acc = __builtin_fmaf((ca * minv) * w2[3 + j], vr[3 + j], acc);
            acc = __builtin_fmaf(u0, vr[6], acc);
            acc = __builtin_fmaf(u1, vr[7], acc);
            acc = __builtin_fmaf(u2, vr[8], acc);
            acc = __builtin_fmaf(dw, vr[9 + j], acc);
        }
        yv[e] = (double)acc;
        if (dbgG) dbgG[(size_t)rid * NV + 3 * ls + j] = acc;
    }
    __syncthreads();

    // ---------------- phase 3: packed symmetric sweep, Mp <- -H^-1 ----------------
    int st = 0;
    {
        const int tx = tid & 15, ty = tid >> 4;
        const int nt = (ns + 15) >> 4;
        for (int k = 0; k < ns; ++k) {
            for (int i = tid; i < ns; i += QR_MPC_THREADS) wv[i] = Mp[pidx(i, k)];
            __syncthreads();
            const double piv = wv[k];
            if (!(piv > 0.0)) st |= QRGPU_ST_MPC_NOTSPD_D;
            const double ip = 1.0 / piv;
            for (int ta = 0; ta < nt; ++ta) {
                const int i = 16 * ta + ty;
                const double ci = (i < ns) ? wv[i] : 0.0;
                for (int tb = 0; tb <= ta; ++tb) {
                    const int j = 16 * tb + tx;
                    if (i < ns && j <= i) {
                        const double cj = wv[j];
                        double *m = &Mp[tri(i) + j];
                        double v;
                        if (i == k) v = (j == k) ? -ip : cj * ip;
                        else if (j == k) v = ci * ip;
                        else v = *m - ci * cj * ip;
                        *m = v;
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---------------- phase 4: x = -M g = Mp g ----------------
    for (int e = tid; e < ns; e += QR_MPC_THREADS) {
        double acc = 0.0;
        for (int j = 0; j < ns; ++j) acc += Mp[pidx(e, j)] * yv[j];
        xv[e] = acc;
    }
    for (int c = tid; c < 6 * nls; c += QR_MPC_THREADS) sPos[c] = -1;
    __syncthreads();
    if (tid >= 64) return;          // the active-set loop is a single wavefront; no barrier below

    // ---------------- phase 5: dual active set (wave 0) ----------------
    // NOTE Mp holds -H^-1: every use below flips the sign.
    const double im = (double)(1.f / C.mu);          // mu_ (:230) as fmat holds it
    const double tol = 1e-9;
    int q = 0, iter = 0;
    const int maxit = 40 * nls + 100;
    bool done = (nls == 0);
    while (!done) {
        // step 1: most violated inactive constraint (ties -> lowest id)
        double bs = -tol; int bc = 0x7fffffff;
        for (int c = lane; c < 6 * nls; c += 64) {
            if (sPos[c] >= 0) continue;
            const Cons cc = decode_cons(c, im);
            double s = cc.ca * xv[cc.ia] + cc.cb * xv[cc.ib];
            if (c - 6 * (c / 6) == 5) s += fmk[c / 6];
            if (s < bs) { bs = s; bc = c; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double os = __shfl_xor(bs, m, 64); const int oc = __shfl_xor(bc, m, 64);
            if (os < bs || (os == bs && oc < bc)) { bs = os; bc = oc; }
        }
        if (bc == 0x7fffffff) break;
        const int p = bc;
        const Cons cp = decode_cons(p, im);
        const double ci0p = (p - 6 * (p / 6) == 5) ? fmk[p / 6] : 0.0;
        double up = 0.0;
        for (;;) {
            if (++iter > maxit) { st |= QRGPU_ST_MPC_MAXITER_D; done = true; break; }
            // w = M c_p
            for (int e = lane; e < ns; e += 64) wv[e] = -(cp.ca * Mp[pidx(e, cp.ia)] + cp.cb * Mp[pidx(e, cp.ib)]);
            wave_sync();
            const double delta = cp.ca * wv[cp.ia] + cp.cb * wv[cp.ib];
            // d = N' w ; r = S^-1 d
            for (int j = lane; j < q; j += 64) { const Cons cj = decode_cons(sAct[j], im); dv[j] = cj.ca * wv[cj.ia] + cj.cb * wv[cj.ib]; }
            wave_sync();
            double dr = 0.0;
            for (int i = lane; i < q; i += 64) {
                double acc = 0.0;
                for (int j = 0; j < q; ++j) acc += Sinv[pidx(i, j)] * dv[j];
                rv[i] = acc;
                dr += acc * dv[i];
            }
            dr = wave_sum(dr);
            wave_sync();
            const double zc = delta - dr;                    // z'c_p
            // dual step length
            double t1 = __builtin_inf(); int lpos = 0x7fffffff;
            for (int j = lane; j < q; j += 64) {
                const double rj = rv[j];
                if (rj > 0.0) { const double tt = uv[j] / rj; if (tt < t1) { t1 = tt; lpos = j; } }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const double ot = __shfl_xor(t1, m, 64); const int ol = __shfl_xor(lpos, m, 64);
                if (ot < t1 || (ot == t1 && ol < lpos)) { t1 = ot; lpos = ol; }
            }
            const double sp = cp.ca * xv[cp.ia] + cp.cb * xv[cp.ib] + ci0p;
            const bool have_z = zc > 1e-13 * delta;
            const double t2 = have_z ? -sp / zc : __builtin_inf();
            const double t = t1 < t2 ? t1 : t2;
            if (!(t < __builtin_inf())) { st |= QRGPU_ST_MPC_INFEAS_D; done = true; break; }
            if (have_z) {
                // y = N r (gathered per variable), z = w - M y, x += t z
                for (int e = lane; e < ns; e += 64) {
                    const int k = e / 3, ax = e - 3 * k;
                    double acc = 0.0;
                    if (ax == 2) {
#pragma unroll
                        for (int tt = 0; tt < 6; ++tt) { const int ps = sPos[6 * k + tt]; if (ps >= 0) acc += (tt == 5 ? -1.0 : 1.0) * rv[ps]; }
                    } else {
                        const int p0 = sPos[6 * k + 2 * ax], p1 = sPos[6 * k + 2 * ax + 1];
                        if (p0 >= 0) acc += im * rv[p0];
                        if (p1 >= 0) acc -= im * rv[p1];
                    }
                    yv[e] = acc;
                }
                wave_sync();
                for (int e = lane; e < ns; e += 64) {
                    double acc = wv[e];
                    for (int k = 0; k < nls; ++k) {
                        const short *pk = sPos + 6 * k;
                        if ((pk[0] & pk[1] & pk[2] & pk[3] & pk[4] & pk[5]) >= 0) {    // any constraint of leg-step k active
                            acc += Mp[pidx(e, 3 * k)] * yv[3 * k] + Mp[pidx(e, 3 * k + 1)] * yv[3 * k + 1] + Mp[pidx(e, 3 * k + 2)] * yv[3 * k + 2];
                        }
                    }
                    zv[e] = acc;
                    xv[e] += t * acc;
                }
            }
            for (int j = lane; j < q; j += 64) uv[j] -= t * rv[j];
            up += t;
            wave_sync();
            if (have_z && t == t2) {
                // full step: p joins the working set; bordered update of S^-1
                if (q >= qcap) { st |= QRGPU_ST_MPC_OVERFLOW_D; done = true; break; }
                const double isg = 1.0 / zc;
                for (int e = lane; e < tri(q); e += 64) {
                    int i = (int)((__builtin_sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
                    while (tri(i + 1) <= e) ++i;
                    while (tri(i) > e) --i;
                    const int j = e - tri(i);
                    Sinv[e] += rv[i] * rv[j] * isg;
                }
                for (int j = lane; j < q; j += 64) Sinv[tri(q) + j] = -rv[j] * isg;
                if (lane == 0) { Sinv[tri(q) + q] = isg; sAct[q] = p; sPos[p] = (short)q; uv[q] = up; }
                ++q;
                wave_sync();
                break;
            }
            // partial or dual-only step: constraint at lpos leaves; downdate S^-1, move last into its slot
            {
                const int l = lpos, last = q - 1;
                for (int i = lane; i < q; i += 64) dv[i] = Sinv[pidx(i, l)];
                wave_sync();
                const double isl = 1.0 / dv[l];
                for (int e = lane; e < tri(q); e += 64) {
                    int i = (int)((__builtin_sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
                    while (tri(i + 1) <= e) ++i;
                    while (tri(i) > e) --i;
                    const int j = e - tri(i);
                    if (i != l && j != l) Sinv[e] -= dv[i] * dv[j] * isl;
                }
                wave_sync();
                if (l != last) {
                    for (int j = lane; j < last; j += 64) rv[j] = (j == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, j)];
                    wave_sync();
                    for (int j = lane; j < last; j += 64) Sinv[pidx(l, j)] = rv[j];
                }
                if (lane == 0) {
                    sPos[sAct[l]] = -1;
                    if (l != last) { sAct[l] = sAct[last]; uv[l] = uv[last]; sPos[sAct[l]] = (short)l; }
                }
                --q;
                wave_sync();
            }
        }
    }

    // ---------------- phase 6: outputs ----------------
    // f(axis,leg) = q_soln[3*leg+axis] for horizon step 0 (GetMPCSolution, :446-451); swing legs are 0.
    if (lane < 12) yv[lane] = 0.0;
    wave_sync();
    for (int e = lane; e < ns; e += 64) { const int ls = sLs[e / 3]; if (ls < 4) yv[3 * ls + e % 3] = xv[e]; }
    wave_sync();
    if (lane < 12) {
        const int leg = lane / 3, j = lane - 3 * leg;
        const float fx = (float)yv[3 * leg], fy = (float)yv[3 * leg + 1], fz = (float)yv[3 * leg + 2];
        g_force[(size_t)lane * n + rid] = (float)yv[lane];
        if (g_force_wbc) g_force_wbc[(size_t)(force_stride + lane) * n + rid] = (float)yv[lane];
        if (g_tau) {
            // f_ff = -R^T f  (R^T = quaternionToRotationMatrix(quat)), tau = J^T f_ff
            float fff[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) fff[i] = (-R[0][i]) * fx + (-R[1][i]) * fy + (-R[2][i]) * fz;
            const float t0 = g_q[(size_t)(3 * leg) * n + rid], t1 = g_q[(size_t)(3 * leg + 1) * n + rid], t2 = g_q[(size_t)(3 * leg + 2) * n + rid];
            const float lu = C.upper_l, ll = C.lower_l;
            const float sh = C.hip_l * ((leg & 1) ? 1.f : -1.f);
            const float lEff = sqrtf(lu * lu + ll * ll + 2 * lu * ll * cosf(t2));
            const float tEff = t1 + t2 / 2;
            float J0, J1, J2;     // column j of the leg Jacobian
            if (j == 0) {
                J0 = 0;
                J1 = -sh * sinf(t0) + lEff * cosf(t0) * cosf(tEff);
                J2 = sh * cosf(t0) + lEff * sinf(t0) * cosf(tEff);
            } else if (j == 1) {
                J0 = -lEff * cosf(tEff);
                J1 = -lEff * sinf(t0) * sinf(tEff);
                J2 = lEff * sinf(tEff) * cosf(t0);
            } else {
                J0 = ll * lu * sinf(t2) * sinf(tEff) / lEff - lEff * cosf(tEff) / 2;
                J1 = -ll * lu * sinf(t0) * sinf(t2) * cosf(tEff) / lEff - lEff * sinf(t0) * sinf(tEff) / 2;
                J2 = ll * lu * sinf(t2) * cosf(t0) * cosf(tEff) / lEff + lEff * sinf(tEff) * cosf(t0) / 2;
            }
            g_tau[(size_t)lane * n + rid] = J0 * fff[0] + J1 * fff[1] + J2 * fff[2];
        }
    }
    if (lane == 0 && g_status) g_status[rid] = st | (iter << 8);
}

}  // namespace qrgpu
