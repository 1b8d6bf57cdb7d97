// ============================================================================
// Force-balance ("VMC") stance QP, one 64-lane wavefront per robot:
//   ComputeContactForce, control-frame overload   quadruped/src/controllers/balance_controller/qr_qp_torque_optimizer.cpp:190-301
//   (ComputeMassMatrix :31-57, ComputeConstraintMatrix :60-110, ComputeObjectiveMatrix :152-179, ComputeWeightMatrix :183-187)
//   qrRobot::MapContactForceToJointTorques         quadruped/src/robots/qr_robot.cpp:241-251
//
// 12 unknowns (one 3-vector per foot), 24 inequality rows (normal-force window 2 x 4, friction pyramid 4 x 4), each row touching
// one foot.  fp32 assembly exactly as the oracle states it (k-ordered fmaf chains, contraction off), then in fp64:
// G^-1 by symmetric sweep, and the Goldfarb-Idnani dual active set in Schur-complement form with QuadProg++'s decisions
// (QX/QuadProgpp/src/QuadProg++.cc): most violated row first, |psi| <= m eps c1 c2 100 termination with c1 = tr G,
// c2 = sum 1/L_jj, "z = 0 => no primal step" (asked relative to n'Mn, see below).  A swing foot carries the contradictory pair n.x >= 1e-7, -n.x >= 1e-7
// (:79-81): QuadProg++ then returns +inf with the iterate it had, and the reference only tests x for NaN (:281-285) -- the
// kernel stops at the same iterate and raises QRGPU_ST_VMC_INFEAS so the caller can see it.
// QuadProg++'s Cholesky reads one triangle of G: the QP solved is the mirrored LOWER triangle of the fp32 G.
// ============================================================================
#include <hip/hip_runtime.h>
#include "qr_device_types.h"
#include "qr_wave_helpers.h"

namespace qrgpu {

namespace {
__device__ __forceinline__ void vsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float chain3(float a0, float b0, float a1, float b1, float a2, float b2)
{
#pragma clang fp contract(off)
    float acc = 0.f;
    acc = __builtin_fmaf(a0, b0, acc); acc = __builtin_fmaf(a1, b1, acc); acc = __builtin_fmaf(a2, b2, acc);
    return acc;
}

// fp32 QP data -> LDS.  sIn: the 37 inputs.  Outputs: Gf[144], af[12], cn[24][3], bf[24].
__device__ __forceinline__ void vmc_assemble(int lane, const VmcType &C, const float *sIn, float *sA /*9*3 scratch*/, float *xc, float *Mm, float *Gf,
                                             float *af, float *cn, float *bf, const float *ratio /* fMinRatio[4], fMaxRatio[4] or null */)
{
#pragma clang fp contract(off)
    const float *pb = sIn, *acc_des = sIn + 12, *ct = sIn + 18, *R = sIn + 22, *gv = sIn + 31, *nrm = sIn + 34;
    float *T = sA, *Ic = sA + 9, *Iinv = sA + 18;
    if (lane < 9) { const int i = lane / 3, j = lane - 3 * i; T[lane] = chain3(R[3 * i], C.inertia[0 + 3 * j], R[3 * i + 1], C.inertia[1 + 3 * j], R[3 * i + 2], C.inertia[2 + 3 * j]); }
    vsync();
    if (lane < 9) { const int i = lane / 3, j = lane - 3 * i; Ic[lane] = chain3(T[3 * i], R[3 * j], T[3 * i + 1], R[3 * j + 1], T[3 * i + 2], R[3 * j + 2]); }
    vsync();
    {
        float cof[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
                cof[3 * i + j] = Ic[3 * i1 + j1] * Ic[3 * i2 + j2] - Ic[3 * i1 + j2] * Ic[3 * i2 + j1];
            }
        const float det = chain3(Ic[0], cof[0], Ic[1], cof[1], Ic[2], cof[2]);
        const float idet = 1.f / det;
        if (lane < 9) { const int i = lane / 3, j = lane - 3 * i; float v = 0.f;
#pragma unroll
            for (int e = 0; e < 9; ++e) if (e == 3 * j + i) v = cof[e];
            Iinv[lane] = v * idet; }
        if (lane >= 16 && lane < 28) { const int e = lane - 16, l = e / 3, i = e - 3 * l; xc[e] = chain3(R[3 * i], pb[3 * l], R[3 * i + 1], pb[3 * l + 1], R[3 * i + 2], pb[3 * l + 2]); }
    }
    vsync();
    const float im = 1.f / C.mass;
    for (int e = lane; e < 72; e += 64) {
        const int k = e / 12, col = e - 12 * k, l = col / 3, j = col - 3 * l;
        float v;
        if (k < 3) v = (k == j) ? im : 0.f;
        else {
            const int i = k - 3;
            const float x0 = xc[3 * l], x1 = xc[3 * l + 1], x2 = xc[3 * l + 2];
            // column j of the skew matrix [0 -x2 x1; x2 0 -x0; -x1 x0 0]
            const float s0 = (j == 0) ? 0.f : (j == 1 ? -x2 : x1), s1 = (j == 0) ? x2 : (j == 1 ? 0.f : -x0), s2 = (j == 0) ? -x1 : (j == 1 ? x0 : 0.f);
            v = chain3(Iinv[3 * i], s0, Iinv[3 * i + 1], s1, Iinv[3 * i + 2], s2);
        }
        Mm[e] = v;
    }
    vsync();
    for (int e = lane; e < 144; e += 64) {
        const int i = e / 12, j = e - 12 * i;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = __builtin_fmaf(Mm[12 * k + i] * C.acc_weight[k], Mm[12 * k + j], acc);
        float g = acc + C.reg_weight;
        g = g + ((i == j) ? 1e-4f : 0.f);
        Gf[e] = g;
    }
    if (lane < 12) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) { const float gk = (k < 3) ? gv[k] : 0.f; acc = __builtin_fmaf((gk + acc_des[k]) * C.acc_weight[k], Mm[12 * k + lane], acc); }
        af[lane] = acc;
    }
    if (lane < 24) {
        const float t2v[3] = {0.f, 1.f, 0.f};
        const float t1v[3] = {t2v[1] * nrm[2] - t2v[2] * nrm[1], t2v[2] * nrm[0] - t2v[0] * nrm[2], t2v[0] * nrm[1] - t2v[1] * nrm[0]};
        const float fMin = C.fmin_ratio * C.mass * 9.8f, fMax = C.fmax_ratio * C.mass * 9.8f;
        float b;
        if (lane < 8) {
            const int l = lane >> 1; const bool neg = lane & 1;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) cn[3 * lane + ax] = neg ? -nrm[ax] : nrm[ax];
            if (ratio) {
                // world-frame overload: lb = ratio[leg] * mass * 9.8 with the double literal (qr_qp_torque_optimizer.cpp:133-134)
                const float rt = ratio[neg ? 4 + l : l];
                const float w = (float)((double)((neg ? -rt : rt) * C.mass) * 9.8);
                b = (ct[l] > 0.f) ? w : 1e-7f;
            } else
            b = (ct[l] > 0.f) ? (neg ? -fMax : fMin) : 1e-7f;
        } else {
            const int r = (lane - 8) & 3;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const float mn = C.friction * nrm[ax];
                cn[3 * lane + ax] = (r == 0) ? mn + t1v[ax] : (r == 1) ? mn - t1v[ax] : (r == 2) ? mn + t2v[ax] : mn - t2v[ax];
            }
            b = 0.f;
        }
        bf[lane] = b;
    }
    vsync();
}
}  // namespace

__global__ void __launch_bounds__(64) qr_vmc_kernel(VmcLaunch P, const int *__restrict__ type_id, const float *__restrict__ g_in,
                                                    const float *__restrict__ g_q, float *__restrict__ g_force, float *__restrict__ g_tau,
                                                    int *__restrict__ g_status)
{
    const int lane = threadIdx.x;
    const int n = P.n;
    const int rid = xcd_robot_index(blockIdx.x, n);
    if (rid < 0) return;
    const VmcType &C = P.type[type_id ? type_id[rid] : 0];

    __shared__ float sIn[48], sA[27], xc[12], Mm[72], Gf[144], af[12], cn[72], bf[24];
    __shared__ double Md[144], colv[12], xd[12], wd[12], zd[12], Sq[13 * 13], dd[12], rr[12], uu[13], mna[12 * 12];
    __shared__ int act[12];

    if (lane < 37) sIn[lane] = g_in[(size_t)lane * n + rid];
    if (P.ratio && lane >= 40 && lane < 48) sIn[lane] = P.ratio[(size_t)(lane - 40) * n + rid];
    vsync();
    vmc_assemble(lane, C, sIn, sA, xc, Mm, Gf, af, cn, bf, P.ratio ? sIn + 40 : nullptr);

    // ---- G (mirrored lower triangle) -> fp64, c1 = tr G
    for (int e = lane; e < 144; e += 64) { const int i = e / 12, j = e - 12 * i; Md[e] = (double)Gf[12 * (i > j ? i : j) + (i > j ? j : i)]; }
    vsync();
    double c1 = 0.0, c2 = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) c1 += Md[13 * i];
    // ---- symmetric sweep: Md <- -G^-1 ; the pivots are the LDL^T pivots, L_jj = sqrt(pivot)
    int st = 0;
    for (int k = 0; k < 12; ++k) {
        if (lane < 12) colv[lane] = Md[12 * lane + k];
        vsync();
        const double piv = colv[k];
        if (!(piv > 0.0)) st |= QRGPU_ST_VMC_INFEAS_D;
        c2 += 1.0 / __builtin_sqrt(piv);
        const double ip = 1.0 / piv;
        for (int e = lane; e < 144; e += 64) {
            const int i = e / 12, j = e - 12 * i;
            double v;
            if (i == k) v = (j == k) ? -ip : colv[j] * ip;
            else if (j == k) v = colv[i] * ip;
            else v = Md[e] - colv[i] * colv[j] * ip;
            Md[e] = v;
        }
        vsync();
    }
    for (int e = lane; e < 144; e += 64) Md[e] = -Md[e];          // M = +G^-1
    vsync();
    // ---- x = -G^-1 g0 = M a   (g0 = -a, :249-252)
    if (lane < 12) { double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc += Md[12 * lane + k] * (double)af[k];
        xd[lane] = acc; }
    vsync();

    // ---- dual active set.  lane c < 24 is inequality row c: foot lc, normal nc, offset ci0 = -b.
    const bool isrow = lane < 24;
    const int lc = isrow ? (lane < 8 ? lane >> 1 : (lane - 8) >> 2) : 0;
    const double n0 = isrow ? (double)cn[3 * lane] : 0.0, n1 = isrow ? (double)cn[3 * lane + 1] : 0.0, n2 = isrow ? (double)cn[3 * lane + 2] : 0.0;
    const double ci0 = isrow ? -(double)bf[lane] : 0.0;
    const double eps = 2.220446049250313e-16, INF = __builtin_inf();
    const double term = 24.0 * eps * c1 * c2 * 100.0;
    unsigned active = 0, excluded = 0;            // bit c (uniform)
    int q = 0, iter = 0;
    const int maxit = 50 * 36 + 100;
    bool stop = (st != 0);
    while (!stop) {
        if (++iter > maxit) { st |= QRGPU_ST_VMC_MAXITER_D; break; }
        double s = isrow ? ci0 + n0 * xd[3 * lc] + n1 * xd[3 * lc + 1] + n2 * xd[3 * lc + 2] : 0.0;
        const double psi = wave_sum_d(s < 0.0 ? s : 0.0);
        const bool cand = isrow && !(((active | excluded) >> lane) & 1u) && s < 0.0;
        const double smin = wave_min_d(cand ? s : INF);
        if (!(smin < INF) || __builtin_fabs(psi) <= term) break;
        const int ip = first_lane(cand && s == smin);
        const int lp = (ip < 8) ? ip >> 1 : (ip - 8) >> 2;
        const double p0 = readlane_d(n0, ip), p1 = readlane_d(n1, ip), p2 = readlane_d(n2, ip);
        double sip = smin;
        double unew = 0.0;
        for (;;) {
            if (++iter > maxit) { st |= QRGPU_ST_VMC_MAXITER_D; stop = true; break; }
            // w = M n_p ; d = N_A' w ; r = S^-1 d ; z = w - M N_A r
            if (lane < 12) wd[lane] = Md[12 * lane + 3 * lp] * p0 + Md[12 * lane + 3 * lp + 1] * p1 + Md[12 * lane + 3 * lp + 2] * p2;
            vsync();
            if (lane < q) {
                const int c = act[lane], l = (c < 8) ? c >> 1 : (c - 8) >> 2;
                dd[lane] = (double)cn[3 * c] * wd[3 * l] + (double)cn[3 * c + 1] * wd[3 * l + 1] + (double)cn[3 * c + 2] * wd[3 * l + 2];
            }
            vsync();
            if (lane < q) {
                double sv[12], dv[12];                       // every load is issued before the first use (q <= 12)
#pragma unroll
                for (int j = 0; j < 12; ++j) { const bool ok = j < q; sv[j] = ok ? Sq[13 * lane + j] : 0.0; dv[j] = ok ? dd[j] : 0.0; }
                double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int j = 0; j < 12; j += 3) { a0 += sv[j] * dv[j]; a1 += sv[j + 1] * dv[j + 1]; a2 += sv[j + 2] * dv[j + 2]; }
                rr[lane] = (a0 + a1) + a2;
            }
            vsync();
            if (lane < 12) {
                double mv[12], rv[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) { const bool ok = i < q; mv[i] = ok ? mna[12 * i + lane] : 0.0; rv[i] = ok ? rr[i] : 0.0; }
                double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int i = 0; i < 12; i += 3) { a0 += mv[i] * rv[i]; a1 += mv[i + 1] * rv[i + 1]; a2 += mv[i + 2] * rv[i + 2]; }
                zd[lane] = wd[lane] - ((a0 + a1) + a2);
            }
            vsync();
            const double znp = zd[3 * lp] * p0 + zd[3 * lp + 1] * p1 + zd[3 * lp + 2] * p2;
            double tt = INF;
            if (lane < q) { const double rj = rr[lane]; if (rj > 0.0) tt = uu[lane] / rj; }
            const double t1 = wave_min_d(tt);
            const int l = (t1 < INF) ? first_lane(lane < q && tt == t1) : -1;
            // QuadProg++ asks |z|^2 > eps of a z built from orthogonal factors; here z = w - M N r cancels to ~1e-16 |w|, so the
            // same question is asked relative to delta = n'Mn (a dependent row -- the second row of a swing foot's 1e-7 pair --
            // gives |z.n| / delta <= 1e-10 on the test batches, an independent one >= 1e-6: scratch/proto_vmc.py)
            const double delta = wd[3 * lp] * p0 + wd[3 * lp + 1] * p1 + wd[3 * lp + 2] * p2;
            const double t2 = (znp > 1e-8 * delta) ? -sip / znp : INF;
            const double t = t1 < t2 ? t1 : t2;
            if (!(t < INF)) { st |= QRGPU_ST_VMC_INFEAS_D; stop = true; break; }       // QuadProg++ returns inf here; x stays as it is
            const bool dual_only = !(t2 < INF);
            if (!dual_only && lane < 12) xd[lane] += t * zd[lane];
            if (lane < q) uu[lane] -= t * rr[lane];
            unew += t;
            vsync();
            if (!dual_only && t == t2) {
                // full step: row ip joins the working set at position q (bordered update of S^-1 with 1 / z'n_p)
                const double isg = 1.0 / znp;
                { const int rq_ = (65536 + q - 1) / q; for (int e = lane; e < q * q; e += 64) { const int i = (e * rq_) >> 16, j = e - i * q; Sq[13 * i + j] += rr[i] * rr[j] * isg; } }
                if (lane < q) { Sq[13 * q + lane] = -rr[lane] * isg; Sq[13 * lane + q] = -rr[lane] * isg; }
                if (lane == 0) { Sq[13 * q + q] = isg; act[q] = ip; uu[q] = unew; }
                if (lane < 12) mna[12 * q + lane] = wd[lane];                            // M n_p, kept for z
                active |= 1u << ip; excluded = 0;
                ++q;
                vsync();
                break;
            }
            // partial or dual-only step: position l leaves the working set
            {
                const int last = q - 1;
                const int cl = __builtin_amdgcn_readfirstlane(act[l]);
                if (lane < q) dd[lane] = Sq[13 * lane + l];
                vsync();
                const double isl = 1.0 / dd[l];
                { const int rq_ = (65536 + q - 1) / q; for (int e = lane; e < q * q; e += 64) { const int i = (e * rq_) >> 16, j = e - i * q; if (i != l && j != l) Sq[13 * i + j] -= dd[i] * dd[j] * isl; } }
                vsync();
                if (l != last) {
                    if (lane < last) rr[lane] = (lane == l) ? Sq[13 * last + last] : Sq[13 * last + lane];
                    double mv = (lane < 12) ? mna[12 * last + lane] : 0.0;
                    vsync();
                    if (lane < last) { Sq[13 * l + lane] = rr[lane]; Sq[13 * lane + l] = rr[lane]; }
                    if (lane < 12) mna[12 * l + lane] = mv;
                    if (lane == 0) { act[l] = act[last]; uu[l] = uu[last]; }
                }
                active &= ~(1u << cl);
                --q;
                vsync();
                if (!dual_only) sip = readlane_d(ci0, ip) + p0 * xd[3 * lp] + p1 * xd[3 * lp + 1] + p2 * xd[3 * lp + 2];
            }
        }
    }

    // ---- X = -x (:293-297), force = (X Rcb)^T (:300), tau = J^T force
    if (lane < 12) {
        const int l = lane / 3, j = lane - 3 * l;
        const float X0 = -(float)xd[3 * l], X1 = -(float)xd[3 * l + 1], X2 = -(float)xd[3 * l + 2];
        const float *R = sIn + 22;
        const float f = chain3(X0, R[j], X1, R[3 + j], X2, R[6 + j]);
        g_force[(size_t)lane * n + rid] = f;
        Gf[lane] = f;
    }
    vsync();
    if (g_tau && g_q && lane < 12) {
        const int leg = lane / 3, j = lane - 3 * leg;
        const float t0 = g_q[(size_t)(3 * leg) * n + rid], t1 = g_q[(size_t)(3 * leg + 1) * n + rid], t2 = g_q[(size_t)(3 * leg + 2) * n + rid];
        float J0, J1, J2;
        leg_jacobian_column(j, t0, t1, t2, C.hip_l * ((leg & 1) ? 1.f : -1.f), C.upper_l, C.lower_l, J0, J1, J2);
        g_tau[(size_t)lane * n + rid] = J0 * Gf[3 * leg] + J1 * Gf[3 * leg + 1] + J2 * Gf[3 * leg + 2];
    }
    if (lane == 0 && g_status) g_status[rid] = st | (iter << 8);
}

}  // namespace qrgpu
