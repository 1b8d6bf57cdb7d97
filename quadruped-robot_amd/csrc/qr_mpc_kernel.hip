// ============================================================================
// Convex-MPC tick for a batch of quadrupeds, one 256-thread workgroup per robot.
// gfx950 (MI355X) only.  Everything between the state load and the 24-float
// result store lives in LDS / registers.
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
// Data layout.  The unknowns are grouped by "leg-step" (one foot at one horizon step, 3 force
// components); only stance leg-steps are free (swing ones are pinned to 0 by U_b = 0).  Every
// matrix is handled as 3x3 blocks indexed by (leg-step, leg-step):
//   phase 2  each thread owns up to MAXB lower-triangle blocks (a >= b) and builds them as fp32
//            k-ordered fmaf chains -- bit-identical to the CPU oracle's dense GEMM, the skipped terms
//            being exact zeros -- H[a][b] and H[b][a] separately, averaged in fp64;
//   phase 3  in-place symmetric *block* sweep (Gauss-Jordan without pivoting on an SPD matrix) with
//            the owned blocks held in registers; per pivot leg-step only the pivot block column goes
//            through LDS (double-buffered, one barrier per pivot);  result  M = H^-1  -> LDS,
//            block-packed lower triangle;
//   phase 4  x = -M g;
//   phase 5  wave 0 alone: Goldfarb-Idnani dual active set in Schur-complement form.  Lane k owns
//            leg-step k (x_k, w_k, z_k in registers).  The pyramid rows touch one leg-step, so
//            M c_p is one block column, S = N'MN is never formed: S^-1 is kept explicitly (packed,
//            LDS) by bordered-inverse updates / downdates.  Reductions use DPP, broadcasts v_readlane;
//   phase 6  f -> J'(-R'f) torques, store.
// ============================================================================
#include <hip/hip_runtime.h>
#include "qr_device_types.h"
#include "qr_wave_helpers.h"

namespace qrgpu {

#define QR_MPC_THREADS 256

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2)
{
#pragma clang fp contract(off)
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
}
__device__ __forceinline__ float det2(float a, float b, float c, float d)
{
#pragma clang fp contract(off)
    return __builtin_fmaf(a, b, -(c * d));
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) >> 1; }
__device__ __forceinline__ int pidx(int i, int j) { return i >= j ? tri(i) + j : tri(j) + i; }

// Pyramid row `t` of a leg-step (rows of f_block, :232-236, with the two-sided f_z row split):
//   c'f + ci0 >= 0,  c = (c0, c1, c2)
__device__ __forceinline__ void cons_vec(int t, double im, double &c0, double &c1, double &c2)
{
    c0 = (t == 0) ? im : ((t == 1) ? -im : 0.0);
    c1 = (t == 2) ? im : ((t == 3) ? -im : 0.0);
    c2 = (t == 5) ? -1.0 : 1.0;
}

struct Blk { double m[9]; };

// 3x3 block (a, b) of M read as seen from row leg-step `k` against column leg-step `kc`
// (block-packed lower triangle, 9 doubles per block, row-major).
__device__ __forceinline__ void load_block(const double *Mb, int k, int kc, Blk &B)
{
    // one code path for both triangles: element (i, j) sits at base + i*si + j*sj with (si, sj) = (3, 1) or (1, 3)
    const bool lower = k >= kc;
    const double *p = Mb + (lower ? tri(k) + kc : tri(kc) + k) * 9;
    const int si = lower ? 3 : 1, sj = lower ? 1 : 3;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) B.m[3 * i + j] = p[i * si + j * sj];
}

// One 3x3 block pair of the Hessian: hab = H[3a.., 3b..], hba = H[3b.., 3a..] as the fp32 k-ordered fmaf
// chains of  qH = temp * Bqp  (:411), then out = (hab + hba')/2 in fp64.  la/lb: original leg-step ids
// (4*step + leg).  The caller fences the scheduler between blocks so that only one block's temporaries are live.
__device__ __forceinline__ Blk hess_block(const float *sT, const float *sU, int la, int lb, int h, float dt, float dt2, float minv,
                                        const float *weights, float alpha, float *Hd, int NV)
{
#pragma clang fp contract(off)      // the fp32 chain must be exactly the written sequence (bit-identical to the oracle)
    Blk out;
    const int ia = la >> 2, pa = la & 3, ib = lb >> 2, pb = lb & 3;   // horizon step, leg
    float w2[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) w2[s] = 2.f * weights[s];
    const float dtm = dt * minv;
    const float two_alpha = 2.f * alpha;
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
    if (la == lb) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { hab[i][i] = hab[i][i] + two_alpha; hba[i][i] = hba[i][i] + two_alpha; }   // + 2 alpha I (:411)
    }
    // the stated QP depends on H only through (H + H')/2: average the two fp32 entries exactly in fp64
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) out.m[3 * i + j] = 0.5 * ((double)hab[i][j] + (double)hba[j][i]);
    if (Hd) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Hd[(size_t)(3 * la + i) * NV + 3 * lb + j] = hab[i][j];
                Hd[(size_t)(3 * lb + j) * NV + 3 * la + i] = hba[j][i];
            }
    }
    return out;
}

// Phase 6: first-step forces (staged in LDS, 12 doubles) -> force[12][n], and tau = J^T (-R^T f) per leg
// (qr_mpc_stance_leg_controller.cpp:402-409,139-153; AnalyticalLegJacobian QS/robots/qr_robot.cpp:148-172).
// Column j of the analytic leg Jacobian of leg `leg` (AnalyticalLegJacobian, QS/robots/qr_robot.cpp:148-172) from the joint angles in HBM:
// thread (leg, j) of phase 0 stores it in LDS, so that the torque map at the very end is three multiply-adds instead of a global-memory
// round trip and twenty sinf / cosf on the critical path of every robot.
__device__ __forceinline__ void mpc_jacobian_column(int leg, int j, int rid, int n, const MpcType &C, const float *__restrict__ g_q, float *sJ3)
{
    const float t0 = g_q[(size_t)(3 * leg) * n + rid], t1 = g_q[(size_t)(3 * leg + 1) * n + rid], t2 = g_q[(size_t)(3 * leg + 2) * n + rid];
    const float lu = C.upper_l, ll = C.lower_l;
    const float sh = C.hip_l * ((leg & 1) ? 1.f : -1.f);
    const float lEff = sqrtf(lu * lu + ll * ll + 2 * lu * ll * cosf(t2));
    const float tEff = t1 + t2 / 2;
    float J0, J1, J2;
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
    sJ3[0] = J0; sJ3[1] = J1; sJ3[2] = J2;
}

// epilogue (MPC-only batches; in the fused tick the WBC kernel applies it after the stance/swing merge): bit 0 = the +-0.9 N m abad
// compensation of qrFSMStateLocomotion::Run (QS/fsm/qr_fsm_state_locomotion.cpp:141-151), bit 1 = the +-23 N m clip of
// qrSafetyChecker::CheckForceFeedForward (QS/fsm/qr_safety_checker.cpp:48-66); legCmd.tua is a double there.
__device__ __forceinline__ float torque_epilogue(float tau, int motor, bool comp, int epilogue)
{
    double t = (double)tau;
    if (comp && (epilogue & 1) && motor % 3 == 0) t += (double)(((motor / 3) & 1) ? 0.9f : -0.9f);     // tua_ * pow(-1, (leg + 1) % 2)
    if (epilogue & 2) t = t > 23.0 ? 23.0 : (t < -23.0 ? -23.0 : t);
    return (float)t;
}

__device__ __forceinline__ void mpc_outputs(int lane, int rid, int n, const double *yl, const float (&R)[3][3], const float *sJ, const MpcType &C,
                                            const float *__restrict__ g_q, float *__restrict__ g_force, float *__restrict__ g_force_wbc,
                                            int force_stride, float *__restrict__ g_tau, int epilogue)
{
    if (lane < 12) {
        const int leg = lane / 3;
        const float fx = (float)yl[3 * leg], fy = (float)yl[3 * leg + 1], fz = (float)yl[3 * leg + 2];
        g_force[(size_t)lane * n + rid] = (float)yl[lane];
        if (g_force_wbc) g_force_wbc[(size_t)(force_stride + lane) * n + rid] = (float)yl[lane];
        if (g_tau) {
            // f_ff = -R^T f  (R^T = quaternionToRotationMatrix(quat)), tau = J^T f_ff
            float fff[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) fff[i] = (-R[0][i]) * fx + (-R[1][i]) * fy + (-R[2][i]) * fz;
            float Jl[3];
            if (!sJ) mpc_jacobian_column(leg, lane - 3 * leg, rid, n, C, g_q, Jl);      // (h = 16 variants: no LDS to spare for the early copy)
            const float J0 = sJ ? sJ[3 * lane] : Jl[0], J1 = sJ ? sJ[3 * lane + 1] : Jl[1], J2 = sJ ? sJ[3 * lane + 2] : Jl[2];
            const float tq = J0 * fff[0] + J1 * fff[1] + J2 * fff[2];
            g_tau[(size_t)lane * n + rid] = epilogue ? torque_epilogue(tq, lane, true, epilogue) : tq;
        }
    }
}

// Longest-processing-time-first dispatch order for the next launch.  A robot's solve time varies 10x with its active-set
// iteration count, and with two resident workgroups per CU a long solve that starts late sets the kernel time.  Block
// dispatch follows blockIdx, so each XCD chunk [x*chunk, (x+1)*chunk) is counting-sorted by the cost the robots had in
// the previous launch, descending (control ticks are temporally coherent; a stale cost only costs speed).  Robots never
// leave their XCD chunk, so the L2 locality of xcd_robot_index() is kept.  grid = 8, one workgroup per chunk.
__device__ __forceinline__ void lpt_order_chunk(int x, int n, const int *__restrict__ cost, int *__restrict__ order, int *hist /* >= 2048 ints of LDS */)
{
    const int chunk = (n + 7) >> 3;
    const int lo = x * chunk;
    const int hi = (lo + chunk < n) ? lo + chunk : n;
    if (hi - lo <= 2048) {
        // rank by comparison: robot i goes to slot #{j : c_j > c_i, or c_j = c_i and j < i}.  Every thread reads the same c_j (an LDS
        // broadcast), no atomics -- the counting sort below spends ~0.5 ms at 512 robots per chunk on atomics to a few hot bins -- and the
        // order is stable, i.e. the same for the same costs.
        const int m = hi - lo;
        for (int i = threadIdx.x; i < m; i += 256) hist[i] = cost[lo + i] & 255;
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 256) {
            const int ci = hist[i];
            int rank = 0;
            for (int j = 0; j < m; ++j) { const int cj = hist[j]; rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
            order[lo + rank] = lo + i;
        }
        return;
    }
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&hist[255 - (cost[i] & 255)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < 256; ++b) { const int c = hist[b]; hist[b] = acc; acc += c; }
    }
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += 256) order[lo + atomicAdd(&hist[255 - (cost[i] & 255)], 1)] = i;
}
__global__ void __launch_bounds__(256) qr_lpt_order_kernel(int n, const int *__restrict__ cost, int *__restrict__ order)
{
    __shared__ int hist[2048];
    lpt_order_chunk(blockIdx.x, n, cost, order, hist);
}

// TAG only names the instance: the rescue launch runs <4, true, 1>, so that a kernel trace keeps it apart from the main launches <4, true, 0>
template <int MAXB, bool MULTI, int TAG>
__global__ __launch_bounds__(QR_MPC_THREADS, (MAXB <= 4 ? 2 : 1))
void qr_mpc_kernel(MpcLaunch P, const int *__restrict__ type_id, const float *__restrict__ g_state,
                   const float *__restrict__ g_traj, const float *__restrict__ g_gait, const float *__restrict__ g_q,
                   float *__restrict__ g_force, float *__restrict__ g_tau, int *__restrict__ g_status,
                   float *__restrict__ dbgH, float *__restrict__ dbgG, float *__restrict__ g_force_wbc, int force_stride,
                   long long *__restrict__ dbgT)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int n = P.n;
    int rid;
    if (P.rescue_mode) {
        if (P.lpt_order_out && blockIdx.x < 8) {       // the histogram borrows the head of the dynamic LDS before the solve carves it
            extern __shared__ double smem_head[];
            lpt_order_chunk(blockIdx.x, n, P.lpt_cost_in, P.lpt_order_out, (int *)smem_head);
            __syncthreads();
        }
        // rescue pass: workgroup b re-solves the b-th robot the main pass could not hold
        int cnt = P.rescue_count[P.rescue_parity];
        cnt = cnt < n ? cnt : n;
        if ((int)blockIdx.x >= cnt) return;
        rid = P.rescue_list[blockIdx.x];
    } else {
        const int slot = xcd_robot_index(blockIdx.x, n);
        if (slot < 0) return;
        rid = P.order ? P.order[slot] : slot;       // same XCD chunk either way (the order permutes inside a chunk)
        if (P.rescue_count && blockIdx.x == 0 && tid == 0) P.rescue_count[P.rescue_parity ^ 1] = 0;     // the next call's counter
    }
    const long long t_begin = P.cost ? clock64() : 0;
    // a type id outside the table, or one that was never set up, would read garbage (mass 0 => 1/mass = inf): the robot is solved with the
    // first valid type's constants and carries QRGPU_ST_BAD_TYPE
    int tyid = type_id ? type_id[rid] : 0;
    const bool bad_type = tyid < 0 || tyid >= QR_MAX_TYPES || !((P.type_ready >> (tyid & (QR_MAX_TYPES - 1))) & 1);
    if (bad_type) tyid = __builtin_ctz(P.type_ready | (1 << QR_MAX_TYPES));
    const MpcType &C = P.type[tyid & (QR_MAX_TYPES - 1)];
    const int h = P.horizon;
    const int NV = 12 * h, NL = 4 * h;

    // ---------------- LDS carve (must match mpc_lds_fixed_bytes) ----------------
    extern __shared__ double smem[];
    double *gl = smem;                 // [NV] gradient (free variables, leg-step major)
    double *wl = gl + NV;              // single-wave: [NV] staging of w          | four-wave: xz[4][NV] partial x / z exchange
    double *yl = wl + NV;              // single-wave: [NV] staging of y = N r
    double *rl = yl + NV;              // single-wave: [QH] staging of r
    double *xz = gl + NV;
    double *xr = xz + 4 * NV;          // four-wave: xr[4][64] partial r exchange
    double *fmk = MULTI ? xr + 4 * 64 : rl + QR_QH;   // [NL] f_z upper bound per free leg-step
    float *sT = (float *)(fmk + NL);   // [4][3][3]
    float *sU = sT + 36;               // [4][3][3]
    float *sSt = sU + 36;              // [28]
    float *sTraj = sSt + 28;           // [12h]
    float *sGait = sTraj + NV;         // [4h]
    float *sV = sGait + NL;            // [13h]
    float *sJ = (MAXB <= 4) ? sV + 13 * h : nullptr;   // [12][3] columns of the leg Jacobians (the torque map of phase 6, computed while the data loads; h <= 11)
    int *sLs = (int *)(sV + 13 * h + (MAXB <= 4 ? 36 : 0));   // [NL] free leg-step -> original leg-step
    int *sAct = sLs + NL;              // [QH] active constraint ids (6*k + t)
    short *sPos = (short *)(sAct + QR_QH);   // [6 NL] constraint id -> position in sAct, or -1
    int *sMisc = (int *)(sPos + 6 * NL + ((6 * NL) & 1));   // [16]: [0] free leg-steps, [8..13] control block
    // (pointer arithmetic only: an integer round trip would drop the LDS address space and turn every access into flat_*)
    double *Mb = smem + (int)(mpc_lds_fixed_bytes(h, MULTI) / 8);   // block-packed M; the sweep panels live here first

#ifdef QR_TRACE
#define QR_TS(i) do { } while (0)
#else
#define QR_TS(i) do { if (dbgT && tid == 0) dbgT[(size_t)rid * 16 + (i)] = clock64(); } while (0)
#endif
    QR_TS(0);
    // ---------------- phase 0: inputs ----------------
    if (tid < 28) sSt[tid] = g_state[(size_t)tid * n + rid];
    for (int i = tid; i < NV; i += QR_MPC_THREADS) sTraj[i] = g_traj[(size_t)i * n + rid];
    for (int i = tid; i < NL; i += QR_MPC_THREADS) sGait[i] = g_gait[(size_t)i * n + rid];
    __syncthreads();

    // ---------------- phase 1: SRBD terms (every thread keeps R in registers) ----------------
    float R[3][3];
    {
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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
    QR_TS(1);
    // LDS loads count as divergent for the compiler; make the sizes scalar so that loops and branches on them are SALU
    const int nls = __builtin_amdgcn_readfirstlane(sMisc[0]);
    const int ns = 3 * nls;
    const int npairs = tri(nls);
    double *Sinv = Mb + npairs * 9;
    int qcap;
    {   // rows of S^-1 that fit behind M
        const long long rem = (long long)(P.lds_bytes / 8) - (long long)(Mb - smem) - (long long)npairs * 9;
        int qc = 0;
        if (rem > 0) { qc = (int)((__builtin_sqrt(8.0 * (double)rem + 1.0) - 1.0) * 0.5); while (tri(qc) > rem) --qc; }
        qcap = qc < QR_QH ? qc : QR_QH;
        if (qcap > ns) qcap = ns;
    }
    bool spilled = false;
    if constexpr (MAXB > 4) {
        // (the pointer is then generic and the S^-1 accesses of these variants compile to flat_* instructions: a few per cent at
        // h = 16, nothing at h <= 11 whose variants never take this branch)
        const int want = ns < QR_QH ? ns : QR_QH;
        if (P.sinv_spill && qcap < want && qcap < 64) { Sinv = P.sinv_spill + (size_t)rid * (size_t)tri(QR_QH); qcap = want; spilled = true; }
    }
    // Four-wave path: what is left behind S^-1 caches W_A = M N_A, one 3*nls vector per working-set position (the `w` of the
    // iteration that added it), so that z = w - W_A r needs no block products.  qW positions fit; the solve falls back to
    // z = w - M (N_A r) for good once the working set outgrows them (all-stance robots at h = 10 never have room).
    // h = 16 only: a four-wave solve whose working set reaches its 64 lanes hands over, in place, to the single-wave loop further down
    // (positions 64..95 in a second register / the big-q path): the state is a consistent dual-feasible point there (see the hand-over)
    bool handoff = false;
    double h_x0 = 0.0, h_x1 = 0.0, h_x2 = 0.0, h_u0 = 0.0;
    unsigned h_amask = 0;
    int h_q = 0, h_iter = 0;
    const int qcap_full = qcap;
    int qW = 0;
    double *Wc = nullptr;
    const int nsp = ns | 1;             // row stride of the cache (odd number of doubles)
    if constexpr (MULTI) {
        const int qs = spilled ? 0 : (qcap < 64 ? qcap : 64);
        const long long rem = (long long)(P.lds_bytes / 8) - (long long)(Mb - smem) - (long long)npairs * 9 - (long long)tri(qs);
        Wc = Mb + npairs * 9 + tri(qs);
        if (rem > 0 && ns > 0) qW = (int)(rem / (ns | 1));
        if (qW > 64) qW = 64;
        if (P.no_wcache == 1) qW = 0;
    }
    int st = bad_type ? QRGPU_ST_BAD_TYPE_D : 0;
    if (npairs > MAXB * QR_MPC_THREADS) { st |= QRGPU_ST_MPC_OVERFLOW_D; }      // cannot happen: the host picks MAXB from the horizon

    float w2[13];
#pragma unroll
    for (int s = 0; s < 12; ++s) w2[s] = 2.f * C.weights[s];
    w2[12] = 0.f;
    const float dtm = dt * minv;
    const float two_alpha = 2.f * C.alpha;

    // ---------------- phase 2: Hessian blocks (registers) + gradient (LDS) ----------------
    // (the torque map's Jacobian columns first, on twelve lanes of the last wave: it owns the fewest blocks, so this hides behind wave 0's)
    if (MAXB <= 4 && g_tau && tid >= 192 && tid < 204) { const int e = tid - 192; mpc_jacobian_column(e / 3, e - 3 * (e / 3), rid, n, C, g_q, sJ + 3 * e); }
    int ba[MAXB], bb[MAXB];
#pragma unroll
    for (int sl = 0; sl < MAXB; ++sl) {
        const int pid = tid + QR_MPC_THREADS * sl;
        ba[sl] = -1; bb[sl] = -1;
        if (pid < npairs) {
            int a = (int)((__builtin_sqrtf(8.f * (float)pid + 1.f) - 1.f) * 0.5f);
            while (tri(a + 1) <= pid) ++a;
            while (tri(a) > pid) --a;
            const int b = pid - tri(a);                 // a >= b
            ba[sl] = a; bb[sl] = b;
            const Blk Hb = hess_block(sT, sU, sLs[a], sLs[b], h, dt, dt2, minv, C.weights, C.alpha,
                                      dbgH ? dbgH + (size_t)rid * NV * NV : nullptr, NV);
            // parked in its final M slot: keeps the 18 VGPRs per block out of the build's register budget
            double *dst = Mb + pid * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) dst[i] = Hb.m[i];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the blocks' temporaries from overlapping (register pressure)
    }
    // gradient: qg[a] = sum_k temp[a][k] v[k], one free variable per thread
    for (int e = tid; e < ns; e += QR_MPC_THREADS) {
#pragma clang fp contract(off)
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
        gl[e] = (double)acc;
        if (dbgG) dbgG[(size_t)rid * NV + 3 * ls + j] = acc;
    }
    for (int c = tid; c < 6 * nls; c += QR_MPC_THREADS) sPos[c] = -1;
    QR_TS(2);

// Two implementations of the sweep.  The default keeps the 3x3 blocks in VALU registers.  QR_SWEEP_MFMA=1 runs the same sweep on the
// fp64 matrix cores (one v_mfma_f64_16x16x4_f64 per 16x16 tile and pivot); parity-green, but measured SLOWER on MI355X (1024 A1
// robots, h = 10: 128 k cycles per robot against 84 k): the f64 matrix instruction holds its wave for 64 cycles and runs at 1.7x the
// v_fma_f64 rate at best (scratch/ubench/mfma64b.hip), while copying the pivot columns out of the accumulator layout, the barrier and
// P^-1 cost ~3 k cycles per pivot whatever the matrix size -- the sweep is a chain of 4h dependent rank-3 steps, not a GEMM.
#ifndef QR_SWEEP_MFMA
#define QR_SWEEP_MFMA 0
#endif
#if QR_SWEEP_MFMA
    // ---------------- phase 3: symmetric sweep on the fp64 matrix cores,  A <- -H^-1 ----------------
    // Pivot leg-step k, its three columns p:  P = A_pp,  C = A[:, p] (n x 3, the lower triangle mirrored),  D = C P^-1;
    //   A <- A - D C'   everywhere,   then   A[:, p] <- D,  A[p, :] <- D',  A_pp <- -P^-1.
    // The whole step is ONE rank-3 update: with V = C except V[p, :] = P - I,
    //   A - (V P^-1) V'  =  A - D C' off the pivot,  D in the pivot columns / rows,  2I - P^-1 in the pivot block
    // (the 2I is taken off the three diagonal entries when they are copied out).
    // A lives in the accumulators of v_mfma_f64_16x16x4_f64 as 16x16 tiles of the lower triangle (diagonal tiles whole): five leg-steps
    // per tile row / column (15 of 16 indices; no leg-step straddles a tile), tile p = tri(R) + C in slot p / 4 of wave p mod 4.  Lane l
    // of a tile holds column l&15, rows (l>>4) + 4r, r = 0..3.  The update of a tile is one instruction: A operand -(V P^-1) rows of
    // tile row R, B operand V rows of tile column C, the fourth k zero.  Only V goes through LDS (rows of 4 doubles, double-buffered, one
    // barrier per pivot); the exact pivot diagonal goes to a side array for P^-1 (P_ii - 1 would lose the low bits of a small pivot
    // there; as an operand it does not matter).
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        constexpr int NTW = (MAXB <= 4) ? 12 : 24;         // tiles per wave: tri(9) = 45 (44 leg-steps), tri(13) = 91 (64 leg-steps)
        const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lc = lane & 15, lr = lane >> 4;
        const int T = (nls + 4) / 5, ntile = tri(T);
        const int nsl = (ntile - wvs + 3) >> 2;              // tiles of this wave
        int tRC[NTW];                                        // R | C << 8 (scalar)
        int offA[NTW], offB[NTW];                            // byte offsets of the lane's operands inside a panel
        d4 acc[NTW];
        __syncthreads();               // the Hessian blocks parked in the M slots are read by other threads now
        const int lcb = lc / 3, lcj = lc - 3 * lcb;
#pragma unroll
        for (int sl = 0; sl < NTW; ++sl) {
            const int pt = 4 * sl + wvs;
            int R = 0, C = 0;
            if (pt < ntile) { while (tri(R + 1) <= pt) ++R; C = pt - tri(R); }
            tRC[sl] = R | (C << 8);
            offA[sl] = (16 * R + lc) * 32; offB[sl] = ((16 * C + lc) * 4 + lr) * 8;
            acc[sl] = (d4){0.0, 0.0, 0.0, 0.0};
            if (pt < ntile) {
                const int b = 5 * C + lcb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ri = lr + 4 * r, a = 5 * R + ri / 3, i = ri - 3 * (ri / 3);
                    if (ri < 15 && lc < 15 && a < nls && b < nls)
                        acc[sl][r] = (a >= b) ? Mb[(tri(a) + b) * 9 + 3 * i + lcj] : Mb[(tri(b) + a) * 9 + 3 * lcj + i];
                }
            }
        }
        __syncthreads();               // every tile is in registers: the M region can now carry the panels
        double *pan0 = Mb + ((int)(Mb - smem) & 1);          // 16-byte aligned rows of 4 doubles: V[row][0..2] and a zero (the fourth k)
        const int pansz = 64 * T;
        double *pd0 = pan0 + 2 * pansz;                      // [2][4] exact pivot diagonals
        for (int e = tid; e < 32 * T; e += QR_MPC_THREADS) pan0[4 * e + 3] = 0.0;
#ifdef QR_SWEEP_STAMPS
        long long sw_t[6] = {0, 0, 0, 0, 0, 0}, sw_0 = clock64();
#define SW_STAMP(i) do { const long long t_ = clock64(); sw_t[i] += t_ - sw_0; sw_0 = t_; } while (0)
#else
#define SW_STAMP(i) do { } while (0)
#endif
        unsigned colmask = 0, rowmask = 0;                   // slots holding a tile of the pivot's tile column / tile row
        int J = -1;
        for (int k = 0; k < nls; ++k) {
            double *pan = pan0 + (k & 1) * pansz, *pd = pd0 + (k & 1) * 4;
            SW_STAMP(3);
            if (k == 5 * (J + 1)) {
                ++J; colmask = 0; rowmask = 0;
#pragma unroll
                for (int sl = 0; sl < NTW; ++sl) {
                    if (sl >= nsl) continue;
                    if ((tRC[sl] >> 8) == J) colmask |= 1u << sl;
                    if ((tRC[sl] & 255) == J) rowmask |= 1u << sl;
                }
            }
            const int y = 3 * (k - 5 * J);                   // first pivot index inside its tile
            SW_STAMP(4);
            // ---- the pivot columns -> panel (from the lower triangle only, so that the panel is exactly symmetric data)
            const int jj = lc - y;
#pragma unroll
            for (int sl = 0; sl < NTW; ++sl) {
                if ((colmask >> sl) & 1u) {
                    const int R = tRC[sl] & 255;
                    if (jj >= 0 && jj < 3) {
                        double *dst = pan + (16 * R + lr) * 4 + jj;
                        if (R > J) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) dst[16 * r] = acc[sl][r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (lr + 4 * r == lc) { pd[jj] = acc[sl][r]; dst[16 * r] = acc[sl][r] - 1.0; acc[sl][r] -= 2.0; }
                                else if (lr + 4 * r > lc) dst[16 * r] = acc[sl][r];
                            }
                        }
                    }
                }
                if ((rowmask >> sl) & 1u) {
                    // row y + ii of this tile sits in register (y + ii - lr) / 4 of the lanes whose lr makes that an integer
                    const int C = tRC[sl] >> 8, ii = (lr - y) & 3, x = y + ii - lr;
                    const double v01 = (x >> 2) == 0 ? acc[sl][0] : acc[sl][1], v23 = (x >> 2) == 2 ? acc[sl][2] : acc[sl][3];
                    const double v = (x >> 2) < 2 ? v01 : v23;
                    if (ii < 3 && (C < J || lc < y + ii)) pan[(16 * C + lc) * 4 + ii] = v;
                }
            }
            SW_STAMP(0);
            __syncthreads();
            SW_STAMP(1);
            // P^-1 (3x3 symmetric, adjugate / determinant), redundantly per lane
            double Pi[9];
            {
                const double *Pk = pan + 4 * (16 * J + y);
                const double p00 = pd[0], p01 = Pk[4], p02 = Pk[8], p11 = pd[1], p12 = Pk[9], p22 = pd[2];
                const double c00 = p11 * p22 - p12 * p12, c01 = p02 * p12 - p01 * p22, c02 = p01 * p12 - p02 * p11;
                const double c11 = p00 * p22 - p02 * p02, c12 = p01 * p02 - p00 * p12, c22 = p00 * p11 - p01 * p01;
                const double det = p00 * c00 + p01 * c01 + p02 * c02;
                if (!(det > 0.0) || !(p00 > 0.0)) st |= QRGPU_ST_MPC_NOTSPD_D;
                const double id = fast_rcp(det);
                Pi[0] = c00 * id; Pi[1] = c01 * id; Pi[2] = c02 * id;
                Pi[3] = Pi[1];    Pi[4] = c11 * id; Pi[5] = c12 * id;
                Pi[6] = Pi[2];    Pi[7] = Pi[5];    Pi[8] = c22 * id;
            }
            // column lr of -P^-1 for the A operand (k index = lr; the fourth k is zero padding)
            const double q0 = -(lr == 0 ? Pi[0] : (lr == 1 ? Pi[1] : (lr == 2 ? Pi[2] : 0.0)));
            const double q1 = -(lr == 0 ? Pi[3] : (lr == 1 ? Pi[4] : (lr == 2 ? Pi[5] : 0.0)));
            const double q2 = -(lr == 0 ? Pi[6] : (lr == 1 ? Pi[7] : (lr == 2 ? Pi[8] : 0.0)));
#ifdef QR_SWEEP_STAMPS
            asm volatile("" :: "v"(q0), "v"(q1), "v"(q2));
#endif
            SW_STAMP(2);
            // three tiles at a time: the operand loads first, then the products and the matrix instructions
            // (a group's slots past the wave's last tile work on tile (0, 0) data and are never stored)
            const char *panb = (const char *)pan;
#pragma unroll
            for (int g = 0; g < NTW / 3; ++g) {
                if (3 * g >= nsl) continue;
                double c0[3], c1[3], c2[3], bo[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const double *cr = (const double *)(panb + offA[3 * g + u]);
                    c0[u] = cr[0]; c1[u] = cr[1]; c2[u] = cr[2];
                    bo[u] = *(const double *)(panb + offB[3 * g + u]);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const double aop = c0[u] * q0 + c1[u] * q1 + c2[u] * q2;
                    acc[3 * g + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bo[u], acc[3 * g + u], 0, 0, 0);
                }
            }
        }
#ifdef QR_SWEEP_STAMPS
        SW_STAMP(3);
        if (dbgT && tid == 0) { dbgT[(size_t)rid * 16 + 8] = sw_t[0]; dbgT[(size_t)rid * 16 + 9] = sw_t[1]; dbgT[(size_t)rid * 16 + 10] = sw_t[2]; dbgT[(size_t)rid * 16 + 11] = sw_t[3]; dbgT[(size_t)rid * 16 + 12] = sw_t[4]; }
#endif
        __syncthreads();           // everybody is done with the panels before M overwrites them
#pragma unroll
        for (int sl = 0; sl < NTW; ++sl) {
            if (sl >= nsl) continue;
            const int R = tRC[sl] & 255, C = tRC[sl] >> 8, b = 5 * C + lcb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ri = lr + 4 * r, a = 5 * R + ri / 3, i = ri - 3 * (ri / 3);
                if (ri < 15 && lc < 15 && a < nls && (R > C || lc <= ri)) {
                    const double v = -acc[sl][r];                              // M = +H^-1
                    Mb[(tri(a) + b) * 9 + 3 * i + lcj] = v;
                    if (a == b && i != lcj) Mb[(tri(a) + a) * 9 + 3 * lcj + i] = v;
                }
            }
        }
        __syncthreads();
    }
#else
    // ---------------- phase 3: symmetric block sweep in registers,  A <- -H^-1 ----------------
    // Pivot leg-step k:  P = A_kk,  C_i = A_ik (i > k) or A_ki' (i < k);
    //   A_ij <- A_ij - C_i P^-1 C_j',   A_ik <- C_i P^-1,   A_kk <- -P^-1.
    // The pivot column is exchanged through a double-buffered LDS panel: one barrier per pivot.
    Blk A[MAXB];
#pragma unroll
    for (int sl = 0; sl < MAXB; ++sl) {
        if (ba[sl] >= 0) {
            const double *src = Mb + (tid + QR_MPC_THREADS * sl) * 9;        // own slot: no barrier needed
#pragma unroll
            for (int i = 0; i < 9; ++i) A[sl].m[i] = src[i];
        }
    }
    __syncthreads();               // every block is in registers: the M region can now carry the pivot panels
    {
        double *panel0 = Mb, *panel1 = Mb + NL * 9;
#ifdef QR_SWEEP_STAMPS
        long long vs_t[4] = {0, 0, 0, 0}, vs_0 = clock64();
#define VS_STAMP(i) do { const long long t_ = clock64(); vs_t[i] += t_ - vs_0; vs_0 = t_; } while (0)
#else
#define VS_STAMP(i) do { } while (0)
#endif
        // the pivot column of step kk out of block sl: block (a, kk), a >= kk, is C_a; block (kk, b), b < kk, is C_b'
        auto write_panel = [&](int sl, int kk, double *pn) {
            if (bb[sl] == kk) {
#pragma unroll
                for (int i = 0; i < 9; ++i) pn[9 * ba[sl] + i] = A[sl].m[i];
            } else if (ba[sl] == kk) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) pn[9 * bb[sl] + 3 * j + i] = A[sl].m[3 * i + j];
            }
        };
#pragma unroll
        for (int sl = 0; sl < MAXB; ++sl) write_panel(sl, 0, panel0);
        for (int k = 0; k < nls; ++k) {
            double *pan = (k & 1) ? panel1 : panel0, *pnext = (k & 1) ? panel0 : panel1;
            VS_STAMP(3);
            VS_STAMP(0);
            __syncthreads();
            VS_STAMP(1);
            // the first block's operands do not depend on P^-1: their LDS round trip hides behind its computation
            double Cx0[9], Cb0[9];
            {
                const int a0 = ba[0] < 0 ? 0 : ba[0], b0 = ba[0] < 0 ? 0 : bb[0];
                const double *Cx = pan + 9 * (a0 == k ? b0 : a0), *Cb = pan + 9 * b0;
#pragma unroll
                for (int i = 0; i < 9; ++i) { Cx0[i] = Cx[i]; Cb0[i] = Cb[i]; }
            }
            // P^-1 (3x3 symmetric, adjugate / determinant), redundantly per thread
            double Pi[9];
            {
                const double *Pk = pan + 9 * k;
                const double p00 = Pk[0], p01 = 0.5 * (Pk[1] + Pk[3]), p02 = 0.5 * (Pk[2] + Pk[6]), p11 = Pk[4], p12 = 0.5 * (Pk[5] + Pk[7]), p22 = Pk[8];
                const double c00 = p11 * p22 - p12 * p12, c01 = p02 * p12 - p01 * p22, c02 = p01 * p12 - p02 * p11;
                const double c11 = p00 * p22 - p02 * p02, c12 = p01 * p02 - p00 * p12, c22 = p00 * p11 - p01 * p01;
                const double det = p00 * c00 + p01 * c01 + p02 * c02;
                if (!(det > 0.0) || !(p00 > 0.0)) st |= QRGPU_ST_MPC_NOTSPD_D;
                const double id = fast_rcp(det);
                Pi[0] = c00 * id; Pi[1] = c01 * id; Pi[2] = c02 * id;
                Pi[3] = Pi[1];    Pi[4] = c11 * id; Pi[5] = c12 * id;
                Pi[6] = Pi[2];    Pi[7] = Pi[5];    Pi[8] = c22 * id;
            }
#ifdef QR_SWEEP_STAMPS
            asm volatile("" :: "v"(Pi[0]), "v"(Pi[4]), "v"(Pi[8]), "v"(Pi[5]));
#endif
            VS_STAMP(2);
#pragma unroll
            for (int sl = 0; sl < MAXB; ++sl) {
                const int a = ba[sl], b = bb[sl];
                if (a < 0) continue;
                // D = C_x P^-1, x = the non-pivot index of the block (any index for the pivot block itself)
                double Cxv[9], Cb[9];
                if (sl == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) { Cxv[i] = Cx0[i]; Cb[i] = Cb0[i]; }
                } else {
                    const double *Cxp = pan + 9 * (a == k ? b : a), *Cbp = pan + 9 * b;
#pragma unroll
                    for (int i = 0; i < 9; ++i) { Cxv[i] = Cxp[i]; Cb[i] = Cbp[i]; }
                }
                const double *Cx = Cxv;
                double D[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) D[3 * i + j] = Cx[3 * i] * Pi[j] + Cx[3 * i + 1] * Pi[3 + j] + Cx[3 * i + 2] * Pi[6 + j];
                if (a != k && b != k) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            A[sl].m[3 * i + j] -= D[3 * i] * Cb[3 * j] + D[3 * i + 1] * Cb[3 * j + 1] + D[3 * i + 2] * Cb[3 * j + 2];
                } else {
                    // pivot row / column: A_xk = D (x > k), A_kx = D' (x < k), A_kk = -P^-1
                    const bool piv = (a == k && b == k), tr = (a == k);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) A[sl].m[3 * i + j] = piv ? -Pi[3 * i + j] : (tr ? D[3 * j + i] : D[3 * i + j]);
                }
                // this block is final for step k: if it lies in the next pivot's column, publish it now (the other panel: its readers
                // passed this step's barrier), so the copy overlaps the remaining blocks instead of being a phase of its own
                write_panel(sl, k + 1, pnext);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef QR_SWEEP_STAMPS
        VS_STAMP(3);
        if (dbgT && tid == 0) { dbgT[(size_t)rid * 16 + 8] = vs_t[0]; dbgT[(size_t)rid * 16 + 9] = vs_t[1]; dbgT[(size_t)rid * 16 + 10] = vs_t[2]; dbgT[(size_t)rid * 16 + 11] = vs_t[3]; dbgT[(size_t)rid * 16 + 12] = 0; }
#endif
        __syncthreads();           // everybody is done with the panels before M overwrites them
#pragma unroll
        for (int sl = 0; sl < MAXB; ++sl) {
            if (ba[sl] >= 0) {
                double *dst = Mb + (tri(ba[sl]) + bb[sl]) * 9;
#pragma unroll
                for (int i = 0; i < 9; ++i) dst[i] = -A[sl].m[i];         // M = +H^-1
            }
        }
        __syncthreads();
    }
#endif
    QR_TS(3);
    // =====================================================================================================
    // Four-wave active set (h <= 11).  All four wavefronts run the SAME control flow on the SAME data, so every
    // decision is identical without communication (same code, same inputs, deterministic reductions); only the
    // three loops whose cost grows with the working set are split four ways and recombined through LDS:
    //   r = S^-1 d        wave v takes columns j = v (mod 4)      -> partial r   -> xr[4][64]  -> barrier -> sum
    //   z = w - M (N r)   wave v takes every 4th active leg-step  -> partial z   -> xz[4][NV]  -> barrier -> sum
    //   S^-1 +/- update   wave v updates columns j = v (mod 4) of every row (disjoint), visible after the next barrier
    // Bookkeeping lives in registers, replicated per wave: working-set position i belongs to lane i (q <= 64):
    // constraint (ck, ct), multiplier u, and d, r during an iteration; lane k knows its leg-step's active rows (amask)
    // and their positions (posk).  Per-lane gathers use ds_bpermute (__shfl), uniform ones v_readlane.
    // =====================================================================================================
    // =====================================================================================================
    // Control / worker active set (QR_GI_CTRL, default).  Wave 0 alone takes the decisions -- row scan, w, delta, d, step lengths,
    // bookkeeping in its registers -- and waves 1-3 are linear-algebra helpers, so that the S^-1 border of one iteration runs while
    // wave 0 already scans for the next row, and the z partials run while wave 0 reduces the step lengths:
    //     wave 0:   scan, w, delta, d  ->X1-> r partial ->B2-> r, dr, t1, t2, t ->B3-> z, x, u, bookkeeping -> scan ...
    //     wave 1-3: [S^-1 border / downdate of the previous iteration] ->X1-> r partial ->B2-> r, z partial ->B3-> S^-1 border ...
    // Three workgroup barriers per working-set change (X1 also publishes the S^-1 update); the control block in LDS carries
    // {command, q, flags, dropped position, 1/z'c}, d travels through xz[0], the row positions of every leg-step through sPos.
    // =====================================================================================================
#ifndef QR_GI_CTRL
#define QR_GI_CTRL 1
#endif
    if constexpr (MULTI && QR_GI_CTRL) {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool own = lane < nls;
        const int kme = own ? lane : 0;
        const double im = (double)(1.f / C.mu);
        const double fmaxk = own ? fmk[kme] : 0.0;
        const int tril = tri(lane);
        const double tol = 1e-9;
        const double INF = __builtin_inf();
        // h = 16 variants: working-set positions 64 .. 95 live in a SECOND set of per-lane registers (position lane + 64), so a solve that
        // outgrows the 64 lanes stays in this loop instead of handing over to the single-wave one (which costs ~40 k cycles per change at
        // 86 rows).  The second half of the r exchange lives in the float arrays of phases 0-2, dead by now (2.2 KB at h = 16).
        constexpr bool BIG = MAXB > 4;
        const bool big2 = BIG && (100 + 29 * h) * 4 >= 2048 && qcap_full > 64;
        double *xr2 = (double *)sT;
        const int tril2 = tri(lane + 64);
        if (qcap > (big2 ? QR_QH : 64)) qcap = big2 ? QR_QH : 64;
        auto rd2 = [&](double v0, double v1, int j) { return j < 64 ? readlane_d(v0, j) : readlane_d(v1, j - 64); };       // j uniform
        int *sCtl = sMisc + 8;                            // [0] command (0 go, 1 exit), [1] q, [2] flags, [3] dropped position, [4,5] 1/z'c
        double *dd = xz;                                  // d of the iteration (wave 0 produces no z partial: its slot is free)
        enum { F_FULL = 1, F_DROP = 2 };
        // ---- phase 4: x = -M g, block columns kc = wv (mod 4) per wave
        double x0 = 0.0, x1 = 0.0, x2 = 0.0;
        {
            double p0 = 0.0, p1 = 0.0, p2 = 0.0;
            if (own) {
                for (int kc = wv; kc < nls; kc += 4) {
                    Blk B; load_block(Mb, kme, kc, B);
                    const double g0 = gl[3 * kc], g1 = gl[3 * kc + 1], g2 = gl[3 * kc + 2];
                    p0 += B.m[0] * g0 + B.m[1] * g1 + B.m[2] * g2;
                    p1 += B.m[3] * g0 + B.m[4] * g1 + B.m[5] * g2;
                    p2 += B.m[6] * g0 + B.m[7] * g1 + B.m[8] * g2;
                }
                xz[wv * NV + 3 * kme] = p0; xz[wv * NV + 3 * kme + 1] = p1; xz[wv * NV + 3 * kme + 2] = p2;
            }
            if (wv == 0) for (int e = lane; e < 6 * nls; e += 64) sPos[e] = (short)-1;
            __syncthreads();
            if (own) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { x0 -= xz[v * NV + 3 * kme]; x1 -= xz[v * NV + 3 * kme + 1]; x2 -= xz[v * NV + 3 * kme + 2]; }
            }
            __syncthreads();
        }
        QR_TS(4);
#ifdef QR_CTRL_NOFASTZ
        bool fastz = false;
#else
        bool fastz = qW > 0;
#endif
        // r partial over columns j = wv (mod 4); (i, j) at tri(i) + j for j <= i, else tri(j) + i
        auto r_partial = [&](int q, double dq, double dq2) {
            if (BIG && q > 64) {
                const int i1 = (lane + 64 < q) ? lane + 64 : 0;
                double pa = 0.0, pb = 0.0;
                for (int j = wv; j < q; j += 4) {
                    const double dj = rd2(dq, dq2, j);
                    pa += Sinv[(j <= lane) ? tril + j : tri(j) + lane] * dj;
                    pb += Sinv[(j <= i1) ? tri(i1) + j : tri(j) + i1] * dj;
                }
                xr[wv * 64 + lane] = pa; xr2[wv * 64 + lane] = pb;
                return;
            }
            const int i0 = (lane < q) ? lane : 0;
            double pr = 0.0;
            int j = wv;
            for (; j + 12 < q; j += 16) {
                double sv[4], dj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int jj = j + 4 * u; sv[u] = Sinv[(jj <= i0) ? tril + jj : tri(jj) + i0]; dj[u] = readlane_d(dq, jj); }
#pragma unroll
                for (int u = 0; u < 4; ++u) pr += sv[u] * dj[u];
            }
            for (; j < q; j += 4) pr += Sinv[(j <= i0) ? tril + j : tri(j) + i0] * readlane_d(dq, j);
            xr[wv * 64 + lane] = pr;
        };
        if (wv != 0) {
            // ================================ workers ================================
            const int g = wv - 1;                         // 0..2
            for (;;) {
                __syncthreads();                          // X1
                if (__builtin_amdgcn_readfirstlane(sCtl[0]) != 0) return;
                const int q = __builtin_amdgcn_readfirstlane(sCtl[1]);
                const bool hi = BIG && q > 64;
                const double dq = (lane < q) ? dd[lane] : 0.0;
                const double dq2 = (hi && lane + 64 < q) ? dd[lane + 64] : 0.0;
                r_partial(q, dq, dq2);
                __syncthreads();                          // B2
                double rq, rq2 = 0.0;
                { const double rs = (xr[lane] + xr[64 + lane]) + (xr[128 + lane] + xr[192 + lane]); rq = (lane < q) ? rs : 0.0; }
                if (hi) { const double rs = (xr2[lane] + xr2[64 + lane]) + (xr2[128 + lane] + xr2[192 + lane]); rq2 = (lane + 64 < q) ? rs : 0.0; }
                // z partial: W_A r over the positions i = g (mod 3), or M (N_A r) over every third active leg-step
                double p0 = 0.0, p1 = 0.0, p2 = 0.0;
                if (fastz) {
                    const double *wk = Wc + 3 * kme;
                    int i = g;
                    for (; i + 3 < q; i += 6) {
                        const double ra = readlane_d(rq, i), rb = readlane_d(rq, i + 3);
                        const double *wa = wk + i * nsp, *wb = wk + (i + 3) * nsp;
                        const double a0 = wa[0], a1 = wa[1], a2 = wa[2], b0 = wb[0], b1 = wb[1], b2 = wb[2];
                        p0 += a0 * ra + b0 * rb; p1 += a1 * ra + b1 * rb; p2 += a2 * ra + b2 * rb;
                    }
                    if (i < q) { const double ra = readlane_d(rq, i); const double *wa = wk + i * nsp; p0 += wa[0] * ra; p1 += wa[1] * ra; p2 += wa[2] * ra; }
                } else if (q > 0) {
                    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
                    bool hasrow = false;
#pragma unroll
                    for (int tq = 0; tq < 6; ++tq) {
                        const int ps = own ? (int)sPos[6 * kme + tq] : -1;
                        double rr = __shfl(rq, ps < 0 ? 0 : (ps & 63), 64);
                        if (hi) { const double r2 = __shfl(rq2, ps < 0 ? 0 : (ps & 63), 64); rr = ps >= 64 ? r2 : rr; }
                        if (ps >= 0) { double a0, a1, a2; cons_vec(tq, im, a0, a1, a2); y0 += a0 * rr; y1 += a1 * rr; y2 += a2 * rr; hasrow = true; }
                    }
                    const unsigned long long kall = __ballot(hasrow);
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(kall >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kall, 0u));
                    const int r3 = rank - 3 * ((rank * 21846) >> 16);        // rank mod 3 for rank < 64
                    unsigned long long km = __ballot(hasrow && r3 == g);
                    while (km) {
                        const int kc = (int)__builtin_ctzll(km);
                        km &= km - 1;
                        Blk B; load_block(Mb, kme, kc, B);
                        const double q0 = readlane_d(y0, kc), q1 = readlane_d(y1, kc), q2 = readlane_d(y2, kc);
                        p0 += B.m[0] * q0 + B.m[1] * q1 + B.m[2] * q2;
                        p1 += B.m[3] * q0 + B.m[4] * q1 + B.m[5] * q2;
                        p2 += B.m[6] * q0 + B.m[7] * q1 + B.m[8] * q2;
                    }
                }
                if (own) { xz[wv * NV + 3 * kme] = p0; xz[wv * NV + 3 * kme + 1] = p1; xz[wv * NV + 3 * kme + 2] = p2; }
                __syncthreads();                          // B3
                const int flags = __builtin_amdgcn_readfirstlane(sCtl[2]);
                if (flags & F_FULL) {
                    // bordered update of S^-1, columns j = g (mod 3); published by the next X1
                    const double isg = __hiloint2double(__builtin_amdgcn_readfirstlane(sCtl[4]), __builtin_amdgcn_readfirstlane(sCtl[5]));
                    const bool act0 = lane < q;
                    const double ri = rq * isg;
                    int j = g;
                    if (hi) {
                        const bool act1 = lane + 64 < q;
                        const double ri2 = rq2 * isg;
                        for (; j < q; j += 3) {
                            const double rj = rd2(rq, rq2, j);
                            if (j <= lane) Sinv[tril + j] += ri * rj;
                            if (act1 && j <= lane + 64) Sinv[tril2 + j] += ri2 * rj;
                        }
                        if (g == 0) {
                            Sinv[tri(q) + lane] = -rq * isg;
                            if (act1) Sinv[tri(q) + 64 + lane] = -rq2 * isg;
                            if (lane == 0) Sinv[tri(q) + q] = isg;
                        }
                    } else {
                    for (; j + 9 < q; j += 12) {
                        double sv[4], rj[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int jj = j + 3 * u; rj[u] = readlane_d(rq, jj); sv[u] = (act0 && jj <= lane) ? Sinv[tril + jj] : 0.0; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int jj = j + 3 * u; if (act0 && jj <= lane) Sinv[tril + jj] = sv[u] + ri * rj[u]; }
                    }
                    for (; j < q; j += 3) { const double rj = readlane_d(rq, j); if (act0 && j <= lane) Sinv[tril + j] += ri * rj; }
                    if (g == 0) {
                        if (act0) Sinv[tri(q) + lane] = -rq * isg;
                        if (lane == 0) Sinv[tri(q) + q] = isg;
                    }
                    }
                    if (fastz && !(q < qW)) fastz = false;      // the same rule wave 0 applies when it stores the cache row
                } else if (flags & F_DROP) {
                    const int l = __builtin_amdgcn_readfirstlane(sCtl[3]), last = q - 1;
                    double sl = 0.0, sl2 = 0.0;
                    if (lane < q) sl = Sinv[pidx(lane, l)];
                    if (hi && lane + 64 < q) sl2 = Sinv[pidx(lane + 64, l)];
                    const double isl = fast_rcp(rd2(sl, sl2, l));
                    __syncthreads();                      // D1: everyone has column l before anyone changes S^-1
                    for (int j = g; j < q; j += 3) {
                        if (j == l) continue;
                        const double sj = rd2(sl, sl2, j) * isl;
                        if (lane < q && lane != l && j <= lane) Sinv[tril + j] -= sl * sj;
                        if (hi && lane + 64 < q && lane + 64 != l && j <= lane + 64) Sinv[tril2 + j] -= sl2 * sj;
                    }
                    __syncthreads();                      // D2
                    double m0 = 0.0, m1 = 0.0;
                    if (l != last && g == 0 && lane < last) m0 = (lane == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, lane)];
                    if (hi && l != last && g == 0 && lane + 64 < last) m1 = (lane + 64 == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, lane + 64)];
                    __syncthreads();                      // D3
                    if (l != last && g == 0 && lane < last) Sinv[pidx(l, lane)] = m0;
                    if (hi && l != last && g == 0 && lane + 64 < last) Sinv[pidx(l, lane + 64)] = m1;
                }
            }
        }
        // ================================ wave 0: control ================================
        int q = 0, iter = 0;
#ifdef QR_GI_STAMPS      // sub-phase cycle accounting of wave 0 (build.py: QRGPU_GI_STAMPS=1)
        long long cs_t[6] = {0, 0, 0, 0, 0, 0}, cs_0 = clock64();
#define CS_STAMP(i) do { const long long t_ = clock64(); cs_t[i] += t_ - cs_0; cs_0 = t_; } while (0)
#else
#define CS_STAMP(i) do { } while (0)
#endif
        unsigned amask = 0, xmask = 0;
        unsigned long long posk = 0;                      // byte t: working-set position of row t of my leg-step
        int ck = 0, ct = 0;                               // constraint (leg-step, row) at working-set position `lane`
        double uq = 0.0;                                  // its multiplier
        int ck2 = 0, ct2 = 0;                             // (h = 16 variants) the same for position lane + 64
        double uq2 = 0.0;
        const int maxit = 40 * nls + 100;
        bool done = (nls == 0);
        while (!done) {
            double bs = INF; int bt = 0;
            {
                const double s[6] = {im * x0 + x2, -im * x0 + x2, im * x1 + x2, -im * x1 + x2, x2, fmaxk - x2};
                const unsigned blocked = own ? (amask | xmask) : 0x3fu;
#pragma unroll
                for (int t = 0; t < 6; ++t) { const bool take = !((blocked >> t) & 1u) && s[t] < bs; bs = take ? s[t] : bs; bt = take ? t : bt; }
            }
            const double smin = wave_min_d(bs);
            if (!(smin < -tol)) {
                double be = 0.0;                          // excluded rows hold, active rows are tight (see the four-wave loop below)
                if (own && (xmask | amask)) {
                    const double s[6] = {im * x0 + x2, -im * x0 + x2, im * x1 + x2, -im * x1 + x2, x2, fmaxk - x2};
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        if (((xmask >> t) & 1u) && s[t] < be) be = s[t];
                        if (((amask >> t) & 1u) && -__builtin_fabs(s[t]) < be) be = -__builtin_fabs(s[t]);
                    }
                }
                if (wave_min_d(be) < -1e-4) st |= QRGPU_ST_MPC_INFEAS_D;
                break;
            }
            const int kp = __builtin_amdgcn_readfirstlane(first_lane(bs == smin));
            const int tp = __builtin_amdgcn_readlane(bt, kp);
            double c0, c1, c2;
            cons_vec(tp, im, c0, c1, c2);
            const double ci0p = (tp == 5) ? readlane_d(fmaxk, kp) : 0.0;
            double up = 0.0;
            CS_STAMP(0);
            for (;;) {
                q = __builtin_amdgcn_readfirstlane(q);
                if (++iter > maxit) { st |= QRGPU_ST_MPC_MAXITER_D; done = true; break; }
                double w0, w1, w2_;
                {
                    Blk B; load_block(Mb, kme, kp, B);
                    w0 = B.m[0] * c0 + B.m[1] * c1 + B.m[2] * c2;
                    w1 = B.m[3] * c0 + B.m[4] * c1 + B.m[5] * c2;
                    w2_ = B.m[6] * c0 + B.m[7] * c1 + B.m[8] * c2;
                }
                const double delta = c0 * readlane_d(w0, kp) + c1 * readlane_d(w1, kp) + c2 * readlane_d(w2_, kp);
                double dq;
                {
                    double a0, a1, a2;
                    cons_vec(ct, im, a0, a1, a2);
                    const double g0 = __shfl(w0, ck, 64), g1 = __shfl(w1, ck, 64), g2 = __shfl(w2_, ck, 64);
                    dq = (lane < q) ? a0 * g0 + a1 * g1 + a2 * g2 : 0.0;
                }
                const bool hi = BIG && q > 64;
                double dq2 = 0.0;
                if (hi) {
                    double a0, a1, a2;
                    cons_vec(ct2, im, a0, a1, a2);
                    const double g0 = __shfl(w0, ck2, 64), g1 = __shfl(w1, ck2, 64), g2 = __shfl(w2_, ck2, 64);
                    dq2 = (lane + 64 < q) ? a0 * g0 + a1 * g1 + a2 * g2 : 0.0;
                    if (lane + 64 < q) dd[lane + 64] = dq2;
                }
                if (lane < q) dd[lane] = dq;
                if (lane == 0) { sCtl[0] = 0; sCtl[1] = q; }
                CS_STAMP(1);
                __syncthreads();                          // X1
                CS_STAMP(2);
                r_partial(q, dq, dq2);
                __syncthreads();                          // B2
                CS_STAMP(3);
                double rq, rq2 = 0.0;
                { const double rs = (xr[lane] + xr[64 + lane]) + (xr[128 + lane] + xr[192 + lane]); rq = (lane < q) ? rs : 0.0; }
                if (hi) { const double rs = (xr2[lane] + xr2[64 + lane]) + (xr2[128 + lane] + xr2[192 + lane]); rq2 = (lane + 64 < q) ? rs : 0.0; }
                const double dr = wave_sum_d(hi ? rq * dq + rq2 * dq2 : rq * dq);
                const double zc = delta - dr;
                double tt, tt2 = INF;
                { const double tq_ = uq * fast_rcp(rq); tt = (lane < q && rq > 0.0) ? tq_ : INF; }
                if (hi) { const double tq_ = uq2 * fast_rcp(rq2); tt2 = (lane + 64 < q && rq2 > 0.0) ? tq_ : INF; }
                const double t1 = wave_min_d(hi ? (tt < tt2 ? tt : tt2) : tt);
                int lpos = (t1 < INF) ? first_lane(tt == t1) : -1;
                if (hi && t1 < INF && lpos < 0) lpos = 64 + first_lane(tt2 == t1);
                const double sp = c0 * readlane_d(x0, kp) + c1 * readlane_d(x1, kp) + c2 * readlane_d(x2, kp) + ci0p;
                const bool have_z = zc > 1e-13 * delta;
                const double izc = fast_rcp(zc);
                const double t2 = have_z ? -sp * izc : INF;
                const double t = t1 < t2 ? t1 : t2;
                const bool degenerate = !(t < INF);
                const bool full = !degenerate && have_z && t == t2;
                const bool over = full && q >= qcap;
                const int flags = (degenerate || over) ? 0 : (full ? F_FULL : F_DROP);
#ifdef QR_TRACE
                if (dbgT && lane == 0 && iter <= 7) { dbgT[(size_t)rid * 16 + 2 * (iter - 1)] = ((long long)kp << 32) | (tp << 24) | (q << 16) | (full ? 1 : 0) | (have_z ? 2 : 0); dbgT[(size_t)rid * 16 + 2 * (iter - 1) + 1] = __double_as_longlong(t); }
#endif
                if (lane == 0) { sCtl[2] = flags; sCtl[3] = lpos; sCtl[4] = __double2hiint(izc); sCtl[5] = __double2loint(izc); }
                CS_STAMP(4);
                __syncthreads();                          // B3
                CS_STAMP(5);
                if (degenerate) { if (lane == kp) xmask |= 1u << tp; break; }
                if (over) {
                    if (!big2 && q == 64 && qcap_full > 64) handoff = true; else st |= QRGPU_ST_MPC_OVERFLOW_D;
                    done = true; break;
                }
                if (have_z) {
                    const double z0 = w0 - ((xz[NV + 3 * kme] + xz[2 * NV + 3 * kme]) + xz[3 * NV + 3 * kme]);
                    const double z1 = w1 - ((xz[NV + 3 * kme + 1] + xz[2 * NV + 3 * kme + 1]) + xz[3 * NV + 3 * kme + 1]);
                    const double z2 = w2_ - ((xz[NV + 3 * kme + 2] + xz[2 * NV + 3 * kme + 2]) + xz[3 * NV + 3 * kme + 2]);
                    x0 += t * z0; x1 += t * z1; x2 += t * z2;
                }
                uq -= t * rq;
                if (hi) uq2 -= t * rq2;
                up += t;
                if (full) {
                    if (fastz) {
                        if (q < qW) { if (own) { double *wq = Wc + q * nsp + 3 * kme; wq[0] = w0; wq[1] = w1; wq[2] = w2_; } }
                        else fastz = false;
                    }
                    if (lane == q) { uq = up; ck = kp; ct = tp; }
                    if (BIG && lane + 64 == q) { uq2 = up; ck2 = kp; ct2 = tp; }
                    if (lane == kp) { amask |= 1u << tp; posk = (posk & ~(0xffull << (8 * tp))) | ((unsigned long long)q << (8 * tp)); sPos[6 * kp + tp] = (short)q; }
                    xmask = 0;
                    ++q;
                    break;
                }
                // partial or dual-only step: position lpos leaves (the workers downdate S^-1 between D1 and D3)
                {
                    const int l = lpos, last = q - 1;
                    int clk, clt, cmk, cmt;
                    double ulast;
                    if (BIG && l >= 64) { clk = __builtin_amdgcn_readlane(ck2, l - 64); clt = __builtin_amdgcn_readlane(ct2, l - 64); }
                    else { clk = __builtin_amdgcn_readlane(ck, l); clt = __builtin_amdgcn_readlane(ct, l); }
                    if (BIG && last >= 64) { cmk = __builtin_amdgcn_readlane(ck2, last - 64); cmt = __builtin_amdgcn_readlane(ct2, last - 64); ulast = readlane_d(uq2, last - 64); }
                    else { cmk = __builtin_amdgcn_readlane(ck, last); cmt = __builtin_amdgcn_readlane(ct, last); ulast = readlane_d(uq, last); }
                    __syncthreads();                      // D1
                    if (fastz && l != last && own) { const double *wl_ = Wc + last * nsp + 3 * kme; double *wd_ = Wc + l * nsp + 3 * kme; wd_[0] = wl_[0]; wd_[1] = wl_[1]; wd_[2] = wl_[2]; }
                    __syncthreads();                      // D2
                    if (l != last) {
                        if (lane == l) { uq = ulast; ck = cmk; ct = cmt; }
                        if (BIG && lane + 64 == l) { uq2 = ulast; ck2 = cmk; ct2 = cmt; }
                    }
                    if (lane == clk) { amask &= ~(1u << clt); sPos[6 * clk + clt] = (short)-1; }
                    if (l != last && lane == cmk) { posk = (posk & ~(0xffull << (8 * cmt))) | ((unsigned long long)l << (8 * cmt)); sPos[6 * cmk + cmt] = (short)l; }
                    __syncthreads();                      // D3
                    xmask = 0;
                    --q;
                }
            }
        }
        if (lane == 0) sCtl[0] = 1;                       // workers leave at their next X1
        __syncthreads();
        QR_TS(5);
        if (handoff) {
            if (lane < q) sAct[lane] = 6 * ck + ct;       // sPos has been kept all along
            h_x0 = x0; h_x1 = x1; h_x2 = x2; h_u0 = (lane < q) ? uq : 0.0; h_amask = amask; h_q = q; h_iter = iter;
            wave_sync();
        } else {
            wave_sync();
            if (lane < 12) xz[lane] = 0.0;
            wave_sync();
            if (own) { const int ls = sLs[kme]; if (ls < 4) { xz[3 * ls] = x0; xz[3 * ls + 1] = x1; xz[3 * ls + 2] = x2; } }
            wave_sync();
            mpc_outputs(lane, rid, n, xz, R, sJ, C, g_q, g_force, g_force_wbc, force_stride, g_tau, P.epilogue);
            if (lane == 0 && g_status) g_status[rid] = st | ((iter & 0xffff) << 8);
            if (lane == 0 && (st & QRGPU_ST_MPC_OVERFLOW_D) && P.rescue_list && !P.rescue_mode) P.rescue_list[atomicAdd(P.rescue_count + P.rescue_parity, 1)] = rid;
            if (lane == 0 && P.cost) { const long long c = (clock64() - t_begin) >> 12; P.cost[rid] = c > 255 ? 255 : (int)c; }
            QR_TS(6);
#ifndef QR_TRACE
            if (lane == 0 && dbgT) { dbgT[(size_t)rid * 16 + 7] = ns; dbgT[(size_t)rid * 16 + 14] = q; }
#ifdef QR_GI_STAMPS
            if (lane == 0 && dbgT) for (int i = 0; i < 6; ++i) dbgT[(size_t)rid * 16 + 8 + i] = cs_t[i];
#endif
#endif
            return;
        }
    } else
    if constexpr (MULTI) {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the partitioned loops below run on the SALU
        const bool own = lane < nls;
        const int kme = own ? lane : 0;
        const double im = (double)(1.f / C.mu);          // mu_ (:230) as fmat holds it
        const double fmaxk = own ? fmk[kme] : 0.0;
        const int tril = tri(lane);
        const double tol = 1e-9;
        const double INF = __builtin_inf();
        if (qcap > 64) qcap = 64;
        // ---- phase 4: x = -M g, block columns kc = wv (mod 4) per wave
        double x0 = 0.0, x1 = 0.0, x2 = 0.0;
        {
            double p0 = 0.0, p1 = 0.0, p2 = 0.0;
            if (own) {
                for (int kc = wv; kc < nls; kc += 4) {
                    Blk B; load_block(Mb, kme, kc, B);
                    const double g0 = gl[3 * kc], g1 = gl[3 * kc + 1], g2 = gl[3 * kc + 2];
                    p0 += B.m[0] * g0 + B.m[1] * g1 + B.m[2] * g2;
                    p1 += B.m[3] * g0 + B.m[4] * g1 + B.m[5] * g2;
                    p2 += B.m[6] * g0 + B.m[7] * g1 + B.m[8] * g2;
                }
                xz[wv * NV + 3 * kme] = p0; xz[wv * NV + 3 * kme + 1] = p1; xz[wv * NV + 3 * kme + 2] = p2;
            }
            __syncthreads();
            if (own) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { x0 -= xz[v * NV + 3 * kme]; x1 -= xz[v * NV + 3 * kme + 1]; x2 -= xz[v * NV + 3 * kme + 2]; }
            }
            __syncthreads();
        }
        QR_TS(4);
        // ---- phase 5
        int q = 0, iter = 0;
        unsigned amask = 0, xmask = 0;
        unsigned long long posk = 0;                      // byte t: working-set position of row t of my leg-step
        int ck = 0, ct = 0;                               // constraint (leg-step, row) at working-set position `lane`
        double uq = 0.0;                                  // its multiplier
        const int maxit = 40 * nls + 100;
        bool done = (nls == 0);
        bool after_drop = false;
        bool fastz = qW > 0;
        long long acc_t[6] = {0, 0, 0, 0, 0, 0}; long long tq0 = dbgT ? clock64() : 0;
#ifdef QR_GI_STAMPS      // sub-phase cycle accounting of the loop below (build.py: QRGPU_GI_STAMPS=1); costs ~15 % when compiled in
#define QM_STAMP(i) do { if (dbgT) { const long long t_ = clock64(); acc_t[i] += t_ - tq0; tq0 = t_; } } while (0)
#else
#define QM_STAMP(i) do { } while (0)
#endif
        while (!done) {
            // step 1: most violated inactive row (ties -> lowest id)
            // (lanes without a leg-step run the same arithmetic on their zeros and are masked by one select: no divergent region)
            double bs = INF; int bt = 0;
            {
                const double s[6] = {im * x0 + x2, -im * x0 + x2, im * x1 + x2, -im * x1 + x2, x2, fmaxk - x2};
                const unsigned blocked = own ? (amask | xmask) : 0x3fu;
#pragma unroll
                for (int t = 0; t < 6; ++t) { const bool take = !((blocked >> t) & 1u) && s[t] < bs; bs = take ? s[t] : bs; bt = take ? t : bt; }
            }
            const double smin = wave_min_d(bs);
            if (!(smin < -tol)) {
                // Rows left out as dependent are combinations of active rows and hold with them -- unless the exclusion was a
                // numerical accident (seen at twice SURVEY 8d's ranges after ~1000 iterations): then this point is not the optimum.
                // Likewise an active row must be tight: the updates keep x = x0 + M N u whatever r was, but a drifted S^-1 lets
                // active rows slip, and those are not scanned.  Either residual beyond 1e-4 N (a tenth of the force tolerance at 100 N) flags the robot.
                double be = 0.0;
                if (own && (xmask | amask)) {
                    const double s[6] = {im * x0 + x2, -im * x0 + x2, im * x1 + x2, -im * x1 + x2, x2, fmaxk - x2};
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        if (((xmask >> t) & 1u) && s[t] < be) be = s[t];
                        if (((amask >> t) & 1u) && -__builtin_fabs(s[t]) < be) be = -__builtin_fabs(s[t]);
                    }
                }
                if (wave_min_d(be) < -1e-4) st |= QRGPU_ST_MPC_INFEAS_D;
                break;
            }
            const int kp = __builtin_amdgcn_readfirstlane(first_lane(bs == smin));
            const int tp = __builtin_amdgcn_readlane(bt, kp);
            double c0, c1, c2;
            cons_vec(tp, im, c0, c1, c2);
            const double ci0p = (tp == 5) ? readlane_d(fmaxk, kp) : 0.0;
            double up = 0.0;
            QM_STAMP(0);
            for (;;) {
                q = __builtin_amdgcn_readfirstlane(q);
                if (++iter > maxit) { st |= QRGPU_ST_MPC_MAXITER_D; done = true; break; }
                // w_k = M_{k,kp} c_p
                double w0, w1, w2_;
                {
                    Blk B; load_block(Mb, kme, kp, B);                // kme is a valid leg-step in every lane
                    w0 = B.m[0] * c0 + B.m[1] * c1 + B.m[2] * c2;
                    w1 = B.m[3] * c0 + B.m[4] * c1 + B.m[5] * c2;
                    w2_ = B.m[6] * c0 + B.m[7] * c1 + B.m[8] * c2;
                }
                const double delta = c0 * readlane_d(w0, kp) + c1 * readlane_d(w1, kp) + c2 * readlane_d(w2_, kp);
                // d = N' w : position i needs w of leg-step ck(i)
                double dq = 0.0;
                {
                    // (taking d from the W_A cache instead -- c_p' (W_A row i)(kp) -- is the same number in exact arithmetic but rounds
                    // differently from delta; for a row that is dependent on active rows of its own leg-step the cancellation in
                    // z'c = delta - d'r is then no longer exact, and one robot in 36 864 ended 8e-4 N off: scratch/diag_outlier.py)
                    double a0, a1, a2;
                    cons_vec(ct, im, a0, a1, a2);
                    const double g0 = __shfl(w0, ck, 64), g1 = __shfl(w1, ck, 64), g2 = __shfl(w2_, ck, 64);
                    dq = (lane < q) ? a0 * g0 + a1 * g1 + a2 * g2 : 0.0;
                }
                QM_STAMP(1);
                // r = S^-1 d, columns j = wv (mod 4) here.  (i, j) at tri(i) + j for j <= i, else tri(j) + i.
                if (after_drop) { __syncthreads(); after_drop = false; }   // B1: wave 0's row move of the last drop is visible
                double rq = 0.0;
                {
                    const int i0 = (lane < q) ? lane : 0;
                    double pr = 0.0;
                    int j = wv;
                    for (; j + 12 < q; j += 16) {
                        double sv[4], dj[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int jj = j + 4 * u;
                            sv[u] = Sinv[(jj <= i0) ? tril + jj : tri(jj) + i0];
                            dj[u] = readlane_d(dq, jj);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) pr += sv[u] * dj[u];
                    }
                    for (; j < q; j += 4) pr += Sinv[(j <= i0) ? tril + j : tri(j) + i0] * readlane_d(dq, j);
                    xr[wv * 64 + lane] = pr;                      // every lane has its slot
                    __syncthreads();                              // B2
                    { const double rs = (xr[lane] + xr[64 + lane]) + (xr[128 + lane] + xr[192 + lane]); rq = (lane < q) ? rs : 0.0; }
                }
                QM_STAMP(2);
                const double dr = wave_sum_d(rq * dq);
                const double zc = delta - dr;                    // z'c_p
                // dual step length: min u_j / r_j over r_j > 0
                double tt = INF;
                { const double tq_ = uq * fast_rcp(rq); tt = (lane < q && rq > 0.0) ? tq_ : INF; }
                const double t1 = wave_min_d(tt);
                const int lpos = (t1 < INF) ? first_lane(tt == t1) : -1;
                const double sp = c0 * readlane_d(x0, kp) + c1 * readlane_d(x1, kp) + c2 * readlane_d(x2, kp) + ci0p;
                const bool have_z = zc > 1e-13 * delta;
                const double izc = fast_rcp(zc);
                const double t2 = have_z ? -sp * izc : INF;
                const double t = t1 < t2 ? t1 : t2;
                if (!(t < INF)) {
                    // degenerate corner (see the single-wave path): leave the row out until the working set changes
                    if (lane == kp) xmask |= 1u << tp;
                    break;
                }
                const bool full = have_z && t == t2;
                QM_STAMP(3);
#ifdef QR_TRACE
                if (dbgT && tid == 0 && iter <= 7) { dbgT[(size_t)rid * 16 + 2 * (iter - 1)] = ((long long)kp << 32) | (tp << 24) | (q << 16) | (full ? 1 : 0) | (have_z ? 2 : 0); dbgT[(size_t)rid * 16 + 2 * (iter - 1) + 1] = __double_as_longlong(t); }
#endif
                if (full) {
                    // bordered update of S^-1 (needs only r and 1/z'c): columns j = wv (mod 4); published by B3 below
                    if (q >= qcap) {
                        // Nothing of this inner iteration has been applied yet, and with q at its maximum no row was dropped in this
                        // outer iteration either (up == 0): (x, u, working set, S^-1) is exactly the state the outer loop started from.
                        if (q == 64 && qcap_full > 64) { handoff = true; done = true; break; }
                        st |= QRGPU_ST_MPC_OVERFLOW_D; done = true; break;
                    }
                    const double isg = izc;
                    const bool act0 = lane < q;
                    const double ri = rq * isg;
                    int j = wv;
                    for (; j + 12 < q; j += 16) {
                        double sv[4], rj[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int jj = j + 4 * u; rj[u] = readlane_d(rq, jj); sv[u] = (act0 && jj <= lane) ? Sinv[tril + jj] : 0.0; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int jj = j + 4 * u; if (act0 && jj <= lane) Sinv[tril + jj] = sv[u] + ri * rj[u]; }
                    }
                    for (; j < q; j += 4) { const double rj = readlane_d(rq, j); if (act0 && j <= lane) Sinv[tril + j] += ri * rj; }
                    if (wv == 0) {
                        if (act0) Sinv[tri(q) + lane] = -rq * isg;
                        if (lane == 0) Sinv[tri(q) + q] = isg;
                    }
                }
                QM_STAMP(4);
                if (have_z) {
                    double p0 = 0.0, p1 = 0.0, p2 = 0.0;
                    if (fastz) {
                        // partial W_A r over the positions i = wv (mod 4), two per trip
                        const double *wk = Wc + 3 * kme;
                        int i = wv;
                        for (; i + 4 < q; i += 8) {
                            const double ra = readlane_d(rq, i), rb = readlane_d(rq, i + 4);
                            {
                                const double *wa = wk + i * nsp, *wb = wk + (i + 4) * nsp;
                                const double a0 = wa[0], a1 = wa[1], a2 = wa[2], b0 = wb[0], b1 = wb[1], b2 = wb[2];
                                p0 += a0 * ra + b0 * rb; p1 += a1 * ra + b1 * rb; p2 += a2 * ra + b2 * rb;
                            }
                        }
                        if (i < q) {
                            const double ra = readlane_d(rq, i);
                            { const double *wa = wk + i * nsp; p0 += wa[0] * ra; p1 += wa[1] * ra; p2 += wa[2] * ra; }
                        }
                    } else {
                        // y_k = sum over the active rows of my leg-step of c_row * r(position)
                        double y0 = 0.0, y1 = 0.0, y2 = 0.0;
#pragma unroll
                        for (int tq = 0; tq < 6; ++tq) {
                            const int ps = (int)((posk >> (8 * tq)) & 0x3full);
                            const double rr = __shfl(rq, ps, 64);
                            if ((amask >> tq) & 1u) {
                                double a0, a1, a2; cons_vec(tq, im, a0, a1, a2);
                                y0 += a0 * rr; y1 += a1 * rr; y2 += a2 * rr;
                            }
                        }
                        // partial z over every 4th active leg-step (rank among the active ones = wv mod 4, found with mbcnt so that every
                        // wave scans only its own bits), two block
                        // columns per trip so that the second LDS load is in flight while the first is used
                        const bool hasrow = own && amask != 0;
                        const unsigned long long kall = __ballot(hasrow);
                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(kall >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kall, 0u));
                        unsigned long long km = __ballot(hasrow && (rank & 3) == wv);
                        while (km) {
                            const int kc = (int)__builtin_ctzll(km);
                            km &= km - 1;
                            int kc2 = -1;
                            if (km) { kc2 = (int)__builtin_ctzll(km); km &= km - 1; }
                            const int kr = kc2 >= 0 ? kc2 : kc;
                            const double sc = kc2 >= 0 ? 1.0 : 0.0;
                            Blk B, B2;
                            if (own) { load_block(Mb, kme, kc, B); load_block(Mb, kme, kr, B2); }
                            const double q0 = readlane_d(y0, kc), q1 = readlane_d(y1, kc), q2 = readlane_d(y2, kc);
                            const double s0 = sc * readlane_d(y0, kr), s1 = sc * readlane_d(y1, kr), s2 = sc * readlane_d(y2, kr);
                            if (own) {
                                p0 += (B.m[0] * q0 + B.m[1] * q1 + B.m[2] * q2) + (B2.m[0] * s0 + B2.m[1] * s1 + B2.m[2] * s2);
                                p1 += (B.m[3] * q0 + B.m[4] * q1 + B.m[5] * q2) + (B2.m[3] * s0 + B2.m[4] * s1 + B2.m[5] * s2);
                                p2 += (B.m[6] * q0 + B.m[7] * q1 + B.m[8] * q2) + (B2.m[6] * s0 + B2.m[7] * s1 + B2.m[8] * s2);
                            }
                        }
                    }
                    if (own) { xz[wv * NV + 3 * kme] = p0; xz[wv * NV + 3 * kme + 1] = p1; xz[wv * NV + 3 * kme + 2] = p2; }
                    __syncthreads();                              // B3
                    {
                        double z0 = w0, z1 = w1, z2 = w2_;
#pragma unroll
                        for (int v = 0; v < 4; ++v) { z0 -= xz[v * NV + 3 * kme]; z1 -= xz[v * NV + 3 * kme + 1]; z2 -= xz[v * NV + 3 * kme + 2]; }
                        x0 += t * z0; x1 += t * z1; x2 += t * z2;      // (lanes without a leg-step shadow lane 0; their x is never read)
                    }
                }
                uq -= t * rq;
                up += t;
                QM_STAMP(5);
                if (full) {
                    // full step: the row joined the working set at position q (S^-1 already updated above)
                    if (fastz) {
                        if (q < qW) { if (wv == 0 && own) { double *wq = Wc + q * nsp + 3 * kme; wq[0] = w0; wq[1] = w1; wq[2] = w2_; } }
                        else fastz = false;
                    }
                    if (lane == q) { uq = up; ck = kp; ct = tp; }
                    if (lane == kp) { amask |= 1u << tp; posk = (posk & ~(0xffull << (8 * tp))) | ((unsigned long long)q << (8 * tp)); }
                    xmask = 0;
                    ++q;
                    break;
                }
                // partial or dual-only step: the constraint at lpos leaves; downdate S^-1, move the last one into its slot
                {
                    const int l = lpos, last = q - 1;
                    double sl = 0.0;                              // column l of S^-1
                    if (lane < q) sl = Sinv[pidx(lane, l)];
                    const double isl = fast_rcp(readlane_d(sl, l));
                    __syncthreads();                              // everyone has column l before anyone changes S^-1
                    for (int j = wv; j < q; j += 4) {
                        if (j == l) continue;
                        const double sj = readlane_d(sl, j) * isl;
                        if (lane < q && lane != l && j <= lane) Sinv[tril + j] -= sl * sj;
                    }
                    __syncthreads();
                    const int clk = __builtin_amdgcn_readlane(ck, l), clt = __builtin_amdgcn_readlane(ct, l);
                    const int cmk = __builtin_amdgcn_readlane(ck, last), cmt = __builtin_amdgcn_readlane(ct, last);
                    if (l != last) {
                        double m0 = 0.0;
                        if (wv == 0 && lane < last) m0 = (lane == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, lane)];
                        __syncthreads();
                        if (wv == 0 && lane < last) Sinv[pidx(l, lane)] = m0;
                        if (fastz && wv == 0 && own) { const double *wl_ = Wc + last * nsp + 3 * kme; double *wd_ = Wc + l * nsp + 3 * kme; wd_[0] = wl_[0]; wd_[1] = wl_[1]; wd_[2] = wl_[2]; }
                        const double ulast = readlane_d(uq, last);
                        if (lane == l) { uq = ulast; ck = cmk; ct = cmt; }
                    }
                    if (lane == clk) amask &= ~(1u << clt);
                    if (l != last && lane == cmk) posk = (posk & ~(0xffull << (8 * cmt))) | ((unsigned long long)l << (8 * cmt));
                    xmask = 0;
                    --q;
                    after_drop = true;
                }
            }
        }
        QR_TS(5);
        if (wv != 0) return;
        if (handoff) {
            // wave 0 carries on alone: working-set tables of the single-wave loop from the per-lane registers
            if (lane < q) sAct[lane] = 6 * ck + ct;
            if (own) {
#pragma unroll
                for (int t = 0; t < 6; ++t) sPos[6 * lane + t] = ((amask >> t) & 1u) ? (short)((posk >> (8 * t)) & 0x3full) : (short)-1;
            }
            h_x0 = x0; h_x1 = x1; h_x2 = x2; h_u0 = (lane < q) ? uq : 0.0; h_amask = amask; h_q = q; h_iter = iter;
            wave_sync();
        } else {
            // ---- phase 6 (wave 0): stage the first-step forces through LDS, then the shared output code below
            wave_sync();
            if (lane < 12) xz[lane] = 0.0;
            wave_sync();
            if (own) { const int ls = sLs[kme]; if (ls < 4) { xz[3 * ls] = x0; xz[3 * ls + 1] = x1; xz[3 * ls + 2] = x2; } }
            wave_sync();
            mpc_outputs(lane, rid, n, xz, R, sJ, C, g_q, g_force, g_force_wbc, force_stride, g_tau, P.epilogue);
            if (lane == 0 && g_status) g_status[rid] = st | ((iter & 0xffff) << 8);
            if (lane == 0 && (st & QRGPU_ST_MPC_OVERFLOW_D) && P.rescue_list && !P.rescue_mode) P.rescue_list[atomicAdd(P.rescue_count + P.rescue_parity, 1)] = rid;
            if (lane == 0 && P.cost) { const long long c = (clock64() - t_begin) >> 12; P.cost[rid] = c > 255 ? 255 : (int)c; }
            QR_TS(6);
#ifndef QR_TRACE
            if (lane == 0 && dbgT) { dbgT[(size_t)rid * 16 + 7] = ns; for (int i = 0; i < 6; ++i) dbgT[(size_t)rid * 16 + 8 + i] = acc_t[i]; dbgT[(size_t)rid * 16 + 14] = q; }
#endif
            return;
        }
    }
    if (tid >= 64) return;          // phases 4-6 are a single wavefront; no workgroup barrier below

    // ---------------- phase 4: x = -M g  (lane k owns leg-step k) ----------------
    const bool own = lane < nls;
    const int kme = own ? lane : 0;
    double x0 = h_x0, x1 = h_x1, x2 = h_x2;
    if (own && !handoff) {
        for (int kc = 0; kc < nls; ++kc) {
            Blk B; load_block(Mb, kme, kc, B);
            const double g0 = gl[3 * kc], g1 = gl[3 * kc + 1], g2 = gl[3 * kc + 2];
            x0 -= B.m[0] * g0 + B.m[1] * g1 + B.m[2] * g2;
            x1 -= B.m[3] * g0 + B.m[4] * g1 + B.m[5] * g2;
            x2 -= B.m[6] * g0 + B.m[7] * g1 + B.m[8] * g2;
        }
    }
    if (!handoff) QR_TS(4);
    if (handoff) qcap = qcap_full;                    // the four-wave loop had clamped it to its 64 lanes

    // ---------------- phase 5: dual active set (wave 0) ----------------
    const double im = (double)(1.f / C.mu);          // mu_ (:230) as fmat holds it
    const double fmaxk = own ? fmk[kme] : 0.0;
    const int tril = tri(lane);
    const double tol = 1e-9;
    const double INF = __builtin_inf();
    long long acc_t[6] = {0, 0, 0, 0, 0, 0}; long long tq0 = 0;
#define QR_STAMP(i) do { if (dbgT) { const long long t_ = clock64(); acc_t[i] += t_ - tq0; tq0 = t_; } } while (0)
    int q = h_q, iter = h_iter;
    unsigned amask = h_amask;                         // active rows of my leg-step (6 bits)
    unsigned xmask = 0;                               // rows found numerically dependent on the working set (skipped until the set changes)
    double u0 = h_u0, u1 = 0.0;                       // multipliers of working-set positions lane, lane+64
    const int maxit = 40 * nls + 100;
    bool done = (nls == 0);
    while (!done) {
        if (dbgT) tq0 = clock64();
        // step 1: most violated inactive row (ties -> lowest id)
        double bs = INF; int bt = 0;
        if (own) {
            const double s[6] = {im * x0 + x2, -im * x0 + x2, im * x1 + x2, -im * x1 + x2, x2, fmaxk - x2};
#pragma unroll
            for (int t = 0; t < 6; ++t) if (!(((amask | xmask) >> t) & 1u) && s[t] < bs) { bs = s[t]; bt = t; }
        }
        const double smin = wave_min_d(bs);
        if (!(smin < -tol)) {
            double be = 0.0;                           // as in the four-wave path: excluded rows hold, active rows are tight
            if (own && (xmask | amask)) {
                const double s[6] = {im * x0 + x2, -im * x0 + x2, im * x1 + x2, -im * x1 + x2, x2, fmaxk - x2};
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    if (((xmask >> t) & 1u) && s[t] < be) be = s[t];
                    if (((amask >> t) & 1u) && -__builtin_fabs(s[t]) < be) be = -__builtin_fabs(s[t]);
                }
            }
            if (wave_min_d(be) < -1e-4) st |= QRGPU_ST_MPC_INFEAS_D;
            break;
        }
        const int kp = __builtin_amdgcn_readfirstlane(first_lane(bs == smin));
        const int tp = __builtin_amdgcn_readlane(bt, kp);
        const int p = 6 * kp + tp;
        double c0, c1, c2;
        cons_vec(tp, im, c0, c1, c2);
        const double ci0p = (tp == 5) ? readlane_d(fmaxk, kp) : 0.0;
        double up = 0.0;
        QR_STAMP(0);
        for (;;) {
            q = __builtin_amdgcn_readfirstlane(q);           // q is wave-uniform by construction; tell the compiler
            if (++iter > maxit) { st |= QRGPU_ST_MPC_MAXITER_D; done = true; break; }
            // w_k = M_{k,kp} c_p
            double w0 = 0.0, w1 = 0.0, w2_ = 0.0;
            if (own) {
                Blk B; load_block(Mb, kme, kp, B);
                w0 = B.m[0] * c0 + B.m[1] * c1 + B.m[2] * c2;
                w1 = B.m[3] * c0 + B.m[4] * c1 + B.m[5] * c2;
                w2_ = B.m[6] * c0 + B.m[7] * c1 + B.m[8] * c2;
                wl[3 * kme] = w0; wl[3 * kme + 1] = w1; wl[3 * kme + 2] = w2_;
            }
            const double delta = c0 * readlane_d(w0, kp) + c1 * readlane_d(w1, kp) + c2 * readlane_d(w2_, kp);
            wave_sync();
            QR_STAMP(1);
            // d = N' w  (position i lives in lane i & 63, slot i >> 6)
            double d0 = 0.0, d1 = 0.0;
            {
                if (lane < q) { const int cj = sAct[lane]; const int kj = cj / 6; double a0, a1, a2; cons_vec(cj - 6 * kj, im, a0, a1, a2);
                                d0 = a0 * wl[3 * kj] + a1 * wl[3 * kj + 1] + a2 * wl[3 * kj + 2]; }
                if (q > 64 && lane + 64 < q) { const int cj = sAct[lane + 64]; const int kj = cj / 6; double a0, a1, a2; cons_vec(cj - 6 * kj, im, a0, a1, a2);
                                     d1 = a0 * wl[3 * kj] + a1 * wl[3 * kj + 1] + a2 * wl[3 * kj + 2]; }
            }
            // r = S^-1 d.  Lane i owns position i (and i + 64 in the rare q > 64 case).  S^-1 is packed lower:
            // (i, j) at tri(i) + j for j <= i, else tri(j) + i.  Four loads are issued before they are consumed.
            double r0 = 0.0, r1 = 0.0;
            {
                const int q0 = q < 64 ? q : 64;
                const int i0 = (lane < q) ? lane : 0;
                int j = 0;
                for (; j + 4 <= q0; j += 4) {
                    double sv[4], dj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int jj = j + u;
                        sv[u] = Sinv[(jj <= i0) ? tril + jj : tri(jj) + i0];
                        dj[u] = readlane_d(d0, jj);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) r0 += sv[u] * dj[u];
                }
                for (; j < q0; ++j) r0 += Sinv[(j <= i0) ? tril + j : tri(j) + i0] * readlane_d(d0, j);
                if (lane >= q) r0 = 0.0;
                if (q > 64) {                     // cold path: positions 64..q-1
                    const int i1 = lane + 64;
                    for (int jj = 64; jj < q; ++jj) { const double dj = readlane_d(d1, jj - 64); if (lane < q) r0 += Sinv[pidx(lane, jj)] * dj; }
                    for (int jj = 0; jj < q; ++jj) {
                        const double dj = (jj < 64) ? readlane_d(d0, jj) : readlane_d(d1, jj - 64);
                        if (i1 < q) r1 += Sinv[pidx(i1, jj)] * dj;
                    }
                }
            }
            QR_STAMP(2);
            const double dr = wave_sum_d(r0 * d0 + r1 * d1);
            const double zc = delta - dr;                    // z'c_p
            // dual step length: min u_j / r_j over r_j > 0
            double tt = INF;
            if (lane < q && r0 > 0.0) tt = u0 * fast_rcp(r0);
            double tt1 = INF;
            if (q > 64 && lane + 64 < q && r1 > 0.0) tt1 = u1 * fast_rcp(r1);
            const double t1 = wave_min_d(fmin(tt, tt1));
            int lpos = -1;
            if (t1 < INF) {
                const int la = first_lane(tt == t1);
                lpos = (la >= 0) ? la : 64 + first_lane(tt1 == t1);
            }
            const double sp = c0 * readlane_d(x0, kp) + c1 * readlane_d(x1, kp) + c2 * readlane_d(x2, kp) + ci0p;
            const bool have_z = zc > 1e-13 * delta;
            const double izc = fast_rcp(zc);
            const double t2 = have_z ? -sp * izc : INF;
            const double t = t1 < t2 ? t1 : t2;
            if (!(t < INF)) {
                // The row is (numerically) in the span of the working set and no multiplier blocks: u = 0 is always
                // feasible, so this is a degenerate corner (e.g. f = 0 with five rows on three unknowns), its violation
                // is rounding.  Leave it out until the working set changes (what QuadProg++ does, QuadProg++.cc "iaexcl").
                if (lane == kp) xmask |= 1u << tp;
                break;
            }
            QR_STAMP(3);
            if (have_z) {
                // y_k = sum of active rows of my leg-step times r;  z = w - M y;  x += t z
                if (lane < q) rl[lane] = r0;
                if (q > 64 && lane + 64 < q) rl[lane + 64] = r1;
                wave_sync();
                double y0 = 0.0, y1 = 0.0, y2 = 0.0;
                if (own && amask) {
#pragma unroll
                    for (int tq = 0; tq < 6; ++tq)
                        if ((amask >> tq) & 1u) {
                            const double rr = rl[sPos[6 * kme + tq]];
                            double a0, a1, a2; cons_vec(tq, im, a0, a1, a2);
                            y0 += a0 * rr; y1 += a1 * rr; y2 += a2 * rr;
                        }
                }
                unsigned long long km = __ballot(own && amask != 0);
                double z0 = w0, z1 = w1, z2 = w2_;
                while (km) {
                    const int kc = (int)__builtin_ctzll(km);
                    km &= km - 1;
                    int kc2 = -1;
                    if (km) { kc2 = (int)__builtin_ctzll(km); km &= km - 1; }
                    Blk B, B2;
                    load_block(Mb, kme, kc, B);
                    load_block(Mb, kme, kc2 >= 0 ? kc2 : kc, B2);
                    const double q0 = readlane_d(y0, kc), q1 = readlane_d(y1, kc), q2 = readlane_d(y2, kc);
                    const int kr = kc2 >= 0 ? kc2 : 0;
                    const double sc = kc2 >= 0 ? 1.0 : 0.0;
                    const double p0 = sc * readlane_d(y0, kr), p1 = sc * readlane_d(y1, kr), p2 = sc * readlane_d(y2, kr);
                    z0 -= (B.m[0] * q0 + B.m[1] * q1 + B.m[2] * q2) + (B2.m[0] * p0 + B2.m[1] * p1 + B2.m[2] * p2);
                    z1 -= (B.m[3] * q0 + B.m[4] * q1 + B.m[5] * q2) + (B2.m[3] * p0 + B2.m[4] * p1 + B2.m[5] * p2);
                    z2 -= (B.m[6] * q0 + B.m[7] * q1 + B.m[8] * q2) + (B2.m[6] * p0 + B2.m[7] * p1 + B2.m[8] * p2);
                }
                x0 += t * z0; x1 += t * z1; x2 += t * z2;
            }
            QR_STAMP(4);
            u0 -= t * r0; u1 -= t * r1;
            up += t;
            if (have_z && t == t2) {
                // full step: p joins the working set; bordered update of S^-1
                if (q >= qcap) { st |= QRGPU_ST_MPC_OVERFLOW_D; done = true; break; }
                const double isg = izc;
                {
                    // row i of the lower triangle belongs to lane i: S^-1(i, j) += r_i r_j / sigma for j <= i
                    const int q0 = q < 64 ? q : 64;
                    const bool act0 = lane < q;
                    const double ri = r0 * isg;
                    int j = 0;
                    for (; j + 4 <= q0; j += 4) {
                        double sv[4], rj[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { rj[u] = readlane_d(r0, j + u); sv[u] = (act0 && j + u <= lane) ? Sinv[tril + j + u] : 0.0; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (act0 && j + u <= lane) Sinv[tril + j + u] = sv[u] + ri * rj[u];
                    }
                    for (; j < q0; ++j) { const double rj = readlane_d(r0, j); if (act0 && j <= lane) Sinv[tril + j] += ri * rj; }
                    if (q > 64) {                 // cold path
                        const int i1 = lane + 64;
                        for (int jj = 0; jj < q; ++jj) {
                            const double rj = ((jj < 64) ? readlane_d(r0, jj) : readlane_d(r1, jj - 64)) * isg;
                            if (i1 < q && jj <= i1) Sinv[tri(i1) + jj] += r1 * rj;
                        }
                        if (i1 < q) Sinv[tri(q) + i1] = -r1 * isg;
                    }
                    if (act0) Sinv[tri(q) + lane] = -r0 * isg;
                }
                if (lane == 0) { Sinv[tri(q) + q] = isg; sAct[q] = p; sPos[p] = (short)q; }
                if (lane == (q & 63)) { if (q < 64) u0 = up; else u1 = up; }
                if (lane == kp) amask |= 1u << tp;
                xmask = 0;
                ++q;
                wave_sync();
                QR_STAMP(5);
                break;
            }
            // partial or dual-only step: the constraint at lpos leaves; downdate S^-1, move the last one into its slot
            {
                const int l = lpos, last = q - 1;
                double s0 = 0.0, s1 = 0.0;                    // column l of S^-1
                if (lane < q) s0 = Sinv[pidx(lane, l)];
                if (lane + 64 < q) s1 = Sinv[pidx(lane + 64, l)];
                const double isl = fast_rcp((l < 64) ? readlane_d(s0, l) : readlane_d(s1, l - 64));
                wave_sync();
                {
                    const int i0 = lane, i1 = lane + 64;
                    for (int j = 0; j < q; ++j) {
                        if (j == l) continue;
                        const double sj = ((j < 64) ? readlane_d(s0, j) : readlane_d(s1, j - 64)) * isl;
                        if (i0 < q && i0 != l && j <= i0) Sinv[tri(i0) + j] -= s0 * sj;
                        if (i1 < q && i1 != l && j <= i1) Sinv[tri(i1) + j] -= s1 * sj;
                    }
                }
                wave_sync();
                const int cl = sAct[l], clast = sAct[last];
                if (l != last) {
                    double m0 = 0.0, m1 = 0.0;
                    if (lane < last) m0 = (lane == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, lane)];
                    if (lane + 64 < last) m1 = (lane + 64 == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, lane + 64)];
                    wave_sync();
                    if (lane < last) Sinv[pidx(l, lane)] = m0;
                    if (lane + 64 < last) Sinv[pidx(l, lane + 64)] = m1;
                    const double ulast = (last < 64) ? readlane_d(u0, last) : readlane_d(u1, last - 64);
                    if (lane == (l & 63)) { if (l < 64) u0 = ulast; else u1 = ulast; }
                }
                wave_sync();
                if (lane == 0) {
                    sPos[cl] = -1;
                    if (l != last) { sAct[l] = clast; sPos[clast] = (short)l; }
                }
                if (lane == cl / 6) amask &= ~(1u << (cl - 6 * (cl / 6)));
                xmask = 0;
                --q;
                wave_sync();
            }
        }
    }
    QR_TS(5);

    // ---------------- phase 6: outputs ----------------
    // f(axis,leg) = q_soln[3*leg+axis] for horizon step 0 (GetMPCSolution, :446-451); swing legs are 0.
    if (lane < 12) yl[lane] = 0.0;
    wave_sync();
    if (own) { const int ls = sLs[kme]; if (ls < 4) { yl[3 * ls] = x0; yl[3 * ls + 1] = x1; yl[3 * ls + 2] = x2; } }
    wave_sync();
    mpc_outputs(lane, rid, n, yl, R, sJ, C, g_q, g_force, g_force_wbc, force_stride, g_tau, P.epilogue);
    if (lane == 0 && g_status) g_status[rid] = st | ((iter & 0xffff) << 8);
    if (tid == 0 && P.cost) { const long long c = (clock64() - t_begin) >> 12; P.cost[rid] = c > 255 ? 255 : (int)c; }
    QR_TS(6);
#ifndef QR_TRACE
    if (lane == 0 && dbgT) { dbgT[(size_t)rid * 16 + 7] = ns; for (int i = 0; i < 6; ++i) dbgT[(size_t)rid * 16 + 8 + i] = acc_t[i]; dbgT[(size_t)rid * 16 + 14] = q; }
#endif
}

#define QR_MPC_INST(MAXB, MULTI, TAG)                                                                                                 \
    template __global__ void qr_mpc_kernel<MAXB, MULTI, TAG>(MpcLaunch, const int *, const float *, const float *, const float *, const float *, \
                                                             float *, float *, int *, float *, float *, float *, int, long long *);
QR_MPC_INST(4, true, 0)
QR_MPC_INST(4, true, 1)
QR_MPC_INST(9, true, 0)
QR_MPC_INST(9, false, 0)

// Self-test of the cross-lane helpers (qrgpu_selftest): a permutation's minimum, a sum, first_lane, readlane.
__global__ void qr_selftest_kernel(double *out)
{
    const int lane = threadIdx.x & 63;
    const double v = (double)((lane * 37 + 11) % 64) - 20.5;          // a permutation of -20.5 .. 42.5
    const double mn = wave_min_d(v);
    const double sm = wave_sum_d((double)(lane + 1));
    const int fl = first_lane(v == mn);
    const double rd = readlane_d(v, 17);
    out[lane] = mn;
    out[64 + lane] = sm;
    out[128 + lane] = (double)fl;
    out[192 + lane] = rd;
}

}  // namespace qrgpu
