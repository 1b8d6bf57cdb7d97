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
#include <type_traits>
#include "qr_device_types.h"
#include "qr_wave_helpers.h"

namespace qrgpu {

// Diagnostic hooks of the pipelined tick's timeline (qrgpu_debug_timeline, bench.py QRGPU_BENCH_TIMELINE=1; the finish-order experiment of the WBC
// launch, QRGPU_WBC_ORDER=1, needs them too): compiled in only with -DQR_TIMELINE (QRGPU_EXTRA_FLAGS).  With the pointers merely null at run
// time the main pass was 1.5-2 % slower (1.373 against 1.346 ms at 8192 robots, A/B on one box).
#ifdef QR_TIMELINE
#define QR_TRACE(rid_, bits_) do { if (P.tl && P.solved) atomicOr((unsigned long long *)(P.tl + 768 + 32768 + 4096 + 64 + (P.solved_epoch & 15u) * 1024 + ((rid_) & 1023)), (unsigned long long)(bits_) | ((unsigned long long)(P.solved_epoch & 15u) << 32) | (1ull << 40)); } while (0)
#define QR_P_TL P.tl
#define QR_P_FTIME P.ftime
#else
#define QR_TRACE(rid_, bits_) do { } while (0)
#define QR_P_TL ((long long *)nullptr)
#define QR_P_FTIME ((int *)nullptr)
#endif

#define QR_AS_THREADS 256        // the four waves of phases 4-6 (control wave + three workers)
// Pivot reciprocals of both sweeps: v_rcp_f64 + one Newton step (2.2e-15 relative, scratch/ubench/rcp.hip) -- 0.7 % of the main pass.  (Before
// the periodic refresh of S^-1 existed, one robot of the stress set at twice the 8d ranges wandered into the iteration cap with it; with the
// refresh the stress run is the same with either form: 21 overflow flags at twice the ranges, none inside them, largest count 202 / 204.)
#define QR_RCP_PIVOT fast_rcp1
#define QR_REFRESH_EVERY 100         // a solve still going after this many working-set changes gets S^-1 rebuilt from its working set, and again every so many
#define QR_MAIN_WAVES_PER_SIMD 3     // register budget of the h <= 11 main pass: 3 workgroups per CU (168 VGPRs); the LDS allotment decides how many run
// The executed-arithmetic counters (qrgpu_enable_flop_count) cost the main pass four live fp64 accumulators and 2.4 % of its time even when the
// pointer is null (0.2242 -> 0.2189 ms with them compiled out), so the kernels exist twice: this file compiles them without the counters,
// qr_mpc_kernel_fl.hip includes it with QR_FLOPS_BUILD and gets the same kernels under the name qr_mpc_kernel_fl with the counters in;
// the host launches those only while counting is switched on.
// The same goes for the inspection outputs (the dense H and g of qrgpu_mpc_assemble_batch, the cycle stamps of qrgpu_debug_cycles): the host
// picks the instrumented kernels for a launch that asks for any of them.
#ifdef QR_FLOPS_BUILD
#define qr_mpc_kernel qr_mpc_kernel_fl
#define mpc_solve_robot mpc_solve_robot_fl
#define QR_PFLOPS P.flops
#define QR_DBGT io.dbgT
#define QR_DBGH io.dbgH
#define QR_DBGG io.dbgG
#else
#define QR_PFLOPS ((double *)nullptr)
#define QR_DBGT ((long long *)nullptr)
#define QR_DBGH ((float *)nullptr)
#define QR_DBGG ((float *)nullptr)
#endif

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

// (st_out: a plain store, or -- pipelined tick, where another launch reads the value while this one still runs -- an agent-scope one:
//  global_store ... sc1, written through to memory)
__device__ __forceinline__ void st_out(float *p, float v, bool through)
{
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ void mpc_outputs(int lane, int rid, int n, const double *yl, const float (&R)[3][3], const float *sJ, const MpcType &C,
                                            const float *__restrict__ g_q, float *__restrict__ g_force, float *__restrict__ g_force_wbc,
                                            int force_stride, float *__restrict__ g_tau, int epilogue, bool through = false)
{
    if (lane < 12) {
        const int leg = lane / 3;
        const float fx = (float)yl[3 * leg], fy = (float)yl[3 * leg + 1], fz = (float)yl[3 * leg + 2];
        st_out(&g_force[(size_t)lane * n + rid], (float)yl[lane], through);
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
            st_out(&g_tau[(size_t)lane * n + rid], epilogue ? torque_epilogue(tq, lane, true, epilogue) : tq, through);
        }
    }
}

// Longest-processing-time-first dispatch order for the next launch.  A robot's solve time varies 10x with its active-set
// iteration count, and with two resident workgroups per CU a long solve that starts late sets the kernel time.  Block
// dispatch follows blockIdx, so each XCD chunk [x*chunk, (x+1)*chunk) is counting-sorted by the cost the robots had in
// the previous launch, descending (control ticks are temporally coherent; a stale cost only costs speed).  Robots never
// leave their XCD chunk, so the L2 locality of xcd_robot_index() is kept.  grid = 8, one workgroup per chunk.
// Overlapped ticks: the words one tick's launches leave for the lane's next tick (plan, lists, counters) and the hand-overs INSIDE a tick (rescue
// list: written by the main pass while the planned launch's workgroups read it) are written through and read from memory (agent scope) when xt
// is set -- nothing here may depend on which XCD's L2 holds a line of these arrays.  (The one failure measured on the way was not a cache's: the
// planned list rewritten by the trailing launch while the tick's planned launch was still going down it, MpcLaunch::pre_list_next.)
template <typename T> __device__ __forceinline__ T ld_xt(const T *p, bool xt)
{
    return xt ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <typename T> __device__ __forceinline__ void st_xt(T *p, T v, bool xt)
{
    if (xt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
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
        for (int i = threadIdx.x; i < m; i += blockDim.x) hist[i] = (cost[lo + i] >> 16) & 0xffff;       // (the fine cost: units of 256 cycles)
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const int ci = hist[i];
            int rank = 0;
            for (int j = 0; j < m; ++j) { const int cj = hist[j]; rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
            order[lo + rank] = lo + i;
        }
        return;
    }
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&hist[255 - (cost[i] & 255)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < 256; ++b) { const int c = hist[b]; hist[b] = acc; acc += c; }
    }
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) order[lo + atomicAdd(&hist[255 - (cost[i] & 255)], 1)] = i;
}
// The WBC launch of a pipelined tick takes its robots, inside each XCD chunk, in the order their solves ended in the LAST tick (ftime: low
// word of the shared 100 MHz clock at the moment the flag went up; compared as wrapping differences): a WBC workgroup that lands on a CU
// while its robot is still being solved holds a slot for nothing, and there are far fewer slots than robots while the solves run.
// Rank by comparison as above: a permutation of the chunk by construction.  Chunks beyond 2048 robots keep slot order.
__device__ __forceinline__ void finish_order_chunk(int x, int n, const int *__restrict__ ftime, int *__restrict__ order, int *hist /* >= 2048 ints of LDS */)
{
    const int chunk = (n + 7) >> 3;
    const int lo = x * chunk;
    const int hi = (lo + chunk < n) ? lo + chunk : n;
    const int m = hi - lo;
    if (m > 2048) { for (int i = threadIdx.x; i < m; i += blockDim.x) order[lo + i] = lo + i; return; }
    for (int i = threadIdx.x; i < m; i += blockDim.x) hist[i] = ftime[lo + i];
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int ti = hist[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) { const int dlt = hist[j] - ti; rank += (dlt < 0 || (dlt == 0 && j < i)) ? 1 : 0; }
        order[lo + rank] = lo + i;
    }
}
#ifndef QR_FLOPS_BUILD
__global__ void __launch_bounds__(256) qr_lpt_order_kernel(int n, const int *__restrict__ cost, int *__restrict__ order, const int *__restrict__ ftime, int *__restrict__ wbc_order)
{
    __shared__ int hist[2048];
    lpt_order_chunk(blockIdx.x, n, cost, order, hist);
    if (ftime && wbc_order) { __syncthreads(); finish_order_chunk(blockIdx.x, n, ftime, wbc_order, hist); }
}
#endif

// Global-memory arguments of one MPC launch (SoA, [field][robot]; see include/qrgpu.h)
struct MpcIO {
    const int *type_id;
    const float *g_state, *g_traj, *g_gait, *g_q;
    float *g_force, *g_tau;
    int *g_status;
    float *dbgH, *dbgG, *g_force_wbc;
    int force_stride;
    long long *dbgT;
};

// One robot's MPC tick by one workgroup of NTHR threads.  MAXB: 3x3 blocks a thread keeps in registers during the sweep (MAXB * NTHR >= number of
// stance leg-step pairs).  BIG: working-set positions 64 .. 95 live in a second set of per-lane registers.  Every wave returns from here
// (the workers at their exit command, wave 0 after the outputs), so a workgroup may solve several robots in a row (list mode below).
template <int MAXB, bool BIG, int NTHR, bool PERSIST = false, bool H16 = (MAXB > 4)>
__device__ __forceinline__ void mpc_solve_robot(const MpcLaunch &P, const MpcIO &io, const int rid, double *smem)
{
    int tid_ = threadIdx.x;
    // (the persistent kernel calls this in a loop: without the opaque copy everything that depends on the thread index alone is hoisted out of
    //  that loop and held in registers across the whole solve -- 98 spilled VGPRs in the 128-register main pass instead of 6)
    if (PERSIST) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    const int n = P.n;
    const long long t_begin = P.cost ? clock64() : 0;
    // Every workgroup barrier of the solve is counted (uniform, an SGPR): a workgroup of the persistent main pass keeps its waves beyond the
    // active set's four alive -- they must come back for the next robot -- and a live wave counts at every s_barrier of its workgroup, so
    // those waves cross barriers in step with the working ones until the count wave 0 leaves in sMisc[15] at the solve's last one (QR_IDLE).
    int nbar = 0;
#define QR_SYNC() do { __syncthreads(); if (PERSIST) ++nbar; } while (0)
#define QR_IDLE() do { for (;;) { QR_SYNC(); if (((volatile int *)sMisc)[15] == nbar) break; } } while (0)
    // a type id outside the table, or one that was never set up, would read garbage (mass 0 => 1/mass = inf): the robot is solved with the
    // first valid type's constants and carries QRGPU_ST_BAD_TYPE
    int tyid = io.type_id ? io.type_id[rid] : 0;
    const bool bad_type = tyid < 0 || tyid >= QR_MAX_TYPES || !((P.type_ready >> (tyid & (QR_MAX_TYPES - 1))) & 1);
    if (bad_type) tyid = __builtin_ctz(P.type_ready | (1 << QR_MAX_TYPES));
    const MpcType &C = P.type[tyid & (QR_MAX_TYPES - 1)];
    const int h = P.horizon;
    const int NV = 12 * h, NL = 4 * h;

    // ---------------- LDS carve (must match mpc_lds_fixed_bytes) ----------------
    double *gl = smem;                 // [NV] gradient (free variables, leg-step major)
    double *xz = gl + NV;              // xz[4][NV] partial x / z exchange (slot 0 carries d of the iteration)
    double *xr = xz + 4 * NV;          // xr[4][64] partial r exchange
    double *fmk = xr + 4 * 64;         // [NL] f_z upper bound per free leg-step
    float *sT = (float *)(fmk + NL);   // [4][3][3]
    float *sU = sT + 36;               // [4][3][3]
    float *sSt = sU + 36;              // [28]
    float *sTraj = sSt + 28;           // [12h]
    float *sGait = sTraj + NV;         // [4h]
    float *sV = sGait + NL;            // [13h]
    float *sJ = !H16 ? sV + 13 * h : nullptr;   // [12][3] columns of the leg Jacobians (the torque map of phase 6, computed while the data loads; h <= 11)
    int *sLs = (int *)(sV + 13 * h + (!H16 ? 36 : 0));   // [NL] free leg-step -> original leg-step
    int *sAct = sLs + NL;              // [QH] (unused since the single-wave loop went; keeps the carve of mpc_lds_fixed_bytes)
    short *sPos = (short *)(sAct + QR_QH);   // [6 NL] constraint id -> working-set position, or -1
    int *sMisc = (int *)(sPos + 6 * NL + ((6 * NL) & 1));   // [16]: [0] free leg-steps, [8..13] control block
    // (pointer arithmetic only: an integer round trip would drop the LDS address space and turn every access into flat_*)
    double *Mb = smem + (int)(mpc_lds_fixed_bytes(h, true) / 8);   // block-packed M; the sweep panels live here first

#define QR_TS(i) do { if (QR_DBGT && tid == 0) QR_DBGT[(size_t)rid * 16 + (i)] = clock64(); } while (0)
    QR_TS(0);
    if (QR_DBGT && tid == 0) QR_DBGT[(size_t)rid * 16 + 12] = wall_clock64();       // (the 100 MHz clock every CU shares: launch-wide concurrency, scratch/diag_util.py)
    // ---------------- phase 0: inputs ----------------
    if (tid == 0) sMisc[15] = -1;
    if (tid < 28) sSt[tid] = io.g_state[(size_t)tid * n + rid];
    for (int i = tid; i < NV; i += NTHR) sTraj[i] = io.g_traj[(size_t)i * n + rid];
    for (int i = tid; i < NL; i += NTHR) sGait[i] = io.g_gait[(size_t)i * n + rid];
    QR_SYNC();

    // ---------------- phase 1: SRBD terms ----------------
    // R = quat.toRotationMatrix() from the quaternion: phases 1-2 keep it in registers; the output phase, a whole active set later, forms it
    // again from the copy of the quaternion in sMisc[4..7] (nine VGPRs less to carry -- and spill -- through the 128-register main pass)
    auto quat_to_R = [](float w, float x, float y, float z, float (&R)[3][3]) {
#pragma clang fp contract(off)
        const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
        const float twx = tx * w, twy = ty * w, twz = tz * w;
        const float txx = tx * x, txy = ty * x, txz = tz * x;
        const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
        R[0][0] = 1.f - (tyy + tzz); R[0][1] = txy - twz;         R[0][2] = txz + twy;
        R[1][0] = txy + twz;         R[1][1] = 1.f - (txx + tzz); R[1][2] = tyz - twx;
        R[2][0] = txz - twy;         R[2][1] = tyz + twx;         R[2][2] = 1.f - (txx + tyy);
    };
    float R[3][3];
    quat_to_R(sSt[6], sSt[7], sSt[8], sSt[9], R);
    if (tid < 4) ((float *)sMisc)[4 + tid] = sSt[6 + tid];
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
        if (tid == 0) { sMisc[0] = __popcll(mask); sMisc[1] = 0; sMisc[2] = (int)(unsigned)mask; sMisc[3] = (int)(unsigned)(mask >> 32); }
        // The operand table of phase 2 (K4), one row per free variable e = 3 pos + i: T_p[0..2][i], dt U_p[0..2][i], (i_a | i << 8).  Lanes 0-3 of
        // this same wave have just written T and U, so a wave-level sync is all it takes -- the table is complete at the barrier below and the
        // tiles' lane constants are one LDS round trip (not a chain leg-step id -> leg -> T / U entry) and no barrier of their own away.
        wave_sync();
        if (fr) {
#pragma clang fp contract(off)
            const int pos = __popcll(mask & ((1ull << tid) - 1ull));
            const int p = tid & 3;
            float *sOp_ = (float *)xz;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float *o = sOp_ + 8 * (3 * pos + i);
                o[0] = sT[9 * p + i]; o[1] = sT[9 * p + 3 + i]; o[2] = sT[9 * p + 6 + i];
                o[3] = C.dt * sU[9 * p + i]; o[4] = C.dt * sU[9 * p + 3 + i]; o[5] = C.dt * sU[9 * p + 6 + i];
                ((int *)o)[6] = (tid >> 2) | (i << 8);
            }
        }
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
    QR_SYNC();
    QR_TS(1);
    // LDS loads count as divergent for the compiler; make the sizes scalar so that loops and branches on them are SALU
    const int nls = __builtin_amdgcn_readfirstlane(sMisc[0]);
    const int ns = 3 * nls;
    const int npairs = tri(nls);
    // (the eight-wave h > 11 kernel that runs two to a CU keeps S^-1 of every robot in the global scratch: typed as a global pointer its accesses are
    //  global_* with a scalar base and a 32-bit offset instead of flat_* on 64-bit addresses -- the generic pointer of the other H16 variants costs
    //  44 spilled VGPRs within 128 registers)
    constexpr bool SG = H16 && MAXB <= 4;
    typedef __attribute__((address_space(1))) double gdouble;
    typedef typename std::conditional<SG, gdouble *, double *>::type SinvPtr;
    SinvPtr Sinv = (SinvPtr)(Mb + npairs * 9);
    // BIG at a horizon whose phase 0-2 float arrays are too small for the second half of the r exchange (h < 14): 256 doubles at the very
    // end of the workgroup's LDS
    const bool xr2_in_floats = (100 + 29 * h) * 4 >= 2048;
    const int tail_doubles = (BIG && !xr2_in_floats) ? 256 : 0;
    constexpr int QMAX = BIG ? QR_QH : 64;          // working-set positions the per-lane registers hold
    int qcap;
    bool m_fits;                        // the block-packed inverse Hessian itself fits this launch's LDS allotment
    {   // rows of S^-1 that fit behind M
        const long long rem = (long long)(P.lds_bytes / 8) - (long long)(Mb - smem) - (long long)npairs * 9 - tail_doubles;
        m_fits = rem >= 0;
        int qc = 0;
        if (rem > 0) { qc = (int)((__builtin_sqrt(8.0 * (double)rem + 1.0) - 1.0) * 0.5); while (tri(qc) > rem) --qc; }
        qcap = qc < QMAX ? qc : QMAX;
        if (qcap > ns) qcap = ns;
    }
    bool spilled = false;
    if constexpr (H16) {
        // (the pointer is then generic and the S^-1 accesses of these variants compile to flat_* instructions: a few per cent at
        // h = 16, nothing at h <= 11 whose variants never take this branch)
        const int want = ns < QMAX ? ns : QMAX;
        if constexpr (SG) {
            if (P.sinv_spill && m_fits) { Sinv = (SinvPtr)(P.sinv_spill + (size_t)rid * (size_t)tri(QR_QH)); qcap = want; spilled = true; }
            else qcap = 0;                      // (to the list launches, below)
        } else {
            if (P.sinv_spill && m_fits && qcap < want && qcap < 64) { Sinv = (SinvPtr)(P.sinv_spill + (size_t)rid * (size_t)tri(QR_QH)); qcap = want; spilled = true; }
        }
    }
    // What is left behind S^-1 caches W_A = M N_A, one 3*nls vector per working-set position (the `w` of the iteration that added it),
    // so that z = w - W_A r needs no block products.  qW positions fit; the solve falls back to z = w - M (N_A r) for good once the
    // working set outgrows them.
    int qW = 0;
    double *Wc = nullptr;
    const int nsp = ns | 1;             // row stride of the cache (odd number of doubles)
    {
        const int qs = spilled ? 0 : qcap;
        const long long rem = (long long)(P.lds_bytes / 8) - (long long)(Mb - smem) - (long long)npairs * 9 - (long long)tri(qs) - tail_doubles;
        Wc = Mb + npairs * 9 + tri(qs);
        if (rem > 0 && ns > 0) qW = (int)(rem / (ns | 1));
        if (qW > 64) qW = 64;
        if (P.no_wcache == 1) qW = 0;
    }
    int st = bad_type ? QRGPU_ST_BAD_TYPE_D : 0;
    if (npairs > MAXB * NTHR) { st |= QRGPU_ST_MPC_OVERFLOW_D; }      // cannot happen: the host picks MAXB from the horizon
    if (!spilled && qcap < (ns < 24 ? ns : 24)) {
        // this launch's LDS allotment cannot hold the inverse Hessian of this robot plus a 24-row S^-1 (the main pass at three workgroups
        // per CU and an all-stance robot): nothing is computed here, the robot goes to the list pass at once -- and, through the `big`
        // bit, onto the planned list of the next call
        if (tid == 0) {
            if (io.g_status) st_xt(io.g_status + rid, st | QRGPU_ST_MPC_OVERFLOW_D, P.main_done != nullptr);      // (its rescuer may already be running: not behind that one's word)
            if (P.main_done) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QR_TRACE(rid, 4 | (P.rescue_mode << 8));
            if (P.rescue_list && !P.rescue_mode) st_xt(P.rescue_list + atomicAdd(P.rescue_count + P.rescue_parity, 1), rid, P.solved != nullptr);
            else if (P.solved) qr_epoch_raise(P.solved + rid, P.solved_epoch);      // (nobody solves it again this tick)
            if (P.cost) st_xt(P.cost + rid, 255 | (P.pre_list ? 256 : 0) | (0xff0 << 16), P.solved != nullptr);
            if (P.done_flag) __hip_atomic_store(P.done_flag + rid, (P.done_epoch << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (QR_P_FTIME) QR_P_FTIME[rid] = (int)wall_clock64();
        }
        return;
    }

    const float dtm = dt * minv;
    const float two_alpha = 2.f * C.alpha;

    // Executed-arithmetic accounting (QR_PFLOPS, off unless asked for): what the workgroup actually computes, by formula from the sizes the
    // solve sees -- [0] fp32 vector flops (operand generation, gradient), [1] fp32 matrix flops issued (2 * 16 * 16 * 4 per
    // v_mfma_f32_16x16x4_f32), [2] fp64 flops of the sweep and x0, [3] fp64 flops of the active set (rebuilds included).  mul and add count 1
    // each, fma 2.  Wave 0 keeps the sums (uniform) and stores them at the end.
    double fl_v32 = 0.0, fl_m32 = 0.0, fl_sw = 0.0, fl_as = 0.0;
    // ---------------- phase 2: Hessian (matrix cores -> fp32 tiles in LDS -> blocks in registers) + gradient (LDS) ----------------
    // (the torque map's Jacobian columns first, on twelve lanes of the last wave)
    // The torque map's Jacobian columns (AnalyticalLegJacobian, QS/robots/qr_robot.cpp:148-172) on twelve lanes of the second-to-last wave (the
    // snake below deals it the cheapest units): lane 3 leg + m takes the sine and cosine of ONE angle -- abad, hip + knee / 2, knee -- and
    // the three lanes of a leg trade them by shuffles, instead of every lane evaluating its column's six to eight sinf / cosf itself.
    if (!H16 && io.g_tau && (tid & ~63) == NTHR - 128) {
        const int l12 = lane < 12 ? lane : 0;
        const int leg = (l12 * 21846) >> 16, m = l12 - 3 * leg;
        const float t0 = io.g_q[(size_t)(3 * leg) * n + rid], t1 = io.g_q[(size_t)(3 * leg + 1) * n + rid], t2 = io.g_q[(size_t)(3 * leg + 2) * n + rid];
        const float tEff = t1 + t2 / 2;
        const float ang = (m == 0) ? t0 : ((m == 1) ? tEff : t2);
        const float sn = sinf(ang), cs = cosf(ang);
        const float s0 = __shfl(sn, 3 * leg, 64), c0 = __shfl(cs, 3 * leg, 64), sE = __shfl(sn, 3 * leg + 1, 64), cE = __shfl(cs, 3 * leg + 1, 64);
        const float s2 = __shfl(sn, 3 * leg + 2, 64), c2 = __shfl(cs, 3 * leg + 2, 64);
        const float lu = C.upper_l, ll = C.lower_l;
        const float sh = C.hip_l * ((leg & 1) ? 1.f : -1.f);
        const float lEff = sqrtf(lu * lu + ll * ll + 2 * lu * ll * c2);
        float J0, J1, J2;
        if (m == 0) {
            J0 = 0;
            J1 = -sh * s0 + lEff * c0 * cE;
            J2 = sh * c0 + lEff * s0 * cE;
        } else if (m == 1) {
            J0 = -lEff * cE;
            J1 = -lEff * s0 * sE;
            J2 = lEff * sE * c0;
        } else {
            J0 = ll * lu * s2 * sE / lEff - lEff * cE / 2;
            J1 = -ll * lu * s0 * s2 * cE / lEff - lEff * s0 * sE / 2;
            J2 = ll * lu * s2 * c0 * cE / lEff + lEff * sE * c0 / 2;
        }
        if (lane < 12) { sJ[3 * lane] = J0; sJ[3 * lane + 1] = J1; sJ[3 * lane + 2] = J2; }
    }
    int ba[MAXB], bb[MAXB];
#pragma unroll
    for (int sl = 0; sl < MAXB; ++sl) {
        const int pid = tid + NTHR * sl;
        ba[sl] = -1; bb[sl] = -1;
        if (pid < npairs) {
            int a = (int)((__builtin_sqrtf(8.f * (float)pid + 1.f) - 1.f) * 0.5f);
            while (tri(a + 1) <= pid) ++a;
            while (tri(a) > pid) --a;
            ba[sl] = a; bb[sl] = pid - tri(a);          // a >= b
        }
    }
    // K4 on the matrix cores:  qH = temp * Bqp (:411) restricted to the stance columns, as 16 x 16 tiles of v_mfma_f32_16x16x4_f32.
    // That instruction IS the k-ordered fp32 fmaf chain (MI355X_MICROARCH.md: "exact f32, == fmaf chain, bitwise"), so the result is bit
    // for bit the oracle's dense GEMM (tests/test_gpu_mpc.py::test_assembly_bit_exact); the terms it adds beyond the hand-written chain
    // are exact zeros.  k runs over (horizon step r, state row s): per step three instructions cover s = 0..11 (s = 12 has weight 0),
    // lane group g = lane >> 4 supplying s = 4 q + g of instruction q.  Operands are generated in registers from the closed form of
    // Adt^a Bdt (no Bqp in memory): G[(r, s)][(a, i)] = c * T_p[s][i] | c / m | dt U_p[s - 6][i] | dt / m, c = (r - i_a + 1/2) dt^2, zero for r < i_a.
    //
    // Round 3 form.  The stated QP needs H[x][y] AND H[y][x] (fp32 rounds them differently; (H + H') / 2 is taken exactly in fp64).  A UNIT of
    // work is one accumulator chain of one tile on or below the tile diagonal:
    //   off-diagonal tile (R, C), C < R:  unit 0:  D[row][col] = sum temp[e_r][k] G[k][e_c] = H[e_r][e_c]
    //                                     unit 1:  D[row][col] = sum G[k][e_r] temp[e_c][k] = H[e_c][e_r]   (same lane and register as unit 0's)
    //   diagonal tile (R, R):             ONE unit: the tile holds both H[e_r][e_c] and, at the transposed position, H[e_c][e_r] -- the second
    //                                     chain round 2 ran there repeated the same products in the same order (a * b = b * a): a quarter of
    //                                     all matrix instructions, and the most expensive tiles (tile (0, 0) runs over every horizon step)
    // A wave takes a whole tile (both chains of an off-diagonal one share every product but the weights: 29 vector instructions per step for
    // the two, against 2 x 22 apart -- the phase is bound by instruction issue, four waves to a SIMD, not by the matrix pipe); tiles are
    // enumerated row by row (cost h - first step of the row: descending) and dealt to the waves in snake order; each chain
    // leaves its 16 x 16 fp32 accumulator in LDS (the M region, idle until the sweep) and the block owners form (H + H') / 2 from there as they
    // load their blocks -- no fp64 conversion, division by three or scattered 8-byte store behind the matrix instructions any more.
    // The lane constants of a tile side come from a per-variable table (sOp, in the xz exchange area, idle until phase 4) built once per
    // robot, instead of a chain of dependent LDS reads (leg-step id -> leg -> T / U entries) per tile and side.
    const int NT = (ns + 15) >> 4;
    const int NOD = (NT * (NT - 1)) >> 1;                  // off-diagonal tiles below the diagonal
    float *Hs = (float *)Mb;                               // [2 NOD + NT][16][16]: buffers 2 t, 2 t + 1 of off-diagonal tile t = tri(R - 1) + C, then one per diagonal tile
    float *sOp = (float *)xz;                              // [ns][8]: T_p[0..2][i], dt U_p[0..2][i], (i_a | i << 8), -
    if ((long long)NT * NT * 1024 > (long long)P.lds_bytes - (long long)((Mb - smem) * 8)) {
        // (h = 11 all stance in the main pass's half-CU allotment: to the list pass, like a robot whose S^-1 does not fit)
        if (tid == 0) {
            if (io.g_status) st_xt(io.g_status + rid, st | QRGPU_ST_MPC_OVERFLOW_D, P.main_done != nullptr);      // (its rescuer may already be running: not behind that one's word)
            if (P.main_done) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QR_TRACE(rid, 4 | (P.rescue_mode << 8));
            if (P.rescue_list && !P.rescue_mode) st_xt(P.rescue_list + atomicAdd(P.rescue_count + P.rescue_parity, 1), rid, P.solved != nullptr);
            else if (P.solved) qr_epoch_raise(P.solved + rid, P.solved_epoch);      // (nobody solves it again this tick)
            if (P.cost) st_xt(P.cost + rid, 255 | (P.pre_list ? 256 : 0) | (0xff0 << 16), P.solved != nullptr);
            if (P.done_flag) __hip_atomic_store(P.done_flag + rid, (P.done_epoch << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (QR_P_FTIME) QR_P_FTIME[rid] = (int)wall_clock64();
        }
        return;
    }
    {
#pragma clang fp contract(off)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int wvb = __builtin_amdgcn_readfirstlane(tid >> 6);
        constexpr int NW = NTHR / 64;
        const int g = lane >> 4, lc = lane & 15;
        const float w2q0 = 2.f * C.weights[g], w2q1 = 2.f * C.weights[4 + g], w2q2 = 2.f * C.weights[8 + g];
        float *Hd = QR_DBGH ? QR_DBGH + (size_t)rid * NV * NV : nullptr;
        // lane constants of one side (row tile or column tile): the variable behind index e = 16 * tile + lc and its operand recipe
        auto side = [&](int tile, int &ia, float &al0, float &al1, float &k1, float &k2) {
            const int e = 16 * tile + lc;
            const bool valid = e < ns;
            const float *o = sOp + 8 * (valid ? e : 0);
            const int meta = ((const int *)o)[6];
            const float o0 = o[g < 3 ? g : 2], o1 = o[3 + (g >= 2 ? g - 2 : 0)], o2 = o[5];
            const int i = meta >> 8;
            ia = valid ? (meta & 255) : h;                         // (rows / columns past the matrix never switch on)
            al0 = (g < 3) ? o0 : ((i == 0) ? minv : 0.f);
            al1 = (i == g + 1) ? minv : 0.f;                       // used by lane groups 0, 1 only (s = 4, 5)
            k1 = o1;                                               // s = 6, 7 for lane groups 2, 3
            k2 = (g == 0) ? o2 : ((i == g - 1) ? dtm : 0.f);       // s = 8 | 9, 10, 11
        };
        const int NU = (NT * (NT + 1)) >> 1;            // one unit per tile on or below the tile diagonal
        // Deal.  Units come tile row by tile row, i.e. in order of descending cost (a row's chains run over h - first step of the row).  The
        // last two waves carry the gradient (below) and the Jacobian columns, so the first NW - 2 units go to waves 0 .. NW - 3, and the rest
        // go back and forth over waves NW - 1 .. 1 (snake) -- wave 0 keeps tile (0, 0) alone, the one chain that runs over every horizon step.
        constexpr int NF = NW > 2 ? NW - 2 : NW, per = 2 * (NW - 1);
        for (int it = 0; it == 0 || (wvb != 0 && NF + (it - 1) * per < NU); ++it) {
          for (int half = 0; half < 2; ++half) {
            int u;
            if (it == 0) { if (half || wvb >= NF) continue; u = wvb; }
            else u = NF + (it - 1) * per + (half == 0 ? NW - 1 - wvb : NW - 2 + wvb);
            if (u >= NU) continue;
            int R = (int)((__builtin_sqrtf(8.f * (float)u + 1.f) - 1.f) * 0.5f);
            while (tri(R + 1) <= u) ++R;
            while (tri(R) > u) --R;
            const int Cc = u - tri(R);
            const bool diag = Cc == R;
            int iaR, iaC;
            float a0R, a1R, k1R, k2R, a0C, a1C, k1C, k2C;
            side(R, iaR, a0R, a1R, k1R, k2R);
            side(Cc, iaC, a0C, a1C, k1C, k2C);
            // the weighted (temp = G * 2w, rounding order (dt U) * 2w as the reference's) forms of the constants
            const float t1R = k1R * w2q1, t2R = k2R * w2q2, t1C = k1C * w2q1, t2C = k2C * w2q2;
            const int r0 = __builtin_amdgcn_readfirstlane(((const int *)(sOp + 8 * 16 * R))[6]) & 255;      // first step at which any entry of the tile switches on
            f4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
            const float bR = 0.5f - (float)iaR, bC = 0.5f - (float)iaC;      // (r - i_a) + 1/2 is exact in fp32 either way
            if (P.hess_mode == 1) {
                // BASELINE.json configs[4]'s arithmetic: the same contraction on the bf16 matrix cores.  Every fp32 operand is cut into three
                // bf16 limbs (8 + 8 + 8 significant bits: x = hi + mid + lo exactly up to the last limb's rounding) and the product a * b is taken as
                // the six cross terms hh + hm + mh + hl + mm + lh (what is left out is below 2^-24 of it), summed in fp32 by
                // v_mfma_f32_16x16x32_bf16: one instruction covers the same four state rows s = 4 q + g as the fp32 instruction it replaces, its
                // K = 32 being 4 lane groups x (6 terms + 2 zeros).  NOT bit-identical to the fp32 chain (another summation order and tree):
                // tolerance in tests/test_gpu_mpc.py::test_bf16x3_hessian.  Operand generation (the limb cuts) dominates here, so this mode is
                // slower than the exact one on this problem; it exists because that configuration names it.
                typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
                union PK { unsigned u[4]; bf8 v; };
                auto limbs = [](float x, unsigned &hi, unsigned &mid, unsigned &lo) {
                    const __bf16 bh = (__bf16)x;
                    const float r1 = x - (float)bh;
                    const __bf16 bm = (__bf16)r1;
                    const float r2 = r1 - (float)bm;
                    const __bf16 bl = (__bf16)r2;
                    hi = (unsigned)__builtin_bit_cast(unsigned short, bh); mid = (unsigned)__builtin_bit_cast(unsigned short, bm); lo = (unsigned)__builtin_bit_cast(unsigned short, bl);
                };
                auto packA = [&](float x) { unsigned a, b, c; limbs(x, a, b, c); PK p; p.u[0] = a | (a << 16); p.u[1] = b | (a << 16); p.u[2] = b | (c << 16); p.u[3] = 0u; return p.v; };   // hi hi mid hi mid lo 0 0
                auto packB = [&](float x) { unsigned a, b, c; limbs(x, a, b, c); PK p; p.u[0] = a | (b << 16); p.u[1] = a | (c << 16); p.u[2] = b | (a << 16); p.u[3] = 0u; return p.v; };   // hi mid hi lo mid hi 0 0
                // the operands that do not change with the horizon step are cut once per tile
                const bf8 Ak2g = packA(k2R), Ak2t = packA(t2R), Bk2g = packB(k2C), Bk2t = packB(t2C);
                const bf8 Ak1g = packA(k1R), Ak1t = packA(t1R), Bk1g = packB(k1C), Bk1t = packB(t1C);
                const bf8 zero8 = packA(0.f);
                for (int r = r0; r < h; ++r) {
                    const bool onR = r >= iaR, onC = r >= iaC;
                    const float rf = (float)r;
                    const float cR = (rf + bR) * dt2, cC = (rf + bC) * dt2;
                    float gR = cR * a0R, gC = cC * a0C;
                    float tR = gR * w2q0, tC = gC * w2q0;
                    gR = onR ? gR : 0.f; tR = onR ? tR : 0.f; gC = onC ? gC : 0.f; tC = onC ? tC : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(packA(tR), packB(gC), acc1, 0, 0, 0);
                    if (!diag) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(packA(gR), packB(tC), acc2, 0, 0, 0);
                    const float hR = cR * a1R, hC = cC * a1C;
                    const bf8 a1t = (g < 2) ? packA(onR ? hR * w2q1 : 0.f) : (onR ? Ak1t : zero8);
                    const bf8 b1g = (g < 2) ? packB(onC ? hC : 0.f) : (onC ? Bk1g : zero8);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1t, b1g, acc1, 0, 0, 0);
                    if (!diag) {
                        const bf8 a1g = (g < 2) ? packA(onR ? hR : 0.f) : (onR ? Ak1g : zero8);
                        const bf8 b1t = (g < 2) ? packB(onC ? hC * w2q1 : 0.f) : (onC ? Bk1t : zero8);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1g, b1t, acc2, 0, 0, 0);
                    }
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(onR ? Ak2t : zero8, onC ? Bk2g : zero8, acc1, 0, 0, 0);
                    if (!diag) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(onR ? Ak2g : zero8, onC ? Bk2t : zero8, acc2, 0, 0, 0);
                }
            } else if (diag) {
                // diagonal tile: one chain, D[row][col] = sum temp[e_r][k] G[k][e_c] = H[e_r][e_c] for BOTH triangles of the tile
                for (int r = r0; r < h; ++r) {
                    const bool onR = r >= iaR;
                    const float cR = ((float)r + bR) * dt2;
                    float gR = cR * a0R;
                    float tR = gR * w2q0;
                    gR = onR ? gR : 0.f; tR = onR ? tR : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tR, gR, acc1, 0, 0, 0);
                    const float hR = cR * a1R;
                    gR = (g < 2) ? hR : k1R; tR = (g < 2) ? hR * w2q1 : t1R;
                    gR = onR ? gR : 0.f; tR = onR ? tR : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tR, gR, acc1, 0, 0, 0);
                    gR = onR ? k2R : 0.f; tR = onR ? t2R : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tR, gR, acc1, 0, 0, 0);
                }
            } else {
                // off-diagonal tile: both chains, sharing every product but the weights:  acc1 = H[e_r][e_c],  acc2[row][col] = H[e_c][e_r]
                for (int r = r0; r < h; ++r) {
                    const bool onR = r >= iaR, onC = r >= iaC;
                    const float rf = (float)r;
                    const float cR = (rf + bR) * dt2, cC = (rf + bC) * dt2;
                    // s = 0..3
                    float gR = cR * a0R, gC = cC * a0C;
                    float tR = gR * w2q0, tC = gC * w2q0;
                    gR = onR ? gR : 0.f; tR = onR ? tR : 0.f; gC = onC ? gC : 0.f; tC = onC ? tC : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tR, gC, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gR, tC, acc2, 0, 0, 0);
                    // s = 4..7
                    const float hR = cR * a1R, hC = cC * a1C;
                    gR = (g < 2) ? hR : k1R; tR = (g < 2) ? hR * w2q1 : t1R;
                    gC = (g < 2) ? hC : k1C; tC = (g < 2) ? hC * w2q1 : t1C;
                    gR = onR ? gR : 0.f; tR = onR ? tR : 0.f; gC = onC ? gC : 0.f; tC = onC ? tC : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tR, gC, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gR, tC, acc2, 0, 0, 0);
                    // s = 8..11
                    gR = onR ? k2R : 0.f; tR = onR ? t2R : 0.f; gC = onC ? k2C : 0.f; tC = onC ? t2C : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tR, gC, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gR, tC, acc2, 0, 0, 0);
                }
            }
            // D[row = 4 g + reg][col = lc] -> the tile's buffer(s), 4-byte stores, lanes lc contiguous.  Row r sits at physical row
            // hs_row(r, Cc) = (4 (r & 3) + (r >> 2)) ^ (Cc & 1): the lane groups g = 0, 1 of a store land in different halves of the 32
            // banks (rows 4 apart would share them), and the block owners' reads, which run across column tiles, alternate halves with the tile
            float *buf = Hs + 256 * (diag ? 2 * NOD + R : 2 * (((R * (R - 1)) >> 1) + Cc));
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) buf[((4 * reg + g) ^ (Cc & 1)) * 16 + lc] = acc1[reg];
            if (!diag) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) buf[256 + ((4 * reg + g) ^ (Cc & 1)) * 16 + lc] = acc2[reg];
            }
            if (Hd) {
                const int ec = 16 * Cc + lc;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int er = 16 * R + 4 * g + reg;
                    if (er < ns && ec < ns) {
                        const int aq = (er * 21846) >> 16, bq = (ec * 21846) >> 16;
                        const int xr_ = 3 * sLs[aq] + (er - 3 * aq), xc_ = 3 * sLs[bq] + (ec - 3 * bq);
                        Hd[(size_t)xr_ * NV + xc_] = acc1[reg] + ((er == ec) ? two_alpha : 0.f);             // + 2 alpha I (:411)
                        if (!diag) Hd[(size_t)xc_ * NV + xr_] = acc2[reg];
                    }
                }
            }
          }
        }
    }
    // gradient: qg[a] = sum_k temp[a][k] v[k], one free variable per thread -- on the last two waves (variable e and, beyond 128 of them, e + 128 per thread), which the deal above leaves out of its first pass
    for (int e = tid - (NTHR - 128); e >= 0 && e < ns; e += 128) {
#pragma clang fp contract(off)
        const int ls = sLs[e / 3], j = e % 3, ia = ls >> 2, p = ls & 3;
        const float t0 = sT[9 * p + j], t1 = sT[9 * p + 3 + j], t2 = sT[9 * p + 6 + j];
        const float wg0 = 2.f * C.weights[0], wg1 = 2.f * C.weights[1], wg2 = 2.f * C.weights[2], wg3 = 2.f * C.weights[3 + j], wg9 = 2.f * C.weights[9 + j];
        const float u0 = (dt * sU[9 * p + j]) * (2.f * C.weights[6]), u1 = (dt * sU[9 * p + 3 + j]) * (2.f * C.weights[7]), u2 = (dt * sU[9 * p + 6 + j]) * (2.f * C.weights[8]);
        const float dw = dtm * wg9;
        float acc = 0.f;
        const float bA = 0.5f - (float)ia;
        for (int r = 0; r < h; ++r) {               // (uniform trip count: the step's broadcast loads of v pipeline; steps before i_a add nothing)
            const float ca = ((float)r + bA) * dt2;
            const float *vr = sV + 13 * r;
            float a2 = acc;
            a2 = __builtin_fmaf((ca * t0) * wg0, vr[0], a2);
            a2 = __builtin_fmaf((ca * t1) * wg1, vr[1], a2);
            a2 = __builtin_fmaf((ca * t2) * wg2, vr[2], a2);
            a2 = __builtin_fmaf((ca * minv) * wg3, vr[3 + j], a2);
            a2 = __builtin_fmaf(u0, vr[6], a2);
            a2 = __builtin_fmaf(u1, vr[7], a2);
            a2 = __builtin_fmaf(u2, vr[8], a2);
            a2 = __builtin_fmaf(dw, vr[9 + j], a2);
            acc = (r >= ia) ? a2 : acc;
        }
        gl[e] = (double)acc;
        if (QR_DBGG) QR_DBGG[(size_t)rid * NV + 3 * ls + j] = acc;
    }
    for (int c = tid; c < 6 * nls; c += NTHR) sPos[c] = -1;
    QR_TS(2);

    // ---------------- phase 3: symmetric block sweep in registers,  A <- -H^-1 ----------------
    // Pivot leg-step k:  P = A_kk,  C_i = A_ik (i > k) or A_ki' (i < k);
    //   A_ij <- A_ij - C_i P^-1 C_j',   A_ik <- C_i P^-1,   A_kk <- -P^-1.
    // The pivot column is exchanged through a double-buffered LDS panel: one barrier per pivot.
    Blk A[MAXB];
    QR_SYNC();               // every unit's tile is in LDS
    // block (a, b), a >= b, of (H + H') / 2: entry (i, j) from H[x][y] and H[y][x], x = 3 a + i >= y = 3 b + j (the block's lower half when
    // a = b), both fp32, averaged exactly in fp64.  Off-diagonal tile (R, C): buffers at 512 (tri(R - 1) + C), H[x][y] at [x & 15][y & 15] of
    // the first, H[y][x] at the same place of the second; diagonal tile: one buffer, H[y][x] at the transposed place.
#pragma unroll
    for (int sl = 0; sl < MAXB; ++sl) {
        if (ba[sl] >= 0) {
            const int a = ba[sl], b = bb[sl];
            int rowb[3], rr[3], rp[3], Rx[3], colb[3], cc[3], cp[3], Cy[3];          // (rp, cp: physical rows before the column tile's parity flip)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int x = 3 * a + i, y = 3 * b + i;
                Rx[i] = x >> 4; rr[i] = x & 15; rp[i] = 4 * (rr[i] & 3) + (rr[i] >> 2); rowb[i] = 256 * Rx[i] * (Rx[i] - 1);
                Cy[i] = y >> 4; cc[i] = y & 15; cp[i] = 4 * (cc[i] & 3) + (cc[i] >> 2); colb[i] = 512 * Cy[i] + cc[i];
            }
            const int dbase = 512 * NOD;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (a == b && j > i) continue;                              // (mirrored below)
                    const bool od = Rx[i] != Cy[j];
                    const int dg = dbase + 256 * Rx[i], par = Cy[j] & 1;
                    const int a1 = od ? rowb[i] + colb[j] + 16 * (rp[i] ^ par) : dg + 16 * (rp[i] ^ par) + cc[j];
                    const int a2 = od ? a1 + 256 : dg + 16 * (cp[j] ^ par) + rr[i];
                    float v1 = Hs[a1], v2 = Hs[a2];
                    if (a == b && i == j) {
#pragma clang fp contract(off)
                        v1 = v1 + two_alpha; v2 = v1;                          // + 2 alpha I (:411), in fp32 as the reference adds it
                    }
                    const double o = 0.5 * ((double)v1 + (double)v2);
                    A[sl].m[3 * i + j] = o;
                    if (a == b) A[sl].m[3 * j + i] = o;
                }
        }
    }
    QR_SYNC();               // every block is in registers: the M region can now carry the pivot panels
    // A wave beyond the active set's four that owns no block (a trotting robot's 300 blocks fill 4.7 of the 8 waves) is done: it would only
    // load panels and wait at barriers (a wave that has ended no longer counts there), competing for the LDS pipe with the ones that work.
    if (NTHR > QR_AS_THREADS && tid >= QR_AS_THREADS && (tid & ~63) >= npairs) { if (PERSIST) QR_IDLE(); return; }
    {
        double *panel0 = Mb, *panel1 = Mb + NL * 9;
#define VS_STAMP(i) do { } while (0)
        // the pivot column of step kk out of block sl: block (a, kk), a > kk, is C_a; block (kk, b), b < kk, is C_b'; the owner of the
        // pivot block (kk, kk) publishes P^-1 (3x3 symmetric, adjugate / determinant) in its place: once per pivot, not once per thread
        auto write_panel = [&](int sl, int kk, double *pn) {
            if (bb[sl] == kk) {
                if (ba[sl] == kk) {
                    const double *Pk = A[sl].m;
                    const double p00 = Pk[0], p01 = 0.5 * (Pk[1] + Pk[3]), p02 = 0.5 * (Pk[2] + Pk[6]), p11 = Pk[4], p12 = 0.5 * (Pk[5] + Pk[7]), p22 = Pk[8];
                    const double c00 = p11 * p22 - p12 * p12, c01 = p02 * p12 - p01 * p22, c02 = p01 * p12 - p02 * p11;
                    const double c11 = p00 * p22 - p02 * p02, c12 = p01 * p02 - p00 * p12, c22 = p00 * p11 - p01 * p01;
                    const double det = p00 * c00 + p01 * c01 + p02 * c02;
                    if (!(det > 0.0) || !(p00 > 0.0)) sMisc[1] = 1;                  // -> QRGPU_ST_MPC_NOTSPD, picked up by thread 0 after the sweep
                    const double id = QR_RCP_PIVOT(det);
                    double *d = pn + 9 * kk;
                    d[0] = c00 * id; d[1] = c01 * id; d[2] = c02 * id; d[4] = c11 * id; d[5] = c12 * id; d[8] = c22 * id;
                } else {
#pragma unroll
                    for (int i = 0; i < 9; ++i) pn[9 * ba[sl] + i] = A[sl].m[i];
                }
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
            QR_SYNC();
            VS_STAMP(1);
            // the first block's operands do not depend on P^-1: their LDS round trip hides behind its computation
            double Cx0[9], Cb0[9];
            {
                const int a0 = ba[0] < 0 ? 0 : ba[0], b0 = ba[0] < 0 ? 0 : bb[0];
                const double *Cx = pan + 9 * (a0 == k ? b0 : a0), *Cb = pan + 9 * b0;
#pragma unroll
                for (int i = 0; i < 9; ++i) { Cx0[i] = Cx[i]; Cb0[i] = Cb[i]; }
            }
            // P^-1 as its owner published it (upper triangle)
            double Pi[9];
            {
                const double *Pk = pan + 9 * k;
                Pi[0] = Pk[0]; Pi[1] = Pk[1]; Pi[2] = Pk[2]; Pi[4] = Pk[4]; Pi[5] = Pk[5]; Pi[8] = Pk[8];
                Pi[3] = Pi[1]; Pi[6] = Pi[2]; Pi[7] = Pi[5];
            }
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
                    for (int j = 0; j < 3; ++j) D[3 * i + j] = __builtin_fma(Cx[3 * i + 2], Pi[6 + j], __builtin_fma(Cx[3 * i + 1], Pi[3 + j], Cx[3 * i] * Pi[j]));
                if (a != k && b != k) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            A[sl].m[3 * i + j] = __builtin_fma(-D[3 * i + 2], Cb[3 * j + 2], __builtin_fma(-D[3 * i + 1], Cb[3 * j + 1], __builtin_fma(-D[3 * i], Cb[3 * j], A[sl].m[3 * i + j])));     // three chained fma, no separate mul / sub
                } else {
                    // pivot row / column: A_xk = D (x > k), A_kx = D' (x < k): the diagonal as it is, one select per off-diagonal pair ...
                    const bool tr = (a == k);
                    A[sl].m[0] = D[0]; A[sl].m[4] = D[4]; A[sl].m[8] = D[8];
                    A[sl].m[1] = tr ? D[3] : D[1]; A[sl].m[3] = tr ? D[1] : D[3];
                    A[sl].m[2] = tr ? D[6] : D[2]; A[sl].m[6] = tr ? D[2] : D[6];
                    A[sl].m[5] = tr ? D[7] : D[5]; A[sl].m[7] = tr ? D[5] : D[7];
                    // ... and A_kk = -P^-1 in the one lane that owns it (every other wave skips the branch)
                    if (a == k && b == k) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) A[sl].m[i] = -Pi[i];
                    }
                }
                // this block is final for step k: if it lies in the next pivot's column, publish it now (the other panel: its readers
                // passed this step's barrier), so the copy overlaps the remaining blocks instead of being a phase of its own
                write_panel(sl, k + 1, pnext);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        QR_SYNC();           // everybody is done with the panels before M overwrites them
#pragma unroll
        for (int sl = 0; sl < MAXB; ++sl) {
            if (ba[sl] >= 0) {
                double *dst = Mb + (tri(ba[sl]) + bb[sl]) * 9;
#pragma unroll
                for (int i = 0; i < 9; ++i) dst[i] = -A[sl].m[i];         // M = +H^-1
            }
        }
        QR_SYNC();
    }
    // phases 0-3 may run on more than four waves (NTHR / 64: one block of a trotting robot's Hessian per thread); the active set is a
    // four-wave protocol, so the others are done here (a wave that has ended no longer counts at the workgroup's barriers)
    if (NTHR > QR_AS_THREADS && tid >= QR_AS_THREADS) { if (PERSIST) QR_IDLE(); return; }
    if (sMisc[1]) st |= QRGPU_ST_MPC_NOTSPD_D;
    QR_TS(3);
    if (QR_PFLOPS) {
        double steps = 0.0;                              // chain x horizon-step pairs: two chains per off-diagonal tile, one per diagonal tile
        for (int Rf = 0; Rf < NT; ++Rf) steps += (double)(2 * Rf + 1) * (double)(h - (sLs[(16 * Rf * 21846) >> 16] >> 2));
        fl_m32 = steps * 3.0 * 2048.0;                   // 3 instructions per chain and step, 2 * 16 * 16 * 4 flops each
        fl_v32 = steps * 64.0 * 8.0;                     // operand generation: ~8 mul / add per lane, chain and step
        double gsteps = 0.0;
        for (int kf = 0; kf < nls; ++kf) gsteps += (double)(h - (sLs[kf] >> 2));
        fl_v32 += gsteps * 3.0 * 24.0;                   // gradient: 8 fma + 8 mul per variable and step
        fl_sw = (double)nls * ((double)npairs * 99.0 + 40.0) + (double)nls * (double)nls * 15.0;     // block sweep (D = C P^-1: 45, update: 54), P^-1 once per pivot; x0 = -M g
    }
    // =====================================================================================================
    // Control / worker active set.  Wave 0 alone takes the decisions -- row scan, w, delta, d, step lengths,
    // bookkeeping in its registers -- and waves 1-3 are linear-algebra helpers, so that the S^-1 border of one iteration runs while
    // wave 0 already scans for the next row, and the z partials run while wave 0 reduces the step lengths:
    //     wave 0:   scan, w, delta, d  ->X1-> r partial ->B2-> r, dr, t1, t2, t ->B3-> z, x, u, bookkeeping -> scan ...
    //     wave 1-3: [S^-1 border / downdate of the previous iteration] ->X1-> r partial ->B2-> r, z partial ->B3-> S^-1 border ...
    // Three workgroup barriers per working-set change (X1 also publishes the S^-1 update); the control block in LDS carries
    // {command, q, flags, dropped position, 1/z'c}, d travels through xz[0], the row positions of every leg-step through sPos.
    // =====================================================================================================
    {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool own = lane < nls;
        const int kme = own ? lane : 0;
        const double im = (double)(1.f / C.mu);
        const double fmaxk = own ? fmk[kme] : 0.0;
        const int tril = tri(lane);
        const double tol = 1e-9;
        const double INF = __builtin_inf();
        // BIG variants: working-set positions 64 .. 95 live in a SECOND set of per-lane registers (position lane + 64), so a solve that
        // outgrows the 64 lanes stays in this loop.  The second half of the r exchange lives in the float arrays of phases 0-2, dead by
        // now (2.2 KB at h = 16), or at the very end of the workgroup's LDS when those are too small (h < 14).
        const bool big2 = BIG && qcap > 64;
        double *xr2 = xr2_in_floats ? (double *)sT : smem + (P.lds_bytes / 8 - 256);
        const int tril2 = tri(lane + 64);
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
            QR_SYNC();
            if (own) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { x0 -= xz[v * NV + 3 * kme]; x1 -= xz[v * NV + 3 * kme + 1]; x2 -= xz[v * NV + 3 * kme + 2]; }
                // the unconstrained optimum stays in LDS (g is dead: every wave read it in front of the barrier above): a re-factorisation or a
                // warm start solves the equality-constrained problem on a whole working set from it
                if (wv == 0) { gl[3 * kme] = x0; gl[3 * kme + 1] = x1; gl[3 * kme + 2] = x2; }
            }
            QR_SYNC();
        }
        QR_TS(4);
        bool fastz = qW > 0;
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
        // ---- S^-1 (and the W_A cache) of a WHOLE working set at once: sAct[0..q) = 6 * leg-step + row per position.
        //   S_ij = c_i' M[k_i, k_j] c_j (one block of M per pair), then an in-place symmetric sweep of the packed triangle (one barrier per
        //   pivot: the next pivot's column is published by the owners of its elements while they update them), then W_A = M N_A while it fits.
        // Used for a warm start (last tick's working set, instead of adding its rows one iteration at a time) and to re-factorise when the
        // exit check finds that S^-1 -- otherwise kept by bordered updates / downdates only -- has drifted.  Every wave calls it (barriers
        // inside); returns false, for everybody alike, when a pivot shows the rows to be dependent.
        auto rebuild = [&](const int q) -> bool {
            // thread t owns the packed elements e = t, t + 256, ... (NE at most) and keeps them in registers through the whole sweep; only the
            // pivot column goes through LDS (colp, double buffered: the owners of the next pivot's column publish it while they update it)
            constexpr int NE = (QMAX * (QMAX + 1) / 2 + QR_AS_THREADS - 1) / QR_AS_THREADS;        // 9 (64 rows), 19 (96 rows)
            const int ne = tri(q);
            double *colp = xr;                            // [2][QMAX] pivot columns (the r exchange is idle during a rebuild)
            double *diag0 = xz + NV;                      // [q] the diagonal of S (q <= 3 nls <= NV)
            // (the 128-register h > 11 variant keeps the elements a 67-row set needs in registers, as the 64-row variants do; what a larger set
            //  adds is swept in place in S^-1's own storage -- slower per pivot, for the one robot in a hundred that gets there unannounced)
            constexpr int NER = (BIG && H16 && MAXB <= 4) ? 9 : NE;
            auto elem_ij = [&](const int e, int &i, int &j) {
                i = (int)((__builtin_sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
                while (tri(i + 1) <= e) ++i;
                while (tri(i) > e) --i;
                j = e - tri(i);
            };
            double el[NER];
            int eij[NER];
            if constexpr (NER < NE) {
                for (int m = NER; QR_AS_THREADS * m < ne; ++m) {
                    const int e = tid + QR_AS_THREADS * m;
                    if (e < ne) {
                        int i, j; elem_ij(e, i, j);
                        const int ci = sAct[i], cj = sAct[j];
                        const int ki = (ci * 10923) >> 16, kj = (cj * 10923) >> 16;
                        Blk B; load_block(Mb, ki, kj, B);
                        double a0, a1, a2, b0, b1, b2;
                        cons_vec(ci - 6 * ki, im, a0, a1, a2); cons_vec(cj - 6 * kj, im, b0, b1, b2);
                        const double v = a0 * (B.m[0] * b0 + B.m[1] * b1 + B.m[2] * b2) + a1 * (B.m[3] * b0 + B.m[4] * b1 + B.m[5] * b2)
                                       + a2 * (B.m[6] * b0 + B.m[7] * b1 + B.m[8] * b2);
                        Sinv[e] = v;
                        if (j == 0) colp[i] = v;
                        if (i == j) diag0[i] = v;
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < NER; ++m) {
                el[m] = 0.0; eij[m] = 0;
                if (QR_AS_THREADS * m < ne) {                                  // (uniform)
                    int te = tid;
                    asm volatile("" : "+v"(te));                               // (opaque: the (i, j) of this thread's first element is otherwise hoisted out of the solve's loop and spilled)
                    const int e = te + QR_AS_THREADS * m;
                    if (e < ne) {
                        int i = (int)((__builtin_sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
                        while (tri(i + 1) <= e) ++i;
                        while (tri(i) > e) --i;
                        const int j = e - tri(i);
                        const int ci = sAct[i], cj = sAct[j];
                        const int ki = (ci * 10923) >> 16, kj = (cj * 10923) >> 16;        // / 6 for ids < 6 * 64
                        Blk B; load_block(Mb, ki, kj, B);
                        double a0, a1, a2, b0, b1, b2;
                        cons_vec(ci - 6 * ki, im, a0, a1, a2); cons_vec(cj - 6 * kj, im, b0, b1, b2);
                        const double v = a0 * (B.m[0] * b0 + B.m[1] * b1 + B.m[2] * b2) + a1 * (B.m[3] * b0 + B.m[4] * b1 + B.m[5] * b2)
                                       + a2 * (B.m[6] * b0 + B.m[7] * b1 + B.m[8] * b2);
                        el[m] = v; eij[m] = (i << 8) | j;
                        if (j == 0) colp[i] = v;
                        if (i == j) diag0[i] = v;
                    }
                }
            }
            QR_SYNC();
            bool ok = true;
            for (int p = 0; p < q; ++p) {
                const double *cur = colp + (p & 1) * QMAX;
                double *nxt = colp + ((p & 1) ^ 1) * QMAX;
                const double d = cur[p];
                if (!(d > 1e-11 * diag0[p])) { ok = false; break; }          // (the same value in every thread: a uniform exit)
                const double ip = QR_RCP_PIVOT(d);
                if constexpr (NER < NE) {
                    for (int m = NER; QR_AS_THREADS * m < ne; ++m) {
                        const int e = tid + QR_AS_THREADS * m;
                        if (e < ne) {
                            int i, j; elem_ij(e, i, j);
                            const double ci = cur[i], cj = cur[j];
                            const bool ip_ = (i == p), jp_ = (j == p);
                            const double sp_ = (ip_ && jp_) ? -ip : (ip_ ? cj : ci) * ip;
                            const double v = (ip_ || jp_) ? sp_ : Sinv[e] - ci * cj * ip;
                            Sinv[e] = v;
                            const bool c1 = (j == p + 1);
                            if (c1 || i == p + 1) nxt[c1 ? i : j] = v;
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < NER; ++m) {
                    if (QR_AS_THREADS * m < ne) {                              // (uniform; threads past the last element work on a zero at (0, 0))
                        const int i = eij[m] >> 8, j = eij[m] & 255;
                        const double ci = cur[i], cj = cur[j];
                        const bool ip_ = (i == p), jp_ = (j == p);
                        const double sp_ = (ip_ && jp_) ? -ip : (ip_ ? cj : ci) * ip;
                        const double v = (ip_ || jp_) ? sp_ : el[m] - ci * cj * ip;
                        el[m] = v;
                        const bool c1 = (j == p + 1);
                        if (c1 || i == p + 1) nxt[c1 ? i : j] = v;
                    }
                }
                QR_SYNC();
            }
            if (ok) {
#pragma unroll
                for (int m = 0; m < NER; ++m) { const int e = tid + QR_AS_THREADS * m; if (e < ne) Sinv[e] = -el[m]; }
                if constexpr (NER < NE) { for (int e = tid + QR_AS_THREADS * NER; e < ne; e += QR_AS_THREADS) Sinv[e] = -Sinv[e]; }
                if (qW > 0 && q <= qW) {
                    for (int i = wv; i < q; i += 4) {
                        const int ci = sAct[i];
                        const int ki = (ci * 10923) >> 16;
                        double a0, a1, a2;
                        cons_vec(ci - 6 * ki, im, a0, a1, a2);
                        if (own) {
                            Blk B; load_block(Mb, kme, ki, B);
                            double *wq = Wc + i * nsp + 3 * kme;
                            wq[0] = B.m[0] * a0 + B.m[1] * a1 + B.m[2] * a2;
                            wq[1] = B.m[3] * a0 + B.m[4] * a1 + B.m[5] * a2;
                            wq[2] = B.m[6] * a0 + B.m[7] * a1 + B.m[8] * a2;
                        }
                    }
                }
            }
            QR_SYNC();
            return ok;
        };
        enum { CMD_GO = 0, CMD_EXIT = 1, CMD_REBUILD = 2 };
        if (wv != 0) {
            // ================================ workers ================================
            const int g = wv - 1;                         // 0..2
            for (;;) {
                QR_SYNC();                          // X1
                const int cmd = __builtin_amdgcn_readfirstlane(sCtl[0]);
                if (cmd == CMD_EXIT) return;
                const int q = __builtin_amdgcn_readfirstlane(sCtl[1]);
                if (cmd == CMD_REBUILD) {
                    const bool ok = rebuild(q);
                    fastz = ok ? (qW > 0 && q <= qW) : (qW > 0);      // as wave 0 decides (a failed rebuild falls back to a cold start)
                    continue;
                }
                const bool hi = BIG && q > 64;
                const double dq = (lane < q) ? dd[lane] : 0.0;
                const double dq2 = (hi && lane + 64 < q) ? dd[lane + 64] : 0.0;
                r_partial(q, dq, dq2);
                QR_SYNC();                          // B2
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
                QR_SYNC();                          // B3
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
                    QR_SYNC();                      // D1: everyone has column l before anyone changes S^-1
                    for (int j = g; j < q; j += 3) {
                        if (j == l) continue;
                        const double sj = rd2(sl, sl2, j) * isl;
                        if (lane < q && lane != l && j <= lane) Sinv[tril + j] -= sl * sj;
                        if (hi && lane + 64 < q && lane + 64 != l && j <= lane + 64) Sinv[tril2 + j] -= sl2 * sj;
                    }
                    QR_SYNC();                      // D2
                    double m0 = 0.0, m1 = 0.0;
                    if (l != last && g == 0 && lane < last) m0 = (lane == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, lane)];
                    if (hi && l != last && g == 0 && lane + 64 < last) m1 = (lane + 64 == l) ? Sinv[tri(last) + last] : Sinv[pidx(last, lane + 64)];
                    QR_SYNC();                      // D3
                    if (l != last && g == 0 && lane < last) Sinv[pidx(l, lane)] = m0;
                    if (hi && l != last && g == 0 && lane + 64 < last) Sinv[pidx(l, lane + 64)] = m1;
                }
            }
        }
        // ================================ wave 0: control ================================
        int q = 0, iter = 0;
#define CS_STAMP(i) do { } while (0)
#define CS_FINE(i) do { } while (0)
        unsigned amask = 0, xmask = 0;
        unsigned long long posk = 0;                      // byte t: working-set position of row t of my leg-step
        int ck = 0, ct = 0;                               // constraint (leg-step, row) at working-set position `lane`
        double uq = 0.0;                                  // its multiplier
        int ck2 = 0, ct2 = 0;                             // (h = 16 variants) the same for position lane + 64
        double uq2 = 0.0;
        const int maxit = 40 * nls + 100;
        bool done = (nls == 0);
        // ---- warm start: last tick's final working set of this robot slot (P.warm: per robot a 6-bit row mask per ORIGINAL leg-step, the
        // contact table it belonged to and a validity tag).  The contact table scrolls as the gait phase advances, so the old masks are
        // taken under the row shift (0, 1 or 2 horizon steps) that matches the old table to the new one best.  Speed only: the QP has
        // one optimum, whatever set the solve starts from.
        bool need_rebuild = false;
        int refac = 0;                                    // re-factorisations spent on a failed exit check (at most two)
        int iter_at_rebuild = 0;                          // (periodic refresh of S^-1, below)
        unsigned char *warm = (P.warm && nls > 0) ? P.warm + (size_t)rid * QR_WARM_STRIDE : nullptr;
        // Overlapped ticks: this robot's previous solve -- last tick's, on another stream set -- may still be running on another CU.  What it hands
        // over (the warm-start words here, the cost word at the end) it stores written through and then raises solved[robot] to its epoch: poll
        // for that (agent-scope loads; bounded at 20 ms of the 100 MHz clock: then the solve starts cold -- the guess is speed only -- and the
        // robot carries QRGPU_ST_PIPE_TIMEOUT) and read the words with loads of the same kind.  Only this wave reads them.
        const bool xtick = P.prev_solved != nullptr;
        bool warm_ok = true;
        long long t_waited = 0;                           // (shader-clock cycles spent in that wait: not part of what the robot COST, below)
        if (xtick) {
            const long long c0_ = clock64();
            const long long t0 = wall_clock64();
            while (!qr_epoch_reached(__hip_atomic_load(P.prev_solved + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), P.prev_epoch)) {
                if (wall_clock64() - t0 > P.xtick_wait) { st |= QRGPU_ST_PIPE_TIMEOUT_D; warm_ok = false; break; }
                __builtin_amdgcn_s_sleep(32);
            }
            t_waited = clock64() - c0_;
            if (lane == 0) QR_TRACE(rid, 32);
            if (lane == 0 && QR_P_TL) QR_P_TL[768 + 1024 * 16 + (P.solved_epoch & 15u) * 1024 + (rid & 1023)] = ((wall_clock64() - t0) << 8) | (long long)(P.rescue_mode & 7) | (warm_ok ? 0 : 8);
        }
        auto warm_ld = [&](const unsigned char *p_) -> unsigned { return xtick ? (unsigned)__hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)*p_; };
        if (warm && warm_ok && warm_ld(warm + QR_WARM_STRIDE - 1) == (unsigned)(unsigned char)h) {
            const unsigned long long cur = ((unsigned long long)(unsigned)sMisc[3] << 32) | (unsigned)sMisc[2];
            const unsigned long long old = xtick ? __hip_atomic_load((const unsigned long long *)(warm + 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                 : *(const unsigned long long *)(warm + 64);
            int sh = 0, best = 1 << 30;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int keep = NL - 4 * k;                                   // table rows that exist under this shift
                const unsigned long long m = keep >= 64 ? ~0ull : ((1ull << keep) - 1ull);
                const int miss = __popcll(((old >> (4 * k)) ^ cur) & m) * 4 + k;
                if (miss < best) { best = miss; sh = k; }
            }
            const int src = own ? sLs[kme] + 4 * sh : NL;
            unsigned gm = (src < NL) ? warm_ld(warm + src) & 0x3fu : 0u;
            // positions row-type-major: rows t of every leg-step, then rows t + 1 ...
            int base = 0;
            unsigned long long pk = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const bool has = (gm >> t) & 1u;
                const unsigned long long bal = __ballot(has);
                const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (has && pos < 256) pk |= (unsigned long long)(pos & 0xff) << (8 * t);
                base += __popcll(bal);
            }
            const int qg = base;
            if (qg > 0 && qg <= qcap) {
                amask = gm; posk = pk; q = qg;
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    if ((gm >> t) & 1u) { const int pos = (int)((pk >> (8 * t)) & 0xffu); sAct[pos] = 6 * kme + t; sPos[6 * kme + t] = (short)pos; }
                wave_sync();
                if (lane < q) { const int c = sAct[lane]; ck = (c * 10923) >> 16; ct = c - 6 * ck; }
                if (BIG && lane + 64 < q) { const int c = sAct[lane + 64]; ck2 = (c * 10923) >> 16; ct2 = c - 6 * ck2; }
                need_rebuild = true;
            }
        }
        // slack of the row at working-set position `lane` (+64) at the unconstrained optimum x_0 (kept in gl)
        auto slack_at_x0 = [&](int k_, int t_) {
            double a0, a1, a2;
            cons_vec(t_, im, a0, a1, a2);
            return a0 * gl[3 * k_] + a1 * gl[3 * k_ + 1] + a2 * gl[3 * k_ + 2] + ((t_ == 5) ? fmk[k_] : 0.0);
        };
        while (!done) {
            if (need_rebuild) {
                // ---- solve the equality-constrained problem on the whole working set: S^-1 from scratch, u = -S^-1 s(x_0), x = x_0 + W_A u;
                // rows whose multiplier comes out negative leave one by one (most negative first) until the point is dual feasible: then
                // (x, u, working set, S^-1) is a state the loop below can carry on from.
                need_rebuild = false;
                xmask = 0;
                if (lane < q) sAct[lane] = 6 * ck + ct;
                if (BIG && lane + 64 < q) sAct[lane + 64] = 6 * ck2 + ct2;
                if (lane == 0) { sCtl[0] = CMD_REBUILD; sCtl[1] = q; }
                QR_SYNC();                          // X1
                const bool ok = rebuild(q);
                if (QR_PFLOPS) fl_as += 17.5 * (double)q * (double)q + 1.5 * (double)q * (double)q * (double)q + 15.0 * (double)q * (double)nls;   // S, sweep of S, W_A
                if (!ok) {
                    // dependent rows in the guess: cold start
                    if (own) {
#pragma unroll
                        for (int t = 0; t < 6; ++t) sPos[6 * kme + t] = (short)-1;
                    }
                    amask = 0; posk = 0; q = 0; uq = 0.0; uq2 = 0.0;
                    x0 = gl[3 * kme]; x1 = gl[3 * kme + 1]; x2 = gl[3 * kme + 2];
                    fastz = qW > 0;
                    refac = 2;
                    continue;
                }
                fastz = qW > 0 && q <= qW;
                for (;;) {
                    if (++iter > maxit) { st |= QRGPU_ST_MPC_MAXITER_D; done = true; break; }
                    q = __builtin_amdgcn_readfirstlane(q);
                    const bool hi = BIG && q > 64;
                    const double dq = (lane < q) ? slack_at_x0(ck, ct) : 0.0;
                    double dq2 = 0.0;
                    if (hi) { dq2 = (lane + 64 < q) ? slack_at_x0(ck2, ct2) : 0.0; if (lane + 64 < q) dd[lane + 64] = dq2; }
                    if (lane < q) dd[lane] = dq;
                    if (lane == 0) { sCtl[0] = CMD_GO; sCtl[1] = q; }
                    QR_SYNC();                      // X1
                    r_partial(q, dq, dq2);
                    QR_SYNC();                      // B2
                    double rq, rq2 = 0.0;
                    { const double rs = (xr[lane] + xr[64 + lane]) + (xr[128 + lane] + xr[192 + lane]); rq = (lane < q) ? rs : 0.0; }
                    if (hi) { const double rs = (xr2[lane] + xr2[64 + lane]) + (xr2[128 + lane] + xr2[192 + lane]); rq2 = (lane + 64 < q) ? rs : 0.0; }
                    if (QR_PFLOPS) fl_as += 3.0 * (double)q * (double)q + 6.0 * (double)nls * (double)q + 8.0 * (double)q;
                    // u = -r; the most negative multiplier beyond rounding leaves
                    const double umax = -wave_min_d(hi ? (rq < rq2 ? rq : rq2) : rq);          // max u (>= 0 when any row is held properly)
                    const double worst = -wave_min_d(hi ? (-rq < -rq2 ? -rq : -rq2) : -rq);    // max r = -min u
                    const bool drop = worst > 1e-10 * (1.0 + (umax > 0.0 ? umax : 0.0));
                    // Many wrong rows in the guess (a warm start keeps 60 % of its rows on average, and at h = 16 some keep a fifth): dropping them one
                    // downdate round at a time costs 3-10 k cycles each, a block solve on what is left about 6 k + 1 k per row.  When at least
                    // four rows and a quarter of the set have negative multipliers they all leave at once and the set is solved afresh.
                    const double thr_neg = 1e-10 * (1.0 + (umax > 0.0 ? umax : 0.0));
                    const bool neg1 = lane < q && rq > thr_neg, neg2 = hi && lane + 64 < q && rq2 > thr_neg;
                    const int nneg = __popcll(__ballot(neg1)) + (hi ? __popcll(__ballot(neg2)) : 0);
                    const bool block_drop = drop && P.no_block_drop != 1 && nneg >= (P.no_block_drop > 1 ? (P.no_block_drop >> 8) : 4) && (P.no_block_drop > 1 ? (P.no_block_drop & 255) : 4) * nneg >= q && nneg < q;
                    int lpos = -1;
                    if (drop && !block_drop) { lpos = first_lane(lane < q && rq == worst); if (hi && lpos < 0) lpos = 64 + first_lane(lane + 64 < q && rq2 == worst); }
                    if (lane == 0) { sCtl[2] = (drop && !block_drop) ? F_DROP : 0; sCtl[3] = lpos; }
                    QR_SYNC();                      // B3
                    if (block_drop) {
                        // (the workers saw a round without a drop and wait at X1 for the next command)
                        const unsigned long long k1 = __ballot(lane < q && !neg1), k2 = hi ? __ballot(lane + 64 < q && !neg2) : 0ull;
                        const int n1 = __popcll(k1);
                        const int np1 = __builtin_amdgcn_mbcnt_hi((unsigned)(k1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)k1, 0u));
                        const int np2 = n1 + __builtin_amdgcn_mbcnt_hi((unsigned)(k2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)k2, 0u));
                        if (own) {
#pragma unroll
                            for (int t = 0; t < 6; ++t) sPos[6 * kme + t] = (short)-1;
                        }
                        wave_sync();
                        if (lane < q && !neg1) { sAct[np1] = 6 * ck + ct; sPos[6 * ck + ct] = (short)np1; }
                        if (hi && lane + 64 < q && !neg2) { sAct[np2] = 6 * ck2 + ct2; sPos[6 * ck2 + ct2] = (short)np2; }
                        q = n1 + (hi ? __popcll(k2) : 0);
                        wave_sync();
                        if (lane < q) { const int c = sAct[lane]; ck = (c * 10923) >> 16; ct = c - 6 * ck; }
                        if (BIG && lane + 64 < q) { const int c = sAct[lane + 64]; ck2 = (c * 10923) >> 16; ct2 = c - 6 * ck2; }
                        amask = 0; posk = 0;
                        if (own) {
#pragma unroll
                            for (int t = 0; t < 6; ++t) { const int ps = sPos[6 * kme + t]; if (ps >= 0) { amask |= 1u << t; posk |= (unsigned long long)(ps & 0xff) << (8 * t); } }
                        }
                        need_rebuild = true;
                        break;
                    }
                    if (!drop) {
                        x0 = gl[3 * kme] - ((xz[NV + 3 * kme] + xz[2 * NV + 3 * kme]) + xz[3 * NV + 3 * kme]);
                        x1 = gl[3 * kme + 1] - ((xz[NV + 3 * kme + 1] + xz[2 * NV + 3 * kme + 1]) + xz[3 * NV + 3 * kme + 1]);
                        x2 = gl[3 * kme + 2] - ((xz[NV + 3 * kme + 2] + xz[2 * NV + 3 * kme + 2]) + xz[3 * NV + 3 * kme + 2]);
                        uq = (lane < q && rq < 0.0) ? -rq : 0.0;
                        if (hi) uq2 = (lane + 64 < q && rq2 < 0.0) ? -rq2 : 0.0;
                        break;
                    }
                    // position lpos leaves: the same bookkeeping as a drop of the loop below (the workers downdate S^-1 between D1 and D3)
                    {
                        const int l = lpos, last = q - 1;
                        int clk, clt, cmk, cmt;
                        if (BIG && l >= 64) { clk = __builtin_amdgcn_readlane(ck2, l - 64); clt = __builtin_amdgcn_readlane(ct2, l - 64); }
                        else { clk = __builtin_amdgcn_readlane(ck, l); clt = __builtin_amdgcn_readlane(ct, l); }
                        if (BIG && last >= 64) { cmk = __builtin_amdgcn_readlane(ck2, last - 64); cmt = __builtin_amdgcn_readlane(ct2, last - 64); }
                        else { cmk = __builtin_amdgcn_readlane(ck, last); cmt = __builtin_amdgcn_readlane(ct, last); }
                        QR_SYNC();                  // D1
                        if (fastz && l != last && own) { const double *wl_ = Wc + last * nsp + 3 * kme; double *wd_ = Wc + l * nsp + 3 * kme; wd_[0] = wl_[0]; wd_[1] = wl_[1]; wd_[2] = wl_[2]; }
                        QR_SYNC();                  // D2
                        if (l != last) {
                            if (lane == l) { ck = cmk; ct = cmt; }
                            if (BIG && lane + 64 == l) { ck2 = cmk; ct2 = cmt; }
                        }
                        if (lane == clk) { amask &= ~(1u << clt); sPos[6 * clk + clt] = (short)-1; }
                        if (l != last && lane == cmk) { posk = (posk & ~(0xffull << (8 * cmt))) | ((unsigned long long)l << (8 * cmt)); sPos[6 * cmk + cmt] = (short)l; }
                        QR_SYNC();                  // D3
                        --q;
                    }
                    if (q == 0) { x0 = gl[3 * kme]; x1 = gl[3 * kme + 1]; x2 = gl[3 * kme + 2]; uq = 0.0; uq2 = 0.0; break; }
                }
                if (done) break;
                if (need_rebuild) continue;               // (a block drop: solve on what is left)
            }
            CS_FINE(1);
            double bs = INF; int bt = 0;
            {
                const double s[6] = {im * x0 + x2, -im * x0 + x2, im * x1 + x2, -im * x1 + x2, x2, fmaxk - x2};
                const unsigned blocked = own ? (amask | xmask) : 0x3fu;
#pragma unroll
                for (int t = 0; t < 6; ++t) { const bool take = !((blocked >> t) & 1u) && s[t] < bs; bs = take ? s[t] : bs; bt = take ? t : bt; }
            }
            CS_FINE(2);
            const double smin = wave_min_d(bs);
            CS_FINE(3);
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
                const double bemin = wave_min_d(be);
                if (bemin < -1e-4) {
                    // S^-1 has drifted (hundreds of bordered updates / downdates) or a row was set aside wrongly: rebuild it from the working
                    // set as it stands, re-solve on that set, restore dual feasibility and carry on; only a solve that fails the check
                    // after two such repairs keeps the flag
                    if (refac < 2 && q > 0) { ++refac; need_rebuild = true; continue; }
                    st |= QRGPU_ST_MPC_INFEAS_D;
                }
                break;
            }
            // S^-1 is kept by bordered updates and downdates; a solve that is still going after a hundred of them (the healthy ones end long
            // before) gets it rebuilt from the working set as it stands, and again every hundred changes after that: a wandering solve
            // (1 678 changes seen with the three-limb Hessian at h = 16) is one whose S^-1 has drifted
            if (iter - iter_at_rebuild >= QR_REFRESH_EVERY && q > 0) { iter_at_rebuild = iter; need_rebuild = true; continue; }
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
                QR_SYNC();                          // X1
                CS_STAMP(2);
                r_partial(q, dq, dq2);
                QR_SYNC();                          // B2
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
                // z'c / c'Mc = |z|^2 / |w|^2 in the Hessian norm: how independent of the working set the row is.  Measured over 6,000 solves
                // (scratch/diag_ratio.py) the rows a solve adds sit at >= 9e-4; the only other values seen were ~1e-13 -- rows that are dependent
                // but slipped through a looser test, after which 1 / z'c = 1e13 in S^-1 cost ten digits and the solve ended with active rows
                // 2e-4 N off (the drift flags of round 1).  Below 1e-9 a row is treated as dependent: no full step, a blocking row leaves first.
                const bool have_z = zc > 1e-13 * delta;
                const bool indep = zc > 1e-9 * delta;
                const double izc = fast_rcp(zc);
                const double t2 = indep ? -sp * izc : INF;
                const double t = t1 < t2 ? t1 : t2;
                const bool degenerate = !(t < INF);
                const bool full = !degenerate && indep && t == t2;
                const bool over = full && q >= qcap;
                const int flags = (degenerate || over) ? 0 : (full ? F_FULL : F_DROP);
                if (lane == 0) { sCtl[2] = flags; sCtl[3] = lpos; sCtl[4] = __double2hiint(izc); sCtl[5] = __double2loint(izc); }
                CS_STAMP(4);
                QR_SYNC();                          // B3
                CS_STAMP(5);
                if (degenerate) { if (lane == kp) xmask |= 1u << tp; break; }
                if (over) { st |= QRGPU_ST_MPC_OVERFLOW_D; done = true; break; }      // (main pass: the robot goes on the rescue list below)
                if (have_z) {
                    const double z0 = w0 - ((xz[NV + 3 * kme] + xz[2 * NV + 3 * kme]) + xz[3 * NV + 3 * kme]);
                    const double z1 = w1 - ((xz[NV + 3 * kme + 1] + xz[2 * NV + 3 * kme + 1]) + xz[3 * NV + 3 * kme + 1]);
                    const double z2 = w2_ - ((xz[NV + 3 * kme + 2] + xz[2 * NV + 3 * kme + 2]) + xz[3 * NV + 3 * kme + 2]);
                    x0 += t * z0; x1 += t * z1; x2 += t * z2;
                }
                uq -= t * rq;
                if (hi) uq2 -= t * rq2;
                up += t;
                CS_FINE(0);
                if (QR_PFLOPS) fl_as += 3.0 * (double)q * (double)q + 6.0 * (double)nls * (double)q + 21.0 * (double)nls + 12.0 * (double)q;
                if (full) {
                    if (fastz) {
                        // (the address from the lane number, behind an opaque copy: hoisted out of the loop as `Wc + 3 * kme` it is spilled under 128
                        //  registers and reloaded from scratch here, once per change, in front of a wait for the load)
                        if (q < qW) { if (own) { int kl = lane; asm volatile("" : "+v"(kl)); double *wq = Wc + q * nsp + 3 * kl; wq[0] = w0; wq[1] = w1; wq[2] = w2_; } }
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
                    QR_SYNC();                      // D1
                    if (fastz && l != last && own) { const double *wl_ = Wc + last * nsp + 3 * kme; double *wd_ = Wc + l * nsp + 3 * kme; wd_[0] = wl_[0]; wd_[1] = wl_[1]; wd_[2] = wl_[2]; }
                    QR_SYNC();                      // D2
                    if (l != last) {
                        if (lane == l) { uq = ulast; ck = cmk; ct = cmt; }
                        if (BIG && lane + 64 == l) { uq2 = ulast; ck2 = cmk; ct2 = cmt; }
                    }
                    if (lane == clk) { amask &= ~(1u << clt); sPos[6 * clk + clt] = (short)-1; }
                    if (l != last && lane == cmk) { posk = (posk & ~(0xffull << (8 * cmt))) | ((unsigned long long)l << (8 * cmt)); sPos[6 * cmk + cmt] = (short)l; }
                    QR_SYNC();                      // D3
                    xmask = 0;
                    --q;
                }
            }
        }
        if (lane == 0) { sCtl[0] = CMD_EXIT; sMisc[15] = nbar + 1; }      // workers leave at their next X1 (and the parked waves of a persistent workgroup with them)
        QR_SYNC();
        QR_TS(5);
        const bool to_rescue = (st & QRGPU_ST_MPC_OVERFLOW_D) && P.rescue_list && !P.rescue_mode;
        if (warm && !to_rescue) {                         // (a robot on its way to the list pass keeps last tick's guess for that pass)
            // this tick's final working set, by original leg-step, for the next tick of this robot slot (only a converged solve is worth it)
            const bool good = (st & 0xff) == 0;
            // (overlapped ticks: written through -- the robot's next solve may read them from another CU while this launch still runs)
            auto warm_st = [&](unsigned char *p_, unsigned char v_) { if (P.solved) __hip_atomic_store(p_, v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p_ = v_; };
            warm_st(warm + lane, 0);
            wave_sync();
            unsigned keep = amask;
            if (P.warm_uthr > 0.0) {
                // only rows held with a multiplier above a fraction of the largest go into the next tick's guess
                const double um = -wave_min_d((lane < q) ? -uq : 0.0);
                const double um2 = (BIG && q > 64) ? -wave_min_d((lane + 64 < q) ? -uq2 : 0.0) : 0.0;
                const double thr_u = P.warm_uthr * (um > um2 ? um : um2);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int pos = (int)((posk >> (8 * t)) & 0xffu);
                    const double ut = (BIG && pos >= 64) ? __shfl(uq2, pos - 64, 64) : __shfl(uq, pos & 63, 64);
                    if (((amask >> t) & 1u) && !(ut > thr_u)) keep &= ~(1u << t);
                }
            }
            if (own && good) warm_st(warm + sLs[kme], (unsigned char)keep);
            if (lane == 0) {
                const unsigned long long tbl = ((unsigned long long)(unsigned)sMisc[3] << 32) | (unsigned)sMisc[2];
                if (P.solved) __hip_atomic_store((unsigned long long *)(warm + 64), tbl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *(unsigned long long *)(warm + 64) = tbl;
                warm_st(warm + QR_WARM_STRIDE - 1, good ? (unsigned char)h : (unsigned char)0);
            }
        }
        wave_sync();
        if (lane < 12) xz[lane] = 0.0;
        wave_sync();
        if (own) { const int ls = sLs[kme]; if (ls < 4) { xz[3 * ls] = x0; xz[3 * ls + 1] = x1; xz[3 * ls + 2] = x2; } }
        wave_sync();
        {
            float Ro[3][3];
            const float *qs = (const float *)sMisc + 4;
            quat_to_R(qs[0], qs[1], qs[2], qs[3], Ro);
            mpc_outputs(lane, rid, n, xz, Ro, sJ, C, io.g_q, io.g_force, io.g_force_wbc, io.force_stride, io.g_tau, P.epilogue, P.done_flag != nullptr);
        }
        if (lane == 0 && io.g_status) {
            if (P.done_flag) __hip_atomic_store(io.g_status + rid, st | ((iter & 0xffff) << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else io.g_status[rid] = st | ((iter & 0xffff) << 8);
        }
        if (to_rescue && P.main_done) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (its rescuer may start at once: everything this solve stored is out first)
        if (lane == 0 && to_rescue) QR_TRACE(rid, 64);
        if (lane == 0 && to_rescue) st_xt(P.rescue_list + atomicAdd(P.rescue_count + P.rescue_parity, 1), rid, P.solved != nullptr);
        if (P.done_flag) {
            // pipelined tick: the forces, torques and status word of this robot are on their way to memory (write-through stores of this very
            // wave): wait for them, then raise the robot's flag for the WBC workgroup that is waiting for it (or, for a robot on its way to the
            // list pass, tell that workgroup to leave it to the WBC pass behind the list launch)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                // (a compare-and-swap, not a store.  A WBC workgroup that gave up waiting for this robot leaves 0x80000000 | epoch << 1 here -- its own
                //  status word, written before ours, is gone, so the time-out is recorded again behind ours.  And in an overlapped tick the word may
                //  already carry a LATER epoch: the lane's next tick, which waits for this tick's WBC launch but not for a solve that launch gave up
                //  on, has been here.  Then the word is left alone and the robot flagged all the same: never silent.)
                const unsigned mine = (P.done_epoch << 1) | (to_rescue ? 1u : 0u);
                unsigned cur = __hip_atomic_load(P.done_flag + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool late = false;
                for (int tries = 0; tries < 64; ++tries) {
                    const unsigned ce = (cur >> 1) & 0x3fffffffu;
                    late = cur == (0x80000000u | (P.done_epoch << 1));
                    if (ce != P.done_epoch && qr_epoch_reached(ce, P.done_epoch)) { late = true; break; }
                    const unsigned seen = atomicCAS(P.done_flag + rid, cur, mine);
                    if (seen == cur) break;
                    cur = seen;
                }
                if (late && io.g_status) __hip_atomic_fetch_or(io.g_status + rid, QRGPU_ST_PIPE_TIMEOUT_D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0 && QR_P_FTIME) QR_P_FTIME[rid] = (int)wall_clock64();
            if (lane == 0 && QR_P_TL) atomicMax(QR_P_TL + (P.done_epoch & 63u) * 8 + 2, wall_clock64());
        }
        if (lane == 0 && QR_PFLOPS) { double *fo = QR_PFLOPS + (size_t)rid * 4; fo[0] = fl_v32; fo[1] = fl_m32; fo[2] = fl_sw; fo[3] = fl_as; }
        if (lane == 0 && P.cost) {
            // what this robot cost, in units of 256 cycles -- smoothed over ticks when there is a history: half of a robot's tick-to-tick variation is
            // the change count of its active set, which does not persist (correlation 0.47 between consecutive ticks), and a longest-first
            // order from last tick's cost alone ends 10 us later than one from the running mean (scratch/analyze_predict.py)
            long long c = (clock64() - t_begin - t_waited) >> 8;
            if (c > 0xffff) c = 0xffff;
            if (P.cost_ema) {
                const int prevc = P.prev_solved ? __hip_atomic_load(P.cost_in + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : P.cost_in[rid];
                c = (c + ((prevc >> 16) & 0xffff) + 1) >> 1;
            }
            const int cfine = (int)c;
            c >>= 4;
            int big = 0;
            if (P.pre_list) {
                // does this robot belong in the planned list next time?  It overflowed, or ended within six rows of what the main pass's LDS
                // holds for its size, or is of the big class
                const long long remm = (long long)(P.lds_main / 8) - (long long)(Mb - smem) - (long long)npairs * 9;
                int qcm = 0;
                if (remm > 0) { qcm = (int)((__builtin_sqrt(8.0 * (double)remm + 1.0) - 1.0) * 0.5); while (tri(qcm) > remm) --qcm; }
                if (qcm > 64) qcm = 64;
                big = ((st & QRGPU_ST_MPC_OVERFLOW_D) || q + P.big_margin >= qcm || (P.big_nls > 0 && nls >= P.big_nls)
                       || (P.big_cost > 0 && cfine >= (P.rescue_mode == 0 ? P.big_cost : P.big_cost_stay))) ? 1 : 0;
            }
            if (P.planned_done || P.solved) __hip_atomic_store(P.cost + rid, (c > 255 ? 255 : (int)c) | (big << 8) | (cfine << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else P.cost[rid] = (c > 255 ? 255 : (int)c) | (big << 8) | (cfine << 16);
        }
        if (P.solved && !to_rescue) {
            // overlapped ticks: the warm-start words and the cost word of this robot are on their way to memory (write-through stores of this
            // very wave): wait for them, then tell the robot's next solve (a robot on its way to the list pass is told by that pass's solve)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) qr_epoch_raise(P.solved + rid, P.solved_epoch);
        }
        if (lane == 0 && QR_P_TL && P.solved) QR_P_TL[768 + (P.solved_epoch & 15u) * 1024 + (rid & 1023)] = (wall_clock64() << 8) | (long long)(P.rescue_mode & 7) | (to_rescue ? 8 : 0) | ((long long)(P.solved_epoch & 15u) << 4);
        QR_TS(6);
        if (lane == 0 && QR_DBGT) QR_DBGT[(size_t)rid * 16 + 13] = wall_clock64();
        if (lane == 0 && QR_DBGT) { QR_DBGT[(size_t)rid * 16 + 7] = ns; QR_DBGT[(size_t)rid * 16 + 14] = q; }
        if (lane == 0 && QR_DBGT) QR_DBGT[(size_t)rid * 16 + 15] = ((long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
}
#undef QR_SYNC
#undef QR_IDLE

// Launch wrappers (one inlined copy of the solve each).
//   LIST = false, main pass: workgroup b solves the robot of slot xcd_robot_index(b) (through the longest-first order when there is one);
//   LIST = true:  workgroups 0-7 first sort the next call's dispatch order, then workgroup b re-solves entries b, b + grid, b + 2 grid ...
//                 of the rescue list (robots whose working set outgrew the main pass's registers or LDS) with this launch's larger LDS
//                 allotment and the BIG register set.
//   MINW (0 = by the rule below): waves per SIMD the register allocation must leave room for.  The h <= 16 four-wave kernels take AGPRs on top of
//                 their 256 VGPRs under the default of one wave per SIMD, so two of their workgroups never share a CU; MINW = 2 is the build that can.
//   H16: the h > 11 form of the solve (no early copy of the torque map's Jacobians in LDS, S^-1 may live in the global scratch) -- every MAXB > 4
//                 variant, and the eight-wave two-blocks-per-thread kernel when it runs a trotting h = 16 robot on half a CU (MINW = 4)
template <int MAXB, bool BIG, bool LIST, int NTHR, int MINW = 0, bool H16 = (MAXB > 4)>
__global__ __launch_bounds__(NTHR, (MINW ? MINW : ((MAXB <= 4 && !LIST && !BIG) ? (NTHR >= 384 ? 4 : QR_MAIN_WAVES_PER_SIMD) : (NTHR >= 512 ? 2 : 1))))
void qr_mpc_kernel(MpcLaunch P, MpcIO io)
{
    extern __shared__ double smem[];
    if (P.started && threadIdx.x == 0) atomicAdd(P.started, 1);        // (planned list launches: see qr_gate_kernel)
    if (!LIST && P.main_started && P.rescue_mode == 0 && threadIdx.x == 0) atomicAdd(P.main_started, 1);     // (pipelined tick: the WBC launch's gate)
    if (QR_P_TL && threadIdx.x == 0) {
        long long *tl = QR_P_TL + (P.done_epoch & 63u) * 8;
        const long long t = wall_clock64();
        if (!LIST && P.rescue_mode == 0) { atomicMin(tl + 0, t); atomicMax(tl + 1, t); }
        if (LIST && P.rescue_mode == 1) atomicMin(tl + 5, t);
    }
    if constexpr (LIST) {
        if (P.rescue_mode == 1 && P.planned_done) {
            // the planned launch beside the main pass must be through before its robots' costs are sorted and its hand-overs re-solved
            if (threadIdx.x == 0) {
                const long long t0 = wall_clock64();
                while ((int)((unsigned)__hip_atomic_load(P.planned_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)P.planned_expect) < 0 && wall_clock64() - t0 < 2000000)
                    __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
        const bool planned = P.rescue_mode == 2;
        if (!planned && blockIdx.x < 8) {
            if (P.lpt_order_out) {                     // the histogram borrows the head of the dynamic LDS before a solve carves it
                lpt_order_chunk(blockIdx.x, P.n, P.lpt_cost_in, P.lpt_order_out, (int *)smem);
                __syncthreads();
            }
            if (QR_P_FTIME && P.wbc_order_out) { finish_order_chunk(blockIdx.x, P.n, QR_P_FTIME, P.wbc_order_out, (int *)smem); __syncthreads(); }
            if (P.pre_list && P.skip && P.lpt_cost_in) {
                // plan the next call: robots whose solve left the `big` bit go on the planned list and are skipped by the main pass
                const int chunk = (P.n + 7) >> 3, lo = blockIdx.x * chunk, hi = (lo + chunk < P.n) ? lo + chunk : P.n;
                for (int i = lo + threadIdx.x; i < hi; i += NTHR) {
                    const int big = (ld_xt(P.lpt_cost_in + i, P.solved != nullptr) >> 8) & 1;
                    st_xt(P.skip + i, (unsigned char)big, P.solved != nullptr);
                    if (big) st_xt(P.pre_list + P.pre_list_next + atomicAdd(P.pre_count + (P.rescue_parity ^ 1), 1), i, P.solved != nullptr);
                }
                __syncthreads();
                // the last of the eight planning workgroups tells the host how long the list is (a write to pinned host memory: no copy
                // command on the stream, no sync)
                if (threadIdx.x == 0 && P.pre_hint) {
                    __threadfence();
                    if (atomicAdd(P.pre_count + 2, 1) == 7) {
                        P.pre_hint[P.rescue_parity ^ 1] = atomicAdd(P.pre_count + (P.rescue_parity ^ 1), 0);
                        if (QR_P_TL) QR_P_TL[768 + 32768 + 4096 + (P.done_epoch & 63u)] = (long long)P.pre_hint[P.rescue_parity ^ 1] | ((long long)(P.rescue_parity ^ 1) << 16) | (1ll << 40);
                        __threadfence_system();
                    }
                }
            }
        }
        const int *list = planned ? P.pre_list : P.rescue_list;
        int cnt = ld_xt((planned ? P.pre_count : P.rescue_count) + P.rescue_parity, P.solved != nullptr);
        cnt = cnt < P.n ? cnt : P.n;
        if (P.plan_only) return;                 // (overlapped tick at h > 11: the tick's planned launch empties the rescue list, MpcLaunch::main_done)
        for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
            mpc_solve_robot<MAXB, BIG, NTHR, false, H16>(P, io, ld_xt(list + e, P.solved != nullptr), smem);
            __syncthreads();                           // every wave is out of the solve before the LDS is carved again
        }
        if (QR_P_TL && threadIdx.x == 0 && P.rescue_mode == 1) atomicMax(QR_P_TL + (P.done_epoch & 63u) * 8 + 6, wall_clock64());
    } else {
        if (P.rescue_mode == 3) {
            // planned list, one robot per workgroup (so that the waves beyond the active set's four may leave after the sweep, which a workgroup
            // striding over a list cannot allow): entry blockIdx.x of the list the last call's planning left.  The grid is the host's
            // unsynchronised copy of the list's length; should the list be longer, the last workgroup hands the remainder to the trailing list
            // launch, and workgroups past the end of a shorter list leave at once.
            // (every workgroup of this launch tells the trailing launch when it is done -- planned_done -- whichever way it leaves)
            // (timeline build: first start / last end of the planned launch's workgroups, behind the gate's two slots of the epoch's extra row)
            if (QR_P_TL && threadIdx.x == 0) atomicMin(QR_P_TL + 640 + (P.done_epoch & 63u) * 2, wall_clock64());
            auto tell_done = [&]() {
                if (QR_P_TL && threadIdx.x == 0) atomicMax(QR_P_TL + 640 + (P.done_epoch & 63u) * 2 + 1, wall_clock64());
                if (P.planned_done && threadIdx.x < 64) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (threadIdx.x == 0) __hip_atomic_fetch_add(P.planned_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            if (P.plan_abort && *P.plan_abort == P.plan_epoch) { tell_done(); return; }       // its gate gave up: the main pass solves everybody (below)
            int cnt = ld_xt(P.pre_count + P.rescue_parity, P.solved != nullptr);
            if (QR_P_TL && threadIdx.x == 0) QR_P_TL[768 + 32768 + (P.done_epoch & 63u) * 64 + (blockIdx.x & 63)] = (long long)cnt | ((long long)P.rescue_parity << 16) | ((long long)gridDim.x << 20) | ((long long)P.planned_stride << 32) | (1ll << 40);
            cnt = cnt < P.n ? cnt : P.n;
            if constexpr (MAXB == 5) {
                // (QRGPU_H16_TWO: a tenth of a mixed h = 16 batch is listed -- more robots than the launch may take CUs.  The workgroup keeps its CU
                //  and goes down the list; the waves its sweep no longer needs are parked as in the persistent main pass.)
                if (P.planned_stride && P.main_done && P.rescue_taken && P.rescue_list) {
                    // Overlapped tick at h > 11 (MpcLaunch::main_done): the workgroups take the list's entries off a head (they start as the last
                    // tick's leave their CUs, not together: no fixed shares), and the first `linger` of them stay for the main pass's hand-overs.
                    if (P.planned_stride == 2) cnt = 0;            // (no plan on the host's side: the main pass skips nobody, this launch only rescues)
                    volatile int *sNext = (volatile int *)smem;        // (the head of the dynamic LDS, dead between two solves)
                    int *const phead = P.rescue_taken + 2 + P.rescue_parity;
                    int *const head = P.rescue_taken + P.rescue_parity;
                    const int *const tail = P.rescue_count + P.rescue_parity;
                    const bool lingers = (int)blockIdx.x < P.linger;
                    const long long t0 = wall_clock64();
                    for (;;) {
                        if (threadIdx.x == 0) {
                            int got = -2;
                            for (;;) {
                                int ph = cnt > 0 ? __hip_atomic_load(phead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                                if (ph < cnt) {
                                    if (!__hip_atomic_compare_exchange_strong(phead, &ph, ph + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                                    got = ld_xt(P.pre_list + ph, true);
                                    QR_TRACE(got, 8);
                                    break;
                                }
                                // (a workgroup that does not stay still takes a hand-over that is there when it looks: a whole class arriving unannounced --
                                //  a new population -- is spread over every workgroup of the launch, not queued for the eight that stay)
                                int taken = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                int avail = __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (!lingers && (P.linger == 0 || !(taken < avail && taken < P.n))) break;          // (linger = 0: the give-up test, nobody takes a hand-over)
                                if (taken < avail && taken < P.n) {
                                    if (!__hip_atomic_compare_exchange_strong(head, &taken, taken + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                                    // (the entry's store follows its writer's bump of the tail: a few hundred nanoseconds at most)
                                    int r = -1;
                                    const long long t1 = wall_clock64();
                                    while ((r = __hip_atomic_load(P.rescue_list + taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0 && wall_clock64() - t1 < 100000)
                                        __builtin_amdgcn_s_sleep(4);
                                    if (r < 0 || r >= P.n) continue;           // (cannot happen: the robot's WBC workgroup then reports it, QRGPU_ST_PIPE_TIMEOUT)
                                    __hip_atomic_store(P.rescue_list + taken, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    got = r;
                                    QR_TRACE(r, 16);
                                    break;
                                }
                                if ((int)((unsigned)__hip_atomic_load(P.main_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)P.main_done_expect) >= 0) {
                                    avail = __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    taken = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    if (taken < avail && taken < P.n) continue;
                                    break;                                     // the main pass is through and the list is empty
                                }
                                if (wall_clock64() - t0 > 5 * P.xtick_wait) break;          // (100 ms: a main pass that never ends)
                                __builtin_amdgcn_s_sleep(32);
                            }
                            *sNext = got;
                        }
                        __syncthreads();
                        const int rid = *sNext;
                        __syncthreads();
                        if (rid < 0) break;
                        mpc_solve_robot<MAXB, BIG, NTHR, true>(P, io, rid, smem);
                        __syncthreads();
                    }
                    tell_done();
                    return;
                }
                if (P.planned_stride) {
                    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
                        if (threadIdx.x == 0) QR_TRACE(ld_xt(P.pre_list + e, P.solved != nullptr), 8);
                        mpc_solve_robot<MAXB, BIG, NTHR, true>(P, io, ld_xt(P.pre_list + e, P.solved != nullptr), smem);
                        __syncthreads();
                    }
                    if (P.main_done && P.rescue_taken && P.rescue_list) {
                        // (cannot happen: a launch with main_done set takes its entries off the list's head, below)
                    }
                    tell_done();
                    return;
                }
            }
            if (blockIdx.x == gridDim.x - 1 && cnt > (int)gridDim.x && P.rescue_list) {
                for (int e2 = (int)gridDim.x + (int)threadIdx.x; e2 < cnt; e2 += NTHR) {
                    const int r2 = ld_xt(P.pre_list + e2, P.solved != nullptr);
                    __hip_atomic_store(P.rescue_list + atomicAdd(P.rescue_count + P.rescue_parity, 1), r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // (pipelined tick: the WBC workgroup of a robot handed on like this must not wait for a flag nobody raises -- the main pass
                    //  skips the robot, the trailing launch raises none -- but leave it to the WBC pass behind the trailing launch)
                    if (P.done_flag) __hip_atomic_store(P.done_flag + r2, (P.done_epoch << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();                       // (every wave's hand-over stores are out before wave 0 can tell anybody)
            }
            if ((int)blockIdx.x >= cnt) { tell_done(); return; }
            mpc_solve_robot<MAXB, BIG, NTHR, false, H16>(P, io, ld_xt(P.pre_list + blockIdx.x, P.solved != nullptr), smem);
            tell_done();                               // (wave 0 is the last to return from the solve and the one that stored its results)
            return;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {     // the next call's counters
            if (P.rescue_count) st_xt(P.rescue_count + (P.rescue_parity ^ 1), 0, P.solved != nullptr);
            if (P.rescue_taken) { st_xt(P.rescue_taken + (P.rescue_parity ^ 1), 0, true); st_xt(P.rescue_taken + 2 + (P.rescue_parity ^ 1), 0, true); }
            if (P.pre_count) { st_xt(P.pre_count + (P.rescue_parity ^ 1), 0, P.solved != nullptr); st_xt(P.pre_count + 2, 0, P.solved != nullptr); }
        }
        const int slot = xcd_robot_index(blockIdx.x, P.n);
        if (slot >= 0) {
            const int rid = P.order ? ld_xt(P.order + slot, P.main_done != nullptr) : slot;       // same XCD chunk either way (the order permutes inside a chunk)
            // solved by the planned list launch, beside this one -- unless that launch's gate gave up waiting for this one's stream (plan_abort)
            if (threadIdx.x == 0) QR_TRACE(rid, 1);
            if (!(P.skip && ld_xt(P.skip + rid, P.main_done != nullptr) && !(P.plan_abort && *P.plan_abort == P.plan_epoch)))
                mpc_solve_robot<MAXB, BIG, NTHR, false, H16>(P, io, rid, smem);
            else if (threadIdx.x == 0) QR_TRACE(rid, 2);
        }
        // (overlapped ticks at h > 11: the planned launch's workgroups take what this pass leaves on the rescue list -- thread 0 is the one that
        //  appends -- and go home when every workgroup of this launch has left: MpcLaunch::main_done)
        if (P.main_done && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(P.main_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#ifndef QR_FLOPS_BUILD
// Persistent main pass (MpcLaunch::persist): one workgroup per resident slot; robots come off the queue of the workgroup's own XCD (slot
// order of its chunk = the longest-first order), then off the others'; every robot taken counts for the WBC launch's gate.
template <int MAXB, bool BIG, int NTHR, int MINW = 0>
__global__ __launch_bounds__(NTHR, (MINW ? MINW : ((MAXB <= 4 && !BIG) ? (NTHR >= 384 ? 4 : QR_MAIN_WAVES_PER_SIMD) : (NTHR >= 512 ? 2 : 1))))
void qr_mpc_persist_kernel(MpcLaunch P, MpcIO io)
{
    extern __shared__ double smem[];
    volatile int *sNext = (volatile int *)smem;        // (the head of the dynamic LDS, dead between two solves: a static word would push the second workgroup off the CU)
    if (QR_P_TL && threadIdx.x == 0) atomicMin(QR_P_TL + (P.done_epoch & 63u) * 8, wall_clock64());
    if (blockIdx.x == 0 && threadIdx.x == 0) {         // the next call's counters
        if (P.rescue_count) st_xt(P.rescue_count + (P.rescue_parity ^ 1), 0, P.solved != nullptr);
        if (P.pre_count) { st_xt(P.pre_count + (P.rescue_parity ^ 1), 0, P.solved != nullptr); st_xt(P.pre_count + 2, 0, P.solved != nullptr); }
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) P.qhead_next[threadIdx.x] = 0;
    const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);      // HW_REG_XCC_ID[3:0]
    const int chunk = (P.n + 7) >> 3;
    for (;;) {
        if (threadIdx.x == 0) {
            int got = -1;
            for (int a = 0; a < 8 && got < 0; ++a) {
                const int y = (xcc + a) & 7, lo = y * chunk;
                const int len = (lo + chunk < P.n ? lo + chunk : P.n) - lo;
                while (len > 0 && __hip_atomic_load(P.qhead + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < len) {
                    const int k = atomicAdd(P.qhead + y, 1);
                    if (k >= len) break;
                    if (P.main_started) atomicAdd(P.main_started, 1);
                    const int r = P.order ? ld_xt(P.order + lo + k, P.main_done != nullptr) : lo + k;
                    if (P.skip && ld_xt(P.skip + r, P.main_done != nullptr) && !(P.plan_abort && *P.plan_abort == P.plan_epoch)) continue;         // solved by the planned list launch, beside this one
                    got = r;
                    if (QR_P_TL) atomicMax(QR_P_TL + (P.done_epoch & 63u) * 8 + 1, wall_clock64());
                    break;
                }
            }
            *sNext = got;
        }
        __syncthreads();
        const int rid = *sNext;
        __syncthreads();
        if (rid < 0) return;
        mpc_solve_robot<MAXB, BIG, NTHR, true>(P, io, rid, smem);
        __syncthreads();                               // every wave is out of the solve before the LDS is carved again
    }
}
template __global__ void qr_mpc_persist_kernel<2, false, 512>(MpcLaunch, MpcIO);
template __global__ void qr_mpc_persist_kernel<5, true, 512>(MpcLaunch, MpcIO);
template __global__ void qr_mpc_persist_kernel<9, true, 256>(MpcLaunch, MpcIO);     // h <= 16 on four waves (no parked waves: every wave is in the active set)
template __global__ void qr_mpc_persist_kernel<9, true, 256, 2>(MpcLaunch, MpcIO);  // the same within 256 registers: two workgroups per CU (QRGPU_H16_TWO)
#endif

template __global__ void qr_mpc_kernel<2, false, false, 512>(MpcLaunch, MpcIO);     // h <= 11, main pass: eight waves build and sweep (128 VGPRs), four solve
template __global__ void qr_mpc_kernel<4, false, false, 256>(MpcLaunch, MpcIO);     // h <= 11, main pass on four waves (QRGPU_MAIN_THREADS=256, A/B)
template __global__ void qr_mpc_kernel<4, true, true, 256>(MpcLaunch, MpcIO);       // h <= 11, list launches (whole CU's LDS, 96 rows)
template __global__ void qr_mpc_kernel<9, true, true, 256>(MpcLaunch, MpcIO);       // h <= 16, list launches (whole CU's LDS, 96 rows)
template __global__ void qr_mpc_kernel<4, true, true, 256, 2, true>(MpcLaunch, MpcIO);   // h <= 11, list launches of an OVERLAPPED tick: half a CU's LDS, S^-1 (96 rows) in the global scratch
template __global__ void qr_mpc_kernel<9, true, false, 256>(MpcLaunch, MpcIO);      // h <= 16, four waves (QRGPU_H16_THREADS=256, A/B)
template __global__ void qr_mpc_kernel<9, true, false, 256, 2>(MpcLaunch, MpcIO);   // h <= 16, four waves within 256 registers: two workgroups per CU (QRGPU_H16_TWO)
template __global__ void qr_mpc_kernel<5, true, false, 512>(MpcLaunch, MpcIO);      // h <= 16: eight waves build and sweep (256 VGPRs, one workgroup per CU)
template __global__ void qr_mpc_kernel<2, false, false, 512, 4, true>(MpcLaunch, MpcIO);
template __global__ void qr_mpc_kernel<2, true, false, 512, 4, true>(MpcLaunch, MpcIO);   // h <= 16 two to a CU: a trotting robot (<= 42 stance leg-steps) on eight waves within 128 registers
template __global__ void qr_mpc_kernel<2, true, false, 512>(MpcLaunch, MpcIO);      // h <= 11, planned list: one robot per workgroup, whole CU's LDS, 96 rows, eight waves build and sweep

#ifndef QR_FLOPS_BUILD
// Holds the stream it is launched on until every workgroup of the planned list launches issued so far (side stream) has started, or
// max_ticks of the 100 MHz clock have passed, whichever comes first.  A listed robot needs a whole CU: left to the dispatcher, the
// main pass's thousand workgroups fill every CU first and the listed robot starts 80-160 us late -- which is then the end of the launch.
// The counter is cumulative and never cleared (`expected_total` is the host's running sum of the grids, compared as a wrapping difference):
// a workgroup that starts after a gate has timed out is counted where it belongs instead of leaking into the next call's count.
// `timed_out` (or null): set to `timed_out_value` when the wait ends on the clock instead of the counter -- neither the join of a pipelined tick
// (a word of pinned host memory: qrgpu_sync reports it) nor the gate of its WBC launch (the tick's epoch in a device word: every robot of
// that tick is flagged QRGPU_ST_PIPE_TIMEOUT) may give up silently.
__global__ void qr_gate_kernel(int *counter, int expected_total, long long max_ticks, int *timed_out, int timed_out_value, int *bump)
{
    if (threadIdx.x != 0) return;
    if (bump) __hip_atomic_fetch_add(bump, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (the "go" of the planned launch's own gate: see qrgpu_api.hip)
    const long long t0 = wall_clock64();
    for (;;) {
        if ((int)((unsigned)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)expected_total) >= 0) return;
        if (wall_clock64() - t0 >= max_ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
    if (timed_out) { *timed_out = timed_out_value; __threadfence_system(); }
}

// The gate in front of an OVERLAPPED tick's launches (qrgpu_set_tick_overlap): tick t + 1 is queued on another stream set and may start in the
// slots tick t's drain leaves empty -- but not before every workgroup of tick t's main pass has started (c0 / e0: the count the WBC launch's
// gate polls too) and every workgroup of its planned launch (c1 / e1, or null): a workgroup of tick t + 1 waits, per robot, for that robot's
// tick-t solve, and must never hold a slot that solve still needs to START.  Bounded; giving up is harmless (the per-robot waits are bounded
// too, and a solve whose wait gives up starts cold and is flagged).
__global__ void qr_gate2_kernel(int *c0, int e0, int *c1, int e1, long long max_ticks, long long *stamp)
{
    if (threadIdx.x != 0) return;
    auto reached = [](int *p, int e) { return (int)((unsigned)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)e) >= 0; };
    const long long t0 = wall_clock64();
    if (stamp) stamp[0] = t0;                   // (diagnostic: when the gate came up / opened)
    while (!reached(c0, e0) || (c1 && !reached(c1, e1))) {
        if (wall_clock64() - t0 >= max_ticks) return;
        __builtin_amdgcn_s_sleep(8);
    }
    if (stamp) stamp[1] = wall_clock64();
}

// The join of a pipelined tick: waits for the WBC launch's waves (counter / expected_total, as qr_gate_kernel) and, while it is at it, for the
// all-gathers queued before the tick (g0 / e0, g1 / e1: the counts qrgpu_allgather_tau's one-thread launches bump behind each gather, per
// source-buffer slot; null when the context has no communicator) -- they were queued a whole tick ago, so this costs the tick nothing, and the
// fence in front of the next tick (qrgpu_allgather_fence) finds its gather already waited for and queues no launch of its own.
// `tick_done` (or null) is bumped once the waits are over: the tick is complete in stream order -- the second WBC pass is ahead of this launch on
// the stream -- which is what the all-gather of its torques polls for (qrgpu_allgather_tau_of_tick).
// (`lane_done` / `lane_expect`, or null: an overlapped tick's launches on its lane's stream -- trailing list launch, second WBC pass -- are through.)
__global__ void qr_join_kernel(int *counter, int expected_total, long long max_ticks, int *timed_out, int *g0, int e0, int *g1, int e1, int *tick_done,
                               int *lane_done, int lane_expect, long long *dbg)
{
    if (threadIdx.x != 0) return;
    auto reached = [](int *p, int e) { return (int)((unsigned)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)e) >= 0; };
    long long t0 = wall_clock64();
    if (dbg) { dbg[0] = t0; dbg[1] = 0; dbg[2] = expected_total; dbg[3] = lane_expect; }
    while (!reached(counter, expected_total) || (lane_done && !reached(lane_done, lane_expect))) {
        if (wall_clock64() - t0 >= max_ticks) {
            if (dbg) { dbg[1] = wall_clock64(); dbg[4] = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dbg[5] = lane_done ? __hip_atomic_load(lane_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1; dbg[6] = 1; }
            if (timed_out) { *timed_out = 1; __threadfence_system(); } return; }
        __builtin_amdgcn_s_sleep(16);
    }
    // the gathers: another rank may be late with its side of the collective (the first one also sets up RCCL's connections), so this wait is
    // a patient one -- 30 s, then it is a hung collective and is reported as such
    t0 = wall_clock64();
    while ((g0 && !reached(g0, e0)) || (g1 && !reached(g1, e1))) {
        if (wall_clock64() - t0 >= 3000000000LL) { if (timed_out) { *timed_out = 1; __threadfence_system(); } return; }
        __builtin_amdgcn_s_sleep(16);
    }
    if (tick_done) __hip_atomic_fetch_add(tick_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dbg) { dbg[1] = wall_clock64(); dbg[6] = 0; }
}
// Probe of qrgpu_set_tick_overlap: do two streams of this process run side by side?  `wait` spins until `flag` is set (by `set`, queued
// afterwards on the other stream) or the bound passes, and says which in out[0].
__global__ void qr_probe_wait_kernel(int *flag, int *out, long long max_ticks, int token)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    bool ok = false;
    while (!(ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == token) && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(8);
    out[0] = ok ? 1 : 2;
}
__global__ void qr_probe_set_kernel(int *flag, int token)          // (a token per probe: the flag words are used again and again)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

#endif

}  // namespace qrgpu
