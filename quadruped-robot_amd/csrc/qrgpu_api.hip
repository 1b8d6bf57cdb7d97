// ============================================================================
// libqrgpu.so host side: the C ABI of include/qrgpu.h on top of the HIP runtime.
// No torch, no CPU compute path: every solve is a kernel launch on gfx950.
// ============================================================================
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "qrgpu_ctx.h"

namespace qrgpu {
struct MpcIO {
    const int *type_id;
    const float *g_state, *g_traj, *g_gait, *g_q;
    float *g_force, *g_tau;
    int *g_status;
    float *dbgH, *dbgG, *g_force_wbc;
    int force_stride;
    long long *dbgT;
};
template <int MAXB, bool BIG, bool LIST, int NTHR, int MINW = 0, bool H16 = (MAXB > 4)> __global__ void qr_mpc_kernel(MpcLaunch P, MpcIO io);
extern template __global__ void qr_mpc_kernel<2, false, false, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<4, false, false, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<4, true, true, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<2, true, false, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<2, true, false, 512, 4, true>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<2, false, false, 512, 4, true>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<9, true, false, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<9, true, false, 256, 2>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<5, true, false, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<9, true, true, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel<4, true, true, 256, 2, true>(MpcLaunch, MpcIO);
template <int MAXB, bool BIG, int NTHR, int MINW = 0> __global__ void qr_mpc_persist_kernel(MpcLaunch P, MpcIO io);
extern template __global__ void qr_mpc_persist_kernel<2, false, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_persist_kernel<5, true, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_persist_kernel<9, true, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_persist_kernel<9, true, 256, 2>(MpcLaunch, MpcIO);
// the same kernels with the executed-arithmetic counters compiled in (qr_mpc_kernel_fl.hip)
template <int MAXB, bool BIG, bool LIST, int NTHR, int MINW = 0, bool H16 = (MAXB > 4)> __global__ void qr_mpc_kernel_fl(MpcLaunch P, MpcIO io);
extern template __global__ void qr_mpc_kernel_fl<2, false, false, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<4, false, false, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<4, true, true, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<2, true, false, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<2, true, false, 512, 4, true>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<2, false, false, 512, 4, true>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<9, true, false, 256>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<9, true, false, 256, 2>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<5, true, false, 512>(MpcLaunch, MpcIO);
extern template __global__ void qr_mpc_kernel_fl<9, true, true, 256>(MpcLaunch, MpcIO);
__global__ void qr_join_kernel(int *counter, int expected_total, long long max_ticks, int *timed_out, int *g0, int e0, int *g1, int e1, int *tick_done,
                               int *lane_done, int lane_expect, long long *dbg);
__global__ void qr_gate2_kernel(int *c0, int e0, int *c1, int e1, long long max_ticks, long long *stamp);
__global__ void qr_probe_wait_kernel(int *flag, int *out, long long max_ticks, int token);
__global__ void qr_probe_set_kernel(int *flag, int token);
__global__ void qr_selftest_kernel(double *out);
__global__ void qr_lpt_order_kernel(int n, const int *cost, int *order, const int *ftime, int *wbc_order);
__global__ void qr_gait_kernel(int n, GaitDesc D, float currentTime, int stop, int fresh, const float *g_contact, float *st, float *g_out, float *g_fe);
__global__ void qr_swing_velocity_kernel(int n, EstimatorDesc D, SwingVelDesc V, const float *g_in, float *g_out);
__global__ void qr_gate_kernel(int *counter, int expected_total, long long max_ticks, int *timed_out, int timed_out_value, int *bump);
__global__ void qr_ground_kernel(int n, int fresh, const float *g_in, double *g_st, float *g_out, float *g_est_in);
__global__ void qr_walk_gait_kernel(int n, WalkDesc D, float currentTime, int stop, int fresh, const float *g_contact, float *st, float *g_out, float *g_ratio,
                                    float *g_vmc_in);
__global__ void qr_swing_kernel(int n, EstimatorDesc D, const float *g_in, float *g_cmd, float *g_tgt_world, float *g_qdes);
__global__ void qr_foothold_kernel(int n, FootholdDesc D, const float *g_in, const float *g_gait_state, const float *g_gait_out, float *g_swing);
__global__ void qr_pack_state_kernel(int n, float c0, float c1, float c2, const float *g_in, const float *g_est, const float *g_rpy, float *g_mpc, float *g_fb);
__global__ void qr_estimator_kernel(int n, EstimatorDesc D, const float *g_in, const unsigned *g_tick, double *st, float *g_out);
__global__ void qr_vmc_kernel(VmcLaunch P, const int *type_id, const float *g_in, const float *g_q, float *g_force, float *g_tau, int *g_status);
__global__ void qr_frontend_kernel(int n, int horizon, int numHorizonL, float dt, float dtMPC, const float *fin, float *fst, float *g_traj,
                                   float *g_gait, float *g_cmd, int *g_updated);
__global__ void qr_wbc_kernel(int n, const WbcConst *types, const int *type_id, const float *g_state, const float *g_cmd,
                              float *g_prev, float *g_tau, float *g_qdes, int *g_status, float *g_dbg, int merge_tau, int status_or, long long *dbgT,
                              const float *g_fr, int type_ready, int epilogue, float *g_qp, WbcPipe pipe);
__global__ void qr_wbc_kernel_dbg(int n, const WbcConst *types, const int *type_id, const float *g_state, const float *g_cmd,
                                  float *g_prev, float *g_tau, float *g_qdes, int *g_status, float *g_dbg, int merge_tau, int status_or, long long *dbgT,
                                  const float *g_fr, int type_ready, int epilogue, float *g_qp, WbcPipe pipe);
}

// MPC kernel variants: 0 = <5, BIG, ., 512> (h <= 16), 1 = <9, BIG, ., 256> (h <= 16, A/B), 2 = <4, ., ., 256> (h <= 11, A/B), 3 = <2, ., ., 512> (h <= 11
// main pass), 4 = <4, BIG, LIST, 256> (list launches), 5 = <2, BIG, ., 512> (planned list, one robot per workgroup); fl: the counting build
static const void *mpc_fn(int var, bool fl)
{
    switch (var) {
    case 0: return fl ? (const void *)qr_mpc_kernel_fl<5, true, false, 512> : (const void *)qr_mpc_kernel<5, true, false, 512>;
    case 1: return fl ? (const void *)qr_mpc_kernel_fl<9, true, false, 256> : (const void *)qr_mpc_kernel<9, true, false, 256>;
    case 2: return fl ? (const void *)qr_mpc_kernel_fl<4, false, false, 256> : (const void *)qr_mpc_kernel<4, false, false, 256>;
    case 3: return fl ? (const void *)qr_mpc_kernel_fl<2, false, false, 512> : (const void *)qr_mpc_kernel<2, false, false, 512>;
    case 4: return fl ? (const void *)qr_mpc_kernel_fl<4, true, true, 256> : (const void *)qr_mpc_kernel<4, true, true, 256>;
    case 6: return (const void *)qr_mpc_persist_kernel<2, false, 512>;        // persistent forms of 3 and 0 (no counting build of these)
    case 7: return (const void *)qr_mpc_persist_kernel<5, true, 512>;
    case 8: return fl ? (const void *)qr_mpc_kernel_fl<9, true, true, 256> : (const void *)qr_mpc_kernel<9, true, true, 256>;     // h > 11, list launches
    case 9: return (const void *)qr_mpc_persist_kernel<9, true, 256>;       // persistent form of 1
    case 10: return fl ? (const void *)qr_mpc_kernel_fl<9, true, false, 256, 2> : (const void *)qr_mpc_kernel<9, true, false, 256, 2>;   // 1 within 256 registers (two per CU)
    case 11: return (const void *)qr_mpc_persist_kernel<9, true, 256, 2>;   // persistent form of 10
    case 13: return fl ? (const void *)qr_mpc_kernel_fl<2, false, false, 512, 4, true> : (const void *)qr_mpc_kernel<2, false, false, 512, 4, true>;   // ... 64 working-set positions
    case 14: return (const void *)qr_mpc_kernel<4, true, true, 256, 2, true>;       // h <= 11 list launches on HALF a CU (overlapped ticks): S^-1 in the global scratch
    case 12: return fl ? (const void *)qr_mpc_kernel_fl<2, true, false, 512, 4, true> : (const void *)qr_mpc_kernel<2, true, false, 512, 4, true>;   // h <= 16 two to a CU on eight waves
    default: return fl ? (const void *)qr_mpc_kernel_fl<2, true, false, 512> : (const void *)qr_mpc_kernel<2, true, false, 512>;
    }
}
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function (per device), not to a context: the cache is process-wide and the
// limit is only ever raised, so that a second context asking for less cannot lower it under the first one's launches.
static int mpc_ensure_lds(qrgpu_ctx *c, int var, bool fl, int bytes)
{
    static std::mutex mu;
    static int configured[16][2][16];          // [device][counting build][variant], zero-initialised
    std::lock_guard<std::mutex> lk(mu);
    int &have = configured[c->device & 15][fl ? 1 : 0][var];
    if (have >= bytes) return QRGPU_OK;
    HIPCHK(c, hipFuncSetAttribute(mpc_fn(var, fl), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
    return QRGPU_OK;
}

static int mpc_main_wgs()
{   // workgroups of the h <= 11 main pass per CU: 2 (80 KB each: every robot fits) or 3 (53 KB: robots above ~32 stance leg-steps go to the list launches)
    static const int v = [] { const char *e = lab_env("QRGPU_MAIN_WGS"); const int k = e ? atoi(e) : 2; return (k == 3) ? 3 : 2; }();
    return v;
}

static int mpc_lds_bytes(const qrgpu_ctx *ctx, int h, bool inspection = false)
{
    // Packed inverse Hessian for the all-stance worst case plus room for S^-1; two (or three) workgroups
    // per CU when that fits, otherwise the whole CU.
    const size_t fixed = mpc_lds_fixed_bytes(h, true);
    const size_t nmax = 12 * (size_t)h;
    const size_t mp = 8 * (nmax * (nmax + 1) / 2);
    const size_t want = fixed + mp + 8 * (size_t)(24 * 25 / 2);     // at least a 24-row S^-1 in the worst case
    const size_t cu = (size_t)ctx->lds_per_cu;
    if (want <= cu / 2) return (4 * h <= 44 && mpc_main_wgs() == 3 && !inspection) ? (int)((cu / 3) & ~(size_t)15) : (int)(cu / 2);
    return (int)cu;
}

struct TimerScope {
    qrgpu_ctx *c; int k; bool on; hipStream_t s;
    TimerScope(qrgpu_ctx *ctx, int kernel, hipStream_t stream = nullptr, bool enabled = true) : c(ctx), k(kernel), on(ctx->timing && enabled), s(stream ? stream : ctx->stream)
    {
        if (on && ctx->timing_every > 1 && (ctx->ev_calls[kernel]++ % (unsigned)ctx->timing_every) != 0) on = false;
        if (!on) return;
        if (c->ev_used[k] == c->ev[k].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            c->ev[k].push_back({a, b});
        }
        hipEventRecord(c->ev[k][c->ev_used[k]].first, s);
    }
    ~TimerScope()
    {
        if (!on) return;
        hipEventRecord(c->ev[k][c->ev_used[k]].second, s);
        c->ev_used[k]++;
    }
};

// ---------------------------------------------------------------------------------------------
// BuildDynamicModel (QS/robots/qr_robot_a1_sim.cpp:176-343; the Lite3 file is a literal copy) reduced
// to rigid-body parameters.  Literals are the reference's float literals, evaluated in double.
// ---------------------------------------------------------------------------------------------
namespace {
struct RB { double m, h[3], I[6]; };     // I: xx yy zz xy xz yz about the frame origin
RB make_rb(double m, const double c[3], const double Ic[9])
{   // SpatialInertia(mass, com, inertia), QI/dynamics/spatial.hpp:390-398: Ibar = I + m [c]x[c]x^T
    RB r; r.m = m;
    for (int i = 0; i < 3; ++i) r.h[i] = m * c[i];
    const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    double Ib[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Ib[i][j] = Ic[3 * i + j] + m * ((i == j ? cc : 0.0) - c[i] * c[j]);
    r.I[0] = Ib[0][0]; r.I[1] = Ib[1][1]; r.I[2] = Ib[2][2]; r.I[3] = Ib[0][1]; r.I[4] = Ib[0][2]; r.I[5] = Ib[1][2];
    return r;
}
RB flip_y(const RB &a)
{   // flipAlongAxis(Y), spatial.hpp:505-534: mirror y
    RB r = a;
    r.h[1] = -a.h[1];
    r.I[3] = -a.I[3];
    r.I[5] = -a.I[5];
    return r;
}
RB add_rb(const RB &a, const RB &b)
{
    RB r; r.m = a.m + b.m;
    for (int i = 0; i < 3; ++i) r.h[i] = a.h[i] + b.h[i];
    for (int i = 0; i < 6; ++i) r.I[i] = a.I[i] + b.I[i];
    return r;
}
void store_rb(double *dst, const RB &r)
{
    dst[0] = r.m; dst[1] = r.h[0]; dst[2] = r.h[1]; dst[3] = r.h[2];
    for (int i = 0; i < 6; ++i) dst[4 + i] = r.I[i];
}
void build_wbc_const(const qrgpu_model_desc &d, WbcConst &K)
{
    auto F = [](double x) { return (double)(float)x; };      // the reference's literals are floats
    const double u = F(1e-6);
    const double abadI[9] = {F(469.2) * u, F(-9.4) * u, F(-0.342) * u, F(-9.4) * u, F(807.5) * u, F(-0.466) * u, F(-0.342) * u, F(-0.466) * u, F(552.9) * u};
    const double abadC[3] = {F(-0.0033), 0, 0};
    const double hipI[9] = {F(5529) * u, F(4.825) * u, F(343.9) * u, F(4.825) * u, F(5139.3) * u, F(22.4) * u, F(343.9) * u, F(22.4) * u, F(1367.8) * u};
    const double hipC[3] = {F(-0.003237), F(-0.022327), F(-0.027326)};
    const double kneeI[9] = {F(2998) * u, 0, F(-141.2) * u, 0, F(3014) * u, 0, F(-141.2) * u, 0, F(32.4) * u};
    const double kneeC[3] = {F(0.006435), 0, F(-0.107)};
    const double bodyI[9] = {F(15853) * u, 0, 0, 0, F(37799) * u, 0, 0, 0, F(45654) * u};
    const double zero3[3] = {0, 0, 0};
    const double m_abad = F(0.696), m_hip = F(1.013), m_knee = F(0.166), m_body = 6.0;
    const RB abadL = make_rb(m_abad, abadC, abadI), hipL = make_rb(m_hip, hipC, hipI), knee = make_rb(m_knee, kneeC, kneeI);
    const RB abadR = flip_y(abadL), hipR = flip_y(hipL);
    const RB base = make_rb(m_body, zero3, bodyI);
    // rotors (:193-198, :244-247): mass 1e-8, inertia (1e-2 * 1e-6) * identity  (setIdentity() overrides 33/33/63)
    const double k_rot = (double)(float)(F(1e-2) * 1e-6), m_rot = F(1e-8);
    auto rotor_at = [&](double x, double y, double z) {
        const double c[3] = {x, y, z};
        const double I[9] = {k_rot, 0, 0, 0, k_rot, 0, 0, 0, k_rot};
        return make_rb(m_rot, c, I);
    };
    RB base_eff = base;
    const double arx = F(0.14), ary = F(0.047);
    for (int leg = 0; leg < 4; ++leg) base_eff = add_rb(base_eff, rotor_at((leg < 2 ? 1 : -1) * arx, ((leg & 1) ? 1 : -1) * ary, 0.0));
    const double hry = F(0.04);
    const RB abadR_eff = add_rb(abadR, rotor_at(0, -hry, 0)), abadL_eff = add_rb(abadL, rotor_at(0, hry, 0));
    const RB hipR_eff = add_rb(hipR, rotor_at(0, 0, 0)), hipL_eff = add_rb(hipL, rotor_at(0, 0, 0));
    store_rb(K.rb[QR_RB_BASE], base);          store_rb(K.rb[QR_RB_BASE_EFF], base_eff);
    store_rb(K.rb[QR_RB_ABAD + 0], abadR);     store_rb(K.rb[QR_RB_ABAD + 1], abadL);
    store_rb(K.rb[QR_RB_ABAD_EFF + 0], abadR_eff); store_rb(K.rb[QR_RB_ABAD_EFF + 1], abadL_eff);
    store_rb(K.rb[QR_RB_HIP + 0], hipR);       store_rb(K.rb[QR_RB_HIP + 1], hipL);
    store_rb(K.rb[QR_RB_HIP_EFF + 0], hipR_eff); store_rb(K.rb[QR_RB_HIP_EFF + 1], hipL_eff);
    store_rb(K.rb[QR_RB_KNEE], knee);
    K.abad_loc[0] = F(0.1805); K.abad_loc[1] = F(0.047); K.abad_loc[2] = 0.0;
    K.hip_l = d.hip_l; K.upper_l = d.upper_l; K.lower_l = d.lower_l; K.foot_y = F(0.004);
    K.k_rot = k_rot;
    const double pi_f = (double)(float)M_PI;                   // coordinateRotation(Z, float(M_PI)) (:299)
    K.hiprot_ex = -std::sin(pi_f); K.hiprot_ey = std::cos(pi_f);
    const double total = m_body + 4.0 * (m_abad + m_hip + m_knee);                  // totalNonRotorMass()
    K.max_fz = (double)(float)total * (double)9.81f;
    K.kp_pos = d.kp_body_pos; K.kd_pos = d.kd_body_pos; K.kp_ori = d.kp_body_ori; K.kd_ori = d.kd_body_ori;
    K.kp_foot = d.kp_foot; K.kd_foot = d.kd_foot;
    K.w_fb = d.weight_fb; K.w_fr = d.weight_fr; K.mu = d.mu;
}
}  // namespace

extern "C" {

void qrgpu_model_desc_default(qrgpu_model_desc *d)
{
    d->hip_l = 0.08505f; d->upper_l = 0.2f; d->lower_l = 0.2f;
    d->body_size[0] = 0.267f; d->body_size[1] = 0.194f; d->body_size[2] = 0.114f;
    d->kp_body_pos = 100.f; d->kd_body_pos = 10.f; d->kp_body_ori = 100.f; d->kd_body_ori = 10.f;
    d->kp_foot = 500.f; d->kd_foot = 10.f; d->weight_fb = 0.1f; d->weight_fr = 1.f; d->mu = 0.4f;
}

// The side stream carries the planned list launch -- a few workgroups that each need a whole CU -- beside the main pass.  Highest priority,
// so that they are placed while the CUs are still empty: at default priority the main pass's workgroups fill every CU first and a listed
// robot starts 80-160 us late, which is then the end of the launch (QRGPU_SIDE_PRIORITY=0 for the default priority).
static hipError_t create_side_stream(hipStream_t *s)
{
    static const int want = [] { const char *e = lab_env("QRGPU_SIDE_PRIORITY"); return e ? atoi(e) : 1; }();
    int least = 0, greatest = 0;
    if (want && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// A lane's buffers, counters and (lanes 1, 2) streams.  Counters start at zero and are never cleared.
static int lane_create(qrgpu_ctx *c, Lane &L, bool own_stream, bool masked = false)
{
    if (L.d_order) return QRGPU_OK;
    const size_t nb = (size_t)c->max_batch;
    auto zalloc = [](auto **p, size_t bytes) { return hipMalloc((void **)p, bytes) == hipSuccess && hipMemset(*p, 0, bytes) == hipSuccess; };
    bool ok = hipMalloc(&L.d_order, sizeof(int) * 2 * nb) == hipSuccess && zalloc(&L.d_rescue, sizeof(int) * (nb + 2)) && zalloc(&L.d_pre, sizeof(int) * (2 * nb + 4)) &&
              zalloc(&L.d_skip, nb) && hipHostMalloc((void **)&L.h_pre_count, 4 * sizeof(int), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void **)&L.d_pre_hint, L.h_pre_count, 0) == hipSuccess && zalloc(&L.d_started, sizeof(int)) &&
              ((own_stream && !masked) || masked || create_side_stream(&L.side_stream) == hipSuccess) && zalloc(&L.d_done_flag, sizeof(unsigned) * nb) && zalloc(&L.d_qhead, 16 * sizeof(int)) &&
              zalloc(&L.d_planned_done, sizeof(int)) && zalloc(&L.d_go, (1 + QR_ABORT_RING) * sizeof(int)) && zalloc(&L.d_lane_done, sizeof(int)) && zalloc(&L.d_main_done, sizeof(int)) && zalloc(&L.d_rescue_taken, 4 * sizeof(int)) &&
              hipMalloc(&L.d_cmd_tick, sizeof(float) * 12 * nb) == hipSuccess &&
              hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming) == hipSuccess;
    if (ok && own_stream) {
        const uint32_t words = (uint32_t)((c->num_cu + 31) / 32);
        ok = (masked ? (hipExtStreamCreateWithCUMask(&L.stream, words, c->mask16_main) == hipSuccess && hipExtStreamCreateWithCUMask(&L.side_stream, words, c->mask16_side) == hipSuccess)
                     : hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess);
        L.own_stream = ok; L.masked = ok && masked;
    }
    if (ok && masked) ok = hipMemset(L.d_rescue + 2, 0xff, sizeof(int) * nb) == hipSuccess;      // (an entry reads -1 until it is written: MpcLaunch::rescue_taken)
    if (ok) { L.h_pre_count[0] = L.h_pre_count[1] = L.h_pre_count[2] = L.h_pre_count[3] = 0; }       // ([2] of lane 0: a pipelined tick's join gave up waiting)
    (void)hipDeviceSynchronize();          // (the fills went to the default stream: none of the context's streams waits for that one)
    return ok ? QRGPU_OK : QRGPU_ERR_ALLOC;
}
static void lane_destroy(Lane &L)
{
    if (L.stream && L.own_stream) { (void)hipStreamSynchronize(L.stream); }
    if (L.side_stream && (!L.own_stream || L.masked)) { (void)hipStreamSynchronize(L.side_stream); hipStreamDestroy(L.side_stream); }     // (lanes 1, 2 borrow lane 0's)
    if (L.stream && L.own_stream) hipStreamDestroy(L.stream);
    if (L.d_order) hipFree(L.d_order);
    if (L.d_rescue) hipFree(L.d_rescue);
    if (L.d_pre) hipFree(L.d_pre);
    if (L.d_skip) hipFree(L.d_skip);
    if (L.h_pre_count) hipHostFree(L.h_pre_count);
    if (L.d_started) hipFree(L.d_started);
    if (L.d_done_flag) hipFree(L.d_done_flag);
    if (L.d_qhead) hipFree(L.d_qhead);
    if (L.d_planned_done) hipFree(L.d_planned_done);
    if (L.d_go) hipFree(L.d_go);
    if (L.d_lane_done) hipFree(L.d_lane_done);
    if (L.d_main_done) hipFree(L.d_main_done);
    if (L.d_rescue_taken) hipFree(L.d_rescue_taken);
    if (L.d_cmd_tick) hipFree(L.d_cmd_tick);
    if (L.ev_fork) hipEventDestroy(L.ev_fork);
    if (L.ev_join) hipEventDestroy(L.ev_join);
    L = Lane{};
}

// Once per process: any QRGPU_* variable of the environment that is neither a supported switch (include/qrgpu.h) nor bench.py's own (QRGPU_BENCH_*)
// is reported on the standard error -- a laboratory switch without QRGPU_LAB=1 is ignored, a misspelt one never did anything.
extern char **environ;
static void warn_unknown_env()
{
    static std::once_flag once;
    std::call_once(once, [] {
        static const char *supported[] = {QRGPU_SUPPORTED_ENV};
        static const char *labs[] = {QRGPU_LAB_ENV};
        const bool lab_on = lab_env("QRGPU_LAB") != nullptr;
        for (char **e = environ; e && *e; ++e) {
            if (strncmp(*e, "QRGPU_", 6) != 0 || strncmp(*e, "QRGPU_BENCH_", 12) == 0) continue;
            const char *eq = strchr(*e, '=');
            const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
            bool ok = false, is_lab = false;
            for (const char *s_ : supported) if (name == s_) ok = true;
            for (const char *s_ : labs) if (name == s_) is_lab = true;
            if (ok || (is_lab && lab_on)) continue;
            if (is_lab) fprintf(stderr, "libqrgpu: %s is a laboratory switch: ignored unless QRGPU_LAB=1 is set (include/qrgpu.h lists the supported ones)\n", name.c_str());
            else fprintf(stderr, "libqrgpu: %s is not an environment switch of this library (include/qrgpu.h lists the supported ones): ignored\n", name.c_str());
        }
    });
}

int qrgpu_create(int device_id, int max_batch, int horizon_max, qrgpu_ctx **out)
{
    if (!out || max_batch <= 0 || horizon_max <= 0 || horizon_max > QRGPU_MAX_HORIZON) return QRGPU_ERR_BAD_ARG;
    warn_unknown_env();
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return QRGPU_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return QRGPU_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return QRGPU_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return QRGPU_ERR_NO_DEVICE;     // kernels are built for gfx950 only
    qrgpu_ctx *c = new qrgpu_ctx();
    c->device = device_id; c->max_batch = max_batch; c->horizon_max = horizon_max;
    c->name = prop.name; c->num_cu = prop.multiProcessorCount;
    c->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (c->lds_per_cu <= 0) c->lds_per_cu = 160 * 1024;
    const size_t in1 = 28 + 12 * QRGPU_MAX_HORIZON + 4 * QRGPU_MAX_HORIZON + 12 + 37 + 67 + 3;
    static_assert(28 + 12 * QRGPU_MAX_HORIZON + 4 * QRGPU_MAX_HORIZON + 12 + 37 + 67 + 3 <= 512, "staging layout");
    {
        const char *e = getenv("QRGPU_SINGLE_COPIES");
        c->zero_copy = !(e && atoi(e) != 0);
    }
    bool stage_ok;
    if (c->zero_copy) {
        // [0, 512) floats in, [512, 576) floats out, [576, 580) status / type words
        void *dp = nullptr;
        stage_ok = hipHostMalloc(&c->h_stage, 640 * sizeof(float), hipHostMallocMapped) == hipSuccess &&
                   hipHostGetDevicePointer(&dp, c->h_stage, 0) == hipSuccess;
        if (stage_ok) {
            memset(c->h_stage, 0, 640 * sizeof(float));
            c->h_in1 = (float *)c->h_stage; c->h_out1 = c->h_in1 + 512; c->h_st1 = (int *)(c->h_in1 + 576);
            c->d_in1 = (float *)dp; c->d_out1 = c->d_in1 + 512; c->d_st1 = (int *)(c->d_in1 + 576);
        }
    } else {
        stage_ok = hipMalloc(&c->d_in1, in1 * sizeof(float)) == hipSuccess && hipMalloc(&c->d_out1, 64 * sizeof(float)) == hipSuccess &&
                   hipMalloc(&c->d_st1, 4 * sizeof(int)) == hipSuccess;
    }
    auto zalloc = [](auto **p, size_t bytes) { return hipMalloc((void **)p, bytes) == hipSuccess && hipMemset(*p, 0, bytes) == hipSuccess; };
    bool ok = stage_ok && hipMalloc(&c->d_wbc, sizeof(WbcConst) * QR_MAX_TYPES) == hipSuccess;
    const size_t nb = (size_t)max_batch;
    ok = ok && zalloc(&c->d_cost[0], sizeof(int) * nb) && zalloc(&c->d_cost[1], sizeof(int) * nb) && hipMalloc(&c->d_warm, (size_t)QR_WARM_STRIDE * nb) == hipSuccess &&
         hipStreamCreateWithFlags(&c->wbc_stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_wbc_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_wbc_join, hipEventDisableTiming) == hipSuccess &&
         zalloc(&c->d_main_started, sizeof(int)) && zalloc(&c->d_tick_done, sizeof(int)) && zalloc(&c->d_gate_abort, QR_ABORT_RING * sizeof(int)) &&
         zalloc(&c->d_wbc_finished, sizeof(int)) && zalloc(&c->d_solved, sizeof(unsigned) * nb) && zalloc(&c->d_wbc_done, sizeof(unsigned) * nb) &&
         hipMalloc(&c->d_ftime, sizeof(int) * nb) == hipSuccess && hipMalloc(&c->d_wbc_order, 2 * sizeof(int) * nb) == hipSuccess;
    // lane 0 always; lanes 1 and 2 (streams of their own) when overlapped ticks are first switched on (qrgpu_set_tick_overlap)
    ok = ok && lane_create(c, c->lane[0], false) == QRGPU_OK;
    if (!ok) {
        qrgpu_destroy(c);
        return QRGPU_ERR_ALLOC;
    }
    {   // The compute stream is the context's own (non-blocking) unless the caller names one (qrgpu_set_stream; NULL there = the default stream).  On the
        // default stream two contexts of one process serialise each other's launches: 16.9 against 34.7 M WBC calls/s for two contexts of 512 robots.
        // QRGPU_OWN_STREAM=0: rounds 1-3's default.
        static const int own = [] { const char *e = lab_env("QRGPU_OWN_STREAM"); return e ? atoi(e) : 1; }();
        if (own && hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess) c->stream = c->own_stream;
    }
    c->lane[0].stream = c->stream;
    (void)hipDeviceSynchronize();          // (the fills of the counters above went to the default stream: none of the context's streams waits for that one)
    memset(&c->mpc, 0, sizeof(c->mpc));
    memset(c->wbc_host, 0, sizeof(c->wbc_host));
    *out = c;
    return QRGPU_OK;
}

void qrgpu_destroy(qrgpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->wbc_stream) (void)hipStreamSynchronize(c->wbc_stream);
    qrgpu_comm_destroy(c);
    for (int l = 0; l < QR_LANES; ++l) lane_destroy(c->lane[l]);         // (synchronises the lanes' own and side streams first)
    for (int k = 0; k < 2; ++k) for (auto &e : c->ev[k]) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : c->marks) hipEventDestroy(e);
    if (c->h_stage) hipHostFree(c->h_stage);
    else {
        if (c->d_in1) hipFree(c->d_in1);
        if (c->d_out1) hipFree(c->d_out1);
        if (c->d_st1) hipFree(c->d_st1);
    }
    if (c->d_wbc) hipFree(c->d_wbc);
    if (c->d_cost[0]) hipFree(c->d_cost[0]);
    if (c->d_cost[1]) hipFree(c->d_cost[1]);
    if (c->d_warm) hipFree(c->d_warm);
    if (c->d_flops) hipFree(c->d_flops);
    if (c->wbc_stream) hipStreamDestroy(c->wbc_stream);
    if (c->wbc_stream_hi) { (void)hipStreamSynchronize(c->wbc_stream_hi); hipStreamDestroy(c->wbc_stream_hi); }
    if (c->wbc_stream_16) { (void)hipStreamSynchronize(c->wbc_stream_16); hipStreamDestroy(c->wbc_stream_16); }
    for (int k = 0; k < 2; ++k) if (c->ev_call[k]) hipEventDestroy(c->ev_call[k]);
    if (c->ev_wbc_fork) hipEventDestroy(c->ev_wbc_fork);
    if (c->ev_wbc_join) hipEventDestroy(c->ev_wbc_join);
    if (c->d_main_started) hipFree(c->d_main_started);
    if (c->d_ftime) hipFree(c->d_ftime);
    if (c->d_wbc_finished) hipFree(c->d_wbc_finished);
    if (c->d_gate_abort) hipFree(c->d_gate_abort);
    if (c->d_solved) hipFree(c->d_solved);
    if (c->d_wbc_done) hipFree(c->d_wbc_done);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    if (c->d_gather_done) hipFree(c->d_gather_done);
    if (c->d_tick_done) hipFree(c->d_tick_done);
    if (c->d_timeline) hipFree(c->d_timeline);
    if (c->d_tlr) hipFree(c->d_tlr);
    if (c->d_wbc_order) hipFree(c->d_wbc_order);
    if (c->d_sinv_spill) hipFree(c->d_sinv_spill);
    delete c;
}

// another population: the dispatch order, the plan and the smoothed costs of every lane mean nothing any more
static void forget_history(qrgpu_ctx *c)
{
    for (auto &L : c->lane) { L.lpt_n = 0; L.plan_n = 0; }
    c->cost_n[0] = c->cost_n[1] = 0;
    c->ov_hold = 0;
}
int qrgpu_set_lpt_schedule(qrgpu_ctx *c, int on)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    c->lpt = on != 0;
    forget_history(c);
    return QRGPU_OK;
}
int qrgpu_set_warm_start(qrgpu_ctx *c, int on)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    c->warm = on != 0;
    c->warm_n = 0;                 // forget what is stored
    return QRGPU_OK;
}
int qrgpu_mpc_set_hessian_mode(qrgpu_ctx *c, int mode)
{
    if (!c || (mode != QRGPU_HESSIAN_F32 && mode != QRGPU_HESSIAN_BF16X3)) return QRGPU_ERR_BAD_ARG;
    c->mpc.hess_mode = mode;
    return QRGPU_OK;
}
int qrgpu_set_planned_list(qrgpu_ctx *c, int on, int big_nls)
{
    if (!c || big_nls < 0) return QRGPU_ERR_BAD_ARG;
    c->planned = on != 0;
    c->big_nls = big_nls;
    for (auto &L : c->lane) L.plan_n = 0;
    return QRGPU_OK;
}
int qrgpu_set_tick_pipeline(qrgpu_ctx *c, int on)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    c->pipeline = on != 0;
    return QRGPU_OK;
}
int qrgpu_set_rescue_pass(qrgpu_ctx *c, int on)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    c->rescue = on != 0;
    for (auto &L : c->lane) L.plan_n = 0;
    return QRGPU_OK;
}
int qrgpu_set_stream(qrgpu_ctx *c, void *s) { if (!c) return QRGPU_ERR_BAD_ARG; c->stream = (hipStream_t)s; c->lane[0].stream = c->stream; c->ov_chain = false; return QRGPU_OK; }
void *qrgpu_get_stream(qrgpu_ctx *c) { return c ? (void *)c->stream : nullptr; }
const char *qrgpu_last_error(const qrgpu_ctx *c) { return c ? c->err.c_str() : "null context"; }
int qrgpu_device_info(const qrgpu_ctx *c, char *name, int len, int *lds)
{
    if (!c) return 0;
    if (name && len > 0) { strncpy(name, c->name.c_str(), len - 1); name[len - 1] = 0; }
    if (lds) *lds = c->lds_per_cu;
    return c->num_cu;
}

int qrgpu_mpc_setup(qrgpu_ctx *c, int type_id, float dt, int horizon, float mu, float fmax, float mass,
                    const float inertia[3], const float weights[12], float alpha)
{
    if (!c || type_id < 0 || type_id >= QR_MAX_TYPES || !inertia || !weights) return QRGPU_ERR_BAD_ARG;
    if (horizon <= 0 || horizon > c->horizon_max) return QRGPU_ERR_BAD_ARG;
    // one horizon per context (the reference has one global problem size, qr_mpc_interface.cpp:35-104):
    // a different horizon re-sizes the problem and invalidates the other types' setup, as a second SetupProblem would
    if (c->mpc.horizon != horizon) for (int t = 0; t < QR_MAX_TYPES; ++t) if (t != type_id) c->mpc_ready[t] = false;
    MpcType &T = c->mpc.type[type_id];
    T.dt = dt; T.mu = mu; T.fmax = fmax; T.mass = mass; T.alpha = alpha;
    for (int i = 0; i < 3; ++i) T.inertia[i] = inertia[i];
    for (int i = 0; i < 12; ++i) T.weights[i] = weights[i];
    if (!c->wbc_ready[type_id]) { T.hip_l = 0.08505f; T.upper_l = 0.2f; T.lower_l = 0.2f; }
    c->mpc.horizon = horizon;
    c->mpc_ready[type_id] = true;
    return QRGPU_OK;
}

int qrgpu_wbc_setup(qrgpu_ctx *c, int type_id, const qrgpu_model_desc *desc)
{
    if (!c || type_id < 0 || type_id >= QR_MAX_TYPES || !desc) return QRGPU_ERR_BAD_ARG;
    build_wbc_const(*desc, c->wbc_host[type_id]);
    MpcType &T = c->mpc.type[type_id];
    T.hip_l = desc->hip_l; T.upper_l = desc->upper_l; T.lower_l = desc->lower_l;    // leg geometry for the MPC torque map
    c->wbc_ready[type_id] = true;
    c->wbc_dirty = true;
    return QRGPU_OK;
}

static int upload_wbc(qrgpu_ctx *c)
{
    if (!c->wbc_dirty) return QRGPU_OK;
    HIPCHK(c, hipMemcpyAsync(c->d_wbc, c->wbc_host, sizeof(WbcConst) * QR_MAX_TYPES, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->wbc_dirty = false;
    return QRGPU_OK;
}

static int ready_mask(const bool *r) { int m = 0; for (int t = 0; t < QR_MAX_TYPES; ++t) if (r[t]) m |= 1 << t; return m; }

// What an overlapped tick adds to its MPC launches (qrgpu_tick_batch): the epoch its solves leave in d_solved, whether they wait -- per robot -- for
// the previous tick's (chained), and the cost buffers they read and write.
struct OvLaunch { unsigned epoch; bool chained; unsigned prev_epoch; bool plan_tick; int *prev_started; unsigned prev_started_total; };
// bound of an overlapped tick's per-robot waits for its predecessor (20 ms; QRGPU_OV_WAIT_US: the give-up tests)
static long long ov_wait_ticks()
{
    static const long long v = [] { const char *e = getenv("QRGPU_OV_WAIT_US"); return e ? 100LL * atoll(e) : 2000000LL; }();
    return v;
}

static int launch_mpc(qrgpu_ctx *c, int n, const int *d_type, const float *d_state, const float *d_traj, const float *d_gait,
                      const float *d_q, float *d_force, float *d_tau, int *d_status, float *dH, float *dG, float *d_force_wbc, int epilogue = 0,
                      bool piped = false, int lane_id = 0, const OvLaunch *ov = nullptr)
{
    Lane &LN = c->lane[lane_id];
    if (!c || n <= 0 || n > c->max_batch || !d_state || !d_traj || !d_gait || !d_force) return QRGPU_ERR_BAD_ARG;
    if (d_tau && !d_q) return QRGPU_ERR_BAD_ARG;
    // without a type array every robot is type 0; with one, the kernel flags robots whose type was never set up (QRGPU_ST_BAD_TYPE)
    if (!(d_type ? ready_mask(c->mpc_ready) != 0 : c->mpc_ready[0])) return QRGPU_ERR_NOT_SETUP;
    HIPCHK(c, hipSetDevice(c->device));
    MpcLaunch P = c->mpc;
    P.n = n;
    P.type_ready = ready_mask(c->mpc_ready);
    P.epilogue = epilogue;
    // pipelined tick: the solves raise per-robot flags for the WBC launch that runs beside them (qrgpu_tick_batch)
    P.done_flag = piped ? LN.d_done_flag : nullptr;
    P.done_epoch = c->tick_epoch;
    // overlapped tick (lanes 1, 2): per-robot hand-over of the warm-start and cost words between consecutive ticks (MpcLaunch::solved)
    const bool ovl = ov != nullptr;
    const bool ov16 = ovl && LN.masked;            // h > 11 overlapped: main pass on the lane's (masked) stream, planned AND trailing launch on its side stream (reserved CUs)
    if (!ovl) { c->ov_chain = false; c->cost_n[0] = c->cost_n[1] = 0; }      // (any other MPC launch: the next overlapped tick waits for the context's stream)
    int *const cost_out = c->d_cost[ovl ? (ov->epoch & 1u) : 0];
    const int *const cost_prev = ovl ? c->d_cost[(ov->epoch & 1u) ^ 1u] : cost_out;
    P.solved = ovl ? c->d_solved : nullptr; P.solved_epoch = ovl ? ov->epoch : 0u;
    P.prev_solved = (ovl && ov->chained) ? c->d_solved : nullptr; P.prev_epoch = ovl ? ov->prev_epoch : 0u;
    P.xtick_wait = ov_wait_ticks();
    P.main_started = piped ? c->d_main_started : nullptr;
    P.tl = piped ? c->d_timeline : nullptr;
    // QRGPU_WBC_ORDER=1 (an experiment, off by default): the solves also leave the moment they ended, from which the launch behind the main pass
    // sorts the NEXT tick's WBC order (robots in the order their solves ended) into the half of d_wbc_order that this tick's WBC launch is not
    // reading.  Measured: nothing at 1024 robots (4.15 against 4.16-4.20 M ticks/s: nine WBC workgroups in ten start AFTER their robot's solve
    // has ended, they are short of slots, not waiting for flags) and -6 % at 8192 (two rank sorts of 1024-robot chunks behind the main pass).
    // (The moments come from the timeline hooks: a library built with -DQR_TIMELINE only.)
#ifdef QR_TIMELINE
    static const int wbc_order_on = [] { const char *e = lab_env("QRGPU_WBC_ORDER"); return e ? atoi(e) : 0; }();
#else
    static const int wbc_order_on = 0;
#endif
    const bool wbc_ord = piped && wbc_order_on != 0 && n <= 16384;
    P.ftime = (wbc_ord || (piped && c->d_tlr)) ? c->d_ftime : nullptr;
    P.wbc_order_out = wbc_ord ? c->d_wbc_order + (size_t)(c->wbc_order_parity ^ 1) * (size_t)c->max_batch : nullptr;
    P.flops = (c->flops_on && !dH) ? c->d_flops : nullptr;
    if (P.flops) c->flops_n = n;
    // warm start from the slot's previous solve: not for inspection launches; a different batch size starts from nothing
    P.warm = (c->warm && !dH) ? c->d_warm : nullptr;
    if (P.warm && c->warm_n != n) {
        HIPCHK(c, hipMemsetAsync(c->d_warm, 0, (size_t)QR_WARM_STRIDE * (size_t)n, LN.stream));
        c->warm_n = n;
    }
    P.lds_bytes = mpc_lds_bytes(c, P.horizon, dH != nullptr);      // (inspection launches have no list pass behind them)
    // longest-first dispatch from the previous launch's per-robot cost; inspection launches (dH) and tiny batches keep slot order
    const bool lpt = c->lpt && n >= 64 && !dH;
    // (the order is two arrays: this tick's trailing launch sorts the next one into the half this tick's launches -- the chunked WBC launches of a large
    //  batch among them, WbcPipe::slot_base -- do not read)
    int *const order_next = LN.d_order + (size_t)(LN.order_parity ^ 1) * (size_t)c->max_batch;
    P.order = (lpt && LN.lpt_n == n) ? LN.d_order + (size_t)LN.order_parity * (size_t)c->max_batch : nullptr;
    LN.order_used = P.order;
    P.cost = lpt ? cost_out : nullptr;
    P.cost_in = cost_prev;
    { static const int ema = [] { const char *e = lab_env("QRGPU_COST_EMA"); return e ? atoi(e) : 1; }(); P.cost_ema = (lpt && ema && (ovl ? c->cost_n[(ov->epoch & 1u) ^ 1u] == n : LN.lpt_n == n)) ? 1 : 0; }
    if (ovl) c->cost_n[ov->epoch & 1u] = lpt ? n : 0;
    // up to 4 register-resident 3x3 blocks per thread cover tri(44) leg-step pairs (h <= 11); 9 cover h = 16
    const bool small = 4 * P.horizon <= 44;
    P.sinv_spill = nullptr;
    if (!small || ovl) {            // (overlapped ticks at h <= 11: the list launches run on half a CU with S^-1 in this scratch)
        if (!c->d_sinv_spill) HIPCHK(c, hipMalloc(&c->d_sinv_spill, sizeof(double) * (size_t)c->max_batch * (size_t)(QR_QH * (QR_QH + 1) / 2)));
        if (!small) P.sinv_spill = c->d_sinv_spill;
    }
    // Batches below 64 robots (the single-robot drop-in calls among them) have a CU per robot to themselves: they run the whole-CU eight-wave
    // variant <2, BIG, ., 512> (96 working-set positions, the CU's whole LDS) as their main pass, so that nothing is left for a trailing
    // list launch -- one launch instead of two on the single-robot path, and the ping-pong parity of the rescue / planned lists, which
    // belongs to the batched calls' plan, is not touched by calls in between (a solve1 between two planned calls used to flip it
    // and the next planned call read the counters of the plan before last).  QRGPU_TINY_WHOLE_CU=0: the old two-launch form.
    static const int tiny_whole_cu = [] { const char *e = lab_env("QRGPU_TINY_WHOLE_CU"); return e ? atoi(e) : 1; }();
    const bool tiny = small && n < 64 && !dH && tiny_whole_cu != 0;
    if (tiny) P.lds_bytes = c->lds_per_cu;
    // rescue pass for the h <= 11 main pass (not for inspection launches or tiny batches)
    // h > 11, batches of 3.5 robots per CU and more (QRGPU_H16_TWO=0: never, =2: from 64 robots on): the main pass runs TWO workgroups per CU on half
    // the LDS each.  A trotting robot's inverse Hessian (<= 42 stance leg-steps at h = 16: <= 65 KB) fits, and its 903 blocks are two per thread
    // of the EIGHT-wave build and sweep of the h <= 11 main pass -- <2, BIG, ., 512, 4, H16>, within 128 registers (four waves leave after the
    // sweep; QRGPU_H16_TWO_WAVES=4: the four-wave kernel within 256 registers, <9, BIG, ., 256, 2> -- under that kernel's default of one wave
    // per SIMD the compiler takes AGPRs on top of the 256 VGPRs and two workgroups never share a CU, which is what the earlier attempts at
    // this measured without knowing).  S^-1 of every robot of the main pass lives in the global scratch (qcap 96 whatever the LDS holds: nobody
    // outgrows the main pass unannounced).  On whole CUs beside the main pass, one robot per eight-wave 256-register workgroup (planned list):
    // the robots whose inverse Hessian does not fit half a CU (three-leg and all-stance gaits: a class known from the gait table, 10 % of the
    // mixed shard) and the tick's long poles -- robots whose smoothed cost says 450 us and more two to a CU (60-80 active rows over the
    // spilled S^-1), which stay listed while they cost 300 us and more on a whole CU.
    // Mixed h = 16 shard, one workgroup per CU -> four waves two to a CU -> eight waves two to a CU: 1.37 -> 1.45 -> 1.52 M ticks/s at 1024
    // robots, 1.45 -> 1.68 -> 1.83 M at 2048, 1.50 -> 1.76 -> 1.97 M at 8192; below 3.5 robots per CU one workgroup per CU is faster (1.28
    // against 1.18 M at 768: fewer rounds than slots).
    static const int h16_two = [] { const char *e = getenv("QRGPU_H16_TWO"); return e ? atoi(e) : 1; }();
    // (it needs the list launches -- a robot of the big class has nowhere else to go -- and the cost words that carry the plan)
    bool two = !small && h16_two != 0 && !dH && n >= (h16_two >= 2 ? 64 : 7 * c->num_cu / 2) && c->rescue && c->planned && lpt;
    if (two) {
        // A shard in which most robots stand is a list, not a main pass: all stance is the class that cannot share a CU, and 1024 of them strided over by
        // the planned launch's workgroups (parked waves, three quarters of the CUs) run at 0.59 M ticks/s against 0.94 M one workgroup per CU
        // (60 % standing: 1.16 against 1.29 M; 30 %: 1.96 against 1.56 M; scratch/ab_h16_stand.py).  So when the list the host last saw is more
        // than 45 % of the batch the calls go back to one workgroup per CU for 31 calls; nobody plans meanwhile, so the call after them runs two
        // to a CU whatever the old count says (on the old plan: consistent, if stale) and the one after that decides on the fresh count.
        static const int hold_calls = [] { const char *e = getenv("QRGPU_H16_TWO_HOLD"); return e ? atoi(e) : 31; }();
        if (ov16) { }                              // (qrgpu_tick_batch has decided: an overlapped tick IS the two-to-a-CU form)
        else if (LN.two_hold > 0) { --LN.two_hold; two = false; }
        else if (LN.two_probe) LN.two_probe = false;
        else if (hold_calls > 0 && LN.plan_n == n && 20 * (long long)LN.h_pre_count[LN.rescue_parity] > 9 * (long long)n) { LN.two_hold = hold_calls; LN.two_probe = true; two = false; }
    }
    if (two) P.lds_bytes = (c->lds_per_cu / 2) & ~15;
    const bool rescue = c->rescue && !dH && (small || two) && !tiny;          // (the whole-CU h > 11 variant holds 96 rows itself)
    P.rescue_mode = 0;
    P.rescue_count = rescue ? LN.d_rescue : nullptr;
    P.rescue_list = rescue ? LN.d_rescue + 2 : nullptr;
    P.rescue_parity = LN.rescue_parity;
    P.lpt_cost_in = nullptr; P.lpt_order_out = nullptr;
    // rows enter the next tick's guess only when their multiplier exceeds 2 % of the solve's largest (weakly held rows are the ones that do
    // not persist: measured 0.2446 -> 0.2211 ms per launch at h = 10, neutral at h = 5; at h = 16, where a missing row costs 7-13 k cycles
    // to add, every threshold measured worse, so none is applied there).  QRGPU_WARM_UTHR overrides.
    { static const double wu = [] { const char *e = lab_env("QRGPU_WARM_UTHR"); return e ? atof(e) : -1.0; }(); P.warm_uthr = wu >= 0.0 ? wu : (4 * P.horizon <= 44 ? 0.02 : 0.0); }
    { static const int nb = [] { const char *e = lab_env("QRGPU_NO_BLOCK_DROP"); return e ? atoi(e) : 0; }(); P.no_block_drop = nb; }
    { static const int nw = [] { const char *e = lab_env("QRGPU_NO_WCACHE"); return e ? atoi(e) : 0; }(); P.no_wcache = nw; }
    // planned list: needs the trailing list launch (it plans) and the per-robot cost words (they carry the `big` bit)
    const bool planned = c->planned && rescue && lpt;
    P.pre_count = planned ? LN.d_pre : nullptr;
    // (h > 11 overlapped: the list is two lists, by the parity the counters ping-pong on -- MpcLaunch::pre_list_next)
    P.pre_list = planned ? LN.d_pre + 4 + ((ov16 && LN.rescue_parity) ? c->max_batch : 0) : nullptr;
    P.pre_list_next = ov16 ? (LN.rescue_parity ? -c->max_batch : c->max_batch) : 0;
    P.pre_hint = planned ? LN.d_pre_hint : nullptr;
    P.skip = nullptr;
    P.big_nls = c->big_nls;
    if (two) {
        // the class that cannot be solved on half a CU: stance leg-steps whose block-packed inverse Hessian does not fit the main pass's LDS
        const long long room = (long long)P.lds_bytes - (long long)mpc_lds_fixed_bytes(P.horizon, true);
        int k = 1;
        while (k <= 4 * P.horizon && (long long)k * (k + 1) / 2 * 72 <= room) ++k;
        if (k > 45) k = 45;                 // (and the eight-wave kernel holds two blocks per thread: 1024 >= tri(44))
        if (P.big_nls <= 0 || P.big_nls > k) P.big_nls = k;        // (a caller's own, stricter class rule stands: qrgpu_set_planned_list)
    }
    { static const int bm = [] { const char *e = lab_env("QRGPU_BIG_MARGIN"); return e ? atoi(e) : 6; }(); P.big_margin = two ? -1000 : bm; }
    P.big_cost = P.big_cost_stay = 0; P.planned_stride = 0;
    // (not in an overlapped tick: a long pole no longer sets a span there -- a tick has two periods to finish -- and the reserved CUs are for the
    //  robots that cannot run anywhere else; with the cost rule on, time spent WAITING counts as cost, the list grows and the reserved CUs fall behind)
    static const int ov16_cost = [] { const char *e = lab_env("QRGPU_OV16_COST"); return e ? atoi(e) : 1; }();
    if (two && (!ov16 || ov16_cost)) {
        // the long poles: a robot whose solve takes most of the tick's span two to a CU (a large working set over the spilled S^-1: 600-800 us
        // against a mean of 200) is planned onto a whole CU, and stays there while its solve costs more than QRGPU_H16_BIG_STAY_US there
        static const int big_us = [] { const char *e = getenv("QRGPU_H16_BIG_US"); return e ? atoi(e) : 450; }();
        static const int stay_us = [] { const char *e = getenv("QRGPU_H16_BIG_STAY_US"); return e ? atoi(e) : 300; }();
        P.big_cost = (int)((long long)big_us * 2250 / 256); P.big_cost_stay = (int)((long long)stay_us * 2250 / 256);
    }
    P.lds_main = P.lds_bytes;
    P.started = nullptr;
    if (planned && LN.plan_n != n) {                 // no plan for this batch size yet: nothing is skipped, both counters start at zero
        HIPCHK(c, hipMemsetAsync(LN.d_pre, 0, 4 * sizeof(int), LN.stream));
        HIPCHK(c, hipMemsetAsync(LN.d_skip, 0, (size_t)n, LN.stream));
    }
    // kernel variant: 3 = h <= 11, eight waves build and sweep (two blocks per thread, 128 VGPRs; the default), 2 = the same on four waves
    // (QRGPU_MAIN_THREADS=256, for A/B runs; six waves were measured too: the second workgroup of a CU then often cannot be placed until
    // the first has shed its extra waves), 1 = <9, positions 64..95 in a second register set> (h <= 16)
    static const int main_threads = [] { const char *e = lab_env("QRGPU_MAIN_THREADS"); return e ? atoi(e) : 512; }();
    static const int h16_threads = [] { const char *e = lab_env("QRGPU_H16_THREADS"); return e ? atoi(e) : 512; }();
    static const int two_waves = [] { const char *e = lab_env("QRGPU_H16_TWO_WAVES"); return e ? atoi(e) : 8; }();
    const int var = tiny ? 5 : (small ? (main_threads == 256 ? 2 : 3) : (two ? (two_waves == 8 ? 12 : (two_waves == 9 ? 13 : 10)) : (h16_threads == 256 ? 1 : 0)));
    // Overlapped ticks (h <= 11): the machine is never empty -- a workgroup that asks for a whole CU's LDS waits until both halves of some CU
    // happen to be free at once, behind every half-CU workgroup of the next tick's main pass and every WBC workgroup.  So the list launches of
    // an overlapped tick run on HALF a CU like the main pass, with S^-1 (96 rows) in the global scratch: <4, BIG, LIST, 256, 2, H16> striding
    // (variant 14) and <2, BIG, ., 512, 4, H16> one robot per workgroup (variant 12: the two-to-a-CU main pass of h > 11).
    // A tick whose lane has a PLAN is not chained (qrgpu_tick_batch): it starts on an empty machine, and its planned launch is the whole-CU one.
    const bool half_lists = small && ovl && !ov->plan_tick;
    const int list_var = small ? (half_lists ? 14 : 4) : 8;             // striding list kernel (trailing launch, long planned lists)
    const int one_var = small ? (half_lists ? 12 : 5) : 0;              // one listed robot per eight-wave workgroup
    const int list_lds = half_lists ? P.lds_bytes : c->lds_per_cu;
    const int one_lds = list_lds;
    // the instrumented kernels (counters, dense H / g, cycle stamps compiled in) only for a launch that asks for one of those
    const bool fl = P.flops != nullptr || dH != nullptr || dG != nullptr || c->d_dbg_cycles != nullptr;
    const void *fn = mpc_fn(var, fl);
    { const int rc_ = mpc_ensure_lds(c, var, fl, P.lds_bytes); if (rc_) return rc_; }
    if (rescue) { const int rc_ = mpc_ensure_lds(c, list_var, fl, list_lds); if (rc_) return rc_; }
    // Persistent main pass (qr_device_types.h): when the batch is more than the machine holds at once, launch one workgroup per resident slot
    // and let them take robots off per-XCD queues.  Default (QRGPU_PERSIST=1): the h > 11 variant only -- 1.31 -> 1.37 M ticks/s on the mixed
    // h = 16 shard.  At h <= 11 (QRGPU_PERSIST=2 to try) it loses what it gains and more: the four waves a solve no longer needs after its sweep
    // cannot leave a workgroup that has another robot to solve, they have to cross every barrier of the active set with the working ones
    // (a live wave counts at s_barrier), and a parked wave's wake-up, look at the exit word and return to the barrier is on the critical
    // path of each of the two hundred barriers of a solve: main pass 0.208 -> 0.231 ms at 1024 robots, 1.39 -> 1.46 ms at 8192.
    // QRGPU_PERSIST=0: one workgroup per robot everywhere, dispatched by the hardware in launch order.
    static const int persist_on = [] { const char *e = getenv("QRGPU_PERSIST"); return e ? atoi(e) : 1; }();
    int main_grid = 8 * ((n + 7) / 8);
    P.persist = 0; P.qhead = nullptr; P.qhead_next = nullptr;
    const void *main_fn = fn;
    if (persist_on && !fl && !tiny && (var == 0 || var == 1 || (var == 10 && persist_on >= 2) || (var == 3 && persist_on >= 2))) {
        const int pvar = var == 3 ? 6 : (var == 1 ? 9 : (var == 10 ? 11 : 7));
        { const int rc_ = mpc_ensure_lds(c, pvar, false, P.lds_bytes); if (rc_) return rc_; }
        if (c->main_slots[pvar][0] == 0 || c->main_slots_lds[pvar][0] != P.lds_bytes) {
            int nb = 0;
            HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mpc_fn(pvar, false), (var == 1 || var == 10) ? 256 : 512, (size_t)P.lds_bytes));
            c->main_slots[pvar][0] = nb > 0 ? nb : 1; c->main_slots_lds[pvar][0] = P.lds_bytes;
        }
        const int slots = 8 * ((c->main_slots[pvar][0] * c->num_cu + 7) / 8);
        if (main_grid > slots) {
            P.persist = 1;
            P.qhead = LN.d_qhead + 8 * LN.qhead_parity; P.qhead_next = LN.d_qhead + 8 * (LN.qhead_parity ^ 1);
            LN.qhead_parity ^= 1;
            main_grid = slots;
            main_fn = mpc_fn(pvar, false);
        }
    }
    MpcIO io;
    io.type_id = d_type; io.g_state = d_state; io.g_traj = d_traj; io.g_gait = d_gait; io.g_q = d_q; io.g_force = d_force; io.g_tau = d_tau;
    io.g_status = d_status; io.dbgH = dH; io.dbgG = dG; io.g_force_wbc = d_force_wbc; io.force_stride = 51; io.dbgT = (long long *)c->d_dbg_cycles;
    // the planned launch (and its two stream events) is only worth issuing when the last plan listed somebody: the list's length comes back
    // through pinned memory without a sync.  A stale zero just means the main pass solves everybody (P.skip stays null): consistent either way.
    if (planned && LN.plan_n != n) { LN.h_pre_count[0] = LN.h_pre_count[1] = 0; static const int ps = [] { const char *e = lab_env("QRGPU_PLAN_SYNC"); return e ? atoi(e) : 2; }(); LN.plan_sync_left = ps; }
    const bool have_plan = planned && LN.plan_n == n && LN.h_pre_count[LN.rescue_parity] > 0;
    // QRGPU_PLANNED_MODE: 0 = planned list on the context's side stream (fork / join events), 1 = planned list and main pass on the SAME
    // stream, the main pass launched with hipExtAnyOrderLaunch so that it may start before the list launch has finished: the list's
    // workgroups (each needs a whole CU's LDS) are dispatched first, the main pass's fill the rest of the machine
    static const int planned_mode = [] { const char *e = lab_env("QRGPU_PLANNED_MODE"); return e ? atoi(e) : 0; }();
    bool poll_join = false;
    P.planned_done = nullptr; P.planned_expect = 0;
    { static const bool dbg = lab_env("QRGPU_OV16_DEBUG") != nullptr;
      if (dbg && ov16) fprintf(stderr, "ov16 tick epoch %u lane %d chained %d: plan_n %d hint[%d] %d (other %d) have_plan %d\n", ov->epoch, lane_id, (int)ov->chained, LN.plan_n, LN.rescue_parity,
                               LN.h_pre_count[LN.rescue_parity], LN.h_pre_count[LN.rescue_parity ^ 1], (int)have_plan); }
    P.main_done = nullptr; P.main_done_expect = 0; P.rescue_taken = nullptr; P.linger = 0;
    if (ov16 && rescue) {
        // (h > 11 overlapped: the planned launch is also the tick's rescuer, plan or no plan -- MpcLaunch::main_done)
        LN.main_done_total += main_grid;
        P.main_done = LN.d_main_done; P.main_done_expect = LN.main_done_total; P.rescue_taken = LN.d_rescue_taken;
    }
    if (have_plan || (ov16 && rescue)) {
        // whole CU's LDS, 96 positions, workgroup b takes entries b, b + grid, ... of the list the last call's planning left
        P.skip = have_plan ? LN.d_skip : nullptr;
        MpcLaunch L = P;
        L.persist = 0; L.qhead = nullptr; L.qhead_next = nullptr;
        L.rescue_mode = 2; L.order = nullptr; L.rescue_count = nullptr; L.rescue_list = nullptr;
        L.lds_bytes = list_lds;                       // (the one-robot-per-workgroup form below: one_lds)
        L.sinv_spill = c->d_sinv_spill;               // (null at h <= 11; the whole-CU kernels of h > 11 put S^-1 there when an all-stance robot's M leaves no room)
        static const int gate_on = [] { const char *e = lab_env("QRGPU_PLANNED_GATE"); return e ? atoi(e) : 1; }();
        const bool gate = gate_on && planned_mode != 1 && !ov16;       // (reserved CUs: nothing to race the main pass for)
        L.started = (gate || ov16) ? LN.d_started : nullptr;
        int gate_expect = 0;
        int pgrid = n / 16;                            // a list of the all-stance twentieth of a batch gets a workgroup per robot
        pgrid = pgrid < 16 ? 16 : (pgrid > c->num_cu ? c->num_cu : pgrid);
        hipStream_t ls = planned_mode == 1 ? LN.stream : LN.side_stream;
        // QRGPU_PLANNED_WAVES=4: the four-wave list kernel, a workgroup striding over the list (this round's first form)
        static const int planned_waves = [] { const char *e = lab_env("QRGPU_PLANNED_WAVES"); return e ? atoi(e) : 8; }();
        // (big batches -- hundreds of listed robots at 8192 per launch -- stay on the striding kernel: one workgroup per robot would take every CU
        // from the main pass, and a stale short count would send most of the list to the trailing launch: 4.54 against 4.72 M ticks/s)
        // (h > 11 two to a CU: always the whole-CU kernel, on at most three quarters of the CUs -- a longer list is strided over, MpcLaunch::planned_stride)
        // (... unless most of the batch is listed -- a shard of standing robots: then the list is the launch, and it gets every CU)
        const int g3_cap = ov16 ? c->ov16_side_cus : ((two && 2 * LN.h_pre_count[LN.rescue_parity] <= n) ? 3 * c->num_cu / 4 : c->num_cu);
        const bool one_per_wg = planned_waves == 8 && (two || (n <= 2048 && LN.h_pre_count[LN.rescue_parity] <= (small ? c->num_cu / 4 : 3 * c->num_cu / 4)));
        // How the side stream learns that the context's stream has reached this call.  An event (QRGPU_PLANNED_FORK=1, and always for the
        // striding kernel and the ungated forms) costs ~10 us before the listed workgroups even launch -- 20 us between a tick's trailing launch and
        // the first workgroup of the next main pass on ticks that have a plan, against 2 on ticks that have none (the kernels' stamps).  Instead: a
        // one-thread launch on the side stream polls a "go" count that the gate in front of the main pass -- a launch on the context's stream --
        // bumps before it waits for the listed workgroups.  Bounded (50 ms, QRGPU_PLAN_GO_MS); a gate that gives up calls the plan off for
        // this call (MpcLaunch::plan_abort): nobody runs on inputs the caller's stream has not produced yet.
        static const int planned_fork = [] { const char *e = lab_env("QRGPU_PLANNED_FORK"); return e ? atoi(e) : 0; }();
        const bool poll_fork = !planned_fork && planned_mode != 1 && gate && one_per_wg;
        P.plan_abort = nullptr; P.plan_epoch = 0; L.plan_abort = nullptr; L.plan_epoch = 0;
        // ... and, in a pipelined tick, how the trailing launch learns that the planned launch is through (MpcLaunch::planned_done); QRGPU_PLANNED_JOIN=1: an event
        static const int planned_join = [] { const char *e = lab_env("QRGPU_PLANNED_JOIN"); return e ? atoi(e) : 0; }();
        poll_join = poll_fork && piped && !planned_join;
        // grid of the one-robot-per-workgroup launch: the list's length as the host last saw it, plus two (below)
        // (h > 11 two to a CU: the cost rule's share of the list comes and goes with the robots' smoothed costs, a dozen entries a tick -- and a
        //  robot handed to the trailing launch is a whole solve BEHIND the main pass: 1.10 M ticks/s with eight spare workgroups, 1.43 M with 24 or 48)
        static const int g3_extra = [] { const char *e = lab_env("QRGPU_PLANNED_EXTRA"); return e ? atoi(e) : -1; }();
        int g3 = LN.h_pre_count[LN.rescue_parity] + (g3_extra >= 0 ? g3_extra : (two ? 24 : 2));
        L.planned_stride = ov16 ? (have_plan ? 1 : 2) : ((two && g3 > g3_cap) ? 1 : 0);      // (2: no list, rescue only)
        g3 = g3 < 1 ? 1 : (g3 > g3_cap ? g3_cap : g3);
        // (how many of its workgroups stay for the hand-overs: all of them while there is no plan -- the whole big class arrives unannounced --
        //  then a few: one or two robots a tick change class, and a workgroup that stays keeps its CU from the next tick's planned launch)
        // (measured on the default configs[4] run, twice each: 16 stay 1.686 M ticks/s, 8: 1.698, 4: 1.715, 2: 1.720, 1: 1.729 -- a workgroup that stays keeps its CU
        //  from the next tick's planned launch; four is what is left of the margin for a tick in which a handful of robots change class at once)
        static const int linger_n = [] { const char *e = lab_env("QRGPU_OV16_LINGER"); return e ? atoi(e) : 4; }();
        LN.last_linger = ov16 ? (have_plan ? (linger_n < g3_cap ? linger_n : g3_cap) : g3_cap) : 0;
        // (QRGPU_OV_FAULT=2, the give-up test of MpcLaunch::main_done: nobody stays, as if every lingering workgroup had run into its bound -- a robot the
        //  main pass hands on afterwards is solved by nobody in that tick, and must carry QRGPU_ST_PIPE_TIMEOUT)
        static const bool linger_fault = [] { const char *e = getenv("QRGPU_OV_FAULT"); return e && atoi(e) == 2; }();
        if (linger_fault) LN.last_linger = 0;
        L.linger = LN.last_linger;
        if (ov16) g3 = g3_cap;                         // (the reserved CUs are this launch's whatever the list's length: it is also the tick's rescuer)
        bool main_gate_queued = false;
        if (poll_fork) {
            static const long long go_ticks = [] { const char *e = getenv("QRGPU_PLAN_GO_MS"); return 100000LL * (e ? atoll(e) : 50LL); }();
            ++LN.go_total;
            if (++LN.plan_epoch >= 0x7fffffff) LN.plan_epoch = 1;
            // (the give-up word is a ring indexed by the plan epoch: two planned ticks queued behind a backlog longer than twice the bound must not
            //  overwrite each other's word before their own kernels have read it)
            int *const abort_word = LN.d_go + 1 + (LN.plan_epoch & (QR_ABORT_RING - 1));
            P.plan_abort = abort_word; P.plan_epoch = LN.plan_epoch; L.plan_abort = P.plan_abort; L.plan_epoch = P.plan_epoch;
            // The gate in front of the main pass -- it gives the "go" -- is queued BEFORE the launch that polls for it: should the two streams
            // ever share a hardware queue (more streams in the process than the device has queues), a poller queued in front of what it polls for
            // would sit out its whole bound; this way round the worst case is the 30 us of the main pass's own gate.
            LN.started_total += g3;
            hipLaunchKernelGGL(qr_gate_kernel, dim3(1), dim3(64), 0, LN.stream, LN.d_started, LN.started_total, (long long)3000, (int *)nullptr, 0, LN.d_go);
            HIPCHK(c, hipGetLastError());
            main_gate_queued = true;
            hipLaunchKernelGGL(qr_gate_kernel, dim3(1), dim3(64), 0, LN.side_stream, LN.d_go, LN.go_total, go_ticks, abort_word, LN.plan_epoch, (int *)nullptr);
            HIPCHK(c, hipGetLastError());
        } else if (planned_mode != 1) {
            // (h > 11 overlapped: the lane's previous planned launch is through before anything of this tick runs -- it normally ended a tick ago, the tick's
            //  join having waited for the WBC workgroups of its robots; but a WBC workgroup that GAVE UP on a robot lets the join pass while the robot's
            //  solve is still going, and this tick's main pass clears the counters that launch's workgroups take their work from.  Found by fault injection:
            //  tests/test_gpu_overlap.py::test_h16_hand_overs_nobody_takes_are_never_silent)
            if (ov16 && LN.join_recorded) HIPCHK(c, hipStreamWaitEvent(LN.stream, LN.ev_join, 0));
            HIPCHK(c, hipEventRecord(LN.ev_fork, LN.stream));
            HIPCHK(c, hipStreamWaitEvent(LN.side_stream, LN.ev_fork, 0));
            if (ov16 && ov->chained && ov->prev_started) {
                // This tick's planned workgroups share the reserved CUs with its predecessor's, and each waits -- per robot -- for that robot's previous
                // solve: not one of them may start before EVERY planned workgroup of the predecessor has (a waiting workgroup holds its CU; one that
                // waits for a robot in the share of a workgroup that cannot start for lack of a CU never sees it: 144 robots timed out a tick, 5.8 ms).
                hipLaunchKernelGGL(qr_gate_kernel, dim3(1), dim3(64), 0, LN.side_stream, ov->prev_started, (int)ov->prev_started_total, (long long)5000000, (int *)nullptr, 0,
                                   (int *)nullptr);
                HIPCHK(c, hipGetLastError());
            }
        }
        if (one_per_wg) {
            // one robot per workgroup of the eight-wave whole-CU kernel; the grid is the list's length as the host last saw it (the kernel
            // hands a longer list's tail to the trailing launch)
            L.rescue_mode = 3; L.rescue_count = P.rescue_count; L.rescue_list = P.rescue_list;
            L.lds_bytes = one_lds;
            { const int rc_ = mpc_ensure_lds(c, one_var, fl, one_lds); if (rc_) return rc_; }
            if (poll_join) { LN.planned_done_total += g3; L.planned_done = LN.d_planned_done; }       // (every workgroup of the launch bumps it once)
            void *largs[2] = {(void *)&L, (void *)&io};
            HIPCHK(c, hipExtLaunchKernel(mpc_fn(one_var, fl), dim3(g3), dim3(512), largs, (size_t)L.lds_bytes, ls, nullptr, nullptr, 0));
            gate_expect = g3;
            if (!main_gate_queued) LN.started_total += g3;               // every workgroup of this launch bumps the counter once, sooner or later

        } else {
            L.started = nullptr;                  // (a long list on the striding kernel competes with the main pass as before: gating it would starve the main pass)
            void *largs[2] = {(void *)&L, (void *)&io};
            HIPCHK(c, hipExtLaunchKernel(mpc_fn(list_var, fl), dim3(pgrid), dim3(256), largs, (size_t)L.lds_bytes, ls, nullptr, nullptr, 0));
        }
        HIPCHK(c, hipGetLastError());
        if (planned_mode != 1 && !poll_join) { HIPCHK(c, hipEventRecord(LN.ev_join, LN.side_stream)); LN.join_recorded = true; }
        // the main pass waits (at most 30 us) until the listed robots' workgroups sit on their CUs
        if (gate && gate_expect > 0 && !main_gate_queued) {
            hipLaunchKernelGGL(qr_gate_kernel, dim3(1), dim3(64), 0, LN.stream, LN.d_started, LN.started_total, (long long)3000, (int *)nullptr, 0, (int *)nullptr);
            HIPCHK(c, hipGetLastError());
        }
    }
    {
        TimerScope ts(c, 0, LN.stream);
        const dim3 grid(main_grid);
        void *kargs[2] = {(void *)&P, (void *)&io};
        const unsigned flags = (have_plan && planned_mode == 1) ? hipExtAnyOrderLaunch : 0;
        const int threads = (var == 3 || var == 0 || var == 5 || var == 12 || var == 13) ? 512 : 256;
        HIPCHK(c, hipExtLaunchKernel(main_fn, grid, dim3(threads), kargs, (size_t)P.lds_bytes, LN.stream, nullptr, nullptr, flags));
    }
    HIPCHK(c, hipGetLastError());
    if (have_plan && planned_mode != 1 && !poll_join && !ov16) HIPCHK(c, hipStreamWaitEvent(LN.stream, LN.ev_join, 0));
    // (h > 11 overlapped: the trailing launch only sorts and plans -- eight small workgroups on the lane's stream, MpcLaunch::plan_only -- and the
    //  tick's planned launch takes the robots the main pass hands on, MpcLaunch::main_done.  A whole-CU trailing launch on the reserved CUs was
    //  measured first: it queues behind the NEXT tick's planned workgroups, 200-350 us instead of 8 -- and those may be waiting for the very
    //  robot it has yet to solve.)
    hipStream_t trail_stream = LN.stream;
    const bool plan_only = ov16 && rescue;
    if (rescue) {
        // trailing list launch: the robots whose working set outgrew the main pass (normally none: the workgroups sort the next call's
        // dispatch order, plan its list and exit) are re-solved with the whole CU's LDS and 96 working-set positions
        MpcLaunch R = P;
        R.persist = 0; R.qhead = nullptr; R.qhead_next = nullptr;
        R.planned_done = poll_join ? LN.d_planned_done : nullptr; R.planned_expect = LN.planned_done_total;
        R.rescue_mode = 1; R.order = nullptr; R.cost = nullptr;
        // (its robots go to the WBC pass queued behind it, not to the one running beside the main pass -- except in an overlapped tick, which has no
        //  second pass: there the robot's WBC workgroup waits for the flag this launch raises, WbcPipe::wait_list)
        R.done_flag = ovl ? LN.d_done_flag : nullptr; R.main_started = nullptr;
        R.skip = planned ? LN.d_skip : nullptr;          // (written by the planning workgroups; only the main pass reads it)
        R.lpt_cost_in = lpt ? cost_out : nullptr; R.lpt_order_out = lpt ? order_next : nullptr;
        R.lds_bytes = list_lds;
        R.sinv_spill = c->d_sinv_spill;
        // (a grid growing with the batch was tried: workgroups that ask for a whole CU's LDS are dispatched one every ~2 us, 0.55 ms for an
        // empty pass at 4096 robots)
        int rgrid = (half_lists || plan_only) ? 16 : (64 < n ? 64 : n);          // (chained ticks: every workgroup of this launch waits for a freed half CU)
        R.plan_only = plan_only ? 1 : 0;
        R.main_done = nullptr; R.rescue_taken = nullptr;
        if (plan_only) R.lds_bytes = 16384;          // (the sort's histogram; MpcLaunch::lds_main still says what the main pass holds)
        if (rgrid < 8 && lpt) rgrid = 8;
        io.dbgH = nullptr; io.dbgG = nullptr; io.dbgT = nullptr;
        void *rargs[2] = {(void *)&R, (void *)&io};
        HIPCHK(c, hipExtLaunchKernel(mpc_fn(list_var, fl), dim3(rgrid), dim3(256), rargs, (size_t)R.lds_bytes, trail_stream, nullptr, nullptr, 0));
        HIPCHK(c, hipGetLastError());
        // (the length of the list just planned reaches h_pre_count by itself).  A trailing launch that does not plan still flips the parity the
        // counters ping-pong on: whatever plan there was now sits under the wrong parity and is forgotten (the next planned call starts afresh)
        LN.plan_n = planned ? n : 0;
        LN.last_rescue_parity = LN.rescue_parity;
        LN.rescue_parity ^= 1;
    }
    LN.last_rescue_active = rescue;
    c->last_main_persist = P.persist != 0;
    if (piped) c->main_started_total += P.persist ? n : (int)(8 * ((n + 7) / 8));      // (persistent: one count per robot taken off a queue)
    if (lpt && rescue) { LN.lpt_n = n; LN.order_parity ^= 1; }               // sorted by workgroups 0-7 of the rescue launch
    else if (lpt) {
        hipLaunchKernelGGL(qr_lpt_order_kernel, dim3(8), dim3(256), 0, LN.stream, n, cost_out, order_next, (const int *)P.ftime, P.wbc_order_out);
        HIPCHK(c, hipGetLastError());
        LN.lpt_n = n; LN.order_parity ^= 1;
    }
    // (the WBC order is sorted by the launch behind the main pass -- the trailing list launch or qr_lpt_order_kernel; any other MPC launch on
    //  this context in between leaves the halves as they are and the next pipelined tick starts from slot order)
    if (wbc_ord && (lpt || rescue)) { c->wbc_order_parity ^= 1; c->wbc_order_n = n; }
    else c->wbc_order_n = 0;
    // The list's length reaches the host through pinned memory, unsynchronised: a caller that queues ticks faster than the GPU runs them
    // decides on a count several ticks old -- and on nothing at all for the first ticks of a new batch, whose listed robots then go through
    // the trailing launch, serially behind the main pass (a 20-step run lost a quarter of its rate on populations with an all-stance robot).
    // The first two calls after a history reset (one per parity) therefore end with a stream sync.
    // (Not while the stream is being captured into a graph: a sync is illegal there, and a replayed graph has a fixed launch shape anyway.)
    if (planned && LN.plan_sync_left > 0) {
        --LN.plan_sync_left;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(LN.stream, &cap) != hipSuccess) { cap = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
        if (cap == hipStreamCaptureStatusNone) HIPCHK(c, hipStreamSynchronize(LN.stream));
    }
    return QRGPU_OK;
}

static int launch_wbc(qrgpu_ctx *c, int n, const int *d_type, const float *d_state, const float *d_cmd, float *d_prev,
                      float *d_tau, float *d_qdes, int *d_status, float *d_dbg, int merge, int status_or, const float *d_fr = nullptr, int epilogue = 0,
                      float *d_qp = nullptr, hipStream_t stream_override = nullptr, WbcPipe pipe = WbcPipe{nullptr, 0u, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0, 0, 0, 0},
                      int grid_wgs = 0 /* > 0: one of the launches a large batch's WBC launch is cut into (WbcPipe::slot_base) */, bool timed = true)
{
    if (!c || n <= 0 || n > c->max_batch || !d_state) return QRGPU_ERR_BAD_ARG;
    if (!d_dbg && (!d_cmd || !d_prev || !d_tau)) return QRGPU_ERR_BAD_ARG;
    if (!(d_type ? ready_mask(c->wbc_ready) != 0 : c->wbc_ready[0])) return QRGPU_ERR_NOT_SETUP;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = upload_wbc(c);
    if (rc) return rc;
    const hipStream_t ws = stream_override ? stream_override : c->stream;
    if (!pipe.wbc_done) c->ov_chain = false;        // (any WBC launch but an overlapped tick's: the next overlapped tick waits for the context's stream)
    {
        TimerScope ts(c, 1, ws, !pipe.second && timed);          // (the second pass of a pipelined tick is not "the WBC launch" of the timing API)
        // (inspection outputs and cycle stamps are compiled into qr_wbc_kernel_dbg only)
        hipLaunchKernelGGL((d_dbg || d_qp || c->d_dbg_cycles_wbc) ? qr_wbc_kernel_dbg : qr_wbc_kernel, dim3(grid_wgs > 0 ? grid_wgs : 8 * ((n + 7) / 8)), dim3(128), 0, ws, n, c->d_wbc, d_type, d_state,
                           d_cmd ? d_cmd : d_state, d_prev, d_tau, d_qdes, d_status, d_dbg, merge, status_or, (long long *)c->d_dbg_cycles_wbc, d_fr,
                           ready_mask(c->wbc_ready), epilogue, d_qp, pipe);
    }
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

int qrgpu_mpc_solve_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_mpc_state, const float *d_traj,
                          const float *d_gait, const float *d_q, float *d_force, float *d_tau_mpc, int *d_status)
{
    return launch_mpc(c, n, d_type_id, d_mpc_state, d_traj, d_gait, d_q, d_force, d_tau_mpc, d_status, nullptr, nullptr, nullptr, c ? c->epilogue : 0);
}

int qrgpu_mpc_assemble_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_mpc_state, const float *d_traj,
                             const float *d_gait, float *d_H, float *d_g)
{
    if (!c || !d_H || !d_g) return QRGPU_ERR_BAD_ARG;
    // the kernel needs somewhere to put the forces; use the head of d_g's robot 0 row?  No: own scratch.
    float *scratch = nullptr;
    HIPCHK(c, hipMalloc(&scratch, sizeof(float) * 12 * (size_t)n));
    int rc = launch_mpc(c, n, d_type_id, d_mpc_state, d_traj, d_gait, nullptr, scratch, nullptr, nullptr, d_H, d_g, nullptr);
    hipStreamSynchronize(c->stream);
    hipFree(scratch);
    return rc;
}

int qrgpu_wbc_run_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_fb_state, const float *d_wbc_cmd,
                        float *d_prev_ori, float *d_tau, float *d_qdes, int *d_status)
{
    return launch_wbc(c, n, d_type_id, d_fb_state, d_wbc_cmd, d_prev_ori, d_tau, d_qdes, d_status, nullptr, 0, 0);
}

int qrgpu_fb_debug_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_fb_state, float *d_out)
{
    if (!d_out) return QRGPU_ERR_BAD_ARG;
    return launch_wbc(c, n, d_type_id, d_fb_state, nullptr, nullptr, nullptr, nullptr, nullptr, d_out, 0, 0);
}

int qrgpu_wbc_inspect_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_fb_state, const float *d_wbc_cmd,
                            float *d_prev_ori, float *d_tau, float *d_qp, int *d_status)
{
    if (!d_qp) return QRGPU_ERR_BAD_ARG;
    return launch_wbc(c, n, d_type_id, d_fb_state, d_wbc_cmd, d_prev_ori, d_tau, nullptr, d_status, nullptr, 0, 0, nullptr, 0, d_qp);
}

void qrgpu_estimator_desc_default(qrgpu_estimator_desc *d)
{
    if (!d) return;
    memset(d, 0, sizeof(*d));
    d->hip_l = 0.08505f; d->upper_l = 0.2f; d->lower_l = 0.2f;
    const float ho[12] = {0.1805f, -0.047f, 0.f, 0.1805f, 0.047f, 0.f, -0.1805f, -0.047f, 0.f, -0.1805f, 0.047f, 0.f};
    memcpy(d->hip_offset, ho, sizeof(ho));
    d->time_step = 0.002f; d->accelerometer_variance = 0.1f; d->sensor_variance = 0.1f; d->window = 120; d->body_height = 0.28f;
}

int qrgpu_estimator_state_doubles(int window) { return window > 0 ? 96 + 3 * window : 0; }

int qrgpu_estimator_update_batch(qrgpu_ctx *c, int n, const qrgpu_estimator_desc *desc, const float *d_est_in, const unsigned *d_tick,
                                 double *d_est_state, float *d_est_out)
{
    if (!c || n <= 0 || n > c->max_batch || !desc || !d_est_in || !d_tick || !d_est_state || !d_est_out) return QRGPU_ERR_BAD_ARG;
    if (desc->window <= 0 || desc->window > 4096) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    EstimatorDesc D;
    D.hip_l = desc->hip_l; D.upper_l = desc->upper_l; D.lower_l = desc->lower_l;
    memcpy(D.hip_offset, desc->hip_offset, sizeof(D.hip_offset));
    D.time_step = desc->time_step; D.accelerometer_variance = desc->accelerometer_variance; D.sensor_variance = desc->sensor_variance;
    D.window = desc->window; D.body_height = desc->body_height;
    hipLaunchKernelGGL(qr_estimator_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, D, d_est_in, d_tick, d_est_state, d_est_out);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

void qrgpu_gait_desc_default(qrgpu_gait_desc *d)
{
    if (!d) return;
    memset(d, 0, sizeof(*d));
    for (int l = 0; l < 4; ++l) { d->stance_duration[l] = 0.5f; d->duty_factor[l] = 0.6f; d->initial_leg_state[l] = 1; }
    d->initial_leg_phase[0] = 0.5f; d->initial_leg_phase[3] = 0.5f;
    d->contact_detection_phase_threshold = 0.5f; d->wait_time = 1.0f; d->advanced_trot = 1;
}

int qrgpu_gait_update_batch(qrgpu_ctx *c, int n, const qrgpu_gait_desc *desc, float current_time, int robot_stop, int reset, const float *d_contact,
                            float *d_gait_state, float *d_gait_out, float *d_fe_in)
{
    if (!c || n <= 0 || n > c->max_batch || !desc || !d_contact || !d_gait_state) return QRGPU_ERR_BAD_ARG;
    for (int l = 0; l < 4; ++l) if (!(desc->duty_factor[l] > 0.001f) || !(desc->stance_duration[l] > 0.f)) return QRGPU_ERR_BAD_ARG;   // USERDEFINED_SWING legs are not built
    HIPCHK(c, hipSetDevice(c->device));
    GaitDesc D;
    memcpy(D.stance_duration, desc->stance_duration, 16); memcpy(D.duty_factor, desc->duty_factor, 16); memcpy(D.initial_leg_phase, desc->initial_leg_phase, 16);
    memcpy(D.initial_leg_state, desc->initial_leg_state, 16);
    D.contact_detection_phase_threshold = desc->contact_detection_phase_threshold; D.wait_time = desc->wait_time; D.advanced_trot = desc->advanced_trot;
    hipLaunchKernelGGL(qr_gait_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, D, current_time, robot_stop, reset, d_contact, d_gait_state, d_gait_out, d_fe_in);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

void qrgpu_walk_gait_desc_default(qrgpu_walk_gait_desc *d)
{   // config/a1_sim/openloop_gait_generator.yaml, gait "walk"
    if (!d) return;
    memset(d, 0, sizeof(*d));
    for (int l = 0; l < 4; ++l) { d->stance_duration[l] = 7.5f; d->duty_factor[l] = 0.75f; d->initial_leg_state[l] = 1; }
    d->initial_leg_phase[0] = 0.5f; d->initial_leg_phase[1] = 0.f; d->initial_leg_phase[2] = 0.75f; d->initial_leg_phase[3] = 0.25f;
    d->contact_detection_phase_threshold = 0.1f;
    d->n_states = 4;
    d->state_switch[0] = 7; d->state_switch[1] = 6; d->state_switch[2] = 8; d->state_switch[3] = 5;
    d->state_ratio[0] = 0.2f; d->state_ratio[1] = 0.3f; d->state_ratio[2] = 0.3f; d->state_ratio[3] = 0.2f;
}

int qrgpu_walk_gait_update_batch(qrgpu_ctx *c, int n, const qrgpu_walk_gait_desc *desc, float current_time, int robot_stop, int reset,
                                 const float *d_contact, float *d_walk_state, float *d_walk_out, float *d_ratio, float *d_vmc_in)
{
    if (!c || n <= 0 || n > c->max_batch || !desc || !d_contact || !d_walk_state || reset < 0 || reset > 2) return QRGPU_ERR_BAD_ARG;
    if (desc->n_states < 1 || desc->n_states > 4) return QRGPU_ERR_BAD_ARG;
    for (int l = 0; l < 4; ++l) if (!(desc->duty_factor[l] > 0.001f) || !(desc->duty_factor[l] < 1.f) || !(desc->stance_duration[l] > 0.f)) return QRGPU_ERR_BAD_ARG;
    // the constructor's bookkeeping (qr_walk_gait_generator.cpp:87-157): sub-states below a ratio of 0.01 are dropped, the stance-like ones
    // in front of true_swing add up to its start, running sums in float
    WalkDesc D;
    memset(&D, 0, sizeof(D));
    float stand = 0.f;
    for (int k = 0; k < desc->n_states; ++k) {
        if (desc->state_ratio[k] < 0.01) continue;
        const int st = desc->state_switch[k];
        if (st != 5 && st != 6 && st != 7 && st != 8) return QRGPU_ERR_BAD_ARG;
        if (st == 8) D.true_swing_start_in_swing = stand; else stand += desc->state_ratio[k];
        D.que[D.nq] = st; D.ratio[D.nq] = desc->state_ratio[k]; ++D.nq;
    }
    if (D.nq < 1) return QRGPU_ERR_BAD_ARG;
    D.accum[0] = 0.f;
    for (int k = 0; k < D.nq; ++k) D.accum[k + 1] = D.accum[k] + D.ratio[k];
    if (!(fabsf(D.accum[D.nq] - 1.0f) < 1e-4f)) return QRGPU_ERR_BAD_ARG;       // "not vaild ratio definition" (:124)
    for (int l = 0; l < 4; ++l) {
        D.duty_factor[l] = desc->duty_factor[l]; D.initial_leg_phase[l] = desc->initial_leg_phase[l]; D.initial_leg_state[l] = desc->initial_leg_state[l];
        D.full[l] = desc->stance_duration[l] / desc->duty_factor[l];
        D.state_index0[l] = 0;
        if (desc->initial_leg_state[l] == 0) {
            const float ph = (desc->initial_leg_phase[l] - desc->duty_factor[l]) / desc->duty_factor[l];
            int k = 0;
            while (k < D.nq && ph > D.accum[k]) k++;
            D.state_index0[l] = k - 1 > 0 ? k - 1 : 0;
        }
    }
    D.contact_detection_phase_threshold = desc->contact_detection_phase_threshold;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(qr_walk_gait_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, D, current_time, robot_stop, reset, d_contact, d_walk_state, d_walk_out,
                       d_ratio, d_vmc_in);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

int qrgpu_ground_update_batch(qrgpu_ctx *c, int n, int reset, const float *d_ground_in, double *d_ground_state, float *d_ground_out, float *d_est_in)
{
    if (!c || n <= 0 || n > c->max_batch || !d_ground_in || !d_ground_state) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(qr_ground_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, reset, d_ground_in, d_ground_state, d_ground_out, d_est_in);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

void qrgpu_foothold_desc_default(qrgpu_foothold_desc *d)
{
    if (!d) return;
    memset(d, 0, sizeof(*d));
    const float ho[12] = {0.1805f, -0.047f, 0.f, 0.1805f, 0.047f, 0.f, -0.1805f, -0.047f, 0.f, -0.1805f, 0.047f, 0.f};
    const float hp[12] = {0.185f, -0.135f, 0.f, 0.185f, 0.135f, 0.f, -0.185f, -0.135f, 0.f, -0.185f, 0.135f, 0.f};     // config/a1_sim/a1_sim.yaml:40-43
    memcpy(d->hip_offset, ho, sizeof(ho)); memcpy(d->default_hip_position, hp, sizeof(hp));
    d->hip_l = 0.08505f; d->swing_kp[0] = d->swing_kp[1] = d->swing_kp[2] = 0.16f; d->foot_clearance = 0.01f;
}

int qrgpu_footholds_batch(qrgpu_ctx *c, int n, const qrgpu_foothold_desc *desc, const float *d_fh_in, const float *d_gait_state,
                          const float *d_gait_out, float *d_swing_in)
{
    if (!c || n <= 0 || n > c->max_batch || !desc || !d_fh_in || !d_swing_in) return QRGPU_ERR_BAD_ARG;
    if ((d_gait_state == nullptr) != (d_gait_out == nullptr)) return QRGPU_ERR_BAD_ARG;      // both or neither
    HIPCHK(c, hipSetDevice(c->device));
    FootholdDesc D;
    memcpy(D.hip_offset, desc->hip_offset, sizeof(D.hip_offset)); memcpy(D.default_hip_position, desc->default_hip_position, sizeof(D.default_hip_position));
    D.hip_l = desc->hip_l; memcpy(D.swing_kp, desc->swing_kp, sizeof(D.swing_kp)); D.foot_clearance = desc->foot_clearance;
    hipLaunchKernelGGL(qr_foothold_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, D, d_fh_in, d_gait_state, d_gait_out, d_swing_in);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

int qrgpu_swing_velocity_batch(qrgpu_ctx *c, int n, const qrgpu_estimator_desc *desc, const qrgpu_swing_velocity_desc *vdesc, const float *d_swing_vel_in,
                               float *d_out)
{
    if (!c || n <= 0 || n > c->max_batch || !desc || !vdesc || !d_swing_vel_in || !d_out) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    EstimatorDesc D;
    memset(&D, 0, sizeof(D));
    D.hip_l = desc->hip_l; D.upper_l = desc->upper_l; D.lower_l = desc->lower_l;
    memcpy(D.hip_offset, desc->hip_offset, sizeof(D.hip_offset));
    SwingVelDesc V;
    memcpy(V.hip_pos_com, vdesc->hip_position_com, sizeof(V.hip_pos_com)); memcpy(V.stance_duration, vdesc->stance_duration, sizeof(V.stance_duration));
    memcpy(V.swing_kp, vdesc->swing_kp, sizeof(V.swing_kp)); V.desired_height = vdesc->desired_height;
    hipLaunchKernelGGL(qr_swing_velocity_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, D, V, d_swing_vel_in, d_out);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

int qrgpu_swing_targets_batch(qrgpu_ctx *c, int n, const qrgpu_estimator_desc *desc, const float *d_swing_in, float *d_wbc_cmd, float *d_foot_target_world,
                              float *d_qdes)
{
    if (!c || n <= 0 || n > c->max_batch || !desc || !d_swing_in || (!d_wbc_cmd && !d_foot_target_world && !d_qdes)) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    EstimatorDesc D;
    memset(&D, 0, sizeof(D));
    D.hip_l = desc->hip_l; D.upper_l = desc->upper_l; D.lower_l = desc->lower_l;
    memcpy(D.hip_offset, desc->hip_offset, sizeof(D.hip_offset));
    hipLaunchKernelGGL(qr_swing_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, D, d_swing_in, d_wbc_cmd, d_foot_target_world, d_qdes);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

int qrgpu_pack_state_batch(qrgpu_ctx *c, int n, const float com_offset[3], const float *d_est_in, const float *d_est_out, const float *d_rpy,
                           float *d_mpc_state, float *d_fb_state)
{
    if (!c || n <= 0 || n > c->max_batch || !com_offset || !d_est_in || !d_est_out || (!d_mpc_state && !d_fb_state)) return QRGPU_ERR_BAD_ARG;
    if (d_mpc_state && !d_rpy) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(qr_pack_state_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, com_offset[0], com_offset[1], com_offset[2], d_est_in, d_est_out,
                       d_rpy, d_mpc_state, d_fb_state);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

void qrgpu_vmc_desc_default(qrgpu_vmc_desc *d)
{
    if (!d) return;
    memset(d, 0, sizeof(*d));
    d->mass = 13.f;
    d->inertia[0] = 0.24f; d->inertia[4] = 0.80f; d->inertia[8] = 1.0f;
    const float w[6] = {1.f, 1.f, 1.f, 10.f, 10.f, 1.f};
    memcpy(d->acc_weight, w, sizeof(w));
    d->reg_weight = 1e-4f; d->friction = 0.5f; d->fmin_ratio = 0.01f; d->fmax_ratio = 10.f;
    d->hip_l = 0.08505f; d->upper_l = 0.2f; d->lower_l = 0.2f;
}

int qrgpu_vmc_setup(qrgpu_ctx *c, int type_id, const qrgpu_vmc_desc *d)
{
    if (!c || !d || type_id < 0 || type_id >= QR_MAX_TYPES) return QRGPU_ERR_BAD_ARG;
    if (!(d->mass > 0.f)) return QRGPU_ERR_BAD_ARG;
    VmcType &t = c->vmc.type[type_id];
    t.mass = d->mass;
    memcpy(t.inertia, d->inertia, sizeof(t.inertia));
    memcpy(t.acc_weight, d->acc_weight, sizeof(t.acc_weight));
    t.reg_weight = d->reg_weight; t.friction = d->friction; t.fmin_ratio = d->fmin_ratio; t.fmax_ratio = d->fmax_ratio;
    t.hip_l = d->hip_l; t.upper_l = d->upper_l; t.lower_l = d->lower_l;
    c->vmc_ready[type_id] = true;
    return QRGPU_OK;
}

static int launch_vmc(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_vmc_in, const float *d_ratio, const float *d_q, float *d_force,
                      float *d_tau, int *d_status)
{
    if (!c || n <= 0 || n > c->max_batch || !d_vmc_in || !d_force) return QRGPU_ERR_BAD_ARG;
    if (d_tau && !d_q) return QRGPU_ERR_BAD_ARG;
    if (!c->vmc_ready[0]) return QRGPU_ERR_NOT_SETUP;
    HIPCHK(c, hipSetDevice(c->device));
    VmcLaunch P = c->vmc;
    P.n = n; P.ratio = d_ratio;
    hipLaunchKernelGGL(qr_vmc_kernel, dim3(8 * ((n + 7) / 8)), dim3(64), 0, c->stream, P, d_type_id, d_vmc_in, d_q, d_force, d_tau, d_status);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

int qrgpu_vmc_force_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_vmc_in, const float *d_q, float *d_force, float *d_tau,
                          int *d_status)
{
    return launch_vmc(c, n, d_type_id, d_vmc_in, nullptr, d_q, d_force, d_tau, d_status);
}

int qrgpu_vmc_force_world_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_vmc_in, const float *d_ratio, const float *d_q,
                                float *d_force, float *d_tau, int *d_status)
{
    if (!d_ratio) return QRGPU_ERR_BAD_ARG;
    return launch_vmc(c, n, d_type_id, d_vmc_in, d_ratio, d_q, d_force, d_tau, d_status);
}

int qrgpu_mpc_frontend_batch(qrgpu_ctx *c, int n, int num_horizon_l, float dt_ctrl, float dt_mpc, const float *d_fe_in, float *d_fe_state,
                             float *d_traj, float *d_gait, float *d_wbc_cmd, int *d_mpc_updated)
{
    if (!c || n <= 0 || n > c->max_batch || !d_fe_in || !d_fe_state || !d_traj || !d_gait) return QRGPU_ERR_BAD_ARG;
    if (num_horizon_l <= 0 || !(dt_ctrl > 0.f) || !(dt_mpc > 0.f)) return QRGPU_ERR_BAD_ARG;
    if (!c->mpc_ready[0]) return QRGPU_ERR_NOT_SETUP;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(qr_frontend_kernel, dim3((n + 63) / 64), dim3(64, c->mpc.horizon), 0, c->stream, n, c->mpc.horizon, num_horizon_l, dt_ctrl, dt_mpc,
                       d_fe_in, d_fe_state, d_traj, d_gait, d_wbc_cmd, d_mpc_updated);
    HIPCHK(c, hipGetLastError());
    return QRGPU_OK;
}

int qrgpu_tick_batch(qrgpu_ctx *c, int n, const int *d_type_id, const float *d_mpc_state, const float *d_traj,
                     const float *d_gait, const float *d_fb_state, const float *d_wbc_cmd, float *d_prev_ori,
                     float *d_force, float *d_tau, float *d_qdes, int *d_status)
{
    if (!c || !d_fb_state || !d_wbc_cmd || !d_tau || !d_prev_ori) return QRGPU_ERR_BAD_ARG;
    if (n <= 0 || n > c->max_batch) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    static const int pipe_env = [] { const char *e = getenv("QRGPU_TICK_PIPELINE"); return e ? atoi(e) : 1; }();
    c->last_tick_piped = false;
    bool piped = c->pipeline && pipe_env != 0 && n >= 64 && !c->d_dbg_cycles && !c->d_dbg_cycles_wbc;
    if (piped) {         // (not while the stream is being captured into a graph: the WBC launch lives on a stream of the context's own)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess) { cap = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
        if (cap != hipStreamCaptureStatusNone) piped = false;
    }
    // the join: QRGPU_PIPE_JOIN=1 (default) a one-thread launch on the context's stream that polls the count of WBC waves whose written-through
    // outputs are in memory; 0: an event of the WBC stream (10-13 us between the last WBC workgroup and the next launch on the context's stream)
    static const int pipe_join = [] { const char *e = lab_env("QRGPU_PIPE_JOIN"); return e ? atoi(e) : 1; }();
    // Overlapped tick (qrgpu_set_tick_overlap; h <= 11): the tick's launches go on lane 1 or 2 -- stream sets of the context's own, alternating --
    // and the context's stream carries only the join.  When the caller's previous call was an overlapped tick of the same batch that wrote OTHER
    // output arrays, this tick is CHAINED to it: its main pass is released as soon as every workgroup of that tick's MPC launches has started
    // and fills the slots that tick's drain leaves empty; every robot waits for its own previous solve / WBC pass (MpcLaunch::solved,
    // WbcPipe::wbc_done).  Otherwise the lane waits for everything queued on the context's stream so far (an event): no overlap, same results.
    const bool was_chain = c->ov_chain;
    const bool small_h = 4 * c->mpc.horizon <= 44;
    bool ovl = piped && c->overlap && pipe_join && !c->flops_on && (small_h ? (c->lane[1].d_order && c->lane[2].d_order) : (c->lane[3].d_order && c->lane[4].d_order));
    // h > 11: only the two-workgroups-per-CU form of the main pass (3.5 robots per CU and more, list launches and cost words on) overlaps, on the
    // CU-masked lanes; a shard in which most robots stand (the planned list beyond 45 % of the batch) goes back to the plain tick for 31 calls
    if (ovl && !small_h) {
        static const int h16_two = [] { const char *e = getenv("QRGPU_H16_TWO"); return e ? atoi(e) : 1; }();
        static const int hold16 = [] { const char *e = getenv("QRGPU_H16_TWO_HOLD"); return e ? atoi(e) : 31; }();
        ovl = h16_two != 0 && n >= (h16_two >= 2 ? 64 : 7 * c->num_cu / 2) && c->rescue && c->planned && c->lpt;
        if (ovl) {
            const Lane &NL = c->lane[3 + c->ov_next];
            if (c->ov_hold > 0) { --c->ov_hold; ovl = false; }
            else if (hold16 > 0 && NL.plan_n == n && 20 * (long long)NL.h_pre_count[NL.rescue_parity] > 9 * (long long)n) { c->ov_hold = hold16; ovl = false; }
        }
    }
    if (ovl && small_h) {
        // A population with a PLAN -- robots that want a whole CU on a list launch beside the main pass -- is not for overlapped ticks: on a machine
        // that is never empty a whole-CU workgroup waits until both halves of some CU happen to be free at once, and the half-CU list kernel that
        // needs no such luck (S^-1 in the global scratch) takes 300 us and more for such a robot, which the pipeline then waits for: 3.0-3.3 against
        // 4.2 M ticks/s on the bench's populations with an all-stance robot at a degenerate vertex.  So when the lane that is next finds a plan (its
        // last trailing launch listed somebody) the context goes back to the plain pipelined tick for 31 calls, then looks again.
        static const int hold_calls = [] { const char *e = getenv("QRGPU_OV_PLAN_HOLD"); return e ? atoi(e) : 31; }();
        const Lane &NL = c->lane[1 + c->ov_next];
        if (c->ov_hold > 0) { --c->ov_hold; ovl = false; }
        else if (hold_calls > 0 && c->planned && c->rescue && c->lpt && NL.plan_n == n && NL.h_pre_count[NL.rescue_parity] > 0) { c->ov_hold = hold_calls; ovl = false; }
    }
    const int lane_id = ovl ? (small_h ? 1 : 3) + c->ov_next : 0;
    Lane &LN = c->lane[lane_id];
    // wbcData.Fr_des = f (:408): the WBC kernel takes its Fr_des rows from the force array the MPC kernel has just written.
    // The K14 tail, when switched on, is applied by the WBC kernel after the stance / swing merge (the MPC launch leaves d_tau raw).
    float *const force = d_force ? d_force : LN.d_cmd_tick;
    if (!piped) {
        int rc = launch_mpc(c, n, d_type_id, d_mpc_state, d_traj, d_gait, d_fb_state + (size_t)13 * n, force, d_tau, d_status, nullptr, nullptr, nullptr, 0);
        if (rc) return rc;
        return launch_wbc(c, n, d_type_id, d_fb_state, d_wbc_cmd, d_prev_ori, d_tau, d_qdes, d_status, nullptr, 1, d_status ? 1 : 0, force, c->epilogue);
    }
    // Pipelined tick.  Of a robot's WBC only the relaxation QP at its very end reads the MPC's forces, and the MPC launch spends its last third
    // with most of its slots empty (two rounds of robots of very different length: DESIGN.md 5).  So the WBC launch goes on a stream of its own
    // beside the MPC launches: a gate holds it until every workgroup of the main pass has started (it must never take a CU from a solve it is
    // going to wait for), then its workgroups settle wherever a solve has left, run the rigid-body dynamics, the task set and the kinematic
    // projection, and wait -- bounded -- at the QP for their robot's flag (qr_wbc_kernel.hip, qr_mpc_kernel.hip).  Robots the main pass hands
    // to its trailing list launch are skipped there and taken by a second, list-driven WBC pass queued behind that launch.
    const unsigned prev_epoch = c->tick_epoch;
    if (++c->tick_epoch >= 0x3fffffffu) c->tick_epoch = 1;        // (below 2^30: bit 31 of a robot's flag word says "its WBC workgroup gave up in this epoch")
    const unsigned epoch = c->tick_epoch;
    // (the give-up word of this tick's WBC gate: a ring indexed by the epoch -- several ticks may be queued behind a backlog)
    int *const gate_abort = c->d_gate_abort + (epoch & (QR_ABORT_RING - 1));
    // (QRGPU_OV_FAULT=1, the give-up tests: chained ticks wait for an epoch nobody ever writes, so that every per-robot wait runs into its bound)
    static const unsigned ov_fault = [] { const char *e = getenv("QRGPU_OV_FAULT"); return (e && atoi(e) == 1) ? 0x10000000u : 0u; }();      // (a quarter of the epochs' range ahead: "not reached yet")
    OvLaunch ov{epoch, false, (prev_epoch + ov_fault) & 0x3fffffffu, false, nullptr, 0u};
    if (ovl) {
        c->ov_next ^= 1;
        const void *outs[4] = {(const void *)d_force, (const void *)d_tau, (const void *)d_qdes, (const void *)d_status};
        bool distinct = true;
        for (int a = 0; a < 4; ++a) if (outs[a]) for (int b = 0; b < 4; ++b) if (outs[a] == c->ov_out[b]) distinct = false;
        // A tick whose lane has a plan -- robots that want a whole CU, on a list launch beside the main pass -- is not chained, nor is its successor: on
        // a machine that is never empty a whole-CU workgroup waits until both halves of some CU happen to be free at once (the half-CU list kernel
        // that needs no such luck takes 300 us and more for such a robot, and the pipeline then waits for it: 3.0 against 4.2 M ticks/s on the
        // populations that hold an all-stance robot at a degenerate vertex).
        ov.plan_tick = small_h && c->planned && c->rescue && c->lpt && LN.plan_n == n && LN.h_pre_count[LN.rescue_parity] > 0;       // (only with QRGPU_OV_PLAN_HOLD=0)
        ov.chained = was_chain && c->ov_n == n && c->ov_epoch == prev_epoch && c->ov_prev_ori == (const void *)d_prev_ori && distinct && !ov.plan_tick && !c->ov_prev_plan;
        c->ov_prev_plan = ov.plan_tick;
        // What the caller had queued on the context's stream when it made the PREVIOUS tick call -- the join of the tick before that one and whatever
        // consumed its outputs, which are the arrays this tick overwrites when the caller double-buffers -- must be through before this tick writes
        // anything: the event recorded at that call.  (It completed about a tick ago: the wait costs the lane nothing.)  An unchained tick waits for
        // the event recorded now: everything queued on the context's stream so far.
        const int ev_now = (c->ev_call_last + 1) & 1;
        const int ev_prev = c->ev_call_last;
        HIPCHK(c, hipEventRecord(c->ev_call[ev_now], c->stream));
        if (ov.chained && ev_prev >= 0) {
            HIPCHK(c, hipStreamWaitEvent(LN.stream, c->ev_call[ev_prev], 0));
            // ... and not before every workgroup of the previous tick's main pass and planned launch has started (bounded: 50 ms; harmless if it gives up)
            Lane &PL = c->lane[c->ov_lane_last];
            // (h > 11: the planned launches live on reserved CUs, the main pass cannot keep them from starting)
            hipLaunchKernelGGL(qr_gate2_kernel, dim3(1), dim3(64), 0, LN.stream, c->d_main_started, (int)c->ov_main_total, small_h ? PL.d_started : (int *)nullptr, (int)PL.started_total, (long long)5000000,
                               c->d_timeline ? c->d_timeline + 512 + (epoch & 63u) * 2 : (long long *)nullptr);      // (diagnostic: qrgpu_debug_gate2)
            HIPCHK(c, hipGetLastError());
        } else {
            ov.chained = false;
            HIPCHK(c, hipStreamWaitEvent(LN.stream, c->ev_call[ev_now], 0));
        }
        c->ev_call_last = ev_now;
        if (ov.chained) { ov.prev_started = c->lane[c->ov_lane_last].d_started; ov.prev_started_total = c->lane[c->ov_lane_last].started_total - (c->lane[c->ov_lane_last].last_linger < 8 ? c->lane[c->ov_lane_last].last_linger : 8); }      // (all but the ones that wait for a CU held by the tick before's rescuers)
        // (the all-gathers the caller fenced since the last tick -- qrgpu_allgather_fence -- still read output arrays this tick overwrites)
        for (int sl = 0; sl < 2; ++sl)
            if (((c->ov_fence_slots >> sl) & 1) && c->ev_gather[sl]) HIPCHK(c, hipStreamWaitEvent(LN.stream, c->ev_gather[sl], 0));
        c->ov_fence_slots = 0;
        for (int a = 0; a < 4; ++a) c->ov_out[a] = outs[a];
    }
    // (No fork event from the context stream: the gate below opens only once this tick's main pass -- queued on the context stream behind
    //  everything the caller put there -- is running, and the WBC launch of the previous tick is ahead of this one on the same stream.
    //  QRGPU_PIPE_FORK=1 puts the event back: 10-15 us of cross-stream hand-over per tick.)
    static const int pipe_fork = [] { const char *e = lab_env("QRGPU_PIPE_FORK"); return e ? atoi(e) : 0; }();
    if (pipe_fork && !ovl) {
        HIPCHK(c, hipEventRecord(c->ev_wbc_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->wbc_stream, c->ev_wbc_fork, 0));
    }
    // (QRGPU_PIPE_EARLY=K opens the gate K workgroups early: an experiment, see LAB_NOTES.md)
    static const int pipe_early = [] { const char *e = lab_env("QRGPU_PIPE_EARLY"); return e ? atoi(e) : 0; }();
    // (the half of d_wbc_order this tick's WBC launch reads: taken before launch_mpc, whose trailing launch writes the other half and flips the parity)
    const int *const wbc_order_in = (c->wbc_order_n == n && !ovl) ? c->d_wbc_order + (size_t)c->wbc_order_parity * (size_t)c->max_batch : nullptr;
    int rc = launch_mpc(c, n, d_type_id, d_mpc_state, d_traj, d_gait, d_fb_state + (size_t)13 * n, force, d_tau, d_status, nullptr, nullptr, nullptr, 0, true, lane_id,
                        ovl ? &ov : nullptr);
    if (rc) return rc;
    const int expect = (int)(c->main_started_total - (unsigned)pipe_early);          // (launch_mpc has added this tick's main-pass units to main_started_total)
    // (bounded at 50 ms; QRGPU_PIPE_GATE_MS for the tests.  A gate that gives up -- the caller had that much work of its own queued in front of
    //  this tick -- turns the tick into the serial one: WbcPipe::gate_abort)
    static const long long gate_ticks = [] { const char *e = getenv("QRGPU_PIPE_GATE_MS"); return 100000LL * (e ? atoll(e) : 50LL); }();
    // (h > 11, laboratory: QRGPU_OV16_WBC_MASK=1 keeps the WBC launches off the reserved CUs -- a masked stream has no priority, and without it tick t's
    //  WBC workgroups lose the freed slots to tick t + 1's solves: LAB_NOTES A.2 item 4)
    static const int wbc16_masked = [] { const char *e = lab_env("QRGPU_OV16_WBC_MASK"); return e ? atoi(e) : 0; }();
    const hipStream_t wbc_stream = ovl ? ((small_h || !wbc16_masked) ? c->wbc_stream_hi : c->wbc_stream_16) : c->wbc_stream;
    // (an overlapped tick has no second pass to fall back on: its gate is patient -- 2 s -- and one that gives up just lets the launch go: every wait
    //  of a WBC workgroup for its robot's forces is bounded and flagged.  What the serial fall-back protects against -- inputs that the caller's stream
    //  has not produced yet -- cannot happen: a chained tick's inputs are ready by contract, an unchained one makes this stream wait for the event too)
    if (ovl && !ov.chained) HIPCHK(c, hipStreamWaitEvent(wbc_stream, c->ev_call[c->ev_call_last], 0));
    static const long long flag_ticks = [] { const char *e = getenv("QRGPU_PIPE_WAIT_US"); return e ? 100LL * atoll(e) : 400000LL; }();
    unsigned *const wbc_done = ovl ? c->d_wbc_done : nullptr;
    const unsigned wait_epoch = (ovl && ov.chained) ? ((prev_epoch + ov_fault) & 0x3fffffffu) : 0u;
    // Large batches, LABORATORY (QRGPU_LAB=1 QRGPU_WBC_CHUNKS=1; LAB_NOTES A.7: bit-identical, 5 % slower at 8192 robots): the WBC launch in launches of
    // 1024 workgroups, each behind its own gate (WbcPipe::slot_base).  The main pass is not persistent at h <= 11, so "started" counts workgroups
    // in dispatch order.
    static const int wbc_chunks_on = [] { const char *e = lab_env("QRGPU_WBC_CHUNKS"); return e ? atoi(e) : 0; }();
    const int total_wgs = 8 * ((n + 7) / 8);
    const bool chunked = wbc_chunks_on && !ovl && n >= 4096 && small_h && !c->last_main_persist && !wbc_order_in && !pipe_early;
    const int chunk_wgs = chunked ? 1024 : total_wgs;
    const int main_before = (int)(c->main_started_total - (unsigned)total_wgs);         // (what the counter stood at before this tick's main pass)
    for (int base = 0; base < total_wgs; base += chunk_wgs) {
        const int wgs = (total_wgs - base < chunk_wgs) ? total_wgs - base : chunk_wgs;
        const int expect_k = chunked ? main_before + base + wgs : expect;
        hipLaunchKernelGGL(qr_gate_kernel, dim3(1), dim3(64), 0, wbc_stream, c->d_main_started, expect_k, ovl ? 200000000LL : gate_ticks, ovl ? (int *)nullptr : gate_abort, (int)epoch,
                           (int *)nullptr);
        HIPCHK(c, hipGetLastError());
        WbcPipe wp{LN.d_done_flag, epoch, nullptr, nullptr, ovl ? (int *)nullptr : gate_abort, 0, pipe_join ? c->d_wbc_finished : nullptr, c->d_tlr, c->d_timeline, chunked ? LN.order_used : wbc_order_in, wbc_done, wait_epoch,
                   ov_wait_ticks(), ovl ? 1 : 0, flag_ticks, chunked ? base : 0};
        rc = launch_wbc(c, n, d_type_id, d_fb_state, d_wbc_cmd, d_prev_ori, d_tau, d_qdes, d_status, nullptr, 1, d_status ? 1 : 0, force, c->epilogue, nullptr,
                        wbc_stream, wp, chunked ? wgs : 0, base == 0);
        if (rc) return rc;
    }
    if (!pipe_join) HIPCHK(c, hipEventRecord(c->ev_wbc_join, c->wbc_stream));
    if (!ovl) {   // the second pass: the robots of the trailing launch's list (there is one at h <= 11) -- or every robot, should the gate have given up
        const bool have_list = LN.last_rescue_active;
        WbcPipe lp{nullptr, epoch, have_list ? LN.d_rescue + 2 : nullptr, have_list ? LN.d_rescue + LN.last_rescue_parity : nullptr, gate_abort, 1, nullptr,
                   nullptr, c->d_timeline, nullptr, nullptr, 0u, 0, 0, flag_ticks};
        rc = launch_wbc(c, n, d_type_id, d_fb_state, d_wbc_cmd, d_prev_ori, d_tau, d_qdes, d_status, nullptr, 1, d_status ? 1 : 0, force, c->epilogue, nullptr, nullptr, lp);
        if (rc) return rc;
    }
    if (pipe_join) {
        c->wbc_finished_total += 2u * (unsigned)n;
        // (... and for the all-gathers queued before this tick, so that the fence in front of the next tick need not queue a launch: qr_join_kernel)
        int *g0 = c->d_gather_done, *g1 = c->d_gather_done ? c->d_gather_done + 1 : nullptr;
        hipLaunchKernelGGL(qr_join_kernel, dim3(1), dim3(64), 0, c->stream, c->d_wbc_finished, (int)c->wbc_finished_total, (long long)2000000, c->lane[0].d_pre_hint + 2,
                           g0, (int)c->gather_total[0], g1, (int)c->gather_total[1], c->d_tick_done, (int *)nullptr, 0,
                           c->d_join_dbg ? c->d_join_dbg + 8 * (c->tick_done_total & 15u) : (long long *)nullptr);
        HIPCHK(c, hipGetLastError());
        c->gather_joined[0] = c->gather_total[0]; c->gather_joined[1] = c->gather_total[1];
        ++c->tick_done_total; c->last_tick_piped = true;
    } else HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_wbc_join, 0));
    if (ovl) {
        ++c->ov_stats[ov.chained ? 0 : 1];
        c->ov_chain = true; c->ov_n = n; c->ov_epoch = epoch; c->ov_main_total = c->main_started_total; c->ov_prev_ori = (const void *)d_prev_ori; c->ov_lane_last = lane_id;
    }
    return QRGPU_OK;
}

// Overlapped ticks: see include/qrgpu.h.  Switching them on creates lanes 1 and 2 and PROBES that two of the context's streams really run side
// by side in this process (a launch on one lane that waits for a launch queued afterwards on the other): with fewer hardware queues than streams
// (GPU_MAX_HW_QUEUES, default 4, against the context's seven) two streams may share one, a chained tick would sit out its gates' bounds behind its
// predecessor, and the mode is refused -- QRGPU_ERR_NOT_SETUP, qrgpu_last_error says why, ticks stay as they were.
int qrgpu_set_tick_overlap(qrgpu_ctx *c, int on)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    c->ov_chain = false;
    if (!on) { c->overlap = 0; return QRGPU_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->ev_call[0]) for (int k = 0; k < 2; ++k) HIPCHK(c, hipEventCreateWithFlags(&c->ev_call[k], hipEventDisableTiming));
    if (!c->wbc_stream_hi) {
        static const int hi = [] { const char *e = lab_env("QRGPU_OV_WBC_PRIORITY"); return e ? atoi(e) : 1; }();
        if (hi) HIPCHK(c, create_side_stream(&c->wbc_stream_hi));
        else HIPCHK(c, hipStreamCreateWithFlags(&c->wbc_stream_hi, hipStreamNonBlocking));
    }
    // The stream sets are made for the horizon the context is set up with at this call (a context whose horizon changes class afterwards calls this
    // again; until then its ticks are plain pipelined ticks): every stream wants a hardware queue of its own, and a context that made both sets
    // would own eleven streams against GPU_MAX_HW_QUEUES = 8.
    //   h <= 11: lanes 1 and 2 (a stream each; the planned launches of consecutive ticks share lane 0's side stream, in tick order).
    //   h > 11 (QRGPU_OV16=0 keeps such contexts on the plain tick): lanes 3 and 4 on a machine split in space by CU masks -- the main pass two to a
    //     CU on 192 CUs, the big class's whole-CU workgroups (and whatever the main pass hands on) on 64 reserved ones, eight of every XCD
    //     (the per-XCD count has to be a multiple of four, LAB_NOTES A.2 item 2; QRGPU_OV16_SIDE_CUS = 32 for the A/B).
    static const bool ov16_on = [] { const char *e = getenv("QRGPU_OV16"); return !e || atoi(e) != 0; }();
    const bool want16 = 4 * c->mpc.horizon > 44;
    hipStream_t st[6]; int nst = 0, npair = 0;
    if (!want16) {
        for (int l = 1; l <= 2; ++l) {
            if (lane_create(c, c->lane[l], true) != QRGPU_OK) { c->err = "qrgpu_set_tick_overlap: allocation of a lane failed"; return QRGPU_ERR_ALLOC; }
            c->lane[l].side_stream = c->lane[0].side_stream;
        }
        st[0] = c->lane[1].stream; st[1] = c->lane[2].stream; st[2] = c->wbc_stream_hi; st[3] = c->stream; nst = 4; npair = 2;
    } else if (ov16_on) {
        if (!c->ov16_side_cus) {
            static const int side_env = [] { const char *e = lab_env("QRGPU_OV16_SIDE_CUS"); return e ? atoi(e) : 64; }();
            int k = (side_env * c->num_cu / 256) & ~31;
            if (k < 32) k = 32;
            if (k > c->num_cu / 2) k = (c->num_cu / 2) & ~31;
            c->ov16_side_cus = k;
            for (int b = 0; b < c->num_cu && b < 512; ++b) { if (b < c->num_cu - k) c->mask16_main[b >> 5] |= 1u << (b & 31); else c->mask16_side[b >> 5] |= 1u << (b & 31); }
            static const int wbc16_masked = [] { const char *e = lab_env("QRGPU_OV16_WBC_MASK"); return e ? atoi(e) : 0; }();
            if (wbc16_masked) HIPCHK(c, hipExtStreamCreateWithCUMask(&c->wbc_stream_16, (uint32_t)((c->num_cu + 31) / 32), c->mask16_main));
        }
        for (int l = 3; l <= 4; ++l)
            if (lane_create(c, c->lane[l], true, true) != QRGPU_OK) { c->err = "qrgpu_set_tick_overlap: allocation of a CU-masked lane failed"; return QRGPU_ERR_ALLOC; }
        st[0] = c->lane[3].stream; st[1] = c->lane[4].stream; st[2] = c->lane[3].side_stream; st[3] = c->lane[4].side_stream; st[4] = c->wbc_stream_hi; st[5] = c->stream; nst = 6; npair = 4;
    } else { c->overlap = 1; return QRGPU_OK; }          // (h > 11 with QRGPU_OV16=0: the mode is on, the ticks stay plain)
    // probe, both ways round: the lanes' streams against each other, and each of them against the WBC stream and the context's
    int *d_probe = nullptr;
    HIPCHK(c, hipMalloc(&d_probe, 16 * sizeof(int)));
    HIPCHK(c, hipMemset(d_probe, 0, 16 * sizeof(int)));
    HIPCHK(c, hipDeviceSynchronize());
    int k = 0, token = 0;
    for (int a = 0; a < nst; ++a)
        for (int b = 0; b < nst; ++b) {
            if (a == b || (a >= npair && b >= npair)) continue;
            // (20 ms, and a pair that fails is asked once more: the waiting launch runs from the moment it is queued, the other one is queued by this
            //  thread right behind it -- on a host busy with something else "right behind" has been seen to take longer than the 2 ms this bound was)
            int res = 0;
            for (int attempt = 0; attempt < 2 && res != 1; ++attempt) {
                ++token;                                   // (the eight flag words go round: every probe has a value of its own)
                hipLaunchKernelGGL(qr_probe_wait_kernel, dim3(1), dim3(64), 0, st[a], d_probe + k, d_probe + 8, (long long)2000000, token);
                hipLaunchKernelGGL(qr_probe_set_kernel, dim3(1), dim3(64), 0, st[b], d_probe + k, token);
                HIPCHK(c, hipStreamSynchronize(st[a]));
                HIPCHK(c, hipStreamSynchronize(st[b]));
                HIPCHK(c, hipMemcpy(&res, d_probe + 8, sizeof(int), hipMemcpyDeviceToHost));
                k = (k + 1) & 7;
            }
            if (res != 1) {
                hipFree(d_probe);
                static char msg[320];
                snprintf(msg, sizeof(msg), "qrgpu_set_tick_overlap: two of the context's streams share a hardware queue in this process (set GPU_MAX_HW_QUEUES=8 before the first HIP call); "
                         "overlapped ticks stay off [a launch on stream %d of the set waited for one queued behind it on stream %d]", a, b);
                c->err = msg;
                c->overlap = 0;
                return QRGPU_ERR_NOT_SETUP;
            }
        }
    hipFree(d_probe);
    c->overlap = 1;
    return QRGPU_OK;
}
int qrgpu_tick_fence(qrgpu_ctx *c) { if (!c) return QRGPU_ERR_BAD_ARG; c->ov_chain = false; return QRGPU_OK; }
int qrgpu_tick_overlap_stats(const qrgpu_ctx *c, int *chained, int *unchained)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    if (chained) *chained = c->ov_stats[0];
    if (unchained) *unchained = c->ov_stats[1];
    return QRGPU_OK;
}

int qrgpu_set_torque_epilogue(qrgpu_ctx *c, int flags)
{
    if (!c || (flags & ~(QRGPU_EPILOGUE_HIP_COMP | QRGPU_EPILOGUE_CLIP))) return QRGPU_ERR_BAD_ARG;
    c->epilogue = flags;
    return QRGPU_OK;
}

// The single-robot calls stage through the context (qrgpu_ctx.h): with zero copy the host fills the pinned block in place, the one
// launch reads and writes it over PCIe, and the only stream command besides the launch is the wait.
static int stage_in(qrgpu_ctx *c, const float *src, size_t nfloat, int type_id, int **d_type)
{
    *d_type = nullptr;
    if (c->zero_copy) {
        memcpy(c->h_in1, src, nfloat * sizeof(float));
        if (type_id != 0) { c->h_st1[1] = type_id; *d_type = c->d_st1 + 1; }
        return QRGPU_OK;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_in1, src, nfloat * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (type_id != 0) {
        *d_type = c->d_st1 + 1;
        c->type_stage = type_id;         // (a context member: the copy is asynchronous)
        HIPCHK(c, hipMemcpyAsync(*d_type, &c->type_stage, sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    return QRGPU_OK;
}
static int stage_out(qrgpu_ctx *c, float *out, size_t nfloat, int *st)
{
    if (c->zero_copy) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(out, c->h_out1, nfloat * sizeof(float));
        *st = c->h_st1[0];
        return QRGPU_OK;
    }
    HIPCHK(c, hipMemcpyAsync(out, c->d_out1, nfloat * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(st, c->d_st1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QRGPU_OK;
}

int qrgpu_mpc_solve1(qrgpu_ctx *c, int type_id, const float p[3], const float v[3], const float quat[4], const float w[3],
                     const float r[12], const float rpy[3], const float *traj, const float *gait, const float q[12],
                     double f_out[12], float tau_out[12], int *status)
{
    if (!c || !p || !v || !quat || !w || !r || !rpy || !traj || !gait || !f_out) return QRGPU_ERR_BAD_ARG;
    if (type_id < 0 || type_id >= QR_MAX_TYPES || !c->mpc_ready[type_id]) return QRGPU_ERR_NOT_SETUP;
    const int h = c->mpc.horizon;
    float in[28 + 16 * QRGPU_MAX_HORIZON + 12];
    const size_t nin = 28 + 16 * (size_t)h + 12;
    memcpy(&in[0], p, 12); memcpy(&in[3], v, 12); memcpy(&in[6], quat, 16); memcpy(&in[10], w, 12);
    memcpy(&in[13], r, 48); memcpy(&in[25], rpy, 12);
    memcpy(&in[28], traj, sizeof(float) * 12 * h);
    memcpy(&in[28 + 12 * h], gait, sizeof(float) * 4 * h);
    if (q) memcpy(&in[28 + 16 * h], q, 48); else memset(&in[28 + 16 * h], 0, 48);
    HIPCHK(c, hipSetDevice(c->device));
    int *d_type = nullptr;
    int rc = stage_in(c, in, nin, type_id, &d_type);
    if (rc) return rc;
    rc = launch_mpc(c, 1, d_type, c->d_in1, c->d_in1 + 28, c->d_in1 + 28 + 12 * h, c->d_in1 + 28 + 16 * h, c->d_out1,
                    (q && tau_out) ? c->d_out1 + 12 : nullptr, c->d_st1, nullptr, nullptr, nullptr);
    if (rc) return rc;
    float out[24]; int st = 0;
    rc = stage_out(c, out, 24, &st);
    if (rc) return rc;
    for (int i = 0; i < 12; ++i) f_out[i] = out[i];
    if (q && tau_out) for (int i = 0; i < 12; ++i) tau_out[i] = out[12 + i];
    if (status) *status = st;
    return QRGPU_OK;
}

int qrgpu_wbc_run1(qrgpu_ctx *c, int type_id, const float fb_state[37], const float wbc_cmd[67], float prev_ori_vel[3],
                   float tau_out[12], float qdes_out[12], float qddes_out[12], int *status)
{
    if (!c || !fb_state || !wbc_cmd || !prev_ori_vel || !tau_out) return QRGPU_ERR_BAD_ARG;
    if (type_id < 0 || type_id >= QR_MAX_TYPES || !c->wbc_ready[type_id]) return QRGPU_ERR_NOT_SETUP;
    float in[37 + 67 + 3];
    memcpy(in, fb_state, 37 * 4); memcpy(in + 37, wbc_cmd, 67 * 4); memcpy(in + 104, prev_ori_vel, 12);
    HIPCHK(c, hipSetDevice(c->device));
    int *d_type = nullptr;
    int rc = stage_in(c, in, 107, type_id, &d_type);
    if (rc) return rc;
    const bool want_q = qdes_out || qddes_out;
    rc = launch_wbc(c, 1, d_type, c->d_in1, c->d_in1 + 37, c->d_in1 + 104, c->d_out1, want_q ? c->d_out1 + 12 : nullptr,
                    c->d_st1, nullptr, 0, 0);
    if (rc) return rc;
    float out[36]; int st = 0;
    if (!c->zero_copy) HIPCHK(c, hipMemcpyAsync(prev_ori_vel, c->d_in1 + 104, 12, hipMemcpyDeviceToHost, c->stream));
    rc = stage_out(c, out, 36, &st);
    if (rc) return rc;
    if (c->zero_copy) memcpy(prev_ori_vel, c->h_in1 + 104, 12);
    memcpy(tau_out, out, 48);
    if (qdes_out) memcpy(qdes_out, out + 12, 48);
    if (qddes_out) memcpy(qddes_out, out + 24, 48);
    if (status) *status = st;
    return QRGPU_OK;
}

static int vmc_force1(qrgpu_ctx *c, int type_id, const float vmc_in[37], const float ratio[8], const float q[12], float force_out[12], float tau_out[12],
                      int *status)
{
    if (!c || !vmc_in || !force_out) return QRGPU_ERR_BAD_ARG;
    if (tau_out && !q) return QRGPU_ERR_BAD_ARG;
    if (type_id < 0 || type_id >= QR_MAX_TYPES || !c->vmc_ready[type_id]) return QRGPU_ERR_NOT_SETUP;
    float in[37 + 12 + 8];
    memset(in, 0, sizeof(in));
    memcpy(in, vmc_in, 37 * 4);
    if (q) memcpy(in + 37, q, 48);
    if (ratio) memcpy(in + 49, ratio, 32);
    HIPCHK(c, hipSetDevice(c->device));
    int *d_type = nullptr;
    { const int rc_ = stage_in(c, in, sizeof(in) / sizeof(float), type_id, &d_type); if (rc_) return rc_; }
    VmcLaunch P = c->vmc;
    P.n = 1; P.ratio = ratio ? c->d_in1 + 49 : nullptr;
    hipLaunchKernelGGL(qr_vmc_kernel, dim3(8), dim3(64), 0, c->stream, P, d_type, c->d_in1, q ? c->d_in1 + 37 : nullptr, c->d_out1,
                       (q && tau_out) ? c->d_out1 + 12 : nullptr, c->d_st1);
    HIPCHK(c, hipGetLastError());
    float out[24]; int st = 0;
    { const int rc_ = stage_out(c, out, 24, &st); if (rc_) return rc_; }
    memcpy(force_out, out, 48);
    if (q && tau_out) memcpy(tau_out, out + 12, 48);
    if (status) *status = st;
    return QRGPU_OK;
}

int qrgpu_vmc_force1(qrgpu_ctx *c, int type_id, const float vmc_in[37], const float q[12], float force_out[12], float tau_out[12], int *status)
{
    return vmc_force1(c, type_id, vmc_in, nullptr, q, force_out, tau_out, status);
}

int qrgpu_vmc_force_world1(qrgpu_ctx *c, int type_id, const float vmc_in[37], const float ratio[8], const float q[12], float force_out[12],
                           float tau_out[12], int *status)
{
    if (!ratio) return QRGPU_ERR_BAD_ARG;
    return vmc_force1(c, type_id, vmc_in, ratio, q, force_out, tau_out, status);
}

int qrgpu_debug_cycles(qrgpu_ctx *c, long long *host_out /* [n][8] or NULL to disable */, int n)
{   // undocumented diagnostic: phase cycle stamps of the last MPC launch (enable by calling once with NULL first; NULL with n < 0 switches them off again)
    if (!c) return QRGPU_ERR_BAD_ARG;
    if (!host_out && n < 0) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_dbg_cycles) hipFree(c->d_dbg_cycles);
        if (c->d_dbg_cycles_wbc) hipFree(c->d_dbg_cycles_wbc);
        c->d_dbg_cycles = nullptr; c->d_dbg_cycles_wbc = nullptr;
        return QRGPU_OK;
    }
    if (!c->d_dbg_cycles) {
        HIPCHK(c, hipMalloc(&c->d_dbg_cycles, sizeof(long long) * 16 * (size_t)c->max_batch));
        HIPCHK(c, hipMalloc(&c->d_dbg_cycles_wbc, sizeof(long long) * 16 * (size_t)(c->max_batch + 8)));
        return QRGPU_OK;
    }
    if (host_out && n < 0) {   // n < 0: fetch the WBC kernel's stamps instead (indexed by workgroup)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(host_out, c->d_dbg_cycles_wbc, sizeof(long long) * 16 * (size_t)(-n), hipMemcpyDeviceToHost));
        return QRGPU_OK;
    }
    if (host_out) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipMemcpy(host_out, c->d_dbg_cycles, sizeof(long long) * 16 * (size_t)n, hipMemcpyDeviceToHost)); }
    return QRGPU_OK;
}

int qrgpu_debug_lists(qrgpu_ctx *c, int *host_out /* [8]: rescue list lengths (both parities), planned list lengths (both parities), "go" count and the
                                                      plan epoch of a planned launch whose gate gave up, the tick epoch of a WBC gate that gave up, the context's plan epoch */)
{   // undocumented diagnostic: how many robots the last MPC launches handed to the trailing list launch / planned for the next call; which gates gave up
    if (!c || !host_out) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // (the lane the context's last launch ran on; the give-up words are rings indexed by epoch: the latest epoch in each is reported)
    const Lane &L = c->lane[c->ov_chain ? c->ov_lane_last : 0];
    HIPCHK(c, hipMemcpy(host_out, L.d_rescue, 2 * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(host_out + 2, L.d_pre, 2 * sizeof(int), hipMemcpyDeviceToHost));
    int ring[1 + QR_ABORT_RING];
    HIPCHK(c, hipMemcpy(ring, L.d_go, sizeof(ring), hipMemcpyDeviceToHost));
    host_out[4] = ring[0]; host_out[5] = 0;
    for (int i = 1; i <= QR_ABORT_RING; ++i) if (ring[i] > host_out[5]) host_out[5] = ring[i];
    HIPCHK(c, hipMemcpy(ring, c->d_gate_abort, QR_ABORT_RING * sizeof(int), hipMemcpyDeviceToHost));
    host_out[6] = 0;
    for (int i = 0; i < QR_ABORT_RING; ++i) if (ring[i] > host_out[6]) host_out[6] = ring[i];
    host_out[7] = L.plan_epoch;
    return QRGPU_OK;
}

int qrgpu_debug_counters(qrgpu_ctx *c, int *host_out /* [12]: device count / host total of main_started, wbc_finished, tick_done, lane_done of lanes 1 and 2, epoch */)
{   // undocumented diagnostic: the cumulative counters the gates and joins poll, as the device and the host see them
    if (!c || !host_out) return QRGPU_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    memset(host_out, 0, 12 * sizeof(int));
    HIPCHK(c, hipMemcpy(host_out + 0, c->d_main_started, sizeof(int), hipMemcpyDeviceToHost)); host_out[1] = (int)c->main_started_total;
    HIPCHK(c, hipMemcpy(host_out + 2, c->d_wbc_finished, sizeof(int), hipMemcpyDeviceToHost)); host_out[3] = (int)c->wbc_finished_total;
    HIPCHK(c, hipMemcpy(host_out + 4, c->d_tick_done, sizeof(int), hipMemcpyDeviceToHost)); host_out[5] = (int)c->tick_done_total;
    for (int l = 1; l <= 2; ++l)
        if (c->lane[l].d_lane_done) { HIPCHK(c, hipMemcpy(host_out + 4 + 2 * l, c->lane[l].d_lane_done, sizeof(int), hipMemcpyDeviceToHost)); host_out[5 + 2 * l] = (int)c->lane[l].lane_done_total; }
    host_out[10] = (int)c->tick_epoch;
    if (!c->d_join_dbg) { HIPCHK(c, hipMalloc(&c->d_join_dbg, 16 * 8 * sizeof(long long))); HIPCHK(c, hipMemset(c->d_join_dbg, 0, 16 * 8 * sizeof(long long))); }
    else {
        long long h[16 * 8];
        HIPCHK(c, hipMemcpy(h, c->d_join_dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) if (h[8 * i]) fprintf(stderr, "  join[%d]: start %lld dur %.1f us expect %lld lane_expect %lld seen %lld lane_seen %lld gave_up %lld\n", i, h[8 * i], (h[8 * i + 1] - h[8 * i]) / 100.0, h[8 * i + 2], h[8 * i + 3], h[8 * i + 4], h[8 * i + 5], h[8 * i + 6]);
    }
    return QRGPU_OK;
}

int qrgpu_debug_gate2(qrgpu_ctx *c, long long *host_out /* [64][2]: when the gate in front of a chained tick's launches came up / opened, by epoch & 63 */)
{
    if (!c || !c->d_timeline || !host_out) return QRGPU_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    HIPCHK(c, hipMemcpy(host_out, c->d_timeline + 512, sizeof(long long) * 256, hipMemcpyDeviceToHost));      // [64][2] gate up / open, then [64][2] planned launch first start / last end
    return QRGPU_OK;
}

int qrgpu_debug_timeline_solves(qrgpu_ctx *c, long long *host_out /* [2][16][1024]: per epoch & 15 and robot: (publish time << 8 | launch kind), (cross-tick wait << 8 | kind | 8 gave up) */)
{
    if (!c || !c->d_timeline || !host_out) return QRGPU_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    HIPCHK(c, hipMemcpy(host_out, c->d_timeline + 768, sizeof(long long) * 2 * 16 * 1024, hipMemcpyDeviceToHost));
    return QRGPU_OK;
}

int qrgpu_debug_timeline_plans(qrgpu_ctx *c, long long *host_out /* [64][64] list length each planned workgroup read, then [64] the length each tick's planning left */)
{
    if (!c || !c->d_timeline || !host_out) return QRGPU_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    HIPCHK(c, hipMemcpy(host_out, c->d_timeline + 768 + 32768, sizeof(long long) * (4096 + 64), hipMemcpyDeviceToHost));
    return QRGPU_OK;
}

int qrgpu_debug_timeline_trace(qrgpu_ctx *c, long long *host_out /* [16][1024] what happened to each robot in each epoch & 15 (QR_TRACE bits) */)
{
    if (!c || !c->d_timeline || !host_out) return QRGPU_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    HIPCHK(c, hipMemcpy(host_out, c->d_timeline + 768 + 32768 + 4096 + 64, sizeof(long long) * 16384, hipMemcpyDeviceToHost));
    return QRGPU_OK;
}

int qrgpu_debug_words(qrgpu_ctx *c, unsigned *solved, unsigned *wbc_done, int n)
{   // undocumented diagnostic: the per-robot epoch words of the overlapped tick
    if (!c || n <= 0 || n > c->max_batch) return QRGPU_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    if (solved) HIPCHK(c, hipMemcpy(solved, c->d_solved, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost));
    if (wbc_done) HIPCHK(c, hipMemcpy(wbc_done, c->d_wbc_done, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost));
    return QRGPU_OK;
}

int qrgpu_debug_timeline(qrgpu_ctx *c, long long *host_out /* [65][8] (row 64, entry 0: the last tick's epoch), or NULL to switch on and reset */)
{   // undocumented diagnostic: per pipelined tick (ring of 64, indexed by the tick's epoch & 63) on the shared 100 MHz clock:
    // 0 first / 1 last start of a main-pass workgroup, 2 last solve published, 3 first WBC workgroup, 4 last WBC workgroup done,
    // 5 trailing list launch started, 6 ended, 7 second WBC pass ended (0 / LLONG_MAX where nothing was recorded)
    if (!c) return QRGPU_ERR_BAD_ARG;
#ifndef QR_TIMELINE
    c->err = "qrgpu_debug_timeline: the stamps are compiled in only with -DQR_TIMELINE (QRGPU_EXTRA_FLAGS)";
    return QRGPU_ERR_NOT_SETUP;
#endif
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->wbc_stream));
    if (!c->d_timeline) { HIPCHK(c, hipMalloc(&c->d_timeline, sizeof(long long) * (768 + 2 * 16 * 1024 + 4096 + 64 + 16384))); HIPCHK(c, hipMemset(c->d_timeline, 0, sizeof(long long) * (768 + 2 * 16 * 1024 + 4096 + 64 + 16384))); }
    {   // (the planned launch's first start is an atomicMin)
        long long ext[128];
        for (int e = 0; e < 64; ++e) { ext[2 * e] = 0x7fffffffffffffffLL; ext[2 * e + 1] = 0; }
        HIPCHK(c, hipMemcpy(c->d_timeline + 640, ext, sizeof(ext), hipMemcpyHostToDevice));
    }
    if (!c->d_tlr) HIPCHK(c, hipMalloc(&c->d_tlr, sizeof(int) * 4 * (size_t)c->max_batch));
    if (host_out) HIPCHK(c, hipMemcpy(host_out, c->d_timeline, sizeof(long long) * 512, hipMemcpyDeviceToHost));
    long long init[512];
    for (int e = 0; e < 64; ++e) for (int k = 0; k < 8; ++k) init[e * 8 + k] = (k == 0 || k == 3 || k == 5) ? 0x7fffffffffffffffLL : 0;
    HIPCHK(c, hipMemcpy(c->d_timeline, init, sizeof(init), hipMemcpyHostToDevice));
    if (host_out) host_out[512] = (long long)c->tick_epoch;
    return QRGPU_OK;
}

int qrgpu_debug_timeline_robots(qrgpu_ctx *c, int *host_out /* [4][n]: WBC started, flag seen, WBC done, the solve's flag raised */, int n)
{   // undocumented diagnostic: per-robot moments of the last pipelined tick (after qrgpu_debug_timeline switched the stamps on)
    if (!c || !c->d_tlr || !host_out || n <= 0 || n > c->max_batch) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->wbc_stream));
    HIPCHK(c, hipMemcpy(host_out, c->d_tlr, sizeof(int) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(host_out + 3 * (size_t)n, c->d_ftime, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    return QRGPU_OK;
}

int qrgpu_selftest(qrgpu_ctx *c, double *host_out256)
{   // cross-lane helper self-test (tests/test_gpu_mpc.py::test_wave_helpers)
    if (!c || !host_out256) return QRGPU_ERR_BAD_ARG;
    double *d = nullptr;
    HIPCHK(c, hipMalloc(&d, 256 * sizeof(double)));
    hipLaunchKernelGGL(qr_selftest_kernel, dim3(1), dim3(64), 0, c->stream, d);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host_out256, d, 256 * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(d);
    return QRGPU_OK;
}

int qrgpu_enable_flop_count(qrgpu_ctx *c, int on)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    if (on && !c->d_flops) HIPCHK(c, hipMalloc(&c->d_flops, sizeof(double) * 4 * (size_t)c->max_batch));
    c->flops_on = on != 0;
    c->flops_n = 0;
    return QRGPU_OK;
}

int qrgpu_mpc_flop_counts(qrgpu_ctx *c, double out[4])
{
    if (!c || !out) return QRGPU_ERR_BAD_ARG;
    if (!c->d_flops || c->flops_n <= 0) return QRGPU_ERR_NOT_SETUP;
    std::vector<double> h(4 * (size_t)c->flops_n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h.data(), c->d_flops, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    out[0] = out[1] = out[2] = out[3] = 0.0;
    for (int i = 0; i < c->flops_n; ++i) for (int k = 0; k < 4; ++k) out[k] += h[4 * (size_t)i + k];
    return QRGPU_OK;
}

int qrgpu_sync(qrgpu_ctx *c)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->lane[0].h_pre_count && c->lane[0].h_pre_count[2]) {
        // the join of a pipelined tick waited 20 ms for its WBC launch and went on without it: outputs of that tick are incomplete
        c->lane[0].h_pre_count[2] = 0;
        (void)hipStreamSynchronize(c->wbc_stream);
        c->err = "a bounded device-side wait gave up: the WBC launch of a pipelined tick did not finish within 20 ms of its join, or an all-gather did not finish "
                 "(or its tick did not) within 30 s: outputs of that call were incomplete when the stream went on";
        return QRGPU_ERR_LAUNCH;
    }
    return QRGPU_OK;
}

int qrgpu_enable_timing(qrgpu_ctx *c, int on)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    if (on < 0) { c->timing = false; c->timing_paused = true; return QRGPU_OK; }     // pause: what was measured so far is kept
    const bool resume = on > 0 && c->timing_paused;
    c->timing = on != 0;
    c->timing_paused = false;
    c->timing_every = on > 1 ? on : 1;
    if (!resume) {
        c->ev_calls[0] = c->ev_calls[1] = 0;
        c->ev_used[0] = c->ev_used[1] = 0;
    }
    if (on) {
        // the event pairs of the first launches are made here, not inside the caller's timed steps (TimerScope still grows the pool beyond them)
        HIPCHK(c, hipSetDevice(c->device));
        for (int k = 0; k < 2; ++k)
            while (c->ev[k].size() < 512) {
                hipEvent_t a, b;
                HIPCHK(c, hipEventCreate(&a));
                HIPCHK(c, hipEventCreate(&b));
                c->ev[k].push_back({a, b});
            }
    }
    return QRGPU_OK;
}

int qrgpu_get_timing(qrgpu_ctx *c, int kernel, double *mean_ms, int *count)
{
    if (!c || kernel < 0 || kernel > 1 || !mean_ms) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (size_t i = 0; i < c->ev_used[kernel]; ++i) {
        float ms = 0.f;
        // (a pipelined tick records the WBC launch's events on the WBC stream, an overlapped one the main pass's on a lane's stream: the tick's join on
        //  the context's stream polls counts the kernels bump BEFORE they retire, so the second event may still be pending)
        HIPCHK(c, hipEventSynchronize(c->ev[kernel][i].second));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[kernel][i].first, c->ev[kernel][i].second));
        tot += ms;
    }
    *mean_ms = c->ev_used[kernel] ? tot / (double)c->ev_used[kernel] : 0.0;
    if (count) *count = (int)c->ev_used[kernel];
    return QRGPU_OK;
}

void *qrgpu_malloc(qrgpu_ctx *c, unsigned long long bytes)
{
    if (!c) return nullptr;
    void *p = nullptr;
    hipSetDevice(c->device);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void qrgpu_free(qrgpu_ctx *c, void *p) { if (c && p) { hipSetDevice(c->device); hipFree(p); } }
int qrgpu_memcpy_h2d(qrgpu_ctx *c, void *dst, const void *src, unsigned long long bytes)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QRGPU_OK;
}
int qrgpu_memcpy_d2h(qrgpu_ctx *c, void *dst, const void *src, unsigned long long bytes)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QRGPU_OK;
}

void *qrgpu_host_alloc(qrgpu_ctx *c, unsigned long long bytes)
{
    if (!c) return nullptr;
    void *p = nullptr;
    hipSetDevice(c->device);
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void qrgpu_host_free(qrgpu_ctx *c, void *p) { if (c && p) { hipSetDevice(c->device); hipHostFree(p); } }
int qrgpu_memcpy_async(qrgpu_ctx *c, void *dst, const void *src, unsigned long long bytes, int kind)
{
    if (!c || kind < 0 || kind > 2) return QRGPU_ERR_BAD_ARG;
    static const hipMemcpyKind k[3] = {hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice};
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, k[kind], c->stream));
    return QRGPU_OK;
}
int qrgpu_memset_async(qrgpu_ctx *c, void *dst, int byte_value, unsigned long long bytes)
{
    if (!c || !dst) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipMemsetAsync(dst, byte_value, bytes, c->stream));
    return QRGPU_OK;
}
int qrgpu_mark(qrgpu_ctx *c, int index)
{
    if (!c || index < 0 || index >= 65536) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    while ((int)c->marks.size() <= index) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreate(&e));
        c->marks.push_back(e);
    }
    HIPCHK(c, hipEventRecord(c->marks[index], c->stream));
    return QRGPU_OK;
}
int qrgpu_mark_elapsed_ms(qrgpu_ctx *c, int from, int to, double *ms)
{
    if (!c || !ms || from < 0 || to < 0 || from >= (int)c->marks.size() || to >= (int)c->marks.size()) return QRGPU_ERR_BAD_ARG;
    HIPCHK(c, hipEventSynchronize(c->marks[to]));
    float f = 0.f;
    HIPCHK(c, hipEventElapsedTime(&f, c->marks[from], c->marks[to]));
    *ms = f;
    return QRGPU_OK;
}

}  // extern "C"
