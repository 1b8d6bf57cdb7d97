// TEST INFRASTRUCTURE (see qr_oracle.h).  Flat C entry points for ctypes
// (tests/, bench.py cpu_baseline, __graft_entry__.smoke()).
//
// Packed layouts (shared with include/qrgpu.h so tests can feed both sides the
// same arrays):
//   mpc_cfg[20]  = dt, mu, fmax, mass, inertia[3], weights[12], alpha
//   mpc_state[28]= p[3], v[3], quat_wxyz[4], w[3], r[12] (3x4 column-major), rpy[3]
//   fb_state[37] = quat_wxyz[4], pos[3], bodyVel[6] (omega_body, v_body), q[12], qd[12]
//   wbc_cmd[67]  = pBody_des[3], vBody_des[3], aBody_des[3], pBody_RPY_des[3], vBody_Ori_des[3],
//                  pFoot_des[12], vFoot_des[12], aFoot_des[12], Fr_des[12], contact[4] (0/1)
//   model[6]     = hip_l, upper_l, lower_l, body_size[3]
#include "qr_oracle.h"
#include <cmath>
#include <thread>
#include <chrono>

using namespace qro;

namespace {

MpcConfig unpack_cfg(const float *c, int horizon)
{
    MpcConfig cfg;
    cfg.dt = c[0]; cfg.mu = c[1]; cfg.fmax = c[2]; cfg.mass = c[3];
    for (int i = 0; i < 3; ++i) cfg.inertia[i] = c[4 + i];
    for (int i = 0; i < 12; ++i) cfg.weights[i] = c[7 + i];
    cfg.alpha = c[19];
    cfg.horizon = horizon;
    return cfg;
}
MpcInput unpack_in(const float *s, const float *traj, const float *gait, int h)
{
    MpcInput in;
    memset(&in, 0, sizeof(in));
    memcpy(in.p, s, 12); memcpy(in.v, s + 3, 12); memcpy(in.quat, s + 6, 16); memcpy(in.w, s + 10, 12);
    memcpy(in.r, s + 13, 48); memcpy(in.rpy, s + 25, 12);
    memcpy(in.traj, traj, sizeof(float) * 12 * h);
    memcpy(in.gait, gait, sizeof(float) * 4 * h);
    return in;
}
ModelDesc unpack_model(const float *m)
{
    ModelDesc md;
    md.hip_l = m[0]; md.upper_l = m[1]; md.lower_l = m[2];
    for (int i = 0; i < 3; ++i) md.body_size[i] = m[3 + i];
    return md;
}
template <typename T> FBState<T> unpack_state(const T *s)
{
    FBState<T> st;
    for (int i = 0; i < 4; ++i) st.quat[i] = s[i];
    for (int i = 0; i < 3; ++i) st.pos[i] = s[4 + i];
    for (int i = 0; i < 6; ++i) st.bodyVel[i] = s[7 + i];
    for (int i = 0; i < 12; ++i) { st.q[i] = s[13 + i]; st.qd[i] = s[25 + i]; }
    return st;
}
template <typename T> WbcCmd<T> unpack_cmd(const T *c)
{
    WbcCmd<T> cmd;
    for (int i = 0; i < 3; ++i) {
        cmd.pBody_des[i] = c[i]; cmd.vBody_des[i] = c[3 + i]; cmd.aBody_des[i] = c[6 + i];
        cmd.pBody_RPY_des[i] = c[9 + i]; cmd.vBody_Ori_des[i] = c[12 + i];
    }
    for (int l = 0; l < 4; ++l)
        for (int i = 0; i < 3; ++i) {
            cmd.pFoot_des[l][i] = c[15 + 3 * l + i]; cmd.vFoot_des[l][i] = c[27 + 3 * l + i];
            cmd.aFoot_des[l][i] = c[39 + 3 * l + i]; cmd.Fr_des[l][i] = c[51 + 3 * l + i];
        }
    for (int l = 0; l < 4; ++l) cmd.contact[l] = c[63 + l] != T(0);
    return cmd;
}
void pack_stats(const QpStats &st, int *out) { if (out) { out[0] = st.iters; out[1] = st.adds; out[2] = st.drops; out[3] = st.n_active; } }

template <typename T>
void fb_compute_c(const float *model, const T *state, T *H, T *G, T *C, T *Jc, T *Jcdqd, T *pGC, T *vGC)
{
    FBResult<T> r;
    fb_compute(unpack_model(model), unpack_state(state), r);
    for (int i = 0; i < 324; ++i) H[i] = r.H.d[i];
    for (int i = 0; i < 18; ++i) { G[i] = r.G[i]; C[i] = r.C[i]; }
    for (int k = 0; k < 4; ++k) {
        for (int i = 0; i < 54; ++i) Jc[54 * k + i] = r.Jc[k].d[i];
        for (int i = 0; i < 3; ++i) { Jcdqd[3 * k + i] = r.Jcdqd[k][i]; pGC[3 * k + i] = r.pGC[k][i]; vGC[3 * k + i] = r.vGC[k][i]; }
    }
}
template <typename T>
int wbc_run_c(const float *model, const T *state, const T *cmd, T *prev_ori_vel, T *tau, T *qdes, T *qddes, T *fr, T *qddot, int *stats)
{
    WbcOut<T> o;
    wbc_run(unpack_model(model), unpack_state(state), unpack_cmd(cmd), prev_ori_vel, o);
    for (int i = 0; i < 12; ++i) { tau[i] = o.tau[i]; if (qdes) qdes[i] = o.qdes[i]; if (qddes) qddes[i] = o.qddes[i]; if (fr) fr[i] = o.fr[i]; }
    if (qddot) for (int i = 0; i < 18; ++i) qddot[i] = o.qddot[i];
    pack_stats(o.qp, stats);
    return o.qp_status;
}

// The part of one robot's tick behind the MPC solve: K7 torque map of the forces `u` (first horizon step), the +-0.9 N m abad compensation,
// the WBC tick fed with Fr_des := u, the stance / swing merge and the +-23 N m clip.  legCmd[motor].tua (a double, qrMotorCommand) as
// qrFSMStateLocomotion::Run leaves it (QS/fsm/qr_fsm_state_locomotion.cpp:131-156): hybridAction's tua, then +-0.9 on every abad motor
// (:141-151), then -- full tick -- UpdateLegCMD overwrites the stance legs (qr_wbc_locomotion_controller.cpp:205-219), then the clip of
// CheckForceFeedForward (QS/fsm/qr_safety_checker.cpp:48-66).  WT: the arithmetic of the WBC (float as the reference computes; double
// for the kernel-side comparison).  Returns the WBC QP's status << 4.
template <typename WT>
int tick_tail(const LegGeom &g, const ModelDesc &md, const float *fs /*fb_state37*/, const float *cmd_in /*wbc_cmd67*/, float *prev3,
              const double *u /*12 forces*/, int mode, int epilogue, float *tau12_out, float *qdes24_out)
{
    int rc = 0;
    float tau[12];
    mpc_force_to_torque(g, fs, fs + 13, u, tau);
    double tua[12];
    for (int k = 0; k < 12; ++k) {
        tua[k] = (double)tau[k];
        if ((epilogue & 1) && k % 3 == 0) { float comp = 0.9f * (float)std::pow(-1.0, (double)((k / 3 + 1) % 2)); tua[k] += comp; }
    }
    if (mode == 1) {
        float cmd[67];
        memcpy(cmd, cmd_in, sizeof(cmd));
        for (int k = 0; k < 12; ++k) cmd[51 + k] = (float)u[k];          // wbcData.Fr_des[leg] = f.col(leg) (:408)
        WT sw[37], cw[67], pw[3];
        for (int k = 0; k < 37; ++k) sw[k] = (WT)fs[k];
        for (int k = 0; k < 67; ++k) cw[k] = (WT)cmd[k];
        for (int k = 0; k < 3; ++k) pw[k] = (WT)prev3[k];
        WbcOut<WT> o;
        wbc_run(md, unpack_state(sw), unpack_cmd(cw), pw, o);
        for (int k = 0; k < 3; ++k) prev3[k] = (float)pw[k];
        for (int l = 0; l < 4; ++l)
            if (cmd[63 + l] != 0.f) for (int j = 0; j < 3; ++j) tua[3 * l + j] = (double)(float)o.tau[3 * l + j];
        if (qdes24_out) for (int k = 0; k < 12; ++k) { qdes24_out[k] = (float)o.qdes[k]; qdes24_out[12 + k] = (float)o.qddes[k]; }
        rc = o.qp_status << 4;
    }
    if (epilogue & 2) for (int k = 0; k < 12; ++k) { if (tua[k] > 23) tua[k] = 23; else if (tua[k] < -23) tua[k] = -23; }
    for (int k = 0; k < 12; ++k) tau12_out[k] = (float)tua[k];
    return rc;
}

// The WBC tick with its relaxation QP laid open: the QP as assembled (QuadProg++ layout), our solver's z or -- with z_in -- the tick
// finished with an externally computed z (the compiled QuadProg++ of oracle/_ref in tests/golden/make_golden.py).  prev is not modified.
template <typename T>
int wbc_qp_c(const float *model, const T *state, const T *cmd, const T *prev_in, const double *z_in, int *dims, double *G, double *g0, double *CE,
             double *ce0, double *CI, double *ci0, double *z_out, T *tau, T *fr)
{
    WbcOut<T> o;
    WbcQpIO io;
    io.z_in = z_in;
    T prev[3] = {prev_in[0], prev_in[1], prev_in[2]};
    wbc_run(unpack_model(model), unpack_state(state), unpack_cmd(cmd), prev, o, &io);
    if (dims) { dims[0] = io.n; dims[1] = io.p; dims[2] = io.m; }
    auto put = [](double *dst, const std::vector<double> &v) { if (dst) memcpy(dst, v.data(), sizeof(double) * v.size()); };
    put(G, io.G); put(g0, io.g0); put(CE, io.CE); put(ce0, io.ce0); put(CI, io.CI); put(ci0, io.ci0); put(z_out, io.z);
    for (int i = 0; i < 12; ++i) { if (tau) tau[i] = o.tau[i]; if (fr) fr[i] = o.fr[i]; }
    return o.qp_status;
}

}  // namespace

extern "C" {

int qro_qp_solve(int n, const double *G, const double *g0, int p, const double *CE, const double *ce0,
                 int m, const double *CI, const double *ci0, double *x, double *lambda, int *stats, double *obj)
{
    QpStats st;
    int rc = qp_solve_gi(n, G, g0, p, CE, ce0, m, CI, ci0, x, lambda, &st);
    pack_stats(st, stats);
    if (obj) *obj = st.obj;
    return rc;
}

void qro_mpc_assemble(const float *cfg, int horizon, const float *state28, const float *traj, const float *gait,
                      int literal, float *H, float *g, float *ub)
{
    MpcAssembly a;
    MpcConfig c = unpack_cfg(cfg, horizon);
    MpcInput in = unpack_in(state28, traj, gait, horizon);
    if (literal) mpc_assemble_literal(c, in, a); else mpc_assemble(c, in, a);
    memcpy(H, a.H.data(), sizeof(float) * a.H.size());
    memcpy(g, a.g.data(), sizeof(float) * a.g.size());
    if (ub) memcpy(ub, a.ub.data(), sizeof(float) * a.ub.size());
}

int qro_mpc_solve(const float *cfg, int horizon, const float *state28, const float *traj, const float *gait,
                  int literal, double *u, int *stats)
{
    MpcAssembly a;
    MpcConfig c = unpack_cfg(cfg, horizon);
    MpcInput in = unpack_in(state28, traj, gait, horizon);
    if (literal) mpc_assemble_literal(c, in, a); else mpc_assemble(c, in, a);
    QpStats st;
    int rc = mpc_solve_qp(a, gait, horizon, u, &st);
    pack_stats(st, stats);
    return rc;
}

// The stated QP of qr_mpc_interface.cpp:396-438 solved from a GIVEN (H, g) -- fp32, n x n row-major and n, as any assembly left them (the
// kernel's bf16-limb Hessian mode of BASELINE configs[4] among them: tests/test_gpu_mpc.py) -- instead of the oracle's own assembly.
// Bounds and friction rows come from the gait table and the configuration exactly as mpc_assemble sets them (:222-240, :386-389); entries of
// H / g that belong to swing variables are never read (mpc_solve_qp eliminates them), so they may hold anything.
int qro_mpc_solve_hg(const float *cfg, int horizon, const float *gait, const float *H, const float *g, double *u, int *stats)
{
    MpcConfig c = unpack_cfg(cfg, horizon);
    MpcAssembly a;
    a.n = 12 * horizon; a.m = 20 * horizon;
    a.H.assign(H, H + (size_t)a.n * a.n);
    a.g.assign(g, g + a.n);
    a.ub.assign(a.m, 0.f);
    for (int k = 0; k < 4 * horizon; ++k) {
        for (int r = 0; r < 4; ++r) a.ub[5 * k + r] = 5e10f;
        a.ub[5 * k + 4] = gait[k] * c.fmax;
    }
    a.invmu = 1.f / c.mu;
    QpStats st;
    int rc = mpc_solve_qp(a, gait, horizon, u, &st);
    pack_stats(st, stats);
    return rc;
}

void qro_mpc_force_to_torque(const float *geom3, const float *quat, const float *q12, const double *f12, float *tau12)
{
    LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2];
    mpc_force_to_torque(g, quat, q12, f12, tau12);
}

void qro_foot_positions(const float *geom3, const float *hipOffset12, const float *q12, float *out12)
{
    LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2];
    foot_positions_in_base_frame(g, hipOffset12, q12, out12);
}

void qro_leg_jacobian(const float *geom3, const float *q3, int leg, float *J9)
{
    LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2];
    analytical_leg_jacobian(g, q3, leg, J9);
}

void qro_rpy_to_quat(const float *rpy, float *quat)
{
    V3<float> r = {{rpy[0], rpy[1], rpy[2]}};
    Q4<float> q = rpyToQuat(r);
    for (int i = 0; i < 4; ++i) quat[i] = q[i];
}

void qro_fb_compute_f32(const float *model, const float *state, float *H, float *G, float *C, float *Jc, float *Jcdqd, float *pGC, float *vGC)
{ fb_compute_c<float>(model, state, H, G, C, Jc, Jcdqd, pGC, vGC); }
void qro_fb_compute_f64(const float *model, const double *state, double *H, double *G, double *C, double *Jc, double *Jcdqd, double *pGC, double *vGC)
{ fb_compute_c<double>(model, state, H, G, C, Jc, Jcdqd, pGC, vGC); }

int qro_wbc_run_f32(const float *model, const float *state, const float *cmd, float *prev, float *tau, float *qdes, float *qddes, float *fr, float *qddot, int *stats)
{ return wbc_run_c<float>(model, state, cmd, prev, tau, qdes, qddes, fr, qddot, stats); }
int qro_wbc_run_f64(const float *model, const double *state, const double *cmd, double *prev, double *tau, double *qdes, double *qddes, double *fr, double *qddot, int *stats)
{ return wbc_run_c<double>(model, state, cmd, prev, tau, qdes, qddes, fr, qddot, stats); }

void qro_pinv_f32(int r, int c, const float *A, double thr, float *out /* c x r */)
{
    Mat<float> m(r, c), inv;
    memcpy(m.d.data(), A, sizeof(float) * r * c);
    pseudoInverse(m, thr, inv);
    memcpy(out, inv.d.data(), sizeof(float) * r * c);
}
void qro_lu_inverse_f32(int n, const float *A, float *out)
{
    Mat<float> m(n, n);
    memcpy(m.d.data(), A, sizeof(float) * n * n);
    Mat<float> inv = luInverse(m);
    memcpy(out, inv.d.data(), sizeof(float) * n * n);
}

int qro_wbc_qp_f32(const float *model, const float *state, const float *cmd, const float *prev, const double *z_in, int *dims, double *G, double *g0,
                   double *CE, double *ce0, double *CI, double *ci0, double *z_out, float *tau, float *fr)
{ return wbc_qp_c<float>(model, state, cmd, prev, z_in, dims, G, g0, CE, ce0, CI, ci0, z_out, tau, fr); }
int qro_wbc_qp_f64(const float *model, const double *state, const double *cmd, const double *prev, const double *z_in, int *dims, double *G, double *g0,
                   double *CE, double *ce0, double *CI, double *ci0, double *z_out, double *tau, double *fr)
{ return wbc_qp_c<double>(model, state, cmd, prev, z_in, dims, G, g0, CE, ce0, CI, ci0, z_out, tau, fr); }

// One robot's tick from given first-step MPC forces (tests: the forces the compiled qpOASES returns when called as the reference calls it).
// wbc_fp64: 0 = the WBC in float (the reference's arithmetic), 1 = in double.  prev3: in/out.
int qro_tick_from_forces(const float *geom3, const float *model, const float *fb_state37, const float *wbc_cmd67, float *prev3, const double *f12,
                         int mode, int epilogue, int wbc_fp64, float *tau12_out, float *qdes24_out)
{
    LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2];
    ModelDesc md = unpack_model(model);
    return wbc_fp64 ? tick_tail<double>(g, md, fb_state37, wbc_cmd67, prev3, f12, mode, epilogue, tau12_out, qdes24_out)
                    : tick_tail<float>(g, md, fb_state37, wbc_cmd67, prev3, f12, mode, epilogue, tau12_out, qdes24_out);
}

// ---------------------------------------------------------------------------
// Batched full tick on host threads: the CPU baseline bench.py reports.
// One tick per robot = MPC (K1-K7) then WBC (K8-K14) fed with that MPC's Fr_des
// (SURVEY.md 8d).  Arrays are AoS per robot here (robot-major); returns seconds.
//   mode: 0 = MPC only (tau = K7 torques), 1 = full tick (tau = WBC torques on stance
//   legs, K7 torques on swing legs untouched by UpdateLegCMD :205-219)
//   epilogue: bit 0 = abad hip compensation, bit 1 = +-23 clip (K14 tail); qdes24_out (may be null): desiredJPos, desiredJVel of K12
// ---------------------------------------------------------------------------
double qro_tick_batch(int nrobots, int nthreads, int mode, const float *cfg, int horizon, const float *geom3, const float *model,
                      const float *mpc_state28, const float *traj, const float *gait,
                      const float *fb_state37, const float *wbc_cmd67, float *prev_ori_vel3,
                      float *force12_out, float *tau12_out, int *status_out, int epilogue, float *qdes24_out)
{
    auto t0 = std::chrono::steady_clock::now();
    MpcConfig c = unpack_cfg(cfg, horizon);
    LegGeom g; g.hip_l = geom3[0]; g.upper_l = geom3[1]; g.lower_l = geom3[2];
    ModelDesc md = unpack_model(model);
    auto work = [&](int lo, int hi) {
        std::vector<double> u(12 * horizon);
        for (int i = lo; i < hi; ++i) {
            MpcInput in = unpack_in(mpc_state28 + 28 * i, traj + (size_t)12 * horizon * i, gait + (size_t)4 * horizon * i, horizon);
            MpcAssembly a;
            mpc_assemble(c, in, a);
            QpStats st;
            int rc = mpc_solve_qp(a, in.gait, horizon, u.data(), &st);
            for (int k = 0; k < 12; ++k) force12_out[12 * i + k] = (float)u[k];
            rc |= tick_tail<float>(g, md, fb_state37 + 37 * i, wbc_cmd67 + 67 * i, prev_ori_vel3 + 3 * i, u.data(), mode, epilogue, tau12_out + 12 * i,
                                   qdes24_out ? qdes24_out + 24 * i : nullptr);
            if (status_out) status_out[i] = rc;
        }
    };
    if (nthreads <= 1) work(0, nrobots);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) {
            int lo = (int)((long long)nrobots * t / nthreads), hi = (int)((long long)nrobots * (t + 1) / nthreads);
            th.emplace_back(work, lo, hi);
        }
        for (auto &x : th) x.join();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
