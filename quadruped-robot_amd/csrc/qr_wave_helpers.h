// Cross-lane helpers for wave64 on gfx950: DPP row all-reduce + v_readlane combine (no LDS traffic, no
// ds_bpermute), uniform broadcasts, fast reciprocal.  Shared by the MPC and WBC kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace qrgpu {

// 1/x by v_rcp_f64 + two Newton steps (<= 1 ulp-ish; the active-set step lengths do not need IEEE division)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// 1/x by v_rcp_f64 (4.6e-8 relative, scratch/ubench/rcp.hip) + one Newton step: 2.2e-15 relative -- for the pivots of a sweep, whose
// reciprocal sits on the dependent chain of every step
__device__ __forceinline__ double fast_rcp1(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// ---- cross-lane helpers (wave64) ---------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// srclane must be wave-uniform
__device__ __forceinline__ double readlane_d(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
// All-reduce inside each 16-lane row: xor 1, xor 2 (quad_perm), rotate by 4 and 8 (row_ror); then
// the four row results are combined through v_readlane.  Result is wave-uniform.
__device__ __forceinline__ double wave_min_d(double v)
{
    v = fmin(v, dpp_d<0xB1>(v));      // quad_perm [1,0,3,2]
    v = fmin(v, dpp_d<0x4E>(v));      // quad_perm [2,3,0,1]
    v = fmin(v, dpp_d<0x124>(v));     // row_ror:4
    v = fmin(v, dpp_d<0x128>(v));     // row_ror:8
    const double a = readlane_d(v, 0), b = readlane_d(v, 16), c = readlane_d(v, 32), d = readlane_d(v, 48);
    return fmin(fmin(a, b), fmin(c, d));
}
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x124>(v);
    v += dpp_d<0x128>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
// lowest lane whose predicate holds (or -1); uniform
__device__ __forceinline__ int first_lane(bool pred)
{
    const unsigned long long m = __ballot(pred);
    return m ? (int)__builtin_ctzll(m) : -1;
}


// Column j of the analytic leg Jacobian (AnalyticalLegJacobian, quadruped/src/robots/qr_robot.cpp:148-172) in fp32.
__device__ __forceinline__ void leg_jacobian_column(int j, float t0, float t1, float t2, float sh, float lu, float ll, float &J0, float &J1, float &J2)
{
    const float lEff = sqrtf(lu * lu + ll * ll + 2 * lu * ll * cosf(t2));
    const float tEff = t1 + t2 / 2;
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
}

}  // namespace qrgpu
