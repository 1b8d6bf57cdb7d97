// Latency microbenchmarks of the primitives in the active-set loop (one wave per SIMD, like the MPC kernel's GI phase).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../quadruped-robot_amd/csrc/qr_wave_helpers.h"
using namespace qrgpu;
#define REP 64
__global__ void __launch_bounds__(256) k(double *out, long long *tk, const double *in)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double x = in[threadIdx.x], y = in[threadIdx.x + 256], acc = 0.0;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i & 511];
    __syncthreads();
    long long t0, t1; int s = 0; float xf = 0.f; int fl = 0; int idx = lane; double a0 = 0.0;
#define FX asm volatile("s_nop 0" : "+v"(x), "+v"(xf), "+v"(fl), "+v"(idx), "+v"(a0) :: "memory");
#define T0 FX t0 = clock64(); FX
#define T1 FX t1 = clock64(); FX if (threadIdx.x == 0) tk[s] = t1 - t0; ++s;
    // 0: clock overhead
    T0 T1
    // 1: dependent fma f64
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) x = __builtin_fma(x, y, 1e-9);
    T1
    // 2: dependent fma f32
    xf = (float)x; float yf = (float)y;
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) xf = __builtin_fmaf(xf, yf, 1e-9f);
    T1
    x += xf;
    // 3: wave_min_d
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) x = wave_min_d(x + (double)lane) + y;
    T1
    // 4: wave_sum_d
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) x = wave_sum_d(x * 1e-3) + y;
    T1
    // 5: readlane_d dependent
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) x = readlane_d(x, (i * 7) & 63) + y;
    T1
    // 6: shfl (bpermute) dependent double
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) x = __shfl(x, (lane + i) & 63, 64) + y;
    T1
    // 7: LDS load dependent (address depends on previous value)
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) { double v = lds[idx]; idx = ((int)v + lane + i) & 4095; acc += v; }
    T1
    // 8: barrier
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) { __syncthreads(); }
    T1
    // 9: LDS write + barrier + read (exchange)
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) { lds[wv * 64 + lane] = x; __syncthreads(); x = lds[lane] + lds[64 + lane] + lds[128 + lane] + lds[192 + lane]; __syncthreads(); }
    T1
    // 10: first_lane + readfirstlane
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) { fl += __builtin_amdgcn_readfirstlane(first_lane(x + fl > (double)lane)); }
    T1
    // 11: fast_rcp dependent
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) x = fast_rcp(x + 2.0);
    T1
    // 12: 9-double block load (b128 x4.5) + 9 fma dependent on the address
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        const double *B = lds + ((idx * 9) & 2047);
        double w = 0; 
#pragma unroll
        for (int j = 0; j < 9; ++j) w += B[j] * y;
        idx = ((int)w + i) & 63; acc += w;
    }
    T1
    // 13: v_cmp + cndmask chain on doubles (select min of 6)
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) { double s0 = x + i; if (y < s0) s0 = y; x = s0 * 1.0000001; }
    T1
    // 14: independent fma f64 x4 (ILP)
    a0 = x; double a1 = y, a2 = x + 1, a3 = y + 1;
    T0
#pragma unroll
    for (int i = 0; i < REP; ++i) { a0 = __builtin_fma(a0, y, 1e-9); a1 = __builtin_fma(a1, y, 1e-9); a2 = __builtin_fma(a2, y, 1e-9); a3 = __builtin_fma(a3, y, 1e-9); }
    T1
    out[threadIdx.x] = x + acc + fl + a0 + a1 + a2 + a3;
}
int main()
{
    double *in, *out; long long *tk;
    hipMalloc(&in, 4096 * 8); hipMalloc(&out, 256 * 8); hipMalloc(&tk, 32 * 8);
    double h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 1.0 + (i % 17) * 0.01;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096 * 8, 0, out, tk, in);
    hipDeviceSynchronize();
    long long t[32]; hipMemcpy(t, tk, sizeof(t), hipMemcpyDeviceToHost);
    const char *nm[] = {"clock overhead", "dep fma f64", "dep fma f32", "wave_min_d(+add)", "wave_sum_d(+mul,add)", "readlane_d(+add)", "shfl double(+add)",
                        "LDS load dependent", "s_barrier (4 waves)", "LDS exchange (write,bar,read4,bar)", "first_lane+readfirstlane", "fast_rcp(+add)",
                        "block load 9 + 9 fma", "cmp+select+mul f64", "4 independent fma f64 (per group)"};
    for (int i = 0; i < 15; ++i) printf("%-36s %8.1f ticks per op\n", nm[i], (double)(t[i] - (i ? t[0] : 0)) / (i ? REP : 1));
    return 0;
}
