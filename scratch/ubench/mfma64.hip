// Throughput / latency of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, and a layout check of the f64 MFMA with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 32
__global__ void __launch_bounds__(512) k(double *out, long long *tk, const double *in)
{
    const int lane = threadIdx.x & 63;
    double a = in[threadIdx.x], b = in[threadIdx.x + 256];
    d4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = (d4){0.0 + i, 1.0, 2.0, 3.0 * i};
    double av[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { av[i] = a + 0.5 * i; bv[i] = b - 0.25 * i; asm volatile("" : "+v"(av[i]), "+v"(bv[i])); }
    long long t0, t1; int s = 0;
#define FX asm volatile("s_nop 0" : "+v"(a), "+v"(b) :: "memory");
#define T0 FX t0 = clock64(); FX
#define T1 FX t1 = clock64(); FX if (threadIdx.x == 0) tk[s] = t1 - t0; ++s;
    T0 T1
    // 1: 8 independent accumulators, back to back
    T0
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], c[i], 0, 0, 0);
    T1
    // 2: dependent chain on one accumulator
    T0
#pragma unroll
    for (int r = 0; r < REP * 2; ++r) c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[0], 0, 0, 0);
    T1
    // 3: independent v_fma_f64 x 8
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = a + i; asm volatile("" : "+v"(x[i])); }
    T0
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], bv[i], av[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
    T1
    // 4: mfma interleaved with independent fma (do the pipes overlap?)
    T0
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) { c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], c[i], 0, 0, 0); x[i] = __builtin_fma(x[i], bv[i], av[i]); x[(i + 1) & 7] = __builtin_fma(x[(i + 1) & 7], bv[i], av[i]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
    T1
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += c[i][0] + c[i][1] + c[i][2] + c[i][3] + x[i];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
// layout check: D = A(16x4) * B(4x16), A[i][k] = i + 100k (asymmetric), B[k][j] = (k+1) * (j + 1) + 1000 * k
__global__ void chk(double *D)
{
    const int l = threadIdx.x;
    const double a = (double)((l & 15) + 100 * (l >> 4));
    const int kk = l >> 4, j = l & 15;
    const double b = (double)((kk + 1) * (j + 1) + 1000 * kk);
    d4 c = (d4){0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main()
{
    double *in, *out, *D; long long *tk;
    hipMalloc(&in, 1024 * 8); hipMalloc(&out, 1024 * 512 * 8); hipMalloc(&tk, 64 * 8); hipMalloc(&D, 256 * 8);
    double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1e-3 * (i % 7);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, tk, in);
    long long t[8]; hipMemcpy(t, tk, sizeof(t), hipMemcpyDeviceToHost);
    printf("1 WG of 4 waves (one per SIMD): clock overhead %lld\n", t[0]);
    printf("mfma_f64_16x16x4 independent: %.1f ticks each\n", (double)(t[1] - t[0]) / (REP * 8));
    printf("mfma_f64_16x16x4 dependent:   %.1f ticks each\n", (double)(t[2] - t[0]) / (REP * 2));
    printf("v_fma_f64 independent:        %.1f ticks each\n", (double)(t[3] - t[0]) / (REP * 8));
    printf("mfma + 2 fma interleaved:     %.1f ticks per group\n", (double)(t[4] - t[0]) / (REP * 8));
    // two WGs per CU: 512 WGs on 256 CUs
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, out, tk, in);
    hipMemcpy(t, tk, sizeof(t), hipMemcpyDeviceToHost);
    printf("512 WGs of 8 waves (2 per SIMD): mfma indep %.1f, fma indep %.1f, interleaved %.1f\n", (double)(t[1] - t[0]) / (REP * 8), (double)(t[3] - t[0]) / (REP * 8), (double)(t[4] - t[0]) / (REP * 8));
    hipLaunchKernelGGL(chk, dim3(1), dim3(64), 0, 0, D);
    double hd[256]; hipMemcpy(hd, D, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double r = 0; for (int kk = 0; kk < 4; ++kk) r += (double)(i + 100 * kk) * (double)((kk + 1) * (j + 1) + 1000 * kk);
        if (r != hd[i * 16 + j]) ++bad;
    }
    printf("layout check: %d mismatches of 256\n", bad);
    return 0;
}
