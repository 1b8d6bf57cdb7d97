// fp64 MFMA / FMA issue rates on gfx950 with 1 and 2 waves per SIMD (per-wave timings, no CSE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 16
__global__ void __launch_bounds__(512) k(double *out, long long *tk, const double *in, int mode)
{
    const int wave = threadIdx.x >> 6;
    double a = in[threadIdx.x], b = in[threadIdx.x + 512];
    d4 c[8]; double av[8], bv[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c[i] = (d4){0.0 + i, 1.0, 2.0, 3.0 * i}; av[i] = a + 0.5 * i; bv[i] = b - 0.25 * i; x[i] = a * i; }
#define FENCE _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(c[i]), "+v"(av[i]), "+v"(bv[i]), "+v"(x[i]));
    FENCE
    __syncthreads();
    long long t0 = clock64();
    FENCE
    if (mode == 0) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], c[i], 0, 0, 0);
    } else if (mode == 1) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], bv[i], av[i]);
    } else if (mode == 2) {       // mfma + 8 independent fma
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) { c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], c[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], bv[i], av[j]); }
    } else {                      // mfma + 16 independent int ops
        int y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = threadIdx.x + j;
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) { c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], c[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) { y[j] = y[j] * 3 + i; y[j] ^= y[j] >> 3; } }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
    }
    FENCE
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) tk[blockIdx.x * 8 + wave] = t1 - t0;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += c[i][0] + c[i][1] + c[i][2] + c[i][3] + x[i];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
int main()
{
    double *in, *out; long long *tk;
    hipMalloc(&in, 1024 * 8); hipMalloc(&out, 512 * 512 * 8); hipMalloc(&tk, 512 * 8 * 8);
    double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1e-3 * (i % 7);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const char *nm[4] = {"8 mfma f64 16x16x4 (independent)", "8 v_fma_f64 (independent)", "mfma + 8 fma", "mfma + 16 int ops"};
    for (int mode = 0; mode < 4; ++mode)
        for (int thr = 256; thr <= 512; thr += 256) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(thr), 0, 0, out, tk, in, mode);
            long long t[512 * 8]; hipMemcpy(t, tk, sizeof(t), hipMemcpyDeviceToHost);
            double s = 0; int nw = thr / 64;
            for (int bI = 0; bI < 256; ++bI) for (int w = 0; w < nw; ++w) s += t[bI * 8 + w];
            s /= 256.0 * nw;
            printf("%-36s %d waves/SIMD: %.1f ticks per inner item (of 8 per round)\n", nm[mode], thr / 256, s / (REP * 8));
        }
    return 0;
}
