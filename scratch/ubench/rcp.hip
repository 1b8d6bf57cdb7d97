// How accurate is v_rcp_f64, and how many Newton steps does fast_rcp need?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r); r1[i] = r;
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r); r2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (double)(s >> 11) / 9007199254740992.0; x[i] = std::ldexp(1.0 + u, (int)(s % 80) - 40); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) { const long double t = 1.0L / (long double)x[i]; e0 = std::fmax(e0, (double)fabsl((a[i] - t) / t)); e1 = std::fmax(e1, (double)fabsl((b[i] - t) / t)); e2 = std::fmax(e2, (double)fabsl((c[i] - t) / t)); }
    printf("max relative error of v_rcp_f64: %.3e | + 1 Newton step: %.3e | + 2: %.3e  (eps = 1.1e-16)\n", e0, e1, e2);
    return 0;
}
