// Occupancy under a CU-masked stream: do two 80 KB / 512-thread workgroups still share a CU?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void __launch_bounds__(512, 4) k(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    lds[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)lds[1];
}
int main()
{
    unsigned *d; hipMalloc(&d, 1 << 20);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    uint32_t mA[8] = {0}, mAll[8];
    for (int b = 0; b < 216; ++b) mA[b >> 5] |= 1u << (b & 31);
    for (int w = 0; w < 8; ++w) mAll[w] = 0xffffffffu;
    hipStream_t s0, sA, sAll;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipExtStreamCreateWithCUMask(&sA, 8, mA);
    hipExtStreamCreateWithCUMask(&sAll, 8, mAll);
    auto run = [&](hipStream_t s, int grid, int lds) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, d, 40000);
        hipDeviceSynchronize();
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    run(s0, 512, 1024);
    for (int ncu : {224, 192, 240, 208}) {
        uint32_t m[8] = {0}, mc[8] = {0};
        for (int b = 0; b < 256; ++b) { if (b < ncu) m[b >> 5] |= 1u << (b & 31); else mc[b >> 5] |= 1u << (b & 31); }
        hipStream_t s, sc; hipExtStreamCreateWithCUMask(&s, 8, m); hipExtStreamCreateWithCUMask(&sc, 8, mc);
        const int w = 256 - ncu;
        printf("first %d CUs, 80 KB: %d WGs %.0f us, %d WGs %.0f us | the other %d CUs, 39 KB: %d WGs %.0f us, %d WGs %.0f us, %d WGs %.0f us\n", ncu, ncu, run(s, ncu, 81920), 2 * ncu,
               run(s, 2 * ncu, 81920), w, w, run(sc, w, 39936), 2 * w, run(sc, 2 * w, 39936), 4 * w, run(sc, 4 * w, 39936));
    }
    return 0;
}
