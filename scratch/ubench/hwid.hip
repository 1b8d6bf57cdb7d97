// Where do the four waves of a 256-thread workgroup land (SIMD, wave slot) when two such workgroups share a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
__global__ void __launch_bounds__(256) k(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    lds[threadIdx.x] = x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
    if (lds[(threadIdx.x + 1) & 255] == 12345.678) out[0] = 0;
}
int main()
{
    const int nb = 1024;
    unsigned *d; hipMalloc(&d, nb * 4 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 80000, 0, d, 20000);
    unsigned h[nb * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx90a+: se_id [14:13])
    std::map<unsigned, int> pat;
    for (int b = 0; b < nb; ++b) {
        unsigned key = 0;
        for (int w = 0; w < 4; ++w) key |= (((h[b * 4 + w] >> 4) & 3) | ((h[b * 4 + w] & 15) << 2)) << (8 * w);
        pat[key]++;
    }
    for (auto &p : pat) {
        printf("%4d WGs: ", p.second);
        for (int w = 0; w < 4; ++w) printf(" wave%d simd %u slot %u |", w, (p.first >> (8 * w)) & 3, (p.first >> (8 * w + 2)) & 15);
        printf("\n");
    }
    printf("first WGs raw:"); for (int i = 0; i < 8; ++i) printf(" %08x", h[i]); printf("\n");
    return 0;
}
