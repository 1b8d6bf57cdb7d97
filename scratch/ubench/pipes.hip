// Do two kernels from two streams have their workgroups dispatched side by side, or does one grid's dispatch wait for the other's to finish?
// Kernel A: 2048 workgroups of 80 KB (the machine holds 512) on stream 0; kernel B: the same on stream k, launched 20 us later.
// Reported: when B's first workgroup started relative to A's first and to A's LAST workgroup start.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ long long wall() { return __builtin_readcyclecounter(); }
__global__ void __launch_bounds__(512, 4) k(long long *first, long long *last, int spin)
{
    extern __shared__ double lds[];
    long long t;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    if (threadIdx.x == 0) { atomicMin((unsigned long long *)first, (unsigned long long)t); atomicMax((unsigned long long *)last, (unsigned long long)t); }
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    lds[threadIdx.x] = x;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 511] == 12345.678) first[2] = 0;
}
int main()
{
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    const int NS = 10;
    hipStream_t s[NS];
    for (int i = 0; i < NS; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    long long *d; hipMalloc(&d, 64);
    for (int b = 1; b < NS; ++b) {
        long long init[4] = {0x7fffffffffffffffLL, 0, 0x7fffffffffffffffLL, 0};
        hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(2048), dim3(512), 81920, s[0], d, d + 1, 4000);
        hipLaunchKernelGGL(k, dim3(2048), dim3(512), 81920, s[b], d + 2, d + 3, 4000);
        hipDeviceSynchronize();
        long long h[4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stream 0 vs %d: A first..last start %.1f us; B first start %.1f us after A's first (%.1f after A's last), B last %.1f\n", b, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0,
               (h[2] - h[1]) / 100.0, (h[3] - h[0]) / 100.0);
    }
    return 0;
}
