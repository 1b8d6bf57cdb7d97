// Does a CU-masked stream confine workgroups on gfx950 (8 XCDs), and how do mask bits map to (XCC, SE, CU)?  And do two streams with disjoint
// masks run side by side at full rate?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <set>
#include <vector>
#include <chrono>
__global__ void __launch_bounds__(256) k(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = (xcc & 0xf) | (x == 12345.678 ? 16 : 0); }
}
static void report(const char *name, const std::vector<unsigned> &h, int nb)
{
    std::set<unsigned> cus; int perx[8] = {0};
    std::set<unsigned> perxset[8];
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[2 * b], x = h[2 * b + 1] & 7;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = (x << 16) | (se << 8) | (sh << 4) | cu;
        cus.insert(key); perxset[x].insert(key);
    }
    printf("%s: %zu distinct CUs; per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", perxset[x].size());
    printf("\n");
}
int main()
{
    const int nb = 4096;
    unsigned *d; hipMalloc(&d, nb * 2 * 4);
    std::vector<unsigned> h(nb * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    // unmasked
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, s0, d, 2000); hipStreamSynchronize(s0);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost); report("no mask", h, nb);
    // first 216 bits / last 40 bits
    uint32_t mA[8], mB[8];
    for (int w = 0; w < 8; ++w) { mA[w] = 0; mB[w] = 0; }
    for (int b = 0; b < 256; ++b) { if (b < 216) mA[b >> 5] |= 1u << (b & 31); else mB[b >> 5] |= 1u << (b & 31); }
    hipStream_t sA, sB;
    hipError_t eA = hipExtStreamCreateWithCUMask(&sA, 8, mA), eB = hipExtStreamCreateWithCUMask(&sB, 8, mB);
    printf("create: %s %s\n", hipGetErrorString(eA), hipGetErrorString(eB));
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, sA, d, 2000); hipStreamSynchronize(sA);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost); report("mask A (bits 0..215)", h, nb);
    std::set<unsigned> setA; for (int b = 0; b < nb; ++b) setA.insert(((h[2*b+1]&7) << 16) | ((h[2*b] >> 8) & 0xff));
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, sB, d, 2000); hipStreamSynchronize(sB);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost); report("mask B (bits 216..255)", h, nb);
    int overlap = 0; std::set<unsigned> setB; for (int b = 0; b < nb; ++b) setB.insert(((h[2*b+1]&7) << 16) | ((h[2*b] >> 8) & 0xff));
    for (auto v : setB) if (setA.count(v)) ++overlap;
    printf("CUs in both: %d\n", overlap);
    // timing: A alone, B alone, both together (work proportional to CU share)
    unsigned *d2; hipMalloc(&d2, nb * 2 * 4);
    auto run = [&](bool a, bool b_) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (a) hipLaunchKernelGGL(k, dim3(216 * 8 * 4), dim3(256), 0, sA, d, 20000);
        if (b_) hipLaunchKernelGGL(k, dim3(40 * 8 * 4), dim3(256), 0, sB, d2, 20000);
        hipDeviceSynchronize();
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    run(true, true);
    printf("A alone %.1f us, B alone %.1f us, both %.1f us\n", run(true, false), run(false, true), run(true, true));
    return 0;
}
