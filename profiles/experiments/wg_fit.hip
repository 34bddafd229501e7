// wg_fit.hip -- do TWO workgroups of W waves fit one CU (gfx950)?  Each workgroup spins for a fixed number of clock ticks;
// 2 x CUs workgroups take ~1 x the spin when two fit per CU and ~2 x when only one does.  Variants: registers per lane
// (forced by clobbering the highest one), dynamic LDS per workgroup, waves per workgroup.
//   hipcc --offload-arch=gfx950 -O2 -o wg_fit wg_fit.hip && ./wg_fit
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W, int VG>
__global__ void __launch_bounds__(64 * W) spin(unsigned long long ticks, unsigned *sink)
{
    extern __shared__ unsigned sm[];
    if (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VG == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    sm[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sm[threadIdx.x] == 0xFFFFFFFFu) sink[0] = 1;
}
template <int W, int VG>
float run(size_t lds, int wgs, unsigned *sink)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)spin<W, VG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    spin<W, VG><<<wgs, 64 * W, lds>>>(1000, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<W, VG><<<wgs, 64 * W, lds>>>(20000000ull, sink);  // ~0.2 s at the 100 MHz counter
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned *sink;
    hipMalloc(&sink, 4);
    const float one = run<4, 0>(1024, cus, sink);
    printf("CUs %d, one workgroup per CU: %.1f ms\n", cus, one);
#define T(W, VG, LDS)                                                                 \
    {                                                                                 \
        printf("W=%2d vgpr=%3d lds=%6d  time / (one workgroup per CU) at n per CU:", W, VG, LDS); \
        for (int n = 2; n <= 6; n++) printf("  %d: %.2f", n, run<W, VG>(LDS, n * cus, sink) / one); \
        printf("\n");                                                                 \
    }
    T(10, 96, 1024);
    T(10, 80, 1024);
    T(10, 96, 62080);
    T(9, 96, 1024);
    T(8, 96, 1024);
    T(7, 96, 1024);
    T(6, 96, 1024);
    T(5, 96, 1024);
    T(4, 96, 1024);
    T(3, 96, 1024);
    T(2, 96, 1024);
    T(5, 128, 1024);
    T(6, 128, 1024);
    T(4, 128, 1024);
    return 0;
}
