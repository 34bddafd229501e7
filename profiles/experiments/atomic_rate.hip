// atomic_rate.hip -- how many device-scope atomicAdd (returning) on ONE address the chip sustains: the work-queue form of the
// frame kernel would fetch one ticket per wave and 16 frames (~90 M tickets/s for the whole batch).
//   hipcc --offload-arch=gfx950 -O2 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) hammer(unsigned *ctr, unsigned n, unsigned *sink, int spin)
{
    unsigned acc = 0;
    if ((threadIdx.x & 63) == 0)
        for (unsigned i = 0; i < n; i++) {
            acc += atomicAdd(ctr, 1u);
            for (int s = 0; s < spin; s++) __builtin_amdgcn_s_sleep(16);
        }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main()
{
    unsigned *ctr, *sink;
    (void)hipMalloc(&ctr, 256);
    (void)hipMalloc(&sink, 4);
    (void)hipMemset(ctr, 0, 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int spin : {0, 4, 32}) {
        for (int wgs : {256, 1024, 4096}) {
            const unsigned n = 2000;
            hammer<<<wgs, 256>>>(ctr, 10, sink, 0);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hammer<<<wgs, 256>>>(ctr, n, sink, spin);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("spin %2d  %4d workgroups x 4 waves, %u tickets per wave: %.2f ms  = %.1f M atomics/s, %.2f us per ticket and wave\n", spin, wgs, n, ms,
                   (double)wgs * 4 * n / ms / 1e3, ms * 1e3 / n);
        }
    }
    return 0;
}
