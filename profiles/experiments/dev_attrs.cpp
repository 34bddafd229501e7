// prints the device attributes the engine's launch-shape decisions rest on (hipcc -o ab_libs/dev_attrs profiles/experiments/dev_attrs.cpp)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t *o) { extern __shared__ uint32_t s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[0] = s[63 - threadIdx.x]; }
int main()
{
    int v;
#define A(x) if (hipDeviceGetAttribute(&v, x, 0) == hipSuccess) std::printf("%-55s %d\n", #x, v); else std::printf("%-55s (query failed)\n", #x);
    A(hipDeviceAttributeMultiprocessorCount)
    A(hipDeviceAttributeMaxSharedMemoryPerBlock)
    A(hipDeviceAttributeSharedMemPerBlockOptin)
    A(hipDeviceAttributeMaxSharedMemoryPerMultiprocessor)
    A(hipDeviceAttributeMaxThreadsPerMultiProcessor)
    A(hipDeviceAttributeClockRate)
    A(hipDeviceAttributeMaxRegistersPerBlock)
    uint32_t *d;
    (void)hipMalloc(&d, 4);
    for (size_t lds : {64u * 1024u, 96u * 1024u, 160u * 1024u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), lds, 0, d);
        hipError_t e = hipGetLastError();
        hipError_t e2 = hipDeviceSynchronize();
        std::printf("launch with %zu bytes of dynamic LDS, no attribute set: %s / %s\n", lds, hipGetErrorString(e), hipGetErrorString(e2));
    }
    return 0;
}
