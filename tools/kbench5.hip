// How many workgroups with a large dynamic LDS allocation does a CU really run at once?  Each workgroup spins for a fixed number
// of clocks; with 256 CUs a grid of 512 such workgroups takes as long as one of 256 only if two of them share a CU.
// hipcc --offload-arch=gfx950 -O3 tools/kbench5.hip -o tools/kbench5.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 4) void spin(unsigned long long clocks, unsigned* sink) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < clocks) acc += lds[(threadIdx.x * 7 + acc) & 511];
    if (acc == 0xdeadbeef) sink[0] = acc;
}
int main() {
    unsigned* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {32768, 65536, 66560, 73728, 79360, 81920}) {
        hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int occ = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)spin, 512, lds);
        for (int wgs : {256, 512, 768, 1024}) {
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(512), lds, 0, 200000ull, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(512), lds, 0, 2000000ull, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("lds=%d occupancy_api=%d wgs=%d ms=%.3f\n", lds, occ, wgs, ms);
        }
    }
    return 0;
}
