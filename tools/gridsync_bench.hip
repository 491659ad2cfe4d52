// How long does a grid-wide barrier take on this chip?  768 blocks x 256 threads (the raster
// kernel's persistent grid), cooperative launch, N syncs.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(256) void k(int n, unsigned long long* out)
{
    extern __shared__ unsigned char smem[];
    cg::grid_group g = cg::this_grid();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) g.sync();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = wall_clock64() - t0;
    if (smem[threadIdx.x] == 77 && n < 0) out[1] = 1;
}
int main()
{
    unsigned long long* d;
    hipMalloc(&d, 16);
    int n = 200;
    for (int lds : {0, 53296}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int per = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, 256, lds);
        void* args[] = {&n, &d};
        for (int blocks : {256, 512, 768}) {
            hipError_t e = hipLaunchCooperativeKernel((const void*)k, dim3(blocks), dim3(256), args, lds, 0);
            hipDeviceSynchronize();
            unsigned long long t = 0;
            hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
            printf("lds %d occupancy %d/CU blocks %d: launch %s, %.2f us per grid sync\n", lds, per, blocks,
                   hipGetErrorString(e), t / 100.0 / n);
            fflush(stdout);
        }
    }
    return 0;
}
