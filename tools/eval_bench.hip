// tools/eval_bench.hip -- what ONE evaluation batch of the F64 Kinect likelihood (rbs_math.h: depth_term + pixel_loglik_f64,
// the tables in LDS as in the raster kernel) costs at the raster kernel's occupancy (3 blocks of 256 threads per CU), on its own:
// cycles per 64-pixel batch per wave, chip-wide rate.  Compare with the share of the raster kernel the stand-in builds charge to it.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I dbot_ros_amd/csrc -o build_variants/eval_bench tools/eval_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "rbs_math.h"

__global__ __launch_bounds__(256) void k(double* out, int iters, int gather, const double* __restrict__ aux, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    constexpr int ne = rbsm::kErfcIntervals * rbsm::kErfcCoefs, nl = rbsm::kLogIntervals * 2;
    for (int i = threadIdx.x; i < ne; i += 256) tab[i] = rbsm::kErfcTab[i];
    for (int i = threadIdx.x; i < nl; i += 256) tab[ne + i] = rbsm::kLogTab[i];
    __syncthreads();
    const rbsm::PixelConsts C = {0.6931471805599453, 0.01 / 6.0, 0.99 / 1.7724538509055159};
    const int lane = threadIdx.x & 63;
    double ll = 0.0;
    float post_acc = 0.f;
    unsigned s = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const float u1 = (float)(s >> 8) * (1.0f / 16777216.0f);
        s = s * 1664525u + 1013904223u;
        const float u2 = (float)(s >> 8) * (1.0f / 16777216.0f);
        const float r = 0.65f + 0.1f * u1;                 // rendered depth
        const float o = r + 0.02f * (u2 - 0.5f);           // observation near it
        const float prior = 0.1f + 0.5f * u1;
        double a0, a1, a2, a3;
        if (gather) {
            const double* a4 = aux + 4 * (size_t)((s >> 12) & 8191u);     // a 256 KB table: L2 resident
            a0 = a4[0]; a1 = a4[1]; a2 = a4[2]; a3 = a4[3];
        } else {
            const double sigma = 0.003 + 0.0014247 * (double)o * o;
            a0 = 0.7071067811865476 / sigma; a1 = C.lambda * sigma * 0.7071067811865476; a2 = o; a3 = 0.2;
        }
        const double g = rbsm::depth_term(C, (double)r);
        float post;
        ll += rbsm::pixel_loglik_f64(C, g, a0, a1, a2, a3, (double)r, prior, tab, tab + ne, post);
        post_acc += post;
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = ll + post_acc;
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
    (void)lane;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    double* out; double* aux; unsigned long long* cyc;
    hipMalloc(&out, 8 * 256 * 768 * 2); hipMalloc(&aux, 8 * 4 * 8192); hipMalloc(&cyc, 8);
    hipMemset(aux, 0, 8 * 4 * 8192);
    {   // plausible table content
        double* h = (double*)malloc(8 * 4 * 8192);
        for (int i = 0; i < 8192; ++i) { const double o = 0.6 + 0.2 * i / 8192.0, sg = 0.003 + 0.0014247 * o * o; h[4 * i] = 0.7071067811865476 / sg; h[4 * i + 1] = 0.6931471805599453 * sg * 0.7071067811865476; h[4 * i + 2] = o; h[4 * i + 3] = 0.2; }
        hipMemcpy(aux, h, 8 * 4 * 8192, hipMemcpyHostToDevice); free(h);
    }
    for (int gather = 0; gather < 2; ++gather)
        for (int bpc = 1; bpc <= 3; ++bpc) {
            const size_t lds = bpc == 1 ? 96 * 1024 : bpc == 2 ? 64 * 1024 : 52 * 1024;
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int blocks = 256 * bpc;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, 16, gather, aux, cyc);
            hipDeviceSynchronize();
            hipMemset(cyc, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, iters, gather, aux, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("gather=%d blocks/CU=%d: %.1f cycles per 64-pixel batch per wave (wave lifetime), %.3f ms, %.2f G pixel-evaluations/s chip-wide\n",
                   gather, bpc, (double)c / blocks / iters, ms, (double)blocks * 256 * iters / (ms * 1e-3) / 1e9);
        }
    return 0;
}
