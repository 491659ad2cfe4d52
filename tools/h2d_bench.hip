// tools/h2d_bench.hip: how long does a 1.2 MB frame take from pinned host memory to the device --
// hipMemcpyAsync (SDMA) against a kernel that reads the mapped pinned buffer over PCIe?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void pull(const floatx4* __restrict__ src, floatx4* __restrict__ dst, int n4)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) dst[i] = __builtin_nontemporal_load(src + i);
}
int main()
{
    const int n = 640 * 480, n4 = n / 4;
    float *h, *d, *hd;
    CK(hipHostMalloc(&h, n * 4, hipHostMallocMapped));
    CK(hipMalloc(&d, n * 4));
    CK(hipHostGetDevicePointer((void**)&hd, h, 0));
    memset(h, 1, n * 4);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int mode = 0; mode < 6; ++mode) {
        const int blocks = mode == 0 ? 0 : (16 << (mode - 1));   // 16..256 blocks of 256 threads
        double best = 1e9, sum = 0;
        for (int it = 0; it < 60; ++it) {
            h[it] = (float)it;
            CK(hipStreamSynchronize(s));
            const auto t0 = now();
            if (mode == 0) CK(hipMemcpyAsync(d, h, n * 4, hipMemcpyHostToDevice, s));
            else hipLaunchKernelGGL(pull, dim3(blocks), dim3(256), 0, s, (const floatx4*)hd, (floatx4*)d, n4);
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(now() - t0).count();
            if (it >= 10) { sum += us; if (us < best) best = us; }
        }
        if (mode == 0) printf("hipMemcpyAsync H2D:           mean %.1f us, best %.1f us (host clock, launch + sync included)\n", sum / 50, best);
        else printf("kernel pull, %3d blocks x 256: mean %.1f us, best %.1f us\n", blocks, sum / 50, best);
    }
    return 0;
}
