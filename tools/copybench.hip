// Micro-benchmark: how fast can the parent->child plane stream go on this box?
// hipcc --offload-arch=gfx950 -O3 tools/copybench.hip -o build_variants/copybench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int U, bool NT>
__global__ __launch_bounds__(256) void flat_copy(const floatx4* __restrict__ s, floatx4* __restrict__ d, size_t n4, float a, float b)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
        floatx4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { size_t i = base + (size_t)k * 256; if (i < n4) v[k] = NT ? __builtin_nontemporal_load(&s[i]) : s[i]; }
#pragma unroll
        for (int k = 0; k < U; ++k) { size_t i = base + (size_t)k * 256; if (i < n4) { floatx4 w = v[k] * a + b; if (NT) __builtin_nontemporal_store(w, &d[i]); else d[i] = w; } }
    }
}

// one block per (particle, band), permuted parent
template<int U, bool NT>
__global__ __launch_bounds__(256) void band_copy(const floatx4* __restrict__ s, floatx4* __restrict__ d, const int* __restrict__ parent,
                                                  int plane4, int band4, int bands, int N, float a, float b)
{
    for (int w = blockIdx.x; w < N * bands; w += gridDim.x) {
        const int p = w / bands, band = w - p * bands;
        const floatx4* sp = s + (size_t)parent[p] * plane4 + (size_t)band * band4;
        floatx4* dp = d + (size_t)p * plane4 + (size_t)band * band4;
        const int n4 = min(band4, plane4 - band * band4);
        for (int base = threadIdx.x; base < n4; base += 256 * U) {
            floatx4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { int i = base + k * 256; if (i < n4) v[k] = NT ? __builtin_nontemporal_load(&sp[i]) : sp[i]; }
#pragma unroll
            for (int k = 0; k < U; ++k) { int i = base + k * 256; if (i < n4) { floatx4 x = v[k] * a + b; if (NT) __builtin_nontemporal_store(x, &dp[i]); else dp[i] = x; } }
        }
    }
}

int main(int argc, char** argv)
{
    const int N = 2000, plane = 640 * 480, plane4 = plane / 4;
    float *s, *d; int* par;
    CK(hipMalloc(&s, (size_t)N * plane * 4)); CK(hipMalloc(&d, (size_t)N * plane * 4)); CK(hipMalloc(&par, N * 4));
    CK(hipMemset(s, 0, (size_t)N * plane * 4));
    std::vector<int> h(N); std::iota(h.begin(), h.end(), 0); std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g);
    CK(hipMemcpy(par, h.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * N * plane * 4;
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%-40s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout);
    };
    const size_t n4 = (size_t)N * plane4;
    if (argc > 1) {  // quick: one known-byte-count kernel for PMC calibration
        time("flat U8 nt   blocks=4096", [&] { hipLaunchKernelGGL((flat_copy<8, true>), dim3(4096), dim3(256), 0, 0, (floatx4*)s, (floatx4*)d, n4, 0.98f, 0.004f); });
        return 0;
    }
    time("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(d, s, (size_t)N * plane * 4, hipMemcpyDeviceToDevice, 0)); });
    for (int blocks : {1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "flat U4 nt   blocks=%d", blocks); time(nm, [&] { hipLaunchKernelGGL((flat_copy<4, true>), dim3(blocks), dim3(256), 0, 0, (floatx4*)s, (floatx4*)d, n4, 0.98f, 0.004f); });
        snprintf(nm, 64, "flat U8 nt   blocks=%d", blocks); time(nm, [&] { hipLaunchKernelGGL((flat_copy<8, true>), dim3(blocks), dim3(256), 0, 0, (floatx4*)s, (floatx4*)d, n4, 0.98f, 0.004f); });
        snprintf(nm, 64, "flat U8 plain blocks=%d", blocks); time(nm, [&] { hipLaunchKernelGGL((flat_copy<8, false>), dim3(blocks), dim3(256), 0, 0, (floatx4*)s, (floatx4*)d, n4, 0.98f, 0.004f); });
    }
    for (int bands : {20, 40, 10, 5}) {
        const int band4 = (plane4 + bands - 1) / bands;
        for (int blocks : {0, 1024, 2048, 4096}) {
            char nm[64];
            const int gx = blocks ? blocks : N * bands;
            snprintf(nm, 64, "band U8 nt bands=%d blocks=%d", bands, gx); time(nm, [&] { hipLaunchKernelGGL((band_copy<8, true>), dim3(gx), dim3(256), 0, 0, (floatx4*)s, (floatx4*)d, par, plane4, band4, bands, N, 0.98f, 0.004f); });
            snprintf(nm, 64, "band U4 nt bands=%d blocks=%d", bands, gx); time(nm, [&] { hipLaunchKernelGGL((band_copy<4, true>), dim3(gx), dim3(256), 0, 0, (floatx4*)s, (floatx4*)d, par, plane4, band4, bands, N, 0.98f, 0.004f); });
            snprintf(nm, 64, "band U16 nt bands=%d blocks=%d", bands, gx); time(nm, [&] { hipLaunchKernelGGL((band_copy<16, true>), dim3(gx), dim3(256), 0, 0, (floatx4*)s, (floatx4*)d, par, plane4, band4, bands, N, 0.98f, 0.004f); });
        }
    }
    return 0;
}
