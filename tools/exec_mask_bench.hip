// tools/exec_mask_bench.hip -- does a wave64 VALU instruction cost less when part of the wave is masked off?
// The per-lane sample loops of the raster kernel run with a third to a half of their lanes active; every attempt
// to run them with fuller lanes (decoupled coverage / depth, compacted samples, row spans: DESIGN.md section 4)
// was slower although it issued fewer instructions.  This measures the issue time of v_fma_f64 / v_fma_f32 under
// EXEC masks: lanes [0, a) active (whole 16-lane rows masked off when a <= 48, 32, 16), and every k-th lane active
// (no row is empty).
// build: hipcc -O3 --offload-arch=gfx950 -o build_variants/exec_mask_bench tools/exec_mask_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int F64>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long mask, unsigned long long* clk)
{
    const unsigned long long w0 = wall_clock64(), t0 = clock64();
    const int lane = threadIdx.x & 63;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float c = 0.999f, e = 1e-3f;
    const double cd = 0.999, ed = 1e-3;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            if (F64) { REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cd), "v"(ed));) }
            else { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(e));) }
        }
    }
    if (blockIdx.x == 5 && threadIdx.x == 0) { clk[0] = clock64() - t0; clk[1] = wall_clock64() - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int F64>
static void run(const char* name, float* out, int iters, unsigned long long mask, const char* what, int per_cu = 3)
{
    const int blocks = 256 * per_cu;     // (three 4-wave blocks per CU: the raster kernel's occupancy)
    static unsigned long long* clk = nullptr;
    if (!clk) hipHostMalloc(&clk, 16);
    hipLaunchKernelGGL(k<F64>, dim3(blocks), dim3(256), 0, 0, out, 8, mask, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<F64>, dim3(blocks), dim3(256), 0, 0, out, iters, mask, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * 128.0;
    hipDeviceSynchronize();
    const double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
    printf("%-10s %-28s active lanes %2d  %8.3f ms  %7.1f G wave-instructions/s   block 5: %.2f GHz shader clock, %.2f cycles per instruction of one wave\n", name, what,
           __builtin_popcountll(mask), ms, inst / (ms * 1e-3) / 1e9, ghz, (double)clk[0] / (iters * 128.0));
}

int main()
{
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
    struct { unsigned long long m; const char* what; } cases[] = {
        {~0ull, "all 64"}, {(1ull << 48) - 1, "lanes 0-47"}, {(1ull << 32) - 1, "lanes 0-31"}, {(1ull << 17) - 1, "lanes 0-16"},
        {(1ull << 16) - 1, "lanes 0-15"}, {0xffull, "lanes 0-7"}, {1ull, "lane 0"},
        {0x5555555555555555ull, "every 2nd lane"}, {0x1111111111111111ull, "every 4th lane"}, {0x0001000100010001ull, "every 16th lane"},
        {0x00000000ffff0000ull, "lanes 16-31"}, {0xffff00000000ffffull, "lanes 0-15 and 48-63"},
        {0x1ffull, "lanes 0-8"}, {0x3ffull, "lanes 0-9"}, {0xfffull, "lanes 0-11"}, {0x7fffull, "lanes 0-14"},
        {0x0101010101010101ull, "every 8th lane (8)"}, {0x0101010101010103ull, "every 8th lane + lane 1 (9)"},
        {0x0000000100000001ull, "lanes 0 and 32"}, {0x8000000000000001ull, "lanes 0 and 63"},
    };
    // is the slow regime a limit of the WAVE (then more waves per SIMD hide it) or of the pipe?
    for (int w = 1; w <= 8; ++w) {
        char what[64];
        snprintf(what, sizeof what, "lane 0, %d waves per SIMD", w);
        run<1>("v_fma_f64", out, 4000, 1ull, what, w);
        snprintf(what, sizeof what, "all 64, %d waves per SIMD", w);
        run<1>("v_fma_f64", out, 4000, ~0ull, what, w);
    }
    for (auto& cse : cases) run<1>("v_fma_f64", out, 20000, cse.m, cse.what);
    for (auto& cse : cases) run<0>("v_fma_f32", out, 40000, cse.m, cse.what);
    hipFree(out);
    return 0;
}
