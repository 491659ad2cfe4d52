// tools/valu_bench.hip -- issue rate of wave64 VALU instructions on gfx950 (MI355X), chip-wide, by
// instruction class and by waves per SIMD (1..8): the ceilings bench.py prices the raster kernel's
// instruction mix against (roofline.peak) and the check that they are what the chip does.
//
// Round 3's version timed 60-180 us kernels and trusted block 0's cycle counter; its cycle-derived and
// wall-derived rates disagreed by 2x at 3 waves per SIMD.  This version
//   * runs every point for >= 5 ms (iterations scaled per class), timed with HIP events;
//   * has EVERY block record its start / end on the 100 MHz wall clock and its shader-cycle count:
//     co-residency is verified (the latest start precedes the earliest end: all blocks ran side by
//     side, `cores`) and the rate is also computed over that common interval only;
//   * derives the shader clock from the two counters (cycles / wall ticks x 100 MHz), so the
//     cycle-derived rate  n_SIMD x clock / (cycles per instruction per SIMD)  and the wall-derived
//     rate  instructions / time  can be compared directly (`agree` = their ratio);
//   * places exactly w blocks of 256 threads (= w waves per SIMD) on a CU through the LDS size, for
//     w = 1..8.
// tools/valu_bench.sh runs it once plainly and once under rocprofv3 --pmc (SQ_WAVES, SQ_BUSY_CYCLES,
// GRBM_GUI_ACTIVE, SQ_INSTS_VALU) and puts both under profiles/.
// build: hipcc -O3 --offload-arch=gfx950 -o build_variants/valu_bench tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

struct BlockLog { unsigned long long w0, w1, cyc; };

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, BlockLog* log)
{
    extern __shared__ unsigned char lds_[];
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 0.999f, e = 1e-3f;
    const double cd = 0.999, ed = 1e-3;
    const float2_ pc = {0.999f, 0.998f}, pe = {1e-3f, 2e-3f};
    const unsigned long long w0 = wall_clock64(), t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(e));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cd), "v"(ed));) }
        if (KIND == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pe));) }
        if (KIND == 3) { REP16(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(ed));) }
        if (KIND == 4) { REP16(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cd));) }
        if (KIND == 5) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));) }
        if (KIND == 6) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
        if (KIND == 7) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 8) { REP16(asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cd));) }
        if (KIND == 9) { REP16(asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));) }
        if (KIND == 10) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));) }
        if (KIND == 11) { REP16(asm volatile("v_cmp_gt_f64 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_gt_f64 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_gt_f64 vcc, %6, %8\n v_cndmask_b32 %7, %7, %2, vcc\n v_cmp_gt_f64 vcc, %0, %8\n v_cndmask_b32 %1, %1, %5, vcc" : "+v"(d0), "+v"(a1), "+v"(a2), "+v"(d3), "+v"(a4), "+v"(a5), "+v"(d6), "+v"(a7) : "v"(cd) : "vcc");) }
        // the raster kernel's own blend: 4 binary64 fma/mul/add + 4 float32/integer per 8 (its measured mix is 45 % binary64)
        if (KIND == 12) { REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f32 %4, %4, %10, %11\n v_mul_f64 %1, %1, %8\n v_add_u32 %5, %5, %11\n v_add_f64 %2, %2, %9\n v_fma_f32 %6, %6, %10, %11\n v_fma_f64 %3, %3, %8, %9\n v_add_f32 %7, %7, %11" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cd), "v"(ed), "v"(c), "v"(e));) }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)lds_[0];
    if (threadIdx.x == 0) { log[blockIdx.x].w0 = w0; log[blockIdx.x].w1 = w1; log[blockIdx.x].cyc = t1 - t0; }
}

static int g_cus = 256;

template <int KIND>
void run(const char* name, float* out, BlockLog* dlog, int iters, const char* only)
{
    if (only && strcmp(only, name) != 0) return;
    static const int lds_kb[9] = {0, 96, 64, 48, 40, 32, 26, 22, 20};   // exactly w blocks of this size fit a CU's 160 KB
    for (int wps = 1; wps <= 8; ++wps) {
        const size_t lds = (size_t)lds_kb[wps] * 1024;
        const int blocks = g_cus * wps;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, out, 8, dlog);   // warm-up
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, out, iters, dlog);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<BlockLog> log(blocks);
        hipMemcpy(log.data(), dlog, sizeof(BlockLog) * blocks, hipMemcpyDeviceToHost);
        unsigned long long first = ~0ull, last_start = 0, first_end = ~0ull, last = 0;
        double cyc = 0, ticks = 0;
        for (const BlockLog& b : log) {
            first = std::min(first, b.w0); last_start = std::max(last_start, b.w0);
            first_end = std::min(first_end, b.w1); last = std::max(last, b.w1);
            cyc += (double)b.cyc; ticks += (double)(b.w1 - b.w0);
        }
        const double n_wave = (double)iters * 16 * 8;              // instructions per wave
        const double n_all = n_wave * blocks * 4;                  // ... in the launch (4 waves per block)
        const bool cores = last_start < first_end;                 // every block was running while every other was
        const double clock_ghz = cyc / ticks * 0.1;                // shader cycles per 100 MHz tick
        const double cyc_per_instr_wave = cyc / blocks / n_wave, cyc_per_instr_simd = cyc_per_instr_wave / wps;
        const double rate_cycles = g_cus * 4.0 * clock_ghz / cyc_per_instr_simd;           // G wave-instr/s
        const double rate_wall = n_all / (ms * 1e-3) / 1e9;
        const double rate_span = n_all / ((double)(last - first) * 1e-8) / 1e9;            // first start .. last end on the device
        printf("%-14s waves/SIMD=%d  coresident=%s  clock %.3f GHz  cycles/instr: per wave %6.2f per SIMD %5.2f  ->  %7.1f G/s by cycles | %7.1f G/s by wall (%.2f ms) | %7.1f by device span   agree %.3f\n",
               name, wps, cores ? "yes" : "NO ", clock_ghz, cyc_per_instr_wave, cyc_per_instr_simd, rate_cycles, rate_wall, ms, rate_span,
               rate_cycles / rate_wall);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main(int argc, char** argv)
{
    const char* only = argc > 1 ? argv[1] : nullptr;
    const int it = argc > 2 ? atoi(argv[2]) : 24000;               // iterations per point: >= 5 ms at one wave per SIMD (rocprofv3 passes use fewer)
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    g_cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, wall_clock64 = 100 MHz; rates in G wave64-instructions/s chip-wide\n", prop.gcnArchName, g_cus);
    float* out; BlockLog* dlog;
    hipMalloc(&out, 4 * 256 * (size_t)g_cus * 8); hipMalloc(&dlog, sizeof(BlockLog) * (size_t)g_cus * 8);
    run<0>("v_fma_f32", out, dlog, it, only); run<5>("v_add_f32", out, dlog, it, only); run<10>("v_add_u32", out, dlog, it, only);
    run<2>("v_pk_fma_f32", out, dlog, it, only);
    run<1>("v_fma_f64", out, dlog, it, only); run<3>("v_add_f64", out, dlog, it, only); run<4>("v_mul_f64", out, dlog, it, only);
    run<8>("v_min_f64", out, dlog, it, only); run<9>("v_cvt_f32_f64", out, dlog, it, only);
    run<6>("v_rcp_f64", out, dlog, it / 3, only); run<7>("v_rcp_f32", out, dlog, it / 2, only);
    run<11>("cmp_f64+cndmask", out, dlog, it, only); run<12>("mix_f64_f32", out, dlog, it, only);
    return 0;
}
