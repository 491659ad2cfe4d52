// tools/valu_bench.hip -- issue cost of wave64 VALU instructions on gfx950, per SIMD: how many shader
// cycles one wave64 instruction of each kind holds the SIMD's VALU for, with 1, 2 and 3 waves per SIMD
// (the raster kernel's occupancy).  Answers "is a float32 instruction cheaper than a binary64 one here".
// build: hipcc -O3 --offload-arch=gfx950 -o build_variants/valu_bench tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 0.999f, e = 1e-3f;
    const double cd = 0.999, ed = 1e-3;
    const float2_ pc = {0.999f, 0.998f}, pe = {1e-3f, 2e-3f};
    const unsigned long long t0 = clock64();
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
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc)
{
    const int iters = 200;
    for (int wps = 1; wps <= 3; ++wps) {
        // wps blocks of 256 threads per CU: LDS sized so that exactly wps blocks fit a CU
        const size_t lds = wps == 1 ? 96 * 1024 : wps == 2 ? 64 * 1024 : 48 * 1024;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), lds, 0, out, 2, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), lds, 0, out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c = 0;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n_inst = (double)iters * 16 * 8;   // per wave
        // shader cycles one wave spends per instruction / waves on the SIMD = cycles the SIMD's VALU is held per instruction
        printf("%-14s waves/SIMD=%d  clock64 cycles per instr per wave = %6.2f  -> per SIMD %5.2f   wall: %.3f ms -> %.2f G wave-instr/s chip\n",
               name, wps, c / n_inst, c / n_inst / wps, ms, n_inst * wps * 1024 / (ms * 1e-3) / 1e9);
    }
}

int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4 * 256 * 256 * 3); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", out, cyc); run<5>("v_add_f32", out, cyc); run<10>("v_add_u32", out, cyc); run<2>("v_pk_fma_f32", out, cyc);
    run<1>("v_fma_f64", out, cyc); run<3>("v_add_f64", out, cyc); run<4>("v_mul_f64", out, cyc); run<8>("v_min_f64", out, cyc);
    run<9>("v_cvt_f32_f64", out, cyc); run<6>("v_rcp_f64", out, cyc); run<7>("v_rcp_f32", out, cyc); run<11>("cmp_f64+cndmask", out, cyc);
    return 0;
}
