// double -> float staging variants on the host (tools/dbg): g++ -O2 -o /tmp/cb convert_bench.cpp && /tmp/cb
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__attribute__((target("avx2"))) static void v_plain(float* d, const double* s, size_t n) {
    for (size_t p = 0; p + 8 <= n; p += 8) {
        const __m128 a = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p)), b = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p + 4));
        _mm256_storeu_ps(d + p, _mm256_set_m128(b, a));
    }
}
__attribute__((target("avx2"))) static void v_nt(float* d, const double* s, size_t n) {
    for (size_t p = 0; p + 8 <= n; p += 8) {
        const __m128 a = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p)), b = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p + 4));
        _mm256_stream_ps(d + p, _mm256_set_m128(b, a));
    }
    _mm_sfence();
}
__attribute__((target("avx2"))) static void v_pf(float* d, const double* s, size_t n) {
    for (size_t p = 0; p + 16 <= n; p += 16) {
        _mm_prefetch((const char*)(s + p + 256), _MM_HINT_NTA);
        _mm_prefetch((const char*)(s + p + 264), _MM_HINT_NTA);
        const __m128 a = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p)), b = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p + 4));
        const __m128 c = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p + 8)), e = _mm256_cvtpd_ps(_mm256_loadu_pd(s + p + 12));
        _mm256_stream_ps(d + p, _mm256_set_m128(b, a));
        _mm256_stream_ps(d + p + 8, _mm256_set_m128(e, c));
    }
    _mm_sfence();
}
__attribute__((target("avx512f"))) static void v_512(float* d, const double* s, size_t n) {
    for (size_t p = 0; p + 16 <= n; p += 16) {
        const __m256 a = _mm512_cvtpd_ps(_mm512_loadu_pd(s + p)), b = _mm512_cvtpd_ps(_mm512_loadu_pd(s + p + 8));
        _mm512_stream_ps(d + p, _mm512_castpd_ps(_mm512_insertf64x4(_mm512_castpd256_pd512(_mm256_castps_pd(a)), _mm256_castps_pd(b), 1)));
    }
    _mm_sfence();
}
int main(int argc, char** argv) {
    const size_t n = 640 * 480; const int F = argc > 1 ? atoi(argv[1]) : 30;
    std::vector<double*> src(F);
    for (auto& p : src) { p = (double*)aligned_alloc(64, n * 8); for (size_t i = 0; i < n; ++i) p[i] = 1.0 + (i % 97) * 0.01; }
    float* dst[2] = {(float*)aligned_alloc(64, n * 4), (float*)aligned_alloc(64, n * 4)};
    memset(dst[0], 0, n * 4); memset(dst[1], 0, n * 4);
    struct { const char* name; void (*f)(float*, const double*, size_t); } V[] = {{"plain", v_plain}, {"nt", v_nt}, {"nt+prefetch", v_pf},
        {"avx512 nt", __builtin_cpu_supports("avx512f") ? v_512 : nullptr}};
    for (int rep = 0; rep < 2; ++rep)
    for (auto& v : V) {
        if (!v.f) continue;
        const int it = 600;
        for (int i = 0; i < 30; ++i) v.f(dst[i & 1], src[i % F], n);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < it; ++i) v.f(dst[i & 1], src[i % F], n);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / it;
        printf("%-14s %d frames cycling: %.1f us per 640x480 frame (%.1f GB/s read)\n", v.name, F, us, n * 8 / us / 1e3);
    }
    return 0;
}
