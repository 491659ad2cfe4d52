// Diagnostic: accuracy of v_rcp_f64, of one Newton step on it, and of q1 = a * r1 against the correctly rounded a / b.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n_per)
{
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    double e0 = 0, e1 = 0, eq = 0, eq0 = 0;
    for (int i = 0; i < n_per; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double b = __longlong_as_double((long long)((s & 0x000fffffffffffffull) | ((1023ull - 20 + (s >> 58)) << 52)));   // 2^-20 .. 2^43
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double a = __longlong_as_double((long long)((s & 0x000fffffffffffffull) | ((1023ull - 10 + (s >> 59)) << 52)));
        const double r0 = __builtin_amdgcn_rcp(b);
        const double r1 = __builtin_fma(r0, __builtin_fma(-b, r0, 1.0), r0);
        const double ex = 1.0 / b, q = a / b;
        e0 = fmax(e0, fabs(r0 - ex) / ex);
        e1 = fmax(e1, fabs(r1 - ex) / ex);
        eq = fmax(eq, fabs(a * r1 - q) / q);
        eq0 = fmax(eq0, fabs(a * r0 - q) / q);
    }
    out[4 * (blockIdx.x * 256 + threadIdx.x) + 0] = e0; out[4 * (blockIdx.x * 256 + threadIdx.x) + 1] = e1;
    out[4 * (blockIdx.x * 256 + threadIdx.x) + 2] = eq; out[4 * (blockIdx.x * 256 + threadIdx.x) + 3] = eq0;
}
int main()
{
    const int nb = 1024, n = nb * 256;
    double* d; hipMalloc(&d, 32 * n);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 4000);
    double* h = new double[4 * n]; hipMemcpy(h, d, 32 * n, hipMemcpyDeviceToHost);
    double m[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 4; ++j) m[j] = fmax(m[j], h[4 * i + j]);
    printf("1e9 samples: max rel err rcp %.3e (2^%.1f), rcp + 1 Newton %.3e (2^%.1f), a * r1 vs a / b %.3e (2^%.1f), a * r0 %.3e (2^%.1f)\n",
           m[0], log2(m[0]), m[1], log2(m[1]), m[2], log2(m[2]), m[3], log2(m[3]));
    return 0;
}
