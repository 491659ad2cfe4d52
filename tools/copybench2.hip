// Sweep of stream-copy shapes to find this box's practical HBM ceiling for a 2.46 GB -> 2.46 GB copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// LD: 0 plain, 1 nt ; ST: 0 plain, 1 nt
template<int TPB, int U, int LD, int ST>
__global__ __launch_bounds__(TPB) void k_copy(const floatx4* __restrict__ s, floatx4* __restrict__ d, size_t n4, size_t chunk4)
{
    // each block owns contiguous chunks of chunk4 float4; grid-stride over chunks
    const size_t nchunks = (n4 + chunk4 - 1) / chunk4;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t lo = c * chunk4, hi = (lo + chunk4 < n4) ? lo + chunk4 : n4;
        for (size_t base = lo + threadIdx.x; base < hi; base += (size_t)TPB * U) {
            floatx4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { size_t i = base + (size_t)k * TPB; if (i < hi) v[k] = LD ? __builtin_nontemporal_load(&s[i]) : s[i]; }
#pragma unroll
            for (int k = 0; k < U; ++k) { size_t i = base + (size_t)k * TPB; if (i < hi) { floatx4 w = v[k] * 0.98f + 0.004f; if (ST) __builtin_nontemporal_store(w, &d[i]); else d[i] = w; } }
        }
    }
}

int main()
{
    const size_t bytes1 = (size_t)2000 * 640 * 480 * 4, n4 = bytes1 / 16;
    float *s, *d; CK(hipMalloc(&s, bytes1)); CK(hipMalloc(&d, bytes1)); CK(hipMemset(s, 0, bytes1));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 8; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
        printf("%-52s %.4f ms  %.0f GB/s\n", name, ms, 2.0 * bytes1 / ms / 1e6); fflush(stdout);
    };
    char nm[96];
#define RUN(TPB,U,LD,ST,GRID,CHUNK) snprintf(nm,96,"tpb=%d U=%d ld=%d st=%d grid=%d chunkKB=%d",TPB,U,LD,ST,(int)(GRID),(int)((CHUNK)*16/1024)); \
    time(nm,[&]{ hipLaunchKernelGGL((k_copy<TPB,U,LD,ST>), dim3(GRID), dim3(TPB), 0, 0, (floatx4*)s,(floatx4*)d,n4,(size_t)(CHUNK)); });
    RUN(256,1,1,1,(n4+255)/256,256)
    RUN(256,1,1,1,8192,256)
    RUN(256,1,1,1,16384,256)
    RUN(256,1,1,1,65536,256)
    RUN(256,2,1,1,(n4+511)/512,512)
    RUN(256,2,1,1,16384,512)
    RUN(256,4,1,1,(n4+1023)/1024,1024)
    RUN(256,4,1,1,16384,1024)
    RUN(256,8,1,1,(n4+2047)/2048,2048)
    RUN(256,8,1,1,16384,2048)
    RUN(512,1,1,1,(n4+511)/512,512)
    RUN(1024,1,1,1,(n4+1023)/1024,1024)
    RUN(128,1,1,1,(n4+127)/128,128)
    RUN(64,1,1,1,(n4+63)/64,64)
    RUN(64,4,1,1,(n4+255)/256,256)
    RUN(256,1,0,0,(n4+255)/256,256)
    RUN(256,1,0,1,(n4+255)/256,256)
    RUN(256,1,1,0,(n4+255)/256,256)
    return 0;
}
