// Diagnostic: hipIpcGetMemHandle in a child process, hipIpcOpenMemHandle in the parent, by buffer size,
// with both processes holding the same allocations (two big buffers + small ones, streams, pinned host memory).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>
static void* big[2];
static void setup(size_t bytes, int mode)
{
    hipSetDevice(0);
    if (mode & 1) { hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
    if (mode & 2) { void* h; hipHostMalloc(&h, 8 << 20, hipHostMallocMapped); }
    if (mode & 4) for (int k = 0; k < 12; ++k) { void* p; hipMalloc(&p, (size_t)(1 + k) << 18); }
    hipMalloc(&big[0], bytes); hipMalloc(&big[1], bytes);
    if (mode & 8) { hipMemset(big[0], 0, 1 << 20); hipDeviceSynchronize(); }
}
int main(int argc, char** argv)
{
    const size_t bytes = (size_t)atof(argv[1]);
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    int to_parent[2], to_child[2];
    pipe(to_parent); pipe(to_child);
    if (fork() == 0) {
        setup(bytes, mode);
        hipIpcMemHandle_t h;
        hipError_t e = hipIpcGetMemHandle(&h, big[0]);
        printf("child: get -> %d\n", (int)e); fflush(stdout);
        write(to_parent[1], &h, sizeof(h));
        char c; read(to_child[0], &c, 1);
        return 0;
    }
    hipIpcMemHandle_t h;
    read(to_parent[0], &h, sizeof(h));
    setup(bytes, mode);
    void* q = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    auto t1 = std::chrono::steady_clock::now();
    printf("parent: open %zu B mode %d -> %d in %.3f s (mine %p, mapped %p)\n", bytes, mode, (int)e, std::chrono::duration<double>(t1 - t0).count(), big[0], q); fflush(stdout);
    char c = 1; write(to_child[1], &c, 1);
    wait(nullptr);
    return 0;
}
