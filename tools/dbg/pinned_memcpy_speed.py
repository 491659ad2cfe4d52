"""Host memcpy rate into pinned memory allocated in different ways (run on the GPU box)."""
import ctypes as C, numpy as np, time
hip = C.CDLL("libamdhip64.so")
N = 307200
F = np.random.rand(30, N).astype(np.float32)
def rate(dst, label):
    for r in range(2):
        t0 = time.perf_counter()
        for i in range(600): np.copyto(dst, F[i % 30])
        dt = (time.perf_counter() - t0) / 600
    print("%-44s %.1f us  %.1f GB/s" % (label, dt * 1e6, N * 4 / dt / 1e9))
rate(np.empty(N, np.float32), "pageable")
for label, flags in (("hipHostMalloc default", 0), ("portable", 1), ("numa-user", 0x20000000), ("non-coherent", 0x80000000),
                     ("coherent", 0x40000000), ("numa-user|non-coherent", 0xA0000000), ("write-combined", 4)):
    p = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(N * 4), C.c_uint(flags))
    if rc != 0: print(label, "rc", rc); continue
    dst = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(N,))
    rate(dst, "hipHostMalloc " + label)
a = np.empty(N + 4096, np.float32)
rc = hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.c_uint(0))
print("hipHostRegister rc", rc)
rate(a[:N], "malloc + hipHostRegister")
