import numpy as np, time
F=np.random.rand(30,307200).astype(np.float32); dst=np.empty(307200,np.float32)
for name,idx in (("cold (30 frames cycling)",lambda i:i%30),("hot (same frame)",lambda i:0)):
    for r in range(2):
        t0=time.perf_counter()
        for i in range(600): np.copyto(dst,F[idx(i)])
        dt=(time.perf_counter()-t0)/600
    print(name,"%.1f us  %.1f GB/s"%(dt*1e6,1.2288e6/dt/1e9))
