import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import scenarios as sc
from dbot_ros_amd import RbSensor
for world, n in ((8, 2000), (8, 25000), (8, 6250), (2, 2000)):
    N = world * n
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=2 * n)
    with RbSensor(om, cam, P, max_particles=2 * n) as s:
        rng = np.random.default_rng(1)
        for spread in (2.0, 200.0):
            ll = torch.from_numpy(rng.normal(-3000, spread, N)).cuda()
            u = torch.sort(torch.rand(N, dtype=torch.float64)).values.cuda()
            outs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4)]
            cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
            stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); st = stream.cuda_stream   # (a null stream selects the handle's own)
            for _ in range(5):
                s.peer_resample(ll.data_ptr(), u.data_ptr(), N, n, 3 % world, 2, 1.0, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), cnt.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                s.peer_resample(ll.data_ptr(), u.data_ptr(), N, n, 3 % world, 2, 1.0, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), cnt.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            print(f"world {world} n {n} N {N} spread {spread}: rbs_peer_resample {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch")
