import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
n = int(sys.argv[1])
sys.argv = sys.argv[:1]
a = bench.parse()
om, cam, P, n_tri, nb = bench.build_scene(a)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
fps = bench.tracker_fps(om, cam, dev, counts=(n,), precision="f64")
