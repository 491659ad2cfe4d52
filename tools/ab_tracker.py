#!/usr/bin/env python3
"""tools/ab_tracker.py [counts...] -- tracker frames/s (bench.py's tracker_fps leg, Python driver) with the library
RBS_LIB_PATH names (default: the in-tree one): A/B of tracker-side changes,
    for v in base variant; do RBS_LIB_PATH=$PWD/build_variants/$v.so python tools/ab_tracker.py 200 2000; done"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

counts = tuple(int(x) for x in sys.argv[1:]) or (200, 2000, 20000)
sys.argv = sys.argv[:1]
a = bench.parse()
om, cam, P, n_tri, nb = bench.build_scene(a)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
fps = bench.tracker_fps(om, cam, dev, counts=counts, precision="f64")
fps.pop("_native", None)
print(os.environ.get("RBS_LIB_PATH", "in-tree"), {k: (round(v["fps"], 1), round(v["fps_pipelined"], 1)) for k, v in fps.items()})
