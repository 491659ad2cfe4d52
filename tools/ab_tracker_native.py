#!/usr/bin/env python3
"""tools/ab_tracker_native.py [counts...] -- the device tracker driven from C++ (tests/cpp/host_bench --tracker, what
bench.py reports as tracker_fps_native_*): frames/s frame by frame and with one frame of look-ahead.  Environment
switches (RBS_TRACKER_BAND=0 ...) pass through: an A/B is two runs."""
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
fps = bench.tracker_fps(om, cam, dev, counts=(200,), precision="f64")     # (for the frames and the initial state)
frames, init = fps.pop("_native")
out = bench.native_tracker_leg(om, cam, P, frames, init, counts, "f64")
print({k: v for k, v in out.items() if not k.endswith("note")})
