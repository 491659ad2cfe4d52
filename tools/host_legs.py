#!/usr/bin/env python3
"""tools/host_legs.py [bench args]: only the C++ host legs of bench.py (tests/cpp/host_bench: host-pointer step, with
look-ahead, and through the plugin surface) -- seconds instead of the whole bench.  Prints the leg's keys as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

a = bench.parse()
om, cam, P, n_tri, nb = bench.build_scene(a)
dev = torch.device("cuda", 0)
W = bench.Workload(a, om, cam, P, nb, dev, 0)
print(json.dumps(bench.native_host_leg(a, om, cam, P, W, 400), indent=1))
