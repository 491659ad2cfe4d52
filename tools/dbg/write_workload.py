#!/usr/bin/env python3
"""tools/dbg/write_workload.py <out.bin> [bench args]: the bench's workload in tests/cpp/host_bench's file format."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = sys.argv.pop(1)
import torch  # noqa: E402
import bench  # noqa: E402

a = bench.parse()
om, cam, P, n_tri, nb = bench.build_scene(a)
W = bench.Workload(a, om, cam, P, nb, torch.device("cuda", 0), 0)
with open(out, "wb") as f:
    bench.write_host_workload(f, om, cam, P, W.frames, W.poses, W.parents, bool(a.update))
print("wrote", out)
