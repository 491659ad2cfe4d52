import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R)
import numpy as np, torch
import bench
from dbot_ros_amd import CameraData, ObjectModel, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
v, t = synth.mesh_m1(); om = ObjectModel([v], [t], center=True); cam = CameraData(synth.camera_matrix(640, 480), 480, 640)
print(bench.tracker_fps(om, cam, torch.device('cuda', 0), counts=(n,), n_frames=20))
