#!/usr/bin/env python3
"""Record a synthetic tracking dataset in dbot_ros's on-disk format (measurements.bag +
ground_truth.txt) and replay it through the tracker: BASELINE config C1's
"recorded bag" path without ROS.
usage: python tools/replay_dataset.py [dir=/tmp/rbs_dataset] [frames=60] [particles=2000] [downsampling=1]"""
import os
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, dataset as ds, node, objloader, pose, synth  # noqa: E402

root = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rbs_dataset"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
f = int(sys.argv[4]) if len(sys.argv) > 4 else 1
shutil.rmtree(root, ignore_errors=True)
os.makedirs(os.path.join(root, "object_models"))
v, t = synth.mesh_m1()
objloader.write_obj(os.path.join(root, "object_models", "m1.obj"), v, t)
params = {
    "particle_filter": {"use_gpu": True, "cpu": {"sample_count": n}, "gpu": {"sample_count": n},
                        "moving_average_update_rate": 1.0, "center_object_frame": True, "max_kl_divergence": 2.0,
                        "observation": {"occlusion": {"initial_occlusion_prob": 0.1, "p_occluded_visible": 0.1,
                                                      "p_occluded_occluded": 0.7},
                                        "kinect": {"tail_weight": 0.01, "model_sigma": 0.003, "sigma_factor": 0.0014247}},
                        "object_transition": {"linear_sigma_x": 0.0025, "linear_sigma_y": 0.0025, "linear_sigma_z": 0.0025,
                                              "angular_sigma_x": 0.02, "angular_sigma_y": 0.02, "angular_sigma_z": 0.02,
                                              "velocity_factor": 0.8}},
    "downsampling_factor": f, "resolution": {"width": 640, "height": 480},
    "object": {"package": "x", "directory": "object_models", "meshes": ["m1.obj"]},
}
K = synth.camera_matrix(640, 480)
om = ObjectModel([v], [t], center=True)


def truth_state(k):
    Rt = synth.truth_pose(1, frame=k)[0]
    s = np.zeros(12)
    s[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
    s[0:3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]
    return s


rng = np.random.default_rng(0)
rec = ds.TrackingDataset(os.path.join(root, "recording"), load=False)
with RbSensor(om, CameraData(K, 480, 640), RbSensorBuilder.Parameters(sample_count=1), max_particles=1) as full:
    for k in range(1, frames + 1):
        kk = k % 60 if k % 60 < 30 else 60 - (k % 60)
        native = synth.make_frame(full.render_depth(synth.truth_pose(1, frame=kk)), 480, 640, rng, occluder=(k % 40 > 30))
        stamp = ds.Stamp.from_sec(1500000000.0 + k / 30.0)
        rec.add_frame(ds.Image(native.reshape(480, 640), stamp, seq=k), ds.CameraInfo(K, 480, 640, stamp, seq=k),
                      ground_truth=truth_state(kk))
rec.store()
t0 = time.perf_counter()
data = ds.TrackingDataset(os.path.join(root, "recording"))
t_load = time.perf_counter() - t0
bag = os.path.join(root, "recording", ds.OBSERVATIONS_FILENAME)
ests, wall = node.replay_dataset(params, data, root, [truth_state(0)], seed=1)
ests2, wall2 = node.replay_dataset(params, data, root, [truth_state(0)], seed=1, look_ahead=True)
err = np.array([np.linalg.norm(ests[i, 0:3] - data.get_ground_truth(i)[0:3]) for i in range(data.size())])
print(f"dataset {bag}: {os.path.getsize(bag) / 1e6:.1f} MB, {data.size()} frames, loaded in {t_load:.2f} s")
print(f"replay: {n} particles, {640 // f}x{480 // f}: {data.size() / wall:.0f} frames/s ({wall / data.size() * 1e3:.3f} ms/frame), "
      f"position error mean {err.mean() * 1e3:.2f} mm, max {err.max() * 1e3:.2f} mm")
print(f"replay with one frame of look-ahead: {data.size() / wall2:.0f} frames/s; estimates identical: {bool(np.array_equal(ests, ests2))}")
