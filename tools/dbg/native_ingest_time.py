"""Time rbs_set_observation_native_f32 (640x480 native -> 80x60 evaluated, factor 8) + a synchronous
loglikes of 200 particles; run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import scenarios as sc
from dbot_ros_amd import RbSensor, synth
n, f = 200, 8
om, cam, P = sc.make_scene(("m1",), 80, 60, max_particles=n)
rng = np.random.default_rng(0)
native = rng.uniform(0.4, 1.5, (480, 640)).astype(np.float32)
poses = synth.particle_poses(synth.truth_pose(1, frame=0), n, rng).reshape(n, 1, 12)
with RbSensor(om, cam, P, max_particles=n) as s:
    idx = np.arange(n, dtype=np.int32)
    for _ in range(20):
        s.set_observation_native(native, f); s.loglikes_poses(poses, idx.copy(), update=True)
    t0 = time.perf_counter()
    for _ in range(300):
        s.set_observation_native(native, f)
    s.synchronize()
    t1 = time.perf_counter()
    for _ in range(300):
        s.set_observation_native(native, f); s.loglikes_poses(poses, idx.copy(), update=True)
    t2 = time.perf_counter()
    print("set_observation_native alone: %.1f us/call; + loglikes(200, update): %.1f us/step" % ((t1 - t0) / 300 * 1e6, (t2 - t1) / 300 * 1e6))
