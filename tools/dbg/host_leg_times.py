"""Where the host-pointer step's time goes: set_observation vs loglikes, copy vs staged (GPU box)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, synth
n = 2000
v, t = synth.mesh_m1()
om = ObjectModel([v], [t]); cam = CameraData(synth.camera_matrix(), 480, 640)
P = RbSensorBuilder.Parameters(sample_count=n)
rng = np.random.default_rng(0)
with RbSensor(om, cam, P, max_particles=n) as s:
    truths = [synth.truth_pose(1, frame=k) for k in range(30)]
    frames = np.stack([synth.make_frame(s.render_depth(tr), 480, 640, rng) for tr in truths]).astype(np.float32)
    poses = np.stack([synth.particle_poses(tr, n, rng).reshape(n, -1) for tr in truths])
    parents = rng.permutation(n).astype(np.int32)
    def run(mode, steps=300):
        ts = tl = 0.0
        for i in range(steps + 10):
            k = i % 30
            t0 = time.perf_counter()
            if mode == "copy": s.set_observation(frames[k])
            elif mode == "staged": s.frame_buffer(); s.commit_frame()
            elif mode == "numpy-into-staging": np.copyto(s.frame_buffer(), frames[k]); s.commit_frame()
            t1 = time.perf_counter()
            s.loglikes_poses(poses[k], parents.copy(), update=True)
            t2 = time.perf_counter()
            if i >= 10: ts += t1 - t0; tl += t2 - t1
        print("%-20s set_observation %.1f us  loglikes %.1f us  -> %.2f M/s" % (mode, ts / steps * 1e6, tl / steps * 1e6, n * steps / (ts + tl) / 1e6))
    for m in ("none", "staged", "copy", "numpy-into-staging", "copy", "staged"):
        run(m)
