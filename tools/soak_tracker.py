#!/usr/bin/env python3
"""Soak: the device tracker over a long back-and-forth sequence with intermittent occluders, once
per occlusion-state layout (windowed planes / whole planes).  Both layouts hold the same planes;
log-likelihoods agree to the order of their additions (1e-15 relative), so while the filter
tracks, the two runs give the same estimates; prints tracking error, the stored window sizes
along the way and the final plane range.
usage: python tools/soak_tracker.py [frames=1500] [particles=2000] [speed=1]
speed > 1 makes the object sweep the image (speed x 2 mm per frame): the windows grow through the
mid-size and wide regimes and back.  (Once the filter has lost the object its weights are
degenerate and last-bit differences decide resampling draws: the runs then part ways, which is
chaos, not an error -- tools/_diff-style plane comparisons are the layout check there.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, pose, synth  # noqa: E402
from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
speed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cols, rows = 640, 480
v, f = synth.mesh_m3()
om = ObjectModel([v], [f])
cam = CameraData(synth.camera_matrix(cols, rows), rows, cols)
P = RbSensorBuilder.Parameters(sample_count=n)


def truth_state(k):
    Rt = synth.truth_pose(1, frame=k)[0]
    st = np.zeros(12)
    st[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
    st[0:3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]
    return st


CONFIGS = {"window": dict(state_layout="window"), "dense": dict(state_layout="dense"),
           "slabs": dict(state_layout="window", slab_px=cols * rows // 8),
           "slabs_growing": dict(state_layout="window", slab_px=4096),     # far too small: the slabs must grow in time
           "shards": dict(state_layout="window", device_ids=[0, 0])}


def run(layout):
    ests = []
    with RbSensor(om, cam, P, max_particles=n, precision="f64", **CONFIGS[layout]) as s:
        tr = DeviceParticleTracker(ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters()).build(), s, om,
                                   ParticleTrackerBuilder.Parameters(evaluation_count=n), device_rng=True, seed=5)
        tr.initialize([truth_state(0)])
        rng = np.random.default_rng(0)
        errs, wins = [], []
        t0 = time.time()
        for k in range(1, frames + 1):
            kk = (k % 120 if k % 120 < 60 else 120 - (k % 120)) * speed - 30 * (speed - 1)   # back and forth within 60 frames
            fr = synth.make_frame(s.render_depth(synth.truth_pose(1, frame=kk)), rows, cols, rng,
                                  occluder=(k % 200 > 150))
            est = tr.track(fr)
            assert np.isfinite(est).all()
            ests.append(est.copy())
            errs.append(np.linalg.norm(est[0:3] - truth_state(kk)[0:3]))
            if k % 250 == 0:
                w = np.array([s.get_window(i) for i in range(0, n, max(1, n // 32))])
                wins.append((k, float(np.mean((w[:, 2] - w[:, 0]) * (w[:, 3] - w[:, 1])) / (cols * rows))))
        errs = np.array(errs)
        print(f"[{layout}] {frames} frames {time.time() - t0:.1f} s  position error max {errs.max():.4f} mean {errs.mean():.4f} "
              f"last-100 max {errs[-100:].max():.4f}  resamplings {tr.n_resamplings}")
        print(f"[{layout}] stored window / plane at frames:", ", ".join(f"{k}: {a:.3f}" for k, a in wins))
        occ = s.get_occlusion(0)
        print(f"[{layout}] plane of slot 0: min {occ.min():.6f} max {occ.max():.6f} NaN {int(np.isnan(occ).sum())} "
              f"background {s.get_background():.9f}")
        tr.close()
    return np.array(ests)


a = run("window")
worst = 0.0
for other in ("dense", "slabs", "slabs_growing", "shards"):
    b = run(other)
    d = np.abs(a - b).max(axis=1)
    worst = max(worst, float(d.max()))
    print("estimates window vs %s: max |difference| %.3e (bitwise identical: %s)" % (other, float(d.max()), bool(np.array_equal(a, b))))
    if d.max() > 1e-6:
        print("first frame differing by more than 1e-6:", int(np.argmax(d > 1e-6)) + 1)
        sys.exit(1)
