#!/usr/bin/env python3
"""Tracker FPS (the second half of BASELINE.json's metric): frames/s of the full per-frame step
-- set_observation + transition + loglikes(update) + weights + KL + resample + mean -- on a
30-frame synthetic sequence (object translating 2 mm and rotating 1 degree per frame), for
{200, 2 000, 20 000} particles on one MI355X.  Two variants per count: filter='device' (rbs_tracker_*: transition, weights, KL,
resampling, mean on the GPU, one host sync per frame) and filter='host' (numpy mirror).  Prints one JSON line per particle count."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, pose, synth  # noqa: E402
from dbot_ros_amd.tracker import (DeviceParticleTracker, ObjectTransitionBuilder, ParticleTracker,  # noqa: E402
                                  ParticleTrackerBuilder)


def main():
    counts = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [200, 2000, 20000]
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["m1"]     # e.g. m1,m2,m3 = BASELINE config C2
    cols, rows, n_frames = 640, 480, 30
    fns = {"m1": synth.mesh_m1, "m2": synth.mesh_m2, "m3": synth.mesh_m3, "m4": synth.mesh_m4}
    meshes = [fns[k]() for k in names]
    nb = len(meshes)
    f = np.concatenate([t for _, t in meshes])
    om = ObjectModel([v for v, _ in meshes], [t for _, t in meshes], center=True)
    cam = CameraData(synth.camera_matrix(cols, rows), rows, cols)
    for n, mode in [(n, m) for n in counts for m in ("device", "host")]:
        P = RbSensorBuilder.Parameters(sample_count=n)
        with RbSensor(om, cam, P, max_particles=max(1, n // nb)) as s:
            rng = np.random.default_rng(0)
            frames = [synth.make_frame(s.render_depth(synth.truth_pose(nb, frame=k)), rows, cols, rng,
                                       occluder=False) for k in range(n_frames + 1)]
            trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=nb)).build()
            tp = ParticleTrackerBuilder.Parameters(evaluation_count=n)
            if mode == "device":   # transition + filter step + mean on the GPU, device RNG
                tr = DeviceParticleTracker(trans, s, om, tp, device_rng=True, seed=1)
            else:                  # host mirror (numpy)
                tr = ParticleTracker(trans, s, om, tp, np.random.default_rng(1))
            init = np.zeros(12 * nb)
            for b in range(nb):
                Rt = synth.truth_pose(nb, frame=0)[b]
                init[12 * b + 3:12 * b + 6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
                init[12 * b:12 * b + 3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[b]
            tr.initialize([init])
            tr.track(frames[0])  # warm-up
            t0 = time.perf_counter()
            sensor_ms = 0.0
            for k in range(1, n_frames + 1):
                est = tr.track(frames[k])
            dt = time.perf_counter() - t0
            sensor_ms = s.timing_summary(n_frames * nb)[0] * n_frames * nb   # sampled kernel timing, outside the loop
            Rt = synth.truth_pose(nb, frame=n_frames)[0]
            err = np.linalg.norm(est[0:3] - (Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]))
            print(json.dumps({"metric": "tracker FPS", "filter": mode, "evaluation_count": n, "objects": nb, "particles": max(1, n // nb), "value": n_frames / dt, "unit": "frames/s",
                              "ms_per_frame": dt / n_frames * 1e3, "sensor_device_ms_per_frame": sensor_ms / n_frames,
                              "resamplings": tr.n_resamplings, "final_position_error_m": float(err),
                              "resolution": [cols, rows], "triangles": int(len(f)), "n_gpus": 1}), flush=True)


if __name__ == "__main__":
    main()
