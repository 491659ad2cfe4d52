#!/usr/bin/env python3
"""Differential stress of the two occlusion-state layouts: the same random poses (some at / behind
the camera plane, some far off screen), frames and resampling patterns into a windowed and a
whole-plane sensor; log-likelihoods must agree to 1e-12 relative and planes bit for bit.
usage: python tools/layout_diff.py [seed]"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from dbot_ros_amd import CameraData, ObjectModel, RbSensor, RbSensorBuilder, synth
from dbot_ros_amd.pose import pack_Rt, rotvec_to_matrix
n, cols, rows = 96, 160, 120
v, t = synth.mesh_m1(level=2)
om = ObjectModel([v], [t]); cam = CameraData(synth.camera_matrix(cols, rows), rows, cols)
P = RbSensorBuilder.Parameters(sample_count=n)
os.environ["RBS_TIMING_EVERY"] = "1"
os.environ["RBS_STATE"] = "window"; a = RbSensor(om, cam, P, max_particles=n)
os.environ["RBS_STATE"] = "dense"; b = RbSensor(om, cam, P, max_particles=n)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ia = np.zeros(n, np.int32); ib = ia.copy()
center = np.array([0.0, 0.0, 0.7])
for k in range(400):
    center = center + rng.normal(0, 0.03, 3)
    center[2] = abs(center[2])
    R = rotvec_to_matrix(rng.normal(size=(n, 1, 3)))
    tt = center[None, None, :] + rng.normal(0, 0.05, (n, 1, 3))
    if k % 7 == 3: tt[:8, 0, 2] = rng.uniform(-0.05, 0.05, 8)          # at / behind the camera plane
    if k % 5 == 2: tt[8:16, 0, 0] += 3.0                                # far off screen
    poses = pack_Rt(R, tt)
    frame = rng.uniform(0.3, 1.5, rows * cols).astype(np.float32)
    frame[rng.random(frame.size) < 0.05] = np.nan
    for s in (a, b): s.set_observation(frame)
    upd = rng.random() < 0.8
    la = a.loglikes_poses(poses, ia, update=upd); lb = b.loglikes_poses(poses, ib, update=upd)
    dd = np.nanmax(np.abs(la - lb) / np.maximum(1, np.abs(lb)))
    worst = max(worst, dd) if 'worst' in dir() else dd
    if dd > 1e-12 or (np.isnan(la) != np.isnan(lb)).any():
        bad = np.nonzero(~((la == lb) | (np.isnan(la) & np.isnan(lb))))[0]
        print("frame", k, "update", upd, "mismatch at particles", bad[:10], "parents", ia[bad[:10]] if not upd else "n/a", "la", la[bad[:3]], "lb", lb[bad[:3]])
        print("window of parent slots:", [a.get_window(int(p)) for p in (par_prev[bad[:3]] if 'par_prev' in dir() else bad[:3])])
        break
    if upd:
        par = rng.integers(0, n, n).astype(np.int32)
        if k % 3 == 0: par[:] = par[0]                                   # everybody inherits one parent
        par_prev = par.copy()
        ia, ib = par.copy(), par.copy()
        if k % 25 == 0:
            for slot in (0, n - 1):
                pa, pb = a.get_occlusion(slot), b.get_occlusion(slot)
                if not np.array_equal(pa, pb):
                    d = np.nonzero(pa != pb)[0]
                    print("frame", k, "plane", slot, "differs at", len(d), "pixels, first", d[:5], pa[d[:3]], pb[d[:3]], "window", a.get_window(slot)); sys.exit(1)
else:
    print("400 frames: worst relative difference", worst)
a.close(); b.close()
