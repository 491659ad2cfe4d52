#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the C oracle (run from the repo root:
    python tests/golden/make_golden.py).

The reference (/root/reference, dbot_ros) holds no tests, fixtures or golden vectors for this
path and the packages that implement it (dbot, fl) are absent, so these vectors pin THIS repo's
restatement (oracle/rbsensor_oracle.c; PARITY UNPINNED against upstream).  They are data:
seeded inputs + the oracle's outputs, small enough to commit, and are checked three ways:
  - the C oracle must keep reproducing them bit-for-bit            (tests/test_oracle.py)
  - the independent numpy twin must reproduce them                 (tests/test_oracle.py)
  - the HIP path must reproduce them within the stated tolerance   (tests/test_gpu_parity.py)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_binding as ob  # noqa: E402
import scenarios as sc  # noqa: E402
from dbot_ros_amd import synth  # noqa: E402


def pixel_model():
    om, cam, P = sc.make_scene(("m1_l2",), 80, 60, max_particles=1)
    o = ob.Oracle(om, cam, P, max_particles=1)
    obs = np.array([0.3, 0.5, 0.69, 0.7, 0.705, 0.72, 0.9, 1.5, 3.0, 5.9])
    ren = np.array([0.4, 0.6, 0.7, 0.71, 1.0, 2.5, np.inf])
    pv = np.array([[o.prob_visible(a, b) for b in ren] for a in obs])
    po = np.array([[o.prob_occluded(a, b) for b in ren] for a in obs])
    dts = np.array([1 / 30, 2 / 30, 1.0])
    occs = np.array([0.0, 0.1, 0.25, 0.5, 0.9, 1.0])
    pr = np.array([[o.propagate(a, d) for d in dts] for a in occs])
    coeffs = np.array([o.eager_coeffs(k) for k in range(4)], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "pixel_model.npz"), obs=obs, rendered=ren, p_visible=pv,
                        p_occluded=po, dts=dts, occs=occs, propagated=pr, eager_coeffs=coeffs)


def coverage():
    out = {}
    for mesh in ("m1_l2", "m3", "box12"):
        for cols, rows in ((80, 60), (160, 120)):
            om, cam, P = sc.make_scene((mesh,), cols, rows, max_particles=1)
            o = ob.Oracle(om, cam, P, max_particles=1)
            rng = np.random.default_rng(17)
            poses, depths = [], []
            for k in range(5):
                pose = synth.particle_poses(synth.truth_pose(1, z=0.45 + 0.12 * k, frame=4 * k), 1, rng, scale=5.0)[0]
                poses.append(pose)
                depths.append(o.render_depth(pose))
            out[f"{mesh}_{cols}x{rows}_poses"] = np.array(poses)
            out[f"{mesh}_{cols}x{rows}_depth"] = np.array(depths)
    np.savez_compressed(os.path.join(HERE, "coverage.npz"), **out)


def coverage_vga():
    """640x480 coverage of the 5 120-triangle meshes, stored sparsely (covered pixel ids + depths)."""
    out = {}
    for mesh in ("m1", "m3", "box12"):
        om, cam, P = sc.make_scene((mesh,), 640, 480, max_particles=1)
        o = ob.Oracle(om, cam, P, max_particles=1)
        rng = np.random.default_rng(23)
        for k in range(5):
            pose = synth.particle_poses(synth.truth_pose(1, z=0.45 + 0.12 * k, frame=4 * k), 1, rng, scale=5.0)[0]
            d = o.render_depth(pose)
            ids = np.nonzero(np.isfinite(d))[0].astype(np.int32)
            out[f"{mesh}_{k}_pose"] = pose
            out[f"{mesh}_{k}_ids"] = ids
            out[f"{mesh}_{k}_depth"] = d[ids]
    np.savez_compressed(os.path.join(HERE, "coverage_vga.npz"), **out)


def sequences():
    out = {}
    for name, meshes, cols, rows, n in (("single", ("m1_l2",), 80, 60, 16), ("multi", ("m1_l2", "box12"), 160, 120, 16)):
        om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
        for mode, tag in ((ob.LAZY, "lazy"), (ob.EAGER, "eager")):
            o = ob.Oracle(om, cam, P, max_particles=n, mode=mode)
            frames = sc.make_frames(o, len(meshes), 3, seed=21)
            lls = sc.run_sequence(o, frames, n, n_bodies=len(meshes))
            out[f"{name}_{tag}_loglik"] = np.array(lls)
            out[f"{name}_{tag}_occ_slot0"] = o.get_occlusion(0)
            out[f"{name}_{tag}_occ_slot5"] = o.get_occlusion(5)
        out[f"{name}_frames"] = np.array([f for _, f in frames])
        out[f"{name}_truth"] = np.array([t for t, _ in frames])
    np.savez_compressed(os.path.join(HERE, "sequences.npz"), **out)


if __name__ == "__main__":
    pixel_model()
    coverage()
    coverage_vga()
    sequences()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
