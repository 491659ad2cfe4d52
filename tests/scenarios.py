"""Seeded scenes shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np

from dbot_ros_amd import CameraData, ObjectModel, RbSensorBuilder, synth
import oracle_binding as ob

MESHES = {"m1": synth.mesh_m1, "m2": synth.mesh_m2, "m3": synth.mesh_m3, "m4": synth.mesh_m4,
          "box12": synth.mesh_box12, "m1_l2": lambda: synth.mesh_m1(level=2),
          # 134 680 triangles = 2 105 clusters: more than one stretch of the shared cluster cull (kCullSteps x 64 clusters)
          "m4_fine": lambda: synth.mesh_m4(260, 260)}


def usable_threads():
    """CPUs this process may use: the affinity mask capped by the cgroup quota (the GPU boxes report
    256 hardware threads and grant 16 CPUs: a team sized by os.cpu_count() is throttled)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def make_scene(mesh_names=("m1",), cols=640, rows=480, max_particles=16, z=0.7):
    vs, fs = zip(*[MESHES[m]() for m in mesh_names])
    om = ObjectModel(list(vs), list(fs), center=True)
    cam = CameraData(synth.camera_matrix(cols, rows), rows, cols)
    params = RbSensorBuilder.Parameters(sample_count=max_particles)
    return om, cam, params


def make_frames(oracle, n_bodies, n_frames, seed=0, z=0.7, **kw):
    """Frames from the ORACLE's renderer (CPU tests / parity tests only)."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_frames):
        truth = synth.truth_pose(n_bodies, z=z, frame=k)
        d = oracle.render_depth(truth)
        out.append((truth, synth.make_frame(d, oracle.rows, oracle.cols, rng, **kw)))
    return out


def run_sequence(sensor, frames, n, seed=1, n_bodies=1, update_every=True, permute=True, abs_sums=None):
    """Drive a sensor (oracle or product: same method names) through a frame sequence the way
    the filter does: set_observation, loglikes(update=True), resample-like index shuffle.
    Returns per-frame log-likelihood arrays."""
    rng = np.random.default_rng(seed)
    sensor.reset()
    indices = np.zeros(n, dtype=np.int32)
    lls = []
    for k, (truth, frame) in enumerate(frames):
        sensor.set_observation(frame)
        poses = synth.particle_poses(truth, n, rng, scale=1.0 + 0.5 * k)
        ll = sensor.loglikes_poses(poses, indices, update=True)
        lls.append(ll)
        if abs_sums is not None:     # oracle only: the conditioning of each particle's sum
            abs_sums.append(sensor.last_abs_sums(n))
        assert (indices == np.arange(n)).all()
        if permute:
            # children inherit from parents drawn with replacement (multinomial resampling)
            w = np.exp(ll - ll.max())
            indices = np.sort(rng.choice(n, size=n, p=w / w.sum())).astype(np.int32)
    return lls
