"""Window-sized slabs for the occlusion state (rbs_config.state_slab_px): a slot holds a fixed
number of floats and stores only the region an updating call writes.  The NUMBERS must be those of
whole planes, bit for bit (same kernels, same values, another address); what does not fit is
contained and reported, never silently wrong."""
import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import RbSensor, RbSensorError, _capi, pose, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("meshes,cols,rows,n", [(("m1",), 640, 480, 24), (("m1", "m2", "m3"), 640, 480, 12),
                                                 (("m1_l2",), 160, 120, 40)])
def test_slabs_hold_the_same_planes_as_whole_planes(gpu_lib, precision, meshes, cols, rows, n):
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    nb = len(meshes)
    eager = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    frames = sc.make_frames(eager, nb, 6, seed=3)
    slab = (cols * rows) // (2 if nb > 1 else 8)
    with RbSensor(om, cam, P, max_particles=n, precision=precision) as whole, \
            RbSensor(om, cam, P, max_particles=n, precision=precision, slab_px=slab) as slabs:
        a = sc.run_sequence(whole, frames, n, n_bodies=nb)
        b = sc.run_sequence(slabs, frames, n, n_bodies=nb)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        for slot in range(n):
            assert whole.get_window(slot) == slabs.get_window(slot)       # (before get_occlusion makes whole planes dense)
        for slot in range(n):
            assert np.array_equal(whole.get_occlusion(slot), slabs.get_occlusion(slot))
        # read-only calls with resampled parents
        rng = np.random.default_rng(1)
        idx = rng.integers(0, n, n).astype(np.int32)
        poses = synth.particle_poses(frames[-1][0], n, rng)
        assert np.array_equal(whole.loglikes_poses(poses, idx.copy(), update=False),
                              slabs.loglikes_poses(poses, idx.copy(), update=False))
    if precision == "f64":
        ref = sc.run_sequence(eager, frames, n, n_bodies=nb)
        for x, y in zip(b, ref):
            assert (np.abs(x - y) / np.maximum(1.0, np.abs(y))).max() <= 1e-9


def test_a_region_that_does_not_fit_is_contained_and_reported(gpu_lib):
    n = 8
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(0)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=16384) as g, \
            RbSensor(om, cam, P, max_particles=n, precision="f64") as whole:
        truth = synth.truth_pose(1)
        frame = synth.make_frame(g.render_depth(truth), 480, 640, rng)
        poses = synth.particle_poses(truth, n, rng)
        near = poses.copy()
        near[3, 0, 9:12] = (0.0, 0.0, 0.25)          # particle 3 three times closer: ~250 x 200 px > 16 384
        for s in (g, whole):
            s.reset()
            s.set_observation(frame)
        ref = whole.loglikes_poses(near, np.zeros(n, np.int32), update=True)
        idx = np.zeros(n, np.int32)
        out = np.empty(n)
        import ctypes as C
        rc = g._lib.rbs_loglikes(g._h, near.reshape(n, -1).ctypes.data_as(C.POINTER(C.c_double)),
                                 idx.ctypes.data_as(C.POINTER(C.c_int32)), n, 1, out.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == _capi.RBS_ERR_OUT_OF_MEMORY
        assert b"state_slab_px" in g._lib.rbs_last_error(g._h)
        assert np.isnan(out[3]) and np.array_equal(np.delete(out, 3), np.delete(ref, 3))     # the others are untouched
        assert g.get_window(3) == (640, 480, 0, 0)                                      # its plane: all background
        assert np.array_equal(g.get_occlusion(2), whole.get_occlusion(2))
        with pytest.raises(RbSensorError):                                               # sticky until reset
            g.loglikes_poses(poses, np.arange(n, dtype=np.int32), update=False)
        g.reset()
        g.set_observation(frame)
        whole.reset()
        whole.set_observation(frame)
        assert np.array_equal(g.loglikes_poses(poses, np.zeros(n, np.int32), update=True),
                              whole.loglikes_poses(poses, np.zeros(n, np.int32), update=True))


def test_plane_hooks_on_slabs(gpu_lib):
    """get / set / export / import speak whole planes whatever the slots are."""
    import torch
    n = 6
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    eager = ob.Oracle(om, cam, P, max_particles=n, mode=ob.EAGER)
    frames = sc.make_frames(eager, 1, 3, seed=5)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=4096) as g:
        sc.run_sequence(g, frames, n)
        ref = sc.run_sequence(eager, frames, n)
        planes = [g.get_occlusion(s) for s in range(n)]
        for s in range(n):
            o = eager.get_occlusion(s)
            assert (planes[s] != o).mean() <= 1e-4
        t = torch.empty(160 * 120, dtype=torch.float32, device="cuda")
        g.export_plane(2, t.data_ptr())
        g.synchronize()
        assert np.array_equal(t.cpu().numpy(), planes[2])
        g.import_plane(4, t.data_ptr())
        assert np.array_equal(g.get_occlusion(4), planes[2])
        g.set_occlusion(5, planes[0])
        assert np.array_equal(g.get_occlusion(5), planes[0])
        assert g.get_window(5) == g.get_window(0) or g.get_window(0)[2] - g.get_window(0)[0] >= g.get_window(5)[2] - g.get_window(5)[0]
        full = np.full(160 * 120, 0.5, np.float32)                      # differs from the background everywhere
        with pytest.raises(RbSensorError) as e:
            g.set_occlusion(1, full)
        assert e.value.code == _capi.RBS_ERR_OUT_OF_MEMORY
        with pytest.raises(RbSensorError):
            g.occlusion_device_ptr(0, next_buffer=True)


def test_slabs_in_a_handle_over_several_shards(gpu_lib):
    n = 30
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    rng = np.random.default_rng(0)
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as one:
        frames = []
        for k in range(4):
            t = synth.truth_pose(1, frame=k)
            frames.append((t, synth.make_frame(one.render_depth(t), 120, 160, rng)))
        ref = sc.run_sequence(one, frames, n, permute=True)
        planes = [one.get_occlusion(s) for s in range(n)]
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=4096, device_ids=[0, 0, 0]) as grp:
        got = sc.run_sequence(grp, frames, n, permute=True)
        for a, b in zip(got, ref):
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
        for s in range(n):
            assert np.array_equal(grp.get_occlusion(s), planes[s])


def test_c3_slice_on_an_eighth_of_the_memory(gpu_lib):
    """25 000 particles at 640x480 in 7.7 GB of occlusion state instead of 61 GB; the same
    log-likelihoods as on whole planes."""
    n = 25000
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(21)
    out = []
    for slab in (0, 640 * 480 // 8):
        with RbSensor(om, cam, P, max_particles=n, slab_px=slab) as g:
            if not out:
                truth = synth.truth_pose(1)
                frame = synth.make_frame(g.render_depth(truth), 480, 640, rng)
                poses = synth.particle_poses(truth, n, rng, scale=2.0)
                perm = rng.permutation(n).astype(np.int32)
            g.reset()
            g.set_observation(frame)
            a = g.loglikes_poses(poses, np.zeros(n, np.int32), update=True)
            g.set_observation(frame)
            b = g.loglikes_poses(poses, perm.copy(), update=True)
            out.append((a, b))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_device_tracker_on_slabs(gpu_lib):
    from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder
    n = 600
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(3)
    ests = []
    for slab in (0, 640 * 480 // 8):
        with RbSensor(om, cam, P, max_particles=n, slab_px=slab) as s:
            if not ests:
                frames = [synth.make_frame(s.render_depth(synth.truth_pose(1, frame=k)), 480, 640, rng, occluder=False).astype(np.float32)
                          for k in range(1, 9)]
                randomness = [(rng.standard_normal((1, n, 6)), rng.random((1, n))) for _ in frames]
            trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters()).build()
            tr = DeviceParticleTracker(trans, s, om, ParticleTrackerBuilder.Parameters(evaluation_count=n), np.random.default_rng(5))
            init = np.zeros(12)
            Rt = synth.truth_pose(1, frame=0)[0]
            init[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
            init[0:3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]
            tr.initialize([init])
            ests.append(np.array([tr.track(f, nz, u) for f, (nz, u) in zip(frames, randomness)]))
            assert tr.n_resamplings >= 1
            tr.close()
    assert np.array_equal(ests[0], ests[1])
