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
            # (before get_occlusion makes whole planes dense; handles that store against a shared background plane -- forced runs of
            # the suite, RBS_STP_ENTER=0 -- re-base on their own schedules, a slab handle's repeated calls included: same planes, other windows)
            if not (whole.shared_trail_state()[0] or slabs.shared_trail_state()[0]):
                assert whole.get_window(slot) == slabs.get_window(slot)
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


def test_a_region_that_does_not_fit_grows_the_slabs(gpu_lib):
    """a region that outgrows its slab in a synchronous call is not an error any
    more -- the call is taken back, the slabs are enlarged, the call runs again: the numbers of whole
    planes, for the particle that did not fit and for everybody else, and the handle goes on."""
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
        a = whole.loglikes_poses(poses, np.zeros(n, np.int32), update=True)
        b = g.loglikes_poses(poses, np.zeros(n, np.int32), update=True)
        assert np.array_equal(a, b)
        for s in (g, whole):
            s.set_observation(frame)
        idx = rng.permutation(n).astype(np.int32)
        ref = whole.loglikes_poses(near, idx.copy(), update=True)
        got = g.loglikes_poses(near, idx.copy(), update=True)           # overflows, is repaired, returns
        assert np.isfinite(got).all() and np.array_equal(got, ref)
        for s_ in range(n):
            assert np.array_equal(g.get_occlusion(s_), whole.get_occlusion(s_))
        assert g.get_window(3) != (640, 480, 0, 0)                       # its plane is there, not contained
        # ... and the handle carries on, a read-only call and further updating ones
        for k in range(3):
            for s in (g, whole):
                s.set_observation(frame)
            idx = rng.integers(0, n, n).astype(np.int32)
            upd = k != 0
            assert np.array_equal(g.loglikes_poses(near, idx.copy(), update=upd), whole.loglikes_poses(near, idx.copy(), update=upd))


def test_a_region_that_does_not_fit_grows_the_slabs_of_every_shard(gpu_lib):
    """The same on a handle over three shards: the whole fan-out is taken back, every shard's slabs
    grow alike (a shard reads its neighbours' planes with its own stride), the call runs again."""
    n = 9
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(2)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=16384, device_ids=[0, 0, 0]) as g, \
            RbSensor(om, cam, P, max_particles=n, precision="f64") as whole:
        truth = synth.truth_pose(1)
        frame = synth.make_frame(whole.render_depth(truth), 480, 640, rng)
        poses = synth.particle_poses(truth, n, rng)
        near = poses.copy()
        near[7, 0, 9:12] = (0.0, 0.0, 0.25)          # a particle of the LAST shard
        for s in (g, whole):
            s.reset()
            s.set_observation(frame)
        assert np.array_equal(g.loglikes_poses(poses, np.zeros(n, np.int32), update=True),
                              whole.loglikes_poses(poses, np.zeros(n, np.int32), update=True))
        for k in range(3):
            for s in (g, whole):
                s.set_observation(frame)
            idx = rng.permutation(n).astype(np.int32)      # parents on other shards
            ref = whole.loglikes_poses(near, idx.copy(), update=True)
            got = g.loglikes_poses(near, idx.copy(), update=True)
            assert np.isfinite(got).all() and np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        for s_ in range(n):
            assert np.array_equal(g.get_occlusion(s_), whole.get_occlusion(s_))


def test_slabs_grow_ahead_of_an_approaching_object(gpu_lib):
    """An object that comes closer frame by frame: the regions grow a little per frame, and the slabs
    are enlarged at a synchronising call BEFORE one overflows (three quarters full) -- no call is
    ever repeated, asynchronous callers included, and the numbers stay those of whole planes."""
    import torch
    n = 16
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(1)
    dev = torch.device("cuda", 0)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=8192) as g, \
            RbSensor(om, cam, P, max_particles=n, precision="f64") as whole:
        for s in (g, whole):
            s.reset()
        idx = np.zeros(n, np.int32)
        d_out = torch.empty(n, dtype=torch.float64, device=dev)
        for k in range(40):
            truth = synth.truth_pose(1, frame=k)
            truth[0, 11] = 0.7 - 0.011 * k                # 0.7 m -> 0.27 m: the footprint grows ~7x
            frame = synth.make_frame(whole.render_depth(truth), 480, 640, rng, occluder=False).astype(np.float32)
            poses = synth.particle_poses(truth, n, rng)
            whole.set_observation(frame)
            ref = whole.loglikes_poses(poses, idx.copy(), update=True)
            # the asynchronous route: device pointers, then one synchronising call per frame
            d_frame = torch.from_numpy(frame).to(dev)
            d_poses = torch.from_numpy(np.ascontiguousarray(poses.reshape(n, -1))).to(dev)
            d_idx = torch.from_numpy(idx).to(dev)
            torch.cuda.synchronize()
            g.set_observation_device(d_frame.data_ptr(), None)
            g.loglikes_device(d_poses.data_ptr(), d_idx.data_ptr(), n, True, d_out.data_ptr(), None)
            g.synchronize()                                # never reports an overflow: the slabs grew in time
            assert np.array_equal(d_out.cpu().numpy(), ref), k
            idx = np.sort(rng.integers(0, n, n)).astype(np.int32)
        for s_ in (0, n - 1):
            assert np.array_equal(g.get_occlusion(s_), whole.get_occlusion(s_))


def test_an_overflow_in_an_asynchronous_call_is_reported_once(gpu_lib):
    """A call that has already returned cannot be repeated: a region that outgrows its slab at once
    (here: a particle three times closer out of the blue) is contained -- NaN, plane reset -- and the
    next synchronising call says so, once; the slabs have grown, the calls that follow fit."""
    import torch
    n = 8
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(0)
    dev = torch.device("cuda", 0)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=16384) as g, \
            RbSensor(om, cam, P, max_particles=n, precision="f64") as whole:
        truth = synth.truth_pose(1)
        frame = synth.make_frame(g.render_depth(truth), 480, 640, rng).astype(np.float32)
        near = synth.particle_poses(truth, n, rng)
        near[3, 0, 9:12] = (0.0, 0.0, 0.25)
        for s in (g, whole):
            s.reset()
            s.set_observation(frame)
        ref = whole.loglikes_poses(near, np.zeros(n, np.int32), update=True)
        d_poses = torch.from_numpy(np.ascontiguousarray(near.reshape(n, -1))).to(dev)
        d_idx = torch.zeros(n, dtype=torch.int32, device=dev)
        d_out = torch.empty(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        g.loglikes_device(d_poses.data_ptr(), d_idx.data_ptr(), n, True, d_out.data_ptr(), None)
        with pytest.raises(RbSensorError) as e:
            g.synchronize()
        assert e.value.code == _capi.RBS_ERR_OUT_OF_MEMORY and "state_slab_px" in str(e.value)
        out = d_out.cpu().numpy()
        assert np.isnan(out[3]) and np.array_equal(np.delete(out, 3), np.delete(ref, 3))     # the others are untouched
        assert g.get_window(3) == (640, 480, 0, 0)                                      # its plane: all background
        g.synchronize()                                                                  # reported once
        # the same poses again now fit (particle 3 starts from the background: compare the others' planes
        # and everybody's second-call likelihood against a whole-plane handle given the same history)
        whole.set_occlusion(3, np.full(640 * 480, whole.get_background(), np.float32))
        for s in (g, whole):
            s.set_observation(frame)
        idx = np.arange(n, dtype=np.int32)
        assert np.array_equal(g.loglikes_poses(near, idx.copy(), update=True), whole.loglikes_poses(near, idx.copy(), update=True))


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
        full = np.full(160 * 120, 0.5, np.float32)                      # differs from the background everywhere:
        g.set_occlusion(1, full)                                        # the slabs grow for it (round 3: OUT_OF_MEMORY)
        assert np.array_equal(g.get_occlusion(1), full) and g.get_window(1) == (0, 0, 160, 120)
        assert np.array_equal(g.get_occlusion(4), planes[2])            # the other slots moved with the reallocation
        idx = np.array([1, 4, 0, 1, 2, 3], np.int32)
        poses = synth.particle_poses(frames[-1][0], n, np.random.default_rng(3))
        assert np.isfinite(g.loglikes_poses(poses, idx, update=True)).all()
        with pytest.raises(RbSensorError):
            g.occlusion_device_ptr(0, next_buffer=True)


def test_library_chosen_slabs_are_sized_before_the_first_asynchronous_call(gpu_lib):
    """state_slab_px = 0 with more than 8 192 particles: the library picks slabs of rows*cols/8 floats.  An object whose
    region is larger than that must not turn a default-config handle's FIRST rbs_loglikes_device into NaNs:
    the regions are probed in front of that call and the slabs enlarged; the numbers are those of whole planes."""
    import torch
    n = 8200
    om, cam, P = sc.make_scene(("m1",), 320, 240, max_particles=n)
    truth = synth.truth_pose(1, z=0.22)                                 # close: the object's rectangle is ~200 x 160 px > 9 600 px
    rng = np.random.default_rng(5)
    poses = synth.particle_poses(truth, n, rng)
    with RbSensor(om, cam, P, max_particles=n) as auto, RbSensor(om, cam, P, max_particles=n, slab_px=-1) as whole:
        frame = synth.make_frame(whole.render_depth(truth), 240, 320, rng)
        assert np.isfinite(frame).any()
        d_poses = torch.from_numpy(poses.reshape(n, -1)).cuda()
        d_idx = torch.zeros(n, dtype=torch.int32, device="cuda")
        out = []
        for s_ in (auto, whole):
            s_.reset()
            s_.set_observation(frame)
            d_out = torch.empty(n, dtype=torch.float64, device="cuda")
            s_.loglikes_device(d_poses.data_ptr(), d_idx.data_ptr(), n, True, d_out.data_ptr())
            s_.synchronize()                                            # (would report a contained overflow)
            out.append(d_out.cpu().numpy())
        assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1])
        x0, y0, x1, y1 = whole.get_window(0)
        assert (x1 - x0) * (y1 - y0) > 320 * 240 // 8, "the scenario must need more than the library's first choice"
        assert np.array_equal(auto.get_occlusion(17), whole.get_occlusion(17))


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
    for slab in (-1, 640 * 480 // 8):      # whole planes (above 8 192 particles 0 = the library's choice = slabs) / slabs
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


def test_a_synchronous_call_reports_an_earlier_asynchronous_overflow(gpu_lib):
    """The overflow of an rbs_loglikes_device call that nobody has synchronised on yet is reported by
    the next rbs_loglikes too -- once, with that call's own results intact (it runs, or is repeated,
    on the enlarged slabs)."""
    import ctypes as C
    import torch
    n = 8
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(0)
    dev = torch.device("cuda", 0)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", slab_px=16384) as g, \
            RbSensor(om, cam, P, max_particles=n, precision="f64") as whole:
        truth = synth.truth_pose(1)
        frame = synth.make_frame(g.render_depth(truth), 480, 640, rng).astype(np.float32)
        near = synth.particle_poses(truth, n, rng)
        near[3, 0, 9:12] = (0.0, 0.0, 0.25)
        for s in (g, whole):
            s.reset()
            s.set_observation(frame)
        d_poses = torch.from_numpy(np.ascontiguousarray(near.reshape(n, -1))).to(dev)
        d_idx = torch.zeros(n, dtype=torch.int32, device=dev)
        d_out = torch.empty(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        g.loglikes_device(d_poses.data_ptr(), d_idx.data_ptr(), n, True, d_out.data_ptr(), None)     # particle 3 overflows, contained
        whole.loglikes_poses(near, np.zeros(n, np.int32), update=True)
        whole.set_occlusion(3, np.full(640 * 480, whole.get_background(), np.float32))           # the same history: 3 starts over
        for s in (g, whole):
            s.set_observation(frame)
        ref = whole.loglikes_poses(near, np.arange(n, dtype=np.int32), update=True)
        idx, out = np.arange(n, dtype=np.int32), np.empty(n)
        rc = g._lib.rbs_loglikes(g._h, near.reshape(n, -1).ctypes.data_as(C.POINTER(C.c_double)),
                                 idx.ctypes.data_as(C.POINTER(C.c_int32)), n, 1, out.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == _capi.RBS_ERR_OUT_OF_MEMORY and b"already returned" in g._lib.rbs_last_error(g._h)
        assert np.isnan(d_out.cpu().numpy()[3])                       # the asynchronous call's particle was contained ...
        assert np.array_equal(out, ref)                                # ... this call's results are those of whole planes
        for s in (g, whole):
            s.set_observation(frame)
        idx = rng.permutation(n).astype(np.int32)
        assert np.array_equal(g.loglikes_poses(near, idx.copy(), update=True), whole.loglikes_poses(near, idx.copy(), update=True))


def test_device_tracker_on_slabs_that_start_too_small(gpu_lib):
    """A tracker frame cannot be repeated, so rbs_tracker_initialize sizes the slabs with a probe call at
    the default pose, and every frame's result enlarges them ahead of the regions: a slab of 4 096 px
    -- smaller than the object's own rectangle -- never overflows, and the estimates are those of whole
    planes, bit for bit, while the object comes closer."""
    from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder
    n = 400
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    rng = np.random.default_rng(3)
    ests = []
    for slab in (-1, 4096):
        with RbSensor(om, cam, P, max_particles=n, slab_px=slab) as s:
            if not ests:
                frames = []
                for k in range(1, 41):
                    t = synth.truth_pose(1, frame=k)
                    t[0, 11] -= 0.006 * k                      # 0.7 m -> 0.46 m: the footprint more than doubles
                    frames.append(synth.make_frame(s.render_depth(t), 480, 640, rng, occluder=False).astype(np.float32))
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
    assert np.isfinite(ests[1]).all() and np.array_equal(ests[0], ests[1])
