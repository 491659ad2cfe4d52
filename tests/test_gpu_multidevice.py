"""Several devices inside ONE handle (rbs_config.n_devices / device_ids): the
reference's tracker node is a single process, so particle sharding lives behind the C-ABI.  On a
one-GPU box the same ordinal is listed several times: the shards then share a device, every
"peer" read is a local read and the log-likelihood exchange falls back from RCCL (which refuses
duplicate devices) to device-to-device copies -- everything else (global slots, cross-shard
ordering, the layout kernel, the replicated filter) is the code that runs on eight GPUs.  With
two or more GPUs visible the distinct-device tests below run RCCL over xGMI."""
import numpy as np
import pytest

import scenarios as sc
from dbot_ros_amd import RbSensor, pose, synth
from dbot_ros_amd.tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTrackerBuilder

pytestmark = pytest.mark.gpu


def _devices(kind):
    import torch
    if kind == "same":
        return None
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    return [0, 1]


def _resampled_sequence(sensor, frames, n, seed=1):
    """set_observation -> loglikes(update) -> multinomial resampling with GLOBAL parent slots."""
    rng = np.random.default_rng(seed)
    sensor.reset()
    idx = np.zeros(n, dtype=np.int32)
    out = []
    for k, (truth, frame) in enumerate(frames):
        sensor.set_observation(frame)
        poses = synth.particle_poses(truth, n, rng, scale=1.0 + 0.5 * k)
        if k % 2 == 1:      # a read-only evaluation first (non-final sampling block)
            out.append(sensor.loglikes_poses(poses, idx.copy(), update=False))
        ll = sensor.loglikes_poses(poses, idx, update=True)
        out.append(ll)
        assert (idx == np.arange(n)).all()
        w = np.exp(ll - ll.max())
        idx = rng.choice(n, size=n, p=w / w.sum()).astype(np.int32)      # unsorted: parents all over the shards
    return out


@pytest.mark.parametrize("layout", ["window", "dense"])
@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0]])
def test_group_sensor_matches_single_device(gpu_lib, layout, ids):
    n = 50          # not a multiple of the shard count: the last shard is partly filled
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", state_layout=layout) as one:
        rng = np.random.default_rng(0)
        frames = []
        for k in range(5):
            t = synth.truth_pose(1, frame=k)
            frames.append((t, synth.make_frame(one.render_depth(t), 120, 160, rng)))
        ref = _resampled_sequence(one, frames, n)
        ref_planes = [one.get_occlusion(s) for s in range(n)]
    with RbSensor(om, cam, P, max_particles=n, precision="f64", state_layout=layout, device_ids=ids) as grp:
        got = _resampled_sequence(grp, frames, n)
        for a, b in zip(got, ref):
            # the same planes and kernels; a particle's tiles may be summed in another order
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), np.abs(a - b).max()
        for s in range(n):
            assert np.array_equal(grp.get_occlusion(s), ref_planes[s]), s
        with pytest.raises(Exception):
            grp.loglikes_poses(synth.particle_poses(frames[0][0], n, rng), np.full(n, 10 ** 6, np.int32))


def _init_state(om, nb):
    init = np.zeros(12 * nb)
    for b in range(nb):
        Rt = synth.truth_pose(nb, frame=0)[b]
        init[12 * b + 3:12 * b + 6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
        init[12 * b:12 * b + 3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[b]
    return init


def _run_tracker(om, cam, P, n, nb, frames, device_ids, randomness):
    with RbSensor(om, cam, P, max_particles=n // nb, precision="f64", device_ids=device_ids) as s:
        trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=nb)).build()
        tr = DeviceParticleTracker(trans, s, om, ParticleTrackerBuilder.Parameters(evaluation_count=n), np.random.default_rng(5))
        tr.initialize([_init_state(om, nb)])
        ests, states = [], []
        for frame, (normals, uniforms) in zip(frames, randomness):
            ests.append(tr.track(frame, normals, uniforms))
            states.append(tr.get_state())
        nres = tr.n_resamplings
        tr.close()
    return np.array(ests), states, nres


@pytest.mark.parametrize("meshes,n,ids", [(("m1_l2",), 96, [0, 0]), (("m1_l2", "box12"), 120, [0, 0, 0]),
                                           (("m1_l2",), 1500, [0, 0])])
def test_group_device_tracker_matches_single_device(gpu_lib, meshes, n, ids):
    """rbs_tracker_* over a handle with several shards against the same tracker over one device,
    same host-supplied randomness: the same particles, weights, resampling decisions and estimates
    (the slot map differs -- it names where planes live -- but names the same planes)."""
    nb = len(meshes)
    cols, rows = 160, 120
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    per = n // nb
    rng = np.random.default_rng(3)
    with RbSensor(om, cam, P, max_particles=1) as r:
        frames = [synth.make_frame(r.render_depth(synth.truth_pose(nb, frame=k)), rows, cols, rng, occluder=False).astype(np.float32)
                  for k in range(1, 8)]
    randomness = [(rng.standard_normal((nb, per, 6)), rng.random((nb, per))) for _ in frames]
    e1, s1, r1 = _run_tracker(om, cam, P, n, nb, frames, None, randomness)
    e2, s2, r2 = _run_tracker(om, cam, P, n, nb, frames, ids, randomness)
    assert r1 == r2 and r1 >= 1
    assert np.abs(e1 - e2).max() <= 1e-9, np.abs(e1 - e2).max()
    for (p1, w1, _), (p2, w2, _) in zip(s1, s2):
        assert np.abs(p1 - p2).max() <= 1e-9
        assert np.abs(w1 - w2).max() <= 1e-6 * max(1.0, np.abs(w1).max())


@pytest.mark.parametrize("meshes,n,ids,device_rng", [(("m1_l2",), 96, None, False), (("m1_l2", "box12"), 600, None, False),
                                                      (("m1_l2",), 4000, None, True), (("m1_l2",), 96, [0, 0], False),
                                                      (("m1_l2",), 20000, None, True)])
def test_pipelined_tracker_is_the_synchronous_tracker(gpu_lib, meshes, n, ids, device_rng):
    """rbs_tracker_submit / rbs_tracker_result with two frames in flight (the caller's buffers are
    overwritten right after submit) against rbs_tracker_track frame by frame: identical estimates,
    resampling counts and final particles; misuse (a third frame, a result with nothing in flight,
    track with frames in flight) is an error that leaves the pipeline usable."""
    nb = len(meshes)
    cols, rows = 160, 120
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    per = n // nb
    rng = np.random.default_rng(3)
    with RbSensor(om, cam, P, max_particles=1) as r:
        frames = [synth.make_frame(r.render_depth(synth.truth_pose(nb, frame=k)), rows, cols, rng, occluder=False).astype(np.float32)
                  for k in range(1, 10)]
    randomness = [(None, None) if device_rng else (rng.standard_normal((nb, per, 6)), rng.random((nb, per))) for _ in frames]

    def run(pipelined):
        with RbSensor(om, cam, P, max_particles=per, precision="f64", device_ids=ids) as s:
            trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=nb)).build()
            tr = DeviceParticleTracker(trans, s, om, ParticleTrackerBuilder.Parameters(evaluation_count=n),
                                       np.random.default_rng(5), device_rng=device_rng, seed=11)
            tr.initialize([_init_state(om, nb)])
            ests, counts = [], []
            if not pipelined:
                for f, (nm, un) in zip(frames, randomness):
                    ests.append(tr.track(f, nm, un))
                    counts.append(tr.n_resamplings)
            else:
                with pytest.raises(Exception, match="no frame in flight"):
                    tr.result()
                scratch_f = np.empty_like(frames[0])
                for k, (f, (nm, un)) in enumerate(zip(frames, randomness)):
                    scratch_f[...] = f
                    nm_, un_ = (None, None) if nm is None else (nm.copy(), un.copy())
                    tr.submit(scratch_f, nm_, un_)
                    scratch_f[...] = np.nan          # the caller's buffers are its own again
                    if nm_ is not None:
                        nm_[...] = 1e9
                        un_[...] = 0.5
                    if k == 1:
                        with pytest.raises(Exception, match="in flight"):
                            tr.submit(f, nm, un)
                        with pytest.raises(Exception, match="in flight"):
                            tr.track(f, nm, un)
                    if k >= 1:
                        ests.append(tr.result())
                        counts.append(tr.n_resamplings)
                ests.append(tr.result())
                counts.append(tr.n_resamplings)
            state = tr.get_state()
            tr.close()
        return np.array(ests), counts, state

    e1, c1, s1 = run(False)
    e2, c2, s2 = run(True)
    assert c1 == c2
    assert np.array_equal(e1, e2)
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])


def test_rccl_all_gather_runs_in_the_group_path(gpu_lib, monkeypatch):
    """RBS_GROUP_SINGLE: a one-shard group on one device, so that the RCCL communicator, the
    ncclGroupStart / ncclAllGather / ncclGroupEnd calls and the run-time binding of librccl are
    executed on hardware even on a one-GPU box (one rank: the all-gather is an in-place copy)."""
    monkeypatch.setenv("RBS_GROUP_SINGLE", "1")
    n, nb = 64, 1
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    rng = np.random.default_rng(3)
    with RbSensor(om, cam, P, max_particles=1) as r:
        frames = [synth.make_frame(r.render_depth(synth.truth_pose(nb, frame=k)), 120, 160, rng, occluder=False).astype(np.float32)
                  for k in range(1, 5)]
    randomness = [(rng.standard_normal((nb, n, 6)), rng.random((nb, n))) for _ in frames]
    e2, _, r2 = _run_tracker(om, cam, P, n, nb, frames, [0], randomness)
    monkeypatch.delenv("RBS_GROUP_SINGLE")
    e1, _, r1 = _run_tracker(om, cam, P, n, nb, frames, None, randomness)
    assert r1 == r2 and np.abs(e1 - e2).max() <= 1e-9


def test_two_gpus_sensor_and_tracker(gpu_lib):
    """Distinct devices: peer reads of remote parents over xGMI, RCCL all-gather.  Skipped on a
    one-GPU box."""
    ids = _devices("distinct")
    n = 64
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as one:
        rng = np.random.default_rng(0)
        frames = []
        for k in range(4):
            t = synth.truth_pose(1, frame=k)
            frames.append((t, synth.make_frame(one.render_depth(t), 120, 160, rng)))
        ref = _resampled_sequence(one, frames, n)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=ids) as grp:
        got = _resampled_sequence(grp, frames, n)
    for a, b in zip(got, ref):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    fr = [f.astype(np.float32) for _, f in frames]
    randomness = [(rng.standard_normal((1, n, 6)), rng.random((1, n))) for _ in fr]
    e1, _, r1 = _run_tracker(om, cam, P, n, 1, fr, None, randomness)
    e2, _, r2 = _run_tracker(om, cam, P, n, 1, fr, ids, randomness)
    assert r1 == r2 and np.abs(e1 - e2).max() <= 1e-9


@pytest.mark.parametrize("ids", [None, [0, 0]])
def test_zero_copy_frame_staging_equals_set_observation(gpu_lib, ids):
    """rbs_acquire_frame_buffer / rbs_commit_frame_buffer against rbs_set_observation_f32, on one
    device and on a handle over two shards (one pinned buffer feeds every device)."""
    n = 32
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    rng = np.random.default_rng(0)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=ids) as a, \
            RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=ids) as b:
        for s in (a, b):
            s.reset()
        ia, ib = np.zeros(n, np.int32), np.zeros(n, np.int32)
        for k in range(5):
            t = synth.truth_pose(1, frame=k)
            frame = synth.make_frame(a.render_depth(t), 120, 160, rng).astype(np.float32)
            poses = synth.particle_poses(t, n, rng)
            a.set_observation(frame)
            buf = b.frame_buffer()
            buf[:] = frame
            b.commit_frame()
            la = a.loglikes_poses(poses, ia, update=True)
            lb = b.loglikes_poses(poses, ib, update=True)
            assert np.array_equal(la, lb)
            ia = rng.integers(0, n, n).astype(np.int32)
            ib = ia.copy()
        assert np.array_equal(a.get_observation(), b.get_observation(), equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ids", [None, [0, 0]])
def test_frame_buffer_calls_must_pair(gpu_lib, ids):
    """a commit without its acquire, or a second acquire, would upload into the image the
    kernels in flight read and count the frame twice -- both are refused, the handle stays usable,
    and an acquire that another rbs_set_observation* call replaced is abandoned."""
    from dbot_ros_amd.sensor import RbSensorError
    n = 8
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    rng = np.random.default_rng(1)
    with RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=ids) as s, \
            RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=ids) as ref:
        s.reset(); ref.reset()
        t = synth.truth_pose(1, frame=0)
        frame = synth.make_frame(ref.render_depth(t), 120, 160, rng).astype(np.float32)
        poses = synth.particle_poses(t, n, rng)
        with pytest.raises(RbSensorError, match="no buffer acquired"):
            s.commit_frame()
        buf = s.frame_buffer()
        with pytest.raises(RbSensorError, match="not been committed"):
            s.frame_buffer()
        buf[:] = frame
        s.commit_frame()
        with pytest.raises(RbSensorError, match="no buffer acquired"):
            s.commit_frame()                       # one commit per acquire: the frame is not counted twice
        ref.set_observation(frame)
        idx = np.zeros(n, np.int32)
        a = s.loglikes_poses(poses, idx.copy(), update=True)
        b = ref.loglikes_poses(poses, idx.copy(), update=True)
        assert np.array_equal(a, b)
        # an acquire replaced by a plain set_observation: abandoned, its commit refused
        s.frame_buffer()
        s.set_observation(frame); ref.set_observation(frame)
        with pytest.raises(RbSensorError, match="no buffer acquired"):
            s.commit_frame()
        idx = np.arange(n, dtype=np.int32)
        a = s.loglikes_poses(poses, idx.copy(), update=True)
        b = ref.loglikes_poses(poses, idx.copy(), update=True)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0]])
def test_device_pointer_entry_points_on_a_group_handle(gpu_lib, ids):
    """rbs_set_observation_device / rbs_loglikes_device on a handle over several shards (VERDICT r2
    #8): frame, poses, parent slots and results are arrays on the first device in global particle
    order; every shard ingests the frame and pulls its slice in place, the caller's stream is
    ordered after all of them.  Same log-likelihoods and planes as the host-pointer calls on a
    single-device handle, over a resampled sequence with read-only and updating calls."""
    import torch
    n = 90                                              # not a multiple of the shard count: a short last shard
    om, cam, P = sc.make_scene(("m1_l2", "box12"), 160, 120, max_particles=n)
    rng = np.random.default_rng(11)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with RbSensor(om, cam, P, max_particles=n, precision="f64") as one, \
            RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=ids) as grp:
        one.reset(); grp.reset()
        idx = np.zeros(n, np.int32)
        for k in range(5):
            t = synth.truth_pose(2, frame=k)
            frame = synth.make_frame(one.render_depth(t), 120, 160, rng).astype(np.float32)
            poses = synth.particle_poses(t, n, rng)
            update = k != 2
            one.set_observation(frame)
            ref = one.loglikes_poses(poses, idx.copy(), update=update)
            with torch.cuda.stream(stream):
                d_frame = torch.from_numpy(frame).to(dev, non_blocking=False)
                d_poses = torch.from_numpy(np.ascontiguousarray(poses.reshape(n, -1))).to(dev)
                d_idx = torch.from_numpy(idx).to(dev)
                d_out = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
                grp.set_observation_device(d_frame.data_ptr(), stream.cuda_stream)
                grp.loglikes_device(d_poses.data_ptr(), d_idx.data_ptr(), n, update, d_out.data_ptr(), stream.cuda_stream)
                got = d_out.cpu().numpy()                 # a copy on `stream`: ordered after every shard
            assert np.array_equal(got, ref), (k, np.abs(got - ref).max())
            if update:
                for s_ in (0, n // 2, n - 1):
                    assert np.array_equal(one.get_occlusion(s_), grp.get_occlusion(s_))
                idx = rng.integers(0, n, n).astype(np.int32)
            grp.synchronize()


def test_a_fan_out_that_fails_half_way_poisons_the_handle_until_reset(gpu_lib, monkeypatch):
    """a call that fails on shard k after shards 0..k-1 were enqueued leaves the shards
    out of step.  (Injected with the library's test hook.)  The failing call reports its error,
    everything enqueued is drained, every later call is REFUSED with the original message -- no
    silently mismatched buffers -- and rbs_reset restores a working handle with the numbers of a
    fresh one."""
    from dbot_ros_amd.sensor import RbSensorError
    import os
    import subprocess
    import sys
    from dbot_ros_amd import _capi
    hooks = os.path.join(os.path.dirname(_capi.LIB_PATH), "librbsensor_mi355x_hooks.so")
    if os.path.abspath(_capi.LIB_PATH) != os.path.abspath(hooks):
        # the hook is compiled into the test build of the library only (`make hooks`): run this test in a process that loads it
        assert os.path.exists(hooks), "build() makes librbsensor_mi355x_hooks.so"
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                            __file__ + "::test_a_fan_out_that_fails_half_way_poisons_the_handle_until_reset"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, RBS_LIB_PATH=hooks))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        return
    n = 30
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    rng = np.random.default_rng(4)
    monkeypatch.setenv("RBS_TEST_FAULT", "1:2")        # the group's third rbs_loglikes fails on its second shard
    with RbSensor(om, cam, P, max_particles=n, precision="f64", device_ids=[0, 0, 0]) as grp:
        monkeypatch.delenv("RBS_TEST_FAULT")
        with RbSensor(om, cam, P, max_particles=n, precision="f64") as one:
            t = synth.truth_pose(1)
            frame = synth.make_frame(one.render_depth(t), 120, 160, rng).astype(np.float32)
            poses = synth.particle_poses(t, n, rng)

            def step(s, idx):
                s.set_observation(frame)
                return s.loglikes_poses(poses, idx, update=True)

            for s in (grp, one):
                s.reset()
            idx = np.zeros(n, np.int32)
            for k in range(2):
                a, b = step(grp, idx.copy()), step(one, idx.copy())
                assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
                idx = rng.integers(0, n, n).astype(np.int32)
            with pytest.raises(RbSensorError, match="injected fault"):
                step(grp, idx.copy())
            for call in (lambda: step(grp, idx.copy()), lambda: grp.set_observation(frame)):
                with pytest.raises(RbSensorError, match="undefined state.*injected fault"):
                    call()
            for s in (grp, one):
                s.reset()
            idx = np.zeros(n, np.int32)
            for k in range(3):
                a, b = step(grp, idx.copy()), step(one, idx.copy())
                assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
                idx = rng.integers(0, n, n).astype(np.int32)
            for s_ in (0, n - 1):
                assert np.array_equal(grp.get_occlusion(s_), one.get_occlusion(s_))


def test_a_communicator_that_cannot_be_made_falls_back_to_peer_copies(gpu_lib, monkeypatch):
    """First contact with a multi-GPU node must not end at ncclCommInitAll: a handle over several devices whose RCCL communicator cannot
    be made exchanges its log-likelihoods by peer copies instead (the devices can read each other: checked at create) and says so on
    stderr; RBS_REQUIRE_RCCL=1 makes it an error.  Provoked on one GPU with the hooks library: RBS_TEST_FORCE_RCCL takes the RCCL branch
    for device_ids = [0, 0], which ncclCommInitAll refuses."""
    import os
    import subprocess
    import sys
    from dbot_ros_amd import _capi
    from dbot_ros_amd.sensor import RbSensorError
    hooks = os.path.join(os.path.dirname(_capi.LIB_PATH), "librbsensor_mi355x_hooks.so")
    if os.path.abspath(_capi.LIB_PATH) != os.path.abspath(hooks):
        assert os.path.exists(hooks), "build() makes librbsensor_mi355x_hooks.so"
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-s",
                            __file__ + "::test_a_communicator_that_cannot_be_made_falls_back_to_peer_copies"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, RBS_LIB_PATH=hooks))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert "RCCL is not used for this handle" in r.stderr + r.stdout
        return
    n, nb = 64, 1
    om, cam, P = sc.make_scene(("m1_l2",), 160, 120, max_particles=n)
    rng = np.random.default_rng(3)
    with RbSensor(om, cam, P, max_particles=1) as r:
        frames = [synth.make_frame(r.render_depth(synth.truth_pose(nb, frame=k)), 120, 160, rng, occluder=False).astype(np.float32)
                  for k in range(1, 5)]
    randomness = [(rng.standard_normal((nb, n, 6)), rng.random((nb, n))) for _ in frames]
    e1, _, r1 = _run_tracker(om, cam, P, n, nb, frames, None, randomness)
    monkeypatch.setenv("RBS_TEST_FORCE_RCCL", "1")
    e2, _, r2 = _run_tracker(om, cam, P, n, nb, frames, [0, 0], randomness)      # the communicator fails, the tracker runs
    assert r1 == r2 and np.abs(e1 - e2).max() <= 1e-9
    monkeypatch.setenv("RBS_REQUIRE_RCCL", "1")
    with pytest.raises(RbSensorError, match="ncclCommInitAll"):
        RbSensor(om, cam, P, max_particles=n, device_ids=[0, 0])
