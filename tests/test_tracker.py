"""End-to-end tracker loop (transition -> loglikes -> weights -> KL -> resample -> mean) around
the sensor: tracks a synthetic object moving 2 mm and 1 degree per frame."""
import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import RbSensor, pose, synth
from dbot_ros_amd.tracker import (ObjectTransitionBuilder, ParticleTracker, ParticleTrackerBuilder)


def _truth_state(frame, centers):
    """State (camera-frame pose of the ORIGINAL mesh frame) of the synthetic truth."""
    Rt = synth.truth_pose(1, frame=frame)[0]
    R, t = Rt[:9].reshape(3, 3), Rt[9:]
    s = np.zeros(12)
    s[3:6] = pose.matrix_to_rotvec(R)
    s[0:3] = t - R @ centers[0]   # truth_pose places the CENTRED mesh
    return s


def _run(sensor, om, cam, n, frames_n, renderer, seed=0):
    tp = ParticleTrackerBuilder.Parameters(evaluation_count=n, max_kl_divergence=2.0)
    tr = ParticleTracker(ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters()).build(), sensor, om, tp,
                         np.random.default_rng(seed))
    init = _truth_state(0, om.centers)
    init[0:3] += [0.004, -0.003, 0.005]      # start a few mm off
    tr.initialize([init])
    rng = np.random.default_rng(100 + seed)
    errs = []
    for k in range(1, frames_n + 1):
        truth = synth.truth_pose(1, frame=k)
        frame = synth.make_frame(renderer(truth), cam.rows, cam.cols, rng, occluder=False)
        est = tr.track(frame)
        ref = _truth_state(k, om.centers)
        dR = pose.rotvec_to_matrix(est[3:6]).T @ pose.rotvec_to_matrix(ref[3:6])
        errs.append((np.linalg.norm(est[0:3] - ref[0:3]), np.linalg.norm(pose.matrix_to_rotvec(dR))))
    return np.array(errs), tr


def test_transition_is_a_velocity_random_walk():
    p = ObjectTransitionBuilder.Parameters(part_count=2)
    t = ObjectTransitionBuilder(p).build()
    s = np.zeros((3, 24))
    s[:, 6:9] = 0.01
    n = np.ones((3, 6))
    out = t.apply(s, n, 0)
    assert np.allclose(out[:, 6:9], 0.8 * 0.01 + 0.0025) and np.allclose(out[:, 0:3], 0.8 * 0.01 + 0.0025)
    assert np.allclose(out[:, 9:12], 0.02) and np.allclose(out[:, 3:6], 0.02)
    assert np.array_equal(out[:, 12:], s[:, 12:]) and np.array_equal(s[:, 0:3], np.zeros((3, 3)))


def test_tracker_follows_the_object_cpu_oracle():
    n = 96
    om, cam, P = sc.make_scene(("m1_l2",), 80, 60, max_particles=n)
    o = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    errs, tr = _run(o, om, cam, n, 8, o.render_depth)
    assert errs[-3:, 0].max() < 0.012 and errs[-3:, 1].max() < 0.2, errs
    assert np.isfinite(errs).all()


@pytest.mark.gpu
def test_tracker_follows_the_object_gpu(gpu_lib):
    n = 2000
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    with RbSensor(om, cam, P, max_particles=n) as g:
        errs, tr = _run(g, om, cam, n, 20, g.render_depth)
    assert errs[-5:, 0].max() < 0.004 and errs[-5:, 1].max() < 0.06, errs
    assert tr.n_resamplings >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("meshes,n,cols,rows", [(("m1_l2",), 64, 160, 120), (("m1_l2", "box12"), 96, 160, 120),
                                                 (("m1",), 500, 640, 480)])
def test_device_tracker_matches_host_tracker(gpu_lib, meshes, n, cols, rows):
    """f1/f2 on the device (rbs_tracker_*) against the host mirror (dbot_ros_amd.tracker) driving
    the same product sensor, with identical host-supplied normals/uniforms: estimated state,
    particle cloud, weights, slot map and resampling decisions agree frame after frame."""
    from dbot_ros_amd.tracker import DeviceParticleTracker
    nb = len(meshes)
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    tp = ParticleTrackerBuilder.Parameters(evaluation_count=n, max_kl_divergence=2.0)
    with RbSensor(om, cam, P, max_particles=n) as s_host, RbSensor(om, cam, P, max_particles=n) as s_dev:
        tr = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=nb)).build()
        host = ParticleTracker(tr, s_host, om, tp, np.random.default_rng(5))
        dev = DeviceParticleTracker(tr, s_dev, om, tp, np.random.default_rng(5))
        assert host.n == dev.n == n // nb
        init = np.zeros(12 * nb)
        for b in range(nb):
            Rt = synth.truth_pose(nb, frame=0)[b]
            init[12 * b + 3:12 * b + 6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
            init[12 * b:12 * b + 3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[b]
        host.initialize([init])
        dev.initialize([init])
        rng = np.random.default_rng(77)
        resampled = 0
        for k in range(1, 7):
            frame = synth.make_frame(s_host.render_depth(synth.truth_pose(nb, frame=k)), rows, cols, rng,
                                     occluder=False)
            normals, uniforms = host.draw_randomness()
            eh = host.track(frame, normals, uniforms)
            ed = dev.track(frame, normals, uniforms)
            assert np.abs(eh - ed).max() <= 1e-9, (k, np.abs(eh - ed).max())
            p, w, idx = dev.get_state()
            assert np.abs(p - host.particles).max() <= 1e-9
            assert np.abs(w - host.log_weights).max() <= 1e-6 * max(1.0, np.abs(host.log_weights).max())
            assert np.array_equal(idx, host.indices)
            assert dev.n_resamplings == host.n_resamplings
        assert host.n_resamplings >= 1
        dev.close()


@pytest.mark.gpu
def test_device_tracker_with_device_rng_tracks(gpu_lib):
    from dbot_ros_amd.tracker import DeviceParticleTracker
    n = 2000
    om, cam, P = sc.make_scene(("m1",), 640, 480, max_particles=n)
    with RbSensor(om, cam, P, max_particles=n) as g:
        tp = ParticleTrackerBuilder.Parameters(evaluation_count=n)
        tr = DeviceParticleTracker(ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters()).build(), g, om, tp,
                                   device_rng=True, seed=1234)
        init = _truth_state(0, om.centers)
        init[0:3] += [0.004, -0.003, 0.005]
        tr.initialize([init])
        rng = np.random.default_rng(3)
        errs = []
        for k in range(1, 21):
            frame = synth.make_frame(g.render_depth(synth.truth_pose(1, frame=k)), 480, 640, rng, occluder=False)
            est = tr.track(frame)
            ref = _truth_state(k, om.centers)
            errs.append(np.linalg.norm(est[0:3] - ref[0:3]))
        assert max(errs[-5:]) < 0.004, errs
        assert tr.n_resamplings >= 1
        tr.close()


def _model_init(om, nb):
    init = np.zeros(12 * nb)
    for b in range(nb):
        Rt = synth.truth_pose(nb, frame=0)[b]
        init[12 * b + 3:12 * b + 6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
        init[12 * b:12 * b + 3] = Rt[9:]            # model coordinates: pose of the centred mesh
    return init


@pytest.mark.parametrize("meshes,n", [(("m1_l2",), 48), (("m1_l2", "box12"), 64)])
def test_host_tracker_mirror_matches_the_c_oracle_tracker(meshes, n):
    """The Python host mirror (dbot_ros_amd.tracker) against the C restatement of the same loop
    (oracle/tracker_oracle.c), both driving oracle sensors, same normals/uniforms."""
    nb = len(meshes)
    om, cam, P = sc.make_scene(meshes, 80, 60, max_particles=n)
    tp = ParticleTrackerBuilder.Parameters(evaluation_count=n, center_object_frame=False)
    per = n // nb
    o1 = ob.Oracle(om, cam, P, max_particles=per, mode=ob.EAGER)
    o2 = ob.Oracle(om, cam, P, max_particles=per, mode=ob.EAGER)
    trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=nb)).build()
    host = ParticleTracker(trans, o1, om, tp, np.random.default_rng(1))
    ref = ob.OracleTracker(o2, per, trans.sigma, trans.vf, tp.max_kl_divergence)
    init = _model_init(om, nb)
    host.initialize([init])
    ref.initialize(init)
    rng = np.random.default_rng(9)
    for k in range(1, 5):
        frame = synth.make_frame(o1.render_depth(synth.truth_pose(nb, frame=k)), 60, 80, rng, occluder=False)
        normals, uniforms = host.draw_randomness()
        eh = host.track(frame, normals, uniforms)
        er, nres = ref.track(frame, normals, uniforms)
        assert np.abs(eh - er).max() <= 1e-12
        p, w, idx = ref.get_state()
        assert np.abs(p - host.particles).max() <= 1e-12 and np.array_equal(idx, host.indices)
        assert np.abs(w - host.log_weights).max() <= 1e-9 and nres == host.n_resamplings
    assert host.n_resamplings >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("f64", 1e-9), ("f32", 1e-6)])
@pytest.mark.parametrize("meshes,n,cols,rows", [(("m1_l2",), 64, 160, 120), (("m1_l2", "box12"), 96, 160, 120)])
def test_device_tracker_matches_the_c_oracle_tracker(gpu_lib, meshes, n, cols, rows, precision, tol):
    """rbs_tracker_* on the GPU against oracle/tracker_oracle.c over the oracle sensor (device
    rule), same host-supplied randomness: the checker is entirely under oracle/."""
    from dbot_ros_amd.tracker import DeviceParticleTracker
    nb = len(meshes)
    om, cam, P = sc.make_scene(meshes, cols, rows, max_particles=n)
    tp = ParticleTrackerBuilder.Parameters(evaluation_count=n, center_object_frame=False)
    per = n // nb
    orc = ob.Oracle(om, cam, P, max_particles=per, mode=ob.EAGER)
    trans = ObjectTransitionBuilder(ObjectTransitionBuilder.Parameters(part_count=nb)).build()
    ref = ob.OracleTracker(orc, per, trans.sigma, trans.vf, tp.max_kl_divergence)
    with RbSensor(om, cam, P, max_particles=per, precision=precision) as s:
        dev = DeviceParticleTracker(trans, s, om, tp, np.random.default_rng(2))
        init = _model_init(om, nb)
        dev.initialize([init])
        ref.initialize(init)
        rng = np.random.default_rng(10)
        for k in range(1, 6):
            frame = synth.make_frame(orc.render_depth(synth.truth_pose(nb, frame=k)), rows, cols, rng, occluder=False)
            normals, uniforms = dev.draw_randomness()
            ed = dev.track(frame, normals, uniforms)
            er, nres = ref.track(frame, normals, uniforms)
            assert np.abs(ed - er).max() <= tol, (k, np.abs(ed - er).max())
            pd, wd, idd = dev.get_state()
            pr, wr, idr = ref.get_state()
            # (F32: the same parents as long as no uniform falls within ~1e-7 of a cdf step)
            assert np.abs(pd - pr).max() <= tol and np.array_equal(idd, idr) and dev.n_resamplings == nres
        assert nres >= 1
        dev.close()
