"""The ROS-free node assembly: the reference's own YAML key tree -> tracker, driven on a
synthetic sequence at the reference's default operating point (640x480 / 8 = 80x60)."""
import numpy as np
import pytest
import yaml

from dbot_ros_amd import node, objloader, pose, synth

# verbatim key structure of R:config/particle_tracker.yaml, R:config/camera.yaml, R:config/object.yaml
FILTER_YAML = """
particle_filter:
  use_gpu: true
  cpu: {sample_count: 100}
  gpu:
    sample_count: 400
    use_custom_shaders: false
    vertex_shader_file: /path/to/custom/vertex_shader.vertexshader
    fragment_shader_file: /path/to/custom/fragment_shader.fragmentshader
    geometry_shader_file: none
  moving_average_update_rate: 1.0
  center_object_frame: true
  max_kl_divergence: 2.0
  observation:
    occlusion: {initial_occlusion_prob: 0.1, p_occluded_visible: 0.1, p_occluded_occluded: 0.7}
    kinect: {tail_weight: 0.01, model_sigma: 0.003, sigma_factor: 0.0014247}
  object_transition:
    linear_sigma_x: 0.0025
    linear_sigma_y: 0.0025
    linear_sigma_z: 0.0025
    angular_sigma_x: 0.02
    angular_sigma_y: 0.02
    angular_sigma_z: 0.02
    velocity_factor: 0.8
  object_color: {R: 10, G: 200, B: 50}
"""
CAMERA_YAML = """
depth_image_topic: /XTION/depth/image
camera_info_topic: /XTION/depth/camera_info
downsampling_factor: 8
resolution: {width: 640, height: 480}
"""
OBJECT_YAML = """
object:
  package: object_meshes
  directory: object_models
  meshes: [ part.obj ]
"""


def _write(tmp_path):
    paths = []
    for name, txt in (("particle_tracker.yaml", FILTER_YAML), ("camera.yaml", CAMERA_YAML), ("object.yaml", OBJECT_YAML)):
        p = tmp_path / name
        p.write_text(txt)
        paths.append(str(p))
    (tmp_path / "object_models").mkdir()
    v, t = synth.mesh_m1(level=3)
    objloader.write_obj(tmp_path / "object_models" / "part.obj", v + np.array([0.2, 0.1, -0.05]), t)
    return paths


def test_rosparam_tree_and_subsampling(tmp_path):
    paths = _write(tmp_path)
    tree = node.load_rosparams(*paths)
    assert tree["particle_filter"]["gpu"]["sample_count"] == 400 and tree["downsampling_factor"] == 8
    assert tree["object"]["meshes"] == ["part.obj"]
    img = np.arange(480 * 640, dtype=np.float32).reshape(480, 640)
    sub = node.to_eigen_vector(img, 8)
    assert sub.shape == (60 * 80,) and sub[1] == img[0, 8] and sub[80] == img[8, 0]


def test_extension_keys_next_to_sample_count(tmp_path):
    """particle_filter/gpu/{likelihood_precision, occlusion_mode, devices}: optional -- the reference's files
    have none of them and give the library's choices -- read into the same fields as dbot_amd::RbSensorBuilder<State>::Parameters."""
    from dbot_ros_amd import RbSensorBuilder
    tree = node.load_rosparams(*_write(tmp_path))
    p = RbSensorBuilder.Parameters.from_rosparam(tree)
    assert (p.likelihood_precision, p.occlusion_mode, p.devices) == ("", "", None)
    tree["particle_filter"]["gpu"].update(likelihood_precision="f64", occlusion_mode="reference", devices=[0, 1])
    p = RbSensorBuilder.Parameters.from_rosparam(tree)
    assert (p.likelihood_precision, p.occlusion_mode, p.devices) == ("f64", "reference", [0, 1])
    tree["particle_filter"]["gpu"]["occlusion_mode"] = "exactly"
    with pytest.raises(ValueError):
        RbSensorBuilder.Parameters.from_rosparam(tree)


REFERENCE_CONFIG = "/root/reference/config"


@pytest.mark.skipif(not __import__("os").path.isdir(REFERENCE_CONFIG),
                    reason="the reference's own config files exist only in the build container (never on the GPU box)")
def test_the_references_own_yaml_files_parse_unchanged():
    """SURVEY 8b "rosparam keys must keep working verbatim": the files the reference's launch file loads
    (R:launch/particle_tracker.launch:13-15 -> R:config/particle_tracker.yaml, camera.yaml, object.yaml),
    read where they lie, give every parameter object of the node assembly its values -- no key renamed,
    none missing.  (Values below are the reference's shipped defaults.)"""
    import os
    from dbot_ros_amd import CameraData, RbSensorBuilder
    from dbot_ros_amd.tracker import ObjectTransitionBuilder, ParticleTrackerBuilder
    tree = node.load_rosparams(*(os.path.join(REFERENCE_CONFIG, f) for f in ("particle_tracker.yaml", "camera.yaml", "object.yaml")))
    ps = RbSensorBuilder.Parameters.from_rosparam(tree)
    assert ps.use_gpu is True and ps.sample_count == tree["particle_filter"]["gpu"]["sample_count"] == 2000
    assert (ps.occlusion.p_occluded_visible, ps.occlusion.p_occluded_occluded, ps.occlusion.initial_occlusion_prob) == (0.1, 0.7, 0.1)
    assert (ps.kinect.tail_weight, ps.kinect.model_sigma, ps.kinect.sigma_factor) == (0.01, 0.003, 0.0014247)
    assert ps.use_custom_shaders is False and ps.geometry_shader_file == "none"      # accepted and ignored
    tree["particle_filter"]["use_gpu"] = False
    assert RbSensorBuilder.Parameters.from_rosparam(tree).sample_count == tree["particle_filter"]["cpu"]["sample_count"]
    pt = ObjectTransitionBuilder.Parameters.from_rosparam(tree, part_count=len(tree["object"]["meshes"]))
    assert (pt.linear_sigma_x, pt.angular_sigma_z, pt.velocity_factor) == (0.0025, 0.02, tree["particle_filter"]["object_transition"]["velocity_factor"])
    pk = ParticleTrackerBuilder.Parameters.from_rosparam(tree, ps.sample_count)
    assert pk.evaluation_count == 2000 and pk.max_kl_divergence == tree["particle_filter"]["max_kl_divergence"]
    assert pk.center_object_frame is True and pk.moving_average_update_rate == 1.0
    f = int(tree["downsampling_factor"])
    cam = CameraData.from_native(synth.camera_matrix(640, 480), int(tree["resolution"]["width"]), int(tree["resolution"]["height"]), f)
    assert f == 8 and (cam.cols, cam.rows) == (80, 60)
    assert tree["object"]["package"] and tree["object"]["directory"] and len(tree["object"]["meshes"]) >= 1


@pytest.mark.gpu
def test_node_assembly_tracks_at_the_reference_operating_point(tmp_path, gpu_lib):
    paths = _write(tmp_path)
    tree = node.load_rosparams(*paths)
    K = synth.camera_matrix(640, 480)
    tracker, om, cam, ori = node.build_particle_tracker(tree, K, str(tmp_path), seed=3)
    assert (cam.rows, cam.cols) == (60, 80) and tracker.n == 400 and ori.count_meshes() == 1
    v0 = synth.mesh_m1(level=3)[0] + np.array([0.2, 0.1, -0.05])   # mesh was written off-centre
    assert np.allclose(om.centers[0], v0.mean(axis=0), atol=1e-12)
    sensor = tracker.sensor
    # native-resolution frames from a full-resolution twin of the sensor, ingested with the reference's sub-sampling
    from dbot_ros_amd import CameraData, RbSensor, RbSensorBuilder
    full = RbSensor(om, CameraData(K, 480, 640), RbSensorBuilder.Parameters(sample_count=1), max_particles=1)
    rng = np.random.default_rng(0)

    def truth_state(k):
        Rt = synth.truth_pose(1, frame=k)[0]
        s = np.zeros(12)
        s[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
        s[0:3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]
        return s

    tracker.initialize([truth_state(0)])
    errs = []
    for k in range(1, 16):
        native = synth.make_frame(full.render_depth(synth.truth_pose(1, frame=k)), 480, 640, rng, occluder=False)
        sensor.set_observation_native(native.reshape(480, 640), tree["downsampling_factor"])
        assert np.array_equal(sensor.get_observation(), node.to_eigen_vector(native.reshape(480, 640), 8), equal_nan=True)
        est = tracker.track(sensor.get_observation())
        errs.append(np.linalg.norm(est[0:3] - truth_state(k)[0:3]))
    full.close()
    assert max(errs[-5:]) < 0.025, errs   # 80x60: one pixel is 8.8 mm at 0.7 m -> a few pixels
    tracker.close()
    sensor.close()


@pytest.mark.gpu
def test_node_assembly_in_occlusion_mode_reference(tmp_path, gpu_lib):
    """particle_filter/gpu/occlusion_mode: reference in the node's YAML -> the sensor the assembly builds keeps the reference's own
    occlusion bookkeeping (its slots are stamped planes: no raw float plane to hand out) and tracks like the default."""
    from dbot_ros_amd import CameraData, RbSensor, RbSensorBuilder, RbSensorError
    paths = _write(tmp_path)
    K = synth.camera_matrix(640, 480)
    ests = {}
    for mode in ("reference", "device"):
        tree = node.load_rosparams(*paths)
        tree["particle_filter"]["gpu"]["occlusion_mode"] = mode
        tracker, om, cam, _ = node.build_particle_tracker(tree, K, str(tmp_path), seed=3)
        sensor = tracker.sensor
        if mode == "reference":
            with pytest.raises(RbSensorError):
                sensor.occlusion_device_ptr(0)
        else:
            assert sensor.occlusion_device_ptr(0)
        full = RbSensor(om, CameraData(K, 480, 640), RbSensorBuilder.Parameters(sample_count=1), max_particles=1)
        rng = np.random.default_rng(0)
        Rt0 = synth.truth_pose(1, frame=0)[0]
        s0 = np.zeros(12)
        s0[3:6] = pose.matrix_to_rotvec(Rt0[:9].reshape(3, 3))
        s0[0:3] = Rt0[9:] - Rt0[:9].reshape(3, 3) @ om.centers[0]
        tracker.initialize([s0])
        out = []
        for k in range(1, 9):
            native = synth.make_frame(full.render_depth(synth.truth_pose(1, frame=k)), 480, 640, rng, occluder=False)
            out.append(tracker.track(node.to_eigen_vector(native.reshape(480, 640), tree["downsampling_factor"])))
        ests[mode] = np.array(out)
        full.close(); tracker.close(); sensor.close()
    # the two bookkeepings differ by ~1e-8 in a log-likelihood: the same track (a resampling draw on the edge may pick a neighbour)
    assert np.abs(ests["reference"][:, 0:3] - ests["device"][:, 0:3]).max() < 5e-3


@pytest.mark.gpu
def test_replay_of_a_recorded_dataset(tmp_path, gpu_lib):
    """SURVEY 8 f4 end to end: a recorded dataset directory (measurements.bag + ground_truth.txt,
    written here by the ROS-free writer from native 640x480 frames) is loaded back and tracked
    through the node assembly at the reference's operating point; the estimates follow the
    recorded ground truth."""
    from dbot_ros_amd import CameraData, RbSensor, RbSensorBuilder
    from dbot_ros_amd import dataset as ds
    paths = _write(tmp_path)
    tree = node.load_rosparams(*paths)
    K = synth.camera_matrix(640, 480)
    vs, ts = objloader.SimpleWavefrontObjectModelLoader(
        objloader.ObjectResourceIdentifier(str(tmp_path), "object_models", ["part.obj"])).load()
    from dbot_ros_amd import ObjectModel
    om = ObjectModel(vs, ts, center=True)

    def truth_state(k):
        Rt = synth.truth_pose(1, frame=k)[0]
        s = np.zeros(12)
        s[3:6] = pose.matrix_to_rotvec(Rt[:9].reshape(3, 3))
        s[0:3] = Rt[9:] - Rt[:9].reshape(3, 3) @ om.centers[0]
        return s

    rng = np.random.default_rng(0)
    rec = ds.TrackingDataset(tmp_path / "recording", load=False)
    with RbSensor(om, CameraData(K, 480, 640), RbSensorBuilder.Parameters(sample_count=1), max_particles=1) as full:
        for k in range(1, 13):
            native = synth.make_frame(full.render_depth(synth.truth_pose(1, frame=k)), 480, 640, rng, occluder=False)
            stamp = ds.Stamp.from_sec(1500000000.0 + k / 30.0)
            rec.add_frame(ds.Image(native.reshape(480, 640), stamp, seq=k), ds.CameraInfo(K, 480, 640, stamp, seq=k),
                          ground_truth=truth_state(k))
    rec.store()
    data = ds.TrackingDataset(tmp_path / "recording")
    assert data.size() == 12 and np.array_equal(data.get_camera_matrix(5), K)
    ests, wall = node.replay_dataset(tree, data, str(tmp_path), [truth_state(0)], seed=3)
    assert ests.shape == (12, 12) and wall > 0
    err = [np.linalg.norm(ests[i, 0:3] - data.get_ground_truth(i)[0:3]) for i in range(12)]
    assert max(err[-4:]) < 0.025, err
    # a recording has the next frame at hand: one frame of look-ahead gives the very same estimates
    ests2, _ = node.replay_dataset(tree, data, str(tmp_path), [truth_state(0)], seed=3, look_ahead=True)
    assert np.array_equal(ests, ests2)
