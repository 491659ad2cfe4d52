"""Host-side mirror of the reference's plugin surface: state layout, delta (+) default pose
composition, rosparam keys -> RbSensorBuilder.Parameters, camera down-sampling."""
import numpy as np
import yaml

from dbot_ros_amd import CameraData, ObjectModel, RbSensorBuilder, pose, synth

# same key tree as R:config/particle_tracker.yaml (values are the reference defaults)
PARTICLE_TRACKER_YAML = """
particle_filter:
  use_gpu: true
  cpu: {sample_count: 100}
  gpu:
    sample_count: 2000
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
"""


def test_parameters_from_rosparam_tree():
    tree = yaml.safe_load(PARTICLE_TRACKER_YAML)
    p = RbSensorBuilder.Parameters.from_rosparam(tree)
    assert p.use_gpu and p.sample_count == 2000
    assert (p.occlusion.p_occluded_visible, p.occlusion.p_occluded_occluded,
            p.occlusion.initial_occlusion_prob) == (0.1, 0.7, 0.1)
    assert (p.kinect.tail_weight, p.kinect.model_sigma, p.kinect.sigma_factor) == (0.01, 0.003, 0.0014247)
    assert p.delta_time == 1.0 / 30.0          # hard-coded by the node, particle_tracker_node.cpp:189
    assert p.geometry_shader_file == "none"    # GL knobs accepted and carried, never used
    tree["particle_filter"]["use_gpu"] = False
    assert RbSensorBuilder.Parameters.from_rosparam(tree).sample_count == 100


def test_camera_downsampling_divides_top_rows():
    """ros_camera_data_provider.cpp:72 and ros_interface.h:158-159."""
    K = [[570.3, 0, 319.5], [0, 570.3, 239.5], [0, 0, 1]]
    cam = CameraData.from_native(K, 640, 480, 8)
    assert (cam.rows, cam.cols) == (60, 80)
    assert np.allclose(cam.camera_matrix, [[570.3 / 8, 0, 319.5 / 8], [0, 570.3 / 8, 239.5 / 8], [0, 0, 1]])


def test_object_model_centering():
    v, f = synth.mesh_m3()
    v = v + np.array([0.3, -0.2, 0.1])
    om = ObjectModel([v], [f], center=True)
    assert np.allclose(om.vertices[0].mean(axis=0), 0.0, atol=1e-15)
    assert np.allclose(om.centers[0], v.mean(axis=0))
    assert np.allclose(ObjectModel([v], [f], center=False).vertices[0], v)


def test_rotation_helpers_roundtrip():
    rng = np.random.default_rng(0)
    for _ in range(20):
        rv = rng.normal(size=3) * rng.choice([1e-12, 0.1, 1.0, 3.0])
        R = pose.rotvec_to_matrix(rv)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and np.isclose(np.linalg.det(R), 1.0)
        assert np.allclose(pose.rotvec_to_matrix(pose.matrix_to_rotvec(R)), R, atol=1e-12)
    # rotation about z by 90 degrees maps x to y
    R = pose.rotvec_to_matrix([0, 0, np.pi / 2])
    assert np.allclose(R @ [1, 0, 0], [0, 1, 0], atol=1e-15)


def test_delta_composition():
    """R = R(delta) R(default), t = t(delta) + t(default); velocities ignored."""
    rng = np.random.default_rng(1)
    nb = 2
    default = rng.normal(size=nb * 12) * 0.3
    deltas = rng.normal(size=(5, nb * 12)) * 0.05
    P = pose.compose_with_default(deltas, default, nb)
    assert P.shape == (5, nb, 12)
    for i in range(5):
        for b in range(nb):
            d, z = deltas[i, 12 * b:12 * b + 12], default[12 * b:12 * b + 12]
            R = pose.rotvec_to_matrix(d[3:6]) @ pose.rotvec_to_matrix(z[3:6])
            assert np.allclose(P[i, b, :9].reshape(3, 3), R, atol=1e-15)
            assert np.allclose(P[i, b, 9:], d[:3] + z[:3])
    zero = pose.compose_with_default(np.zeros((1, nb * 12)), default, nb)
    assert np.allclose(zero[0], pose.states_to_Rt(default[None], nb)[0])


def test_synthetic_meshes_have_the_advertised_sizes():
    assert synth.mesh_m1()[1].shape == (5120, 3) and synth.mesh_m1()[0].shape == (2562, 3)
    assert synth.mesh_m3()[1].shape == (5120, 3)
    assert synth.mesh_m4()[1].shape == (50880, 3)
    assert 5000 < len(synth.mesh_m2()[1]) < 5500
    for m in (synth.mesh_m1, synth.mesh_m2, synth.mesh_m3, synth.mesh_m4, synth.mesh_box12):
        v, f = m()
        assert f.min() == 0 and f.max() == len(v) - 1 and np.isfinite(v).all()
