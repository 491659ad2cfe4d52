"""ROS-free equivalent of the reference's particle_tracker node assembly
(R:source/dbot_ros/tracker/particle_tracker_node.cpp:38-252): reads the SAME three YAML files
`particle_tracker.launch` loads into the node's private namespace
(R:launch/particle_tracker.launch:13-15 -> R:config/particle_tracker.yaml, camera.yaml, object.yaml),
with the same keys, and assembles object model -> camera data -> transition builder -> sensor
builder -> tracker in the node's order.  What ROS supplied at run time is passed in explicitly:
the native camera matrix (camera_info topic), the mesh root (ros::package::getPath) and the
initial poses (interactive marker)."""
import os

import numpy as np
import yaml

from .objloader import ObjectResourceIdentifier, SimpleWavefrontObjectModelLoader
from .sensor import CameraData, ObjectModel, RbSensorBuilder
from .tracker import DeviceParticleTracker, ObjectTransitionBuilder, ParticleTracker, ParticleTrackerBuilder


def load_rosparams(*yaml_paths):
    """Merge YAML files the way `rosparam load` into one namespace does (top-level keys)."""
    tree = {}
    for p in yaml_paths:
        with open(p) as f:
            tree.update(yaml.safe_load(f) or {})
    return tree


def build_particle_tracker(params, native_camera_matrix, mesh_package_path, device_id=0, rng=None,
                           device_filter=True, seed=0):
    """params: merged rosparam tree.  Returns (tracker, object_model, camera_data, ori)."""
    pre = params["particle_filter"]
    # ---- object model (node :78-97)
    obj = params["object"]
    ori = ObjectResourceIdentifier(mesh_package_path, obj["directory"], obj["meshes"])
    vs, ts = SimpleWavefrontObjectModelLoader(ori).load()
    object_model = ObjectModel(vs, ts, center=bool(pre["center_object_frame"]))
    # ---- camera data (node :102-121, ros_camera_data_provider.cpp:66-76)
    camera_data = CameraData.from_native(native_camera_matrix, int(params["resolution"]["width"]),
                                         int(params["resolution"]["height"]), int(params["downsampling_factor"]))
    # ---- state transition (node :138-159)
    params_state = ObjectTransitionBuilder.Parameters.from_rosparam(params, part_count=ori.count_meshes())
    transition = ObjectTransitionBuilder(params_state).build()
    # ---- observation model (node :164-203)
    params_obsrv = RbSensorBuilder.Parameters.from_rosparam(params)
    sensor = RbSensorBuilder(object_model, camera_data, params_obsrv, device_id=device_id).build()
    # ---- filter & tracker (node :208-218)
    params_tracker = ParticleTrackerBuilder.Parameters.from_rosparam(params, params_obsrv.sample_count)
    if device_filter:
        tracker = DeviceParticleTracker(transition, sensor, object_model, params_tracker, rng,
                                        device_rng=rng is None, seed=seed)
    else:
        tracker = ParticleTracker(transition, sensor, object_model, params_tracker, rng)
    return tracker, object_model, camera_data, ori


def to_eigen_vector(native_image, downsampling_factor):
    """ri::to_eigen_vector (R:source/dbot_ros/util/ros_interface.h:152-168) on the host."""
    img = np.asarray(native_image)
    f = int(downsampling_factor)
    rows, cols = img.shape[0] // f, img.shape[1] // f
    return np.ascontiguousarray(img[: rows * f: f, : cols * f: f]).ravel()
