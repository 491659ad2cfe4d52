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


def replay_dataset(params, dataset, mesh_package_path, initial_states, device_id=0, seed=0, max_frames=None,
                   look_ahead=False):
    """Run the tracker over a recorded TrackingDataset (dbot_ros_amd.dataset) the way
    the node runs over live topics: K from the bag's camera_info (frame 0, as GetCameraMatrix does,
    R:source/dbot_ros/util/tracking_dataset.cpp:157-164), every depth image sub-sampled by
    `downsampling_factor` (R:source/dbot_ros/util/ros_interface.h:152-168) and handed to
    tracker.track (R:source/dbot_ros/object_tracker_ros.hpp:44-49).
    look_ahead: a recorded sequence has the next frame at hand, so frame k+1 is submitted before
    frame k's estimate is collected (tracker.submit / tracker.result: two frames in flight, the
    same numbers); False drives tracker.track frame by frame as a live camera would.
    Returns (estimates [frames, parts*12], wall seconds of the tracking loop)."""
    import time
    tracker, object_model, camera_data, _ = build_particle_tracker(params, dataset.get_camera_matrix(0),
                                                                   mesh_package_path, device_id=device_id, seed=seed)
    f = int(params["downsampling_factor"])
    try:
        tracker.initialize(initial_states)
        n = dataset.size() if max_frames is None else min(max_frames, dataset.size())
        frames = [dataset.frame_vector(i, f) for i in range(n)]      # host decode outside the timed loop
        ests = []
        t0 = time.perf_counter()
        if look_ahead and hasattr(tracker, "submit") and frames:
            tracker.submit(frames[0])
            for fr in frames[1:]:
                tracker.submit(fr)
                ests.append(tracker.result())
            ests.append(tracker.result())
        else:
            for fr in frames:
                ests.append(tracker.track(fr))
        wall = time.perf_counter() - t0
    finally:
        tracker.close()
        tracker.sensor.close()
    return np.array(ests), wall
