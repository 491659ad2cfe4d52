"""Host-side mirror (Python flavour, used by tests and bench.py) of the plugin surface
dbot_ros drives for this path -- same names and argument meaning as the reference's call sites:

  ObjectModel, CameraData                R:source/dbot_ros/tracker/particle_tracker_node.cpp:89-121
  RbSensorBuilder.Parameters             R:...particle_tracker_node.cpp:164-199
  RbSensorBuilder(object_model, camera_data, params).build()   R:...particle_tracker_node.cpp:201-203
  RbSensor.set_observation / loglikes / reset   driven from tracker_->track,
                                         R:source/dbot_ros/object_tracker_ros.hpp:49

Everything numeric happens in librbsensor_mi355x.so through the C-ABI; the C++ flavour of the
same mirror is include/dbot_amd/rb_sensor_builder.hpp.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _capi
from .pose import compose_with_default


class RbSensorError(RuntimeError):
    """Raised for any non-zero status of the C-ABI (the C++ shim throws std::runtime_error,
    which the reference's service thread already catches,
    R:source/dbot_ros/tracker/object_tracker_service_node.cpp:252-255)."""

    def __init__(self, code, message):
        super().__init__(f"[rbs {code}] {message}")
        self.code = code


class ObjectModel:
    """Triangle meshes of the tracked rigid bodies (dbot::ObjectModel as constructed at
    R:source/dbot_ros/tracker/particle_tracker_node.cpp:94-97).

    center=True re-expresses every part around the mean of its vertices
    (center_object_frame, R:config/particle_tracker.yaml:27-30)."""

    def __init__(self, vertices, triangles, center=True):
        self.vertices, self.triangles, self.centers = [], [], []
        for v, t in zip(vertices, triangles):
            v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 3)
            t = np.ascontiguousarray(t, dtype=np.int32).reshape(-1, 3)
            c = v.mean(axis=0) if center else np.zeros(3)
            self.vertices.append(v - c)
            self.triangles.append(t)
            self.centers.append(c)

    @property
    def count_parts(self):
        return len(self.vertices)


@dataclass
class CameraData:
    """dbot::CameraData as the sensor sees it: intrinsics already divided by the
    down-sampling factor (R:source/dbot_ros/util/ros_camera_data_provider.cpp:66-76) and the
    evaluated resolution (R:config/camera.yaml:5-9)."""
    camera_matrix: np.ndarray
    rows: int
    cols: int
    downsampling_factor: int = 1

    @classmethod
    def from_native(cls, K, width, height, downsampling_factor):
        K = np.array(K, dtype=np.float64).reshape(3, 3).copy()
        K[:2, :] /= downsampling_factor
        return cls(K, height // downsampling_factor, width // downsampling_factor,
                   downsampling_factor)


class RbSensorBuilder:
    """dbot::RbSensorBuilder<State> mirror."""

    @dataclass
    class Occlusion:
        p_occluded_visible: float = 0.1
        p_occluded_occluded: float = 0.7
        initial_occlusion_prob: float = 0.1

    @dataclass
    class Kinect:
        tail_weight: float = 0.01
        model_sigma: float = 0.003
        sigma_factor: float = 0.0014247

    @dataclass
    class Parameters:
        use_gpu: bool = True
        sample_count: int = 2000
        occlusion: "RbSensorBuilder.Occlusion" = field(default_factory=lambda: RbSensorBuilder.Occlusion())
        kinect: "RbSensorBuilder.Kinect" = field(default_factory=lambda: RbSensorBuilder.Kinect())
        delta_time: float = 1.0 / 30.0
        # GL-only knobs of the CUDA/OpenGL model: accepted and ignored (no GL in this path)
        use_custom_shaders: bool = False
        vertex_shader_file: str = ""
        fragment_shader_file: str = ""
        geometry_shader_file: str = ""
        # extension keys next to particle_filter/gpu/sample_count (the same fields as
        # dbot_amd::RbSensorBuilder<State>::Parameters in include/dbot_amd/rb_sensor_builder.hpp), "" / None: the library's choice
        likelihood_precision: str = ""        # "f64" | "f32"
        occlusion_mode: str = ""              # "reference" | "device"
        devices: "list | None" = None         # HIP ordinals: particle sharding inside the handle

        @classmethod
        def from_rosparam(cls, tree):
            """Build from the dict loaded from R:config/particle_tracker.yaml (same keys the
            node reads at R:source/dbot_ros/tracker/particle_tracker_node.cpp:165-199)."""
            pf = tree["particle_filter"]
            p = cls()
            p.use_gpu = bool(pf["use_gpu"])
            p.sample_count = int(pf["gpu" if p.use_gpu else "cpu"]["sample_count"])
            o, k = pf["observation"]["occlusion"], pf["observation"]["kinect"]
            p.occlusion = RbSensorBuilder.Occlusion(float(o["p_occluded_visible"]),
                                                    float(o["p_occluded_occluded"]),
                                                    float(o["initial_occlusion_prob"]))
            p.kinect = RbSensorBuilder.Kinect(float(k["tail_weight"]), float(k["model_sigma"]),
                                              float(k["sigma_factor"]))
            g = pf.get("gpu", {})
            p.use_custom_shaders = bool(g.get("use_custom_shaders", False))
            p.vertex_shader_file = str(g.get("vertex_shader_file", ""))
            p.fragment_shader_file = str(g.get("fragment_shader_file", ""))
            p.geometry_shader_file = str(g.get("geometry_shader_file", ""))
            p.likelihood_precision = str(g.get("likelihood_precision", "") or "")
            p.occlusion_mode = str(g.get("occlusion_mode", "") or "")
            for key, val, known in (("likelihood_precision", p.likelihood_precision, _capi.PRECISIONS),
                                    ("occlusion_mode", p.occlusion_mode, _capi.OCC_MODES)):
                if val and val not in known:
                    raise ValueError(f"particle_filter/gpu/{key}: {val!r} (one of {sorted(k for k in known if k)})")
            dev = g.get("devices")
            p.devices = [int(d) for d in dev] if dev else None
            return p

    def __init__(self, object_model, camera_data, params, device_id=0):
        self.object_model, self.camera_data, self.params = object_model, camera_data, params
        self.device_id = device_id

    def build(self):
        if not self.params.use_gpu:
            raise RbSensorError(_capi.RBS_ERR_UNSUPPORTED,
                                "use_gpu:false selects dbot's CPU model; this package only "
                                "provides the MI355X implementation (no CPU fallback)")
        p = self.params
        return RbSensor(self.object_model, self.camera_data, p, self.device_id, precision=p.likelihood_precision or None,
                        occlusion=p.occlusion_mode or None, device_ids=p.devices)


class RbSensor:
    """dbot RbSensor mirror over the C-ABI handle."""

    def __init__(self, object_model, camera_data, params, device_id=0, max_particles=None,
                 precision=None, state_layout=None, device_ids=None, slab_px=0, occlusion=None):
        """precision: None (library default) | "f64" | "f32" (rbs_config.likelihood_precision);
        state_layout: None | "window" | "dense"; device_ids: several HIP ordinals = particle
        sharding inside the handle (max_particles is then the total); slab_px: floats per
        occlusion slot (rbs_config.state_slab_px; 0 = the library's choice: whole planes up to 8 192 particles per
        device, growing slabs of rows*cols/8 above; -1 = whole planes always); occlusion: None (library default) |
        "reference" (rbs_config.occlusion_mode REFERENCE: the reference CPU model's per-pixel time stamps, propagated in
        binary64 at use -- oracle mode LAZY) | "device" (the float-stepped rule, oracle mode EAGER)."""
        self._lib = _capi.load()
        self._h = C.c_void_p()
        self.n_bodies = object_model.count_parts
        self.rows, self.cols = int(camera_data.rows), int(camera_data.cols)
        self.max_particles = int(max_particles or params.sample_count)
        self.integrated_poses = np.zeros(12 * self.n_bodies)  # default pose, SURVEY A.1

        verts = np.ascontiguousarray(np.concatenate(object_model.vertices), dtype=np.float64)
        tris = np.ascontiguousarray(np.concatenate(object_model.triangles), dtype=np.int32)
        vcnt = np.array([len(v) for v in object_model.vertices], dtype=np.int32)
        tcnt = np.array([len(t) for t in object_model.triangles], dtype=np.int32)
        cfg = _capi.RbsConfig()
        cfg.abi_version = _capi.RBS_ABI_VERSION
        cfg.device_id = device_id
        cfg.rows, cfg.cols = self.rows, self.cols
        cfg.K = (C.c_double * 9)(*np.asarray(camera_data.camera_matrix, dtype=np.float64).ravel())
        cfg.max_particles = self.max_particles
        cfg.n_objects = self.n_bodies
        cfg.vertices = verts.ctypes.data_as(C.POINTER(C.c_double))
        cfg.vertex_counts = vcnt.ctypes.data_as(C.POINTER(C.c_int32))
        cfg.triangles = tris.ctypes.data_as(C.POINTER(C.c_int32))
        cfg.triangle_counts = tcnt.ctypes.data_as(C.POINTER(C.c_int32))
        cfg.p_occluded_visible = params.occlusion.p_occluded_visible
        cfg.p_occluded_occluded = params.occlusion.p_occluded_occluded
        cfg.initial_occlusion_prob = params.occlusion.initial_occlusion_prob
        cfg.tail_weight = params.kinect.tail_weight
        cfg.model_sigma = params.kinect.model_sigma
        cfg.sigma_factor = params.kinect.sigma_factor
        cfg.delta_time = params.delta_time
        cfg.likelihood_precision = _capi.PRECISIONS[precision]
        cfg.state_layout = _capi.LAYOUTS[state_layout]
        cfg.state_slab_px = int(slab_px)
        cfg.occlusion_mode = _capi.OCC_MODES[occlusion]
        if device_ids is not None and len(device_ids) >= 1:
            ids = np.ascontiguousarray(device_ids, dtype=np.int32)
            cfg.device_id = int(ids[0])
            cfg.n_devices = len(ids)
            cfg.device_ids = ids.ctypes.data_as(C.POINTER(C.c_int32))
        else:
            cfg.n_devices = 0
            cfg.device_ids = None
        rc = self._lib.rbs_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self._lib.rbs_last_error(None).decode()
            self._h = C.c_void_p()
            raise RbSensorError(rc, msg)

    # -- life cycle -----------------------------------------------------------------
    def close(self):
        # device trackers borrow this handle: they must go first
        for ref in getattr(self, "_dependents", []):
            dep = ref()
            if dep is not None:
                dep.close()
        self._dependents = []
        if getattr(self, "_h", None) and self._h.value:
            self._lib.rbs_destroy(self._h)
            self._h = C.c_void_p()

    def _register_dependent(self, obj):
        import weakref
        if not hasattr(self, "_dependents"):
            self._dependents = []
        self._dependents.append(weakref.ref(obj))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RbSensorError(rc, self._lib.rbs_last_error(self._h).decode())

    # -- RbSensor interface ---------------------------------------------------------
    def reset(self):
        self._check(self._lib.rbs_reset(self._h))

    def set_observation(self, image):
        """image: rows*cols depths in metres, row-major, NaN = no reading
        (layout of ri::to_eigen_vector, R:source/dbot_ros/util/ros_interface.h:152-168)."""
        img = np.asarray(image)
        if img.dtype == np.float32:
            a = np.ascontiguousarray(img).ravel()
            self._check(self._lib.rbs_set_observation_f32(
                self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size))
        else:
            a = np.ascontiguousarray(img, dtype=np.float64).ravel()
            self._check(self._lib.rbs_set_observation(
                self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size))

    def set_observation_borrowed(self, image):
        """rbs_set_observation_borrowed: `image` (float64, contiguous) is NOT copied now -- keep it alive and unchanged until
        the next loglikes* / synchronize has returned (this object holds a reference until then)."""
        if isinstance(image, np.ndarray) and image.dtype == np.float32:
            img = np.ascontiguousarray(image).ravel()
            self._borrowed = img
            self._check(self._lib.rbs_set_observation_borrowed_f32(self._h, img.ctypes.data_as(C.POINTER(C.c_float)), img.size))
            return
        img = np.ascontiguousarray(image, dtype=np.float64).ravel()
        self._borrowed = img
        self._check(self._lib.rbs_set_observation_borrowed(self._h, img.ctypes.data_as(C.POINTER(C.c_double)), img.size))

    def frame_buffer(self):
        """The handle's pinned staging buffer for the NEXT frame as a numpy float32 view
        [rows*cols]: write the frame into it, then commit_frame() -- rbs_set_observation_f32
        without the host copy."""
        p = C.POINTER(C.c_float)()
        self._check(self._lib.rbs_acquire_frame_buffer(self._h, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(self.rows * self.cols,))

    def commit_frame(self):
        self._check(self._lib.rbs_commit_frame_buffer(self._h))

    def set_observation_native(self, image, downsampling_factor):
        """image: the driver's full-resolution float32 frame [height, width]; sub-sampled on the
        device by the reference's rule eval(r, c) = native(r*f, c*f)."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        if img.ndim != 2:
            raise RbSensorError(_capi.RBS_ERR_INVALID_ARGUMENT, "native frame must be 2-D [height, width]")
        self._check(self._lib.rbs_set_observation_native_f32(
            self._h, img.ctypes.data_as(C.POINTER(C.c_float)), img.shape[1], img.shape[0],
            int(downsampling_factor)))

    def get_observation(self):
        out = np.empty(self.rows * self.cols, dtype=np.float32)
        self._check(self._lib.rbs_get_observation(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def loglikes(self, deltas, indices, update=False):
        """deltas: [n, n_bodies*12] state deltas around integrated_poses; indices: int32[n],
        modified in place to identity when update is true. Returns float64[n]."""
        poses = compose_with_default(deltas, self.integrated_poses, self.n_bodies)
        return self.loglikes_poses(poses, indices, update)

    def loglikes_deltas(self, deltas, indices, update=False):
        """The same call with the composition delta (+) default pose done by the LIBRARY on the device
        (rbs_loglikes_deltas: what the dbot binding calls).  deltas: [n, n_bodies*12] as for loglikes()."""
        d = np.ascontiguousarray(deltas, dtype=np.float64).reshape(-1, self.n_bodies * 12)
        n = d.shape[0]
        dflt = np.ascontiguousarray(self.integrated_poses, dtype=np.float64).reshape(self.n_bodies * 12)
        if not (isinstance(indices, np.ndarray) and indices.dtype == np.int32
                and indices.flags.c_contiguous and indices.size == n):
            raise RbSensorError(_capi.RBS_ERR_INVALID_ARGUMENT,
                                "indices must be a contiguous int32 array of length n")
        out = np.empty(n, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.rbs_loglikes_deltas(
            self._h, d.ctypes.data_as(dp), dflt.ctypes.data_as(dp), 12,
            indices.ctypes.data_as(C.POINTER(C.c_int32)), n, int(bool(update)), out.ctypes.data_as(dp)))
        return out

    def get_poses(self, n):
        """The absolute poses [n, n_bodies, 12] the last host-pointer likelihood call evaluated (test hook)."""
        out = np.empty((n, self.n_bodies, 12), dtype=np.float64)
        self._check(self._lib.rbs_get_poses(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return out

    def loglikes_poses(self, poses, indices, update=False):
        """poses: absolute R|t, [n, n_bodies, 12]."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, self.n_bodies * 12)
        n = poses.shape[0]
        if not (isinstance(indices, np.ndarray) and indices.dtype == np.int32
                and indices.flags.c_contiguous and indices.size == n):
            raise RbSensorError(_capi.RBS_ERR_INVALID_ARGUMENT,
                                "indices must be a contiguous int32 array of length n")
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.rbs_loglikes(
            self._h, poses.ctypes.data_as(C.POINTER(C.c_double)),
            indices.ctypes.data_as(C.POINTER(C.c_int32)), n, int(bool(update)),
            out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def loglikes_poses_prefetch(self, poses, indices, next_image, update=False):
        """loglikes_poses with the NEXT frame (float32 [rows*cols]) uploaded behind the call's kernels;
        set_observation_prefetched() then makes it the observation."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, self.n_bodies * 12)
        n = poses.shape[0]
        nxt = np.ascontiguousarray(next_image, dtype=np.float32).ravel()
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.rbs_loglikes_prefetch(
            self._h, poses.ctypes.data_as(C.POINTER(C.c_double)), indices.ctypes.data_as(C.POINTER(C.c_int32)), n, int(bool(update)),
            out.ctypes.data_as(C.POINTER(C.c_double)), nxt.ctypes.data_as(C.POINTER(C.c_float)), nxt.size))
        return out

    def set_observation_prefetched(self):
        self._check(self._lib.rbs_set_observation_prefetched(self._h))

    def set_observation_device(self, d_depth_ptr, stream=None):
        """Frame already on the device (float32 [rows*cols], raw address); asynchronous."""
        self._check(self._lib.rbs_set_observation_device(self._h, d_depth_ptr, stream))

    def loglikes_device(self, d_poses_ptr, d_indices_ptr, n, update, d_out_ptr, stream=None):
        """Asynchronous device-pointer variant (raw addresses, e.g. torch.Tensor.data_ptr())."""
        self._check(self._lib.rbs_loglikes_device(self._h, d_poses_ptr, d_indices_ptr, int(n),
                                                  int(bool(update)), d_out_ptr, stream))

    def synchronize(self):
        self._check(self._lib.rbs_synchronize(self._h))

    # -- inspection -----------------------------------------------------------------
    def get_occlusion(self, slot):
        out = np.empty(self.rows * self.cols, dtype=np.float32)
        self._check(self._lib.rbs_get_occlusion(self._h, int(slot),
                                                out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def set_occlusion(self, slot, plane):
        a = np.ascontiguousarray(plane, dtype=np.float32).ravel()
        if a.size != self.rows * self.cols:
            raise RbSensorError(_capi.RBS_ERR_INVALID_ARGUMENT, "plane has the wrong size")
        self._check(self._lib.rbs_set_occlusion(self._h, int(slot),
                                                a.ctypes.data_as(C.POINTER(C.c_float))))

    def get_window(self, slot):
        """(x0, y0, x1, y1) of the slot's stored window; outside it the plane is the background."""
        w = (C.c_int32 * 4)()
        self._check(self._lib.rbs_get_window(self._h, int(slot), w))
        return tuple(int(x) for x in w)

    def shared_trail_state(self):
        """(active, rebases): has the handle switched to a shared background plane, how often has it re-based (rbsensor_mi355x.h)."""
        a, r = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.rbs_shared_trail_state(self._h, C.byref(a), C.byref(r)))
        return bool(a.value), int(r.value)

    def set_option(self, name, value):
        """rbs_set_option: "shared_trail" (0 / 1), "shared_trail_enter" (window fraction), "shared_trail_every" (calls),
        "tracker_split_max" (evaluations), "timing_every"."""
        self._check(self._lib.rbs_set_option(self._h, _capi.OPTIONS[name], float(value)))

    def shared_trail_rebase(self, global_slot):
        """rbs_shared_trail_rebase: at the next updating call enter the shared-trail representation (if need be) and re-base the shared
        plane on that GLOBAL slot's plane; -2: back to the scalar background.  Attached ranks: the same call on every rank."""
        self._check(self._lib.rbs_shared_trail_rebase(self._h, int(global_slot)))

    def window_fraction(self):
        """The window area the handle sampled last, as a fraction of the frame (what the shared trail's policy looks at)."""
        v = C.c_double()
        self._check(self._lib.rbs_window_fraction(self._h, C.byref(v)))
        return float(v.value)

    def get_background(self):
        v = C.c_float()
        self._check(self._lib.rbs_get_background(self._h, C.byref(v)))
        return float(np.float32(v.value))

    def set_timing_every(self, every):
        self._check(self._lib.rbs_set_timing_every(self._h, int(every)))

    def raster_kernel_ms(self, last_n=64):
        v = C.c_float()
        self._check(self._lib.rbs_raster_kernel_ms(self._h, int(last_n), C.byref(v)))
        return float(v.value)

    def export_plane(self, slot, d_dst_ptr, stream=None):
        """Device-to-device copy of a slot's plane into caller-owned device memory (raw address)."""
        self._check(self._lib.rbs_export_plane(self._h, int(slot), d_dst_ptr, stream))

    def import_plane(self, slot, d_src_ptr, stream=None):
        self._check(self._lib.rbs_import_plane(self._h, int(slot), d_src_ptr, stream))

    def export_window(self, slot, d_payload_ptr, capacity_floats, stream=None):
        """The slot's window (x0, y0, x1, y1) and, packed row-major into caller-owned device memory, its values:
        the window-sized form of export_plane."""
        r = (C.c_int32 * 4)()
        self._check(self._lib.rbs_export_window(self._h, int(slot), r, d_payload_ptr, int(capacity_floats), stream))
        return tuple(int(x) for x in r)

    def import_window(self, slot, rect, d_payload_ptr, stream=None):
        r = (C.c_int32 * 4)(*[int(x) for x in rect])
        self._check(self._lib.rbs_import_window(self._h, int(slot), r, d_payload_ptr, stream))

    def stream_join(self, stream=None):
        """`stream` waits (on the device) for the planes of the last updating call, side stream included."""
        self._check(self._lib.rbs_stream_join(self._h, stream))

    def ipc_export(self):
        buf = (C.c_ubyte * _capi.RBS_IPC_BLOB_BYTES)()
        self._check(self._lib.rbs_ipc_export(self._h, buf))
        return bytes(buf)

    def ipc_attach(self, rank, blobs):
        """blobs: every rank's ipc_export(), in rank order.  Afterwards parent indices are GLOBAL slots
        (rank * max_particles + local slot) and parents of other ranks are read in place."""
        raw = b"".join(blobs)
        assert len(raw) == len(blobs) * _capi.RBS_IPC_BLOB_BYTES
        self._check(self._lib.rbs_ipc_attach(self._h, int(rank), len(blobs), raw))
        self.peer_rank, self.peer_world = int(rank), len(blobs)

    def stage_windows(self, d_src_global_ptr, d_dst_local_ptr, n, stream=None):
        self._check(self._lib.rbs_stage_windows(self._h, d_src_global_ptr, d_dst_local_ptr, int(n), stream))

    def peer_resample(self, d_loglik_all_ptr, d_uniforms_sorted_ptr, n_total, n_local, rank, min_share, temperature,
                      d_parent_idx_ptr, d_stage_src_ptr, d_stage_dst_ptr, d_parents_local_ptr, d_counts_ptr, stream=None):
        """rbs_peer_resample: multinomial resampling over all ranks' particles at SORTED uniforms + this rank's plan
        (parents, what to stage) in one launch; device pointers, nothing synchronises after the first call."""
        self._check(self._lib.rbs_peer_resample(self._h, d_loglik_all_ptr, d_uniforms_sorted_ptr, int(n_total), int(n_local), int(rank),
                                                int(min_share), float(temperature), d_parent_idx_ptr, d_stage_src_ptr, d_stage_dst_ptr,
                                                d_parents_local_ptr, d_counts_ptr, stream))

    def occlusion_device_ptr(self, slot, next_buffer=False):
        p = C.c_void_p()
        fn = self._lib.rbs_occlusion_next_device_ptr if next_buffer else self._lib.rbs_occlusion_device_ptr
        self._check(fn(self._h, int(slot), C.byref(p)))
        return p.value

    def render_depth(self, pose):
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(self.n_bodies * 12)
        out = np.empty(self.rows * self.cols, dtype=np.float32)
        self._check(self._lib.rbs_render_depth(self._h, pose.ctypes.data_as(C.POINTER(C.c_double)),
                                               out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self._lib.rbs_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def timing_summary(self, last_n):
        """(call_ms, copy_kernel_ms, n_used) averaged over the last calls, from the library's
        own HIP events on the streams its kernels run on."""
        a, b, n = C.c_float(), C.c_float(), C.c_int32()
        self._check(self._lib.rbs_timing_summary(self._h, int(last_n), C.byref(a), C.byref(b),
                                                 C.byref(n)))
        return float(a.value), float(b.value), int(n.value)
