"""Host-side mirror of the callers either side of the hot path ("next" rows):
the object state transition, the Rao-Blackwellised coordinate particle filter loop and the
tracker that the reference node builds and drives

    ObjectTransitionBuilder<State>::Parameters   R:source/dbot_ros/tracker/particle_tracker_node.cpp:138-159
    ParticleTrackerBuilder<Tracker>::Parameters  R:...particle_tracker_node.cpp:208-218
    tracker->initialize(initial_poses)           R:...particle_tracker_node.cpp:252
    tracker_->track(image) -> State              R:source/dbot_ros/object_tracker_ros.hpp:49

restated from SURVEY.md Appendix A.1/A.6 (recalled upstream behaviour; the dbot sources are not
available, so this part is unpinned like the oracle).  The sensor is anything with
set_observation / loglikes_poses / reset (the product's RbSensor, or the oracle in CPU tests).
Randomness comes from a caller-supplied numpy Generator (fl's mt19937 streams cannot be
reproduced without fl).
"""
from dataclasses import dataclass

import numpy as np

from . import filter as flt
from .pose import matrix_to_rotvec, pack_Rt, rotvec_to_matrix

BODY = 12


class ObjectTransitionBuilder:
    @dataclass
    class Parameters:
        linear_sigma_x: float = 0.0025
        linear_sigma_y: float = 0.0025
        linear_sigma_z: float = 0.0025
        angular_sigma_x: float = 0.02
        angular_sigma_y: float = 0.02
        angular_sigma_z: float = 0.02
        velocity_factor: float = 0.8
        part_count: int = 1

        @classmethod
        def from_rosparam(cls, tree, part_count):
            t = tree["particle_filter"]["object_transition"]
            return cls(*(float(t[k]) for k in ("linear_sigma_x", "linear_sigma_y", "linear_sigma_z",
                                               "angular_sigma_x", "angular_sigma_y", "angular_sigma_z",
                                               "velocity_factor")), part_count=part_count)

    def __init__(self, params):
        self.params = params

    def build(self):
        return ObjectTransition(self.params)


class ObjectTransition:
    """Random walk on the velocities (R:config/particle_tracker.yaml:51-63): per body and per
    frame, with 6 standard normals n = (n_lin, n_ang):
        vel'  = velocity_factor * vel + sigma o n
        pose' = pose + vel'."""

    def __init__(self, p):
        self.vf = p.velocity_factor
        self.sigma = np.array([p.linear_sigma_x, p.linear_sigma_y, p.linear_sigma_z,
                               p.angular_sigma_x, p.angular_sigma_y, p.angular_sigma_z])
        self.part_count = p.part_count

    def apply(self, states, noise, body):
        """states [n, parts*12] (not modified), noise [n, 6] for `body` -> new states."""
        out = states.copy()
        s = out[:, BODY * body: BODY * body + BODY]
        s[:, 6:12] = self.vf * s[:, 6:12] + self.sigma * noise
        s[:, 0:6] = s[:, 0:6] + s[:, 6:12]
        return out


class ParticleTrackerBuilder:
    @dataclass
    class Parameters:
        evaluation_count: int = 2000
        moving_average_update_rate: float = 1.0
        max_kl_divergence: float = 2.0
        center_object_frame: bool = True

        @classmethod
        def from_rosparam(cls, tree, evaluation_count):
            pf = tree["particle_filter"]
            return cls(int(evaluation_count), float(pf["moving_average_update_rate"]),
                       float(pf["max_kl_divergence"]), bool(pf["center_object_frame"]))

    def __init__(self, transition_builder, sensor_builder, object_model, params):
        self.transition_builder, self.sensor_builder = transition_builder, sensor_builder
        self.object_model, self.params = object_model, params

    def build(self, rng=None):
        return ParticleTracker(self.transition_builder.build(), self.sensor_builder.build(),
                               self.object_model, self.params, rng)


class ParticleTracker:
    """dbot::ParticleTracker mirror: one sampling block per object, particle count =
    evaluation_count / #blocks."""

    def __init__(self, transition, sensor, object_model, params, rng=None):
        self.transition, self.sensor, self.params = transition, sensor, params
        self.parts = object_model.count_parts
        self.centers = np.array(object_model.centers)  # mesh re-centring offsets (center_object_frame)
        self.n = max(1, params.evaluation_count // self.parts)
        self.rng = rng if rng is not None else np.random.default_rng(0)
        self.default = np.zeros(self.parts * BODY)       # integrated pose the deltas live around
        self.particles = np.zeros((self.n, self.parts * BODY))
        self.log_weights = np.zeros(self.n)
        self.loglikes = np.zeros(self.n)
        self.indices = np.zeros(self.n, dtype=np.int32)
        self.moving_average = None
        self.n_resamplings = 0

    # -- State <-> model coordinates ---------------------------------------------------------
    def _to_model(self, state):
        """Camera-frame pose of the ORIGINAL mesh frame -> pose of the centred mesh frame."""
        s = np.array(state, dtype=np.float64).reshape(self.parts, BODY).copy()
        if self.params.center_object_frame:
            for b in range(self.parts):
                s[b, 0:3] += rotvec_to_matrix(s[b, 3:6]) @ self.centers[b]
        return s.ravel()

    def _from_model(self, state):
        s = np.array(state, dtype=np.float64).reshape(self.parts, BODY).copy()
        if self.params.center_object_frame:
            for b in range(self.parts):
                s[b, 0:3] -= rotvec_to_matrix(s[b, 3:6]) @ self.centers[b]
        return s.ravel()

    # -- Tracker interface -------------------------------------------------------------------
    def initialize(self, initial_states):
        """initial_states: list of State vectors (parts*12); the first is used as the default
        pose, particles start as zero deltas (R:...particle_tracker_node.cpp:242-252)."""
        self.default = self._to_model(initial_states[0])
        self.default.reshape(self.parts, BODY)[:, 6:12] = 0.0
        self.particles[:] = 0.0
        self.log_weights[:] = 0.0
        self.loglikes[:] = 0.0
        self.indices[:] = 0
        self.moving_average = None
        self.sensor.reset()

    def absolute_poses(self, particles):
        d = particles.reshape(-1, self.parts, BODY)
        z = self.default.reshape(self.parts, BODY)
        R = rotvec_to_matrix(d[..., 3:6]) @ rotvec_to_matrix(z[:, 3:6])[None]
        return pack_Rt(R, d[..., 0:3] + z[None, :, 0:3])

    def draw_randomness(self):
        """Per frame: standard normals [parts, n, 6] and uniforms [parts, n], drawn
        unconditionally so host and device trackers consume identical streams."""
        return (self.rng.standard_normal((self.parts, self.n, 6)), self.rng.random((self.parts, self.n)))

    def track(self, image, normals=None, uniforms=None):
        """One depth frame -> estimated State (camera-frame poses + velocities per object)."""
        if normals is None or uniforms is None:
            normals, uniforms = self.draw_randomness()
        self.sensor.set_observation(image)
        old = self.particles
        noises = np.zeros((self.n, self.parts, 6))
        new = old
        for b in range(self.parts):
            noises[:, b] = normals[b]
            # every block restarts from the SAME old particles; noise accumulates over blocks
            new = old
            for bb in range(b + 1):
                new = self.transition.apply(new, noises[:, bb], bb)
            last = b == self.parts - 1
            idx = self.indices.copy()
            new_ll = self.sensor.loglikes_poses(self.absolute_poses(new), idx, update=last)
            if last:
                self.indices = idx
            self.log_weights += new_ll - self.loglikes
            self.loglikes = new_ll
            w = flt.normalized_weights(self.log_weights)
            if flt.kl_to_uniform(w) > self.params.max_kl_divergence:
                parents = flt.multinomial_resample(w, uniforms[b])
                self.n_resamplings += 1
                self.indices = self.indices[parents].copy()
                old, new, noises = old[parents], new[parents], noises[parents]
                self.loglikes = self.loglikes[parents]
                self.log_weights = np.zeros(self.n)
        self.particles = new
        # fold the weighted mean delta into the default pose and re-centre the particles
        w = flt.normalized_weights(self.log_weights)
        mean = flt.weighted_mean(w, self.particles).reshape(self.parts, BODY)
        z = self.default.reshape(self.parts, BODY)
        p = self.particles.reshape(self.n, self.parts, BODY)
        for b in range(self.parts):
            Rm = rotvec_to_matrix(mean[b, 3:6])
            z[b, 0:3] += mean[b, 0:3]
            z[b, 3:6] = matrix_to_rotvec(Rm @ rotvec_to_matrix(z[b, 3:6]))
            z[b, 6:12] = mean[b, 6:12]
            p[:, b, 0:3] -= mean[b, 0:3]
            Rd = rotvec_to_matrix(p[:, b, 3:6]) @ Rm.T
            p[:, b, 3:6] = _rotvecs(Rd)
        est = self._from_model(self.default)
        rate = self.params.moving_average_update_rate
        self.moving_average = est if self.moving_average is None else rate * est + (1 - rate) * self.moving_average
        return self.moving_average.copy()


def _rotvecs(R):
    """Vectorised matrix -> rotation vector (atan2 form; particle deltas are far from pi)."""
    s = 0.5 * np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], -1)
    sn = np.linalg.norm(s, axis=-1)
    cs = 0.5 * (np.trace(R, axis1=1, axis2=2) - 1.0)
    ang = np.arctan2(sn, cs)
    k = np.where(sn > 1e-8, ang / np.where(sn > 1e-8, sn, 1.0), 1.0)
    out = s * k[:, None]
    flip = (sn <= 1e-8) & (cs <= 0.0)      # angle ~ pi: axis from the symmetric part (pose.matrix_to_rotvec)
    for i in np.nonzero(flip)[0]:
        out[i] = matrix_to_rotvec(R[i])
    return out


class DeviceParticleTracker(ParticleTracker):
    """Same interface, but the transition, filter step and mean run on the sensor's device
    (rbs_tracker_* in librbsensor_mi355x.so): one host synchronisation per frame.  With
    device_rng=True the normals/uniforms are drawn on the device (Philox), otherwise they come
    from this object's numpy Generator exactly as in the host tracker."""

    def __init__(self, transition, sensor, object_model, params, rng=None, device_rng=False, seed=0):
        import ctypes as C
        from . import _capi
        super().__init__(transition, sensor, object_model, params, rng)
        self._C, self._capi = C, _capi
        self._lib = _capi.load()
        self.device_rng, self.seed = device_rng, seed
        tp = _capi.RbsTrackerParams()
        tp.linear_sigma = (C.c_double * 3)(*transition.sigma[:3])
        tp.angular_sigma = (C.c_double * 3)(*transition.sigma[3:])
        tp.velocity_factor = transition.vf
        tp.max_kl_divergence = params.max_kl_divergence
        tp.n_particles = self.n
        self._t = C.c_void_p()
        rc = self._lib.rbs_tracker_create(sensor._h, C.byref(tp), C.byref(self._t))
        if rc != 0:
            sensor._check(rc)
        sensor._register_dependent(self)   # the C tracker borrows the sensor handle

    def close(self):
        if getattr(self, "_t", None) is not None and self._t.value:
            self._lib.rbs_tracker_destroy(self._t)
            self._t = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def initialize(self, initial_states):
        self.default = self._to_model(initial_states[0])
        self.default.reshape(self.parts, BODY)[:, 6:12] = 0.0
        d = np.ascontiguousarray(self.default, dtype=np.float64)
        self.sensor._check(self._lib.rbs_tracker_initialize(self._t, d.ctypes.data_as(self._C.POINTER(self._C.c_double))))
        self.moving_average = None
        self.n_resamplings = 0

    def _frame_args(self, image, normals, uniforms):
        C = self._C
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        if not self.device_rng and (normals is None or uniforms is None):
            normals, uniforms = self.draw_randomness()
        # a float64 image (what dbot's tracker receives) goes in as it is: rbs_tracker_track_f64 / _submit_f64 convert while staging
        f64 = isinstance(image, np.ndarray) and image.dtype == np.float64
        img = np.ascontiguousarray(image, dtype=np.float64 if f64 else np.float32).ravel()
        nptr = uptr = None
        if normals is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float64)
            nptr = normals.ctypes.data_as(dp)
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
            uptr = uniforms.ctypes.data_as(dp)
        self._f64 = f64
        return (img, normals, uniforms), (img.ctypes.data_as(dp if f64 else fp), nptr, uptr, C.c_uint64(self.seed))

    def _estimate(self, out, nres):
        self.default = out
        self.n_resamplings = int(nres.value)
        est = self._from_model(self.default)
        rate = self.params.moving_average_update_rate
        self.moving_average = est if self.moving_average is None else rate * est + (1 - rate) * self.moving_average
        return self.moving_average.copy()

    def track(self, image, normals=None, uniforms=None):
        C = self._C
        _keep, args = self._frame_args(image, normals, uniforms)
        out = np.empty(self.parts * BODY)
        nres = C.c_int32()
        fn = self._lib.rbs_tracker_track_f64 if self._f64 else self._lib.rbs_tracker_track
        self.sensor._check(fn(self._t, *args, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nres)))
        return self._estimate(out, nres)

    def submit(self, image, normals=None, uniforms=None):
        """Enqueue one frame (rbs_tracker_submit) and return at once; at most two frames may be in
        flight.  result() hands out the estimates in submission order."""
        _keep, args = self._frame_args(image, normals, uniforms)
        self.sensor._check((self._lib.rbs_tracker_submit_f64 if self._f64 else self._lib.rbs_tracker_submit)(self._t, *args))

    def result(self):
        """The moving-average estimate of the oldest submitted frame (rbs_tracker_result)."""
        C = self._C
        out = np.empty(self.parts * BODY)
        nres = C.c_int32()
        self.sensor._check(self._lib.rbs_tracker_result(self._t, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nres)))
        return self._estimate(out, nres)

    def get_state(self):
        """(particle deltas [n, parts*12], log-weights [n], occlusion slot map [n]) from the device."""
        C = self._C
        p = np.empty((self.n, self.parts * BODY))
        w = np.empty(self.n)
        i = np.empty(self.n, dtype=np.int32)
        self.sensor._check(self._lib.rbs_tracker_get(self._t, p.ctypes.data_as(C.POINTER(C.c_double)),
                                                     w.ctypes.data_as(C.POINTER(C.c_double)),
                                                     i.ctypes.data_as(C.POINTER(C.c_int32))))
        return p, w, i
