"""Rigid-body state helpers mirroring dbot's FreeFloatingRigidBodiesState<> as the reference
uses it (R:source/dbot_ros/tracker/particle_tracker_node.cpp:126,244-249,
R:source/dbot_ros/util/ros_interface.h:49-133): per body 12 doubles =
position(3), orientation as rotation ("Euler") vector(3), linear velocity(3), angular velocity(3).

The sensor's C-ABI takes absolute camera-frame poses as R|t (12 doubles per body); these
helpers do the delta (+) default-pose composition on the host:
    R = R(delta) . R(default),   t = t(delta) + t(default)
"""
import numpy as np

STATE_DIM = 12  # per body


def rotvec_to_matrix(rv):
    """Rotation vector(s) [...,3] -> rotation matrices [...,3,3] (angle-axis via quaternion)."""
    rv = np.asarray(rv, dtype=np.float64)
    angle = np.linalg.norm(rv, axis=-1)
    half = 0.5 * angle
    # sin(a/2)/a with the series limit near 0
    small = angle < 1e-9
    safe = np.where(small, 1.0, angle)
    k = np.where(small, 0.5 - angle * angle / 48.0, np.sin(half) / safe)
    w = np.cos(half)
    x, y, z = rv[..., 0] * k, rv[..., 1] * k, rv[..., 2] * k
    return quat_to_matrix(np.stack([w, x, y, z], axis=-1))


def quat_to_matrix(q):
    """Unit quaternion(s) [...,4] as (w,x,y,z) -> rotation matrices [...,3,3]."""
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3), dtype=np.float64)
    R[..., 0, 0] = 1.0 - 2.0 * (y * y + z * z)
    R[..., 0, 1] = 2.0 * (x * y - w * z)
    R[..., 0, 2] = 2.0 * (x * z + w * y)
    R[..., 1, 0] = 2.0 * (x * y + w * z)
    R[..., 1, 1] = 1.0 - 2.0 * (x * x + z * z)
    R[..., 1, 2] = 2.0 * (y * z - w * x)
    R[..., 2, 0] = 2.0 * (x * z - w * y)
    R[..., 2, 1] = 2.0 * (y * z + w * x)
    R[..., 2, 2] = 1.0 - 2.0 * (x * x + y * y)
    return R


def matrix_to_rotvec(R):
    """Rotation matrix [3,3] -> rotation vector [3] (atan2 form: accurate for tiny angles)."""
    R = np.asarray(R, dtype=np.float64)
    s = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])  # sin(a) * axis
    sn = np.linalg.norm(s)
    cs = 0.5 * (np.trace(R) - 1.0)
    angle = np.arctan2(sn, cs)
    if sn > 1e-8:
        return s * (angle / sn)
    if cs > 0.0:  # angle ~ 0: sin(a)/a -> 1
        return s
    # angle ~ pi: axis from the symmetric part
    d = np.sqrt(np.maximum((np.diag(R) + 1.0) * 0.5, 0.0))
    i = int(np.argmax(d))
    axis = (R[:, i] + np.eye(3)[i]) / (2.0 * d[i])
    return axis / np.linalg.norm(axis) * angle


def pack_Rt(R, t):
    """R [...,3,3], t [...,3] -> [...,12] = R row-major then t (the C-ABI pose layout)."""
    R = np.asarray(R, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    return np.concatenate([R.reshape(R.shape[:-2] + (9,)), t], axis=-1)


def states_to_Rt(states, n_bodies):
    """State array [n, n_bodies*12] -> absolute poses [n, n_bodies, 12] (no default pose)."""
    s = np.asarray(states, dtype=np.float64).reshape(-1, n_bodies, STATE_DIM)
    return pack_Rt(rotvec_to_matrix(s[..., 3:6]), s[..., 0:3])


def compose_with_default(deltas, default_state, n_bodies):
    """delta (+) default pose: deltas [n, n_bodies*12], default [n_bodies*12]
    -> absolute poses [n, n_bodies, 12]."""
    d = np.asarray(deltas, dtype=np.float64).reshape(-1, n_bodies, STATE_DIM)
    z = np.asarray(default_state, dtype=np.float64).reshape(n_bodies, STATE_DIM)
    R = rotvec_to_matrix(d[..., 3:6]) @ rotvec_to_matrix(z[:, 3:6])[None]
    t = d[..., 0:3] + z[None, :, 0:3]
    return pack_Rt(R, t)
