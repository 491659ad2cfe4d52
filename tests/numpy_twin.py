"""Independent numpy restatement of the same specification as oracle/rbsensor_oracle.c, used
ONLY to cross-check the C oracle on small cases (a second pair of eyes on the formulas and on
the written coverage rule).  Test infrastructure; pure-Python loops, small inputs only.

Follows SURVEY.md Appendix A.2-A.5 (recalled upstream formulas) and the reference's call-site
contract (parameters R:source/dbot_ros/tracker/particle_tracker_node.cpp:164-199, defaults
R:config/particle_tracker.yaml:38-49).
"""
import math

import numpy as np
from scipy.special import erf

MAX_DEPTH = 6.0
HALF_LIFE_DEPTH = 1.0
LAMBDA = -math.log(0.5) / HALF_LIFE_DEPTH


def prob_visible(o, r, tw, ms, sf):
    sigma = ms + sf * o * o
    d = r - o
    return tw / MAX_DEPTH + (1.0 - tw) * np.exp(-(d * d) / (2.0 * sigma * sigma)) / (math.sqrt(2.0 * math.pi) * sigma)


def prob_occluded(o, r, tw, ms, sf):
    sigma = ms + sf * o * o
    lam = LAMBDA
    return (tw / MAX_DEPTH + (1.0 - tw) * lam * np.exp(0.5 * lam * (2.0 * r - 2.0 * o + lam * sigma * sigma))
            * (1.0 + erf((r - o + lam * sigma * sigma) / (math.sqrt(2.0) * sigma)))
            / (2.0 * (np.exp(r * lam) - 1.0)))


def prob_background(o, tw, ms, sf):
    sigma = ms + sf * o * o
    lam = LAMBDA
    return tw / MAX_DEPTH + (1.0 - tw) * lam * np.exp(0.5 * lam * (-2.0 * o + lam * sigma * sigma))


def propagate(occ, dt, p_ov, p_oo):
    c = p_oo - p_ov
    pow_c = np.exp(dt * math.log(c))
    new_visible = pow_c * (1.0 - occ) + (1.0 - p_oo) * (pow_c - 1.0) / (c - 1.0)
    return 1.0 - new_visible


def render(vertices, triangles, poses, K, rows, cols):
    """Depth image (float32, +inf uncovered) under the written coverage rule: integer pixel
    sample points, closed triangles (all edge functions >= 0 or all <= 0), no culling, vertices
    with Z <= 0 or zero projected area skip the triangle, plane/ray depth rounded to float,
    z-min.  Same binary64 operation order as the C oracle."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    depth = np.full((rows, cols), np.inf, dtype=np.float32)
    for verts, tris, Rt in zip(vertices, triangles, np.asarray(poses).reshape(-1, 12)):
        R, t = Rt[:9].reshape(3, 3), Rt[9:]
        for tri in tris:
            P = verts[tri]  # [3,3]
            X = ((R[0, 0] * P[:, 0] + R[0, 1] * P[:, 1]) + R[0, 2] * P[:, 2]) + t[0]
            Y = ((R[1, 0] * P[:, 0] + R[1, 1] * P[:, 1]) + R[1, 2] * P[:, 2]) + t[1]
            Z = ((R[2, 0] * P[:, 0] + R[2, 1] * P[:, 1]) + R[2, 2] * P[:, 2]) + t[2]
            if not (Z > 0.0).all():
                continue
            iz = 1.0 / Z
            u = fx * (X * iz) + cx
            v = fy * (Y * iz) + cy
            e01u, e01v = u[1] - u[0], v[1] - v[0]
            e12u, e12v = u[2] - u[1], v[2] - v[1]
            e20u, e20v = u[0] - u[2], v[0] - v[2]
            area2 = e01u * (v[2] - v[0]) - e01v * (u[2] - u[0])
            if not (area2 != 0.0 and abs(area2) < np.inf):
                continue
            ax, ay, az = X[1] - X[0], Y[1] - Y[0], Z[1] - Z[0]
            bx, by, bz = X[2] - X[0], Y[2] - Y[0], Z[2] - Z[0]
            nx, ny, nz = ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx
            nv0 = (nx * X[0] + ny * Y[0]) + nz * Z[0]
            pa, pb = nx / fx, ny / fy
            pc = (nz - pa * cx) - pb * cy
            xlo, xhi = max(math.ceil(u.min()), 0), min(math.floor(u.max()), cols - 1)
            ylo, yhi = max(math.ceil(v.min()), 0), min(math.floor(v.max()), rows - 1)
            if xlo > xhi or ylo > yhi:
                continue
            py, px = np.meshgrid(np.arange(ylo, yhi + 1, dtype=np.float64),
                                 np.arange(xlo, xhi + 1, dtype=np.float64), indexing="ij")
            E0 = e01u * (py - v[0]) - e01v * (px - u[0])
            E1 = e12u * (py - v[1]) - e12v * (px - u[1])
            E2 = e20u * (py - v[2]) - e20v * (px - u[2])
            inside = ((E0 >= 0) & (E1 >= 0) & (E2 >= 0)) | ((E0 <= 0) & (E1 <= 0) & (E2 <= 0))
            with np.errstate(divide="ignore", invalid="ignore"):
                zf = (nv0 / ((pa * px + pb * py) + pc)).astype(np.float32)
            ok = inside & (zf > 0) & np.isfinite(zf)
            sub = depth[ylo:yhi + 1, xlo:xhi + 1]
            sub[ok] = np.minimum(sub[ok], zf[ok])
    return depth.ravel()


class TwinSensor:
    """Lazy-occlusion (reference CPU semantics) sensor in numpy: same interface subset as the
    oracle binding."""

    def __init__(self, object_model, camera_data, params, max_particles):
        self.om, self.cam, self.p = object_model, camera_data, params
        self.rows, self.cols = camera_data.rows, camera_data.cols
        self.npx = self.rows * self.cols
        self.n = max_particles
        self.reset()

    def reset(self):
        self.occ = np.full((self.n, self.npx), np.float32(self.p.occlusion.initial_occlusion_prob), np.float32)
        self.stamp = np.zeros((self.n, self.npx), np.int64)
        self.clock = 0
        self.frame = np.full(self.npx, np.nan, np.float32)

    def set_observation(self, image):
        self.frame = np.asarray(image, dtype=np.float64).astype(np.float32).ravel()
        self.clock += 1

    def render_depth(self, pose):
        return render(self.om.vertices, self.om.triangles, pose, np.asarray(self.cam.camera_matrix),
                      self.rows, self.cols)

    def loglikes_poses(self, poses, indices, update=False):
        poses = np.asarray(poses, dtype=np.float64).reshape(len(indices), -1)
        k, o_ = self.p.kinect, self.p.occlusion
        out = np.zeros(len(indices))
        new_occ, new_stamp = self.occ.copy(), self.stamp.copy()
        for i, parent in enumerate(indices):
            r = self.render_depth(poses[i])
            m = np.isfinite(r) & np.isfinite(self.frame)
            occ_p, st_p = self.occ[parent].copy(), self.stamp[parent].copy()
            o = self.frame[m].astype(np.float64)
            rr = r[m].astype(np.float64)
            dt = (self.clock - st_p[m]) * self.p.delta_time
            prior = propagate(occ_p[m].astype(np.float64), dt, o_.p_occluded_visible,
                              o_.p_occluded_occluded).astype(np.float32)
            a = (prob_visible(o, rr, k.tail_weight, k.model_sigma, k.sigma_factor) * (1.0 - prior.astype(np.float64))).astype(np.float32)
            b = (prob_occluded(o, rr, k.tail_weight, k.model_sigma, k.sigma_factor) * prior.astype(np.float64)).astype(np.float32)
            pbg = prob_background(o, k.tail_weight, k.model_sigma, k.sigma_factor).astype(np.float32)
            s = a + b
            out[i] = np.log((s / pbg).astype(np.float64)).sum()
            if update:
                occ_p[m] = b / s
                st_p[m] = self.clock
                new_occ[i], new_stamp[i] = occ_p, st_p
        if update:
            self.occ, self.stamp = new_occ, new_stamp
            indices[:] = np.arange(len(indices))
        return out
