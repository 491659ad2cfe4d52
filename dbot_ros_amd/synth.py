"""Deterministic synthetic inputs for tests and bench.py (SURVEY.md 8d / BASELINE.md 2): the
reference ships no meshes, bags or frames, so every workload is generated here.

  meshes   M1 bumped ellipsoid from an icosphere (level 4 -> 2 562 v / 5 120 tris)
           M2 subdivided box 8x6x4 cm, M3 torus R=4 cm r=1.5 cm (non-convex), M4 UV ellipsoid
  camera   fx = fy = 570.3, cx = 319.5, cy = 239.5 at 640x480 (scaled with resolution)
  frames   rendered object depth (supplied by the caller: the product's render hook on the GPU,
           the oracle's renderer in CPU tests) + tilted background plane + occluding slab +
           Kinect-like noise + NaN drop-outs
  particles  pose deltas ~ N(0, diag(sigma_lin, sigma_ang)) around a ground-truth pose
"""
import numpy as np

from .pose import pack_Rt, rotvec_to_matrix


# ---------------------------------------------------------------------------- meshes
def icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t),
         (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts), np.array(faces, dtype=np.int32)


def mesh_m1(level=4):
    """'Duck-size' bumped ellipsoid: radii (5,4,3) cm, 1 cm sinusoidal bump."""
    v, f = icosphere(level)
    bump = 1.0 + 0.2 * np.sin(3.0 * v[:, 0]) * np.cos(2.0 * v[:, 1] + 0.5)
    return v * np.array([0.05, 0.04, 0.03]) * bump[:, None], f


def _grid_patch(origin, du, dv, nu, nv):
    us, vs = np.meshgrid(np.arange(nu + 1) / nu, np.arange(nv + 1) / nv, indexing="ij")
    pts = origin[None, None] + us[..., None] * du[None, None] + vs[..., None] * dv[None, None]
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3),
                           np.stack([a, c, d], -1).reshape(-1, 3)])
    return pts.reshape(-1, 3), tris


def mesh_m2(n=21):
    """Subdivided box 8x6x4 cm, 6 * 2 * n^2 triangles (n=21 -> 5 292); faces are not welded."""
    hx, hy, hz = 0.04, 0.03, 0.02
    faces = [
        (np.array([-hx, -hy, hz]), np.array([2 * hx, 0, 0]), np.array([0, 2 * hy, 0])),
        (np.array([-hx, -hy, -hz]), np.array([0, 2 * hy, 0]), np.array([2 * hx, 0, 0])),
        (np.array([hx, -hy, -hz]), np.array([0, 2 * hy, 0]), np.array([0, 0, 2 * hz])),
        (np.array([-hx, -hy, -hz]), np.array([0, 0, 2 * hz]), np.array([0, 2 * hy, 0])),
        (np.array([-hx, hy, -hz]), np.array([0, 0, 2 * hz]), np.array([2 * hx, 0, 0])),
        (np.array([-hx, -hy, -hz]), np.array([2 * hx, 0, 0]), np.array([0, 0, 2 * hz])),
    ]
    vs, ts, off = [], [], 0
    for o, du, dv in faces:
        v, t = _grid_patch(o, du, dv, n, n)
        vs.append(v)
        ts.append(t + off)
        off += len(v)
    return np.concatenate(vs), np.concatenate(ts).astype(np.int32)


def mesh_m3(nu=64, nv=40):
    """Torus R=4 cm, r=1.5 cm, 2*nu*nv triangles (64x40 -> 5 120)."""
    R, r = 0.04, 0.015
    u = np.arange(nu) * 2 * np.pi / nu
    w = np.arange(nv) * 2 * np.pi / nv
    U, W = np.meshgrid(u, w, indexing="ij")
    v = np.stack([(R + r * np.cos(W)) * np.cos(U), (R + r * np.cos(W)) * np.sin(U),
                  r * np.sin(W)], -1).reshape(-1, 3)
    idx = np.arange(nu * nv).reshape(nu, nv)
    a, b = idx, np.roll(idx, -1, 0)
    c, d = np.roll(b, -1, 1), np.roll(idx, -1, 1)
    t = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3),
                        np.stack([a, c, d], -1).reshape(-1, 3)])
    return v, t.astype(np.int32)


def mesh_m4(n_lon=160, n_lat=160):
    """UV ellipsoid radii (9,7,5) cm; n_lon*(2*n_lat-2) triangles (160x160 -> 50 880)."""
    lat = np.pi * np.arange(1, n_lat) / n_lat
    lon = 2 * np.pi * np.arange(n_lon) / n_lon
    LA, LO = np.meshgrid(lat, lon, indexing="ij")
    ring = np.stack([np.sin(LA) * np.cos(LO), np.sin(LA) * np.sin(LO), np.cos(LA)], -1)
    v = np.concatenate([[[0, 0, 1.0]], ring.reshape(-1, 3), [[0, 0, -1.0]]])
    idx = 1 + np.arange((n_lat - 1) * n_lon).reshape(n_lat - 1, n_lon)
    nxt = np.roll(idx, -1, 1)
    t = [np.stack([np.zeros(n_lon, int), idx[0], nxt[0]], -1)]
    a, b, c, d = idx[:-1], idx[1:], nxt[1:], nxt[:-1]
    t.append(np.stack([a, b, c], -1).reshape(-1, 3))
    t.append(np.stack([a, c, d], -1).reshape(-1, 3))
    t.append(np.stack([np.full(n_lon, len(v) - 1), nxt[-1], idx[-1]], -1))
    return v * np.array([0.09, 0.07, 0.05]), np.concatenate(t).astype(np.int32)


def mesh_box12():
    """12-triangle box 8x6x4 cm: every triangle is 'big' on screen (cooperative raster path)."""
    return mesh_m2(n=1)


# ---------------------------------------------------------------------------- camera / poses
def camera_matrix(cols=640, rows=480):
    s = cols / 640.0
    return np.array([[570.3 * s, 0.0, (cols - 1) * 0.5], [0.0, 570.3 * s, (rows - 1) * 0.5],
                     [0.0, 0.0, 1.0]])


def truth_pose(n_bodies=1, z=0.7, frame=0):
    """Ground-truth absolute pose [n_bodies,12]: bodies side by side at depth z, translating
    2 mm and rotating 1 degree per frame."""
    R, t = [], []
    for b in range(n_bodies):
        rv = np.array([0.3 + 0.2 * b, -0.5 + 0.3 * b, 0.2]) + np.radians(1.0) * frame * np.array([0, 1.0, 0])
        x = (b - 0.5 * (n_bodies - 1)) * 0.14
        R.append(rotvec_to_matrix(rv))
        t.append(np.array([x + 0.002 * frame, 0.01 * b, z + 0.03 * b]))
    return pack_Rt(np.array(R), np.array(t))


def particle_poses(truth, n, rng, sigma_lin=0.0025, sigma_ang=0.02, scale=1.0):
    """n particles around `truth` [n_bodies,12]: R = R(d_ang) R_truth, t = t_truth + d_lin with
    d ~ N(0, diag(sigma_lin, sigma_ang)) * scale (R:config/particle_tracker.yaml:55-60)."""
    truth = np.asarray(truth).reshape(-1, 12)
    nb = truth.shape[0]
    dl = rng.normal(0.0, sigma_lin * scale, size=(n, nb, 3))
    da = rng.normal(0.0, sigma_ang * scale, size=(n, nb, 3))
    R0 = truth[:, :9].reshape(nb, 3, 3)
    R = rotvec_to_matrix(da) @ R0[None]
    return pack_Rt(R, truth[None, :, 9:12] + dl)


# ---------------------------------------------------------------------------- frames
def make_frame(object_depth, rows, cols, rng, K=None, noise=True, nan_frac=0.05,
               occluder=True, bg_depth=1.5):
    """Synthetic Kinect frame (float32 metres, NaN = no reading) from a rendered object depth
    image (+inf where the object is absent): tilted background plane at ~bg_depth, an occluding
    slab at 0.5 m over the right ~25 % of the object, N(0,(0.003+0.0014247 z^2)^2) noise, i.i.d.
    NaN drop-outs."""
    d = np.asarray(object_depth, dtype=np.float64).reshape(rows, cols)
    cc, rr = np.meshgrid(np.arange(cols), np.arange(rows))
    bg = bg_depth + 0.2 * (cc / cols - 0.5) + 0.1 * (rr / rows - 0.5)
    z = np.where(np.isfinite(d), d, bg)
    if occluder and np.isfinite(d).any():
        ys, xs = np.nonzero(np.isfinite(d))
        x_lo = xs.min() + 0.75 * (xs.max() - xs.min())
        slab = (cc >= x_lo) & (cc <= xs.max() + 10) & (rr >= ys.min() - 10) & (rr <= ys.max() + 10)
        z = np.where(slab, 0.5, z)
    if noise:
        z = z + rng.normal(size=z.shape) * (0.003 + 0.0014247 * z * z)
    z = z.astype(np.float32)
    if nan_frac > 0:
        z[rng.random(z.shape) < nan_frac] = np.nan
    return z.ravel()


def resample_like_indices(n, rng, concentration=None):
    """Parent indices as the filter's multinomial resampling would produce them. With
    concentration=None a random permutation (every child a distinct parent: the worst case
    for HBM traffic, no plane is read twice); otherwise parents drawn from weights
    ~ Dirichlet(concentration)."""
    if concentration is None:
        return rng.permutation(n).astype(np.int32)
    w = rng.dirichlet(np.full(n, concentration))
    return np.sort(rng.choice(n, size=n, p=w)).astype(np.int32)
