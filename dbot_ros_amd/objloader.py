"""Wavefront .obj mesh loading -- the role of dbot's
SimpleWavefrontObjectModelLoader / ObjectResourceIdentifier as the node uses them
(R:source/dbot_ros/tracker/particle_tracker_node.cpp:84-97; mesh list R:config/object.yaml:3-5):
one .obj per tracked part -> vertices [nv,3] float64 + triangle indices [nt,3] int32.

Supported: `v x y z [w]`, `f` with v, v/vt, v//vn, v/vt/vn references, negative (relative)
indices, polygons (fan-triangulated); everything else (vt, vn, g, o, s, usemtl, mtllib, comments)
is ignored, as a geometry-only tracker needs."""
import os

import numpy as np


class ObjectResourceIdentifier:
    """package path + directory + mesh file names (R:...particle_tracker_node.cpp:84-87)."""

    def __init__(self, package_path="", directory="", meshes=()):
        self.package_path, self.directory, self.meshes = package_path, directory, list(meshes)

    def count_meshes(self):
        return len(self.meshes)

    def mesh_path(self, i):
        return os.path.join(self.package_path, self.directory, self.meshes[i])


def parse_obj(text):
    """-> (vertices [nv,3] float64, triangles [nt,3] int32). Raises ValueError on bad geometry."""
    verts, tris = [], []
    for ln, line in enumerate(text.splitlines(), 1):
        line = line.split("#", 1)[0].strip()
        if not line:
            continue
        tok = line.split()
        if tok[0] == "v":
            if len(tok) < 4:
                raise ValueError(f"line {ln}: vertex needs 3 coordinates")
            verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
        elif tok[0] == "f":
            idx = []
            for ref in tok[1:]:
                v = int(ref.split("/")[0])
                v = v - 1 if v > 0 else len(verts) + v   # negative = relative to the vertices so far
                if v < 0 or v >= len(verts):
                    raise ValueError(f"line {ln}: vertex reference {ref} out of range")
                idx.append(v)
            if len(idx) < 3:
                raise ValueError(f"line {ln}: face needs at least 3 vertices")
            for k in range(1, len(idx) - 1):              # fan triangulation
                tris.append((idx[0], idx[k], idx[k + 1]))
    if not verts or not tris:
        raise ValueError("mesh has no vertices or no faces")
    return np.array(verts, dtype=np.float64), np.array(tris, dtype=np.int32)


def load_obj(path):
    with open(path, "r") as f:
        return parse_obj(f.read())


class SimpleWavefrontObjectModelLoader:
    """Loads every mesh an ObjectResourceIdentifier names."""

    def __init__(self, ori):
        self.ori = ori

    def load(self):
        vs, ts = [], []
        for i in range(self.ori.count_meshes()):
            v, t = load_obj(self.ori.mesh_path(i))
            vs.append(v)
            ts.append(t)
        return vs, ts


def write_obj(path, vertices, triangles):
    """Writes a mesh back out (used to materialise the synthetic meshes as .obj files)."""
    with open(path, "w") as f:
        f.write("# written by dbot_ros_amd.objloader.write_obj\n")
        for v in np.asarray(vertices, dtype=np.float64):
            f.write(f"v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\n")
        for t in np.asarray(triangles):
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
