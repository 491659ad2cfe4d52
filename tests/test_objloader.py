"""Wavefront .obj loading."""
import numpy as np
import pytest

from dbot_ros_amd import ObjectModel, objloader, synth

OBJ = """
# a unit quad + a triangle, mixed reference styles
mtllib x.mtl
o thing
v 0 0 0
v 1 0 0
v 1 1 0     # trailing comment
v 0 1 0 1.0
vt 0 0
vn 0 0 1
g grp
usemtl m
s off
f 1/1/1 2/1/1 3/1/1 4/1/1
v 2 0 0.5
f -1 -3//1 -4/1
"""


def test_parse_polygons_and_reference_styles():
    v, t = objloader.parse_obj(OBJ)
    assert v.shape == (5, 3) and v.dtype == np.float64 and t.dtype == np.int32
    assert t.tolist() == [[0, 1, 2], [0, 2, 3], [4, 2, 1]]
    assert np.allclose(v[4], [2, 0, 0.5])


@pytest.mark.parametrize("bad", ["v 0 0\nf 1 2 3", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9", "v 0 0 0\nf 1 1",
                                 "v 0 0 0\nv 1 0 0\nv 0 1 0", "f 1 2 3"])
def test_bad_meshes_raise(bad):
    with pytest.raises(ValueError):
        objloader.parse_obj(bad)


def test_roundtrip_of_synthetic_meshes_through_files(tmp_path):
    """The synthetic meshes written as .obj and loaded through the ORI/loader pair give back the
    same geometry exactly (repr round-trips doubles), and build the same ObjectModel."""
    names = []
    for name, fn in (("m1.obj", synth.mesh_m1), ("m3.obj", synth.mesh_m3)):
        v, t = fn()
        objloader.write_obj(tmp_path / name, v, t)
        names.append(name)
    ori = objloader.ObjectResourceIdentifier(str(tmp_path), "", names)
    assert ori.count_meshes() == 2
    vs, ts = objloader.SimpleWavefrontObjectModelLoader(ori).load()
    for (v, t), fn in zip(zip(vs, ts), (synth.mesh_m1, synth.mesh_m3)):
        v0, t0 = fn()
        assert np.array_equal(v, v0) and np.array_equal(t, t0)
    om = ObjectModel(vs, ts, center=True)
    assert om.count_parts == 2 and np.allclose(om.vertices[0].mean(0), 0, atol=1e-15)
