"""the minimal OBJ reader (rt_load_obj) -- host-only, no GPU.  The reference has no loader, so the
expectations are the OBJ format's own rules (fan triangulation, 1-based / negative indices, optional normals)."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import scenes

CUBE = """# unit cube, quads, no normals
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vt 0.5 0.5
g cube
usemtl whatever
f 1 2 3 4
f 5/1 6/1 7/1 8/1
f -8 -7 -3 -4
"""

WITH_NORMALS = """v 0 0 0
v 2 0 0
v 0 2 0
vn 0 0 1
vn 0 1 0
f 1//1 2//1 3//2
f 1/7/1 2/7/1 3/7/1
"""


def write(tmp_path, text, name="m.obj"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_quads_fans_negative_indices_and_geometric_normals(tmp_path):
    tris = rt.load_obj(write(tmp_path, CUBE))
    assert tris.dtype == rt.TRIANGLE and len(tris) == 6
    # f 1 2 3 4 -> (1,2,3), (1,3,4)
    assert np.array_equal(tris[0]["v0"]["position"], [0, 0, 0]) and np.array_equal(tris[0]["v1"]["position"], [1, 0, 0])
    assert np.array_equal(tris[0]["v2"]["position"], [1, 1, 0])
    assert np.array_equal(tris[1]["v1"]["position"], [1, 1, 0]) and np.array_equal(tris[1]["v2"]["position"], [0, 1, 0])
    for k in (0, 1):
        for v in ("v0", "v1", "v2"):
            assert np.array_equal(tris[k][v]["normal"], [0, 0, 1])           # cross((1,0,0),(1,1,0)) normalised
    # f -8 -7 -3 -4 = vertices 1 2 6 5
    assert np.array_equal(tris[4]["v0"]["position"], [0, 0, 0]) and np.array_equal(tris[4]["v2"]["position"], [1, 0, 1])
    assert np.array_equal(tris[4]["v0"]["normal"], [0, -1, 0])
    # default material
    assert np.allclose(tris["material"]["albedo"], 0.8) and np.all(tris["material"]["roughness"] == 0.5)


def test_vertex_normals_and_material(tmp_path):
    mat = np.zeros(1, dtype=rt.MATERIAL)
    mat["albedo"] = (0.1, 0.2, 0.3); mat["emission"] = (1, 1, 1); mat["emissionStrength"] = 4.0; mat["metallic"] = 1.0
    tris = rt.load_obj(write(tmp_path, WITH_NORMALS), mat)
    assert len(tris) == 2
    assert np.array_equal(tris[0]["v0"]["normal"], [0, 0, 1]) and np.array_equal(tris[0]["v2"]["normal"], [0, 1, 0])
    assert np.array_equal(tris[1]["v2"]["normal"], [0, 0, 1])
    assert tris["material"].tobytes() == mat.tobytes() * 2


def test_round_trip_of_a_generated_mesh(tmp_path):
    """Write the sphereGeometry stand-in as OBJ text, read it back: positions and normals survive exactly (%.9g)."""
    src = scenes.sphere_geometry((0.5, 0.25, -1.0), 0.75, 6, 9)
    lines = []
    for t in src:
        for v in ("v0", "v1", "v2"):
            lines.append("v %.9g %.9g %.9g" % tuple(t[v]["position"]))
            lines.append("vn %.9g %.9g %.9g" % tuple(t[v]["normal"]))
    for k in range(len(src)):
        a = 3 * k + 1
        lines.append(f"f {a}//{a} {a+1}//{a+1} {a+2}//{a+2}")
    got = rt.load_obj(write(tmp_path, "\n".join(lines) + "\n"), src[:1]["material"])
    assert len(got) == len(src)
    for v in ("v0", "v1", "v2"):
        assert got[v]["position"].tobytes() == src[v]["position"].tobytes()
        assert got[v]["normal"].tobytes() == src[v]["normal"].tobytes()


@pytest.mark.parametrize("text", ["v 0 0\n", "v 0 0 0\nf 1 2 3\n", "v 0 0 0\nv 1 0 0\nf 1 2\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2//1 3//1\n"])
def test_malformed_files_are_rejected(tmp_path, text):
    with pytest.raises(RuntimeError):
        rt.load_obj(write(tmp_path, text))


def test_missing_file_and_empty_file(tmp_path):
    with pytest.raises(RuntimeError):
        rt.load_obj(str(tmp_path / "nope.obj"))
    assert len(rt.load_obj(write(tmp_path, "# nothing\n"))) == 0


def test_real_mesh_hook_of_the_scene_setup(tmp_path, monkeypatch):
    """C2/C3 use the procedural stand-ins unless RT_BUNNY_OBJ / RT_SPONZA_OBJ name a mesh."""
    assert len(scenes.bunny_class()[0]) == 69194
    monkeypatch.setenv("RT_BUNNY_OBJ", write(tmp_path, CUBE, "bunny.obj"))
    tris, sph, lights, cam = scenes.bunny_class()
    assert len(tris) == 6 and len(sph) == 0 and len(lights) == 1 and cam.dtype == rt.CAMERA
    monkeypatch.setenv("RT_BUNNY_OBJ", str(tmp_path / "absent.obj"))
    assert len(scenes.bunny_class()[0]) == 69194
