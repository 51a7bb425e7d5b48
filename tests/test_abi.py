"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares;
host-only entry points (scene generators, camera) behave; device entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _abi, scenes

from conftest import ROOT, _has_gpu

HEADER = os.path.join(ROOT, "include", "raytracer_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_abi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/raytracer_amd.h but not exported"
    bound = {s[0] for s in _abi.SYMBOLS}
    assert bound == set(names), f"ctypes binding out of sync with header: {bound ^ set(names)}"


def test_no_torch_types_in_the_abi():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)      # declarations only, comments stripped
    assert "torch" not in src and "at::" not in src and "std::" not in src and "hipStream_t" not in src


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", "/nonexistent/libraytracer_amd.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _abi.load()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(d, f), errors="ignore").read()
                assert "liboracle" not in text and "import oracle" not in text and "oracle/" not in text.replace("the CPU oracle", ""), f


def test_struct_layouts():
    assert rt.TRIANGLE.itemsize == 144 and rt.SPHERE.itemsize == 64 and rt.LIGHT.itemsize == 32
    assert rt.CAMERA.itemsize == 112 and rt.MATERIAL.itemsize == 48 and rt.VERTEX.itemsize == 32
    assert rt.TRIANGLE.fields["material"][1] == 96 and rt.TRIANGLE.fields["v2"][1] == 64
    assert rt.CAMERA.fields["fov"][1] == 92 and rt.CAMERA.fields["farPlane"][1] == 104
    assert rt.MATERIAL.fields["emissionStrength"][1] == 28 and rt.SPHERE.fields["radius"][1] == 12


def test_cornell_scene_values():
    """Value-for-value against src/vulkan/VkRenderer.cpp:171-250 (spot checks + totals)."""
    tris, sph, lights = scenes.cornell()
    assert len(tris) == 12 and len(sph) == 3 and len(lights) == 0
    assert tuple(tris[0]["v0"]["position"]) == (-2, 2, 0) and tuple(tris[0]["v0"]["normal"]) == (0, 0, 1)
    assert tuple(tris[3]["v2"]["position"]) == (-2, -2, 2)
    assert tuple(tris[4]["material"]["albedo"]) == (1, 0, 0) and tuple(tris[6]["material"]["albedo"]) == (0, 1, 0)
    assert tris[10]["v0"]["position"][2] == np.float32(1.99) and tris[11]["material"]["emissionStrength"] == 0.5
    assert tuple(tris[10]["material"]["emission"]) == (1, 1, 1)
    assert [tuple(s["center"]) for s in sph] == [(-1, 0, np.float32(0.2)), (0, 0, np.float32(0.2)), (1, 0, np.float32(0.2))]
    assert all(s["radius"] == np.float32(0.2) for s in sph)
    assert tuple(sph[0]["material"]["albedo"]) == (1, np.float32(0.9), 0) and sph[0]["material"]["metallic"] == 1
    assert sph[2]["material"]["roughness"] == 1 and sph[1]["material"]["roughness"] == np.float32(0.1)
    pos = np.stack([tris[k]["position"] for k in ("v0", "v1", "v2")])
    assert pos.min() == -2 and pos.max() == 2


def test_sphere_geometry_counts_and_shape():
    """Sphere::sphereGeometry (src/math/Sphere.h:17-79): 2*stacks*slices triangles on the sphere."""
    t = scenes.sphere_geometry((1, 2, 3), 0.5, 5, 7)
    assert len(t) == 70
    for k in ("v0", "v1", "v2"):
        d = t[k]["position"] - np.array([1, 2, 3], dtype=np.float32)
        assert np.allclose(np.linalg.norm(d, axis=1), 0.5, atol=1e-6)
        assert np.allclose(t[k]["normal"], d / 0.5, atol=1e-5)
    assert np.array_equal(t[0]["v0"]["position"], t[1]["v0"]["position"])      # both triangles of a quad start at v[i][j]
    t32, s32, _ = scenes.cornell32()
    assert len(t32) == 32 and len(s32) == 2
    assert len(scenes.bunny_class()[0]) == 69194 and len(scenes.sponza_class()[0]) == 259212


def test_soup_and_rays_are_counter_based():
    a = scenes.soup(1000)
    b = scenes.soup(500)
    assert a[:500].tobytes() == b.tobytes()
    p = np.stack([a[k]["position"] for k in ("v0", "v1", "v2")])
    assert p.min() > -0.006 and p.max() < 1.006
    r = scenes.random_rays(1000)
    r2 = scenes.random_rays(300, first=700)
    assert r[700:].tobytes() == r2.tobytes()
    assert np.allclose(np.linalg.norm(r["d"], axis=1), 1, atol=1e-6) and r["o"].min() >= 0 and r["o"].max() < 1
    assert np.all(r["tmin"] == np.float32(1e-6)) and np.all(r["tmax"] == np.float32(1e20))
    assert abs(r["d"].mean()) < 0.05                                           # uniform on the sphere


def test_camera_basis_follows_update_camera_vectors():
    """Camera::updateCameraVectors (src/application/Camera.h:36-41) with worldUp forced to (0,0,-1)."""
    c = rt.camera_look_at((3, -2, 5), (0.5, 1, 0), 60.0, 1.5)[0]
    f = np.array([0.5 - 3, 1 + 2, 0 - 5], dtype=np.float64)
    f /= np.linalg.norm(f)
    r = np.cross(f, (0, 0, -1)); r /= np.linalg.norm(r)
    u = np.cross(r, f); u /= np.linalg.norm(u)
    assert np.allclose(c["front"], f, atol=1e-6) and np.allclose(c["right"], r, atol=1e-6) and np.allclose(c["up"], u, atol=1e-6)
    assert c["fov"] == 60 and c["aspectRatio"] == 1.5 and tuple(c["worldUp"]) == (0, 0, -1)


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU error path")
def test_create_without_gpu_raises():
    with pytest.raises(RuntimeError, match="rt_create failed"):
        rt.Renderer((0,))
