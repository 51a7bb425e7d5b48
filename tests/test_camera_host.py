"""the reference's camera mutators (src/application/Camera.h:56-90) as restated in
raytracer_amd/host/rt_camera.hpp and exported as rt_camera_move / rt_camera_rotate.  Expected values are derived by hand
from the reference's formulas (normalize, cross, Rodrigues rotation = glm::rotate applied to a direction); glm itself is
an un-vendored submodule, so its rounding is not pinned: tolerances are a few fp32 ulps of the coordinates."""
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = math.sqrt(0.5)


def vec(c, name):
    return np.asarray(c[name][0], dtype=np.float64)


def close(a, b, tol=2e-6):
    return np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), atol=tol, rtol=0)


def test_default_camera_basis():
    c = rt.default_camera()                      # VkRenderer.cpp:3-12: position (0,4,1), lookAt (0,0,1), worldUp (0,0,-1)
    assert close(vec(c, "front"), (0, -1, 0)) and close(vec(c, "right"), (1, 0, 0)) and close(vec(c, "up"), (0, 0, -1))


def test_moves_translate_position_and_look_at_together():
    c = rt.default_camera()
    f = rt.camera_move(c, forward=1.5)           # front = (0,-1,0)
    assert close(vec(f, "position"), (0, 2.5, 1)) and close(vec(f, "lookAt"), (0, -1.5, 1))
    r = rt.camera_move(c, right=2.0)             # right = (1,0,0)
    assert close(vec(r, "position"), (2, 4, 1)) and close(vec(r, "lookAt"), (2, 0, 1))
    u = rt.camera_move(c, up=0.5)                # up = (0,0,-1): "up" on screen is -z in the world (Vulkan y-down)
    assert close(vec(u, "position"), (0, 4, 0.5)) and close(vec(u, "lookAt"), (0, 0, 0.5))
    for m in (f, r, u):                          # a pure translation keeps the basis
        assert close(vec(m, "front"), (0, -1, 0)) and close(vec(m, "right"), (1, 0, 0)) and close(vec(m, "up"), (0, 0, -1))
    all3 = rt.camera_move(c, 1.5, 2.0, 0.5)      # order forward, right, up; the basis does not change in between
    assert close(vec(all3, "position"), (2, 2.5, 0.5))
    same = rt.camera_move(c)                     # zero arguments: every field untouched
    assert all(np.array_equal(same[n], np.ascontiguousarray(c)[n]) for n in rt.CAMERA.names)


def test_rotate_around_world_up_by_90_degrees():
    # direction (0,-4,0) about k = (0,0,-1): k x v = (-4,0,0), k.v = 0 => v' = v cos 90 + (k x v) sin 90 = (-4, 0, 0)
    c = rt.camera_rotate(rt.default_camera(), around_up_deg=90.0)
    assert close(vec(c, "position"), (0, 4, 1)) and close(vec(c, "lookAt"), (-4, 4, 1), 1e-6 * 4)
    assert close(vec(c, "front"), (-1, 0, 0)) and close(vec(c, "right"), (0, -1, 0)) and close(vec(c, "up"), (0, 0, -1))


def test_rotate_around_right_by_45_degrees():
    # direction (0,-4,0) about k = (1,0,0): k x v = (0,0,-4) => v' = (0, -4 cos 45, -4 sin 45)
    c = rt.camera_rotate(rt.default_camera(), around_right_deg=45.0)
    assert close(vec(c, "lookAt"), (0, 4 - 4 * S, 1 - 4 * S), 4e-6)
    assert close(vec(c, "front"), (0, -S, -S)) and close(vec(c, "right"), (1, 0, 0)) and close(vec(c, "up"), (0, S, -S))


def test_rotations_keep_distance_and_orthonormal_basis_and_close_after_a_full_turn():
    c = rt.default_camera()
    for k in range(36):
        c = rt.camera_rotate(c, around_up_deg=10.0, around_right_deg=(3.0 if k % 2 else -3.0))
        f, r, u = vec(c, "front"), vec(c, "right"), vec(c, "up")
        assert abs(np.linalg.norm(vec(c, "lookAt") - vec(c, "position")) - 4.0) < 1e-4
        for a in (f, r, u):
            assert abs(np.linalg.norm(a) - 1.0) < 1e-6
        assert abs(f @ r) < 1e-6 and abs(f @ u) < 1e-6 and abs(r @ u) < 1e-6
        assert close(np.cross(r, f), u, 1e-6)            # up = right x front (Camera.h:40)
    full = rt.default_camera()
    for _ in range(36):
        full = rt.camera_rotate(full, around_up_deg=10.0)
    assert close(vec(full, "lookAt"), (0, 0, 1), 2e-5) and close(vec(full, "front"), (0, -1, 0), 1e-5)
    assert float(full["fov"][0]) == 45.0 and float(full["farPlane"][0]) == 100.0     # scalars ride along untouched


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_camera_class_matches_the_c_abi(tmp_path):
    """The C++ class (same member names as the reference's Camera) and the C entry points are one implementation."""
    src = tmp_path / "cam.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstring>
#include "rt_camera.hpp"
#include "raytracer_amd.h"
int main() {
    rt::Camera cam({0.f, 4.f, 1.f}, {0.f, 0.f, 1.f}, {0.f, 0.f, -1.f}, 45.f, 1.2f, 0.1f, 100.f);
    rt::CameraUBO ref;
    const float p[3] = {0.f, 4.f, 1.f}, l[3] = {0.f, 0.f, 1.f};
    rt_camera_look_at(p, l, 45.f, 1.2f, &ref);
    rt::CameraUBO got; cam.updateCameraUBO(got, 0.016f);
    got.m_lookAt = cam.m_cameraUBO.m_lookAt;
    if (std::memcmp(&got.m_front, &ref.m_front, 12) || std::memcmp(&got.m_right, &ref.m_right, 12)) return 1;
    cam.moveForward(0.25f); cam.moveRight(-1.f); cam.moveUp(2.f); cam.rotateAroundUp(33.f); cam.rotateAroundRight(-7.f);
    rt_camera_move(&ref, 0.25f, 0.f, 0.f); rt_camera_move(&ref, 0.f, -1.f, 0.f); rt_camera_move(&ref, 0.f, 0.f, 2.f);
    rt_camera_rotate(&ref, 33.f, 0.f); rt_camera_rotate(&ref, 0.f, -7.f);
    cam.updateCameraUBO(got, 0.f);
    if (std::memcmp(&got.m_position, &ref.m_position, 12) || std::memcmp(&got.m_lookAt, &ref.m_lookAt, 12) ||
        std::memcmp(&got.m_front, &ref.m_front, 12) || std::memcmp(&got.m_up, &ref.m_up, 12) || std::memcmp(&got.m_right, &ref.m_right, 12)) return 2;
    std::printf("camera ok %g %g %g\n", got.m_front.x, got.m_front.y, got.m_front.z);
    return 0;
}
''')
    exe = tmp_path / "cam"
    csrc = os.path.join(ROOT, "raytracer_amd", "csrc")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "raytracer_amd", "host"),
                        str(src), "-o", str(exe), "-L", csrc, "-lraytracer_amd", f"-Wl,-rpath,{csrc}"], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "camera ok" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])
