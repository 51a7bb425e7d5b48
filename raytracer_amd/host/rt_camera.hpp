// rt_camera.hpp -- the interactive camera of the reference (src/application/Camera.h:4-92) for hosts of the C ABI.
//
// Same struct, member and method names as the reference's `Camera`, so that the GLFW callbacks that drive it there
// (src/application/Application.h:53-174: W/A/S/D/Q/E keys -> moveForward / moveRight / moveUp, mouse drag ->
// rotateAroundUp / rotateAroundRight) compile unchanged against this header; the 112-byte m_cameraUBO goes to
// rt_set_camera as it is.  glm is an un-vendored submodule of the reference, so the vector math is
// restated here: normalize = v / sqrt(dot(v, v)), the textbook cross product, and glm::rotate(mat4(1), angle, axis)
// applied to a direction = Rodrigues' rotation about the normalised axis, written out as the same 3x3 matrix glm
// builds (c + (1-c) a_i a_i on the diagonal, (1-c) a_i a_j +- s a_k off it).  No test of the reference pins glm's
// rounding: parity of the camera basis is unpinned at that boundary (harmless -- the basis is an INPUT that the kernels
// and the oracle receive identically).
#pragma once
#include <cmath>

#include "rt_types.hpp"

namespace rt {

namespace cam_detail {
inline Vec3 add(const Vec3& a, const Vec3& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec3 sub(const Vec3& a, const Vec3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 mul(const Vec3& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3& a, const Vec3& b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline Vec3 normalize(const Vec3& v) { const float l = std::sqrt(dot(v, v)); return { v.x / l, v.y / l, v.z / l }; }
// glm::rotate(mat4(1), radians, axis) * vec4(v, 0): rotation of a direction about `axis` (normalised first, as glm does)
inline Vec3 rotate(const Vec3& v, float radians, const Vec3& axis_in) {
    const float c = std::cos(radians), s = std::sin(radians);
    const Vec3 a = normalize(axis_in);
    const Vec3 t = mul(a, 1.0f - c);
    // columns of the rotation matrix
    const Vec3 c0 = { c + t.x * a.x,         t.x * a.y + s * a.z,   t.x * a.z - s * a.y };
    const Vec3 c1 = { t.y * a.x - s * a.z,   c + t.y * a.y,         t.y * a.z + s * a.x };
    const Vec3 c2 = { t.z * a.x + s * a.y,   t.z * a.y - s * a.x,   c + t.z * a.z };
    return { c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z, c0.z * v.x + c1.z * v.y + c2.z * v.z };
}
inline float radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }
}  // namespace cam_detail

struct Camera {
    using UniformBufferObject = CameraUBO;            // src/application/Camera.h:6-18, 112 bytes
    UniformBufferObject m_cameraUBO{};

    Camera() {}

    // Camera.h:24-34 (the `up` argument is overwritten by updateCameraVectors there too; worldUp is forced to (0,0,-1))
    Camera(Vec3 position, Vec3 lookAt, Vec3 up, float fov, float aspectRatio, float nearPlane, float farPlane) {
        m_cameraUBO.m_position = position;
        m_cameraUBO.m_lookAt = lookAt;
        m_cameraUBO.m_up = up;
        m_cameraUBO.m_fov = fov;
        m_cameraUBO.m_aspectRatio = aspectRatio;
        m_cameraUBO.m_nearPlane = nearPlane;
        m_cameraUBO.m_farPlane = farPlane;
        m_cameraUBO.m_worldUp = { 0.0f, 0.0f, -1.0f };
        updateCameraVectors();
    }

    // Camera.h:36-41
    inline void updateCameraVectors() {
        using namespace cam_detail;
        m_cameraUBO.m_front = normalize(sub(m_cameraUBO.m_lookAt, m_cameraUBO.m_position));
        m_cameraUBO.m_right = normalize(cross(m_cameraUBO.m_front, m_cameraUBO.m_worldUp));
        m_cameraUBO.m_up = normalize(cross(m_cameraUBO.m_right, m_cameraUBO.m_front));
    }

    // Camera.h:43-54 (deltaTime is unused there as well)
    inline void updateCameraUBO(UniformBufferObject& ubo, float /*deltaTime*/) const {
        ubo.m_position = m_cameraUBO.m_position;
        ubo.m_lookAt = m_cameraUBO.m_lookAt;
        ubo.m_front = m_cameraUBO.m_front;
        ubo.m_up = m_cameraUBO.m_up;
        ubo.m_right = m_cameraUBO.m_right;
        ubo.m_worldUp = { 0.0f, 0.0f, -1.0f };
        ubo.m_aspectRatio = m_cameraUBO.m_aspectRatio;
        ubo.m_fov = m_cameraUBO.m_fov;
        ubo.m_nearPlane = m_cameraUBO.m_nearPlane;
        ubo.m_farPlane = m_cameraUBO.m_farPlane;
    }

    // Camera.h:56-75: translate position and look-at point together along one basis vector
    inline void moveForward(float distance) { move(m_cameraUBO.m_front, distance); }
    inline void moveRight(float distance) { move(m_cameraUBO.m_right, distance); }
    inline void moveUp(float distance) { move(m_cameraUBO.m_up, distance); }

    // Camera.h:77-91: turn the view direction (angle in degrees) about worldUp / about the camera's right vector
    inline void rotateAroundUp(float angle) { turn(angle, m_cameraUBO.m_worldUp); }
    inline void rotateAroundRight(float angle) { turn(angle, m_cameraUBO.m_right); }

private:
    inline void move(const Vec3& basis, float distance) {
        using namespace cam_detail;
        const Vec3 step = mul(normalize(basis), distance);
        m_cameraUBO.m_position = add(m_cameraUBO.m_position, step);
        m_cameraUBO.m_lookAt = add(m_cameraUBO.m_lookAt, step);
        updateCameraVectors();
    }
    inline void turn(float angle, const Vec3& axis) {
        using namespace cam_detail;
        const Vec3 direction = rotate(sub(m_cameraUBO.m_lookAt, m_cameraUBO.m_position), radians(angle), axis);
        m_cameraUBO.m_lookAt = add(m_cameraUBO.m_position, direction);
        updateCameraVectors();
    }
};

}  // namespace rt
