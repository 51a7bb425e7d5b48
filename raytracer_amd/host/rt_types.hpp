// rt_types.hpp -- host-side scene/camera structs, byte-compatible with the reference's.
//
// The reference keeps std140-compatible PODs in src/math/ and src/application/Camera.h and memcpy's
// them into its SSBOs/UBO (src/vulkan/VkRenderer.cpp:260-326, :855-862).  These twins have the same
// names, member names and byte layout, with a build-owned Vec3 instead of glm::vec3 (glm is an
// un-vendored submodule).  A host that already holds the reference's arrays can pass them to
// rt_scene_upload / rt_set_camera unchanged.
#pragma once
#include <cstddef>
#include <cstdint>

namespace rt {

struct Vec3 {
    float x, y, z;
};

// src/math/Vertex.h:38-41
struct Vertex3D {
    alignas(16) Vec3 m_position;
    alignas(16) Vec3 m_normal;
};

// src/math/Material.h:6-11
struct Material {
    alignas(16) Vec3 m_albedo;
    alignas(16) Vec3 m_emission;
    alignas(4) float m_emissionStrength;
    alignas(4) float m_roughness;
    alignas(4) float m_metallic;
};

// src/math/Triangle.h:10-15
struct Triangle {
    Vertex3D m_v0, m_v1, m_v2;
    Material m_material;
};

// src/math/Sphere.h:10-15
struct Sphere {
    alignas(16) Vec3 m_center;
    alignas(4) float m_radius;
    Material m_material;
};

// src/math/Light.h:6-10
struct Light {
    alignas(16) Vec3 m_position;
    alignas(16) Vec3 m_color;
    alignas(4) float m_intensity;
};

// src/application/Camera.h:6-18
struct CameraUBO {
    alignas(16) Vec3 m_position;
    alignas(16) Vec3 m_lookAt;
    alignas(16) Vec3 m_front;
    alignas(16) Vec3 m_up;
    alignas(16) Vec3 m_right;
    alignas(16) Vec3 m_worldUp;
    alignas(4) float m_fov;
    alignas(4) float m_aspectRatio;
    alignas(4) float m_nearPlane;
    alignas(4) float m_farPlane;
};

// Sizes and offsets probed from the reference's declarations.
static_assert(sizeof(Vertex3D) == 32 && offsetof(Vertex3D, m_normal) == 16, "Vertex3D layout");
static_assert(sizeof(Material) == 48 && offsetof(Material, m_emission) == 16 &&
              offsetof(Material, m_emissionStrength) == 28 && offsetof(Material, m_roughness) == 32 &&
              offsetof(Material, m_metallic) == 36, "Material layout");
static_assert(sizeof(Triangle) == 144 && offsetof(Triangle, m_v1) == 32 && offsetof(Triangle, m_v2) == 64 &&
              offsetof(Triangle, m_material) == 96, "Triangle layout");
static_assert(sizeof(Sphere) == 64 && offsetof(Sphere, m_radius) == 12 && offsetof(Sphere, m_material) == 16,
              "Sphere layout");
static_assert(sizeof(Light) == 32 && offsetof(Light, m_color) == 16 && offsetof(Light, m_intensity) == 28,
              "Light layout");
static_assert(sizeof(CameraUBO) == 112 && offsetof(CameraUBO, m_front) == 32 && offsetof(CameraUBO, m_up) == 48 &&
              offsetof(CameraUBO, m_right) == 64 && offsetof(CameraUBO, m_worldUp) == 80 &&
              offsetof(CameraUBO, m_fov) == 92 && offsetof(CameraUBO, m_aspectRatio) == 96 &&
              offsetof(CameraUBO, m_nearPlane) == 100 && offsetof(CameraUBO, m_farPlane) == 104, "CameraUBO layout");

}  // namespace rt
